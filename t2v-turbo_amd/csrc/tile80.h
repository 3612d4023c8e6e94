// Shared pieces of the kernels built on 80 x 80 wave tiles of v_mfma_f32_16x16x32_bf16 (conv_halo.hip, gemm2.hip):
// the LDS-DMA helpers, the counted waits and the workgroup-level epilogue.
#pragma once
#include "common.h"
#include <type_traits>

namespace tile80 {

constexpr unsigned kOutOfRange = 0x80000000u;   // a buffer offset no descriptor of ours covers: the DMA delivers zeros

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ bool static_for_until(F&& f) {   // f(integral_constant<int, I>) -> true: stop
    if constexpr (I < N) {
        if (f(std::integral_constant<int, I>{})) return true;
        return static_for_until<I + 1, N>(f);
    } else {
        return false;
    }
}

// LDS the epilogue needs for a BM x BN tile (the caller's dynamic allocation must cover it): the bf16 tile, or — when a residual /
// late activation makes the row pass work in fp32 — one PHASE of the tile in fp32; plus the tile's global-row table.
template <int BM, int BN>
struct Epi {
    static constexpr int P16 = BN * 2 + 16;                       // bf16 row pitch (bytes)
    static constexpr int P32 = BN * 4 + 16;                       // fp32 row pitch
    static constexpr int PH = (BM * P32 > 112 * 1024) ? 2 : 1;    // fp32 phases (halves of the tile's rows)
    static constexpr int ROWS_PH = BM / PH;
    static_assert(ROWS_PH % 32 == 0, "a phase holds whole 32-row statistics slabs");
    static constexpr int STAGE_BYTES = (BM * P16 > ROWS_PH * P32) ? BM * P16 : ROWS_PH * P32;
    static constexpr int BYTES = STAGE_BYTES + BM * 4;
    static constexpr int CPR = BN / 8;                            // 16-byte output chunks per tile row
};

// Workgroup-level epilogue of a BM x BN tile whose accumulators sit in 80 x 16 NB (tokens x channels) wave tiles (acc[channel block]
// [token block], NB = 5 or 4 channel blocks, a lane owns 4 consecutive channels of one token per block).  `owner`: this wave's accumulators are the final ones (k-group 0).
//   row_of(bm)   -> tile row (0 .. BM) of this lane's token in block bm
//   grow_of(r)   -> global output row of tile row r, < 0 outside the problem
//   col0         =  wave_n * 16 NB + 4 * lq: this lane's first channel within the tile (block bn adds 16 bn)
// Values: v = acc (bias / row vector already inside) [+ residual] [-> SiLU] -> bf16.  Without a residual the tile is rounded
// once and parked as bf16.  With one, the accumulators are parked in FP32 (one phase = half of a large tile at a time), and the
// row pass adds the residual — read in full 16-byte row chunks, not in the accumulator layout's 8-byte pieces — before the ONE
// rounding; the statistics are taken from the final bf16 values, written back in place.
template <int BM, int BN, int NB, class RowOf, class GRowOf>
__device__ __forceinline__ void epilogue(char* smem, const t2v_gemm_desc& d, f32x4_t (&acc)[NB][5], bool owner, int col0, int tid, int n0,
                                         RowOf row_of, GRowOf grow_of, bool skip_stores) {
    using E = Epi<BM, BN>;
    constexpr int P16 = E::P16, P32 = E::P32, CPR = E::CPR;
    int* row_tab = (int*)(smem + E::STAGE_BYTES);
    if (tid < BM) row_tab[tid] = grow_of(tid);
    bf16_t* obase = (bf16_t*)d.out;
    const bool f32_path = d.residual != nullptr;
    if (!f32_path) {
        if (owner) {
#pragma unroll
            for (int bm = 0; bm < 5; ++bm) {
                char* st = smem + row_of(bm) * P16 + col0 * 2;
#pragma unroll
                for (int bn = 0; bn < NB; ++bn) {
                    f32x4_t v = acc[bn][bm];
                    if (d.act == T2V_ACT_SILU) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
                    *(uint2*)(st + bn * 32) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int NIT = (BM * CPR + 511) / 512;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 512 + tid;
            if (idx >= BM * CPR) break;
            const int r = idx / CPR, c = idx - r * CPR;
            const int gm = row_tab[r], ch = n0 + c * 8;
            if (gm >= 0 && ch < d.N && !skip_stores) *(uint4*)(obase + (long long)gm * d.ldo + ch) = *(const uint4*)(smem + r * P16 + c * 16);
        }
        if (d.colstat_out) {
            constexpr int NSL = BM / 32, NCP = BN / 2;
            for (int it = tid; it < NSL * NCP; it += 512) {
                const int sl = it / NCP, cp = it - sl * NCP;
                const int gm0 = row_tab[sl * 32], col = n0 + 2 * cp;
                if (gm0 < 0 || col >= d.N) continue;
                float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) {
                    const uint32_t w2 = *(const uint32_t*)(smem + (sl * 32 + r) * P16 + cp * 4);
                    const float x0 = __uint_as_float(w2 << 16), x1 = __uint_as_float(w2 & 0xffff0000u);
                    a0 += x0; q0 = fmaf(x0, x0, q0);
                    a1 += x1; q1 = fmaf(x1, x1, q1);
                }
                *(float4*)(d.colstat_out + ((long long)(gm0 >> 5) * d.N + col) * 2) = make_float4(a0, q0, a1, q1);   // (host-checked: a tile-order slab is 32 consecutive global rows)
            }
        }
        return;
    }
    // ---- fp32 path: residual (+ activation) added before the one rounding, phase by phase ------------------------------------------
    constexpr int PH = E::PH, ROWS = E::ROWS_PH;
    constexpr int NIT = (ROWS * CPR + 511) / 512;
    // (the wave's five blocks lie in one phase: 80 consecutive tile rows, or five image rows of one half, never straddle BM / 2)
    const int my_phase = row_of(0) / ROWS;
    // this thread's residual chunks of the row passes, ALL in flight before the first accumulators are parked
    uint4 rres[PH][NIT];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 512 + tid;
            rres[ph][it] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < ROWS * CPR) {
                const int r = idx / CPR, c = idx - r * CPR;
                const int gm = grow_of(ph * ROWS + r), ch = n0 + c * 8;
                if (gm >= 0 && ch < d.N) rres[ph][it] = *(const uint4*)((const bf16_t*)d.residual + (long long)gm * d.ldr + ch);
            }
        }
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph > 0) {   // the previous phase's statistics pass is done with the staging area
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (owner && my_phase == ph) {
#pragma unroll
            for (int bm = 0; bm < 5; ++bm) {
                char* st = smem + (row_of(bm) - ph * ROWS) * P32 + col0 * 4;
#pragma unroll
                for (int bn = 0; bn < NB; ++bn) *(f32x4_t*)(st + bn * 64) = acc[bn][bm];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 512 + tid;
            if (idx >= ROWS * CPR) break;
            const int r = idx / CPR, c = idx - r * CPR;
            const int gm = row_tab[ph * ROWS + r], ch = n0 + c * 8;
            const float4 lo = *(const float4*)(smem + r * P32 + c * 32), hi = *(const float4*)(smem + r * P32 + c * 32 + 16);
            float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}, y[8];
            unpack8(rres[ph][it], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[e] += y[e];
                if (d.act == T2V_ACT_SILU) x[e] = silu_f(x[e]);
            }
            const uint4 val = pack8(x);
            if (d.colstat_out) *(uint4*)(smem + r * P32 + c * 32) = val;   // the statistics are of what is stored
            if (gm >= 0 && ch < d.N && !skip_stores) *(uint4*)(obase + (long long)gm * d.ldo + ch) = val;
        }
        if (d.colstat_out) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            constexpr int NSL = ROWS / 32, NCP = BN / 2;
            for (int it = tid; it < NSL * NCP; it += 512) {
                const int sl = it / NCP, cp = it - sl * NCP;
                const int gm0 = row_tab[ph * ROWS + sl * 32], col = n0 + 2 * cp;
                if (gm0 < 0 || col >= d.N) continue;
                float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) {
                    const uint32_t w2 = *(const uint32_t*)(smem + (sl * 32 + r) * P32 + (cp >> 2) * 32 + (cp & 3) * 4);
                    const float x0 = __uint_as_float(w2 << 16), x1 = __uint_as_float(w2 & 0xffff0000u);
                    a0 += x0; q0 = fmaf(x0, x0, q0);
                    a1 += x1; q1 = fmaf(x1, x1, q1);
                }
                *(float4*)(d.colstat_out + ((long long)(gm0 >> 5) * d.N + col) * 2) = make_float4(a0, q0, a1, q1);
            }
        }
    }
}

// k-groups: sum the partial accumulators of groups 1 .. KG-1 into group 0's through LDS (fixed order: deterministic).  Every wave of
// the workgroup calls it after the ring is dead; WPG = waves per k-group, wv = this wave's index within its group.
template <int KG, int WPG, int NB>
__device__ __forceinline__ void reduce_kgroups(char* smem, f32x4_t (&acc)[NB][5], int kgroup, int wv, int lane) {
    if constexpr (KG > 1) {
        if (kgroup > 0) {
            char* dst = smem + ((kgroup - 1) * WPG + wv) * (NB * 5 * 1024) + lane * 16;
#pragma unroll
            for (int bn = 0; bn < NB; ++bn)
#pragma unroll
                for (int bm = 0; bm < 5; ++bm) *(f32x4_t*)(dst + (bn * 5 + bm) * 1024) = acc[bn][bm];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kgroup == 0) {
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const char* src = smem + ((g - 1) * WPG + wv) * (NB * 5 * 1024) + lane * 16;
#pragma unroll
                for (int bn = 0; bn < NB; ++bn)
#pragma unroll
                    for (int bm = 0; bm < 5; ++bm) acc[bn][bm] += *(const f32x4_t*)(src + (bn * 5 + bm) * 1024);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

}  // namespace tile80
