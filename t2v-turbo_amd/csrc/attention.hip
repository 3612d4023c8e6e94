// Attention kernels for gfx950, head dim 64.
//
// attn_spatial_kernel: flash-style fused softmax(QK^T)V for the spatial self-attention and the
//   77-token text cross-attention.  4 waves x 32 queries per workgroup, K / V^T tiles of 64 keys
//   staged through LDS (register prefetch of tile t+1 under the MFMAs of tile t), scores computed
//   *transposed* (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that a lane owns one query column:
//   the online-softmax max/sum are lane-local plus one 32-lane exchange, and the probabilities are
//   already in B-operand position for O^T += V^T P^T — the MFMA contraction index is simply
//   enumerated in the order the score registers come out, and V^T is read from LDS in that order,
//   so P never goes through LDS or cross-lane shuffles (K rows sit in LDS with bits 2/3 of the key index
//   swapped, which makes each lane's 8 contraction keys one contiguous 16-byte chunk of V^T).
// attn_temporal_kernel: the sequence is the F (=16) frames of one pixel; one wave per
//   (clip, pixel, head), rows fetched with the frame stride straight from the token-major
//   buffers (no rearrange copies); tiny FLOPs, bandwidth bound.
#include "common.h"
#include <cstdlib>

// ablation bits (tools/attn_one.py --debug) only exist in a -DT2V_ATTN_ABLATE build: runtime branches inside the
// KV loop split its basic block (inexact s_waitcnt, no MFMA / VALU interleave)
#ifdef T2V_ATTN_ABLATE
#define ATT_ABL(bit) (debug & (bit))
#else
#define ATT_ABL(bit) false
#endif

// A/B switches of the spatial kernel's issue order (tools builds; the product build defines none of them):
//   -DT2V_ATTN_NOFENCE   drop the three sched_barrier fences of the KV loop (the compiler may interleave fragment reads, softmax and MFMAs)
//   -DT2V_ATTN_SETPRIO   s_setprio 1 around the two MFMA clusters (QK^T, PV), 0 for the softmax in between (cdna guide T5)
#ifdef T2V_ATTN_NOFENCE
#define T2V_ATTN_FENCE() do {} while (0)
#else
#define T2V_ATTN_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef T2V_ATTN_SETPRIO
#define T2V_ATTN_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define T2V_ATTN_PRIO(n) do {} while (0)
#endif

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int K_TILE_BYTES = KT * 128;
constexpr int VT_TILE_BYTES = 64 * 128;
constexpr int AT_STAGE = K_TILE_BYTES + VT_TILE_BYTES;

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// K tile [64 keys][64 d] and V^T tile [64 d][64 keys], both 128-byte rows with the 16-byte chunk index
// XOR-swizzled by (row>>1)&7, are filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write): 16 wave-instructions per tile pair, 4 per wave.  Keys past seq_kv: K rows come from a zero
// page; the V^T buffer's padding columns must be finite (the engine zero-fills them) since their P is 0.
#ifndef T2V_ATTN_WPE
#define T2V_ATTN_WPE 3  // waves per SIMD the register budget is sized for (tools: -DT2V_ATTN_WPE=2 / 4 variants for A/B runs)
#endif
// NW = waves per workgroup (4 or 8), 32 queries each: every workgroup streams ALL of its (image, head)'s K and V^T through LDS, so the
// L2 -> LDS fill is seq_kv * 256 bytes per NW * 32 queries — 1.05 GB per launch at 2 560 tokens x 5 heads x 16 images with four waves
// (5 TB/s in a 200 us launch), half of it with eight.  MEASURED (round 5, profiles/r05_attn_issue_order_variants.txt): halving that
// stream changes nothing (224-228 vs 224 us at four waves per SIMD; 240-244 vs 211-220 us at three, where only one eight-wave workgroup
// fits a CU) — the launch is not fill-bound; with r05_attn_pmc.csv (VALU active 45 %, matrix pipe 26 %, 11.7 VALU per MFMA) it runs at
// the sum of the softmax's VALU work and the MFMA work of the same wave.  The product launches NW = 4; T2V_ATTN_NW=8 is a tools switch.
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(T2V_ATTN_WPE, T2V_ATTN_WPE))) void attn_spatial_kernel(const bf16_t* __restrict__ q, int ldq,
                                                           const bf16_t* __restrict__ k, int ldk,
                                                           const bf16_t* __restrict__ vt, int ld_vt, long long vt_img_stride,
                                                           bf16_t* __restrict__ out, int ldo, int seq_q, int seq_kv,
                                                           int heads, int kv_div, float scale, const bf16_t* zero, int debug) {
    __shared__ __attribute__((aligned(16))) char smem[2 * AT_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y, img = blockIdx.z, img_kv = img / kv_div;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;
    const int qi = q0 + l31;
    const bool q_ok = qi < seq_q;

    // Q fragments (B operand: column = query, k = d)
    bf16x8_t qf[4];
    {
        const bf16_t* qp = q + ((long long)img * seq_q + (q_ok ? qi : 0)) * ldq + head * 64 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 u = q_ok ? *(const uint4*)(qp + kk * 16) : make_uint4(0, 0, 0, 0);
            qf[kk] = *(bf16x8_t*)&u;
        }
    }
    const bf16_t* kbase = k + (long long)img_kv * seq_kv * ldk + head * 64;
    const bf16_t* vbase = vt + (long long)img_kv * vt_img_stride + (long long)head * 64 * ld_vt;
    const int ntile = (seq_kv + KT - 1) / KT;

    // DMA lane roles: wave-instruction i (= wave + NW*j, j < 8 / NW) covers tile rows [8i, 8i+8)
    constexpr int DJ = 8 / NW;
    int drow[DJ], dchunk[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        drow[j] = (wave + NW * j) * 8 + (lane >> 3);
        dchunk[j] = ((lane & 7) ^ ((drow[j] >> 1) & 7)) * 8;
    }
    auto stage = [&](int t, int buf) {
        char* sk = smem + buf * AT_STAGE;
        char* sv = sk + K_TILE_BYTES;
        const int key0 = t * KT;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            // LDS row r of the K tile holds key perm(r) = r with bits 2 and 3 swapped: the 8 score registers a lane
            // feeds into one PV K-step (rows {4hi + 0..3, 8 + 4hi + 0..3} of a 16-row group) are then 8 CONSECUTIVE
            // keys, i.e. one aligned 16-byte chunk of the V^T row (a single ds_read_b128 per fragment)
            const int key = key0 + ((drow[j] & ~12) | ((drow[j] & 4) << 1) | ((drow[j] & 8) >> 1));
            const bf16_t* ksrc = key < seq_kv ? kbase + (long long)key * ldk + dchunk[j] : zero;
            dma16(ksrc, sk + (wave + NW * j) * 1024);
            dma16(vbase + (long long)drow[j] * ld_vt + key0 + dchunk[j], sv + (wave + NW * j) * 1024);
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int swz = (lane >> 1) & 7;
    const float c2 = scale * 1.4426950408889634f;  // softmax in base 2

    stage(0, 0);
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!ATT_ABL(16)) __syncthreads();
        if (t + 1 < ntile && !ATT_ABL(4)) stage(t + 1, buf ^ 1);
        const char* sk = smem + buf * AT_STAGE;
        const char* sv = sk + K_TILE_BYTES;
        // ---- S^T = K Q^T: all 8 K fragments are read up front (one batch of ds_read_b128, counted waits) so the
        // MFMAs run back to back instead of read -> wait -> MFMA eight times -------------------------------
        f32x16_t s[2];
#ifdef T2V_ATTN_LOWREG   // K fragments of one 32-key half at a time (the second half's reads go out before the first half's MFMAs)
        {
            bf16x8_t kf0[4], kf1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf0[kk] = *(const bf16x8_t*)(sk + l31 * 128 + (((kk * 2 + hi) ^ swz) << 4));
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf1[kk] = *(const bf16x8_t*)(sk + (32 + l31) * 128 + (((kk * 2 + hi) ^ swz) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[kk], qf[kk], s[0], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[kk], qf[kk], s[1], 0, 0, 0);
        }
        T2V_ATTN_FENCE();
#else
        bf16x8_t kf[2][4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                kf[h2][kk] = *(const bf16x8_t*)(sk + (h2 * 32 + l31) * 128 + (((kk * 2 + hi) ^ swz) << 4));
        T2V_ATTN_FENCE();
        T2V_ATTN_PRIO(1);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
            if (!ATT_ABL(8))
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[h2][kk], qf[kk], s[h2], 0, 0, 0);
        }
        T2V_ATTN_PRIO(0);
        T2V_ATTN_FENCE();
#endif
        // V^T fragments for the PV product: issued now, they land under the softmax VALU work.  K-step ks = 2*h2 + st
        // contracts keys h2*32 + st*16 + 8*hi + 0..7 = chunk 4*h2 + 2*st + hi of the V^T row.
        bf16x8_t vfr[4][2];
#ifdef T2V_ATTN_LOWREG   // (-DT2V_ATTN_LOWREG: only K-steps 0, 1 before the softmax; 2, 3 follow it, under the first four PV MFMAs — 16 registers fewer live across the softmax, for a four-waves-per-SIMD build)
        constexpr int VF_EARLY = 2;
#else
        constexpr int VF_EARLY = 4;
#endif
#pragma unroll
        for (int ks = 0; ks < VF_EARLY; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                vfr[ks][db] = *(const bf16x8_t*)(sv + (db * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
        T2V_ATTN_FENCE();
        // ---- online softmax (this lane: one query, 32 of the 64 keys; partner lane^32 the rest) ---
        // scores stay raw; max is taken on them and exp2(fma(s, c, -c*max)) folds scale*log2(e): 3 VALU
        // ops per element (max, fma, exp2) instead of 5
        const int key_base = t * KT + 8 * hi;  // register r of half h2 is key key_base + h2*32 + (r>>3)*16 + (r&7)
        const bool tail = (t + 1) * KT > seq_kv;
        if (!ATT_ABL(1)) {
            float mloc = -INFINITY;
            if (tail) {  // only the last tile has keys past seq_kv: a scalar branch, not 32 compare/select pairs per tile
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = s[h2][r];
                        if (key_base + h2 * 32 + (r >> 3) * 16 + (r & 7) >= seq_kv) v = -INFINITY;
                        asm volatile("" : "+v"(v));  // keep the masking inside this branch (no if-conversion into the hot path)
                        s[h2][r] = v;
                    }
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[h2][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);  // raw-score units
            const bool grew = m_new > m_run;
            const float mc = m_new * c2;
            float lsum = 0.f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[h2][r], c2, -mc));
                    s[h2][r] = p;
                    lsum += p;
                }
            if (__any(grew)) {  // wave-uniform: rescale only when some query's running max moved
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                m_run = m_new;
            }
            l_run += lsum;
        }
        // ---- O^T += V^T P^T: the score registers already come out in contraction order ----------------------
#pragma unroll
        for (int ks = VF_EARLY; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                vfr[ks][db] = *(const bf16x8_t*)(sv + (db * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
        T2V_ATTN_PRIO(1);
        if (!ATT_ABL(2))
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int h2 = ks >> 1, st = ks & 1;
            uint4 pu;
            pu.x = pack2bf(s[h2][st * 8 + 0], s[h2][st * 8 + 1]);
            pu.y = pack2bf(s[h2][st * 8 + 2], s[h2][st * 8 + 3]);
            pu.z = pack2bf(s[h2][st * 8 + 4], s[h2][st * 8 + 5]);
            pu.w = pack2bf(s[h2][st * 8 + 6], s[h2][st * 8 + 7]);
            const bf16x8_t pb = *(bf16x8_t*)&pu;
#pragma unroll
            for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks][db], pb, o[db], 0, 0, 0);
        }
        T2V_ATTN_PRIO(0);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (q_ok) {
        bf16_t* op = out + ((long long)img * seq_q + qi) * ldo + head * 64 + 4 * hi;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv);
                w.y = pack2bf(o[db][g * 4 + 2] * inv, o[db][g * 4 + 3] * inv);
                *(uint2*)(op + db * 32 + g * 8) = w;
            }
    }
}

#ifdef T2V_EXPERIMENTAL   // (measured no faster: in the T2V_EXPERIMENTAL=1 library only)
// ------------------------------------------------------------------------------------------------
// attn_spatial_q64_kernel (round 5): the same arithmetic with SIXTY-FOUR queries per wave — two 32-query sets A and B that share every K / V^T
// fragment read — and the two sets' phases offset by one in program order, so that one set's softmax VALU work is issued in the gaps of the
// other set's MFMAs INSIDE one wave:
//     QK_A | QK_B + softmax_A | PV_A + softmax_B | PV_B
// Why: the counters of the 32-query kernel (profiles/r05_attn_pmc.csv) show the launch running at the SUM of a wave's softmax VALU time (45 % of
// the SIMD cycles) and its MFMA time (26 %) at any occupancy — at head dimension 64 a 64-key tile is 16 MFMAs against ~190 VALU issue slots, and a
// wave issues in order: with one query set per wave nothing of its own softmax can sit under its own MFMAs.  Per query the operations and their
// order are exactly those of attn_spatial_kernel (bit-identical results).  The rescale of the running output (taken only when a running max
// moved) and the masking of the last tile's padding keys are scalar branches BETWEEN the interleaved regions.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_spatial_q64_kernel(
    const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk, const bf16_t* __restrict__ vt, int ld_vt,
    long long vt_img_stride, bf16_t* __restrict__ out, int ldo, int seq_q, int seq_kv, int heads, int kv_div, float scale, const bf16_t* zero) {
    __shared__ __attribute__((aligned(16))) char smem[2 * AT_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y, img = blockIdx.z, img_kv = img / kv_div;
    const int q0 = blockIdx.x * 256 + wave * 64;
    int qi[2];
    bool q_ok[2];
    bf16x8_t qf[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        qi[a] = q0 + a * 32 + l31;
        q_ok[a] = qi[a] < seq_q;
        const bf16_t* qp = q + ((long long)img * seq_q + (q_ok[a] ? qi[a] : 0)) * ldq + head * 64 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 u = q_ok[a] ? *(const uint4*)(qp + kk * 16) : make_uint4(0, 0, 0, 0);
            qf[a][kk] = *(bf16x8_t*)&u;
        }
    }
    const bf16_t* kbase = k + (long long)img_kv * seq_kv * ldk + head * 64;
    const bf16_t* vbase = vt + (long long)img_kv * vt_img_stride + (long long)head * 64 * ld_vt;
    const int ntile = (seq_kv + KT - 1) / KT;
    int drow[2], dchunk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        drow[j] = (wave + 4 * j) * 8 + (lane >> 3);
        dchunk[j] = ((lane & 7) ^ ((drow[j] >> 1) & 7)) * 8;
    }
    auto stage = [&](int t, int buf) {   // (as in attn_spatial_kernel: K rows with bits 2 / 3 of the key index swapped, V^T rows as they lie)
        char* sk = smem + buf * AT_STAGE;
        char* sv = sk + K_TILE_BYTES;
        const int key0 = t * KT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = key0 + ((drow[j] & ~12) | ((drow[j] & 4) << 1) | ((drow[j] & 8) >> 1));
            const bf16_t* ksrc = key < seq_kv ? kbase + (long long)key * ldk + dchunk[j] : zero;
            dma16(ksrc, sk + (wave + 4 * j) * 1024);
            dma16(vbase + (long long)drow[j] * ld_vt + key0 + dchunk[j], sv + (wave + 4 * j) * 1024);
        }
    };
    f32x16_t o[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[a][0][r] = 0.f; o[a][1][r] = 0.f; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int swz = (lane >> 1) & 7;
    const float c2 = scale * 1.4426950408889634f;

    // one query set's scores -> probabilities (bf16, PV B-operand order) + the tile's contribution to its running sum; returns whether any
    // lane's running max moved.  No control flow: the caller interleaves this with the other set's MFMAs.
    auto softmax_set = [&](f32x16_t (&sc)[2], float m_old, float& m_new, float& lsum, bf16x8_t (&pb)[4]) {
        float mloc = -INFINITY;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sc[h2][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        m_new = fmaxf(m_old, mloc);
        const float mc = m_new * c2;
        lsum = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sc[h2][r], c2, -mc));
                sc[h2][r] = p;
                lsum += p;
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int h2 = ks >> 1, st = ks & 1;
            uint4 pu;
            pu.x = pack2bf(sc[h2][st * 8 + 0], sc[h2][st * 8 + 1]);
            pu.y = pack2bf(sc[h2][st * 8 + 2], sc[h2][st * 8 + 3]);
            pu.z = pack2bf(sc[h2][st * 8 + 4], sc[h2][st * 8 + 5]);
            pu.w = pack2bf(sc[h2][st * 8 + 6], sc[h2][st * 8 + 7]);
            pb[ks] = *(bf16x8_t*)&pu;
        }
    };
    auto mask_tail = [&](f32x16_t (&sc)[2], int key_base) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = sc[h2][r];
                if (key_base + h2 * 32 + (r >> 3) * 16 + (r & 7) >= seq_kv) v = -INFINITY;
                asm volatile("" : "+v"(v));
                sc[h2][r] = v;
            }
    };
    auto rescale = [&](int a, float m_new) {   // (wave-uniform branch at the call site)
        const float alpha = __builtin_amdgcn_exp2f((m_run[a] - m_new) * c2);
        l_run[a] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[a][0][r] *= alpha; o[a][1][r] *= alpha; }
        m_run[a] = m_new;
    };
    // MFMA / VALU interleave of one overlapped region: 8 MFMAs, the other set's softmax spread behind them
    auto interleave_hint = [&]() {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
        }
    };

    stage(0, 0);
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntile) stage(t + 1, buf ^ 1);
        const char* sk = smem + buf * AT_STAGE;
        const char* sv = sk + K_TILE_BYTES;
        const int key_base = t * KT + 8 * hi;
        const bool tail = (t + 1) * KT > seq_kv;
        f32x16_t sA[2], sB[2];
        // (the K fragments are read once per SET, right where they are multiplied: keeping one copy alive across both sets' QK phases
        // costs 32 registers at the kernel's register peak — and LDS reads are not what this kernel waits for)
        auto kfrag = [&](int h2, int kk) { return *(const bf16x8_t*)(sk + (h2 * 32 + l31) * 128 + (((kk * 2 + hi) ^ swz) << 4)); };
        // ---- phase 1: QK_A ------------------------------------------------------------------------------------------------------
        {
            bf16x8_t kf[2][4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[h2][kk] = kfrag(h2, kk);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sA[h2][r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) sA[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[h2][kk], qf[0][kk], sA[h2], 0, 0, 0);
            }
        }
        if (tail) mask_tail(sA, key_base);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: QK_B under softmax_A ------------------------------------------------------------------------------------------
        float mA, lsA, mB, lsB;
        bf16x8_t pA[4], pB[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sB[h2][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) sB[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(h2, kk), qf[1][kk], sB[h2], 0, 0, 0);
        }
        softmax_set(sA, m_run[0], mA, lsA, pA);
#pragma unroll
        for (int g = 0; g < 8; ++g) {   // fragment read, its MFMA, a slice of the other set's softmax
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tail) mask_tail(sB, key_base);
        if (__any(mA > m_run[0])) rescale(0, mA);
        l_run[0] += lsA;
        // V^T fragments (shared by both sets)
        bf16x8_t vfr[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                vfr[ks][db] = *(const bf16x8_t*)(sv + (db * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 3: PV_A under softmax_B ------------------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) o[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks][db], pA[ks], o[0][db], 0, 0, 0);
        softmax_set(sB, m_run[1], mB, lsB, pB);
        interleave_hint();
        __builtin_amdgcn_sched_barrier(0);
        if (__any(mB > m_run[1])) rescale(1, mB);
        l_run[1] += lsB;
        // ---- phase 4: PV_B --------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks][db], pB[ks], o[1][db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float l_tot = l_run[a] + __shfl_xor(l_run[a], 32, 64);
        const float inv = 1.f / l_tot;
        if (q_ok[a]) {
            bf16_t* op = out + ((long long)img * seq_q + qi[a]) * ldo + head * 64 + 4 * hi;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w;
                    w.x = pack2bf(o[a][db][g * 4 + 0] * inv, o[a][db][g * 4 + 1] * inv);
                    w.y = pack2bf(o[a][db][g * 4 + 2] * inv, o[a][db][g * 4 + 3] * inv);
                    *(uint2*)(op + db * 32 + g * 8) = w;
                }
        }
    }
}

#endif   // T2V_EXPERIMENTAL

// ------------------------------------------------------------------------------------------------
// Temporal attention: the sequence is the F frames of one pixel.  One wave per (clip, pixel, head), matrix cores for
// both products (the VALU form of this kernel was instruction-bound at a third of the HBM rate):
//   S^T = K Q^T   v_mfma_f32_16x16x32_bf16, A = K rows (key j = lane&15), B = Q rows (query i = lane&15); every lane
//                 fetches ITS 16-byte slices of q_i / k_j / v_j straight from the token-major buffers at the frame
//                 stride (rows of 128 contiguous bytes per head), no staging for Q and K.
//   softmax       a lane owns query column i and keys j = 4g + r (g = lane>>4): lane-local + 2 shuffles (xor 16, 32).
//   O^T = V^T P^T v_mfma_f32_16x16x16_bf16: P^T is already in B-operand position (k = 4g + r); V^T (contraction index =
//                 frame, the strided one) goes through a 2.5 KiB per-wave LDS transpose (16-bit writes, ds_read_b64).
// F > 16 loops over 16-frame query / key blocks with an online softmax.
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short bf16x4s_t;
constexpr int VT_LD = 20;  // V^T row stride in elements (40 B: keeps the 8-byte fragment reads aligned, spreads banks)
__global__ __launch_bounds__(256) void attn_temporal_kernel(const bf16_t* __restrict__ q, int ldq,
                                                            const bf16_t* __restrict__ k, int ldk,
                                                            const bf16_t* __restrict__ v, int ldv,
                                                            bf16_t* __restrict__ out, int ldo, long long n_items, int F,
                                                            int HW, int heads, float scale, float* __restrict__ probs) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[4][64 * VT_LD];  // [wave] V^T [d][frame]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long item = (long long)blockIdx.x * 4 + wv;
    if (item >= n_items) return;  // whole wave exits together (item is wave-uniform)
    const int head = (int)(item % heads);
    const long long bp = item / heads;
    const int p = (int)(bp % HW);
    const long long b = bp / HW;
    const long long row0 = b * F * HW + p;  // row of frame f = row0 + f*HW
    const int l15 = lane & 15, g = lane >> 4;
    const int col = head * 64 + g * 8;     // this lane's 8-channel slice of K-step 0 (K-step 1: +32)
    bf16_t* vt = lds[wv];
    const int nblk = (F + 15) / 16;
    const float c2 = scale * 1.4426950408889634f;  // softmax in base 2 on raw scores
    for (int qb = 0; qb < nblk; ++qb) {
        const int qi = qb * 16 + l15;
        const bool q_ok = qi < F;
        bf16x8_t qf[2];
        {
            const bf16_t* qp = q + (row0 + (long long)(q_ok ? qi : 0) * HW) * ldq + col;
            uint4 u0 = *(const uint4*)qp, u1 = *(const uint4*)(qp + 32);
            qf[0] = *(bf16x8_t*)&u0; qf[1] = *(bf16x8_t*)&u1;
        }
        float m_run = -INFINITY, l_run = 0.f;
        f32x4_t o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) o[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nblk; ++kb) {
            const int kj = kb * 16 + l15;
            const bool k_ok = kj < F;
            const long long r = row0 + (long long)(k_ok ? kj : 0) * HW;
            uint4 k0 = *(const uint4*)(k + r * ldk + col), k1 = *(const uint4*)(k + r * ldk + col + 32);
            uint4 v0 = *(const uint4*)(v + r * ldv + col), v1 = *(const uint4*)(v + r * ldv + col + 32);
            if (!k_ok) { k0 = k1 = v0 = v1 = make_uint4(0, 0, 0, 0); }  // padding frames: V must be finite (P is 0 there)
            // ---- S^T[j][i] = sum_d K[j][d] Q[i][d]: lane holds column i = l15, rows j = 4g + r
            f32x4_t st = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8_t*)&k0, qf[0], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8_t*)&k1, qf[1], st, 0, 0, 0);
            // ---- V^T through LDS (frame index becomes the contiguous one)
            __builtin_amdgcn_wave_barrier();  // previous block's fragment reads are done
            {
                const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = g * 8 + (e >> 2) * 32 + (e & 3) * 2;
                    vt[d * VT_LD + l15] = (bf16_t)(w[e] & 0xffffu);
                    vt[(d + 1) * VT_LD + l15] = (bf16_t)(w[e] >> 16);
                }
            }
            // ---- online softmax down column i
            float sc[4];
            float mloc = -INFINITY;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                sc[rr] = (kb * 16 + 4 * g + rr < F) ? st[rr] : -INFINITY;
                mloc = fmaxf(mloc, sc[rr]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            float lsum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { sc[rr] = __builtin_amdgcn_exp2f((sc[rr] - m_new) * c2); lsum += sc[rr]; }
            lsum += __shfl_xor(lsum, 16, 64);
            lsum += __shfl_xor(lsum, 32, 64);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
            if (probs && nblk == 1 && q_ok) {  // attention_probs [item][i][j], complete after the only key block
                const float inv1 = 1.f / l_run;
                float* pr = probs + (item * F + qi) * F + 4 * g;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (4 * g + rr < F) pr[rr] = sc[rr] * inv1;
            }
            uint2 pu;
            pu.x = pack2bf(sc[0], sc[1]);
            pu.y = pack2bf(sc[2], sc[3]);
            const bf16x4s_t pb = *(bf16x4s_t*)&pu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- O^T[d][i] += sum_j V^T[d][j] P^T[j][i]: 4 d-blocks of 16
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint2 a = *(const uint2*)(vt + (m * 16 + l15) * VT_LD + 4 * g);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[m][rr] *= alpha;
                o[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*(const bf16x4s_t*)&a, pb, o[m], 0, 0, 0);
            }
        }
        if (q_ok) {
            const float inv = 1.f / l_run;
            bf16_t* op = out + (row0 + (long long)qi * HW) * ldo + head * 64 + 4 * g;
#pragma unroll
            for (int m = 0; m < 4; ++m) {  // O^T rows d = 16m + 4g + r of column i: 4 consecutive channels = 8 bytes
                uint2 w;
                w.x = pack2bf(o[m][0] * inv, o[m][1] * inv);
                w.y = pack2bf(o[m][2] * inv, o[m][3] * inv);
                *(uint2*)(op + m * 16) = w;
            }
        }
    }
}

// attention_probs for F > 16 (rare: the reference records them at F = 16): separate exact pass
__global__ __launch_bounds__(256) void attn_temporal_probs_kernel(const bf16_t* __restrict__ q, int ldq,
                                                                  const bf16_t* __restrict__ k, int ldk, long long n_items,
                                                                  int F, int HW, int heads, float scale,
                                                                  float* __restrict__ probs) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // (item, i)
    if (idx >= n_items * F) return;
    const long long item = idx / F;
    const int i = (int)(idx % F);
    const int head = (int)(item % heads);
    const long long bp = item / heads;
    const long long row0 = (bp / HW) * F * HW + (bp % HW);
    const bf16_t* qp = q + (row0 + (long long)i * HW) * ldq + head * 64;
    float mx = -INFINITY;
    float* pr = probs + idx * F;
    for (int j = 0; j < F; ++j) {
        const bf16_t* kp = k + (row0 + (long long)j * HW) * ldk + head * 64;
        float d = 0.f;
        for (int e = 0; e < 64; ++e) d += bf2f(qp[e]) * bf2f(kp[e]);
        pr[j] = d * scale;
        mx = fmaxf(mx, pr[j]);
    }
    float sum = 0.f;
    for (int j = 0; j < F; ++j) { pr[j] = __expf(pr[j] - mx); sum += pr[j]; }
    for (int j = 0; j < F; ++j) pr[j] /= sum;
}

}  // namespace

static int g_attn_debug = 0;
extern "C" int t2v_attn_debug(int bits) { g_attn_debug = bits; return T2V_OK; }
// Which form of the spatial forward t2v_attn_spatial launches: 0 = the product kernel (4 waves x 32 queries), 8 = eight waves per workgroup,
// 64 = 64 queries per wave on launches with >= 512 queries and keys, 65 = 64 queries per wave always.  The last three are MEASURED NO FASTER
// (profiles/r05_attn_issue_order_variants.txt) and exist for tools and tests; -1 (initial) = from T2V_ATTN_NW / T2V_ATTN_Q64 in the environment.
#ifdef T2V_EXPERIMENTAL
static int g_attn_form = -1;
extern "C" int t2v_attn_spatial_form(int form) { g_attn_form = form; return T2V_OK; }
#endif

extern "C" int t2v_attn_spatial(const void* q, int ldq, const void* k, int ldk, const void* vt, int ld_vt,
                                long long vt_img_stride, void* out, int ldo, int n_img, int seq_q, int seq_kv, int heads,
                                int kv_div, float scale, void* stream) {
    T2V_REQUIRE(q && k && vt && out, T2V_EINVAL, "t2v_attn_spatial: null pointer");
    T2V_REQUIRE(n_img > 0 && seq_q > 0 && seq_kv > 0 && heads > 0 && kv_div > 0, T2V_EINVAL, "t2v_attn_spatial: bad size");
    T2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ld_vt % 8 == 0 && ldo % 4 == 0, T2V_ESHAPE, "t2v_attn_spatial: strides");
    T2V_REQUIRE(ld_vt >= ((seq_kv + 63) / 64) * 64, T2V_ESHAPE, "t2v_attn_spatial: V^T rows must be padded to 64 keys");
    T2V_REQUIRE(heads <= 65535 && n_img <= 65535, T2V_ESHAPE, "t2v_attn_spatial: grid");
    if (vt_img_stride <= 0) vt_img_stride = (long long)heads * 64 * ld_vt;
    T2V_REQUIRE(vt_img_stride % 8 == 0, T2V_ESHAPE, "t2v_attn_spatial: vt_img_stride");
#ifdef T2V_EXPERIMENTAL
    if (g_attn_form < 0) {   // T2V_ATTN_NW=8 / T2V_ATTN_Q64=1 (long launches) / 2 (always): tools switches, see t2v_attn_spatial_form
        const int nw = getenv("T2V_ATTN_NW") ? atoi(getenv("T2V_ATTN_NW")) : 0, q = getenv("T2V_ATTN_Q64") ? atoi(getenv("T2V_ATTN_Q64")) : 0;
        g_attn_form = q == 2 ? 65 : (q == 1 ? 64 : (nw == 8 ? 8 : 0));
    }
    const bool eight = g_attn_form == 8;
    if (g_attn_form == 65 || (g_attn_form == 64 && seq_kv >= 512 && seq_q >= 512 && g_attn_debug == 0)) {
        dim3 grid((seq_q + 255) / 256, heads, n_img);
        hipLaunchKernelGGL(attn_spatial_q64_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                           (const bf16_t*)vt, ld_vt, vt_img_stride, (bf16_t*)out, ldo, seq_q, seq_kv, heads, kv_div, scale,
                           (const bf16_t*)t2v_zero_page());
    } else if (eight) {
        dim3 grid((seq_q + 255) / 256, heads, n_img);
        hipLaunchKernelGGL(attn_spatial_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                           (const bf16_t*)k, ldk, (const bf16_t*)vt, ld_vt, vt_img_stride, (bf16_t*)out, ldo, seq_q, seq_kv, heads,
                           kv_div, scale, (const bf16_t*)t2v_zero_page(), g_attn_debug);
    } else
#endif
    {
        dim3 grid((seq_q + 127) / 128, heads, n_img);
        hipLaunchKernelGGL(attn_spatial_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                           (const bf16_t*)k, ldk, (const bf16_t*)vt, ld_vt, vt_img_stride, (bf16_t*)out, ldo, seq_q, seq_kv, heads,
                           kv_div, scale, (const bf16_t*)t2v_zero_page(), g_attn_debug);
    }
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_attn_temporal(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                                 int ldo, int n_clips, int frames, int hw, int heads, float scale, float* probs,
                                 void* stream) {
    T2V_REQUIRE(q && k && v && out, T2V_EINVAL, "t2v_attn_temporal: null pointer");
    T2V_REQUIRE(n_clips > 0 && frames > 0 && hw > 0 && heads > 0, T2V_EINVAL, "t2v_attn_temporal: bad size");
    T2V_REQUIRE(frames <= 1024, T2V_ESHAPE, "t2v_attn_temporal: more than 1024 frames not supported");
    T2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, T2V_ESHAPE, "t2v_attn_temporal: strides");
    const long long n_items = (long long)n_clips * hw * heads;
    const unsigned blocks = (unsigned)((n_items + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
hipLaunchKernelGGL(attn_temporal_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)v, ldv, (bf16_t*)out, ldo, n_items, frames, hw, heads, scale, probs);
    if (probs && frames > 16) {
        T2V_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_temporal_probs_kernel, dim3((unsigned)((n_items * frames + 255) / 256)), dim3(256), 0, s,
                           (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, n_items, frames, hw, heads, scale, probs);
    }
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
