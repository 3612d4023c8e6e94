// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   out[M,N] = epilogue(alpha * gather(A)[M,K] x W[N,K]^T)
//
// Design (CDNA4):
//  * 4 or 8 wavefronts (64 lanes) per workgroup, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//    The MFMA "A" operand is the weight tile (rows = output channels) and the "B" operand the
//    activation tile (columns = tokens), so in the accumulator a lane owns ONE token and 4-channel
//    runs of it: the epilogue (bias, time-embedding row vector, residual, SiLU, GEGLU with value and
//    gate in the same lane) runs straight out of registers with 8/16-byte row accesses — no LDS
//    round trip, no barrier.
//  * K is walked in 64-element steps.  For the conv modes a K step is one 64-channel slab of one
//    filter tap: the im2col matrix is never built, each lane aims its 16-byte LDS-DMA
//    (global_load_lds_dwordx4) at the shifted source row, or at a zero page for padding taps.
//    Nearest-x2 upsampling, stride 2, the (3,1,1) temporal taps and the skip-connection channel
//    concat are all address arithmetic in that gather (every mode is a kh x kw conv over an
//    (n, H, W) grid; LINEAR is 1x1 over (M,1,1)).
//  * LDS tiles are [rows][64] bf16 (128 B rows) with the 16-byte chunk index XOR-swizzled by
//    (row>>1)&7 — applied on the DMA *source* side (the LDS image of a DMA is lane-linear) and on
//    the ds_read_b128 fragment reads, which makes those reads bank-conflict free.
//  * STAGES-deep ring of tiles; waits are counted (s_waitcnt vmcnt(N)) with a raw s_barrier so
//    that STAGES-2 later tiles stay in flight across the barrier instead of being drained.
//  * split-K (blockIdx.z) for problems with too few tiles to fill 256 CUs: fp32 partial slabs in a
//    caller workspace, reduced (fixed order, deterministic) + epilogue by a second small kernel.
//  * workgroup id -> tile mapping is XCD-aware: each of the 8 XCDs (private L2s) gets a contiguous
//    run of tiles, N-tile fastest, so the tiles that share an A panel hit the same L2.
#include "common.h"
#ifdef T2V_EXPERIMENTAL
#include "gemm2.h"
#endif
#include "gelu_poly.h"
#include <cstdlib>
#include <type_traits>

struct GemmParams {
    t2v_gemm_desc d;
    const void* zero;
    int K, nk, taps, nsrc;
    int gh, gw;  // source grid
    int kh, kw, stride, pad_y, pad_x, ups;
    int h_out, w_out;
    int tiles_m, tiles_n;
    int xcd_m, xcd_n;  // the 8 XCDs as an xcd_m x xcd_n grid over the tile grid (product 8)
    int nblk;          // blocks of that grid; per block: first linear tile index, first tile row / column, width in tiles
    int blk_start[8], blk_r0[8], blk_c0[8], blk_w[8];
    int vec4;          // 4-channel runs may use vector accesses
    int splits, nk_per_split;
    float* ws;         // split-K partials [splits][M][N]
    int debug;         // ablation bits (tools only): 1 = skip DMA in the main loop, 2 = skip MFMA, 4 = skip epilogue
};

// Ablation switches (tools only): a -DT2V_GEMM_ABLATE build honours GemmParams::debug bits inside the main loop
// (1 = no DMA, 2 = no MFMA, 8 = no LDS fragment reads, 16 = no barrier, 4 = no epilogue, 32 = epilogue arithmetic and LDS
// slabs but no global stores, 64 = accumulators start at zero: no bias / row-vector / residual loads, 128 = leave after the
// prologue: launch + descriptor + first DMA latency only; LoRA epilogue: 256 = keep everything (no mask arithmetic), 512 = no LoRA
// MFMAs, 1024 = no t rows, 2048 = no U staging).  The product build
// compiles them out: runtime branches inside the K loop split its basic block and cost ~20 % (exact s_waitcnt
// counts and the MFMA / ds_read interleave both need straight-line code).
#ifdef T2V_GEMM_ABLATE
#define ABL(bit) (p.debug & (bit))
#else
#define ABL(bit) false
#endif

namespace {

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// epilogue on a run of 16 consecutive output channels of one token row (one lane's share of a
// 32x32 accumulator tile)
struct Epi {
    const t2v_gemm_desc* d;
    long long o_off;
    int n_out, vec;
    // pre_res: the run's 16 residual values already in registers (issued for the whole wave tile before the first
    // store, so their latency is paid once instead of once per run behind the previous run's stores)
    // FAST: every run is a full, 16-byte aligned run (host-checked: vector-aligned operands, n_out % 16 == 0): the
    // per-element partial paths compile away.  The epilogue is unrolled over the wave tile, so this is most of the kernel's
    // code size, and these kernels pay for instruction fetch (a ~10 % longer epilogue measured 6 % slower end to end).
    template <bool FAST = false>
    __device__ __forceinline__ void run(float* v, float* gate, int gm, int ch_in, int ch_out, bool has_pre = false,
                                        uint4 pre0 = uint4{0, 0, 0, 0}, uint4 pre1 = uint4{0, 0, 0, 0}) const {
        if (ch_out >= n_out) return;
        compute<FAST>(v, gate, gm, ch_in, ch_out, has_pre, pre0, pre1);
        store<FAST>(v, gm, ch_out);
    }
    // everything up to the final values of the run (v is updated in place)
    template <bool FAST = false>
    __device__ __forceinline__ void compute(float* v, float* gate, int gm, int ch_in, int ch_out, bool has_pre = false,
                                            uint4 pre0 = uint4{0, 0, 0, 0}, uint4 pre1 = uint4{0, 0, 0, 0}) const {
        const t2v_gemm_desc& dd = *d;
        const bool full = FAST || (vec && ch_out + 16 <= n_out);
        // FAST kernels (alpha == 1, host-checked) start their accumulators at bias (+ rowvec + residual when there is no
        // gate), so those terms cost nothing here and the residual's latency hides under the main loop
        constexpr bool FOLD = FAST;
        if (!FOLD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] *= dd.alpha;
        }
        if (!FOLD && dd.bias) {
            if (full) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = *(const float4*)(dd.bias + ch_in + 4 * q);
                    v[4 * q] += b.x; v[4 * q + 1] += b.y; v[4 * q + 2] += b.z; v[4 * q + 3] += b.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (ch_in + e < dd.N) v[e] += dd.bias[ch_in + e];
            }
        }
        if (gate) {  // GEGLU: value * gelu(gate); gate columns sit 32 packed rows after the values
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!FOLD && dd.bias) b = *(const float4*)(dd.bias + ch_in + 32 + 4 * q);
                const float al = FOLD ? 1.0f : dd.alpha;
                v[4 * q] *= fast_gelu(gate[4 * q] * al + b.x);
                v[4 * q + 1] *= fast_gelu(gate[4 * q + 1] * al + b.y);
                v[4 * q + 2] *= fast_gelu(gate[4 * q + 2] * al + b.z);
                v[4 * q + 3] *= fast_gelu(gate[4 * q + 3] * al + b.w);
            }
        }
        if constexpr (!FAST) {  // dropout(alpha * acc + bias) / (1 - p): only the generic kernels carry it (the host keeps launches
                                // with a dropout threshold off the FAST ones, whose epilogue is sized for the inference step)
            if (dd.drop_thr) {
                const uint64_t key = dropout_key(*(const uint64_t*)dd.drop_seed, dd.drop_site);
                const uint64_t quad0 = ((uint64_t)gm * (uint64_t)dd.drop_ncols + (uint64_t)(dd.drop_col0 + ch_out)) >> 2;
                const uint32_t keep = dropout_keep_mask<4>(key, quad0, dd.drop_thr >> 16);
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * dd.drop_inv_keep : 0.f;
            }
        }
        const bool folded_rr = FOLD && !gate;  // rowvec / residual already inside the accumulators
        if (dd.rowvec && !folded_rr) {
            const float* rv = dd.rowvec + (long long)(gm / dd.rowvec_div) * dd.ld_rowvec + ch_out;
            if (full) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 r = *(const float4*)(rv + 4 * q);
                    v[4 * q] += r.x; v[4 * q + 1] += r.y; v[4 * q + 2] += r.z; v[4 * q + 3] += r.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (ch_out + e < n_out) v[e] += rv[e];
            }
        }
        if (dd.residual && !folded_rr) {
            const bf16_t* rp = (const bf16_t*)dd.residual + o_off + (long long)gm * dd.ldr + ch_out;
            if (full) {
                float rf[16];
                if (!has_pre) { pre0 = *(const uint4*)rp; pre1 = *(const uint4*)(rp + 8); }
                unpack8(pre0, rf);
                unpack8(pre1, rf + 8);
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += rf[e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (ch_out + e < n_out) v[e] += bf2f(rp[e]);
            }
        }
        if (dd.act == T2V_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = silu_f(v[e]);
        }
    }
    template <bool FAST = false>
    __device__ __forceinline__ void store(const float* v, int gm, int ch_out) const {
        const t2v_gemm_desc& dd = *d;
        const bool full = FAST || (vec && ch_out + 16 <= n_out);
        if (dd.out_f32) {
            float* op = (float*)dd.out + o_off + (long long)gm * dd.ldo + ch_out;
            if (full) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *(float4*)(op + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (ch_out + e < n_out) op[e] = v[e];
            }
        } else {
            bf16_t* op = (bf16_t*)dd.out + o_off + (long long)gm * dd.ldo + ch_out;
            if (full) {
                *(uint4*)op = pack8(v);
                *(uint4*)(op + 8) = pack8(v + 8);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (ch_out + e < n_out) op[e] = f2bf(v[e]);
            }
        }
    }
};

// Staged epilogue (FAST kernels): a wave parks one 32-row slab of its finished bf16 tile in LDS (row pitch P bytes) and
// writes it out row-contiguously: LPR lanes cover one row's LPR*16 bytes, so a store instruction touches 64/LPR rows with
// whole 64..512-byte runs instead of 32 rows with 32 bytes each (the accumulator layout gives a lane 16 channels of ONE
// row, and the texture path spends a cycle per distinct cache line of an instruction, not per byte).
template <int P, int LPR>
__device__ __forceinline__ void flush_slab(const char* st, int lane, bf16_t* obase, int ldo, int gm0, int M, int col0, int n_out) {
    static_assert((32 * LPR) % 64 == 0, "slab pieces per wave instruction");
#pragma unroll
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
        const int q = it * 64 + lane;
        const int r = q / LPR, c = q - r * LPR;
        const uint4 val = *(const uint4*)(st + r * P + c * 16);
        const int gm = gm0 + r, ch = col0 + c * 8;
        if (gm < M && ch < n_out) *(uint4*)(obase + (long long)gm * ldo + ch) = val;
    }
}
// ablation bit 32: same slab traffic, but the global stores sit behind a condition no finite tile meets
template <int P, int LPR>
__device__ __forceinline__ void flush_slab_nostore(const char* st, int lane, bf16_t* obase, int ldo, int gm0, int M, int col0, int n_out) {
#pragma unroll
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
        const int q = it * 64 + lane;
        const int r = q / LPR, c = q - r * LPR;
        const uint4 val = *(const uint4*)(st + r * P + c * 16);
        const int gm = gm0 + r, ch = col0 + c * 8;
        if (gm < M && ch < n_out && val.x == 0x7fc17fc1u && val.y == 0x7fc27fc2u) *(uint4*)(obase + (long long)gm * ldo + ch) = val;
    }
}
// dynamic LDS of a kernel instantiation: the DMA ring, or the epilogue slabs of all waves if those need more
template <int BM, int BN, int WM, int WN, int STAGES, int BK, bool FAST>
constexpr int gemm_smem_bytes() {  // (the same for the DMA-staged and the register-staged kernels)
    const int ring = STAGES * (BM + BN) * BK * 2;
#ifndef T2V_GEMM_DIRECT_EPI
    const int slabs = FAST ? WM * WN * 32 * ((BN / WN) * 2 + 16) : 0;
    return slabs > ring ? slabs : ring;
#else
    return ring;
#endif
}

// BK = K elements per pipeline step (LDS rows of BK*2 bytes); WPE = waves per SIMD the register budget is
// sized for (2 x 4-wave workgroups or one 8-wave workgroup per CU at WPE = 2)
// RS ("register-staged", experimental): operands travel global -> VGPR -> ds_write_b128 -> LDS instead of by LDS-DMA.  One
// register set per wave holds the NEXT K step while the current one is computed; the LDS ring has exactly two slots.
// LNOUT (FAST kernels whose workgroup tile spans the whole row, i.e. the 160x320 tiles at N = 320): the epilogue also writes
// LayerNorm(out) * gamma + beta to a second tensor — the transformer blocks' LayerNorms (attention.py:300-311) as a by-product
// of the GEMM that produces their input, instead of 60 read-modify-write launches per UNet step.
// FUSE (fast kernels only; instantiated in gemm_fuse.hip): 1 = row statistics of the output (t2v_gemm_desc::rowstat_out), 2 = column
// statistics per 32-row slab (colstat_out), 4 = this GEMM consumes a LayerNorm folded into its weights (lnf_*).
template <int BM, int BN, int WM, int WN, int STAGES, int BK = 64, int WPE = (WM * WN) / 4, bool FAST = false, bool RS = false, bool LNOUT = false,
          int FUSE = 0>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void gemm_kernel(const GemmParams p) {
    static_assert(FUSE == 0 || (FAST && !RS && !LNOUT), "fused statistics: fast DMA-staged kernels only");
    constexpr bool F_ROW = (FUSE & 1) != 0, F_COL = (FUSE & 2) != 0, F_LNF = (FUSE & 4) != 0;
    // F_DRP: the LoRA up-projection's dropout epilogue on the fast kernels: out = residual + dropout(acc + bias).  The residual cannot
    // start the accumulators here (the mask applies to the product only): each 32-row slab's residual runs are loaded at the top of
    // its epilogue pass and land under the mask arithmetic (~35 VALU per column pair).  The generic kernels, which carried every
    // dropout launch before, store 32-byte runs per lane and row.
    constexpr bool F_DRP = (FUSE & 8) != 0;
    // F_LRA: the LoRA up-projection (rank 64) of the leaf as an MFMA phase of the epilogue: see t2v_gemm_desc::lora_t
    constexpr bool F_LRA = (FUSE & 16) != 0;
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWB = BK * 2;       // bytes per LDS row
    constexpr int CPR = ROWB / 16;     // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;      // tile rows filled by one wave-wide 1 KiB DMA
    constexpr int A_IT = BM / RPI / NW, B_IT = BN / RPI / NW;  // DMA wave-instructions per wave per stage
    constexpr int LOADS = A_IT + B_IT;
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "every wave must own a whole number of 1 KiB DMA row groups");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiles are whole 32x32 MFMA tiles");
    static_assert(LOADS * (STAGES - 1) < 64, "vmcnt field");
    static_assert(!RS || STAGES == 2, "register-staged kernels use a two-slot ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const t2v_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar LDS/M0 math
    const int wave_m = wave / WN, wave_n = wave % WN;

    // ---- XCD-aware tile assignment (bijective form) -------------------------------------------
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // `tile` walks the tile grid block by block (xcd_m x xcd_n rectangular blocks, n fastest inside a block), so the
    // contiguous run an XCD owns is (about) one block: its private L2 then fetches 1/xcd_m of the activations and
    // 1/xcd_n of the weights instead of all of one operand.  The host picks the split that minimises the bytes the
    // eight L2s pull in total (xcd_n * A_bytes + xcd_m * W_bytes).
    int tile_m, tile_n;
    {   // block table precomputed on the host: a scalar scan instead of per-block integer divisions
        int b = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (i < p.nblk && tile >= p.blk_start[i]) b = i;
        const int t = tile - p.blk_start[b], bw = p.blk_w[b];
        const int q = t / bw;
        tile_m = p.blk_r0[b] + q;
        tile_n = p.blk_c0[b] + t - q * bw;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- batch / split offsets --------------------------------------------------------------------
    const int z = blockIdx.y;
    const int z0 = z / d.batch_inner, z1 = z % d.batch_inner;
    const bf16_t* a0 = (const bf16_t*)d.a0 + z0 * d.a_stride0 + z1 * d.a_stride1;
    const bf16_t* a1 = d.a1 ? (const bf16_t*)d.a1 + z0 * d.a_stride0 + z1 * d.a_stride1 : nullptr;
    const bf16_t* wbase = (const bf16_t*)d.w + z0 * d.w_stride0 + z1 * d.w_stride1;
    const long long o_off = z0 * d.o_stride0 + z1 * d.o_stride1;
    const bf16_t* zero = (const bf16_t*)p.zero;
    const int split = blockIdx.z;
    const int kt_begin = split * p.nk_per_split;
    const int kt_end = min(p.nk, kt_begin + p.nk_per_split);
    const int nk = kt_end - kt_begin;

    // ---- per-lane DMA row bookkeeping -----------------------------------------------------------
    // wave-instruction i (= wave + NW*j) fills tile rows [RPI*i, RPI*(i+1)): lane -> row RPI*i + lane/CPR,
    // 16-byte slot lane%CPR of that row, which holds source chunk slot ^ swz(row)
    // (swz(row) = (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte rows: conflict-free ds_read_b128).
    auto swz_of = [](int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    const int H = p.gh, W = p.gw;
    int a_n[A_IT], a_y[A_IT], a_x[A_IT];  // decomposed output coordinates (a_n < 0: row >= M)
    int a_chunk[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int r = (wave + NW * j) * RPI + lane / CPR;
        a_chunk[j] = ((lane % CPR) ^ swz_of(r)) * 8;
        const int m = m0 + r;
        if (m >= d.M) {
            a_n[j] = -1; a_y[j] = 0; a_x[j] = 0;
        } else if (p.taps == 1 && p.h_out == 1) {  // LINEAR: the grid is (M, 1, 1), no divisions
            a_n[j] = m; a_y[j] = 0; a_x[j] = 0;
        } else {
            const int hw_o = p.h_out * p.w_out;
            const int n = m / hw_o, rem = m - n * hw_o;
            a_n[j] = n; a_y[j] = rem / p.w_out; a_x[j] = rem - a_y[j] * p.w_out;
        }
    }
    const bf16_t* aptr[A_IT];
    int ainc[A_IT];
    const bf16_t* wptr[B_IT];
    int winc[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int r = (wave + NW * j) * RPI + lane / CPR;
        const int chunk = ((lane % CPR) ^ swz_of(r)) * 8;
        // LDS row r of the weight tile holds channel perm(r): within each 32-row MFMA tile the rows a
        // lane's 16 accumulator registers cover ((q&3) + 8*(q>>2) + 4*hi) are fed 16 CONSECUTIVE
        // channels (16*hi + q), so the epilogue moves 32-byte runs per lane, 64 bytes per row
        const int r32 = r & 31;
        const int n = n0 + (r & ~31) + ((((r32 >> 2) & 1) << 4) | ((r32 >> 3) << 2) | (r32 & 3));
        if (n < d.N) { wptr[j] = wbase + (long long)n * d.ldw + chunk + (long long)kt_begin * BK; winc[j] = BK; }
        else { wptr[j] = zero; winc[j] = 0; }
    }

    // staging iterator: segment = (tap, source); starts at K step kt_begin
    const int steps0 = d.c0 / BK, steps1 = d.c1 / BK, steps_tap = steps0 + steps1;
    int seg, seg_left, seg_skip;
    {
        const int tap = kt_begin / steps_tap, rem = kt_begin - tap * steps_tap;
        const int src = rem >= steps0 ? 1 : 0;
        seg = tap * p.nsrc + src;
        seg_skip = src ? rem - steps0 : rem;
    }
    auto begin_segment = [&]() {
        const int tap = seg / p.nsrc, src = seg - tap * p.nsrc;
        const bf16_t* base = src ? a1 : a0;
        const int ld = src ? d.lda1 : d.lda0;
        seg_left = (src ? steps1 : steps0) - seg_skip;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int uy = a_y[j] * p.stride + ky - p.pad_y, ux = a_x[j] * p.stride + kx - p.pad_x;
            const bool ok = a_n[j] >= 0 && uy >= 0 && uy < (H << p.ups) && ux >= 0 && ux < (W << p.ups);
            const long long row = ((long long)a_n[j] * H + (uy >> p.ups)) * W + (ux >> p.ups);
            if (ok) { aptr[j] = base + row * ld + a_chunk[j] + seg_skip * BK; ainc[j] = BK; }
            else { aptr[j] = zero; ainc[j] = 0; }
        }
        seg_skip = 0;
    };
    // DMAs of one K step into ring slot `slot` + pointer advance (segment bookkeeping stays OUTSIDE the hot loop)
    auto issue = [&](int slot) {
        char* sa = smem + slot * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) dma16(aptr[j], sa + (wave + NW * j) * 1024);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) dma16(wptr[j], sb + (wave + NW * j) * 1024);
#pragma unroll
        for (int j = 0; j < A_IT; ++j) aptr[j] += ainc[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) wptr[j] += winc[j];
    };
    // register-staged twin of issue(): gload() pulls the step the pointers stand at into this wave's register set,
    // gstore(slot) parks that set in ring slot `slot` at exactly the addresses the DMAs would have written
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));  // (an array of the uint4 struct is not promoted to registers)
    u32x4_t greg[RS ? LOADS : 1];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) greg[j] = *(const u32x4_t*)aptr[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) greg[A_IT + j] = *(const u32x4_t*)wptr[j];
#pragma unroll
        for (int j = 0; j < A_IT; ++j) aptr[j] += ainc[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) wptr[j] += winc[j];
    };
    auto gstore = [&](int slot) {
        char* sa = smem + slot * STAGE_BYTES + lane * 16;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) *(u32x4_t*)(sa + (wave + NW * j) * 1024) = greg[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) *(u32x4_t*)(sb + (wave + NW * j) * 1024) = greg[A_IT + j];
    };
    auto next_segment_if_done = [&]() {
        if (seg_left == 0) { ++seg; if (seg < p.taps * p.nsrc) begin_segment(); }
    };

    // ---- fragment read addressing ---------------------------------------------------------------
    const int frow = lane & 31, hi = lane >> 5, swz = swz_of(frow);
    const int ch_lane = n0 + wave_n * WTN + 16 * hi;
    f32x16_t acc[TM][TN];

    // ---- software-pipelined K loop ------------------------------------------------------------------
    // Ring of STAGES slots, all filled by the prologue.  Fragments are double-buffered in registers (fa/fw[2]):
    // the ds_read_b128s of K-slice kk+1 are issued BEFORE the MFMAs of slice kk and land under them.  The
    // hand-over to the next K step happens before the LAST slice's MFMAs: wait for my own DMAs of step kt+1
    // (counted vmcnt: YOUNGER later slots stay in flight), s_barrier (now step kt+1 is visible from every wave
    // and slot kt is drained by every wave), slice-0 fragments of step kt+1 are read, the DMAs of step kt+STAGES
    // go out into the freed slot interleaved with the last slice's MFMAs.  So neither LDS latency nor DMA issue
    // is ever exposed inside the loop.  Each step body is ONE basic block (segment bookkeeping sits outside):
    // exact counted s_waitcnt and a fixed issue order (sched_barrier fences) depend on that.
    int buf = 0;
    bf16x8_t fa[2][TM], fw[2][TN];
    auto read_frags = [&](int slot, int kk, int which) {
        const char* sa = smem + slot * STAGE_BYTES + (wave_m * WTM + frow) * ROWB;
        const char* sb = smem + slot * STAGE_BYTES + A_BYTES + (wave_n * WTN + frow) * ROWB;
        const int coff = ((kk * 2 + hi) ^ swz) << 4;
        if (ABL(8)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { fa[which][i] = (bf16x8_t){0}; asm volatile("" : "+v"(fa[which][i])); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { fw[which][j] = (bf16x8_t){0}; asm volatile("" : "+v"(fw[which][j])); }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[which][i] = *(const bf16x8_t*)(sa + i * 32 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < TN; ++j) fw[which][j] = *(const bf16x8_t*)(sb + j * 32 * ROWB + coff);
    };
    // MFMAs [first, last) of one K slice in (i, j) order
    auto mfmas = [&](int which, int first, int last) {
        if (ABL(2)) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (i * TN + j >= first && i * TN + j < last)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[which][j], fa[which][i], acc[i][j], 0, 0, 0);
    };
    constexpr int NS = BK / 16;  // K slices (one MFMA deep) per step
    constexpr int NM = TM * TN;  // MFMAs per slice
    // ISSUE: stage step kt+STAGES into the slot this step frees.  NEXT: there is a step kt+1 (hand-over).
    // Issue order per slice: first MFMA (the compiler's wait for this slice's fragments is lgkmcnt(0), exact here
    // because nothing younger is outstanding), THEN the next slice's ds_reads, then the other NM-1 MFMAs which
    // cover the LDS latency.
    auto kstep = [&](auto issue_tag, auto next_tag, auto younger) {
        constexpr bool ISSUE = decltype(issue_tag)::value, NEXT = decltype(next_tag)::value;
        if constexpr (RS) {
            // Step kt computes from slot buf.  Under its first slice the register set (step kt+1, loaded one whole step
            // ago) is written to the other slot - free since every wave passed the previous hand-over barrier - and
            // refilled with step kt+2.  The hand-over barrier then publishes slot kt+1 and retires slot kt.
            const int nxt = buf ^ 1;
#pragma unroll
            for (int kk = 0; kk < NS - 1; ++kk) {
                mfmas(kk & 1, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                read_frags(buf, kk + 1, (kk + 1) & 1);
                if (kk == 0) {
                    if constexpr (NEXT) gstore(nxt);
                    if constexpr (ISSUE) gload();
                }
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kk & 1, 1, NM);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NEXT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my fragment reads of slot kt and my writes of slot kt+1
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                mfmas((NS - 1) & 1, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                read_frags(nxt, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                mfmas((NS - 1) & 1, 0, 1);
            }
            mfmas((NS - 1) & 1, 1, NM);
            __builtin_amdgcn_sched_barrier(0);
            buf = nxt;
            return;
        }
#pragma unroll
        for (int kk = 0; kk < NS - 1; ++kk) {
            mfmas(kk & 1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(buf, kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(kk & 1, 1, NM);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int nxt = (buf + 1 == STAGES) ? 0 : buf + 1;
        if constexpr (NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of this slot have landed
            wait_vmcnt<LOADS * decltype(younger)::value>();       // my DMAs of step kt+1 have landed
            if (!ABL(16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mfmas((NS - 1) & 1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(nxt, 0, 0);  // NS is even: slice 0 always lives in fragment set 0
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mfmas((NS - 1) & 1, 0, 1);
        }
        if constexpr (ISSUE) {
            if (!ABL(1)) issue(buf);
        }
        mfmas((NS - 1) & 1, 1, NM);
        if constexpr (ISSUE) {  // spread the DMA issue over the last slice's MFMAs
            constexpr int PER = (LOADS + NM - 2) / (NM - 1);
#pragma unroll
            for (int q = 0; q < NM - 1; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, PER, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        buf = nxt;
    };
    using std::integral_constant;
    constexpr integral_constant<bool, true> kYes{};
    constexpr integral_constant<bool, false> kNo{};

    // prologue: fill the whole ring (steps 0 .. STAGES-1)
    begin_segment();
    int staged = 0;
    if constexpr (RS) {  // step 0 on its way to the registers (parked in slot 0 below, after the accumulator-init loads went out)
        gload(); --seg_left; ++staged; if (staged < nk) next_segment_if_done();
    } else {
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (s < nk) { issue(s); --seg_left; ++staged; if (staged < nk) next_segment_if_done(); }
    }
    // FAST kernels: everything linear in the epilogue (bias, and without a gate the time-embedding row vector and the
    // residual tile) becomes the accumulators' initial value.  The loads are issued right AFTER the DMA prologue, so
    // their latency overlaps the first tiles' flight, and they are the YOUNGEST vector-memory operations when consumed:
    // the wait is then vmcnt(0).  (Issuing them before the DMAs made the compiler wait with a counted vmcnt over the
    // younger LDS-DMAs; LDS-DMA and register loads do not retire in order with each other, and one run in four of the
    // 4-step pipeline came out NaN.)  The counted waits of the K loop only ever see LDS-DMAs.
    constexpr bool FOLD = FAST;
    const bool fold_rr = FOLD && d.act != T2V_ACT_GEGLU && p.splits == 1;
    uint4 rinit[FOLD ? TM : 1][FOLD ? TN : 1][2];
    float4 binit[FOLD ? TN : 1][4];
    if constexpr (FOLD) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ch = ch_lane + j * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                binit[j][q] = (!F_LNF && d.bias && ch < d.N && !ABL(64)) ? *(const float4*)(d.bias + ch + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ch = ch_lane + j * 32;
                rinit[i][j][0] = rinit[i][j][1] = uint4{0, 0, 0, 0};
                if (fold_rr && !F_DRP && d.residual && gm < d.M && ch < d.N && !ABL(64)) {
                    const bf16_t* rp = (const bf16_t*)d.residual + o_off + (long long)gm * d.ldr + ch;
                    rinit[i][j][0] = *(const uint4*)rp;
                    rinit[i][j][1] = *(const uint4*)(rp + 8);
                }
            }
        }
    }
    // LayerNorm fold: (rstd, rstd * mean) of this lane's TM rows from the producer's per-32-column (sum, sum of squares) pairs,
    // added in block order (deterministic); loaded here, with the other initial register loads, so that the main loop hides them
    float lnr[F_LNF ? TM : 1], lnrm[F_LNF ? TM : 1];
    if constexpr (F_LNF) {
        // a row's partials are lnf_nblk (sum, sumsq) pairs = nq float4; the two lanes that own a row (hi = 0 / 1) take one half
        // each — at most LNQ 16-byte loads, all issued before the first add — and exchange their sums (lane ^ 32)
        constexpr int LNQ = 10;  // C <= 1280
        const float inv_c = 1.0f / (float)(d.lnf_nblk * 32);
        const int nq = d.lnf_nblk >> 1, halfq = (nq + 1) >> 1;
        const int q0 = hi * halfq, q1 = min(nq, q0 + halfq);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
            const float4* sp = (const float4*)(d.lnf_stats + (long long)min(gm, d.M - 1) * d.lnf_ld);
            float4 t[LNQ];
#pragma unroll
            for (int k = 0; k < LNQ; ++k) t[k] = (q0 + k < q1) ? sp[q0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < LNQ; ++k) { s1 += t[k].x; s2 += t[k].y; s1 += t[k].z; s2 += t[k].w; }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float mean = s1 * inv_c;
            const float var = fmaxf(s2 * inv_c - mean * mean, 0.f);
            lnr[i] = rsqrtf(var + d.lnf_eps);
            lnrm[i] = lnr[i] * mean;
        }
    }
    // accumulators start at bias (+ row vector + residual) in FAST kernels, at zero otherwise
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int gm = m0 + wave_m * WTM + i * 32 + frow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float init[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) init[e] = 0.f;
            if constexpr (FOLD) {
                const int ch = ch_lane + j * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    init[4 * q] = binit[j][q].x; init[4 * q + 1] = binit[j][q].y;
                    init[4 * q + 2] = binit[j][q].z; init[4 * q + 3] = binit[j][q].w;
                }
                if (fold_rr && !F_DRP) {
                    float rf[16];
                    unpack8(rinit[i][j][0], rf);
                    unpack8(rinit[i][j][1], rf + 8);
#pragma unroll
                    for (int e = 0; e < 16; ++e) init[e] += rf[e];
                    if (d.rowvec && gm < d.M && ch < d.N) {
                        const float* rv = d.rowvec + (long long)(gm / d.rowvec_div) * d.ld_rowvec + ch;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 r4 = *(const float4*)(rv + 4 * q);
                            init[4 * q] += r4.x; init[4 * q + 1] += r4.y; init[4 * q + 2] += r4.z; init[4 * q + 3] += r4.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = init[e];
        }
    }
    if constexpr (RS) {
        gstore(0);
        if (nk > 1) { gload(); --seg_left; ++staged; if (staged < nk) next_segment_if_done(); }  // step 1 stays in registers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        if (nk >= STAGES) wait_vmcnt<LOADS*(STAGES - 1)>();
        else wait_vmcnt<0>();
    }
    if (!ABL(16)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL(128)) return;
    read_frags(0, 0, 0);
    // steady state: per (tap, source) segment a branch-free run of steps, each computing step kt and staging kt+STAGES
    int remaining = nk - staged;
    while (remaining > 0) {
        const int n = min(seg_left, remaining);
        for (int i = 0; i < n; ++i) kstep(kYes, kYes, integral_constant<int, STAGES - 2>{});
        remaining -= n;
        seg_left -= n;
        if (remaining > 0) next_segment_if_done();
    }
    // drain: the last min(nk, STAGES) steps have nothing left to stage; step t of them leaves tail-2-t slots in flight
    const int tail = min(nk, STAGES);
    if constexpr (STAGES >= 4) { if (tail >= 4) kstep(kNo, kYes, integral_constant<int, 2>{}); }
    if constexpr (STAGES >= 3) { if (tail >= 3) kstep(kNo, kYes, integral_constant<int, 1>{}); }
    if (tail >= 2) kstep(kNo, kYes, integral_constant<int, 0>{});
    kstep(kNo, kNo, integral_constant<int, 0>{});

    // ---- epilogue straight from the accumulators: lane = token (frow), regs = 4-channel runs -------
    if (ABL(4) && acc[0][0][0] != 12345.678f) return;
    if (!FAST && p.splits > 1) {  // raw fp32 partial slab (FAST kernels are never split: host-checked); the reduce kernel applies the epilogue
        float* ws = p.ws + ((long long)(z * p.splits + split) * d.M) * d.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
            if (gm >= d.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = ch_lane + j * 32 + 4 * q;
                    if (ch < d.N)
                        *(float4*)(ws + (long long)gm * d.N + ch) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
        }
        return;
    }
    Epi epi;
    epi.d = &d; epi.o_off = o_off; epi.vec = p.vec4;
#ifndef T2V_GEMM_DIRECT_EPI
    if constexpr (FAST) {  // host-checked: bf16 output, full 16-channel runs
        constexpr int P = WTN * 2 + 16;  // +16: consecutive rows start 4 banks apart, 16 lanes x 16 bytes cover all 64
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the ring is dead
        char* st = smem + wave * (32 * P);
        char* st_w = st + frow * P + hi * 32;
        bf16_t* obase = (bf16_t*)d.out + o_off;
        // LayerNorm fold: out = rstd * (acc - mean * s[n]) + t[n] (W holds W diag(gamma), s its row sums, bias = t = b + W beta;
        // value and gate columns of a GEGLU projection alike).  s / t of a 32-column block depend on the column only: wave tiles
        // with several row slabs load them ONCE (the loads of slab i+1 cannot move above the stores of slab i: possible alias).
        constexpr bool LN_HOIST = F_LNF && TM > 1;
        float4 lns[LN_HOIST ? TN : 1][4], lnt[LN_HOIST ? TN : 1][4];
        if constexpr (LN_HOIST) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = min(ch_lane + j * 32 + 4 * q, d.N - 4);
                    lns[j][q] = *(const float4*)(d.lnf_s + ch);
                    lnt[j][q] = *(const float4*)(d.bias + ch);
                }
        }
        auto ln_fold = [&](float* v, int j, float r, float rmu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 s4, t4;
                if constexpr (LN_HOIST) { s4 = lns[j][q]; t4 = lnt[j][q]; }
                else {
                    const int ch = min(ch_lane + j * 32 + 4 * q, d.N - 4);
                    s4 = *(const float4*)(d.lnf_s + ch);
                    t4 = *(const float4*)(d.bias + ch);
                }
                v[4 * q] = fmaf(r, v[4 * q], fmaf(-rmu, s4.x, t4.x));
                v[4 * q + 1] = fmaf(r, v[4 * q + 1], fmaf(-rmu, s4.y, t4.y));
                v[4 * q + 2] = fmaf(r, v[4 * q + 2], fmaf(-rmu, s4.z, t4.z));
                v[4 * q + 3] = fmaf(r, v[4 * q + 3], fmaf(-rmu, s4.w, t4.w));
            }
        };
        if (d.act == T2V_ACT_GEGLU) {
            epi.n_out = d.N / 2;
            if constexpr (TN >= 2 && TN % 2 == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int gm = m0 + wave_m * WTM + i * 32 + frow;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int u = 0; u < TN / 2; ++u) {
                        float v[16], gt[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) { v[e] = acc[i][2 * u][e]; gt[e] = acc[i][2 * u + 1][e]; }
                        const int ch_out = (n0 + wave_n * WTN) / 2 + u * 32 + 16 * hi;
                        if constexpr (F_LNF) { ln_fold(v, 2 * u, lnr[i], lnrm[i]); ln_fold(gt, 2 * u + 1, lnr[i], lnrm[i]); }
                        if (gm < d.M && ch_out < epi.n_out) epi.template compute<true>(v, gt, gm, ch_lane + u * 64, ch_out);
                        *(uint4*)(st_w + u * 64) = pack8(v);
                        *(uint4*)(st_w + u * 64 + 16) = pack8(v + 8);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ABL(32)) flush_slab_nostore<P, TN * 2>(st, lane, obase, d.ldo, m0 + wave_m * WTM + i * 32, d.M, (n0 + wave_n * WTN) / 2, epi.n_out);
                    else flush_slab<P, TN * 2>(st, lane, obase, d.ldo, m0 + wave_m * WTM + i * 32, d.M, (n0 + wave_n * WTN) / 2, epi.n_out);
                }
            }
            return;
        }
        epi.n_out = d.N;
        if constexpr (LNOUT) {
            static_assert(FAST && TM == 1 && WN == 2, "LNOUT: one 32-row MFMA tile row per wave, two waves per output row");
            // (host-checked: one N tile, i.e. BN == N, no batch, no activation)
            const int gm = m0 + wave_m * WTM + frow;
            float vals[TN][16];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int e = 0; e < 16; ++e) vals[j][e] = acc[0][j][e];
                if (gm < d.M) epi.template compute<true>(vals[j], nullptr, gm, ch_lane + j * 32, ch_lane + j * 32);
                *(uint4*)(st_w + j * 64) = pack8(vals[j]);
                *(uint4*)(st_w + j * 64 + 16) = pack8(vals[j] + 8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            flush_slab<P, TN * 4>(st, lane, obase, d.ldo, m0 + wave_m * WTM, d.M, n0 + wave_n * WTN, d.N);
            // row statistics over the BN columns: this lane's TN*16 values, its partner lane (other 16-channel half of every
            // 32-column tile: lane ^ 32) and the partner wave (the other WTN columns: wave ^ 1); two-pass (mean, then centred)
            float* xch = (float*)(smem + NW * 32 * P);  // [2][NW][32] floats behind the slabs
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += vals[j][e];
            sum += __shfl_xor(sum, 32, 64);
            if (hi == 0) xch[wave * 32 + frow] = sum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const float mean = (sum + xch[(wave ^ 1) * 32 + frow]) * (1.0f / (float)BN);
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float dl = vals[j][e] - mean; sq += dl * dl; }
            sq += __shfl_xor(sq, 32, 64);
            if (hi == 0) xch[NW * 32 + wave * 32 + frow] = sq;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const float rstd = rsqrtf((sq + xch[NW * 32 + (wave ^ 1) * 32 + frow]) * (1.0f / (float)BN) + d.ln_eps);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ch = ch_lane + j * 32;
                float o[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 g4 = *(const float4*)(d.ln_gamma + ch + 4 * q4), b4 = *(const float4*)(d.ln_beta + ch + 4 * q4);
                    o[4 * q4] = (vals[j][4 * q4] - mean) * rstd * g4.x + b4.x;
                    o[4 * q4 + 1] = (vals[j][4 * q4 + 1] - mean) * rstd * g4.y + b4.y;
                    o[4 * q4 + 2] = (vals[j][4 * q4 + 2] - mean) * rstd * g4.z + b4.z;
                    o[4 * q4 + 3] = (vals[j][4 * q4 + 3] - mean) * rstd * g4.w + b4.w;
                }
                *(uint4*)(st_w + j * 64) = pack8(o);
                *(uint4*)(st_w + j * 64 + 16) = pack8(o + 8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            flush_slab<P, TN * 4>(st, lane, (bf16_t*)d.ln_out, d.ld_ln_out, m0 + wave_m * WTM, d.M, n0 + wave_n * WTN, d.N);
            return;
        }
        // LoRA epilogue operands.  U (the up-projection rows of this workgroup tile's BN columns, 128 B each) is staged ONCE per tile
        // in LDS behind the slabs — round 3 fetched every 32-column block's fragments from global memory again for every 32-row slab
        // of every wave: 4 bytes of L2 traffic per output element, which is what the epilogue's +0.6-0.9 us per million outputs were
        // (profiles/r04_student_gemm_shapes.csv).  16-byte chunk c of row r sits at chunk c ^ f(r), f(r) = ((r >> 1) & 3) | 4 ((r >> 4) & 1):
        // the 16 lanes of a ds_read_b128 pass (rows in the accumulator's channel order, below) fall into 16 different bank groups.
        // The rank-64 rows t of a 32-row slab (B fragments: column = token) come straight from global memory, slab i + 1's while
        // slab i is worked on; slab 0's are requested BEFORE the staging so that the two latencies overlap.
        char* const lra_lds = smem + NW * 32 * P;
        auto lra_swz = [](int r) { return ((r >> 1) & 3) | (((r >> 4) & 1) << 2); };
        auto lra_load_t = [&](int i, bf16x8_t (*dst)[4], int& leaf0) {
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
            leaf0 = (n0 + wave_n * WTN) / d.lora_n_leaf;
            const int last = min(n0 + wave_n * WTN + WTN - 1, d.N - 1) / d.lora_n_leaf;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16_t* tp = (const bf16_t*)d.lora_t + (long long)min(gm, d.M - 1) * d.ld_lora_t + (leaf0 + q) * 64 + hi * 8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    uint4 u = make_uint4(0, 0, 0, 0);
                    if (gm < d.M && leaf0 + q <= last && !ABL(1024)) u = *(const uint4*)(tp + ks * 16);
                    dst[q][ks] = *(bf16x8_t*)&u;
                }
            }
        };
        // (wave tiles whose accumulators fill half the register file have no room for a second set of t rows: they fetch per slab)
        constexpr bool LRA_T2 = F_LRA && TM > 1 && TM * TN * 16 < 128 && WPE < 4;
        bf16x8_t lra_tb[LRA_T2 ? 2 : 1][2][4];
        int lra_leaf0 = 0;
        if constexpr (F_LRA) {
            lra_load_t(0, lra_tb[0], lra_leaf0);
            for (int idx = tid; idx < BN * 8 && !ABL(2048); idx += NW * 64) {
                const int r = idx >> 3, c = idx & 7, n = n0 + r;
                uint4 u = make_uint4(0, 0, 0, 0);
                if (n < d.N) u = *(const uint4*)((const bf16_t*)d.lora_u + (long long)n * d.ld_lora_u + c * 8);
                *(uint4*)(lra_lds + r * 128 + ((c ^ lra_swz(r)) << 4)) = u;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // a 32-column block's U rows as A fragments (rows in the accumulator's channel order: MFMA row r <-> channel
        // 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3) of the block, the permutation the weight tile's LDS rows carry), 4 K steps of 16
        auto lra_load_u = [&](int j, bf16x8_t* dst) {
            const int cperm = (((frow >> 2) & 1) << 4) | ((frow >> 3) << 2) | (frow & 3);
            const int r = wave_n * WTN + j * 32 + cperm;
            const char* up = lra_lds + r * 128;
            const int sw = lra_swz(r);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const bf16x8_t*)(up + (((2 * ks + hi) ^ sw) << 4));
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
            asm volatile("" ::: "memory");
            float rs1[F_ROW ? TN : 1], rs2[F_ROW ? TN : 1];
            // this slab's rank-64 rows t; a wave tile spans at most two leaves of a group (host-checked)
            bf16x8_t (*lra_t)[4] = lra_tb[LRA_T2 ? (i & 1) : 0];
            if constexpr (LRA_T2) {
                int unused;
                if (i + 1 < TM) lra_load_t(i + 1, lra_tb[(i + 1) & 1], unused);
            } else if constexpr (F_LRA && TM > 1) {
                int unused;
                if (i > 0) lra_load_t(i, lra_tb[0], unused);
            }
            bf16x8_t lra_u[2][4];
            if constexpr (F_LRA) lra_load_u(0, lra_u[0]);
            uint4 rdrp[F_DRP ? TN : 1][2];
            if constexpr (F_DRP) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    rdrp[j][0] = rdrp[j][1] = uint4{0, 0, 0, 0};
                    if (d.residual && gm < d.M && ch_lane + j * 32 < d.N) {
                        const bf16_t* rp = (const bf16_t*)d.residual + o_off + (long long)gm * d.ldr + ch_lane + j * 32;
                        rdrp[j][0] = *(const uint4*)rp;
                        rdrp[j][1] = *(const uint4*)(rp + 8);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[i][j][e];
                if constexpr (F_LNF) ln_fold(v, j, lnr[i], lnrm[i]);
                if constexpr (F_LRA) {   // v += lora_scale * dropout(t u^T): four MFMAs over the rank, mask on the product only
                    f32x16_t lp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) lp[e] = 0.f;
                    const int q = (n0 + wave_n * WTN + j * 32) / d.lora_n_leaf - lra_leaf0;
                    if (j + 1 < TN) lra_load_u(j + 1, lra_u[(j + 1) & 1]);
                    if (!ABL(512)) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            lp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lra_u[j & 1][ks], q ? lra_t[1][ks] : lra_t[0][ks], lp, 0, 0, 0);
                    }
                    if (gm < d.M && ch_lane + j * 32 < d.N) {
                        if (d.drop_thr) {
                            // (drawing the keep bits in the prologue instead — under the DMA flight, parked in LDS — was tried and is
                            // slower: 32.2 vs 30.7 us at 40960 x 320 x 320; the workgroup's waves reach the first barrier later)
                            const uint64_t key = dropout_key(*(const uint64_t*)d.drop_seed, d.drop_site);
                            const uint64_t quad0 = ((uint64_t)gm * (uint64_t)d.drop_ncols + (uint64_t)(d.drop_col0 + ch_lane + j * 32)) >> 2;
                            const uint32_t keep = ABL(256) ? 0xffffu : dropout_keep_mask<4>(key, quad0, d.drop_thr >> 16);
                            const float sk = d.lora_scale * d.drop_inv_keep;
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] = ((keep >> e) & 1u) ? fmaf(sk, lp[e], v[e]) : v[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] = fmaf(d.lora_scale, lp[e], v[e]);
                        }
                    }
                }
                if constexpr (F_DRP) {   // keep(row, col) = the counter-based mask of t2v_dropout_bf16, then the residual tile
                    if (gm < d.M && ch_lane + j * 32 < d.N) {
                        const uint64_t key = dropout_key(*(const uint64_t*)d.drop_seed, d.drop_site);
                        const uint64_t quad0 = ((uint64_t)gm * (uint64_t)d.drop_ncols + (uint64_t)(d.drop_col0 + ch_lane + j * 32)) >> 2;
                        const uint32_t keep = dropout_keep_mask<4>(key, quad0, d.drop_thr >> 16);
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * d.drop_inv_keep : 0.f;
                        float rf[16];
                        unpack8(rdrp[j][0], rf);
                        unpack8(rdrp[j][1], rf + 8);
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] += rf[e];
                    }
                }
                if (gm < d.M && ch_lane + j * 32 < d.N) epi.template compute<true>(v, nullptr, gm, ch_lane + j * 32, ch_lane + j * 32);
                if constexpr (F_ROW) {
                    // (sum, sum of squares) of this row over the 32-column block j of the wave tile: the lane's 16 fp32 epilogue
                    // values plus its partner's (lane ^ 32 holds the block's other 16 channels), in a fixed order
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
                    rs1[j] = s1 + __shfl_xor(s1, 32, 64);
                    rs2[j] = s2 + __shfl_xor(s2, 32, 64);
                }
                *(uint4*)(st_w + j * 64) = pack8(v);
                *(uint4*)(st_w + j * 64 + 16) = pack8(v + 8);
            }
            if constexpr (F_ROW) {  // one store per PAIR of blocks where the wave tile starts on an even block (16-byte aligned)
                float* rp = d.rowstat_out + (long long)gm * d.ld_rowstat + 2 * ((n0 + wave_n * WTN) >> 5);
                if (hi == 0 && gm < d.M) {
                    if constexpr (TN % 2 == 0 && WTN % 64 == 0) {
#pragma unroll
                        for (int j = 0; j < TN; j += 2) {
                            if (ch_lane + j * 32 + 32 < d.N) *(float4*)(rp + 2 * j) = make_float4(rs1[j], rs2[j], rs1[j + 1], rs2[j + 1]);
                            else if (ch_lane + j * 32 < d.N) *(float2*)(rp + 2 * j) = make_float2(rs1[j], rs2[j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (ch_lane + j * 32 < d.N) *(float2*)(rp + 2 * j) = make_float2(rs1[j], rs2[j]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (F_COL) {
                // (sum, sum of squares) per column over the slab's 32 rows, from the bf16 values parked in LDS (what the consumer
                // will read): lane = (column pair, row half); the two halves meet through lane ^ 32.  host-checked: M % 32 == 0
                const int gm0 = m0 + wave_m * WTM + i * 32, col0 = n0 + wave_n * WTN;
                const int rh = hi * 16;
#pragma unroll
                for (int c2 = 0; c2 < (WTN / 2 + 31) / 32; ++c2) {
                    const int cpair = c2 * 32 + frow;
                    float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
                    if (cpair < WTN / 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t w2 = *(const uint32_t*)(st + (rh + r) * P + cpair * 4);
                            const float x0 = __uint_as_float(w2 << 16), x1 = __uint_as_float(w2 & 0xffff0000u);
                            a0 += x0; q0 = fmaf(x0, x0, q0);
                            a1 += x1; q1 = fmaf(x1, x1, q1);
                        }
                    }
                    a0 += __shfl_xor(a0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
                    a1 += __shfl_xor(a1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
                    if (hi == 0 && cpair < WTN / 2 && gm0 < d.M && col0 + 2 * cpair < d.N)
                        *(float4*)(d.colstat_out + ((long long)(gm0 >> 5) * d.N + col0 + 2 * cpair) * 2) = make_float4(a0, q0, a1, q1);
                }
            }
            if (ABL(32)) flush_slab_nostore<P, TN * 4>(st, lane, obase, d.ldo, m0 + wave_m * WTM + i * 32, d.M, n0 + wave_n * WTN, d.N);
            else flush_slab<P, TN * 4>(st, lane, obase, d.ldo, m0 + wave_m * WTM + i * 32, d.M, n0 + wave_n * WTN, d.N);
        }
        return;
    }
#endif
    if (d.act == T2V_ACT_GEGLU) {
        epi.n_out = d.N / 2;
        if constexpr (TN >= 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gm = m0 + wave_m * WTM + i * 32 + frow;
                if (gm >= d.M) continue;
#pragma unroll
                for (int u = 0; u < TN / 2; ++u) {
                    float v[16], gt[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) { v[e] = acc[i][2 * u][e]; gt[e] = acc[i][2 * u + 1][e]; }
                    epi.template run<FAST>(v, gt, gm, ch_lane + u * 64, (n0 + wave_n * WTN) / 2 + u * 32 + 16 * hi);
                }
            }
        }
        return;
    }
    epi.n_out = d.N;
    // residual tiles are fetched PF rows of MFMA tiles at a time, before any store of that batch: their latency is
    // paid once per batch instead of once per 16-channel run (the compiler cannot hoist a load over the previous run's
    // stores: residual and out may alias).  Register-starved variants (4 waves per SIMD) batch one tile row.
    constexpr int PF = (WPE >= 4) ? 1 : TM;
    const bool pre = d.residual && p.vec4 && !FOLD;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += PF) {
        uint4 rres[PF][TN][2];
        if (pre) {
#pragma unroll
            for (int ii = 0; ii < PF; ++ii) {
                const int gm = m0 + wave_m * WTM + (i0 + ii) * 32 + frow;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int ch = ch_lane + j * 32;
                    rres[ii][j][0] = rres[ii][j][1] = uint4{0, 0, 0, 0};
                    if (gm < d.M && ch + 16 <= d.N) {
                        const bf16_t* rp = (const bf16_t*)d.residual + o_off + (long long)gm * d.ldr + ch;
                        rres[ii][j][0] = *(const uint4*)rp;
                        rres[ii][j][1] = *(const uint4*)(rp + 8);
                    }
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < PF; ++ii) {
            const int i = i0 + ii;
            const int gm = m0 + wave_m * WTM + i * 32 + frow;
            if (gm >= d.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[i][j][e];
                epi.template run<FAST>(v, nullptr, gm, ch_lane + j * 32, ch_lane + j * 32, pre, rres[ii][j][0], rres[ii][j][1]);
            }
        }
    }
}

// split-K: out = epilogue(sum_s ws[s]) in a fixed order; thread = (row, 16 channels)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const t2v_gemm_desc& d = p.d;
    const int nq = (d.N + 15) / 16;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)d.M * nq) return;
    const int gm = (int)(idx / nq), ch = (int)(idx % nq) * 16;
    const int z = blockIdx.y, z0 = z / d.batch_inner, z1 = z % d.batch_inner;
    const float* ws = p.ws + ((long long)z * p.splits * d.M + gm) * d.N + ch;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    for (int s = 0; s < p.splits; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ch + 4 * q < d.N) {
                const float4 t = *(const float4*)(ws + (long long)s * d.M * d.N + 4 * q);
                v[4 * q] += t.x; v[4 * q + 1] += t.y; v[4 * q + 2] += t.z; v[4 * q + 3] += t.w;
            }
        }
    }
    Epi epi;
    epi.d = &d; epi.o_off = z0 * d.o_stride0 + z1 * d.o_stride1; epi.vec = p.vec4; epi.n_out = d.N;
    epi.run(v, nullptr, gm, ch, ch);
}

template <int BM, int BN, int WM, int WN, int STAGES, int BK, int WPE, bool FAST, bool RS = false, bool LNOUT = false, int FUSE = 0>
int launch_impl(GemmParams& p, hipStream_t s);

// does this launch qualify for the fast kernels (accumulators start at bias + row vector + residual, bf16 slab epilogue)?
inline bool gemm_is_fast(const GemmParams& p, bool allow_dropout = false) {
    const int n_out = p.d.act == T2V_ACT_GEGLU ? p.d.N / 2 : p.d.N;
    static const bool no_fast = getenv("T2V_GEMM_NOFAST") != nullptr;  // diagnostics: force the generic epilogue
    return !no_fast && (allow_dropout || !p.d.drop_thr) && p.vec4 && n_out % 16 == 0 && p.d.N % 16 == 0 && p.splits == 1 && p.d.alpha == 1.0f &&
           !p.d.out_f32 && (!p.d.rowvec || ((uintptr_t)p.d.rowvec % 16 == 0 && p.d.ld_rowvec % 4 == 0));
}

template <int BM, int BN, int WM, int WN, int STAGES, int BK = 64, int WPE = (WM * WN) / 4, bool RS = false>
int launch(GemmParams& p, hipStream_t s) {
    const bool fast = gemm_is_fast(p);
    // 128x128 wave tiles keep their 256 accumulator registers in AGPRs and have no room for the generic epilogue's
    // partial-run paths (it spills 2 KiB per lane): shapes that need it run the 8-wave sibling of the same workgroup tile
    if constexpr (BM / WM >= 128 && BN / WN >= 128) {
        return fast ? launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true>(p, s) : launch<BM, BN, WM, WN * 2, 4, 32, 2>(p, s);
    } else {
        return fast ? launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, RS>(p, s) : launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, false, RS>(p, s);
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int BK, int WPE, bool FAST, bool RS, bool LNOUT, int FUSE>
int launch_impl(GemmParams& p, hipStream_t s) {
    p.tiles_m = (p.d.M + BM - 1) / BM;
    p.tiles_n = (p.d.N + BN - 1) / BN;
    {   // XCD grid: minimise xcd_n * A_bytes + xcd_m * W_bytes over the factorizations of 8 that the tile grid allows
        const double a_bytes = (double)p.d.M * (p.d.c0 + p.d.c1), w_bytes = (double)p.d.N * p.K;
        double best = 1e300;
        p.xcd_m = 8; p.xcd_n = 1;
        static const int xcd_policy = getenv("T2V_GEMM_XCD_GRID") ? atoi(getenv("T2V_GEMM_XCD_GRID")) : 2;
        const bool model = xcd_policy == 1 || (xcd_policy == 2 && p.taps != 9);
        for (int xm = 8; xm >= (model ? 1 : 8); xm >>= 1) {
            const int xn = 8 / xm;
            if (xm > p.tiles_m || xn > p.tiles_n) continue;
            const double cost = xn * a_bytes + xm * w_bytes;
            if (cost < best) { best = cost; p.xcd_m = xm; p.xcd_n = xn; }
        }
        if (p.xcd_m > p.tiles_m || p.xcd_n > p.tiles_n) { p.xcd_m = 1; p.xcd_n = 1; }
        p.nblk = 0;
        int start = 0;
        for (int bi = 0; bi < p.xcd_m; ++bi) {
            const int r0 = bi * p.tiles_m / p.xcd_m, r1 = (bi + 1) * p.tiles_m / p.xcd_m;
            for (int bj = 0; bj < p.xcd_n; ++bj) {
                const int c0 = bj * p.tiles_n / p.xcd_n, c1 = (bj + 1) * p.tiles_n / p.xcd_n;
                p.blk_start[p.nblk] = start; p.blk_r0[p.nblk] = r0; p.blk_c0[p.nblk] = c0; p.blk_w[p.nblk] = c1 - c0;
                start += (r1 - r0) * (c1 - c0);
                ++p.nblk;
            }
        }
        for (int i = p.nblk; i < 8; ++i) { p.blk_start[i] = 1 << 30; p.blk_r0[i] = 0; p.blk_c0[i] = 0; p.blk_w[i] = 1; }
    }
    dim3 grid(p.tiles_m * p.tiles_n, p.d.batch, p.splits);
    // LNOUT: the row-statistics exchange area ([2][waves][32] floats) sits behind the epilogue slabs
    constexpr int slab_end = WM * WN * 32 * ((BN / WN) * 2 + 16);
    constexpr int base_smem = gemm_smem_bytes<BM, BN, WM, WN, STAGES, BK, FAST>();
    constexpr int lra_end = slab_end + BN * 128;   // FUSE & 16: the workgroup tile's LoRA up-projection rows behind the slabs
    constexpr int smem = LNOUT ? (base_smem > slab_end + 2 * WM * WN * 32 * 4 ? base_smem : slab_end + 2 * WM * WN * 32 * 4)
                               : ((FUSE & 16) && lra_end > base_smem ? lra_end : base_smem);
    static_assert(smem <= 160 * 1024, "t2v_gemm: LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, STAGES, BK, WPE, FAST, RS, LNOUT, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, STAGES, BK, WPE, FAST, RS, LNOUT, FUSE>), grid, dim3(WM * WN * 64), smem, s, p);
    T2V_CHECK_LAUNCH();
    if (p.splits > 1) {
        const long long work = (long long)p.d.M * ((p.d.N + 15) / 16);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256), p.d.batch), dim3(256), 0, s, p);
        T2V_CHECK_LAUNCH();
    }
    return T2V_OK;
}

}  // namespace

// The compile-verified tile ids (24+) are instantiated in their own translation unit (gemm_exp.hip = this file with
// T2V_GEMM_EXP_ONLY): what the compiler emits for an instantiation depends on what else the unit holds (with all ids in one
// unit the 256x128 / 4-waves-per-SIMD kernel picked up a scratch reload inside its K loop), and the validated ids' code
// should be exactly what ran on hardware.
int t2v_gemm_launch_experimental(int cfg, GemmParams& p, hipStream_t s);
int t2v_gemm_launch_ln(int cfg, GemmParams& p, hipStream_t s);  // the 160x320 tile (id 23) with the LayerNorm second output
// The fast kernels with fused normalisation statistics (FUSE = 1 row statistics out, 2 column statistics out, 4 LayerNorm folded
// in) live in gemm_fuse.hip (= this file with T2V_GEMM_FUSE_ONLY) for the same reason as the experimental ids: the validated
// kernels' code must not depend on them.  t2v_gemm_fuse_tile maps a tile id to the id whose fused variants exist (same workgroup
// tile where possible), 0 if none.
int t2v_gemm_launch_fused(int cfg, int fuse, GemmParams& p, hipStream_t s);
int t2v_gemm_fuse_tile(int cfg, int act, int fuse);
#ifdef T2V_GEMM_FUSE_ONLY
namespace {
template <int BM, int BN, int WM, int WN, int STAGES, int BK, int WPE>
int launch_fused(int fuse, GemmParams& p, hipStream_t s) {
    switch (fuse) {
        case 1: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 1>(p, s);
        case 2: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 2>(p, s);
        case 4: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 4>(p, s);
        case 8: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 8>(p, s);
        case 16: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 16>(p, s);
        case 18: return launch_impl<BM, BN, WM, WN, STAGES, BK, WPE, true, false, false, 18>(p, s);   // + column statistics
        default: return T2V_EINVAL;
    }
}
}  // namespace
// tiles with fused variants: one per workgroup-tile shape the tuned table uses for the UNet's norm producers / consumers
int t2v_gemm_fuse_tile(int cfg, int act, int fuse) {
    if ((fuse == 4 || (fuse & 16)) && cfg == 19) cfg = 7;  // the 4-waves-per-SIMD 256x128 tile has no registers to spare for the fold / the LoRA epilogue: its 2-per-SIMD twin
    switch (cfg) {
        case 1: case 4: case 10: case 18: case 26: case 30: case 33: return 4;     // 128x128
        case 31: case 32: return act == T2V_ACT_GEGLU ? 4 : 31;                   // 128x128, eight waves (32-wide wave tiles: no GEGLU)
        case 6: case 7: case 13: case 16: case 19: case 25: return cfg == 19 ? 19 : (cfg == 16 ? 16 : 7);   // 256x128
        case 11: case 14: case 17: case 29: return cfg == 17 ? 17 : 11;           // 128x256
        case 12: case 15: case 20: case 21: case 24: case 27: return cfg == 20 ? 20 : 12;   // 256x256
        case 22: case 23: case 28: return act == T2V_ACT_GEGLU ? 17 : 23;         // 160x320 (no GEGLU on its 32-wide value / gate split)
        case 3: case 9: return 9;                                                 // 256x64
        case 2: case 5: return 5;                                                 // 128x64
        default: return 0;
    }
}
int t2v_gemm_launch_fused(int cfg, int fuse, GemmParams& p, hipStream_t s) {
    switch (cfg) {
        case 4: return launch_fused<128, 128, 2, 2, 3, 64, 1>(fuse, p, s);
        case 5: return launch_fused<128, 64, 2, 2, 3, 64, 1>(fuse, p, s);
        case 7: return launch_fused<256, 128, 4, 2, 3, 64, 2>(fuse, p, s);
        case 9: return launch_fused<256, 64, 4, 2, 3, 64, 2>(fuse, p, s);
        case 11: return launch_fused<128, 256, 2, 4, 3, 64, 2>(fuse, p, s);
        case 12: return launch_fused<256, 256, 2, 4, 2, 64, 2>(fuse, p, s);
        case 16: return launch_fused<256, 128, 2, 2, 3, 32, 2>(fuse, p, s);
        case 17: return launch_fused<128, 256, 1, 4, 3, 32, 2>(fuse, p, s);
        case 19: return launch_fused<256, 128, 4, 2, 3, 32, 4>(fuse, p, s);
        case 20: return launch_fused<256, 256, 2, 4, 4, 32, 2>(fuse, p, s);
        case 23: return launch_fused<160, 320, 5, 2, 3, 32, 3>(fuse, p, s);
        case 31: return launch_fused<128, 128, 2, 4, 3, 64, 2>(fuse, p, s);
        default: return T2V_EINVAL;
    }
}
#elif defined(T2V_GEMM_EXP_ONLY)
int t2v_gemm_launch_ln(int cfg, GemmParams& p, hipStream_t s) {  // tile id 23 only (its 64-deep twin, id 22, spills 22 registers with the extra epilogue)
    (void)cfg;
    return launch_impl<160, 320, 5, 2, 3, 32, 3, true, false, true>(p, s);
}
int t2v_gemm_launch_experimental(int cfg, GemmParams& p, hipStream_t s) {
    switch (cfg) {
        case 24: return launch<256, 256, 2, 2, 3, 32, 1>(p, s);
        case 25: return launch<256, 128, 4, 2, 2, 64, 2, true>(p, s);
        case 26: return launch<128, 128, 2, 2, 2, 64, 2, true>(p, s);
        case 27: return launch<256, 256, 2, 4, 2, 32, 2, true>(p, s);
        case 28: return launch<160, 320, 5, 2, 2, 32, 3, true>(p, s);
        case 29: return launch<128, 256, 1, 4, 2, 32, 2, true>(p, s);
        // EIGHT waves on the small workgroup tiles (two waves per SIMD instead of one): the 10x16 and 5x8 levels give fewer
        // than 256 tiles of any size, so a CU runs ONE workgroup and a 4-wave one leaves every SIMD with a single wave —
        // nothing to issue while it waits at the hand-over barrier or for its fragments
        case 30: return launch<128, 128, 4, 2, 3, 64, 2>(p, s);
        case 31: return launch<128, 128, 2, 4, 3, 64, 2>(p, s);
        case 32: return launch<128, 128, 2, 4, 4, 32, 2>(p, s);
        case 33: return launch<128, 128, 4, 2, 4, 32, 2>(p, s);
        default: return T2V_EINVAL;
    }
}
#else

namespace {
struct TileCfg { int bm, bn, wtn, bk = 64; };
// id -> (BM, BN, per-wave N width); waves / stages: see dispatch()
const TileCfg kCfg[] = {{0, 0, 0},      {128, 128, 64}, {128, 64, 32},  {256, 64, 64}, {128, 128, 64},
                        {128, 64, 32},  {256, 128, 64}, {256, 128, 64}, {64, 128, 32}, {256, 64, 32},
                        {128, 128, 64}, {128, 256, 64},
                        // 128x64 (tokens x channels) wave tiles: 25 % fewer LDS reads per MFMA than 64x64
                        {256, 256, 64}, {256, 128, 64}, {128, 256, 64}, {256, 256, 64},
                        // 32-deep K steps (64-byte LDS rows): <= 80 KiB of LDS and <= 256 VGPRs per 4-wave workgroup, so TWO
                        // workgroups share a CU and one's prologue / barriers / epilogue hide under the other's MFMAs
                        {256, 128, 64, 32}, {128, 256, 64, 32}, {128, 128, 64, 32}, {256, 128, 64, 32},
                        // 256x256 with a 4-slot ring of 32-deep steps (128 KiB): 1.5 K-steps of DMA lookahead instead of 1,
                        // for weight panels that stream from HBM (cold) rather than from L2
                        {256, 256, 64, 32}, {256, 256, 64, 32},
                        // 160x320 tiles, 10 waves (5x2, wave tile 32x160): for the 320-channel level, where 128-wide N tiles
                        // waste a sixth of the MFMAs on padding and M = 40960 gives exactly 256 tiles = one per CU, with the
                        // activation panel read once instead of once per N tile (wtn = 32 here just keeps GEGLU off it)
                        {160, 320, 32, 64}, {160, 320, 32, 32},
                        // 256x256 on FOUR waves, wave tile 128x128 (accumulators in AGPRs): half the LDS fragment bytes per
                        // MFMA of the 8-wave 256x256 tiles.  Compiles with a spill-free main loop; correct on MI355X and slower than the 8-wave tiles on every UNet shape (round 2)
                        // (added after the round's GPU budget was spent) - the tuned table never selects it.
                        {256, 256, 128, 32},
                        // register-staged twins (global -> VGPR -> ds_write_b128, two-slot ring) of 6, 1, 20, 23 and 17: the
                        // candidates for the question tools/fill_rate.hip asks (is LDS-DMA's 16 B/clk per CU the operand
                        // delivery limit).  Compile-verified only, like 24.
                        {256, 128, 64, 64}, {128, 128, 64, 64}, {256, 256, 64, 32}, {160, 320, 32, 32}, {128, 256, 64, 32},
                        // 8-wave small tiles (ids 30-33, see t2v_gemm_launch_experimental)
                        {128, 128, 64, 64}, {128, 128, 32, 64}, {128, 128, 32, 32}, {128, 128, 64, 32}};
constexpr int kNumCfg = 33;  // (a 4-wave 128x128-wave-tile variant spills: 3 KB/lane scratch, 72 TF/s - dropped)

int dispatch(int cfg, GemmParams& p, hipStream_t s) {
    switch (cfg) {
        case 1: return launch<128, 128, 2, 2, 2>(p, s);
        case 2: return launch<128, 64, 2, 2, 2>(p, s);
        case 3: return launch<256, 64, 4, 1, 2>(p, s);
        case 4: return launch<128, 128, 2, 2, 3>(p, s);
        case 5: return launch<128, 64, 2, 2, 3>(p, s);
        case 6: return launch<256, 128, 4, 2, 2>(p, s);
        case 7: return launch<256, 128, 4, 2, 3>(p, s);
        case 8: return launch<64, 128, 1, 4, 3>(p, s);
        case 9: return launch<256, 64, 4, 2, 3>(p, s);
        case 10: return launch<128, 128, 2, 2, 4>(p, s);
        case 11: return launch<128, 256, 2, 4, 3>(p, s);
        case 12: return launch<256, 256, 2, 4, 2>(p, s);
        case 13: return launch<256, 128, 2, 2, 2>(p, s);
        case 14: return launch<128, 256, 1, 4, 2>(p, s);
        case 15: return launch<256, 256, 4, 2, 2>(p, s);
        case 16: return launch<256, 128, 2, 2, 3, 32, 2>(p, s);
        case 17: return launch<128, 256, 1, 4, 3, 32, 2>(p, s);
        case 18: return launch<128, 128, 2, 2, 4, 32, 2>(p, s);
        case 19: return launch<256, 128, 4, 2, 3, 32, 4>(p, s);
        case 20: return launch<256, 256, 2, 4, 4, 32, 2>(p, s);
        case 21: return launch<256, 256, 4, 2, 4, 32, 2>(p, s);
        case 22: return launch<160, 320, 5, 2, 2, 64, 3>(p, s);
        case 23: return launch<160, 320, 5, 2, 3, 32, 3>(p, s);
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31: case 32: case 33:
            return t2v_gemm_launch_experimental(cfg, p, s);
        default: return T2V_EINVAL;
    }
}

}  // namespace

// tuning / test overrides: tile config id (0 = heuristic) and split-K factor (0 = heuristic)
static int g_force_cfg = 0, g_force_split = 0, g_debug = 0;
extern "C" int t2v_gemm_debug(int bits) { g_debug = bits; return T2V_OK; }
extern "C" int t2v_gemm_force_config(int cfg) { g_force_cfg = cfg; return T2V_OK; }
extern "C" int t2v_gemm_force_split(int s) { g_force_split = s; return T2V_OK; }
extern "C" int t2v_gemm_num_configs(void) { return kNumCfg; }

// Everything t2v_gemm decides before it launches: argument checks, geometry, tile id, split-K factor and — when the descriptor
// asks for fused normalisation statistics — whether the launch can carry them (fuse = FUSE bits, fuse_cfg = the tile that does;
// fuse_ok = false: the caller has to use the standalone kernels).  Shared by t2v_gemm and t2v_gemm_fuse_supported.
static int gemm_prepare(const t2v_gemm_desc* dd, GemmParams& p, int& cfg_out, int& fuse, int& fuse_cfg, bool& fuse_ok) {
    T2V_REQUIRE(dd && dd->a0 && dd->w && dd->out, T2V_EINVAL, "t2v_gemm: null pointer");
    T2V_REQUIRE(!dd->ln_in && !dd->gn_coef, T2V_EINVAL, "t2v_gemm: ln_in / gn_coef (a norm of the input rows) are t2v_linear_pr's; normalise first or ask t2v_linear_pr_supported");
    p.d = *dd;
    t2v_gemm_desc& d = p.d;
    if (!d.a1) { d.c1 = 0; d.lda1 = 0; }
    if (d.batch <= 0) d.batch = 1;
    if (d.batch_inner <= 0) d.batch_inner = 1;
    T2V_REQUIRE(d.M > 0 && d.N > 0, T2V_EINVAL, "t2v_gemm: empty problem");
    T2V_REQUIRE(d.c0 > 0 && d.c0 % 64 == 0 && d.c1 % 64 == 0, T2V_ESHAPE, "t2v_gemm: channels must be multiples of 64");
    T2V_REQUIRE(d.lda0 % 8 == 0 && d.lda1 % 8 == 0 && d.ldw % 8 == 0, T2V_ESHAPE, "t2v_gemm: lda/ldw must be multiples of 8");
    T2V_REQUIRE(((uintptr_t)d.a0 % 16 == 0) && ((uintptr_t)d.w % 16 == 0) && (!d.a1 || (uintptr_t)d.a1 % 16 == 0), T2V_ESHAPE,
                "t2v_gemm: operands must be 16-byte aligned");
    T2V_REQUIRE(d.a_stride0 % 8 == 0 && d.a_stride1 % 8 == 0 && d.w_stride0 % 8 == 0 && d.w_stride1 % 8 == 0, T2V_ESHAPE,
                "t2v_gemm: batch strides must be multiples of 8");
    T2V_REQUIRE(d.batch <= 65535, T2V_ESHAPE, "t2v_gemm: batch too large");
    if (d.drop_thr)
        T2V_REQUIRE(d.drop_seed && d.act == T2V_ACT_NONE && d.drop_ncols > 0 && d.drop_ncols % 4 == 0 && d.drop_col0 % 4 == 0 &&
                        d.drop_col0 >= 0 && d.drop_col0 + d.N <= d.drop_ncols && d.batch == 1,
                    T2V_EINVAL, "t2v_gemm: dropout epilogue (seed pointer, column geometry in multiples of 4, no activation, no batch)");
    p.nsrc = d.a1 ? 2 : 1;
    p.gh = d.h_in; p.gw = d.w_in;
    p.kh = 3; p.kw = 3; p.stride = 1; p.pad_y = 1; p.pad_x = 1; p.ups = 0;
    int n_grid = d.n_img;
    switch (d.mode) {
        case T2V_GEMM_LINEAR: p.kh = p.kw = 1; p.pad_y = p.pad_x = 0; p.gh = p.gw = 1; n_grid = d.M; break;
        case T2V_GEMM_CONV3X3: break;
        case T2V_GEMM_CONV3X3_S2: p.stride = 2; break;
        case T2V_GEMM_CONV3X3_S2_PAD01: p.stride = 2; p.pad_y = p.pad_x = 0; break;
        case T2V_GEMM_CONV3X3_UP2: p.ups = 1; break;
        case T2V_GEMM_TCONV3:
            T2V_REQUIRE(d.frames > 0 && d.n_img % d.frames == 0, T2V_EINVAL, "t2v_gemm: frames");
            p.kw = 1; p.pad_x = 0; p.gh = d.frames; p.gw = d.h_in * d.w_in; n_grid = d.n_img / d.frames;
            break;
        default: T2V_REQUIRE(false, T2V_EINVAL, "t2v_gemm: bad mode");
    }
    if (d.mode != T2V_GEMM_LINEAR) T2V_REQUIRE(d.n_img > 0 && d.h_in > 0 && d.w_in > 0, T2V_EINVAL, "t2v_gemm: conv geometry");
    p.taps = p.kh * p.kw;
    p.h_out = ((p.gh << p.ups) + 2 * p.pad_y - p.kh) / p.stride + 1;
    p.w_out = ((p.gw << p.ups) + 2 * p.pad_x - p.kw) / p.stride + 1;
    if (d.mode == T2V_GEMM_CONV3X3_S2_PAD01) {  // pad only right/bottom by 1
        p.h_out = (p.gh + 1 - 3) / 2 + 1; p.w_out = (p.gw + 1 - 3) / 2 + 1;
    }
    T2V_REQUIRE((long long)d.M == (long long)n_grid * p.h_out * p.w_out, T2V_EINVAL, "t2v_gemm: M does not match the geometry");
    p.K = p.taps * (d.c0 + d.c1);
    p.nk = p.K / 64;  // refined below once the tile config (BK) is known
    T2V_REQUIRE(d.ldw >= p.K, T2V_EINVAL, "t2v_gemm: ldw < K");
    if (d.act == T2V_ACT_GEGLU) T2V_REQUIRE(d.N % 128 == 0, T2V_ESHAPE, "t2v_gemm: GEGLU needs N % 128 == 0");
    if (d.rowvec) T2V_REQUIRE(d.rowvec_div > 0, T2V_EINVAL, "t2v_gemm: rowvec_div");
    p.zero = t2v_zero_page();
    T2V_REQUIRE(p.zero, T2V_EHIP, "t2v_gemm: zero page allocation failed");
    const int n_out = d.act == T2V_ACT_GEGLU ? d.N / 2 : d.N;
    p.vec4 = (d.ldo % 8 == 0) && (n_out % 4 == 0) && ((uintptr_t)d.out % 16 == 0) && (d.o_stride0 % 8 == 0) &&
             (d.o_stride1 % 8 == 0) && (!d.residual || (d.ldr % 8 == 0 && (uintptr_t)d.residual % 16 == 0)) &&
             (!d.bias || (uintptr_t)d.bias % 16 == 0) && (!d.rowvec || ((uintptr_t)d.rowvec % 16 == 0 && d.ld_rowvec % 4 == 0));
    if (d.act == T2V_ACT_GEGLU) T2V_REQUIRE(p.vec4, T2V_ESHAPE, "t2v_gemm: GEGLU needs 8-byte aligned rows");

    // ---- tile configuration -------------------------------------------------------------------------
    int cfg = g_force_cfg ? g_force_cfg : d.tile_cfg;
    if (cfg < 0 || cfg > kNumCfg) cfg = 0;
    if (cfg == 0) {
        if (d.N % 128 == 0 || d.N >= 1024) cfg = (d.M >= 8192 && p.nk >= 8) ? 7 : 4;
        else if (d.M >= 8192 && d.N % 64 == 0) cfg = 9;
        else cfg = 5;
    }
    if (d.act == T2V_ACT_GEGLU && kCfg[cfg].wtn < 64) cfg = 4;  // value+gate pairs need a 64-wide wave tile in N
    p.nk = p.K / kCfg[cfg].bk;
    // ---- split-K: only when the grid cannot fill the chip and K is deep ----------------------------------
    int splits = g_force_split ? g_force_split : d.split_k;
    const long long tiles = (long long)((d.M + kCfg[cfg].bm - 1) / kCfg[cfg].bm) * ((d.N + kCfg[cfg].bn - 1) / kCfg[cfg].bn) * d.batch;
    const bool can_split = d.ws && d.act != T2V_ACT_GEGLU && d.N % 4 == 0 && p.vec4;
    if (splits <= 0) {
        splits = 1;
        if (can_split && tiles < 160 && p.K / 64 >= 16) {
            splits = (int)((448 + tiles - 1) / tiles);
            if (splits > p.K / 512) splits = p.K / 512;
            if (splits > 16) splits = 16;
        }
    }
    if (!can_split) splits = 1;
    if (splits > 1 && (long long)splits * d.batch * d.M * d.N * 4 > d.ws_bytes) {
        splits = (int)(d.ws_bytes / ((long long)d.batch * d.M * d.N * 4));
        if (splits < 2) splits = 1;
    }
    if (splits > p.nk) splits = p.nk;
    p.nk_per_split = (p.nk + splits - 1) / splits;
    splits = (p.nk + p.nk_per_split - 1) / p.nk_per_split;
    p.splits = splits;
    p.ws = (float*)d.ws;
    p.debug = g_debug;
    cfg_out = cfg;
    fuse = (d.rowstat_out ? 1 : 0) | (d.colstat_out ? 2 : 0) | (d.lnf_stats ? 4 : 0) | (d.lora_t ? 16 : 0);
    // a dropout epilogue (the LoRA up-projections of the training path) rides on the fast kernels' staged epilogue where the launch
    // qualifies for them otherwise (FUSE bit 8 of gemm_fuse.hip); T2V_GEMM_FAST_DROPOUT=0: the generic kernels, as before round 3
    static const bool fast_dropout = !(getenv("T2V_GEMM_FAST_DROPOUT") && getenv("T2V_GEMM_FAST_DROPOUT")[0] == '0');
    const bool auto_drop = !fuse && !d.lora_t && d.drop_thr && fast_dropout && !d.rowvec && d.act == T2V_ACT_NONE && !d.ln_out && d.batch == 1;
    if (auto_drop) fuse = 8;
    fuse_cfg = 0;
    fuse_ok = false;
    if (fuse) {
        T2V_REQUIRE(fuse == 1 || fuse == 2 || fuse == 4 || fuse == 8 || fuse == 16 || fuse == 18, T2V_EINVAL,
                    "t2v_gemm: one of rowstat_out / colstat_out / lnf_stats per launch (lora_t alone or with colstat_out)");
        if (fuse & 16)
            T2V_REQUIRE(d.lora_u && d.lora_n_leaf > 0 && d.lora_n_leaf % 32 == 0 && d.N % d.lora_n_leaf == 0 && d.ld_lora_t % 8 == 0 &&
                            d.ld_lora_u % 8 == 0 && d.ld_lora_u >= 64 && d.ld_lora_t >= 64 * (d.N / d.lora_n_leaf) &&
                            (uintptr_t)d.lora_t % 16 == 0 && (uintptr_t)d.lora_u % 16 == 0 && d.act == T2V_ACT_NONE &&
                            (!d.drop_thr || (d.drop_seed && d.drop_ncols % 4 == 0 && d.drop_col0 % 4 == 0)),
                        T2V_ESHAPE, "t2v_gemm: lora_t: rank-64 rows [M][64 leaves], lora_u [N][64], N a multiple of lora_n_leaf (% 32), no activation");
        T2V_REQUIRE(!d.ln_out && d.batch == 1, T2V_EINVAL, "t2v_gemm: fused statistics: no batch, no LayerNorm second output");
        if (fuse == 1)
            T2V_REQUIRE(d.N % 32 == 0 && d.ld_rowstat >= d.N / 16 && d.ld_rowstat % 4 == 0 && (uintptr_t)d.rowstat_out % 16 == 0 &&
                            d.act == T2V_ACT_NONE, T2V_ESHAPE, "t2v_gemm: rowstat_out needs N % 32 == 0, ld_rowstat >= N / 16 (% 4), no activation");
        if (fuse & 2)
            T2V_REQUIRE(d.M % 32 == 0 && d.N % 2 == 0 && (uintptr_t)d.colstat_out % 16 == 0 && d.act != T2V_ACT_GEGLU, T2V_ESHAPE,
                        "t2v_gemm: colstat_out needs M % 32 == 0");
        if (fuse == 4)
            T2V_REQUIRE(d.mode == T2V_GEMM_LINEAR && !d.a1 && d.lnf_s && d.bias && d.lnf_nblk > 0 && d.lnf_nblk % 2 == 0 && d.lnf_nblk <= 40 &&
                            d.c0 == 32 * d.lnf_nblk && d.lnf_ld >= 2 * d.lnf_nblk && d.lnf_ld % 4 == 0 && (uintptr_t)d.lnf_stats % 16 == 0 &&
                            (uintptr_t)d.lnf_s % 16 == 0 && !d.residual && !d.rowvec && !d.drop_thr && d.act != T2V_ACT_SILU,
                        T2V_ESHAPE, "t2v_gemm: lnf_stats: LINEAR over C = 32 lnf_nblk <= 1280 channels, lnf_s and bias given, no residual / "
                                    "row vector / dropout");
        fuse_cfg = t2v_gemm_fuse_tile(cfg, d.act, fuse);
        // the fused epilogue rides on the fast kernels of one K split; GEGLU needs a 64-wide wave tile (its value / gate pairs)
        fuse_ok = fuse_cfg != 0 && p.splits == 1 && gemm_is_fast(p, fuse == 8 || (fuse & 16) != 0) && !(d.act == T2V_ACT_GEGLU && kCfg[fuse_cfg].wtn < 64);
        // LoRA epilogue: a wave tile loads the rank-64 rows of at most TWO leaves (lra_t[2]); a group whose leaves are narrow could put
        // a third leaf under one wave tile, which would then be multiplied with leaf 1's rows.  Wave tiles start at multiples of
        // their width, so the check is exact: walk them.
        if ((fuse & 16) && fuse_cfg != 0 && d.N != d.lora_n_leaf) {
            const int wtn = fuse_cfg == 23 ? 160 : kCfg[fuse_cfg].wtn;   // (the 160x320 tile: 32 in the table keeps GEGLU off it; its wave tile is 160 wide)
            for (int s = 0; s < d.N && fuse_ok; s += wtn)
                if ((s + wtn - 1 < d.N ? s + wtn - 1 : d.N - 1) / d.lora_n_leaf - s / d.lora_n_leaf > 1) fuse_ok = false;
        }
        // the dropout epilogue moves to the fast kernels only where that keeps the launch on (a twin of) the tile the table chose
        // for it: measured per shape on MI355X (profiles/r03_student_gemm_fast_dropout.csv), 160x320 (id 28 -> 23) wins 20-35 %,
        // ids that map to a different workgroup tile or lose their register-staged prologue (18 -> 4, 29 -> 11) lose 5-45 %
        if (fuse == 8 && !(fuse_cfg == cfg || cfg == 28 || cfg == 32)) fuse_ok = false;
        if (fuse_ok) { p.nk = p.K / kCfg[fuse_cfg].bk; p.nk_per_split = p.nk; }
        if (fuse == 8 && !fuse_ok) fuse = 0;   // (not a request: the generic kernel carries the dropout epilogue)
        return T2V_OK;
    }
    return T2V_OK;
}

// The second kernel family (gemm2.hip: static-schedule main loop, 80x80 wave tiles; tile ids 50 / 51) for LINEAR / TCONV3 launches
// with a plain epilogue.  MEASURED SLOWER than the tuned first-family tiles on every UNet shape (profiles/r04_gemm2_*.csv: 62 vs
// 44 us at 40960 x 320 x 1280, 54 vs 43 us at 10240 x 640 x 2560): with both operands restaged per K step a 320x160 tile needs
// 3.75 LDS-DMA pieces per wave per 25 MFMAs (conv_halo: 1.25) and 64-byte row pieces fetch every cache line twice — the loop is
// bound by DMA issue and L2 -> LDS bytes, not by its schedule.  Kept as a tested, measured negative result: OFF unless asked for
// (t2v_gemm2_enable(1) / T2V_GEMM2=1: the library's own routing rule; tile id 50 / 51 forced: that tile).
#ifdef T2V_EXPERIMENTAL
static int g_gemm2 = -1;
extern "C" int t2v_gemm2_enable(int on) { g_gemm2 = on ? 1 : 0; return T2V_OK; }
// c2 > 0: the second family takes this launch.  It implements the column statistics (fuse == 2) itself; every other fused request
// (row statistics, LayerNorm fold, LoRA / dropout epilogue) and the LayerNorm second output stay with the first family.
static int gemm2_route(const t2v_gemm_desc* dd, int fuse, Gemm2Params& p2, int& c2) {
    c2 = 0;
    if (g_gemm2 < 0) g_gemm2 = (getenv("T2V_GEMM2") && getenv("T2V_GEMM2")[0] == '1') ? 1 : 0;
    const int want = g_force_cfg ? g_force_cfg : dd->tile_cfg;
    const int forced = (want == 50 || want == 51) ? want - 49 : 0;
    if ((fuse & ~2) || dd->ln_out || !(forced || (g_gemm2 && !g_force_cfg))) return T2V_OK;
    return t2v_gemm2_prepare(dd, p2, forced, c2);
}
#endif   // T2V_EXPERIMENTAL

extern "C" int t2v_gemm_fuse_supported(const t2v_gemm_desc* dd) {
    GemmParams p;
    int cfg = 0, fuse = 0, fuse_cfg = 0;
    bool ok = false;
    const int rc = gemm_prepare(dd, p, cfg, fuse, fuse_cfg, ok);
    if (rc != T2V_OK) return rc;
#ifdef T2V_EXPERIMENTAL
    if (fuse == 2) {   // column statistics: the second kernel family carries them on the launches it takes
        Gemm2Params p2;
        int c2 = 0;
        const int rc2 = gemm2_route(dd, fuse, p2, c2);
        if (rc2 != T2V_OK) return rc2;
        if (c2) return 1;
    }
#endif
    return ((fuse & ~8) && ok) ? 1 : 0;   // (bit 8 is the library's own choice for dropout launches, not a request)
}

// What the library would launch for this descriptor: tile id (the fused twin's where the fused epilogue is taken) and K splits.
// The gradient engine asks before it attaches a LoRA epilogue (which rides on ONE split): where the plain launch would split K
// — the 5x8 / 10x16 levels' long-K convs — the epilogue form loses more in the main loop than it saves after it.
extern "C" int t2v_gemm_plan(const t2v_gemm_desc* dd, int* tile_cfg, int* splits) {
    GemmParams p;
    int cfg = 0, fuse = 0, fuse_cfg = 0;
    bool ok = false;
    const int rc = gemm_prepare(dd, p, cfg, fuse, fuse_cfg, ok);
    if (rc != T2V_OK) return rc;
    if (tile_cfg) *tile_cfg = (fuse && ok) ? fuse_cfg : cfg;
    if (splits) *splits = p.splits;
    return T2V_OK;
}

extern "C" int t2v_gemm(const t2v_gemm_desc* dd, void* stream) {
    GemmParams p;
    int cfg = 0, fuse = 0, fuse_cfg = 0;
    bool fuse_ok = false;
    const int rc = gemm_prepare(dd, p, cfg, fuse, fuse_cfg, fuse_ok);
    if (rc != T2V_OK) return rc;
    t2v_gemm_desc& d = p.d;
    hipStream_t s = (hipStream_t)stream;
#ifdef T2V_EXPERIMENTAL
    {
        Gemm2Params p2;
        int c2 = 0;
        const int rc2 = gemm2_route(dd, fuse, p2, c2);
        if (rc2 != T2V_OK) return rc2;
        if (c2) return t2v_gemm2_dispatch(c2, p2, s);
    }
#endif
    if (fuse) {
        T2V_REQUIRE(fuse_ok, T2V_ESHAPE, "t2v_gemm: this launch cannot carry fused statistics (ask t2v_gemm_fuse_supported first)");
        return t2v_gemm_launch_fused(fuse_cfg, fuse, p, s);
    }
    if (d.ln_out) {  // LayerNorm second output: only the full-row 160x320 FAST kernels carry it
        cfg = 23;
        p.nk = p.K / kCfg[cfg].bk;
        p.splits = 1; p.nk_per_split = p.nk;
        const bool fast = p.vec4 && d.N % 16 == 0 && d.alpha == 1.0f && !d.out_f32 && !d.drop_thr &&
                          (!d.rowvec || ((uintptr_t)d.rowvec % 16 == 0 && d.ld_rowvec % 4 == 0));
        T2V_REQUIRE(d.N == 320 && d.batch == 1 && d.act == T2V_ACT_NONE && fast && d.ln_gamma && d.ln_beta && d.ld_ln_out % 8 == 0 &&
                        (uintptr_t)d.ln_out % 16 == 0 && (uintptr_t)d.ln_gamma % 16 == 0 && (uintptr_t)d.ln_beta % 16 == 0,
                    T2V_ESHAPE, "t2v_gemm: the LayerNorm output needs N == 320, bf16 out, alpha 1, aligned operands, no batch / activation");
        return t2v_gemm_launch_ln(cfg, p, s);
    }
    return dispatch(cfg, p, s);
}
#endif  // T2V_GEMM_EXP_ONLY
