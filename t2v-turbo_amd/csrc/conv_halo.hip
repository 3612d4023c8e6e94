// Implicit-GEMM 3x3 convolution (stride 1, pad 1; optionally over a nearest-x2 upsampled source) with an LDS-RESIDENT HALO SLAB (gfx950, bf16 MFMA, fp32 accumulate).
//
//   out[M,N] = epilogue( sum over taps t, channels c:  A[shift_t(m)][c] * W[n][t][c] )
//
// Why a second conv kernel (round 4).  The ablation of t2v_gemm on its long-K conv shapes (profiles/r04_gemm_ablation_*.csv)
// shows the L2 -> LDS fill to be as long as the MFMA work (DMA-only time == MFMA-only time; 13 TB/s chip-wide = 65 GB/s per
// CU on the 256x128 tile): that kernel re-stages the activation tile once per filter tap, 9 x (BM + BN) rows per channel slab.
// Here a workgroup's output tile is a RECTANGLE of the image grid (S rows x FX columns of tokens); the activation rows it
// needs for ALL taps of one 32-channel sub-slab are the rectangle plus its one-pixel halo, (S+2) x (FX+2) rows, staged ONCE
// and read at nine row offsets; only the weights stream per tap.  Bytes staged per MAC fall from 2/BM + 2/BN to
// 2/BM + 0.3/BN, which makes TALL tiles the right shape (320 tokens x 160 / 80 channels): 2.3 - 2.9 x less L2 -> LDS traffic
// than the tuned t2v_gemm tiles at the same levels.
//
//  * 8 waves; every wave owns an 80 x 80 (tokens x channels) tile = 5 x 5 v_mfma_f32_16x16x32_bf16 blocks (100 accumulator
//    registers): 0.4 ds_read_b128 per MFMA.  A wave's five token blocks are the same 16 columns of five consecutive image rows,
//    so their fragment addresses differ by a compile-time constant.  The workgroup is KG k-groups x WM x WN waves: 320x160
//    (KG 1), 320x80 (KG 2), 160x80 (KG 4) — one tile per CU at the 40x64 / 20x32 / 10x16 levels of the UNet (256 tiles each).
//    K-groups take alternate (sub-slab, tap) pairs and are summed through LDS at the end (fixed order: deterministic).
//  * K order is (32-channel sub-slab, tap, channel): the weight pack is [N][C/32][9][32], zero-padded to whole weight stages
//    (packed once per weight version by the caller), so that a stage of NP consecutive pairs is NP*64 contiguous bytes of a row.
//  * LDS: NA activation sub-slab buffers [(S+2) * PT rows][64 B], PT = FX + 8 (a multiple of 8: the swizzle term of a fragment
//    address then depends on the tap's column offset only) + a ring of NWS weight stages [NP][BN rows][64 B]; 16-byte chunks
//    are XOR-swizzled by 2*((row>>2)&1): conflict-free ds_read_b128 for the 16x16x32 fragment pattern (lane -> row l&15,
//    chunk l>>4) at ANY start row, which the tap offsets need.
//  * All staging is LDS-DMA in the raw-buffer form (buffer_load_dwordx4 ... offen lds): the per-lane offsets are constants, the
//    advance per stage / sub-slab is a SCALAR offset, padding lanes are out of range and read as zeros — no address arithmetic
//    and no zero page in the loop.  Waves 0-3 stage weights, waves 4-7 the activation slabs: a wave only ever counts its own
//    kind of DMA (counted s_waitcnt vmcnt); one raw s_barrier per weight stage, placed before the LAST pair's MFMAs of the stage
//    so that the next stage's first fragment reads and the DMA issue hide under them.
//  * Epilogue: accumulators start at bias + time-embedding row vector + residual; the bf16 tile is parked in LDS (whole
//    workgroup), written out in full rows, and the per-32-row column statistics of the NEXT GroupNorm are taken from it.
#include "tile80.h"
#include <cstdlib>
#include <type_traits>

struct HaloParams {
    t2v_gemm_desc d;
    int U, V;              // per-image OUTPUT grid (rows, columns): token m = (img * U + u) * V + v
    int ups;               // 1: nearest-x2 upsampling folded into the gather (T2V_GEMM_CONV3X3_UP2): the source grid is (U / 2, V / 2)
    int tiles_s, tiles_f;  // tiles per image along the rows / the columns
    int tiles_m, tiles_n;
    int nsub, nq, nstage;  // 32-channel sub-slabs, (sub-slab, tap) pairs, weight stages
    int xcd_m, xcd_n, nblk;
    int blk_start[8], blk_r0[8], blk_c0[8], blk_w[8];
    int debug;             // ablation bits (T2V_HALO_ABLATE builds)
};

// Ablation switches (tools only; a -DT2V_HALO_ABLATE build honours t2v_conv_halo_debug bits inside the main loop: 1 = no weight
// DMA after the prologue, 2 = no activation DMA after the prologue, 4 = no fragment reads after the first, 8 = no barrier,
// 16 = no MFMA, 32 = no output stores, 64 = no MFMA / ds_read interleave hints, 128 = leave after the prologue, 256 = leave after
// the main loop).  Compiled out of the product.
#ifdef T2V_HALO_ABLATE
#define HABL(bit) (p.debug & (bit))
#else
#define HABL(bit) false
#endif

namespace {

using tile80::kOutOfRange;
using tile80::static_for_until;
template <int N>
__device__ __forceinline__ void hwait_vmcnt() { tile80::wait_vmcnt<N>(); }

// NWL of the 8 waves stage weights (W_IT pieces of 1 KiB per stage each), the other 8 - NWL the activation sub-slabs (A_IT pieces
// per sub-slab each).
template <int WM, int WN, int KG, int NP, int NWS, int NA, int NWL, int A_IT, int FX, int NB = 5>
struct HaloCfg {
    static_assert(WM * WN * KG == 8, "eight waves");
    static_assert(NB == 5 || NB == 4, "channel blocks (of 16) per wave tile");
    static constexpr int WTN = 16 * NB;                    // channels per wave tile
    static_assert(FX == 16 || FX == 32, "tile width");
    static constexpr int HALVES = FX / 16;                 // 16-token MFMA blocks per tile row
    static_assert(WM % HALVES == 0, "a wave owns five rows of one 16-column half");
    static constexpr int S = 5 * (WM / HALVES);            // tile rows
    static constexpr int PT = FX + 8;                      // slab row pitch (rows of 64 B): halo + padding to a multiple of 8
    static constexpr int SLAB_ROWS = (S + 2) * PT;
    static constexpr int NAL = 8 - NWL;
    static_assert(SLAB_ROWS <= A_IT * NAL * 16, "activation buffer too small for the halo slab");
    static constexpr int BM = WM * 80, BN = WN * WTN;
    static_assert(BM == S * FX, "tile geometry");
    static constexpr int W_PIECES = NP * BN / 16;          // 1 KiB pieces per weight stage
    static_assert(W_PIECES % NWL == 0, "weight pieces per loader wave");
    static constexpr int W_IT = W_PIECES / NWL;
    static constexpr int A_BYTES = A_IT * NAL * 1024;      // one activation sub-slab buffer
    static constexpr int WS_BYTES = NP * BN * 64;
    static constexpr int RING_BYTES = NA * A_BYTES + NWS * WS_BYTES;
    static constexpr int OUT_BYTES = tile80::Epi<BM, BN>::BYTES;   // the epilogue's staging area + the tile's global row table
    static constexpr int RED_BYTES = (KG - 1) * WM * WN * NB * 5 * 1024;
    static constexpr int SMEM = RING_BYTES > OUT_BYTES ? (RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES) : (OUT_BYTES > RED_BYTES ? OUT_BYTES : RED_BYTES);
    static_assert(SMEM <= 160 * 1024, "LDS");
    static_assert(NP % KG == 0, "every k-group takes the same number of pairs of a stage");
    // ---- the static schedule.  The main loop is unrolled over SUPER stages = a whole number of sub-slabs, ring cycles and
    // fragment-set alternations, so that taps, ring slots, activation buffers and DMA wait counts are compile-time constants.
    static constexpr int PPS = NP / KG;                            // pairs per stage and k-group
    static constexpr int SUPER = (PPS % 2 == 0) ? 9 : 18;          // stages per unrolled super-iteration
    static constexpr int SUBS = SUPER * NP / 9;                    // sub-slabs per super-iteration
    static_assert(SUPER % NWS == 0 && SUBS % NA == 0, "ring slots / activation buffers must be periodic in the super-iteration");
    static constexpr int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
    // sub-slab s (global index) is issued after the barrier of stage issue_at(s) (the one that retires the last reader of
    // sub-slab s - NA; the first NA are issued in the prologue) and must have landed before the barrier of stage due_at(s)
    static constexpr int issue_at(int s) { return s < NA ? -(1 << 20) : fdiv(9 * (s - NA) + 8, NP); }
    static constexpr int due_at(int s) { return fdiv(9 * s, NP) - 1; }
    static constexpr int max_issued(int k) { int s = NA - 1; while (issue_at(s + 1) <= k - 1) ++s; return s; }   // at hand-overs < k
    static constexpr int needed_by(int k) { int s = 0; while (due_at(s + 1) <= k) ++s; return s; }
    static constexpr bool a_wait_here(int k) { for (int s = 0; due_at(s) <= k; ++s) if (due_at(s) == k) return true; return false; }
    static constexpr int a_younger(int k) { return max_issued(k) - needed_by(k); }   // sub-slabs in flight behind the needed one
    static constexpr int first_issue(int k) { return max_issued(k) + 1; }            // sub-slabs issued at hand-over k: [first, first + count)
    static constexpr int issue_count(int k) { int n = 0; while (issue_at(max_issued(k) + 1 + n) == k) ++n; return n; }
    static constexpr bool schedule_ok() {
        for (int k = 0; k < 4 * SUPER; ++k) {
            if (a_younger(k) < 0) return false;                       // the needed sub-slab must have been issued
            if ((a_younger(k) + 1) * A_IT > 60) return false;         // vmcnt field
        }
        return true;
    }
    static_assert(schedule_ok(), "activation-buffer schedule");
    static_assert(W_IT * (NWS - 1) <= 60, "vmcnt field");
};

template <int WM, int WN, int KG, int NP, int NWS, int NA, int NWL, int A_IT, int FX, int NB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_halo_kernel(const HaloParams p) {
    // (the host pass of hipcc does not know the buffer-descriptor builtins and would silently drop the kernel's stub: it sees an empty body)
#if defined(__HIP_DEVICE_COMPILE__) || defined(T2V_HOSTSIM)
    using C = HaloCfg<WM, WN, KG, NP, NWS, NA, NWL, A_IT, FX, NB>;
    constexpr int BM = C::BM, BN = C::BN, W_IT = C::W_IT, PPS = C::PPS, PT = C::PT, S = C::S, HALVES = C::HALVES, Fx = FX, WTN = C::WTN;
    constexpr int NMF = 5 * NB;   // MFMAs per (sub-slab, tap) pair and wave
    constexpr int T = 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const t2v_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgroup = wave / (WM * WN), wv = wave % (WM * WN);
    const int wave_m = wv / WN, wave_n = wv % WN;
    const int rg = wave_m / HALVES, half = wave_m % HALVES;   // my five tile rows 5 rg .. 5 rg + 4, columns 16 half .. + 15
    const bool w_loader = wave < NWL;
    constexpr int NAL = C::NAL;
    const int lw = w_loader ? wave : wave - NWL;   // index among the loaders of my kind

    // ---- tile assignment: XCD-aware, block table from the host (as in t2v_gemm) ---------------------------------
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int tile_m, tile_n;
    {
        int b = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (i < p.nblk && tile >= p.blk_start[i]) b = i;
        const int t = tile - p.blk_start[b], bw = p.blk_w[b];
        const int q = t / bw;
        tile_m = p.blk_r0[b] + q;
        tile_n = p.blk_c0[b] + t - q * bw;
    }
    const int n0 = tile_n * BN;
    // tile_m -> (image, tile row, tile column)
    const int tpi = p.tiles_s * p.tiles_f;
    const int img = tile_m / tpi, trem = tile_m - img * tpi;
    const int ts_i = trem / p.tiles_f, tf_i = trem - ts_i * p.tiles_f;
    const int s0 = ts_i * S, f0 = tf_i * Fx;     // tile origin (image row, image column)
    const int U = p.U, V = p.V;
    // global OUTPUT row of tile-relative (s, f) (< 0: outside the image)
    auto token_of = [&](int s, int f) -> int {
        const int u = s0 + s, v = f0 + f;
        return (u >= 0 && u < U && v >= 0 && v < V) ? (img * U + u) * V + v : -1;
    };
    // global SOURCE row the conv reads at (tile-relative) position (s, f) of the output grid: the same token, or with nearest-x2
    // upsampling (openaimodel3d.py:98-112: interpolate, then conv) the source pixel (u >> 1, v >> 1) of the half-size grid — the
    // slab then holds every source row up to four times, which costs L1 / L2 hits, not a different kernel
    const int ups = p.ups;
    auto src_token_of = [&](int s, int f) -> int {
        const int u = s0 + s, v = f0 + f;
        return (u >= 0 && u < U && v >= 0 && v < V) ? (img * (U >> ups) + (u >> ups)) * (V >> ups) + (v >> ups) : -1;
    };

    char* const a_base = smem;
    char* const w_base = smem + NA * C::A_BYTES;

    // ---- loader bookkeeping: raw-buffer LDS-DMA ------------------------------------------------------------------------
    // Descriptors span 2 GiB (the per-lane offsets are checked on the host to stay below that); a lane that must deliver zeros
    // (channel row >= N, halo outside the image, slab padding) carries an out-of-range offset instead of a zero-page pointer.
    //  * weight loaders: piece i = lw + 4 j of the stage image [pair][BN rows][64 B]; lane -> row 16 (i % (BN/16)) + lane/4;
    //    ld_off[j] = BYTE offset of (row, pair, my chunk) in the pack at stage 0; the stage advance (NP*64 B) is the scalar offset
    //  * activation loaders: piece i = lw + 4 j of the slab image [(S+2) * PT rows][64 B]; ld_off[j] = global token (< 0: zeros)
    // The 16-byte chunk a lane fetches is (lane & 3) ^ swizzle(row), and bit 2 of the row is bit 4 of the lane in every piece.
    constexpr int L_IT = W_IT > A_IT ? W_IT : A_IT;
    int ld_off[L_IT];
    const int ld_ch16 = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
    // (exact sizes: the loaders run ahead blindly past the last stage / sub-slab, and what they fetch there must not fault)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, (unsigned)d.N * (unsigned)d.ldw * 2u, 0x00020000);
    const unsigned m_src = (unsigned)d.M >> (2 * p.ups);   // source rows
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.a0, 0, m_src * (unsigned)d.lda0 * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = d.a1 ? __builtin_amdgcn_make_buffer_rsrc((void*)d.a1, 0, m_src * (unsigned)d.lda1 * 2u, 0x00020000) : rs_a0;
    if (w_loader) {
#pragma unroll
        for (int j = 0; j < L_IT; ++j) {
            const int i = lw + NWL * j;
            const int pair = i / (BN / 16), r16 = i - pair * (BN / 16);
            const int n = n0 + r16 * 16 + (lane >> 2);
            ld_off[j] = (j < W_IT && n < d.N) ? (n * d.ldw + pair * 32) * 2 + ld_ch16 : (int)kOutOfRange;
        }
    } else {
#pragma unroll
        for (int j = 0; j < L_IT; ++j) {
            const int row = (lw + NAL * j) * 16 + (lane >> 2);
            const int sr = row / PT, fc = row - sr * PT;
            ld_off[j] = (j < A_IT && sr < S + 2 && fc < Fx + 2) ? src_token_of(sr - 1, fc - 1) : -1;
        }
    }
    auto issue_w = [&](int stage, int slot_byte_off) {   // weight stage `stage` into the ring slot at this byte offset
        char* dst = w_base + slot_byte_off;
        const int w_soff = min(stage, p.nstage - 1) * (NP * 64);   // byte offset of the stage within a pack row (run-ahead past the end: the last stage again)
#pragma unroll
        for (int j = 0; j < W_IT; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + (lw + NWL * j) * 1024), 16, ld_off[j], w_soff, 0, 0);
    };
    auto issue_a = [&](int a_sub, int buf_off) {   // activation sub-slab a_sub (global index) into the buffer at byte offset buf_off
        char* dst = a_base + buf_off;
        const int ch0 = min(a_sub, p.nsub - 1) * 32;   // (run-ahead past the end: the last sub-slab again — nothing is read outside the operands)
        const bool second = ch0 >= d.c0 && d.a1;
        const int ld2 = (second ? d.lda1 : d.lda0) * 2, soff = (second ? ch0 - d.c0 : ch0) * 2;
        const __amdgpu_buffer_rsrc_t rs = second ? rs_a1 : rs_a0;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const unsigned voff = ld_off[j] >= 0 ? (unsigned)(ld_off[j] * ld2 + ld_ch16) : kOutOfRange;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (lw + NAL * j) * 1024), 16, voff, soff, 0, 0);
        }
    };

    // ---- fragment addressing -------------------------------------------------------------------------------------
    // weights (MFMA A operand: rows = channels): row 16 bn + l15 of the pair tile, chunk lq ^ swizzle(l15): one lane constant,
    // the five blocks 1 KiB apart
    const int w_lane = (wave_n * WTN + l15) * 64 + ((lq ^ (((l15 >> 2) & 1) << 1)) << 4) + kgroup * (BN * 64);
    // activations (MFMA B operand: columns = tokens): block bm, tap (ty, tx) reads slab row (5 rg + bm + ty) PT + 16 half + tx + l15.
    // PT and 16 half are multiples of 8, so the swizzle bit (row & 4) is ((tx + l15) & 4): the per-lane part of the address depends
    // on tx only, the rest is a scalar, and the five blocks are PT rows = PT*64 bytes apart.
    // per tap column tx: byte offset of my row 5 rg, column 16 half + tx + l15 (+ swizzled chunk) within a buffer
    const int a_rows0 = (rg * 5 * PT + half * 16) * 64;
    auto a_lane_of = [&](int tx) { const int t = l15 + tx; return a_rows0 + (t << 6) + ((lq << 4) ^ ((t & 4) << 3)); };
    const int a_lane0 = a_lane_of(0), a_lane1 = a_lane_of(1), a_lane2 = a_lane_of(2);
    bf16x8_t fa[2][5], fw[2][NB];
    f32x4_t acc[NB][5];   // [channel block][token block]

    // Fragments of one pair into register set `which`.  QB = JS * NP + I * KG: the pair of k-group 0 at this point of the
    // super-iteration; mine is QB + kgroup.  With one k-group everything is a compile-time constant (sub-slab QB / 9 in buffer
    // (QB / 9) % NA, tap QB % 9, weight tile QB % NP of ring slot (QB / NP) % NWS).  With several, the weight offset still is
    // (my tile's distance from k-group 0's sits in w_lane), while the tap and the activation buffer of MY pair are tracked at
    // run time (rd_tap, rd_abuf: a handful of scalar operations per pair).
    int rd_tap = kgroup, rd_abuf = 0;   // (KG > 1 only)
    auto read_frags = [&](auto qb_tag, int which) {
        if (HABL(4)) return;
        constexpr int QB = decltype(qb_tag)::value % (C::SUPER * NP);   // (the first pair of the next super-iteration = the first of this one)
        constexpr int W_OFF = NA * C::A_BYTES + ((QB / NP) % NWS) * C::WS_BYTES + (QB % NP) * (BN * 64);
        const char* wb = smem + W_OFF + w_lane;
        const char* ab;
        if constexpr (KG == 1) {
            constexpr int SUB = QB / 9, TAP = QB % 9, TY = TAP / 3, TX = TAP % 3;
            ab = smem + ((SUB % NA) * C::A_BYTES + TY * (PT * 64)) + (TX == 0 ? a_lane0 : (TX == 1 ? a_lane1 : a_lane2));
        } else {
            const int ty = (rd_tap * 11) >> 5, tx = rd_tap - 3 * ty;
            const int t = l15 + tx;
            ab = smem + (rd_abuf + ty * (PT * 64) + a_rows0) + ((t << 6) + ((lq << 4) ^ ((t & 4) << 3)));
            rd_tap += KG;
            if (rd_tap >= 9) { rd_tap -= 9; rd_abuf = rd_abuf + C::A_BYTES == NA * C::A_BYTES ? 0 : rd_abuf + C::A_BYTES; }
        }
#pragma unroll
        for (int bm = 0; bm < 5; ++bm) fa[which][bm] = *(const bf16x8_t*)(ab + bm * (PT * 64));
#pragma unroll
        for (int bn = 0; bn < NB; ++bn) fw[which][bn] = *(const bf16x8_t*)(wb + bn * 1024);
    };
    auto mfmas = [&](int which, int first, int last) {
        if (HABL(16)) return;
#pragma unroll
        for (int bn = 0; bn < NB; ++bn)
#pragma unroll
            for (int bm = 0; bm < 5; ++bm)
                if (bn * 5 + bm >= first && bn * 5 + bm < last)
                    acc[bn][bm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[which][bn], fa[which][bm], acc[bn][bm], 0, 0, 0);
    };

    // ---- prologue: fill the weight ring and the first activation buffers -------------------------------------------
    if (w_loader) {
#pragma unroll
        for (int st = 0; st < NWS; ++st) issue_w(st, st * C::WS_BYTES);
    } else {
#pragma unroll
        for (int st = 0; st < NA; ++st) issue_a(st, st * C::A_BYTES);
    }
    // accumulators of k-group 0 start at bias + time-embedding row vector (the loads are the youngest vector-memory operations
    // when consumed: see t2v_gemm); the other k-groups start at zero.  The residual is NOT folded in here: in this accumulator
    // layout a lane owns 4 channels of a row, so the tile would arrive in 8-byte pieces and the main loop could not start before
    // they had (measured: +11 us at 40960 x 320); it is added in the epilogue's row pass instead, with full-width loads.
    const int ch_lane = n0 + wave_n * WTN + lq * 4;
#pragma unroll
    for (int bn = 0; bn < NB; ++bn)
#pragma unroll
        for (int bm = 0; bm < 5; ++bm) acc[bn][bm] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (kgroup == 0) {
        // (a tile lies inside one image, and rowvec_div is a whole number of images — host-checked — so the row vector is one
        // row for the whole tile: five 16-byte loads per lane next to the five of the bias)
        const long long rv_row = d.rowvec ? (long long)(((long long)img * U * V) / d.rowvec_div) * d.ld_rowvec : 0;
#pragma unroll
        for (int bn = 0; bn < NB; ++bn) {
            const int ch = ch_lane + bn * 16;
            float4 v = (d.bias && ch < d.N) ? *(const float4*)(d.bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.rowvec && ch < d.N) {
                const float4 r4 = *(const float4*)(d.rowvec + rv_row + ch);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
#pragma unroll
            for (int bm = 0; bm < 5; ++bm) acc[bn][bm] = (f32x4_t){v.x, v.y, v.z, v.w};
        }
    }
    hwait_vmcnt<0>();                       // everything staged so far has landed (mine)
    __builtin_amdgcn_s_barrier();           // ... and everybody's
    asm volatile("" ::: "memory");

    // ---- main loop ---------------------------------------------------------------------------------------------------------
    // Unrolled over a super-iteration of SUPER stages (see HaloCfg): every stage body is straight-line code with compile-time
    // LDS offsets and DMA wait counts.  Per stage and k-group PPS pairs; in the last pair of a stage the hand-over: my fragment
    // reads have landed, my DMAs of what stage k + 1 reads have landed (counted vmcnt: later ones stay in flight), s_barrier,
    // fragment reads of stage k + 1's first pair, DMA issue into what the barrier freed, then the pair's MFMAs over all of that.
    // The loaders run ahead BLINDLY: past the last stage / sub-slab they fetch what lies behind (out of range: zeros) into
    // slots nobody reads, so the counts never change; everything is drained before the LDS is reused.  One scalar compare per
    // stage leaves the loop after the last stage.
    if (HABL(128)) return;   // (launch + prologue only)
    // static priority for the second-dispatched half of the workgroup (it loses every arbitration by age otherwise): measured
    // -1.6 / -8.7 / -3.2 % per launch at the three levels (ablation bit 512 switches it off)
    if (!HABL(512) && wave >= 4) asm volatile("s_setprio 1");
    int k = 0;          // global stage index
    int sub_base = 0;   // first sub-slab of the super-iteration BEFORE the current one (the schedule constants are taken one period in)
    read_frags(std::integral_constant<int, 0>{}, 0);
    for (;; sub_base += C::SUBS) {
        const bool done = static_for_until<0, C::SUPER>([&](auto js_tag) -> bool {
            constexpr int JS = decltype(js_tag)::value;
            static_for_until<0, PPS>([&](auto i_tag) -> bool {
                constexpr int I = decltype(i_tag)::value;
                constexpr int CUR = (JS * PPS + I) & 1;
                constexpr int QB = JS * NP + I * KG;      // k-group 0's pair within the super-iteration (mine: + kgroup)
                if constexpr (I < PPS - 1) {
                    mfmas(CUR, 0, 1);
                    read_frags(std::integral_constant<int, QB + KG>{}, CUR ^ 1);
                    mfmas(CUR, 1, NMF);
                    if (!HABL(64)) {   // one fragment read behind each of the first 5 + NB MFMAs, the other MFMAs after them
#pragma unroll
                        for (int r = 0; r < 5 + NB; ++r) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, NMF - (5 + NB), 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my fragment reads of this stage have landed
                    if (w_loader) {
                        hwait_vmcnt<W_IT*(NWS - 2)>();                    // stage k + 1 has landed, NWS - 2 later ones stay in flight
                    } else {
                        if constexpr (C::a_wait_here(C::SUPER + JS)) hwait_vmcnt<A_IT * C::a_younger(C::SUPER + JS)>();
                    }
                    if (!HABL(8)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    mfmas(CUR, 0, 1);
                    read_frags(std::integral_constant<int, (JS + 1) * NP>{}, CUR ^ 1);
                    if (w_loader) {
                        if (!HABL(1)) issue_w(k + NWS, (JS % NWS) * C::WS_BYTES);   // every wave is past its reads of stage k: the slot is free
                    } else {
                        constexpr int CNT = C::issue_count(C::SUPER + JS), FIRST = C::first_issue(C::SUPER + JS);
#pragma unroll
                        for (int c = 0; c < CNT; ++c)
                            if (!HABL(2)) issue_a(sub_base - C::SUBS + FIRST + c, ((FIRST + c) % NA) * C::A_BYTES);
                    }
                    mfmas(CUR, 1, NMF);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return false;
            });
            return ++k == p.nstage;
        });
        if (done) break;
    }
    hwait_vmcnt<0>();   // the loaders ran ahead: nothing may still be landing when the LDS is reused
    if (HABL(256)) { if (acc[0][0][0] == 12345.678f) ((float*)d.out)[0] = acc[1][1][1]; return; }

    // ---- k-groups summed through LDS, then the workgroup-level epilogue (tile80.h) ------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the ring is dead
    asm volatile("" ::: "memory");
    tile80::reduce_kgroups<KG, WM * WN, NB>(smem, acc, kgroup, wv, lane);
    tile80::epilogue<BM, BN, NB>(
        smem, d, acc, kgroup == 0, wave_n * WTN + lq * 4, tid, n0,
        [&](int bm) { return (rg * 5 + bm) * Fx + half * 16 + l15; },                       // tile order: row-major over the S x FX rectangle
        [&](int r) { const int s = r / Fx; return token_of(s, r - s * Fx); }, HABL(32));
#endif
}

template <int WM, int WN, int KG, int NP, int NWS, int NA, int NWL, int A_IT, int FX, int NB = 5>
int halo_launch(HaloParams& p, hipStream_t s) {
    using C = HaloCfg<WM, WN, KG, NP, NWS, NA, NWL, A_IT, FX, NB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_halo_kernel<WM, WN, KG, NP, NWS, NA, NWL, A_IT, FX, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<WM, WN, KG, NP, NWS, NA, NWL, A_IT, FX, NB>), dim3(p.tiles_m * p.tiles_n), dim3(512), C::SMEM, s, p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

struct HaloTile { int s, fx, bn, np, na; };
// ids 1..5 (tile_cfg 40..44): 320x160 (10 rows x 32, one k-group), 320x80 (two k-groups), 160x80 on a 16-wide grid (10 x 16, four
// k-groups), 160x80 on a 32-wide grid (5 x 32, four k-groups)
// id 5 (tile_cfg 44): 320x128 on 80 x 64 wave tiles (4 channel blocks per wave: 9 fragment reads per 20 MFMAs instead of 10 per 25) for
// widths that are multiples of 128 but not of 80 — the KL-VAE decoder's 128 / 256 / 512 channels, which the 80-wide tiles pad by a fifth
const HaloTile kHalo[] = {{0, 0, 0, 0, 0}, {10, 32, 160, 2, 2}, {10, 32, 80, 2, 2}, {10, 16, 80, 4, 4}, {5, 32, 80, 4, 4}, {10, 32, 128, 2, 2}};
constexpr int kNumHalo = 5;

int halo_dispatch(int cfg, HaloParams& p, hipStream_t s) {
    switch (cfg) {
        //                  WM WN KG NP NWS NA NWL A_IT FX
        case 1: return halo_launch<4, 2, 1, 2, 3, 2, 4, 8, 32>(p, s);
        case 2: return halo_launch<4, 1, 2, 2, 6, 2, 2, 5, 32>(p, s);
        case 3: return halo_launch<2, 1, 4, 4, 3, 4, 4, 5, 16>(p, s);
        case 4: return halo_launch<2, 1, 4, 4, 3, 4, 4, 5, 32>(p, s);
        case 5: return halo_launch<4, 2, 1, 2, 3, 2, 4, 8, 32, 4>(p, s);
        default: return T2V_EINVAL;
    }
}

}  // namespace

static int g_halo_force = 0, g_halo_debug = 0;
extern "C" int t2v_conv_halo_force_config(int cfg) { g_halo_force = cfg; return T2V_OK; }
extern "C" int t2v_conv_halo_debug(int bits) { g_halo_debug = bits; return T2V_OK; }
// columns (elements) a slab-major pack row must have for a conv of C input channels: an EVEN number of the widest weight stage (4 pairs)
extern "C" int t2v_conv_halo_pack_cols(int channels) { return ((channels / 32 * 9 + 7) / 8) * 8 * 32; }

// Geometry + tile choice.  Returns T2V_OK with cfg > 0 if the halo kernel takes the launch, cfg = 0 if it does not (the caller
// uses t2v_gemm with the tap-major pack), a negative code on an invalid descriptor.
static int halo_prepare(const t2v_gemm_desc* dd, HaloParams& p, int& cfg) {
    cfg = 0;
    T2V_REQUIRE(dd && dd->a0 && dd->w && dd->out, T2V_EINVAL, "t2v_conv_halo: null pointer");
    T2V_REQUIRE(!dd->ln_in && !dd->gn_coef, T2V_EINVAL, "t2v_conv_halo: ln_in / gn_coef are t2v_linear_pr's");
    p.d = *dd;
    t2v_gemm_desc& d = p.d;
    if (!d.a1) { d.c1 = 0; d.lda1 = 0; }
    if (d.mode != T2V_GEMM_CONV3X3 && d.mode != T2V_GEMM_CONV3X3_UP2) return T2V_OK;
    p.ups = d.mode == T2V_GEMM_CONV3X3_UP2 ? 1 : 0;
    if (d.batch > 1 || d.alpha != 1.0f || d.out_f32 || d.split_k > 1 || d.drop_thr || d.ln_out || d.rowstat_out || d.lnf_stats || d.lora_t ||
        (d.act != T2V_ACT_NONE && d.act != T2V_ACT_SILU))
        return T2V_OK;
    if (d.c0 <= 0 || d.c0 % 64 || d.c1 % 64 || d.N % 16 || d.lda0 % 8 || d.lda1 % 8 || d.ldw % 8 || d.ldo % 8) return T2V_OK;
    if (((uintptr_t)d.a0 | (uintptr_t)d.w | (uintptr_t)d.out | (uintptr_t)d.a1) % 16) return T2V_OK;
    if (d.residual && (d.ldr % 8 || (uintptr_t)d.residual % 16)) return T2V_OK;
    if (d.bias && (uintptr_t)d.bias % 16) return T2V_OK;
    if (d.rowvec && ((uintptr_t)d.rowvec % 16 || d.ld_rowvec % 4 || d.rowvec_div <= 0 || d.rowvec_div % ((d.h_in << p.ups) * (d.w_in << p.ups)))) return T2V_OK;
    T2V_REQUIRE(d.n_img > 0 && d.h_in > 0 && d.w_in > 0, T2V_EINVAL, "t2v_conv_halo: conv geometry");
    const int C = d.c0 + d.c1;
    p.U = d.h_in << p.ups; p.V = d.w_in << p.ups;
    const int n_img = d.n_img;
    T2V_REQUIRE((long long)d.M == (long long)n_img * p.U * p.V, T2V_EINVAL, "t2v_conv_halo: M does not match the geometry");
    const int T = 9;
    p.nsub = C / 32;
    p.nq = p.nsub * T;
    T2V_REQUIRE(d.ldw >= t2v_conv_halo_pack_cols(C), T2V_EINVAL, "t2v_conv_halo: ldw < the padded pack width (t2v_conv_halo_pack_cols)");
    // the DMA's per-lane byte offsets are 31-bit
    if (((long long)d.M >> (2 * p.ups)) * (d.lda0 > d.lda1 ? d.lda0 : d.lda1) * 2 >= (1ll << 31) || (long long)d.N * d.ldw * 2 >= (1ll << 31)) return T2V_OK;
    // tile choice: the largest tile whose grid fills >= 85 % of a whole number of 256-CU rounds; otherwise the best filler
    const int pick = d.tile_cfg >= 40 && d.tile_cfg < 40 + kNumHalo ? d.tile_cfg - 39 : g_halo_force;
    double best = -1.0;
    int best_id = 0;
    for (int id = 1; id <= kNumHalo; ++id) {
        if (pick && id != pick) continue;
        const HaloTile& t = kHalo[id];
        if (p.V % t.fx || (t.fx == 16 && p.V >= 32)) continue;
        if (t.s >= 2 * p.U) continue;   // tile more than twice the image
        const int bm = t.s * t.fx;
        const long long tiles = (long long)n_img * ((p.U + t.s - 1) / t.s) * (p.V / t.fx) * ((d.N + t.bn - 1) / t.bn);
        const double rounds = (double)((tiles + 255) / 256);
        double eff = (double)tiles / (256.0 * rounds);
        eff *= (double)p.U / (double)(((p.U + t.s - 1) / t.s) * t.s) * (double)d.N / (double)(((d.N + t.bn - 1) / t.bn) * t.bn);
        const double score = eff >= 0.85 ? 2.0 + (double)(bm * t.bn) * 1e-6 : eff;
        if (score > best) { best = score; best_id = id; }
    }
    if (!best_id) return T2V_OK;
    const HaloTile& t = kHalo[best_id];
    p.tiles_s = (p.U + t.s - 1) / t.s;
    p.tiles_f = p.V / t.fx;
    p.tiles_m = n_img * p.tiles_s * p.tiles_f;
    p.tiles_n = (d.N + t.bn - 1) / t.bn;
    p.nstage = (p.nq + t.np - 1) / t.np;
    if (d.colstat_out) {   // whole slabs only: every tile row inside the image, a tile-order slab = 32 consecutive global rows
        if (p.U % t.s || d.M % 32 || d.N % 2 || (uintptr_t)d.colstat_out % 16) return T2V_OK;
        if (t.fx == 16 && (p.V != 16 || t.s % 2 || (p.U * p.V) % 32)) return T2V_OK;
    }
    {   // XCD grid over the tile grid: minimise the bytes the eight private L2s pull in total
        const double a_bytes = (double)d.M * C, w_bytes = (double)d.N * T * C;
        double bestc = 1e300;
        p.xcd_m = 8; p.xcd_n = 1;
        for (int xm = 8; xm >= 1; xm >>= 1) {
            const int xn = 8 / xm;
            if (xm > p.tiles_m || xn > p.tiles_n) continue;
            const double cost = xn * a_bytes + xm * w_bytes;
            if (cost < bestc) { bestc = cost; p.xcd_m = xm; p.xcd_n = xn; }
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
    p.debug = g_halo_debug;
    cfg = best_id;
    return T2V_OK;
}

extern "C" int t2v_conv_halo_supported(const t2v_gemm_desc* dd) {
    HaloParams p;
    int cfg = 0;
    const int rc = halo_prepare(dd, p, cfg);
    return rc != T2V_OK ? rc : (cfg > 0 ? 1 : 0);
}

extern "C" int t2v_conv_halo(const t2v_gemm_desc* dd, void* stream) {
    HaloParams p;
    int cfg = 0;
    const int rc = halo_prepare(dd, p, cfg);
    if (rc != T2V_OK) return rc;
    T2V_REQUIRE(cfg > 0, T2V_ESHAPE, "t2v_conv_halo: this launch is not taken by the halo kernel (ask t2v_conv_halo_supported first)");
    return halo_dispatch(cfg, p, (hipStream_t)stream);
}
