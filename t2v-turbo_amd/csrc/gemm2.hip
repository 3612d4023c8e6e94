// t2v_gemm, second kernel family: the STATIC-SCHEDULE main loop of conv_halo.hip for operands that are restaged per K step
// (linear layers, the (3,1,1) temporal conv).  gfx950, bf16 MFMA, fp32 accumulate.
//
//   out[M,N] = epilogue( gather(A)[M,K] x W[N,K]^T ),   K walked in "pairs" of 32 (one v_mfma_f32_16x16x32_bf16 deep)
//
// What it shares with conv_halo.hip (tile80.h): 8 waves, every wave an 80 x 80 tile of 5 x 5 MFMA blocks (0.4 ds_read_b128 per
// MFMA, balanced over the four SIMDs), raw-buffer LDS-DMA whose per-lane offsets are constants and whose advance is a scalar,
// a ring of NWS stages unrolled over a super-iteration so that every LDS offset and every s_waitcnt count is a compile-time
// constant, loaders that run ahead blindly past K (drained before the LDS is reused), one s_barrier per stage placed before the
// last pair's MFMAs, fp32 row-pass epilogue with the residual added before the one rounding.  What differs: both operands are
// ring-staged ([pair][BM rows | BN rows][64 B] per slot); the weight pack is t2v_gemm's own ([N][K], tap-major for the temporal
// conv): no repack, the same descriptor, and t2v_gemm routes here by itself (gemm.hip) — tile ids 50 (320x160, one k-group) and
// 51 (160x160, two k-groups).
//
// Why (round 4): the round-3 kernel's loop carries 3 VALU + 2 SALU per MFMA and runs its two waves per SIMD in lock-step
// (profiles/r04_gemm_pmc_long_k_families.csv: MFMA busy 36-45 % on active CUs); the same arithmetic on this schedule measured
// 20-30 % faster for the 3x3 convs.  The short-K launches (K <= 640) stay on the old tiles: their time is prologue / epilogue, and
// one 512-thread workgroup per CU cannot overlap those with another workgroup's loop.
#include "tile80.h"
#include "gemm2.h"
#include <cstdlib>
#include <type_traits>

namespace {

using tile80::kOutOfRange;
using tile80::static_for_until;

// NAL waves stage the activation rows (A_IT pieces of 1 KiB per stage each), the next NWL the weight rows (W_IT each)
template <int WM, int WN, int KG, int NP, int NWS, int NAL, int NWL>
struct G2Cfg {
    static_assert(WM * WN * KG == 8, "eight waves");
    static constexpr int BM = WM * 80, BN = WN * 80;
    static constexpr int A_PIECES = NP * BM / 16, W_PIECES = NP * BN / 16;
    static_assert(A_PIECES % NAL == 0 && W_PIECES % NWL == 0 && NAL + NWL <= 8, "loader waves");
    static constexpr int A_IT = A_PIECES / NAL, W_IT = W_PIECES / NWL;
    static constexpr int PAIR_BYTES = (BM + BN) * 64;              // [BM activation rows | BN weight rows] x 64 B
    static constexpr int SLOT_BYTES = NP * PAIR_BYTES;
    static constexpr int RING_BYTES = NWS * SLOT_BYTES;
    static constexpr int OUT_BYTES = tile80::Epi<BM, BN>::BYTES;
    static constexpr int RED_BYTES = (KG - 1) * WM * WN * 25 * 1024;
    static constexpr int SMEM = RING_BYTES > OUT_BYTES ? (RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES) : (OUT_BYTES > RED_BYTES ? OUT_BYTES : RED_BYTES);
    static_assert(SMEM <= 160 * 1024, "LDS");
    static_assert(NP % KG == 0, "every k-group takes the same number of pairs of a stage");
    static constexpr int PPS = NP / KG;
    static constexpr int SUPER = NWS * ((PPS % 2) ? 2 : 1);        // stages per unrolled super-iteration: ring slots and fragment sets periodic
    static_assert(A_IT * (NWS - 1) <= 60 && W_IT * (NWS - 1) <= 60, "vmcnt field");
};

template <int WM, int WN, int KG, int NP, int NWS, int NAL, int NWL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm2_kernel(const Gemm2Params p) {
    // (the host pass of hipcc does not know the buffer-descriptor builtins and would silently drop the kernel's stub: it sees an empty body)
#if defined(__HIP_DEVICE_COMPILE__) || defined(T2V_HOSTSIM)
    using C = G2Cfg<WM, WN, KG, NP, NWS, NAL, NWL>;
    constexpr int BM = C::BM, BN = C::BN, A_IT = C::A_IT, W_IT = C::W_IT, PPS = C::PPS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const t2v_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgroup = wave / (WM * WN), wv = wave % (WM * WN);
    const int wave_m = wv / WN, wave_n = wv % WN;
    const bool a_loader = wave < NAL, w_loader = !a_loader && wave < NAL + NWL;
    const int lw = a_loader ? wave : wave - NAL;

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
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader bookkeeping: raw-buffer LDS-DMA (see conv_halo.hip) ---------------------------------------------------------
    // activation loaders: piece i = lw + NAL j of the stage image: pair i / (BM/16), rows 16 (i % (BM/16)) + lane/4 of the tile;
    // per tap t a lane's BYTE offset of (source row, my chunk) — LINEAR: one "tap", and a second set for the second source of a
    // virtual concat; TCONV3: the row one frame earlier / this frame / one frame later (out of range at the clip's ends).
    // weight loaders: piece i = lw + NWL j: pair i / (BN/16), rows 16 (i % (BN/16)) + lane/4 of the channel tile.
    constexpr int L_IT = A_IT > W_IT ? A_IT : W_IT;
    // ONE per-lane array: activation loaders keep the source ROW of each of their pieces (< 0: past M -> zeros), the byte offset is
    // row * (2 lda) + chunk at issue time (one v_mad_u32_u24: the row stride differs between the two sources of a virtual concat
    // and the temporal taps shift the row by whole frames); weight loaders keep the byte offset itself.  A select between several
    // such arrays on a run-time tap / source index would send them to scratch.
    int ldv[L_IT];
    unsigned edge = 0;   // TCONV3: bit 2j: piece j's row is in the first frame of its clip (tap 0 is padding), bit 2j+1: in the last
    const int ld_ch16 = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, (unsigned)d.N * (unsigned)d.ldw * 2u, 0x00020000);
    const long long a_rows = p.taps == 1 ? (long long)d.M : (long long)d.n_img * d.h_in * d.w_in;
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.a0, 0, (unsigned)(a_rows * d.lda0 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = d.a1 ? __builtin_amdgcn_make_buffer_rsrc((void*)d.a1, 0, (unsigned)(a_rows * d.lda1 * 2), 0x00020000) : rs_a0;
#pragma unroll
    for (int j = 0; j < L_IT; ++j) ldv[j] = -1;
    if (a_loader) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int r16 = (lw + NAL * j) % (BM / 16);
            const int m = m0 + r16 * 16 + (lane >> 2);
            if (m < d.M) {
                ldv[j] = m;
                if (p.taps > 1) {   // (3,1,1): token (clip, f, pixel): the taps are frames f - 1, f, f + 1 of the same pixel
                    const int f = (m / p.tap_stride) % p.frames;
                    edge |= (f == 0 ? 1u : 0u) << (2 * j) | (f + 1 == p.frames ? 2u : 0u) << (2 * j);
                }
            }
        }
    } else if (w_loader) {
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int i = lw + NWL * j;
            const int pair = i / (BN / 16), r16 = i - pair * (BN / 16);
            const int n = n0 + r16 * 16 + (lane >> 2);
            ldv[j] = n < d.N ? (n * d.ldw + pair * 32) * 2 + ld_ch16 : (int)kOutOfRange;
        }
    }
    // stage `stage` (its NP pairs) into the ring slot at byte offset slot_off.  Past the last stage (the loaders run ahead blindly,
    // so that the DMA counts never change) the last stage is fetched again: nothing is read outside the operands.
    auto issue = [&](int stage_raw, int slot_off) {
        const int stage = min(stage_raw, p.nstage - 1);
        if (a_loader) {
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                const int i = lw + NAL * j;
                const int pr = i / (BM / 16), r16 = i - pr * (BM / 16);
                char* dst = smem + slot_off + pr * C::PAIR_BYTES + r16 * 1024;
                const int q = stage * NP + pr;                      // K pair -> (tap, 32-channel column)
                int tap = 1, col = q * 32;
                if (p.taps > 1) { tap = q >= 2 * p.nsub ? 2 : (q >= p.nsub ? 1 : 0); col = (q - tap * p.nsub) * 32; }
                const bool second = p.taps == 1 && col >= d.c0 && d.a1;
                const int soff = (second ? col - d.c0 : col) * 2;
                const int ld2 = (second ? d.lda1 : d.lda0) * 2;
                const unsigned pad_bit = tap == 0 ? (1u << (2 * j)) : (tap == 2 ? (2u << (2 * j)) : 0u);
                const int row = ldv[j] + (tap - 1) * p.tap_stride;
                const unsigned voff = (ldv[j] >= 0 && !(edge & pad_bit)) ? (unsigned)row * (unsigned)ld2 + (unsigned)ld_ch16 : kOutOfRange;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs_a1 : rs_a0, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
            }
        } else if (w_loader) {
            const int soff = stage * (NP * 64);
#pragma unroll
            for (int j = 0; j < W_IT; ++j) {
                const int i = lw + NWL * j;
                const int pr = i / (BN / 16), r16 = i - pr * (BN / 16);
                char* dst = smem + slot_off + pr * C::PAIR_BYTES + BM * 64 + r16 * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, ldv[j], soff, 0, 0);
            }
        }
    };

    // ---- fragment addressing: everything but one lane constant per operand is compile-time ------------------------------------
    // row r of a [rows][64 B] tile holds chunk c at slot c ^ 2 ((r >> 2) & 1); both operands' blocks start at multiples of 16 rows
    const int lane_c = l15 * 64 + ((lq ^ (((l15 >> 2) & 1) << 1)) << 4) + kgroup * C::PAIR_BYTES;
    const int a_lane = lane_c + wave_m * 80 * 64;
    const int w_lane = lane_c + BM * 64 + wave_n * 80 * 64;
    bf16x8_t fa[2][5], fw[2][5];
    f32x4_t acc[5][5];   // [channel block][token block]
    auto read_frags = [&](auto qb_tag, int which) {   // QB: k-group 0's pair within the super-iteration (mine: + kgroup, folded into the lane constants)
        constexpr int QB = decltype(qb_tag)::value % (C::SUPER * NP);
        constexpr int OFF = ((QB / NP) % NWS) * C::SLOT_BYTES + (QB % NP) * C::PAIR_BYTES;
        const char* ab = smem + OFF + a_lane;
        const char* wb = smem + OFF + w_lane;
#pragma unroll
        for (int bm = 0; bm < 5; ++bm) fa[which][bm] = *(const bf16x8_t*)(ab + bm * 1024);
#pragma unroll
        for (int bn = 0; bn < 5; ++bn) fw[which][bn] = *(const bf16x8_t*)(wb + bn * 1024);
    };
    auto mfmas = [&](int which, int first, int last) {
#pragma unroll
        for (int bn = 0; bn < 5; ++bn)
#pragma unroll
            for (int bm = 0; bm < 5; ++bm)
                if (bn * 5 + bm >= first && bn * 5 + bm < last)
                    acc[bn][bm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[which][bn], fa[which][bm], acc[bn][bm], 0, 0, 0);
    };

    // ---- prologue --------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < NWS; ++st) issue(st, st * C::SLOT_BYTES);
    const int ch_lane = n0 + wave_n * 80 + lq * 4;
#pragma unroll
    for (int bn = 0; bn < 5; ++bn) {
        const int ch = ch_lane + bn * 16;
        const float4 v = (kgroup == 0 && d.bias && ch < d.N) ? *(const float4*)(d.bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int bm = 0; bm < 5; ++bm) acc[bn][bm] = (f32x4_t){v.x, v.y, v.z, v.w};
    }
    tile80::wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave >= 4) asm volatile("s_setprio 1");   // static priority for the second-dispatched half (conv_halo.hip)

    // ---- main loop (see conv_halo.hip: same hand-over, here both operands come from the ring slot of the stage) ----------------
    int k = 0;
    read_frags(std::integral_constant<int, 0>{}, 0);
    for (;;) {
        const bool done = static_for_until<0, C::SUPER>([&](auto js_tag) -> bool {
            constexpr int JS = decltype(js_tag)::value;
            static_for_until<0, PPS>([&](auto i_tag) -> bool {
                constexpr int I = decltype(i_tag)::value;
                constexpr int CUR = (JS * PPS + I) & 1;
                constexpr int QB = JS * NP + I * KG;
                if constexpr (I < PPS - 1) {
                    mfmas(CUR, 0, 1);
                    read_frags(std::integral_constant<int, QB + KG>{}, CUR ^ 1);
                    mfmas(CUR, 1, 25);
#pragma unroll
                    for (int r = 0; r < 10; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my fragment reads of this stage have landed
                    if (a_loader) tile80::wait_vmcnt<A_IT*(NWS - 2)>();     // my part of stage k + 1 has landed, NWS - 2 later stages stay in flight
                    else if (w_loader) tile80::wait_vmcnt<W_IT*(NWS - 2)>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    mfmas(CUR, 0, 1);
                    read_frags(std::integral_constant<int, (JS + 1) * NP>{}, CUR ^ 1);
                    issue(k + NWS, (JS % NWS) * C::SLOT_BYTES);   // every wave is past its reads of stage k: the slot is free
                    mfmas(CUR, 1, 25);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return false;
            });
            return ++k == p.nstage;
        });
        if (done) break;
    }
    tile80::wait_vmcnt<0>();   // the loaders ran ahead: nothing may still be landing when the LDS is reused
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the ring is dead
    asm volatile("" ::: "memory");
    tile80::reduce_kgroups<KG, WM * WN, 5>(smem, acc, kgroup, wv, lane);
    tile80::epilogue<BM, BN, 5>(
        smem, d, acc, kgroup == 0, wave_n * 80 + lq * 4, tid, n0,
        [&](int bm) { return wave_m * 80 + bm * 16 + l15; },
        [&](int r) { return m0 + r < d.M ? m0 + r : -1; }, false);
#endif
}

template <int WM, int WN, int KG, int NP, int NWS, int NAL, int NWL>
int g2_launch(Gemm2Params& p, hipStream_t s) {
    using C = G2Cfg<WM, WN, KG, NP, NWS, NAL, NWL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm2_kernel<WM, WN, KG, NP, NWS, NAL, NWL>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm2_kernel<WM, WN, KG, NP, NWS, NAL, NWL>), dim3(p.tiles_m * p.tiles_n), dim3(512), C::SMEM, s, p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

struct G2Tile { int bm, bn, np; };
const G2Tile kG2[] = {{0, 0, 0}, {320, 160, 1}, {160, 160, 2}};   // tile ids 50, 51
constexpr int kNumG2 = 2;

}  // namespace

// Decide whether the second kernel family takes a t2v_gemm launch (cfg > 0) and prepare it.  `forced`: tile id 50 / 51 asked for
// (t2v_gemm_force_config or the descriptor's tile_cfg); otherwise the library's own rule: long enough K, a grid that fills the
// chip in whole rounds, an epilogue it implements.
int t2v_gemm2_prepare(const t2v_gemm_desc* dd, Gemm2Params& p, int forced, int& cfg) {
    cfg = 0;
    p.d = *dd;
    t2v_gemm_desc& d = p.d;
    if (!d.a1) { d.c1 = 0; d.lda1 = 0; }
    if (d.mode != T2V_GEMM_LINEAR && d.mode != T2V_GEMM_TCONV3) return T2V_OK;
    if (d.batch > 1 || d.alpha != 1.0f || d.out_f32 || d.split_k > 1 || d.drop_thr || d.ln_out || d.rowstat_out || d.lnf_stats || d.lora_t || d.rowvec ||
        (d.act != T2V_ACT_NONE && d.act != T2V_ACT_SILU))
        return T2V_OK;
    if (d.c0 <= 0 || d.c0 % 64 || d.c1 % 64 || d.N % 16 || d.lda0 % 8 || d.lda1 % 8 || d.ldw % 8 || d.ldo % 8) return T2V_OK;
    if (((uintptr_t)d.a0 | (uintptr_t)d.w | (uintptr_t)d.out | (uintptr_t)d.a1) % 16) return T2V_OK;
    if (d.residual && (d.ldr % 8 || (uintptr_t)d.residual % 16)) return T2V_OK;
    if (d.bias && (uintptr_t)d.bias % 16) return T2V_OK;
    const int C = d.c0 + d.c1;
    p.taps = 1; p.tap_stride = 0; p.frames = 1;
    if (d.mode == T2V_GEMM_TCONV3) {
        if (d.a1 || d.frames <= 0 || d.n_img % d.frames || (long long)d.M != (long long)d.n_img * d.h_in * d.w_in) return T2V_OK;
        p.taps = 3; p.tap_stride = d.h_in * d.w_in; p.frames = d.frames;
    }
    p.nsub = C / 32;
    p.nq = p.taps * p.nsub;
    if (d.ldw < p.taps * C) return T2V_OK;
    // the DMA's per-lane byte offsets are 31-bit
    const long long rows = d.mode == T2V_GEMM_LINEAR ? d.M : (long long)d.n_img * d.h_in * d.w_in;
    if (rows * (d.lda0 > d.lda1 ? d.lda0 : d.lda1) * 2 >= (1ll << 31) || (long long)d.N * d.ldw * 2 >= (1ll << 31)) return T2V_OK;
    if (d.colstat_out && (d.M % 32 || d.N % 2 || (uintptr_t)d.colstat_out % 16)) return T2V_OK;
    int pick = forced;
    if (!pick) {
        // library rule: K long enough that the loop, not the per-workgroup fixed cost, is what is timed (measured: profiles/r04_gemm2_*.csv),
        // and the largest tile whose grid fills >= 85 % of whole 256-CU rounds without padding N or M by more than 10 %
        if (p.nq * 32 < 1280 && !(d.mode == T2V_GEMM_TCONV3 && p.nq * 32 >= 960)) return T2V_OK;
        double best = 0.0;
        for (int id = 1; id <= kNumG2; ++id) {
            const G2Tile& t = kG2[id];
            if (p.nq % t.np) continue;
            const long long tm = (d.M + t.bm - 1) / t.bm, tn = (d.N + t.bn - 1) / t.bn, tiles = tm * tn;
            const double fill = (double)tiles / (256.0 * (double)((tiles + 255) / 256));
            const double pad = (double)d.M / (double)(tm * t.bm) * (double)d.N / (double)(tn * t.bn);
            if (fill * pad < 0.85) continue;
            const double score = fill * pad + 1e-7 * t.bm * t.bn;
            if (score > best) { best = score; pick = id; }
        }
        if (!pick) return T2V_OK;
    }
    const G2Tile& t = kG2[pick];
    if (p.nq % t.np) return T2V_OK;
    p.tiles_m = (d.M + t.bm - 1) / t.bm;
    p.tiles_n = (d.N + t.bn - 1) / t.bn;
    p.nstage = p.nq / t.np;
    {   // XCD grid over the tile grid: minimise the bytes the eight private L2s pull in total
        const double a_bytes = (double)d.M * C * p.taps, w_bytes = (double)d.N * p.taps * C;
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
    cfg = pick;
    return T2V_OK;
}

int t2v_gemm2_dispatch(int cfg, Gemm2Params& p, hipStream_t s) {
    switch (cfg) {
        //               WM WN KG NP NWS NAL NWL
        case 1: return g2_launch<4, 2, 1, 1, 4, 5, 2>(p, s);
        case 2: return g2_launch<2, 2, 2, 2, 3, 4, 4>(p, s);
        default: return T2V_EINVAL;
    }
}
