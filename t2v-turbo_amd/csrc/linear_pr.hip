// Short-K linear layers with the ACTIVATION PANEL RESIDENT in LDS and the weights streamed straight into registers
// (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   out[M, N'] = epilogue( A[M, K] x W[N, K]^T ),   K = 320 / 640 (the transformer blocks of the 40x64 / 20x32 levels)
//
// Replaces t2v_gemm on: the GEGLU projection (lvdm/modules/attention.py:516-523), q | k | v and q | k (:71-76),
// to_out (:164), the feed-forward output projection where K is short, proj_in / proj_out (:373-389, :471-513).
//
// Why a second linear kernel (round 6).  t2v_gemm stages BOTH operands through an LDS ring with one s_barrier per K step:
// at K = 320 a workgroup tile is fill -> 5 steps -> epilogue, every wave of the CU (and every CU of the chip) in the same
// phase at the same time; the counters of round 5 (profiles/r05_gemm_pmc_short_k_linears.csv) show the matrix pipe 12-29 %
// busy and the waves parked or issue-stalled 70-80 % of their life.  Here NOTHING in the main loop is shared between waves:
//
//  * A workgroup owns BM = 32 TB token rows (160 at K = 320, 96 at K = 640) for the WHOLE launch.  Their K columns are
//    written into LDS once (100-120 KB, layout [K/8][BM][16 B]: a fragment read is two linear 512-byte runs, conflict-free)
//    and are read-only afterwards: after the one barrier behind that fill the kernel has no barrier and no LDS write.
//  * The weights come PRE-PACKED in MFMA fragment order (native.pack_linear_pr, once per weight version): chunk q = 64
//    output rows = 2 blocks of 32; [chunk][K/16 steps][2 blocks][64 lanes][16 B].  A wave owns whole chunks (q = wave,
//    wave + 8, ...), so no other wave of the CU needs its weights: each fragment is ONE coalesced 1 KiB global load
//    into registers, issued D - 1 = 3 steps ahead through a 4-slot register ring that runs on across chunk boundaries.
//    Bytes into the CU per MAC: 2 / BM (t2v_gemm's 256x128 tile: 2/256 + 2/128 — 1.9x at BM = 160).
//  * Wave tile = BM tokens x 64 rows: per 16-deep step TB activation fragments (ds_read_b128) + 2 weight fragments feed
//    2 TB MFMAs.  The waves drift apart: while one runs its epilogue (GEGLU: ~15 VALU per output, as long as the chunk's MFMAs)
//    the other wave of its SIMD has the matrix pipe to itself.  That overlap is what a lock-step tile loop cannot have.
//  * The weight rows of a 32-row block are permuted at pack time so that a lane's 16 accumulator registers are 16
//    CONSECUTIVE output channels of ONE token: bias starts the accumulators (from an LDS copy), GEGLU pairs a value block
//    with its gate block in the same lane, and the epilogue is 2 x 16-byte stores per block straight from registers.
//
// Algorithmic work per launch: 2 M N K FLOP; bytes: M K 2 (A) + N K 2 (W, L2-resident) + M N' 2 (out) [+ M N' 2 residual].
#include "gelu_poly.h"
#include <cstdlib>
#include <type_traits>

struct LprParams {
    t2v_gemm_desc d;
    int chunks;         // 64-row chunks of the packed weights (N / 64)
    int chunks_per_y;   // chunks one blockIdx.y walks (its 8 waves take them round-robin)
    int n_out;          // output columns: N, or N / 2 with GEGLU
    int debug;          // ablation bits (T2V_LPR_ABLATE builds)
};

// Ablation switches (tools only; a -DT2V_LPR_ABLATE build honours t2v_linear_pr_debug bits: 1 = no weight loads after the ring is
// primed, 2 = no MFMA, 4 = no activation fragment reads after the first, 8 = no epilogue arithmetic (raw accumulators are packed),
// 16 = no output stores, 32 = leave after the panel fill, 64 = the scalar GELU polynomial instead of the packed one).  Compiled out of the product.
#ifdef T2V_LPR_ABLATE
#define LABL(bit) (p.debug & (bit))
#else
#define LABL(bit) false
#endif

// -DT2V_LPR_TRACE (lab builds): lane 0 of every wave of workgroups 0, 1 and 100 stamps s_memtime at its phase boundaries into the
// descriptor's (otherwise unused) split-K workspace: [workgroup slot][wave][32] 64-bit words; word 31 = HW_ID.
#ifdef T2V_LPR_TRACE
#define LPR_STAMP() do { if (tr && ti < 31) { if (lane == 0) tr[ti] = __builtin_amdgcn_s_memtime(); ++ti; } } while (0)
#else
#define LPR_STAMP() do { } while (0)
#endif

namespace {

constexpr int kLprWaves = 8;
// Build-time knobs (the lab builds variants of this one file against the product's other objects: tools/build_lpr_variant.sh)
#ifndef T2V_LPR_RING
#define T2V_LPR_RING 5        // weight-fragment ring slots (steps): fragments are requested T2V_LPR_RING - 1 steps ahead; must divide K / 16
#endif
#ifndef T2V_LPR_STAGGER
#define T2V_LPR_STAGGER 0     // waves 4-7 (the second wave of every SIMD) start this many times 64 cycles late
#endif
// GELU of the gate: the scalar polynomial (14 VALU per element).  The packed form (v_pk_fma_f32: 7.5 per element) measured 2-4 % SLOWER
// here (-DT2V_LPR_PACKED_GELU): packed fp32 instructions contend with the other wave's MFMAs on the SIMD.
#ifdef T2V_LPR_PACKED_GELU
constexpr bool kLprScalarGelu = false;
#else
constexpr bool kLprScalarGelu = true;
#endif
#ifndef T2V_LPR_AF1
#define T2V_LPR_AF1 1         // 1: ONE set of activation fragments (a block's fragment of step ks + 1 is requested right behind its last MFMA of step ks)
#endif
constexpr int kLprRing = T2V_LPR_RING;
constexpr bool kLprAf1 = T2V_LPR_AF1 != 0;

template <int I, int N, class F>
__device__ __forceinline__ void lpr_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lpr_static_for<I + 1, N>(f);
    }
}

// TB: 32-token blocks per panel; KS: 16-deep K steps; EPI: 0 = bias, 1 = bias -> GEGLU (chunk = [32 value rows | 32 gate rows] -> 32
// output columns), 2 = bias + residual; PRE: what the panel fill does to the A rows on their way into LDS — 0 nothing, 1 LayerNorm
// (t2v_gemm_desc::ln_in), 2 the per-(unit, channel) affine of a GroupNorm whose statistics are known (t2v_gemm_desc::gn_coef)
template <int TB, int KS, int EPI, int PRE>
__global__ __launch_bounds__(kLprWaves * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_pr_kernel(const LprParams p) {
    constexpr int BM = 32 * TB, K = 16 * KS, D = KS % kLprRing == 0 ? kLprRing : 4;   // (K = 512: 32 steps, a four-slot ring)
    // XPF: the weight ring runs on across chunk boundaries (the first D - 1 steps of the wave's next chunk are requested in this
    // chunk's last steps and land under its epilogue).  The GEGLU epilogue leaves the registers for that; the residual epilogue
    // (32 bytes of residual per block in flight beside the accumulators) does not: there the ring is primed at every chunk start.
    constexpr bool GEGLU = EPI == 1, RES = EPI == 2;
    constexpr bool XPF = !RES;
    constexpr int PANEL_BYTES = K * BM * 2;
    constexpr int CHUNK_BYTES = KS * 2 * 1024;   // one chunk of the pack
    static_assert(KS % D == 0, "the ring must be periodic in a chunk");
    static_assert(K % 64 == 0, "panel pieces are 8 rows x 128 bytes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const t2v_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
#ifdef T2V_LPR_TRACE
    unsigned long long* tr = nullptr;
    int ti = 0;
    {
        const int slot = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? 1 : (blockIdx.x == 100 ? 2 : -1));
        if (slot >= 0 && blockIdx.y == 0 && d.ws) {
            tr = (unsigned long long*)d.ws + (slot * kLprWaves + wave) * 32;
            if (lane == 0) tr[31] = (unsigned)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        }
    }
    LPR_STAMP();   // 0: kernel entry
    unsigned long long* wg_tr = d.ws ? (unsigned long long*)d.ws + 3 * kLprWaves * 32 + (blockIdx.y * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (wg_tr && tid == 0) { wg_tr[0] = __builtin_amdgcn_s_memtime(); wg_tr[2] = (unsigned)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4); wg_tr[3] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); }
#endif
    const int c_begin = blockIdx.y * p.chunks_per_y, c_end = min(p.chunks, c_begin + p.chunks_per_y);

    // chunk walk: this wave takes chunks c_begin + wave, + 8, ... of the workgroup's run.  (Measured and dropped: starting the
    // workgroups of a launch at different chunks — no effect, the weights are not hot-spotted in L2 — and starting the second wave of
    // every SIMD late — no effect either: profiles/r06_linear_pr_variants.csv.)
    const int n_run = c_end - c_begin;
    const bool active = wave < n_run;   // (fewer chunks than waves: N = 320 has five; idle waves still fill the panel)
    int chunk = c_begin + (active ? wave : 0);

    // RES: the residual runs of the wave's chunk (lane: 16 channels of one token per block: 2 x 16 bytes), requested a whole chunk
    // ahead — the first chunk's before the panel-fill barrier, so that they land under the fill — and consumed by the chunk's first K
    // step, whose C operand is bias + residual: the epilogue itself loads nothing (no wait can queue behind its stores)
    uint4 rres[RES ? TB : 1][2][2];
    auto load_res_all = [&](int ch) {
        if constexpr (RES) {
            // (host-checked with a residual: M % 32 == 0, so a block of the panel is whole or absent; an absent block reads the
            // panel's first — never stored.  ONE per-lane offset for every block: nothing per block is kept in registers)
            unsigned loff = (unsigned)lane;
            asm volatile("" : "+v"(loff));
            loff = ((loff & 31u) * (unsigned)d.ldr + 16u * (loff >> 5)) * 2u;
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const int row0 = m0 + 32 * i < d.M ? m0 + 32 * i : m0;
                const char* rp = (const char*)d.residual + ((long long)row0 * d.ldr + ch * 64) * 2 + loff;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    rres[i][b][0] = *(const uint4*)(rp + 64 * b);
                    rres[i][b][1] = *(const uint4*)(rp + 64 * b + 16);
                }
            }
        }
    };

    // ---- panel fill: global rows (8 rows x 128 B per wave instruction: whole cache lines) -> registers -> [K/8][BM][16 B] ----------
    // lane -> row 8 rg + (lane & 7), 16-byte column 8 cg + (lane >> 3): eight consecutive lanes write 128 contiguous LDS bytes
    constexpr bool LNIN = PRE == 1, GNIN = PRE == 2;
    if constexpr (PRE == 0) {
        constexpr int CG = K / 64, RG = BM / 8, PIECES = RG * CG, PPW = (PIECES + kLprWaves - 1) / kLprWaves;
        uint4 stage[PPW];
        const bf16_t* a = (const bf16_t*)d.a0;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int pc = wave + kLprWaves * j;
            const int rg = pc / CG, cg = pc - rg * CG;
            const int row = 8 * rg + (lane & 7), c = 8 * cg + (lane >> 3);
            const int gm = m0 + row;
            // (rows past M: the last row is fetched, zeros are written — no divergent branch and no wait between the loads)
            if (pc < PIECES) stage[j] = *(const uint4*)(a + (long long)min(gm, d.M - 1) * d.lda0 + c * 8);
        }
        if (active) load_res_all(chunk);   // (lands under the fill)
        // bias of this workgroup's chunks -> LDS behind the panel (fp32; zeros without a bias)
        float* sb = (float*)(smem + PANEL_BYTES);
        const int nb = (c_end - c_begin) * 64;
        for (int i = tid; i < nb; i += kLprWaves * 64) sb[i] = d.bias ? d.bias[c_begin * 64 + i] : 0.f;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int pc = wave + kLprWaves * j;
            const int rg = pc / CG, cg = pc - rg * CG;
            const int row = 8 * rg + (lane & 7), c = 8 * cg + (lane >> 3);
            if (pc < PIECES) *(uint4*)(smem + (c * BM + row) * 16) = m0 + row < d.M ? stage[j] : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        // LayerNorm of the rows on their way into the panel (attention.py:300-311: norm1 / norm2 / norm3 of BasicTransformerBlock feed
        // exactly one Linear each where this kernel is used): the panel holds ALL K = C columns of its rows, so the statistics are the
        // workgroup's own.  A wave takes WHOLE row groups (rg = wave, wave + 8, ...: all CG pieces of 8 rows), a row's K columns then
        // sit in the 8 lanes with the same (lane & 7): two-pass statistics (mean, then centred variance — t2v_layernorm's arithmetic)
        // by three butterfly stages, the normalised values rounded to bf16 as the separate launch would have written them.
        constexpr int CG = K / 64, RG = BM / 8, RPW = (RG + kLprWaves - 1) / kLprWaves;
        uint4 stage[RPW][CG];
        const bf16_t* a = (const bf16_t*)d.a0;
#pragma unroll
        for (int jr = 0; jr < RPW; ++jr) {
            const int rg = wave + kLprWaves * jr;
            const int gm = m0 + 8 * rg + (lane & 7);
            const bf16_t* ap = a + (long long)min(gm, d.M - 1) * d.lda0 + (lane >> 3) * 8;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
                if (rg < RG) stage[jr][cg] = *(const uint4*)(ap + cg * 64);
        }
        float* sb = (float*)(smem + PANEL_BYTES);
        const int nb = (c_end - c_begin) * 64;
        for (int i = tid; i < nb; i += kLprWaves * 64) sb[i] = d.bias ? d.bias[c_begin * 64 + i] : 0.f;
        float mean[RPW], rstd[RPW];
        constexpr float inv_k = 1.0f / (float)K;
        if constexpr (LNIN) {
#pragma unroll
            for (int jr = 0; jr < RPW; ++jr) {
                if (wave + kLprWaves * jr < RG) {
                    float sm = 0.f;
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) {
                        float f[8];
                        unpack8(stage[jr][cg], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sm += f[e];
                    }
                    sm += __shfl_xor(sm, 8, 64); sm += __shfl_xor(sm, 16, 64); sm += __shfl_xor(sm, 32, 64);
                    mean[jr] = sm * inv_k;
                    float q = 0.f;
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) {
                        float f[8];
                        unpack8(stage[jr][cg], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float dlt = f[e] - mean[jr]; q += dlt * dlt; }
                    }
                    q += __shfl_xor(q, 8, 64); q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
                    rstd[jr] = rsqrtf(q * inv_k + d.ln_eps);
                }
            }
        }
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const int c = 8 * cg + (lane >> 3);
            // LayerNorm: the layer's affine; GroupNorm: rstd gamma / beta - mean rstd gamma per channel of the panel's statistics unit
            // (host-checked: a panel lies inside one unit), coef [unit][2][K] as t2v_gn_coef_cs wrote it
            const float* ga = LNIN ? d.ln_gamma : d.gn_coef + (long long)(m0 / d.gn_rows_per_unit) * 2 * K;
            const float* gb = LNIN ? d.ln_beta : ga + K;
            const float4 g0 = *(const float4*)(ga + c * 8), g1 = *(const float4*)(ga + c * 8 + 4);
            const float4 b0 = *(const float4*)(gb + c * 8), b1 = *(const float4*)(gb + c * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int jr = 0; jr < RPW; ++jr) {
                const int rg = wave + kLprWaves * jr;
                if (rg < RG) {
                    const int row = 8 * rg + (lane & 7);
                    float f[8];
                    unpack8(stage[jr][cg], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = LNIN ? (f[e] - mean[jr]) * rstd[jr] * gg[e] + bb[e] : f[e] * gg[e] + bb[e];
                    *(uint4*)(smem + (c * BM + row) * 16) = m0 + row < d.M ? pack8(f) : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
    }
    LPR_STAMP();       // 1: panel written to LDS
    __syncthreads();
    LPR_STAMP();       // 2: past the fill barrier
    if (LABL(32) || !active) return;


#if T2V_LPR_STAGGER > 0 && !defined(T2V_HOSTSIM)
    if (wave >= 4)
        for (int i = 0; i < T2V_LPR_STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
#endif
#if defined(T2V_LPR_PRIO) && !defined(T2V_HOSTSIM)
    if (wave >= 4) asm volatile("s_setprio 1");
#endif

    // ---- fragment addressing ---------------------------------------------------------------------------------------------------
    // activations (MFMA B operand: columns = tokens): lane l holds token 32 i + (l & 31), K 16 ks + 8 (l >> 5) .. + 8
    const int h = lane >> 5, l31 = lane & 31;
    const unsigned a_off = (unsigned)((h * BM + l31) * 16);   // (the ONE per-lane LDS offset: everything else is derived from it)
    const char* a_lane = smem + a_off;
    // weights (MFMA A operand: rows = channels): fragment (step, block) of the chunk is 1 KiB of the pack, lane-major
    const char* const wlane = (const char*)d.w + lane * 16;
    const char* wcur = wlane + (long long)chunk * CHUNK_BYTES;
    bf16x8_t wr[D][2], af[kLprAf1 ? 1 : 2][TB];
    f32x16_t acc[TB][2];
    auto load_w = [&](const char* base, int step, int slot) {
        if (LABL(1)) { wr[slot][0] = wr[0][0]; wr[slot][1] = wr[0][1]; return; }   // (no weight loads: every slot a copy of the first)
        wr[slot][0] = *(const bf16x8_t*)(base + (step * 2) * 1024);
        wr[slot][1] = *(const bf16x8_t*)(base + (step * 2 + 1) * 1024);
    };
    auto read_a1 = [&](int ks, int which, int i) {
        if (LABL(4) && ks > 0) return;
        af[which][i] = *(const bf16x8_t*)(a_lane + ks * (2 * BM * 16) + i * 512);
    };
    auto read_a = [&](int ks, int which) {
#pragma unroll
        for (int i = 0; i < TB; ++i) read_a1(ks, which, i);
    };
    if constexpr (XPF) {
#pragma unroll
        for (int s = 0; s < D - 1; ++s) {
            wr[s][0] = *(const bf16x8_t*)(wcur + (s * 2) * 1024);
            wr[s][1] = *(const bf16x8_t*)(wcur + (s * 2 + 1) * 1024);
        }
    }

    const float* sb = (const float*)(smem + PANEL_BYTES);
    bf16_t* const obase = (bf16_t*)d.out;
    for (;;) {
        const bool has_next = chunk + kLprWaves < c_end;
        const int nchunk = has_next ? chunk + kLprWaves : chunk;                           // (past the end: this chunk again, never used)
        const char* wnext = wlane + (long long)nchunk * CHUNK_BYTES;
        // the bias (lane = 16 consecutive channels 16 h .. of each block) is the C operand of the chunk's FIRST step: the ten
        // accumulator blocks are never initialised by copies
        f32x16_t bv[2];
        {
            // (h re-derived from the fragment address behind an opaque copy: a separate per-lane bias address would be one more
            // register live across the K loop — the one that spilled, with an s_waitcnt vmcnt(0) on its reload at every chunk top)
            unsigned aoff = a_off;
            asm volatile("" : "+v"(aoff));
            const float* bq = sb + (chunk - c_begin) * 64 + (aoff >= (unsigned)(BM * 16) ? 16 : 0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *(const float4*)(bq + 32 * b + 4 * q);
                    bv[b][4 * q] = t.x; bv[b][4 * q + 1] = t.y; bv[b][4 * q + 2] = t.z; bv[b][4 * q + 3] = t.w;
                }
        }
        if constexpr (!XPF) {   // the ring is primed per chunk: nothing of it lives across the epilogue
#pragma unroll
            for (int s = 0; s < D - 1; ++s) {
                wr[s][0] = *(const bf16x8_t*)(wcur + (s * 2) * 1024);
                wr[s][1] = *(const bf16x8_t*)(wcur + (s * 2 + 1) * 1024);
            }
        }
        read_a(0, 0);
        LPR_STAMP();   // 3 + 3 j: chunk j starts
        // K loop: D steps per iteration (ring slots and fragment sets are compile-time constants); step ks requests the weights of
        // step ks + D - 1 — in the chunk's last iteration those are the first D - 1 steps of this wave's NEXT chunk — reads the
        // activation fragments of step ks + 1, and runs its 2 TB MFMAs.  The fences keep the issue order: without them the
        // scheduler hoists every load of the unrolled body to its top and spills the accumulators.
        auto kstep = [&](const char* wb, int ks_base, auto u_tag, auto last_tag, auto first_tag) {
            constexpr int U = decltype(u_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value, FIRST = decltype(first_tag)::value && U == 0;
            static_assert(kLprAf1 || D % 2 == 0, "fragment sets alternate with the step: an even ring keeps the set of step U a constant");
            constexpr int AP = kLprAf1 ? 0 : (U & 1);
            if constexpr (LAST && U > 0) { if constexpr (XPF) load_w(wnext, U - 1, (U + D - 1) % D); }
            else load_w(wb, U + D - 1, (U + D - 1) % D);
            if constexpr (FIRST) {
                // block-major over the bias: bv[0] is dead after TB MFMAs, and the next step's fragments are requested between the
                // two halves (the accumulators come into being here: bias + fragments + ring would not fit beside all of them)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int i = 0; i < TB; ++i) {
                        f32x16_t c0 = bv[b];
                        if constexpr (RES) {
                            float r[16];
                            unpack8(rres[i][b][0], r);
                            unpack8(rres[i][b][1], r + 8);
#pragma unroll
                            for (int e = 0; e < 16; ++e) c0[e] += r[e];
                        }
                        if (LABL(2)) acc[i][b] = c0;
                        else acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[0][b], af[0][i], c0, 0, 0, 0);
                        if (kLprAf1 && b == 1) read_a1(ks_base + 1, 0, i);
                    }
                    if (!kLprAf1 && b == 0) read_a(ks_base + 1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            constexpr bool READ_NEXT = !(LAST && U == D - 1);
            if constexpr (!kLprAf1 && READ_NEXT) read_a(ks_base + U + 1, (AP + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                if (!LABL(2)) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[U % D][b], af[AP][i], acc[i][b], 0, 0, 0);
                }
                if constexpr (kLprAf1 && READ_NEXT) read_a1(ks_base + U + 1, 0, i);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        {
            using T = integral_constant<bool, true>;
            using F = integral_constant<bool, false>;
            const char* wb = wcur;
            static_assert(KS / D >= 2, "a first and a last iteration");
            lpr_static_for<0, D>([&](auto u) { kstep(wb, 0, u, F{}, T{}); });
            wb += D * 2048;
#pragma unroll 1
            for (int it = 1; it < KS / D - 1; ++it, wb += D * 2048) lpr_static_for<0, D>([&](auto u) { kstep(wb, it * D, u, F{}, F{}); });
            lpr_static_for<0, D>([&](auto u) { kstep(wb, KS - D, u, T{}, F{}); });
        }
        LPR_STAMP();   // 4 + 3 j: K loop of chunk j done
        // (the epilogue's addresses are made from an opaque copy of the chunk index: computed from `chunk` itself the compiler forms
        // all of them BEFORE the K loop and spills them across it — scratch reloads with s_waitcnt vmcnt(0) between the stores)
        int chunk_e = chunk;
        asm volatile("" : "+s"(chunk_e));
        // ---- epilogue straight from the accumulators ---------------------------------------------------------------------------------
        // A lane owns 32 bytes of ONE row (16 consecutive channels): two 16-byte stores per block.  (Measured and dropped: parking each
        // block in a wave-private LDS tile and storing 16 rows x 64 contiguous bytes per instruction — a quarter of the pieces per
        // store instruction, and 11 % SLOWER: the epilogue's length is what costs, not the shape of its stores.)
        {
            unsigned ao = a_off;   // (lane geometry re-derived here: nothing of it is kept in registers across the K loop)
            asm volatile("" : "+v"(ao));
            const unsigned he = ao >= (unsigned)(BM * 16) ? 1u : 0u, le = (ao >> 4) - he * BM;   // h, l31
            const unsigned g_off = (le * (unsigned)d.ldo + 16u * he) * 2u;                       // row l31, channels 16 h .. of a block
            // this lane's 16 channels of block (rows row0 .., columns col ..)
            auto put = [&](const float* v, int row0, int col) {
                if (row0 + (int)le < d.M && !LABL(16)) {
                    char* op = (char*)obase + ((long long)row0 * d.ldo + col) * 2 + g_off;
                    *(uint4*)op = pack8(v);
                    *(uint4*)(op + 16) = pack8(v + 8);
                }
            };
            if constexpr (GEGLU) {
#pragma unroll
                for (int i = 0; i < TB; ++i) {
                    float v[16];
                    if (LABL(8)) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = acc[i][0][e] + acc[i][1][e];
                    } else if (LABL(64) || kLprScalarGelu) {   // (the scalar polynomial: 14 VALU per element)
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = acc[i][0][e] * fast_gelu(acc[i][1][e]);
                    } else {                 // value * gelu(gate), two channels per packed instruction
#pragma unroll
                        for (int e = 0; e < 16; e += 2) {
                            const f32x2_t gg = fast_gelu2((f32x2_t){acc[i][1][e], acc[i][1][e + 1]});
                            const f32x2_t vv = (f32x2_t){acc[i][0][e], acc[i][0][e + 1]} * gg;
                            v[e] = vv[0]; v[e + 1] = vv[1];
                        }
                    }
#ifdef T2V_LPR_TRACE
                    if (chunk < c_begin + kLprWaves) { asm volatile("" :: "v"(v[0]), "v"(v[15])); LPR_STAMP(); }   // first chunk: block i's arithmetic done
#endif
                    put(v, m0 + 32 * i, chunk_e * 32);
#ifdef T2V_LPR_TRACE
                    if (chunk < c_begin + kLprWaves) LPR_STAMP();   // ... and its stores issued
#endif
                }
            } else {
                // (a residual is already inside the accumulators: it started them, with the bias)
#pragma unroll
                for (int ib = 0; ib < 2 * TB; ++ib) {
                    const int i = ib >> 1, b = ib & 1;
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc[i][b][e];
                    put(v, m0 + 32 * i, chunk_e * 64 + 32 * b);
                }
            }
        }
        LPR_STAMP();   // 5 + 3 j: epilogue of chunk j issued
#ifdef T2V_LPR_TRACE
        if (!has_next && wg_tr && lane == 0) atomicMax(&wg_tr[1], (unsigned long long)__builtin_amdgcn_s_memtime());   // the workgroup's last wave to finish
#endif
        if (!has_next) break;
        chunk = nchunk;
        wcur = wnext;
        load_res_all(chunk);

    }
}

template <int TB, int KS, int EPI, int PRE = 0>
int lpr_launch(const LprParams& p, int tiles_m, int ny, hipStream_t s) {
    const int smem = 16 * KS * 32 * TB * 2 + p.chunks_per_y * 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)linear_pr_kernel<TB, KS, EPI, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((linear_pr_kernel<TB, KS, EPI, PRE>), dim3(tiles_m, ny), dim3(kLprWaves * 64), smem, s, p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

}  // namespace

static int g_lpr_debug = 0, g_lpr_force_ny = 0;
extern "C" int t2v_linear_pr_debug(int bits) { g_lpr_debug = bits; return T2V_OK; }
extern "C" int t2v_linear_pr_force_split(int ny) { g_lpr_force_ny = ny; return T2V_OK; }

// Geometry.  cfg: 0 = not taken, 1 = K 320 on 160-row panels, 2 = K 640 on 96-row panels, 3 = K 512 on 96-row panels (the 8-head
// temporal transformer behind the entry conv, init_attn: openaimodel3d.py:441-452; GEGLU / plain epilogue, optional ln_in).
static int lpr_prepare(const t2v_gemm_desc* dd, LprParams& p, int& cfg, int& tiles_m, int& ny) {
    cfg = 0;
    T2V_REQUIRE(dd && dd->a0 && dd->w && dd->out, T2V_EINVAL, "t2v_linear_pr: null pointer");
    p.d = *dd;
    const t2v_gemm_desc& d = p.d;
    if (d.mode != T2V_GEMM_LINEAR || d.a1 || d.c1 || d.batch > 1 || d.alpha != 1.0f || d.out_f32 || d.split_k > 1 || d.drop_thr || d.ln_out ||
        d.rowstat_out || d.colstat_out || d.lnf_stats || d.lora_t || d.rowvec || (d.act != T2V_ACT_NONE && d.act != T2V_ACT_GEGLU))
        return T2V_OK;
    if (d.c0 != 320 && d.c0 != 640 && d.c0 != 512) return T2V_OK;
    if (d.M <= 0 || d.N <= 0 || d.N % 64 || d.lda0 % 8 || d.ldo % 8) return T2V_OK;
    if ((long long)d.M * d.ldo >= (1ll << 31) || (long long)d.M * d.ldr >= (1ll << 31) || (long long)d.N * d.c0 * 2 >= (1ll << 31)) return T2V_OK;   // 32-bit offsets in the kernel
    if (((uintptr_t)d.a0 | (uintptr_t)d.w | (uintptr_t)d.out) % 16) return T2V_OK;
    if (d.residual && (d.act == T2V_ACT_GEGLU || d.ldr % 8 || (uintptr_t)d.residual % 16 || d.M % 32)) return T2V_OK;
    if (d.bias && (uintptr_t)d.bias % 16) return T2V_OK;
    if (d.ln_in) {   // LayerNorm of the A rows in the panel fill: the affine is required, a residual epilogue is not combined with it
        T2V_REQUIRE(d.ln_gamma && d.ln_beta, T2V_EINVAL, "t2v_linear_pr: ln_in without ln_gamma / ln_beta");
        if (d.residual || ((uintptr_t)d.ln_gamma | (uintptr_t)d.ln_beta) % 16) return T2V_OK;
    }
    const int bm = d.c0 == 320 ? 160 : 96;
    if (d.c0 == 512 && (d.residual || d.gn_coef)) return T2V_OK;
    if (d.gn_coef) {   // GroupNorm affine in the panel fill: every panel inside one statistics unit, not combined with ln_in / a residual
        T2V_REQUIRE(!d.ln_in && d.gn_rows_per_unit > 0, T2V_EINVAL, "t2v_linear_pr: gn_coef with ln_in, or without gn_rows_per_unit");
        if (d.residual || (uintptr_t)d.gn_coef % 16 || d.gn_rows_per_unit % bm || d.M % d.gn_rows_per_unit) return T2V_OK;
    }
    p.chunks = d.N / 64;
    p.n_out = d.act == T2V_ACT_GEGLU ? d.N / 2 : d.N;
    tiles_m = (d.M + bm - 1) / bm;
    // column split: as many workgroups as fill the 256 CUs once, each with at least one chunk per wave
    ny = 1;
    if (g_lpr_force_ny > 0) ny = g_lpr_force_ny;
    else if (tiles_m < 256) {
        ny = (256 + tiles_m / 2) / tiles_m;
        const int max_ny = p.chunks / kLprWaves > 0 ? p.chunks / kLprWaves : 1;
        if (ny > max_ny) ny = max_ny;
    }
    if (ny > p.chunks) ny = p.chunks;
    if (ny < 1) ny = 1;
    p.chunks_per_y = (p.chunks + ny - 1) / ny;
    ny = (p.chunks + p.chunks_per_y - 1) / p.chunks_per_y;
    if (d.c0 * bm * 2 + p.chunks_per_y * 256 > 160 * 1024) return T2V_OK;
    p.debug = g_lpr_debug;
    cfg = d.c0 == 320 ? 1 : (d.c0 == 640 ? 2 : 3);
    return T2V_OK;
}

extern "C" int t2v_linear_pr_supported(const t2v_gemm_desc* dd) {
    LprParams p;
    int cfg = 0, tiles_m = 0, ny = 0;
    const int rc = lpr_prepare(dd, p, cfg, tiles_m, ny);
    return rc != T2V_OK ? rc : (cfg > 0 ? 1 : 0);
}

extern "C" int t2v_linear_pr(const t2v_gemm_desc* dd, void* stream) {
    LprParams p;
    int cfg = 0, tiles_m = 0, ny = 0;
    const int rc = lpr_prepare(dd, p, cfg, tiles_m, ny);
    if (rc != T2V_OK) return rc;
    T2V_REQUIRE(cfg > 0, T2V_ESHAPE, "t2v_linear_pr: this launch is not taken by the panel-resident kernel (ask t2v_linear_pr_supported first)");
    hipStream_t s = (hipStream_t)stream;
    const int epi = p.d.act == T2V_ACT_GEGLU ? 1 : (p.d.residual ? 2 : 0);
    if (cfg == 3) {
        T2V_REQUIRE(epi != 2 && !p.d.gn_coef, T2V_ESHAPE, "t2v_linear_pr: K = 512 takes the plain and the GEGLU epilogue (ask t2v_linear_pr_supported first)");
        if (p.d.ln_in) return epi == 1 ? lpr_launch<3, 32, 1, 1>(p, tiles_m, ny, s) : lpr_launch<3, 32, 0, 1>(p, tiles_m, ny, s);
        return epi == 1 ? lpr_launch<3, 32, 1>(p, tiles_m, ny, s) : lpr_launch<3, 32, 0>(p, tiles_m, ny, s);
    }
    if (p.d.ln_in) {
        if (cfg == 1) return epi == 1 ? lpr_launch<5, 20, 1, 1>(p, tiles_m, ny, s) : lpr_launch<5, 20, 0, 1>(p, tiles_m, ny, s);
        return epi == 1 ? lpr_launch<3, 40, 1, 1>(p, tiles_m, ny, s) : lpr_launch<3, 40, 0, 1>(p, tiles_m, ny, s);
    }
    if (p.d.gn_coef) {   // (the transformers' proj_in: plain epilogue)
        T2V_REQUIRE(epi == 0, T2V_ESHAPE, "t2v_linear_pr: gn_coef goes with the plain epilogue");
        return cfg == 1 ? lpr_launch<5, 20, 0, 2>(p, tiles_m, ny, s) : lpr_launch<3, 40, 0, 2>(p, tiles_m, ny, s);
    }
    if (cfg == 1) return epi == 1 ? lpr_launch<5, 20, 1>(p, tiles_m, ny, s) : (epi == 2 ? lpr_launch<5, 20, 2>(p, tiles_m, ny, s) : lpr_launch<5, 20, 0>(p, tiles_m, ny, s));
    return epi == 1 ? lpr_launch<3, 40, 1>(p, tiles_m, ny, s) : (epi == 2 ? lpr_launch<3, 40, 2>(p, tiles_m, ny, s) : lpr_launch<3, 40, 0>(p, tiles_m, ny, s));
}
