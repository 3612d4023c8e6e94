// The GEGLU feed-forward of a BasicTransformerBlock in ONE launch (gfx950):
//
//     out = x + W2 . [ value . gelu(gate) ] + b2,   [value | gate] = W1 . LayerNorm(x) + b1        (attention.py:300-311,516-542)
//
// As three launches (LayerNorm, GEGLU projection, output projection) the 4C-wide hidden activation of the 320-channel level
// (40960 x 1280 bf16 = 105 MB) is written to and read back from HBM and the projection's epilogue is a write phase as long as its
// matrix work.  Here nothing but x comes in and nothing but out goes out:
//
//  * a wave owns 48 tokens (three 16-token MFMA column tiles).  Their rows are loaded once, LayerNorm'ed in registers (a row is
//    spread over the 4 lanes of a k-group quartet: two shuffles), rounded to bf16 and KEPT as the B operands of the first GEMM
//    for the whole kernel: C/32 fragments per column tile (120 VGPRs at C = 320).
//  * the hidden dimension is walked in chunks of 32 channels.  Per chunk: S = W1[64 rows: 16 value, 16 gate, 16 value, 16 gate] . x^
//    (v_mfma_f32_16x16x32_bf16, 4 row tiles x C/32 k-steps x 3 column tiles), GEGLU in registers — value and gate of a hidden
//    channel sit in the same lane because the two row tiles are fed matching rows — and the packed result IS the B operand of the
//    second GEMM (its k order is the accumulator's register order, W2 is packed to match; two 16-deep halves,
//    v_mfma_f32_16x16x16_bf16, so that one pair's GEGLU arithmetic hides under the other's matrix work): the hidden activation
//    never leaves the registers.  O[C rows x 48 tokens] += W2[:, chunk] . P accumulates in AGPRs (240 at C = 320).
//  * the weights are the only LDS traffic: both matrices are PRE-PACKED on the host in MFMA fragment order (one 1 KiB piece =
//    one fragment of all 64 lanes), so a chunk is 60 contiguous KiB that LDS-DMA copies piece by piece and a fragment read is
//    one conflict-free ds_read_b128 at base + 16 * lane.  Double-buffered: chunk j+1 streams in under the 180 MFMAs of chunk j;
//    one barrier per chunk.  Every workgroup streams the same 2.4 MB (L2-resident) once per 192 tokens.
//  * epilogue: O goes through LDS (fp32, padded rows) once to come out row-major; bias, residual (the raw x rows) and the bf16
//    rounding happen on 16-byte row pieces with coalesced stores.
//
// Algorithmic work: 2 * M * (2 * 4C * C + 4C * C) FLOP; HBM bytes: 2 * M * C * 2 (x in, out) + the packed weights once.
#include "common.h"
#include <cstdlib>

#ifdef T2V_FFN_NOPIN   // diagnostic build: leave the issue order of reads / MFMAs / GEGLU arithmetic to the compiler
#define FFN_PIN() ((void)0)
#else
#define FFN_PIN() __builtin_amdgcn_sched_barrier(0)
#endif

namespace {

__device__ __forceinline__ void ffn_dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// exact (erf) GELU as x * Phi(x) with a degree-9 polynomial in clamp(x, +-4.5)^2 (|error| < 4.2e-5; the fit of gemm.hip's GEGLU
// epilogue, which this kernel replaces at the widths it covers)
__device__ __forceinline__ float ffn_gelu(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
    const float t = xc * xc;
    float q = -1.684528927e-12f;
    q = fmaf(q, t, 1.983680165e-10f);
    q = fmaf(q, t, -1.041734787e-08f);
    q = fmaf(q, t, 3.247777158e-07f);
    q = fmaf(q, t, -6.776206646e-06f);
    q = fmaf(q, t, 1.014731897e-04f);
    q = fmaf(q, t, -1.141749439e-03f);
    q = fmaf(q, t, 9.890335612e-03f);
    q = fmaf(q, t, -6.642110646e-02f);
    q = fmaf(q, t, 3.989264667e-01f);
    return x * fmaf(xc, q, 0.5f);
}

constexpr int FFN_WAVES = 4;                             // waves per workgroup (one per SIMD); a wave owns CT 16-token column tiles

template <int C, int CT>
struct FfnGeom {
    static constexpr int TOK = FFN_WAVES * CT * 16;      // tokens per workgroup
    static constexpr int KS = C / 32;                     // k-steps of the first GEMM (contraction over C)
    static constexpr int RT = C / 16;                     // 16-row output tiles of the second GEMM
    static constexpr int NCH = C / 8;                     // hidden chunks of 32 (hidden = 4C)
    static constexpr int W1_PIECES = 4 * KS, W2_PIECES = RT, PIECES = W1_PIECES + W2_PIECES;   // 1 KiB fragment pieces per chunk
    static constexpr int BIAS_OFF = PIECES * 1024;        // b1 of the chunk: [4 row tiles][16] fp32 = 256 bytes behind the pieces
    static constexpr int BUF = PIECES * 1024 + 256;
    static constexpr int STAGE_LD = C + 4;                // fp32 row pitch of the epilogue staging (bank spread)
    static constexpr int STAGE = 16 * STAGE_LD * 4;       // per wave
    static constexpr int SMEM = (2 * BUF > FFN_WAVES * STAGE) ? 2 * BUF : FFN_WAVES * STAGE;
};

template <int C, int FFN_CT>
__global__ __launch_bounds__(FFN_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffn_fused_kernel(const bf16_t* __restrict__ x, int ldx, int M, const bf16_t* __restrict__ w1p, const float* __restrict__ b1p,
                      const bf16_t* __restrict__ w2p, const float* __restrict__ b2, float eps, bf16_t* __restrict__ out, int ldo) {
    using G = FfnGeom<C, FFN_CT>;
    constexpr int KS = G::KS, RT = G::RT, NCH = G::NCH, FFN_TOK = G::TOK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int tok0 = blockIdx.x * FFN_TOK + wave * (FFN_CT * 16);

    // ---- chunk 0 of the weights on its way while the rows are loaded and normalised ------------------------------------------
    // wave w copies the pieces w, w + 4, ... of a chunk (W1 pieces first, then W2's) and wave 0 its 256 bytes of b1 (4 bytes per
    // lane).  A real loop with one running address: unrolled it costs an address pair per piece.
    auto issue = [&](int chunk, int buf) {
        const char* src1 = (const char*)w1p + ((size_t)chunk * G::W1_PIECES + wave) * 1024 + lane * 16;
        const char* src2 = (const char*)w2p + ((size_t)chunk * G::W2_PIECES + wave) * 1024 + lane * 16;
        char* dst = smem + buf * G::BUF + wave * 1024;
        static_assert(G::W1_PIECES % FFN_WAVES == 0 && G::W2_PIECES % FFN_WAVES == 0, "pieces per wave");
#pragma unroll 1
        for (int p = 0; p < G::W1_PIECES / FFN_WAVES; ++p) { ffn_dma16(src1, dst); src1 += FFN_WAVES * 1024; dst += FFN_WAVES * 1024; }
#pragma unroll 1
        for (int p = 0; p < G::W2_PIECES / FFN_WAVES; ++p) { ffn_dma16(src2, dst); src2 += FFN_WAVES * 1024; dst += FFN_WAVES * 1024; }
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1p + (size_t)chunk * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(smem + buf * G::BUF + G::BIAS_OFF), 4, 0, 0);
    };
    // Every workgroup streams the SAME weights: walking the chunks in the same order, the 32 CUs of an XCD would ask its L2 for
    // the same lines at the same time (one channel at a time instead of all of them).  Each workgroup therefore starts at its own
    // chunk and wraps around (the order of the fp32 accumulation over chunks differs between workgroups, never between runs).
    const int rot = blockIdx.x % NCH;
    auto chunk_of = [&](int j) { const int c = j + rot; return c >= NCH ? c - NCH : c; };
    issue(chunk_of(0), 0);

    // ---- x rows -> LayerNorm (no affine: gamma / beta are folded into W1 / b1 on the host) -> bf16 B fragments -----------------
    // lane (col, kg) holds channels [32 s + 8 kg, +8) of token col for every k-step s: a row lives in the 4 lanes of its quartet
    bf16x8_t X[FFN_CT][KS];
#pragma unroll
    for (int c = 0; c < FFN_CT; ++c) {
        const int row = min(tok0 + 16 * c + col, M - 1);
        const bf16_t* xp = x + (long long)row * ldx + 8 * kg;
        uint4 u[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) u[s] = *(const uint4*)(xp + 32 * s);
        float f[KS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            unpack8(u[s], f[s]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[s][e];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / (float)C);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[s][e] -= mean; sq = fmaf(f[s][e], f[s][e], sq); }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / (float)C) + eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[s][e] *= rstd;
            const uint4 pk = pack8(f[s]);
            X[c][s] = *(const bf16x8_t*)&pk;
        }
    }

    f32x4_t O[RT][FFN_CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < FFN_CT; ++c) O[t][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- hidden chunks ------------------------------------------------------------------------------------------------------
    // Per chunk (32 hidden channels = two value / gate pairs of 16): the weight fragments are read through a three-deep register
    // ring — the ds_read_b128 of piece n + 2 goes out before the MFMAs of piece n, so the LDS latency (longer than the three
    // MFMAs one fragment feeds) is never waited for — and the order is
    //     GEMM1 pair a | GEMM1 pair b  +  GEGLU(a) on the VALU | GEMM2 half a  +  GEGLU(b) | GEMM2 half b
    // The second GEMM runs as two 16-deep halves (v_mfma_f32_16x16x16_bf16): half a only needs pair a's activations, so the
    // GEGLU arithmetic of pair b hides under it; its A fragment is the same 16-byte piece (low / high 8 bytes).
    typedef __attribute__((__vector_size__(4 * sizeof(short)))) short bf16x4s_t;
    for (int j = 0; j < NCH; ++j) {
        const int buf = j & 1;
        if (j + 1 < NCH) issue(chunk_of(j + 1), buf ^ 1);
        const char* wb = smem + buf * G::BUF + lane * 16;
        // fragment rings: a piece feeds only CT MFMAs (16 cycles each), the LDS round trip is > 100 cycles: RD - 1 pieces in flight
        constexpr int RD1 = FFN_CT >= 3 ? 4 : 6, RD2 = FFN_CT >= 3 ? 6 : 8;
        bf16x8_t ring[RD1];
#pragma unroll
        for (int n = 0; n < RD1 - 1; ++n) ring[n] = *(const bf16x8_t*)(wb + n * 1024);
        f32x4_t S[4][FFN_CT];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int c = 0; c < FFN_CT; ++c) S[T][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x4s_t Pa[FFN_CT], Pb[FFN_CT];
        constexpr int GS = (2 * KS) / FFN_CT > 0 ? (2 * KS) / FFN_CT : 1;   // pieces of pair b per GEGLU column tile of pair a
        constexpr int G2 = RT / FFN_CT > 0 ? RT / FFN_CT : 1;               // row tiles of GEMM2 half a per GEGLU column tile of pair b
        constexpr int GO = GS > 1 ? 1 : 0, G2O = G2 > 1 ? 1 : 0;           // slot offset inside a group (after the group's first piece)
        static_assert(GS * (FFN_CT - 1) + GO < 2 * KS && G2 * (FFN_CT - 1) + G2O < RT, "every GEGLU tile must have a slot");
        auto geglu_tile = [&](int pair, int c, bf16x4s_t* P) {   // value * gelu(gate) of one pair and column tile, packed: a 16-deep B fragment
            const float* bl = (const float*)(smem + buf * G::BUF + G::BIAS_OFF) + 4 * kg;   // rows 4 kg .. +3 of [value a | gate a | value b | gate b]
            const float4 bv = *(const float4*)(bl + 32 * pair), bg = *(const float4*)(bl + 32 * pair + 16);
            const float bvr[4] = {bv.x, bv.y, bv.z, bv.w}, bgr[4] = {bg.x, bg.y, bg.z, bg.w};
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = (S[2 * pair][c][r] + bvr[r]) * ffn_gelu(S[2 * pair + 1][c][r] + bgr[r]);
            const uint2 pk = make_uint2(pack2bf(p[0], p[1]), pack2bf(p[2], p[3]));
            P[c] = *(const bf16x4s_t*)&pk;
        };
        // GEMM1, pieces 0 .. 4 KS - 1 (row tile T = piece / KS, k-step s = piece % KS).  Pair a's GEGLU is cut into its three
        // column tiles and dealt out between pair b's pieces (VALU work beside MFMAs that do not depend on it); scheduling fences
        // keep the issue order as written — the compiler otherwise hoists every fragment read to the top and spills
#pragma unroll
        for (int n = 0; n < G::W1_PIECES; ++n) {
            if (n + RD1 - 1 < G::W1_PIECES) ring[(n + RD1 - 1) % RD1] = *(const bf16x8_t*)(wb + (n + RD1 - 1) * 1024);
            const int T = n / KS, sk = n % KS;
#pragma unroll
            for (int c = 0; c < FFN_CT; ++c) S[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[n % RD1], X[c][sk], S[T][c], 0, 0, 0);
            if (n >= 2 * KS && (n - 2 * KS) % GS == GO && (n - 2 * KS) / GS < FFN_CT) geglu_tile(0, (n - 2 * KS) / GS, Pa);
            FFN_PIN();
        }
        // GEMM2 as two 16-deep halves over all row tiles: a W2 piece holds [4 k of pair a | 4 k of pair b] per lane, each pass
        // reads its 8 bytes (ds_read_b64) through its own ring; pair b's GEGLU is dealt out between half a's row tiles
        const char* w2b = wb + G::W1_PIECES * 1024;
        bf16x4s_t r2[RD2];
#pragma unroll
        for (int t = 0; t < RD2 - 1; ++t)
            if (t < RT) r2[t] = *(const bf16x4s_t*)(w2b + t * 1024);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (t + RD2 - 1 < RT) r2[(t + RD2 - 1) % RD2] = *(const bf16x4s_t*)(w2b + (t + RD2 - 1) * 1024);
#pragma unroll
            for (int c = 0; c < FFN_CT; ++c) O[t][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(r2[t % RD2], Pa[c], O[t][c], 0, 0, 0);
            if (t % G2 == G2O && t / G2 < FFN_CT) geglu_tile(1, t / G2, Pb);
            FFN_PIN();
        }
#pragma unroll
        for (int t = 0; t < RD2 - 1; ++t)
            if (t < RT) r2[t] = *(const bf16x4s_t*)(w2b + t * 1024 + 8);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (t + RD2 - 1 < RT) r2[(t + RD2 - 1) % RD2] = *(const bf16x4s_t*)(w2b + (t + RD2 - 1) * 1024 + 8);
#pragma unroll
            for (int c = 0; c < FFN_CT; ++c) O[t][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(r2[t % RD2], Pb[c], O[t][c], 0, 0, 0);
            FFN_PIN();
        }
        // my share of chunk j+1 has landed, and (after the barrier) everybody's; everybody is also done reading buffer `buf`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: O (accumulator layout: lane = token, 4-channel runs) -> LDS rows -> + b2 + x -> bf16, 16-byte stores -----------
    float* stage = (float*)(smem + wave * G::STAGE);
    constexpr int CPR = C / 8;                           // 8-channel pieces per row
#pragma unroll
    for (int c = 0; c < FFN_CT; ++c) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
            *(float4*)(stage + col * G::STAGE_LD + 16 * t + 4 * kg) = make_float4(O[t][c][0], O[t][c][1], O[t][c][2], O[t][c][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < (16 * CPR + 63) / 64; ++it) {
            const int q = it * 64 + lane;
            const int r = q / CPR, ch = (q - r * CPR) * 8;
            const int token = tok0 + 16 * c + r;
            if (q < 16 * CPR && token < M) {
                const float4 o0 = *(const float4*)(stage + r * G::STAGE_LD + ch), o1 = *(const float4*)(stage + r * G::STAGE_LD + ch + 4);
                const float4 c0 = *(const float4*)(b2 + ch), c1 = *(const float4*)(b2 + ch + 4);
                float rf[8];
                unpack8(*(const uint4*)(x + (long long)token * ldx + ch), rf);
                float v[8] = {o0.x + c0.x + rf[0], o0.y + c0.y + rf[1], o0.z + c0.z + rf[2], o0.w + c0.w + rf[3],
                              o1.x + c1.x + rf[4], o1.y + c1.y + rf[5], o1.z + c1.z + rf[6], o1.w + c1.w + rf[7]};
                *(uint4*)(out + (long long)token * ldo + ch) = pack8(v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int C, int FFN_CT>
int ffn_launch(const void* x, int ldx, int M, const void* w1p, const float* b1p, const void* w2p, const float* b2, float eps, void* out,
               int ldo, hipStream_t s) {
    using G = FfnGeom<C, FFN_CT>;
    constexpr int FFN_TOK = G::TOK;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)ffn_fused_kernel<C, FFN_CT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((ffn_fused_kernel<C, FFN_CT>), dim3((M + FFN_TOK - 1) / FFN_TOK), dim3(FFN_WAVES * 64), G::SMEM, s, (const bf16_t*)x, ldx, M,
                       (const bf16_t*)w1p, b1p, (const bf16_t*)w2p, b2, eps, (bf16_t*)out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

}  // namespace

extern "C" int t2v_ffn_fused_supported(int C) { return C == 320 || C == 64; }

extern "C" int t2v_ffn_fused(const void* x, int ldx, int M, int C, const void* w1p, const float* b1p, const void* w2p, const float* b2,
                             float ln_eps, void* out, int ldo, void* stream) {
    T2V_REQUIRE(x && w1p && b1p && w2p && b2 && out && M > 0, T2V_EINVAL, "t2v_ffn_fused: null pointer / empty problem");
    T2V_REQUIRE(ldx % 8 == 0 && ldo % 8 == 0 && ldx >= C && ldo >= C, T2V_ESHAPE, "t2v_ffn_fused: row strides must be multiples of 8");
    T2V_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1p | (uintptr_t)w2p | (uintptr_t)b1p | (uintptr_t)b2) % 16 == 0, T2V_ESHAPE,
                "t2v_ffn_fused: operands must be 16-byte aligned");
    T2V_REQUIRE(x != out, T2V_EINVAL, "t2v_ffn_fused: in-place operation is not supported (the epilogue re-reads x as the residual)");
    hipStream_t s = (hipStream_t)stream;
    static const int ct = getenv("T2V_FFN_CT") ? atoi(getenv("T2V_FFN_CT")) : 3;   // column tiles per wave (2: 128-token workgroups)
    switch (C) {
        case 320: return ct == 2 ? ffn_launch<320, 2>(x, ldx, M, w1p, b1p, w2p, b2, ln_eps, out, ldo, s)
                                 : ffn_launch<320, 3>(x, ldx, M, w1p, b1p, w2p, b2, ln_eps, out, ldo, s);
        case 64: return ct == 2 ? ffn_launch<64, 2>(x, ldx, M, w1p, b1p, w2p, b2, ln_eps, out, ldo, s)
                                : ffn_launch<64, 3>(x, ldx, M, w1p, b1p, w2p, b2, ln_eps, out, ldo, s);
        default: T2V_REQUIRE(false, T2V_ESHAPE, "t2v_ffn_fused: built for C = 320 (and 64, the test width)");
    }
    return T2V_OK;
}
