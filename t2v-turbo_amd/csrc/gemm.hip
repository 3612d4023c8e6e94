// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   out[M,N] = epilogue(alpha * gather(A)[M,K] x W[N,K]^T)
//
// Design (CDNA4):
//  * 4 wavefronts (64 lanes) per workgroup, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//    MFMA-M = tokens (A operand = activations), MFMA-N = output channels (B operand = weights).
//  * K is walked in 64-element steps.  For the conv modes a K step is one 64-channel slab of one
//    filter tap: the im2col matrix is never built, each lane aims its 16-byte LDS-DMA
//    (global_load_lds_dwordx4) at the shifted source row, or at a zero page for padding taps.
//    Nearest-x2 upsampling, stride 2, the (3,1,1) temporal taps and the skip-connection channel
//    concat are all just address arithmetic in the same gather.
//  * LDS tiles are [rows][64] bf16 (128 B rows) with the 16-byte chunk index XOR-swizzled by
//    (row>>1)&7, applied on the DMA *source* side (the LDS image of a DMA is lane-linear) and on
//    the ds_read_b128 fragment reads, which makes those reads bank-conflict free.
//  * double-buffered tiles: the DMA for step k+1 is issued right after the barrier that publishes
//    step k and overlaps its MFMAs.
//  * epilogue: accumulators are staged through (the now free) LDS so that every global store /
//    residual load is a full 16-byte-per-lane, 128-byte-per-row coalesced access; bias,
//    time-embedding row vector, residual add, SiLU and GEGLU are fused here.
//  * workgroup id -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a contiguous
//    run of tiles, N-tile fastest, so the tiles that share an A panel hit the same L2.
#include "common.h"

struct GemmParams {
    t2v_gemm_desc d;
    const void* zero;
    int K, nk, taps, nsrc;
    // every mode is a (kh x kw) conv over an (n, H, W) grid: LINEAR = 1x1 over (M,1,1);
    // TCONV3 = 3x1 over (clip, frame, pixel); UP2 reads the grid through a >>1 (ups = 1)
    int gh, gw;        // source grid
    int kh, kw, stride, pad_y, pad_x, ups;
    int h_out, w_out;
    int tiles_m, tiles_n;
    int vec_store;
};

namespace {

template <int BM, int BN>
struct Smem {
    static constexpr int A_BYTES = BM * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int TILE_BYTES = 2 * STAGE;
    static constexpr int EPI_BYTES = BM * BN * 4;
    static constexpr int BYTES = TILE_BYTES > EPI_BYTES ? TILE_BYTES : EPI_BYTES;
};

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;  // DMA wave-instructions per wave per stage
    using S = Smem<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const t2v_gemm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WN, wave_n = wave % WN;

    // ---- XCD-aware tile assignment (bijective form) -------------------------------------------
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- batch offsets --------------------------------------------------------------------------
    const int z = blockIdx.y;
    const int z0 = z / d.batch_inner, z1 = z % d.batch_inner;
    const bf16_t* a0 = (const bf16_t*)d.a0 + z0 * d.a_stride0 + z1 * d.a_stride1;
    const bf16_t* a1 = d.a1 ? (const bf16_t*)d.a1 + z0 * d.a_stride0 + z1 * d.a_stride1 : nullptr;
    const bf16_t* wbase = (const bf16_t*)d.w + z0 * d.w_stride0 + z1 * d.w_stride1;
    const long long o_off = z0 * d.o_stride0 + z1 * d.o_stride1;
    const bf16_t* zero = (const bf16_t*)p.zero;

    // ---- per-lane DMA row bookkeeping -----------------------------------------------------------
    // wave-instruction i (= wave + 4*j) fills tile rows [8i, 8i+8): lane -> row 8i + (lane>>3),
    // 16-byte slot (lane&7) of that row, which holds source chunk (lane&7) ^ swz(row).
    const int H = p.gh, W = p.gw;
    int a_n[A_IT], a_y[A_IT], a_x[A_IT];  // decomposed output coordinates (a_n < 0: row >= M)
    int a_chunk[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int r = (wave + 4 * j) * 8 + (lane >> 3);
        a_chunk[j] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        const int m = m0 + r;
        if (m >= d.M) {
            a_n[j] = -1; a_y[j] = 0; a_x[j] = 0;
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
        const int r = (wave + 4 * j) * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        const int n = n0 + r;
        if (n < d.N) { wptr[j] = wbase + (long long)n * d.ldw + chunk; winc[j] = 64; }
        else { wptr[j] = zero; winc[j] = 0; }
    }

    int seg = 0, seg_left = 0;  // staging iterator: segment = (tap, source)
    auto begin_segment = [&]() {
        const int tap = seg / p.nsrc, src = seg - tap * p.nsrc;
        const bf16_t* base = src ? a1 : a0;
        const int ld = src ? d.lda1 : d.lda0;
        seg_left = (src ? d.c1 : d.c0) >> 6;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int uy = a_y[j] * p.stride + ky - p.pad_y, ux = a_x[j] * p.stride + kx - p.pad_x;
            const bool ok = a_n[j] >= 0 && uy >= 0 && uy < (H << p.ups) && ux >= 0 && ux < (W << p.ups);
            const long long row = ((long long)a_n[j] * H + (uy >> p.ups)) * W + (ux >> p.ups);
            if (ok) { aptr[j] = base + row * ld + a_chunk[j]; ainc[j] = 64; }
            else { aptr[j] = zero; ainc[j] = 0; }
        }
    };
    auto stage = [&](int buf) {
        char* sa = smem + buf * S::STAGE;
        char* sb = sa + S::A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) dma16(aptr[j], sa + (wave + 4 * j) * 1024);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) dma16(wptr[j], sb + (wave + 4 * j) * 1024);
        // advance to the next K step
#pragma unroll
        for (int j = 0; j < A_IT; ++j) aptr[j] += ainc[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) wptr[j] += winc[j];
        if (--seg_left == 0) { ++seg; if (seg < p.taps * p.nsrc) begin_segment(); }
    };

    // ---- fragment read addressing ---------------------------------------------------------------
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    begin_segment();
    stage(0);
    for (int kt = 0; kt < p.nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int buf = kt & 1;
        if (kt + 1 < p.nk) stage(buf ^ 1);
        const char* sa = smem + buf * S::STAGE + (wave_m * WTM + frow) * 128;
        const char* sb = smem + buf * S::STAGE + S::A_BYTES + (wave_n * WTN + frow) * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int coff = ((kk * 2 + hi) ^ swz) << 4;
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(sa + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(sb + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: stage through LDS, then coalesced 16-byte rows --------------------------------
    __syncthreads();
    float* st = (float*)smem + wave * (WTM * WTN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                st[row * WTN + j * 32 + frow] = acc[i][j][r];
            }
    __syncthreads();

    const bool geglu = d.act == T2V_ACT_GEGLU;
    const int lpr = geglu ? 4 : WTN / 8;  // lanes per row
    const int rpp = 64 / lpr;             // rows per pass
    const int c8 = (lane % lpr) * 8;
    const int gn_in = n0 + wave_n * WTN + c8;                     // column in W-row space (bias index)
    const int gn = geglu ? (n0 + wave_n * WTN) / 2 + c8 : gn_in;  // output column
    const int n_out = geglu ? d.N / 2 : d.N;
    float bias[8], bias2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = (d.bias && gn_in + e < d.N) ? d.bias[gn_in + e] : 0.f;
        bias2[e] = (geglu && d.bias && gn_in + 32 + e < d.N) ? d.bias[gn_in + 32 + e] : 0.f;
    }
    for (int row = lane / lpr; row < WTM; row += rpp) {
        const int gm = m0 + wave_m * WTM + row;
        if (gm >= d.M || gn >= n_out) continue;
        float v[8];
        const float4 lo = *(const float4*)(st + row * WTN + c8);
        const float4 hi4 = *(const float4*)(st + row * WTN + c8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
        v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + bias[e];
        if (geglu) {
            if constexpr (WTN == 64) {
                const float4 g0 = *(const float4*)(st + row * WTN + 32 + c8);
                const float4 g1 = *(const float4*)(st + row * WTN + 32 + c8 + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * gelu_f(g[e] * d.alpha + bias2[e]);
            }
        }
        if (d.rowvec) {
            const float* rv = d.rowvec + (long long)(gm / d.rowvec_div) * d.ld_rowvec + gn;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (gn + e < n_out) v[e] += rv[e];
        }
        const bool full = p.vec_store && gn + 8 <= n_out;
        if (d.residual) {
            const bf16_t* rp = (const bf16_t*)d.residual + o_off + (long long)gm * d.ldr + gn;
            if (full) {
                float rf[8];
                unpack8(*(const uint4*)rp, rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rf[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gn + e < n_out) v[e] += bf2f(rp[e]);
            }
        }
        if (d.act == T2V_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        if (d.out_f32) {
            float* op = (float*)d.out + o_off + (long long)gm * d.ldo + gn;
            if (full) {
                *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gn + e < n_out) op[e] = v[e];
            }
        } else {
            bf16_t* op = (bf16_t*)d.out + o_off + (long long)gm * d.ldo + gn;
            if (full) {
                *(uint4*)op = pack8(v);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gn + e < n_out) op[e] = f2bf(v[e]);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(GemmParams& p, hipStream_t s) {
    p.tiles_m = (p.d.M + BM - 1) / BM;
    p.tiles_n = (p.d.N + BN - 1) / BN;
    dim3 grid(p.tiles_m * p.tiles_n, p.d.batch, 1);
    constexpr int smem = Smem<BM, BN>::BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), grid, dim3(256), smem, s, p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

}  // namespace

// tile-shape override for tuning / tests: 0 = heuristic, 1 = 128x128, 2 = 128x64, 3 = 256x64
static int g_force_cfg = 0;
extern "C" int t2v_gemm_force_config(int cfg) { g_force_cfg = cfg; return T2V_OK; }

extern "C" int t2v_gemm(const t2v_gemm_desc* dd, void* stream) {
    T2V_REQUIRE(dd && dd->a0 && dd->w && dd->out, T2V_EINVAL, "t2v_gemm: null pointer");
    GemmParams p;
    p.d = *dd;
    t2v_gemm_desc& d = p.d;
    if (!d.a1) { d.c1 = 0; d.lda1 = 0; }
    if (d.batch <= 0) { d.batch = 1; }
    if (d.batch_inner <= 0) d.batch_inner = 1;
    T2V_REQUIRE(d.M > 0 && d.N > 0, T2V_EINVAL, "t2v_gemm: empty problem");
    T2V_REQUIRE(d.c0 > 0 && d.c0 % 64 == 0 && d.c1 % 64 == 0, T2V_ESHAPE, "t2v_gemm: channels must be multiples of 64");
    T2V_REQUIRE(d.lda0 % 8 == 0 && d.lda1 % 8 == 0 && d.ldw % 8 == 0, T2V_ESHAPE, "t2v_gemm: lda/ldw must be multiples of 8");
    T2V_REQUIRE(((uintptr_t)d.a0 % 16 == 0) && ((uintptr_t)d.w % 16 == 0) && (!d.a1 || (uintptr_t)d.a1 % 16 == 0), T2V_ESHAPE,
                "t2v_gemm: operands must be 16-byte aligned");
    T2V_REQUIRE(d.a_stride0 % 8 == 0 && d.a_stride1 % 8 == 0 && d.w_stride0 % 8 == 0 && d.w_stride1 % 8 == 0, T2V_ESHAPE,
                "t2v_gemm: batch strides must be multiples of 8");
    T2V_REQUIRE(d.batch <= 65535, T2V_ESHAPE, "t2v_gemm: batch too large");
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
    p.nk = p.K / 64;
    T2V_REQUIRE(d.ldw >= p.K, T2V_EINVAL, "t2v_gemm: ldw < K");
    if (d.act == T2V_ACT_GEGLU) T2V_REQUIRE(d.N % 128 == 0, T2V_ESHAPE, "t2v_gemm: GEGLU needs N % 128 == 0");
    if (d.rowvec) T2V_REQUIRE(d.rowvec_div > 0, T2V_EINVAL, "t2v_gemm: rowvec_div");
    p.zero = t2v_zero_page();
    T2V_REQUIRE(p.zero, T2V_EHIP, "t2v_gemm: zero page allocation failed");
    const int n_out = d.act == T2V_ACT_GEGLU ? d.N / 2 : d.N;
    const int esz = d.out_f32 ? 4 : 2;
    p.vec_store = (d.ldo % 8 == 0) && (n_out % 8 == 0) && ((uintptr_t)d.out % 16 == 0) && (d.o_stride0 % 8 == 0) &&
                  (d.o_stride1 % 8 == 0) && (!d.residual || (d.ldr % 8 == 0 && (uintptr_t)d.residual % 16 == 0));
    (void)esz;
    hipStream_t s = (hipStream_t)stream;
    int cfg = g_force_cfg;
    if (d.act == T2V_ACT_GEGLU) cfg = 1;
    if (cfg == 0) {
        if (d.N % 128 == 0) cfg = 1;
        else if (d.M >= 8192 && d.N % 64 == 0) cfg = 3;
        else cfg = 2;
    }
    switch (cfg) {
        case 1: return launch<128, 128, 2, 2>(p, s);
        case 3: return launch<256, 64, 4, 1>(p, s);
        default: return launch<128, 64, 2, 2>(p, s);
    }
}
