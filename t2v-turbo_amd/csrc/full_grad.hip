// Parameter gradients of the base weights for FULL fine-tuning of the student UNet (train_latent_t2v_turbo_v2.py:798-816 param
// groups, :1262 accelerator.backward: every UNet parameter is trainable there, not only LoRA tensors).  Two pieces that the
// token-contracted weight-gradient kernel (wgrad_tn.hip: dW = dy^T x) does not cover by itself:
//
//  * t2v_im2col_bf16 — the shifted-row matrix of a conv leaf, xcol[m][tap * C + c] = x[src(m, tap)][c] (zero outside the grid), in the
//    K order of the forward's tap-major weight pack, for every gather mode of t2v_gemm (3x3 stride 1 / stride 2 / stride 2 padded
//    right-bottom / over the nearest-x2 upsampled input, (3,1,1) temporal).  dW[n][tap][c] = sum_m dy[m][n] xcol[m][tap * C + c] is then
//    ONE t2v_wgrad_tn product (openaimodel3d.py:155-159,179-184 ResBlock convs, :63-79 Down/Upsample, :257-309 TemporalConvBlock under
//    autograd).  HBM-bound copy: 2 C bytes in, 2 taps C bytes out per output row.
//  * t2v_norm_affine_grad — the per-channel reductions over token rows that the normalisation layers' affine parameters, every bias
//    and the time-embedding row vector need:
//        dgamma[u][c] = sum_{rows of u} dz[m][c] xhat[m][c],   dbeta[u][c] = sum_{rows of u} dz[m][c]
//    with xhat from GroupNorm statistics ((mean, rstd) per (unit, group): lvdm/basics.py:78-89 GroupNormSpecific) or recomputed per row
//    (LayerNorm: attention.py:300-311), dz = dy, or dy * silu'(xhat gamma + beta) where the forward applied SiLU after the norm
//    (openaimodel3d.py:223-254 in_layers / out_layers), or plain column sums of dy (kind 2: biases; per-clip sums = d(loss)/d(emb)).
//    `sum_rows` consecutive rows make one output row u.  Two launches: per-block partial sums in a fixed order, then their sum in block
//    order — deterministic, no float atomics.  HBM-bound: reads x and dy once.
//  * t2v_repack_conv_f32 — the weights move every optimizer step, so the bf16 packs the recorded launch lists point at are re-made per step
//    (engine.Packer.refresh): a conv leaf's fp32 parameter [N][C][taps] into its tap-major forward pack [N][taps][C] or its data-gradient
//    pack [C][taps flipped][N], cast included, through an LDS tile so that both sides are coalesced (torch's strided cast-and-copy did
//    37 us per pack, ~ 300 GB/s: profiles/r06_full_finetune_kernel_stats.csv).
#include "common.h"

namespace {

constexpr int IM_KIND_S1 = T2V_GEMM_CONV3X3, IM_KIND_S2 = T2V_GEMM_CONV3X3_S2, IM_KIND_UP2 = T2V_GEMM_CONV3X3_UP2, IM_KIND_T = T2V_GEMM_TCONV3,
              IM_KIND_S2P = T2V_GEMM_CONV3X3_S2_PAD01;

// one thread per 16-byte chunk (8 channels) of one (output row, tap)
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ x0, int c0, int ld0, const bf16_t* __restrict__ x1, int c1, int ld1,
                                                     int mode, int n_img, int h, int w, int ho, int wo, int frames, int taps,
                                                     bf16_t* __restrict__ out, int ldo, long long total) {
    const int C = c0 + c1, cpr = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % cpr);
        const long long mt = i / cpr;
        const int tap = (int)(mt % taps);
        const long long m = mt / taps;
        long long src = -1;
        if (mode == IM_KIND_T) {
            const int hw = h * w;
            const long long img = m / hw;            // (clip, frame)
            const int f = (int)(img % frames), sf = f + tap - 1;
            if (sf >= 0 && sf < frames) src = m + (long long)(tap - 1) * hw;
        } else {
            const int xo = (int)(m % wo), yo = (int)((m / wo) % ho);
            const long long img = m / ((long long)wo * ho);
            const int ty = tap / 3, tx = tap - 3 * ty;
            int sy, sx;
            bool ok;
            if (mode == IM_KIND_S1) { sy = yo + ty - 1; sx = xo + tx - 1; ok = sy >= 0 && sy < h && sx >= 0 && sx < w; }
            else if (mode == IM_KIND_S2) { sy = 2 * yo + ty - 1; sx = 2 * xo + tx - 1; ok = sy >= 0 && sy < h && sx >= 0 && sx < w; }
            else if (mode == IM_KIND_S2P) { sy = 2 * yo + ty; sx = 2 * xo + tx; ok = sy < h && sx < w; }
            else { const int uy = yo + ty - 1, ux = xo + tx - 1; ok = uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w; sy = uy >> 1; sx = ux >> 1; }
            if (ok) src = (img * h + sy) * (long long)w + sx;
        }
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (src >= 0) {
            const int c = ci * 8;
            v = c < c0 ? *(const uint4*)(x0 + src * ld0 + c) : *(const uint4*)(x1 + src * ld1 + (c - c0));
        }
        *(uint4*)(out + m * ldo + (long long)tap * C + ci * 8) = v;
    }
}

// ---- per-channel row reductions ------------------------------------------------------------------------------------------------------
// Block = 4 waves; a wave owns whole rows (lane l holds the 16-byte chunks l, l + 64, ... of the row: NJ of them), so a LayerNorm's row
// statistics are two butterflies.  Block (bx, u) walks rows [bx * rows_per_blk, ...) of output row u; its four waves' register sums meet
// in LDS in wave order; partial[(u * nblk + bx)][2][C].
constexpr int AG_MAXJ = 5;   // C <= 64 * 8 * 5 = 2560 per launch (the widest GroupNorm: a 1280 + 1280 concat); plain column sums go out in column chunks
// NJ = chunks per lane (1, 2, 3, 5 by width), UR = rows a wave has in flight per trip (4 / 2 / 1): the first form (NJ = 5 for every width, one
// row per trip, the GroupNorm group of every element by an integer division per row) kept ~ 200 registers and 1.3 KB per wave in flight and
// ran at ~ 1 TB/s (profiles/r06_full_finetune_kernel_stats.csv: 55 us per GroupNorm launch).  Rows are still taken in the order
// r, r + 4, r + 8, ... per wave: the sums are bit-identical to the first form's.
template <int KIND, int NJ, int UR>   // KIND 0: GroupNorm statistics given, 1: LayerNorm (row statistics recomputed), 2: column sums of dy only
__global__ __launch_bounds__(256) void affine_grad_partial_kernel(const bf16_t* __restrict__ x0, int c0, int ld0, const bf16_t* __restrict__ x1,
                                                                  int c1, int ld1, long long sum_rows, int rows_per_blk, int rows_per_unit,
                                                                  int groups, const float* __restrict__ stats, float ln_eps,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                                  const bf16_t* __restrict__ dy, int ldy, float* __restrict__ partial) {
    const int C = c0 + c1, cpr = C >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.y, bx = blockIdx.x, nblk = gridDim.x;
    const long long r_begin = (long long)u * sum_rows + (long long)bx * rows_per_blk;
    const long long r_end = min((long long)(u + 1) * sum_rows, r_begin + rows_per_blk);
    float ag[NJ][8], ab[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; }
    const int cpg = KIND == 0 ? C / groups : 1;
    const float inv_c = 1.0f / (float)C;
    // GroupNorm: with >= 8 channels per group a chunk lies in at most two groups — the first one and where the second begins, once per lane
    const bool two_groups = KIND == 0 && cpg >= 8;
    int glo[NJ], eb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = (lane + 64 * j) * 8;
        glo[j] = KIND == 0 ? c / cpg : 0;
        eb[j] = KIND == 0 ? (glo[j] + 1) * cpg - c : 8;   // elements e >= eb are in group glo + 1
    }
    for (long long r0 = r_begin + wave; r0 < r_end; r0 += 4 * UR) {
        uint4 xr[UR][NJ], dr[UR][NJ];
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const long long r = r0 + 4 * k;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = lane + 64 * j;
                xr[k][j] = make_uint4(0u, 0u, 0u, 0u); dr[k][j] = make_uint4(0u, 0u, 0u, 0u);
                if (r < r_end && ci < cpr) {
                    dr[k][j] = *(const uint4*)(dy + r * ldy + ci * 8);
                    if (KIND != 2) {
                        const int c = ci * 8;
                        xr[k][j] = c < c0 ? *(const uint4*)(x0 + r * ld0 + c) : *(const uint4*)(x1 + r * ld1 + (c - c0));
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const long long r = r0 + 4 * k;
            if (r >= r_end) break;
            float xv[NJ][8], dv[NJ][8];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { unpack8(dr[k][j], dv[j]); unpack8(xr[k][j], xv[j]); }
            float mean = 0.f, rstd = 1.f;
            if (KIND == 1) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += xv[j][e];
                mean = wave_sum(s) * inv_c;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (lane + 64 * j < cpr) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float dlt = xv[j][e] - mean; q += dlt * dlt; }
                    }
                rstd = rsqrtf(wave_sum(q) * inv_c + ln_eps);
            }
            const long long unit = KIND == 0 ? r / rows_per_unit : 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = lane + 64 * j;
                if (ci < cpr) {
                    float gm[8], bt[8];   // (re-read per row from L1: kept in registers beside the sums they would not fit at 2560 channels)
                    if (KIND != 2 && silu) {
                        const float4 g0 = *(const float4*)(gamma + ci * 8), g1 = *(const float4*)(gamma + ci * 8 + 4);
                        const float4 b0 = *(const float4*)(beta + ci * 8), b1 = *(const float4*)(beta + ci * 8 + 4);
                        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
                        bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
                    }
                    float2 s_lo = make_float2(0.f, 1.f), s_hi = s_lo;
                    if (two_groups) {
                        const float* st = stats + (unit * groups + glo[j]) * 2;
                        s_lo = *(const float2*)st;
                        if (eb[j] < 8) s_hi = *(const float2*)(st + 2);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xh = 0.f, dz = dv[j][e];
                        if (KIND == 0) {
                            float2 st2;
                            if (two_groups) st2 = e < eb[j] ? s_lo : s_hi;
                            else st2 = *(const float2*)(stats + (unit * groups + (ci * 8 + e) / cpg) * 2);
                            xh = (xv[j][e] - st2.x) * st2.y;
                        } else if (KIND == 1) {
                            xh = (xv[j][e] - mean) * rstd;
                        }
                        if (KIND != 2 && silu) {   // y = silu(z), z = xhat gamma + beta: dz = dy * sigma(z) (1 + z (1 - sigma(z)))
                            const float z = xh * gm[e] + bt[e];
                            const float sg = 1.0f / (1.0f + __expf(-z));
                            dz *= sg * (1.0f + z * (1.0f - sg));
                        }
                        ag[j][e] += dz * xh;
                        ab[j][e] += dz;
                    }
                }
            }
        }
    }
    // the four waves' sums, added in wave order
    extern __shared__ float red[];   // [4][2][C]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = lane + 64 * j;
        if (ci < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * C + ci * 8 + e] = ag[j][e];
                red[(wave * 2 + 1) * C + ci * 8 + e] = ab[j][e];
            }
        }
    }
    __syncthreads();
    float* dst = partial + ((long long)u * nblk + bx) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = ((red[i] + red[2 * C + i]) + red[4 * C + i]) + red[6 * C + i];
}

// out[u][i] = sum over the unit's blocks.  Block (x, u) owns 64 consecutive values i (coalesced rows of the partial array); its 16
// waves take the blocks b = wave, wave + 16, ... (four loads in flight per trip), meet in LDS and are added in wave order: a fixed
// summation tree, deterministic.  (The first form walked all blocks with ONE thread per value: 116 us per call on ~1 000 blocks — a
// third of the full fine-tuning step, profiles/r06_full_finetune_kernel_stats.csv.)
constexpr int AGF_WAVES = 16;
__global__ __launch_bounds__(AGF_WAVES * 64) void affine_grad_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dgamma,
                                                                           int ld_g, float* __restrict__ dbeta, int ld_b) {
    __shared__ float red[AGF_WAVES][64];
    const int u = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool ok = i < 2 * C;
    const float* src = partial + (long long)u * nblk * 2 * C + (ok ? i : 0);
    const long long stride = 2LL * C;
    float s = 0.f;
    int b = wave;
    for (; b + 3 * AGF_WAVES < nblk; b += 4 * AGF_WAVES) {
        const float v0 = src[(long long)b * stride], v1 = src[(long long)(b + AGF_WAVES) * stride];
        const float v2 = src[(long long)(b + 2 * AGF_WAVES) * stride], v3 = src[(long long)(b + 3 * AGF_WAVES) * stride];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < nblk; b += AGF_WAVES) s += src[(long long)b * stride];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && ok) {
        float t = red[0][lane];
#pragma unroll
        for (int w = 1; w < AGF_WAVES; ++w) t += red[w][lane];
        if (i < C) { if (dgamma) dgamma[(long long)u * ld_g + i] = t; }
        else if (dbeta) dbeta[(long long)u * ld_b + (i - C)] = t;
    }
}

// ---- conv weight packs from the fp32 parameter ------------------------------------------------------------------------------------------
// kind 0: out[n][t][c] = w[n][c][t].  Block (c chunk of RP_C channels, n): reads RP_C * taps consecutive floats, writes taps runs of RP_C bf16.
// kind 1: out[c][taps - 1 - t][n] = w[n][c][t] (the conv over dy that gives dx: channels and filters swapped, taps mirrored).  Block
//         (c chunk of RP_DC channels, n chunk of RP_DN filters): per filter RP_DC * taps consecutive floats in, per (c, t) a run of RP_DN bf16 out.
constexpr int RP_C = 256, RP_DC = 16, RP_DN = 64, RP_MAXT = 9;
__global__ __launch_bounds__(256) void repack_conv_fwd_kernel(const float* __restrict__ w, int C, int taps, bf16_t* __restrict__ out, int ldo) {
    __shared__ float tile[RP_C * RP_MAXT];
    const int n = blockIdx.y, c0 = blockIdx.x * RP_C, nc = min(RP_C, C - c0);
    const float* src = w + ((long long)n * C + c0) * taps;
    for (int i = threadIdx.x; i < nc * taps; i += 256) tile[i] = src[i];
    __syncthreads();
    bf16_t* dst = out + (long long)n * ldo + c0;
    for (int i = threadIdx.x; i < nc * taps; i += 256) {
        const int t = i / nc, c = i - t * nc;
        dst[(long long)t * C + c] = f2bf(tile[c * taps + t]);
    }
}
__global__ __launch_bounds__(256) void repack_conv_dgrad_kernel(const float* __restrict__ w, int N, int C, int taps, bf16_t* __restrict__ out, int ldo) {
    __shared__ float tile[RP_DN][RP_DC * RP_MAXT + 1];
    const int c0 = blockIdx.x * RP_DC, n0 = blockIdx.y * RP_DN, nc = min(RP_DC, C - c0), nn = min(RP_DN, N - n0);
    const int run = nc * taps;
    for (int i = threadIdx.x; i < nn * run; i += 256) {
        const int j = i / run, k = i - j * run;
        tile[j][k] = w[((long long)(n0 + j) * C + c0) * taps + k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < run * nn; i += 256) {
        const int k = i / nn, j = i - k * nn;          // k = c * taps + t of the source
        const int c = k / taps, t = k - c * taps;
        out[(long long)(c0 + c) * ldo + (long long)(taps - 1 - t) * N + n0 + j] = f2bf(tile[j][k]);
    }
}

}  // namespace

extern "C" long long t2v_im2col_rows(int mode, int n_img, int h, int w) {
    if (mode == T2V_GEMM_CONV3X3 || mode == T2V_GEMM_TCONV3) return (long long)n_img * h * w;
    if (mode == T2V_GEMM_CONV3X3_S2) return (long long)n_img * ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1);
    if (mode == T2V_GEMM_CONV3X3_S2_PAD01) return (long long)n_img * ((h - 2) / 2 + 1) * ((w - 2) / 2 + 1);
    if (mode == T2V_GEMM_CONV3X3_UP2) return (long long)n_img * 4 * h * w;
    return -1;
}

extern "C" int t2v_im2col_bf16(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int mode, int n_img, int h, int w, int frames,
                               void* out, int ldo, void* stream) {
    T2V_REQUIRE(x0 && out && n_img > 0 && h > 0 && w > 0 && c0 > 0 && c1 >= 0 && (c1 == 0 || x1), T2V_EINVAL, "t2v_im2col_bf16: bad argument");
    const long long rows = t2v_im2col_rows(mode, n_img, h, w);
    T2V_REQUIRE(rows > 0, T2V_EINVAL, "t2v_im2col_bf16: mode must be one of the conv gather modes of t2v_gemm");
    const int taps = mode == T2V_GEMM_TCONV3 ? 3 : 9, C = c0 + c1;
    T2V_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && ld0 % 8 == 0 && (c1 == 0 || ld1 % 8 == 0) && ldo % 8 == 0 && ldo >= taps * C &&
                    (uintptr_t)x0 % 16 == 0 && (uintptr_t)x1 % 16 == 0 && (uintptr_t)out % 16 == 0,
                T2V_ESHAPE, "t2v_im2col_bf16: channels / row strides in multiples of 8, 16-byte aligned rows, ldo >= taps * C");
    T2V_REQUIRE(mode != T2V_GEMM_TCONV3 || (frames > 0 && n_img % frames == 0), T2V_ESHAPE, "t2v_im2col_bf16: n_img must be clips x frames");
    int ho = h, wo = w;
    if (mode == T2V_GEMM_CONV3X3_S2) { ho = (h - 1) / 2 + 1; wo = (w - 1) / 2 + 1; }
    else if (mode == T2V_GEMM_CONV3X3_S2_PAD01) { ho = (h - 2) / 2 + 1; wo = (w - 2) / 2 + 1; }
    else if (mode == T2V_GEMM_CONV3X3_UP2) { ho = 2 * h; wo = 2 * w; }
    const long long total = rows * taps * (C / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256LL * 16) blocks = 256LL * 16;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1,
                       ld1, mode, n_img, h, w, ho, wo, frames, taps, (bf16_t*)out, ldo, total);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

static int affine_grad_blocks(long long sum_rows, long long n_out) {
    // about 1024 blocks over the whole launch, at least 16 rows (four per wave) each
    long long nblk = (1024 + n_out - 1) / n_out;
    const long long max_blk = (sum_rows + 15) / 16;
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

extern "C" long long t2v_norm_affine_grad_ws_floats(long long rows, long long sum_rows, int channels) {
    if (rows <= 0 || sum_rows <= 0 || rows % sum_rows || channels <= 0) return -1;
    const long long n_out = rows / sum_rows;
    return n_out * affine_grad_blocks(sum_rows, n_out) * 2 * channels;
}

extern "C" int t2v_norm_affine_grad(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, long long rows, long long sum_rows, int kind,
                                    int rows_per_unit, int groups, const float* stats, float ln_eps, const float* gamma, const float* beta,
                                    int silu, const void* dy, int ldy, float* dgamma, int ld_dgamma, float* dbeta, int ld_dbeta, float* ws,
                                    void* stream) {
    T2V_REQUIRE(dy && ws && rows > 0 && sum_rows > 0 && rows % sum_rows == 0 && kind >= 0 && kind <= 2 && (dgamma || dbeta), T2V_EINVAL,
                "t2v_norm_affine_grad: bad argument");
    if (kind == 2) {
        c1 = 0; x1 = nullptr; silu = 0;
        constexpr int CHUNK = 2048;
        if (c0 > CHUNK) {   // plain column sums are independent per column: wider matrices (the 8 C-wide GEGLU pre-activation) go out in chunks
            for (int c = 0; c < c0; c += CHUNK) {
                const int n = c0 - c < CHUNK ? c0 - c : CHUNK;
                const int rc = t2v_norm_affine_grad(nullptr, n, 0, nullptr, 0, 0, rows, sum_rows, 2, 0, 0, nullptr, 0.f, nullptr, nullptr, 0,
                                                    (const bf16_t*)dy + c, ldy, dgamma ? dgamma + c : nullptr, ld_dgamma, dbeta ? dbeta + c : nullptr,
                                                    ld_dbeta, ws, stream);
                if (rc != T2V_OK) return rc;
            }
            return T2V_OK;
        }
    }
    const int C = c0 + c1;
    T2V_REQUIRE(C > 0 && C % 8 == 0 && C <= 64 * 8 * AG_MAXJ && c0 % 8 == 0 && ldy % 8 == 0 && (uintptr_t)dy % 16 == 0, T2V_ESHAPE,
                "t2v_norm_affine_grad: channels a multiple of 8 (<= 2560 behind a norm), 16-byte aligned rows");
    if (kind != 2 && silu)
        T2V_REQUIRE(((uintptr_t)gamma | (uintptr_t)beta) % 16 == 0, T2V_ESHAPE, "t2v_norm_affine_grad: gamma / beta 16-byte aligned");
    if (kind != 2) {
        T2V_REQUIRE(x0 && ld0 % 8 == 0 && (uintptr_t)x0 % 16 == 0 && (c1 == 0 || (x1 && ld1 % 8 == 0 && (uintptr_t)x1 % 16 == 0)), T2V_ESHAPE,
                    "t2v_norm_affine_grad: the normalised tensor's rows must be 16-byte aligned");
        T2V_REQUIRE(!silu || (gamma && beta), T2V_EINVAL, "t2v_norm_affine_grad: SiLU behind the norm needs gamma and beta");
        T2V_REQUIRE(dgamma, T2V_EINVAL, "t2v_norm_affine_grad: dgamma");
    }
    if (kind == 0)
        T2V_REQUIRE(stats && groups > 0 && C % groups == 0 && rows_per_unit > 0 && rows % rows_per_unit == 0, T2V_ESHAPE,
                    "t2v_norm_affine_grad: GroupNorm statistics [rows / rows_per_unit][groups][2]");
    const long long n_out = rows / sum_rows;
    T2V_REQUIRE(n_out <= 65535, T2V_ESHAPE, "t2v_norm_affine_grad: grid");
    const int nblk = affine_grad_blocks(sum_rows, n_out);
    const int rows_per_blk = (int)((sum_rows + nblk - 1) / nblk);
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)8 * C * sizeof(float);
    dim3 grid(nblk, (unsigned)n_out);
#define T2V_AG_LAUNCH(K, NJ, UR)                                                                                                              \
    do {                                                                                                                                      \
        static bool attr_set = false;   /* (8 C floats of LDS: 80 KB at 2560 channels, above the 64 KB a launch gets unasked) */             \
        if (!attr_set) { hipFuncSetAttribute((const void*)affine_grad_partial_kernel<K, NJ, UR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set = true; } \
        hipLaunchKernelGGL((affine_grad_partial_kernel<K, NJ, UR>), grid, dim3(256), smem, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, sum_rows, \
                           rows_per_blk, rows_per_unit, groups, stats, ln_eps, gamma, beta, silu, (const bf16_t*)dy, ldy, ws);               \
    } while (0)
#define T2V_AG_BY_WIDTH(K)                                                                                                                    \
    do {                                                                                                                                      \
        if (C <= 512) T2V_AG_LAUNCH(K, 1, 4);                                                                                                 \
        else if (C <= 1024) T2V_AG_LAUNCH(K, 2, 2);                                                                                           \
        else if (C <= 1536) T2V_AG_LAUNCH(K, 3, 1);                                                                                           \
        else T2V_AG_LAUNCH(K, 5, 1);                                                                                                          \
    } while (0)
    if (kind == 0) T2V_AG_BY_WIDTH(0);
    else if (kind == 1) T2V_AG_BY_WIDTH(1);
    else T2V_AG_BY_WIDTH(2);
#undef T2V_AG_BY_WIDTH
#undef T2V_AG_LAUNCH
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(affine_grad_final_kernel, dim3((2 * C + 63) / 64, (unsigned)n_out), dim3(AGF_WAVES * 64), 0, s, (const float*)ws, nblk, C,
                       dgamma, ld_dgamma, dbeta, ld_dbeta);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_repack_conv_f32(const float* w, int N, int C, int taps, int kind, void* out, int ldo, void* stream) {
    T2V_REQUIRE(w && out && N > 0 && C > 0 && taps >= 1 && taps <= RP_MAXT && (kind == 0 || kind == 1), T2V_EINVAL,
                "t2v_repack_conv_f32: bad argument (taps <= 9, kind 0 = forward pack, 1 = data-gradient pack)");
    T2V_REQUIRE(ldo >= taps * (kind == 0 ? C : N), T2V_ESHAPE, "t2v_repack_conv_f32: ldo smaller than a pack row");
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) {
        T2V_REQUIRE(N <= 65535, T2V_ESHAPE, "t2v_repack_conv_f32: grid");
        hipLaunchKernelGGL(repack_conv_fwd_kernel, dim3((C + RP_C - 1) / RP_C, N), dim3(256), 0, s, w, C, taps, (bf16_t*)out, ldo);
    } else {
        T2V_REQUIRE((N + RP_DN - 1) / RP_DN <= 65535, T2V_ESHAPE, "t2v_repack_conv_f32: grid");
        hipLaunchKernelGGL(repack_conv_dgrad_kernel, dim3((C + RP_DC - 1) / RP_DC, (N + RP_DN - 1) / RP_DN), dim3(256), 0, s, w, N, C, taps,
                           (bf16_t*)out, ldo);
    }
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
