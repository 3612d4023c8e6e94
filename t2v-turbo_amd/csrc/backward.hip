// Backward (dX) kernels for the KL-VAE decoder's reward-gradient branch (SURVEY.md §8(f) rank 2;
// reference call site train_t2v_turbo_v1_lora.py:1047-1098: d reward / d latents through vae.decode).
// The convolution / linear data gradients are plain t2v_gemm launches on re-packed weights (flipped taps,
// swapped channel roles); this file holds what is not a GEMM: GroupNorm(+SiLU) backward, softmax backward,
// the bf16 transposes the attention gradients need (our GEMM wants both operands K-contiguous), and the 2x2
// sum-pool that is the adjoint of the nearest-x2 upsampling folded into the forward conv gather.
#include "common.h"
#include "gn_bwd_common.h"

namespace {

struct GbGeom { int cpr, tx, ty; };
__host__ __device__ inline GbGeom gb_geom(int C) {
    GbGeom g;
    g.cpr = C / 8;
    g.tx = g.cpr;            // C <= 2048: one 16-byte chunk per thread
    g.ty = 256 / g.tx;
    if (g.ty < 1) g.ty = 1;
    return g;
}
inline int gb_slab_rows(int C, int rows_per_unit) {
    const GbGeom g = gb_geom(C);
    int s = g.ty * GB_RPT * 2;
    const int need = (rows_per_unit + 1023) / 1024;
    if (s < need) s = (need + g.ty - 1) / g.ty * g.ty;
    return s;
}

// partial[unit][slab][2*group + {0: sum g, 1: sum g*xhat}] (fixed order: deterministic)
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const bf16_t* x, int ldx, const bf16_t* dy, int ldy, int C,
                                                             int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                             const float* gamma, const float* beta, int silu, float* partial) {
    extern __shared__ float sred[];  // [2][ty][C]
    const int cpg = C / groups;
    const GbGeom g = gb_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x, cx = tid % g.tx, ry = tid / g.tx;
    const int r0 = slab * slab_rows, r1 = min(r0 + slab_rows, rows_per_unit);
    float* s1 = sred;
    float* s2 = sred + g.ty * C;
    if (ry < g.ty && cx < g.cpr) {
        GbChan k;
        gb_load_chan(k, stats, gamma, beta, unit, groups, cpg, cx * 8);
        float a1[8], a2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
        for (int r = r0 + ry; r < r1; r += GB_RPT * g.ty) {
            uint4 ux[GB_RPT], ud[GB_RPT];
#pragma unroll
            for (int t = 0; t < GB_RPT; ++t) {
                const int rr = r + t * g.ty;
                const long long row = (long long)unit * rows_per_unit + rr;
                ux[t] = rr < r1 ? *(const uint4*)(x + row * ldx + cx * 8) : make_uint4(0, 0, 0, 0);
                ud[t] = rr < r1 ? *(const uint4*)(dy + row * ldy + cx * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < GB_RPT; ++t) {
                if (r + t * g.ty >= r1) continue;
                float fx[8], fd[8];
                unpack8(ux[t], fx);
                unpack8(ud[t], fd);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xh, gg;
                    gb_elem(k, e, fx[e], fd[e], silu, xh, gg);
                    a1[e] += gg;
                    a2[e] += gg * xh;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[ry * C + cx * 8 + e] = a1[e]; s2[ry * C + cx * 8 + e] = a2[e]; }
    }
    __syncthreads();
    for (int i = tid; i < groups * 2; i += 256) {
        const int grp = i >> 1;
        const float* src = ((i & 1) ? s2 : s1) + grp * cpg;
        float b0 = 0.f, b1 = 0.f;
        for (int y = 0; y < g.ty; ++y) {
            const float* row = src + y * C;
            int c = 0;
            for (; c + 2 <= cpg; c += 2) { b0 += row[c]; b1 += row[c + 1]; }
            for (; c < cpg; ++c) b0 += row[c];
        }
        partial[((long long)unit * nslab + slab) * groups * 2 + i] = b0 + b1;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) (+ resid)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* x, int ldx, const bf16_t* dy, int ldy, int C,
                                                           int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                           const float* gamma, const float* beta, int silu, const float* bstats,
                                                           const bf16_t* resid, int ldr, bf16_t* dx, int ldo) {
    const int cpg = C / groups;
    const GbGeom g = gb_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x;
    const int tid = threadIdx.x, cx = tid % g.tx, ry = tid / g.tx;
    if (ry >= g.ty || cx >= g.cpr) return;
    GbChan k;
    gb_load_chan(k, stats, gamma, beta, unit, groups, cpg, cx * 8);
    float m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int grp = (cx * 8 + e) / cpg;
        m1[e] = bstats[((long long)unit * groups + grp) * 2];
        m2[e] = bstats[((long long)unit * groups + grp) * 2 + 1];
    }
    const int r0 = slab * slab_rows, r1 = min(r0 + slab_rows, rows_per_unit);
    for (int r = r0 + ry; r < r1; r += GB_RPT * g.ty) {
        uint4 ux[GB_RPT], ud[GB_RPT], ur[GB_RPT];
#pragma unroll
        for (int t = 0; t < GB_RPT; ++t) {
            const int rr = r + t * g.ty;
            const long long row = (long long)unit * rows_per_unit + rr;
            if (rr < r1) {
                ux[t] = *(const uint4*)(x + row * ldx + cx * 8);
                ud[t] = *(const uint4*)(dy + row * ldy + cx * 8);
                ur[t] = resid ? *(const uint4*)(resid + row * ldr + cx * 8) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < GB_RPT; ++t) {
            const int rr = r + t * g.ty;
            if (rr >= r1) continue;
            float fx[8], fd[8], fr[8], o[8];
            unpack8(ux[t], fx);
            unpack8(ud[t], fd);
            unpack8(ur[t], fr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xh, gg;
                gb_elem(k, e, fx[e], fd[e], silu, xh, gg);
                o[e] = k.rs[e] * (gg - m1[e] - xh * m2[e]) + fr[e];
            }
            *(uint4*)(dx + ((long long)unit * rows_per_unit + rr) * ldo + cx * 8) = pack8(o);
        }
    }
}

// ds = p * (dp - sum_j p_j dp_j), in place on dp; one wave per row, row in registers (n_pad <= 4096)
constexpr int SB_MAX = 8;
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const bf16_t* p, bf16_t* dp, long long rows, int n, int n_pad, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* pp = p + row * ld;
    bf16_t* dd = dp + row * ld;
    const int cpr = n_pad / 8;
    float pv[SB_MAX][8], dv[SB_MAX][8];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < SB_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            unpack8(*(const uint4*)(pp + ci * 8), pv[j]);
            unpack8(*(const uint4*)(dd + ci * 8), dv[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ci * 8 + e < n) dot += pv[j][e] * dv[j][e];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < SB_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ci * 8 + e < n ? pv[j][e] * (dv[j][e] - dot) : 0.f;
            *(uint4*)(dd + ci * 8) = pack8(o);
        }
    }
}

// out[b][c][r] = in[b][r][c] (bf16), 64x64 tiles through LDS; columns of `out` beyond `rows` up to ld_out are left alone
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, int ld_in, int rows, int cols, bf16_t* out, int ld_out,
                                                        long long in_stride, long long out_stride) {
    __shared__ bf16_t tile[64][66];
    const bf16_t* ib = in + blockIdx.z * in_stride;
    bf16_t* ob = out + blockIdx.z * out_stride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? ib[(long long)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) ob[(long long)c * ld_out + r] = tile[tx][i];
    }
}

// adjoint of nearest-x2 upsampling: out[n][y][x][:] = sum of the 2x2 block of in[n][2y..][2x..][:]
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const bf16_t* in, int n_img, int h, int w, int C, bf16_t* out) {
    const int cpr = C / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n_img * h * w * cpr;
    if (idx >= total) return;
    const int ci = (int)(idx % cpr);
    const long long pix = idx / cpr;
    const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
    const long long n = pix / ((long long)w * h);
    const bf16_t* base = in + ((n * 2 * h + 2 * yy) * 2 * w + 2 * xx) * C + ci * 8;
    const uint4 a = *(const uint4*)base, b = *(const uint4*)(base + C);
    const uint4 c = *(const uint4*)(base + (long long)2 * w * C), d = *(const uint4*)(base + (long long)2 * w * C + C);
    float fa[8], fb[8], fc[8], fd[8], o[8];
    unpack8(a, fa); unpack8(b, fb); unpack8(c, fc); unpack8(d, fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (fa[e] + fb[e]) + (fc[e] + fd[e]);
    *(uint4*)(out + pix * C + ci * 8) = pack8(o);
}

}  // namespace

extern "C" long long t2v_gn_bwd_ws_floats(int n_units, int rows_per_unit, int groups) {
    long long nslab = (rows_per_unit + GB_RPT * 2 - 1) / (GB_RPT * 2);
    if (nslab > 1024) nslab = 1024;
    return (long long)n_units * nslab * groups * 2 + (long long)n_units * groups * 2;
}

extern "C" int t2v_gn_bwd(const void* x, int ldx, int C, int n_units, int rows_per_unit, int groups, const float* stats,
                          const float* gamma, const float* beta, int silu, const void* dy, int ldy, const void* resid, int ldr,
                          float* ws, void* dx, int ldo, void* stream) {
    T2V_REQUIRE(x && dy && stats && gamma && beta && ws && dx, T2V_EINVAL, "t2v_gn_bwd: null pointer");
    T2V_REQUIRE(n_units > 0 && rows_per_unit > 0 && groups > 0 && groups <= 128, T2V_EINVAL, "t2v_gn_bwd: bad size");
    T2V_REQUIRE(C % 8 == 0 && C <= 2048 && C % groups == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && (!resid || ldr % 8 == 0),
                T2V_ESHAPE, "t2v_gn_bwd: channels <= 2048, multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const GbGeom gg = gb_geom(C);
    const int slab_rows = gb_slab_rows(C, rows_per_unit);
    const int nslab = (rows_per_unit + slab_rows - 1) / slab_rows;
    float* partial = ws;
    float* bstats = ws + (long long)n_units * nslab * groups * 2;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nslab, n_units), dim3(256), (size_t)2 * gg.ty * C * sizeof(float), s,
                       (const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, C, rows_per_unit, groups, slab_rows, stats, gamma, beta, silu, partial);
    T2V_CHECK_LAUNCH();
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(n_units), dim3(GB_FINAL_THREADS), 0, s, (const float*)partial, nslab, groups, inv_count, bstats);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, C,
                       rows_per_unit, groups, slab_rows, stats, gamma, beta, silu, (const float*)bstats, (const bf16_t*)resid, ldr,
                       (bf16_t*)dx, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_softmax_bwd_rows(const void* p, void* dp, long long rows, int n, int n_pad, int ld, void* stream) {
    T2V_REQUIRE(p && dp && rows > 0 && n > 0 && n_pad >= n, T2V_EINVAL, "t2v_softmax_bwd_rows: bad argument");
    T2V_REQUIRE(n_pad % 8 == 0 && ld % 8 == 0 && n_pad <= ld && n_pad <= 64 * 8 * SB_MAX, T2V_ESHAPE, "t2v_softmax_bwd_rows: row length");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p,
                       (bf16_t*)dp, rows, n, n_pad, ld);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_transpose_bf16(const void* in, int ld_in, int rows, int cols, void* out, int ld_out, int batch,
                                  long long in_stride, long long out_stride, void* stream) {
    T2V_REQUIRE(in && out && rows > 0 && cols > 0 && batch > 0 && ld_in >= cols && ld_out >= rows, T2V_EINVAL,
                "t2v_transpose_bf16: bad argument");
    T2V_REQUIRE(batch <= 65535, T2V_ESHAPE, "t2v_transpose_bf16: batch");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, ld_in, rows, cols, (bf16_t*)out, ld_out, in_stride, out_stride);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_sumpool2x2(const void* in, int n_img, int h, int w, int C, void* out, void* stream) {
    T2V_REQUIRE(in && out && n_img > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, T2V_EINVAL, "t2v_sumpool2x2: bad argument");
    const long long total = (long long)n_img * h * w * (C / 8);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       n_img, h, w, C, (bf16_t*)out);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
