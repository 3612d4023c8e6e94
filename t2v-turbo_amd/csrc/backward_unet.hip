// Backward (dX) kernels of the UNet data-gradient path (engine_unet_bwd.py; include/t2v_hip.h "backward (dX) pieces of the UNet").
// Written in round 1 after its GPU budget was spent, validated on MI355X in round 2 (tests/test_gpu_unet_grad.py).  Also executed thread by thread on the host SIMT simulator
// (tests/hostsim, tests/test_hostsim_kernels.py) against the emulated backend's definitions (tests/emu_ops.py).  Kept in a translation unit of their own so that the validated kernels of backward.hip
// compile to exactly the code that ran (adding kernels to that file changed the code generated for gn_bwd_apply_kernel).
#include "common.h"
#include "gn_bwd_common.h"

namespace {

// ---- two-part / wide GroupNorm backward (the UNet's skip concats: up to 2560 channels from two tensors).  A separate copy of
// the kernels above, so that the validated single-tensor path of the VAE decoder stays byte-for-byte what ran on hardware.
struct Gb2Geom { int cpr, tx, ty, cpt; };
__host__ __device__ inline Gb2Geom gb2_geom(int C) {
    Gb2Geom g;
    g.cpr = C / 8;
    g.tx = g.cpr < 256 ? g.cpr : 256;  // threads across a row; wider rows (the 2560-channel skip concats) take cpt chunks each
    g.cpt = (g.cpr + g.tx - 1) / g.tx;
    g.ty = 256 / g.tx;
    if (g.ty < 1) g.ty = 1;
    return g;
}
// row `row`, channel c of the (virtually concatenated) input [x0 | x1]
__device__ __forceinline__ const bf16_t* gb_src(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int ld1, long long row, int c) {
    return c < c0 ? x0 + row * ld0 + c : x1 + row * ld1 + (c - c0);
}
inline int gb2_slab_rows(int C, int rows_per_unit) {
    const Gb2Geom g = gb2_geom(C);
    int s = g.ty * GB_RPT * 2;
    const int need = (rows_per_unit + 1023) / 1024;
    if (s < need) s = (need + g.ty - 1) / g.ty * g.ty;
    return s;
}
// partial[unit][slab][2*group + {0: sum g, 1: sum g*xhat}] (fixed order: deterministic)
__global__ __launch_bounds__(256) void gn_bwd2_partial_kernel(const bf16_t* x, int xc0, int ldx, const bf16_t* x1, int ldx1,
                                                             const bf16_t* dy, int ldy, int C,
                                                             int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                             const float* gamma, const float* beta, int silu, float* partial) {
    extern __shared__ float sred[];  // [2][ty][C]
    const int cpg = C / groups;
    const Gb2Geom g = gb2_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x, cx = tid % g.tx, ry = tid / g.tx;
    const int r0 = slab * slab_rows, r1 = min(r0 + slab_rows, rows_per_unit);
    float* s1 = sred;
    float* s2 = sred + g.ty * C;
    for (int jc = 0; jc < g.cpt; ++jc) {
    const int ci = cx + jc * g.tx;
    if (ry < g.ty && ci < g.cpr) {
        GbChan k;
        gb_load_chan(k, stats, gamma, beta, unit, groups, cpg, ci * 8);
        float a1[8], a2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
        for (int r = r0 + ry; r < r1; r += GB_RPT * g.ty) {
            uint4 ux[GB_RPT], ud[GB_RPT];
#pragma unroll
            for (int t = 0; t < GB_RPT; ++t) {
                const int rr = r + t * g.ty;
                const long long row = (long long)unit * rows_per_unit + rr;
                ux[t] = rr < r1 ? *(const uint4*)gb_src(x, xc0, ldx, x1, ldx1, row, ci * 8) : make_uint4(0, 0, 0, 0);
                ud[t] = rr < r1 ? *(const uint4*)(dy + row * ldy + ci * 8) : make_uint4(0, 0, 0, 0);
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
        for (int e = 0; e < 8; ++e) { s1[ry * C + ci * 8 + e] = a1[e]; s2[ry * C + ci * 8 + e] = a2[e]; }
    }
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
__global__ __launch_bounds__(256) void gn_bwd2_apply_kernel(const bf16_t* x, int xc0, int ldx, const bf16_t* x1, int ldx1,
                                                           const bf16_t* dy, int ldy, int C,
                                                           int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                           const float* gamma, const float* beta, int silu, const float* bstats,
                                                           const bf16_t* resid, int ldr, bf16_t* dx, int ldo) {
    const int cpg = C / groups;
    const Gb2Geom g = gb2_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x;
    const int tid = threadIdx.x, cx = tid % g.tx, ry = tid / g.tx;
    if (ry >= g.ty) return;
    for (int jc = 0; jc < g.cpt; ++jc) {
    const int ci = cx + jc * g.tx;
    if (ci >= g.cpr) break;
    GbChan k;
    gb_load_chan(k, stats, gamma, beta, unit, groups, cpg, ci * 8);
    float m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int grp = (ci * 8 + e) / cpg;
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
                ux[t] = *(const uint4*)gb_src(x, xc0, ldx, x1, ldx1, row, ci * 8);
                ud[t] = *(const uint4*)(dy + row * ldy + ci * 8);
                ur[t] = resid ? *(const uint4*)(resid + row * ldr + ci * 8) : make_uint4(0, 0, 0, 0);
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
            *(uint4*)(dx + ((long long)unit * rows_per_unit + rr) * ldo + ci * 8) = pack8(o);
        }
    }
    }
}

// ---- UNet data-gradient pieces (engine_unet_bwd.py).  validated on MI355X (see the file header); also run on the host SIMT
// simulator against the emulated backend's definitions (tests/emu_ops.py). -------------------------------------------------

// dx = d/dx LayerNorm(x) . dy (+ resid): one wave per row, the row in registers (NJ 16-byte chunks per lane)
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* x, int ldx, int M, int C, const float* gamma, float eps,
                                                            const bf16_t* dy, int ldy, const bf16_t* resid, int ldr, bf16_t* dx, int ldo) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cpr = C / 8;
    float xv[NJ][8], gv[NJ][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            unpack8(*(const uint4*)(x + row * ldx + ci * 8), xv[j]);
            float d[8];
            unpack8(*(const uint4*)(dy + row * ldy + ci * 8), d);
            const float4 g0 = *(const float4*)(gamma + ci * 8), g1 = *(const float4*)(gamma + ci * 8 + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { gv[j][e] = d[e] * ga[e]; sum += xv[j][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[j][e] = 0.f; gv[j][e] = 0.f; }
        }
    }
    const float inv = 1.0f / (float)C;
    const float mean = wave_sum(sum) * inv;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (lane + j * 64 < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = xv[j][e] - mean; var += t * t; }
        }
    const float rstd = rsqrtf(wave_sum(var) * inv + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (lane + j * 64 < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[j][e] = (xv[j][e] - mean) * rstd;  // xhat
                s1 += gv[j][e];
                s2 += gv[j][e] * xv[j][e];
            }
        }
    const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            float r[8], o[8];
            if (resid) unpack8(*(const uint4*)(resid + row * ldr + ci * 8), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (gv[j][e] - m1 - xv[j][e] * m2) + (resid ? r[e] : 0.f);
            *(uint4*)(dx + row * ldo + ci * 8) = pack8(o);
        }
    }
}

// GEGLU on a packed pre-activation row: 64-column groups [32 value | 32 gate]; thread = 8 outputs
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* h, int ldh, long long M, int inner, bf16_t* out, int ldo) {
    const int cpr = inner / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * cpr) return;
    const long long row = idx / cpr;
    const int ci = (int)(idx - row * cpr), grp = ci >> 2, sub = ci & 3;
    const bf16_t* hp = h + row * ldh + grp * 64 + sub * 8;
    float v[8], g[8], o[8];
    unpack8(*(const uint4*)hp, v);
    unpack8(*(const uint4*)(hp + 32), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] * gelu_f(g[e]);
    *(uint4*)(out + row * ldo + ci * 8) = pack8(o);
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* h, int ldh, const bf16_t* dy, int ldy, long long M, int inner,
                                                        bf16_t* dh, int ldd) {
    const int cpr = inner / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * cpr) return;
    const long long row = idx / cpr;
    const int ci = (int)(idx - row * cpr), grp = ci >> 2, sub = ci & 3;
    const bf16_t* hp = h + row * ldh + grp * 64 + sub * 8;
    float v[8], g[8], d[8], dv[8], dg[8];
    unpack8(*(const uint4*)hp, v);
    unpack8(*(const uint4*)(hp + 32), g);
    unpack8(*(const uint4*)(dy + row * ldy + ci * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float cdf = 0.5f * (1.0f + erff(g[e] * 0.70710678118654752f));
        const float pdf = __expf(-0.5f * g[e] * g[e]) * 0.3989422804014327f;
        dv[e] = d[e] * g[e] * cdf;
        dg[e] = d[e] * v[e] * (cdf + g[e] * pdf);
    }
    bf16_t* op = dh + row * ldd + grp * 64 + sub * 8;
    *(uint4*)op = pack8(dv);
    *(uint4*)(op + 32) = pack8(dg);
}

// adjoint of the stride-2 sampling of a 3x3 s2 p1 conv: out[n][2y][2x] = src[n][y][x], zero elsewhere; thread = 8 channels
__global__ __launch_bounds__(256) void scatter2x_kernel(const bf16_t* src, int n_img, int h, int w, int C, int H, int W, bf16_t* out) {
    const int cpr = C / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n_img * H * W * cpr;
    if (idx >= total) return;
    const int ci = (int)(idx % cpr);
    const long long pix = idx / cpr;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!(X & 1) && !(Y & 1) && (Y >> 1) < h && (X >> 1) < w) v = *(const uint4*)(src + ((n * h + (Y >> 1)) * w + (X >> 1)) * C + ci * 8);
    *(uint4*)(out + pix * C + ci * 8) = v;
}

// out = a + b over [M][C] bf16 with row strides (the second operand is usually a column slice of a concat gradient)
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* out, int ldo,
                                                       long long M, int C) {
    const int cpr = C / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * cpr) return;
    const long long row = idx / cpr;
    const int ci = (int)(idx - row * cpr);
    float fa[8], fb[8], o[8];
    unpack8(*(const uint4*)(a + row * lda + ci * 8), fa);
    unpack8(*(const uint4*)(b + row * ldb + ci * 8), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fa[e] + fb[e];
    *(uint4*)(out + row * ldo + ci * 8) = pack8(o);
}

// Temporal attention backward: one wave per (clip, pixel, head), F <= 16 frames, head dim 64.  Plain VALU through LDS
// (80 K MACs per problem); the forward's MFMA formulation is the obvious next step once this one is validated.
//   P = softmax(scale Q K^T); dP = dO V^T (+ dprobs); dS = P (dP - rowsum(P dP)); dQ = scale dS K; dK = scale dS^T Q; dV = P^T dO
constexpr int TB_F = 16, TB_LD = 65;
__global__ __launch_bounds__(128) void attn_temporal_bwd_kernel(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv,
                                                                const bf16_t* dout, int ldo, const float* dprobs, bf16_t* dq, int ldq2,
                                                                bf16_t* dk, int ldk2, bf16_t* dv, int ldv2, int n_clips, int F, int hw,
                                                                int heads, float scale) {
    __shared__ float sm[2][4][TB_F][TB_LD];   // per wave: Q, K, V, dO as fp32 [frame][channel]
    __shared__ float sp[2][2][TB_F][TB_F + 1];  // per wave: P, dS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long total = (long long)n_clips * hw * heads;
    float (*Q)[TB_LD] = sm[wave][0];
    float (*K)[TB_LD] = sm[wave][1];
    float (*V)[TB_LD] = sm[wave][2];
    float (*DO)[TB_LD] = sm[wave][3];
    float (*P)[TB_F + 1] = sp[wave][0];
    float (*DS)[TB_F + 1] = sp[wave][1];
    const int i = lane >> 2, jq = lane & 3;  // score layout: lane -> query frame i, key frames 4*jq .. 4*jq+3
    for (long long base = (long long)blockIdx.x * 2; base < total; base += (long long)gridDim.x * 2) {
        const long long prob = base + wave;
        const bool active = prob < total;  // (the loop trip count is the same for both waves: __syncthreads below is safe)
        const int hd = active ? (int)(prob % heads) : 0;
        const long long bp = active ? prob / heads : 0;
        const int pix = (int)(bp % hw);
        const long long clip = bp / hw;
        if (active) {
            for (int f = 0; f < F; ++f) {
                const long long row = (clip * F + f) * hw + pix;
                Q[f][lane] = bf2f(q[row * ldq + hd * 64 + lane]);
                K[f][lane] = bf2f(k[row * ldk + hd * 64 + lane]);
                V[f][lane] = bf2f(v[row * ldv + hd * 64 + lane]);
                DO[f][lane] = bf2f(dout[row * ldo + hd * 64 + lane]);
            }
        }
        __syncthreads();
        float s[4], dp[4];
        if (active) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { s[jj] = 0.f; dp[jj] = 0.f; }
            if (i < F) {
                for (int c = 0; c < 64; ++c) {
                    const float qv = Q[i][c], dov = DO[i][c];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * jq + jj;
                        if (j < F) { s[jj] += qv * K[j][c]; dp[jj] += dov * V[j][c]; }
                    }
                }
            }
            // softmax over the F keys of row i: 4 lanes x 4 entries
            float mx = -3.0e38f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                s[jj] *= scale;
                if (i < F && 4 * jq + jj < F) mx = fmaxf(mx, s[jj]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            float den = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                s[jj] = (i < F && 4 * jq + jj < F) ? __expf(s[jj] - mx) : 0.f;
                den += s[jj];
            }
            den += __shfl_xor(den, 1, 64);
            den += __shfl_xor(den, 2, 64);
            const float rden = den > 0.f ? 1.0f / den : 0.f;
            float dot = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * jq + jj;
                s[jj] *= rden;  // P[i][j]
                if (dprobs && i < F && j < F) dp[jj] += dprobs[(prob * F + i) * F + j];
                dot += s[jj] * dp[jj];
            }
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            if (i < F) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = 4 * jq + jj;
                    if (j < F) { P[i][j] = s[jj]; DS[i][j] = s[jj] * (dp[jj] - dot); }
                }
            }
        }
        __syncthreads();
        if (active) {
            for (int f = 0; f < F; ++f) {  // lane = channel
                float aq = 0.f, ak = 0.f, av = 0.f;
                for (int t = 0; t < F; ++t) {
                    aq += DS[f][t] * K[t][lane];   // dQ[f] = sum_j dS[f][j] K[j]
                    ak += DS[t][f] * Q[t][lane];   // dK[f] = sum_i dS[i][f] Q[i]
                    av += P[t][f] * DO[t][lane];   // dV[f] = sum_i P[i][f] dO[i]
                }
                const long long row = (clip * F + f) * hw + pix;
                dq[row * ldq2 + hd * 64 + lane] = f2bf(aq * scale);
                dk[row * ldk2 + hd * 64 + lane] = f2bf(ak * scale);
                dv[row * ldv2 + hd * 64 + lane] = f2bf(av);
            }
        }
        __syncthreads();
    }
}

// Temporal attention backward on the matrix cores: one wave per (clip, pixel, head), F <= 16 frames, head dim 64, no LDS.
// Every lane fetches ITS 16-byte slices of q_f / k_f / v_f / dO_f (frame f = lane & 15) straight from the token-major buffers,
// exactly the A / B fragments of v_mfma_f32_16x16x32_bf16 (contraction over the 64 channels):
//   S^T = K Q^T, dP^T = V dO^T   lane (i = lane&15, g = lane>>4) holds rows j = 4g + r of query column i: softmax statistics
//                                 (max, sum, rowsum(P dP)) are lane-local + 2 shuffles                         "layout A"
//   S   = Q K^T, dP   = dO V^T   the same products with the operands swapped: lane (j, g) holds rows i = 4g + r   "layout B"
//                                 (the statistics of query 4g + r come from lane 4g + r: three ds_bpermute each)
// The second-stage products contract over the FRAMES (v_mfma_f32_16x16x16_bf16, k = 4g + r), so dS^T (layout A) and dS, P
// (layout B) are already B operands, and the A operands are the transposed tiles X^T[c][f].  Those come from the matrix core
// too: X^T tile m = X-fragment(k step m>>1) x E, E the 0/1 matrix that selects channel 16m + n into column n — the result
// registers (rows f = 4g + r of column n) ARE the A fragment of X^T (row c = 16m + n, k = f); exact, so the bf16 repack is a
// truncation.  44 MFMAs (P, dS as hi + lo bf16 parts), ~180 VALU and no LDS traffic per problem; the VALU form above did ~2 800 LDS reads per lane.
//   dQ^T = K^T dS^T (scale folded into dS), dK^T = Q^T dS, dV^T = dO^T P: lane (row of the output token, g) holds channels
//   16m + 4g + r — 8-byte stores.
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short tb_bf16x4_t;
__device__ __forceinline__ tb_bf16x4_t tb_trunc4(f32x4_t t) {   // exact bf16 values held in fp32 -> bf16x4
    uint2 u;
    u.x = (__float_as_uint(t[1]) & 0xffff0000u) | (__float_as_uint(t[0]) >> 16);
    u.y = (__float_as_uint(t[3]) & 0xffff0000u) | (__float_as_uint(t[2]) >> 16);
    return *(tb_bf16x4_t*)&u;
}
// four fp32 values as hi + lo bf16 parts (hi = bf16(v), lo = bf16(v - hi)): the second-stage products take both, so P and dS enter
// them with ~16 bits of mantissa instead of 8 — the VALU kernel this one replaces kept them in fp32, and d/d(latents) of the whole
// student sits at 5.4-5.9e-2 of a 6e-2 tolerance; twelve more MFMAs on a kernel that waits for HBM
__device__ __forceinline__ void tb_split4(float a, float b, float c, float d, tb_bf16x4_t& hi, tb_bf16x4_t& lo) {
    uint2 h, l;
    h.x = pack2bf(a, b);
    h.y = pack2bf(c, d);
    l.x = pack2bf(a - __uint_as_float(h.x << 16), b - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack2bf(c - __uint_as_float(h.y << 16), d - __uint_as_float(h.y & 0xffff0000u));
    hi = *(tb_bf16x4_t*)&h;
    lo = *(tb_bf16x4_t*)&l;
}
__global__ __launch_bounds__(256) void attn_temporal_bwd_mfma_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                                     const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ dout, int ldo,
                                                                     const float* __restrict__ dprobs, bf16_t* __restrict__ dq, int ldq2,
                                                                     bf16_t* __restrict__ dk, int ldk2, bf16_t* __restrict__ dv, int ldv2,
                                                                     long long total, int F, int hw, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const long long prob = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prob >= total) return;  // wave-uniform
    const int hd = (int)(prob % heads);
    const long long bp = prob / heads;
    const int pix = (int)(bp % hw);
    const long long clip = bp / hw;
    const int l15 = lane & 15, g = lane >> 4;
    const bool f_ok = l15 < F;
    const long long row = (clip * F + (f_ok ? l15 : 0)) * hw + pix;
    const int col = hd * 64 + g * 8;
    // ---- fragments: frame l15, channels 8g .. 8g+7 (k step 0) and 32 + 8g .. (k step 1)
    bf16x8_t qf[2], kf[2], vf[2], of[2];
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 a0 = z, a1 = z, b0 = z, b1 = z, c0 = z, c1 = z, d0 = z, d1 = z;
        if (f_ok) {
            const bf16_t* qp = q + row * ldq + col;
            const bf16_t* kp = k + row * ldk + col;
            const bf16_t* vp = v + row * ldv + col;
            const bf16_t* op = dout + row * ldo + col;
            a0 = *(const uint4*)qp; a1 = *(const uint4*)(qp + 32);
            b0 = *(const uint4*)kp; b1 = *(const uint4*)(kp + 32);
            c0 = *(const uint4*)vp; c1 = *(const uint4*)(vp + 32);
            d0 = *(const uint4*)op; d1 = *(const uint4*)(op + 32);
        }
        qf[0] = *(bf16x8_t*)&a0; qf[1] = *(bf16x8_t*)&a1;
        kf[0] = *(bf16x8_t*)&b0; kf[1] = *(bf16x8_t*)&b1;
        vf[0] = *(bf16x8_t*)&c0; vf[1] = *(bf16x8_t*)&c1;
        of[0] = *(bf16x8_t*)&d0; of[1] = *(bf16x8_t*)&d1;
    }
    const f32x4_t zero4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t sT = zero4, dpT = zero4, sB = zero4, dpB = zero4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        sT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[s], sT, 0, 0, 0);    // S^T[j][i]
        dpT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s], of[s], dpT, 0, 0, 0);  // dP^T[j][i]
        sB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf[s], sB, 0, 0, 0);    // S[i][j]
        dpB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of[s], vf[s], dpB, 0, 0, 0);  // dP[i][j]
    }
    const float c2 = scale * 1.4426950408889634f;
    // ---- layout A: query i = l15, keys j = 4g + r
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * g + r < F) mx = fmaxf(mx, sT[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float pT[4], den = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pT[r] = (4 * g + r < F) ? __builtin_amdgcn_exp2f((sT[r] - mx) * c2) : 0.f;
        den += pT[r];
    }
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    const float rden = 1.0f / den;   // F >= 1: the maximum contributes 1
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pT[r] *= rden;
        if (dprobs && f_ok && 4 * g + r < F) dpT[r] += dprobs[(prob * F + l15) * F + 4 * g + r];
        dot += pT[r] * dpT[r];
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    tb_bf16x4_t dsT_b, dsT_l;
    tb_split4(pT[0] * (dpT[0] - dot) * scale, pT[1] * (dpT[1] - dot) * scale, pT[2] * (dpT[2] - dot) * scale,
              pT[3] * (dpT[3] - dot) * scale, dsT_b, dsT_l);
    // ---- layout B: key j = l15, queries i = 4g + r (their statistics live in lane i)
    float pB[4], dsB[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        const float mi = __shfl(mx, i, 64), ri = __shfl(rden, i, 64), di = __shfl(dot, i, 64);
        float dp = dpB[r];
        if (dprobs && f_ok && i < F) dp += dprobs[(prob * F + i) * F + l15];
        pB[r] = f_ok ? __builtin_amdgcn_exp2f((sB[r] - mi) * c2) * ri : 0.f;
        dsB[r] = pB[r] * (dp - di) * scale;
    }
    tb_bf16x4_t p_b, p_l, ds_b, ds_l;
    tb_split4(pB[0], pB[1], pB[2], pB[3], p_b, p_l);
    tb_split4(dsB[0], dsB[1], dsB[2], dsB[3], ds_b, ds_l);
    // ---- selection operands: E_h[c][n] = (c == 16h + n) within one 32-channel k step
    bf16x8_t sel[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (g == 2 * h + (l15 >> 3)) {
            const uint32_t one = (l15 & 1) ? 0x3f800000u : 0x00003f80u;
            const int wi = (l15 & 7) >> 1;
            w.x = wi == 0 ? one : 0u; w.y = wi == 1 ? one : 0u; w.z = wi == 2 ? one : 0u; w.w = wi == 3 ? one : 0u;
        }
        sel[h] = *(bf16x8_t*)&w;
    }
    bf16_t* dqp = dq + row * ldq2 + hd * 64 + 4 * g;
    bf16_t* dkp = dk + row * ldk2 + hd * 64 + 4 * g;
    bf16_t* dvp = dv + row * ldv2 + hd * 64 + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const tb_bf16x4_t kT = tb_trunc4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[m >> 1], sel[m & 1], zero4, 0, 0, 0));
        const tb_bf16x4_t qT = tb_trunc4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[m >> 1], sel[m & 1], zero4, 0, 0, 0));
        const tb_bf16x4_t oT = tb_trunc4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(of[m >> 1], sel[m & 1], zero4, 0, 0, 0));
        f32x4_t gq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kT, dsT_l, zero4, 0, 0, 0);  // dQ^T[16m + 4g + r][i = l15]
        f32x4_t gk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, ds_l, zero4, 0, 0, 0);   // dK^T[..][j = l15]
        f32x4_t gv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(oT, p_l, zero4, 0, 0, 0);    // dV^T[..][j = l15]
        gq = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kT, dsT_b, gq, 0, 0, 0);
        gk = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, ds_b, gk, 0, 0, 0);
        gv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(oT, p_b, gv, 0, 0, 0);
        if (f_ok) {
            uint2 w;
            w.x = pack2bf(gq[0], gq[1]); w.y = pack2bf(gq[2], gq[3]);
            *(uint2*)(dqp + 16 * m) = w;
            w.x = pack2bf(gk[0], gk[1]); w.y = pack2bf(gk[2], gk[3]);
            *(uint2*)(dkp + 16 * m) = w;
            w.x = pack2bf(gv[0], gv[1]); w.y = pack2bf(gv[2], gv[3]);
            *(uint2*)(dvp + 16 * m) = w;
        }
    }
}

}  // namespace

extern "C" int t2v_gn_bwd2(const void* x, int xc0, int ldx, const void* x1, int xc1, int ldx1, int n_units, int rows_per_unit,
                           int groups, const float* stats, const float* gamma, const float* beta, int silu, const void* dy, int ldy,
                           const void* resid, int ldr, float* ws, void* dx, int ldo, void* stream) {
    T2V_REQUIRE(x && dy && stats && gamma && beta && ws && dx, T2V_EINVAL, "t2v_gn_bwd: null pointer");
    T2V_REQUIRE(n_units > 0 && rows_per_unit > 0 && groups > 0 && groups <= 128, T2V_EINVAL, "t2v_gn_bwd: bad size");
    if (!x1) { xc1 = 0; ldx1 = 0; }
    const int C = xc0 + xc1;
    T2V_REQUIRE(xc0 % 8 == 0 && xc1 % 8 == 0 && C <= 4096 && C % groups == 0 && ldx % 8 == 0 && ldx1 % 8 == 0 && ldy % 8 == 0 &&
                ldo % 8 == 0 && (!resid || ldr % 8 == 0), T2V_ESHAPE, "t2v_gn_bwd: channels <= 4096, multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const Gb2Geom gg = gb2_geom(C);
    const int slab_rows = gb2_slab_rows(C, rows_per_unit);
    const int nslab = (rows_per_unit + slab_rows - 1) / slab_rows;
    float* partial = ws;
    float* bstats = ws + (long long)n_units * nslab * groups * 2;
    hipLaunchKernelGGL(gn_bwd2_partial_kernel, dim3(nslab, n_units), dim3(256), (size_t)2 * gg.ty * C * sizeof(float), s,
                       (const bf16_t*)x, xc0, ldx, (const bf16_t*)x1, ldx1, (const bf16_t*)dy, ldy, C, rows_per_unit, groups, slab_rows,
                       stats, gamma, beta, silu, partial);
    T2V_CHECK_LAUNCH();
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(n_units), dim3(GB_FINAL_THREADS), 0, s, (const float*)partial, nslab, groups, inv_count, bstats);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_bwd2_apply_kernel, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x, xc0, ldx, (const bf16_t*)x1, ldx1,
                       (const bf16_t*)dy, ldy, C,
                       rows_per_unit, groups, slab_rows, stats, gamma, beta, silu, (const float*)bstats, (const bf16_t*)resid, ldr,
                       (bf16_t*)dx, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// ---- UNet data-gradient entry points (see the kernels' note: not yet validated on hardware) -------------------------------------
extern "C" int t2v_layernorm_bwd(const void* x, int ldx, int M, int C, const float* gamma, float eps, const void* dy, int ldy,
                                 const void* resid, int ldr, void* dx, int ldo, void* stream) {
    T2V_REQUIRE(x && gamma && dy && dx && M > 0 && C > 0, T2V_EINVAL, "t2v_layernorm_bwd: bad argument");
    T2V_REQUIRE(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && (!resid || ldr % 8 == 0), T2V_ESHAPE,
                "t2v_layernorm_bwd: C <= 2048, multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 3) / 4)), blk(256);
    const int nj = (C / 8 + 63) / 64;
#define T2V_LNB(NJ)                                                                                                              \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NJ>, grid, blk, 0, s, (const bf16_t*)x, ldx, M, C, gamma, eps, (const bf16_t*)dy, ldy, \
                       (const bf16_t*)resid, ldr, (bf16_t*)dx, ldo)
    if (nj == 1) T2V_LNB(1); else if (nj == 2) T2V_LNB(2); else if (nj == 3) T2V_LNB(3); else T2V_LNB(4);
#undef T2V_LNB
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_geglu_fwd(const void* h, int ldh, long long M, int inner, void* out, int ldo, void* stream) {
    T2V_REQUIRE(h && out && M > 0 && inner > 0, T2V_EINVAL, "t2v_geglu_fwd: bad argument");
    T2V_REQUIRE(inner % 32 == 0 && ldh % 8 == 0 && ldo % 8 == 0 && ldh >= 2 * inner && ldo >= inner, T2V_ESHAPE, "t2v_geglu_fwd: inner % 32");
    const long long work = M * (inner / 8);
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ldh, M,
                       inner, (bf16_t*)out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_geglu_bwd(const void* h, int ldh, const void* dy, int ldy, long long M, int inner, void* dh, int ldd, void* stream) {
    T2V_REQUIRE(h && dy && dh && M > 0 && inner > 0, T2V_EINVAL, "t2v_geglu_bwd: bad argument");
    T2V_REQUIRE(inner % 32 == 0 && ldh % 8 == 0 && ldy % 8 == 0 && ldd % 8 == 0 && ldh >= 2 * inner && ldd >= 2 * inner && ldy >= inner,
                T2V_ESHAPE, "t2v_geglu_bwd: inner % 32");
    const long long work = M * (inner / 8);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, ldh,
                       (const bf16_t*)dy, ldy, M, inner, (bf16_t*)dh, ldd);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_scatter2x(const void* src, int n_img, int h, int w, int C, int H, int W, void* out, void* stream) {
    T2V_REQUIRE(src && out && n_img > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, T2V_EINVAL, "t2v_scatter2x: bad argument");
    T2V_REQUIRE((H == 2 * h || H == 2 * h - 1) && (W == 2 * w || W == 2 * w - 1), T2V_ESHAPE, "t2v_scatter2x: H in {2h-1, 2h}");
    const long long total = (long long)n_img * H * W * (C / 8);
    hipLaunchKernelGGL(scatter2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, n_img,
                       h, w, C, H, W, (bf16_t*)out);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, long long M, int C, void* stream) {
    T2V_REQUIRE(a && b && out && M > 0 && C > 0, T2V_EINVAL, "t2v_add_bf16: bad argument");
    T2V_REQUIRE(C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 &&
                (uintptr_t)out % 16 == 0, T2V_ESHAPE, "t2v_add_bf16: 16-byte aligned rows");
    const long long work = M * (C / 8);
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                       (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, C);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_attn_temporal_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                                     const float* dprobs, void* dq, int ldq2, void* dk, int ldk2, void* dv, int ldv2, int n_clips,
                                     int frames, int hw, int heads, float scale, void* stream) {
    T2V_REQUIRE(q && k && v && dout && dq && dk && dv && n_clips > 0 && frames > 0 && hw > 0 && heads > 0, T2V_EINVAL,
                "t2v_attn_temporal_bwd: bad argument");
    T2V_REQUIRE(frames <= TB_F, T2V_ESHAPE, "t2v_attn_temporal_bwd: at most 16 frames");
    const long long total = (long long)n_clips * hw * heads;
    T2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldq2 % 4 == 0 && ldk2 % 4 == 0 && ldv2 % 4 == 0, T2V_ESHAPE,
                "t2v_attn_temporal_bwd: row strides (16-byte fragment loads, 8-byte stores)");
    static const bool valu = [] { const char* e = getenv("T2V_TATTN_BWD_VALU"); return e && e[0] == '1'; }();  // the first (VALU / LDS) form, kept for A/B runs
    if (!valu) {
        hipLaunchKernelGGL(attn_temporal_bwd_mfma_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                           ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)dout, ldo, dprobs, (bf16_t*)dq, ldq2, (bf16_t*)dk,
                           ldk2, (bf16_t*)dv, ldv2, total, frames, hw, heads, scale);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
    long long blocks = (total + 1) / 2;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(attn_temporal_bwd_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                       (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)dout, ldo, dprobs, (bf16_t*)dq, ldq2, (bf16_t*)dk, ldk2,
                       (bf16_t*)dv, ldv2, n_clips, frames, hw, heads, scale);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
