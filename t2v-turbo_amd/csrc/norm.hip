// GroupNorm (two-phase, virtual-concat aware), LayerNorm and row softmax for gfx950.
// All three are HBM-bandwidth bound: 16-byte (8 x bf16) per-lane accesses, fp32 math,
// wave-shuffle (64-lane butterfly) reductions, no re-reads beyond what the algorithm needs.
#include "common.h"

namespace {

// Thread decomposition shared by gn_stats / gn_apply: a block owns a slab of rows of ONE
// statistics unit; thread (cx, ry) owns channel chunks cx + j*tx (8 channels each) for rows
// ry + i*ty.  Each thread therefore sees the same channels on every row it touches, so the
// per-channel scale/shift (apply) or partial sums (stats) stay in registers.
struct GnGeom {
    int cpr;  // 16-byte chunks per row = C/8
    int cpt;  // chunks per thread
    int tx, ty;
};
__host__ __device__ inline GnGeom gn_geom(int C) {
    GnGeom g;
    g.cpr = C / 8;
    g.cpt = (g.cpr + 255) / 256;
    g.tx = (g.cpr + g.cpt - 1) / g.cpt;
    g.ty = 256 / g.tx;
    if (g.ty < 1) g.ty = 1;
    return g;
}
constexpr int GN_MAX_CPT = 2;      // C <= 4096
// rows per block: sized so that a call launches ~1024 blocks (>= 32, <= 256 rows)
inline int gn_slab_rows(int n_units, int rows_per_unit) {
    long long total = (long long)n_units * rows_per_unit;
    long long s = (total + 1023) / 1024;
    s = (s + 7) / 8 * 8;
    if (s < 32) s = 32;
    if (s > 256) s = 256;
    return (int)s;
}

__device__ __forceinline__ const bf16_t* gn_src(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int ld1,
                                               long long row, int c) {
    return c < c0 ? x0 + row * ld0 + c : x1 + row * ld1 + (c - c0);
}

// partial[unit][slab][group][2] = (sum, sumsq).  Deterministic: per-thread register sums -> LDS
// [row-lane][channel] -> fixed-order reduction over row-lanes, then over the channels of a group.
__global__ __launch_bounds__(256) void gn_stats_partial(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int c1,
                                                        int ld1, int rows_per_unit, int groups, int slab_rows, float* partial) {
    extern __shared__ float sred[];  // [2][ty][C] then reused as [2][C]
    const int C = c0 + c1, cpg = C / groups;
    const GnGeom g = gn_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x;
    const int cx = tid % g.tx, ry = tid / g.tx;
    const int r0 = slab * slab_rows;
    const int r1 = min(r0 + slab_rows, rows_per_unit);
    float s[GN_MAX_CPT][8], q[GN_MAX_CPT][8];
#pragma unroll
    for (int j = 0; j < GN_MAX_CPT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }
    float* ssum = sred;
    float* ssq = sred + g.ty * C;
    if (ry < g.ty) {
        // 4 rows per trip: the 4 (x cpt) 16-byte loads are issued before any is consumed
        for (int r = r0 + ry; r < r1; r += 4 * g.ty) {
#pragma unroll
            for (int j = 0; j < GN_MAX_CPT; ++j) {
                const int ci = cx + j * g.tx;
                if (j < g.cpt && ci < g.cpr) {
                    uint4 u[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rr = r + t * g.ty;
                        u[t] = rr < r1 ? *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, (long long)unit * rows_per_unit + rr, ci * 8)
                                       : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float f[8];
                        unpack8(u[t], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s[j][e] += f[e]; q[j][e] += f[e] * f[e]; }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GN_MAX_CPT; ++j) {
            const int ci = cx + j * g.tx;
            if (j < g.cpt && ci < g.cpr) {
                *(float4*)(ssum + ry * C + ci * 8) = make_float4(s[j][0], s[j][1], s[j][2], s[j][3]);
                *(float4*)(ssum + ry * C + ci * 8 + 4) = make_float4(s[j][4], s[j][5], s[j][6], s[j][7]);
                *(float4*)(ssq + ry * C + ci * 8) = make_float4(q[j][0], q[j][1], q[j][2], q[j][3]);
                *(float4*)(ssq + ry * C + ci * 8 + 4) = make_float4(q[j][4], q[j][5], q[j][6], q[j][7]);
            }
        }
    }
    __syncthreads();
    // reduce over row-lanes: thread -> channel(s); result parked in row-lane 0's slots
    for (int c = tid; c < C; c += 256) {
        float a = ssum[c], b = ssq[c];
        for (int y = 1; y < g.ty; ++y) { a += ssum[y * C + c]; b += ssq[y * C + c]; }
        ssum[c] = a; ssq[c] = b;
    }
    __syncthreads();
    // partial is [unit][value = 2*group + stat][slab]: the final pass reads each value's slabs contiguously
    for (int i = tid; i < groups * 2; i += 256) {
        const int grp = i >> 1;
        const float* src = (i & 1) ? ssq : ssum;
        float a = 0.f;
        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) a += src[c];
        partial[((long long)unit * groups * 2 + i) * nslab + slab] = a;
    }
}

// one block per unit: thread = (part, value) walks slabs part, part+parts, ... in a fixed order;
// values = (sum, sumsq) of each group (groups*2 <= 256)
__global__ __launch_bounds__(256) void gn_stats_final(const float* partial, int nslab, int groups, float inv_count, float eps, float* stats) {
    __shared__ double sh[256];
    const int unit = blockIdx.x, tid = threadIdx.x;
    const int width = groups * 2, parts = 256 / width;
    const int v = tid % width, part = tid / width;
    const float* base = partial + ((long long)unit * width + v) * nslab;
    const int chunk = (nslab + parts - 1) / parts;
    double acc = 0.0;
    if (part < parts) {
        const int k1 = min(nslab, (part + 1) * chunk);
#pragma unroll 8
        for (int k = part * chunk; k < k1; ++k) acc += (double)base[k];
    }
    sh[tid] = acc;
    __syncthreads();
    double t = 0.0;
    if (tid < width)
        for (int pz = 0; pz < parts; ++pz) t += sh[pz * width + tid];
    __syncthreads();
    if (tid < width) sh[tid] = t;
    __syncthreads();
    if (tid < groups) {
        const double mean = sh[2 * tid] * inv_count;
        double var = sh[2 * tid + 1] * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((long long)unit * groups + tid) * 2] = (float)mean;
        stats[((long long)unit * groups + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int c1,
                                                       int ld1, int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                       const float* gamma, const float* beta, int silu, bf16_t* out,
                                                       int ldo) {
    const int C = c0 + c1, cpg = C / groups;
    const GnGeom g = gn_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x;
    const int tid = threadIdx.x;
    const int cx = tid % g.tx, ry = tid / g.tx;
    if (ry >= g.ty) return;
    float sc[GN_MAX_CPT][8], sh[GN_MAX_CPT][8];
#pragma unroll
    for (int j = 0; j < GN_MAX_CPT; ++j) {
        const int ci = cx + j * g.tx;
        if (j < g.cpt && ci < g.cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = ci * 8 + e;
                const float* st = stats + ((long long)unit * groups + c / cpg) * 2;
                const float a = st[1] * gamma[c];
                sc[j][e] = a;
                sh[j][e] = beta[c] - st[0] * a;
            }
        }
    }
    const int r0 = slab * slab_rows;
    const int r1 = min(r0 + slab_rows, rows_per_unit);
    for (int r = r0 + ry; r < r1; r += 4 * g.ty) {
#pragma unroll
        for (int j = 0; j < GN_MAX_CPT; ++j) {
            const int ci = cx + j * g.tx;
            if (j < g.cpt && ci < g.cpr) {
                uint4 u[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = r + t * g.ty;
                    if (rr < r1) u[t] = *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, (long long)unit * rows_per_unit + rr, ci * 8);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = r + t * g.ty;
                    if (rr < r1) {
                        float f[8];
                        unpack8(u[t], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = f[e] * sc[j][e] + sh[j][e];
                            f[e] = silu ? silu_f(v) : v;
                        }
                        *(uint4*)(out + ((long long)unit * rows_per_unit + rr) * ldo + ci * 8) = pack8(f);
                    }
                }
            }
        }
    }
}

// ---- LayerNorm: one wave per row, row held in registers (C <= 64*8*LN_MAX = 4096) -------------
constexpr int LN_MAX = 8;
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* x, int ldx, int M, int C, const float* gamma,
                                                        const float* beta, float eps, bf16_t* out, int ldo) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cpr = C / 8;
    float v[LN_MAX][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            unpack8(*(const uint4*)(x + row * ldx + ci * 8), v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mean; q += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < LN_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            float o[8];
            const float4 g0 = *(const float4*)(gamma + ci * 8), g1 = *(const float4*)(gamma + ci * 8 + 4);
            const float4 b0 = *(const float4*)(beta + ci * 8), b1 = *(const float4*)(beta + ci * 8 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * gg[e] + bb[e];
            *(uint4*)(out + row * ldo + ci * 8) = pack8(o);
        }
    }
}

// ---- row softmax, one wave per row, row in registers (n_pad <= 64*8*SM_MAX = 4096) -----------
constexpr int SM_MAX = 8;
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* s, long long rows, int n, int n_pad, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* p = s + row * ld;
    const int cpr = n_pad / 8;
    float v[SM_MAX][8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SM_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            unpack8(*(const uint4*)(p + ci * 8), v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (ci * 8 + e >= n) v[j][e] = -INFINITY;
                mx = fmaxf(mx, v[j][e]);
            }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SM_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[j][e] = __expf(v[j][e] - mx); sum += v[j][e]; }
        }
    }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int j = 0; j < SM_MAX; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] *= inv;
            *(uint4*)(p + ci * 8) = pack8(v[j]);
        }
    }
}

int gn_check(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units, int rows_per_unit, int groups) {
    T2V_REQUIRE(x0 && n_units > 0 && rows_per_unit > 0 && groups > 0, T2V_EINVAL, "groupnorm: bad argument");
    if (!x1) c1 = 0;
    const int C = c0 + c1;
    T2V_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && ld0 % 8 == 0 && (c1 == 0 || ld1 % 8 == 0), T2V_ESHAPE,
                "groupnorm: channels / strides must be multiples of 8");
    T2V_REQUIRE(C % groups == 0 && C / 8 <= 256 * GN_MAX_CPT, T2V_ESHAPE, "groupnorm: unsupported channel count");
    T2V_REQUIRE(groups * 2 <= 256, T2V_ESHAPE, "groupnorm: too many groups");
    return T2V_OK;
}

}  // namespace

static int gn_nslab(int n_units, int rows_per_unit) {
    const int sr = gn_slab_rows(n_units, rows_per_unit);
    return (rows_per_unit + sr - 1) / sr;
}

extern "C" long long t2v_gn_ws_floats(int n_units, int rows_per_unit, int groups) {
    return (long long)n_units * gn_nslab(n_units, rows_per_unit) * groups * 2;
}

extern "C" int t2v_gn_stats(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                            int rows_per_unit, int groups, float eps, float* ws, float* stats, void* stream) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups);
    if (rc) return rc;
    T2V_REQUIRE(ws && stats, T2V_EINVAL, "t2v_gn_stats: null workspace");
    if (!x1) { c1 = 0; ld1 = 0; }
    hipStream_t s = (hipStream_t)stream;
    const int nslab = gn_nslab(n_units, rows_per_unit);
    const int slab_rows = gn_slab_rows(n_units, rows_per_unit);
    const GnGeom gg = gn_geom(c0 + c1);
    hipLaunchKernelGGL(gn_stats_partial, dim3(nslab, n_units), dim3(256), (size_t)2 * gg.ty * (c0 + c1) * sizeof(float), s,
                       (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, rows_per_unit, groups, slab_rows, ws);
    T2V_CHECK_LAUNCH();
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)((c0 + c1) / groups));
    hipLaunchKernelGGL(gn_stats_final, dim3(n_units), dim3(256), 0, s, (const float*)ws, nslab, groups, inv_count, eps, stats);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_gn_apply(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                            int rows_per_unit, int groups, const float* stats, const float* gamma, const float* beta,
                            int silu, void* out, int ldo, void* stream) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups);
    if (rc) return rc;
    T2V_REQUIRE(stats && gamma && beta && out && ldo % 8 == 0, T2V_EINVAL, "t2v_gn_apply: bad argument");
    if (!x1) { c1 = 0; ld1 = 0; }
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_nslab(n_units, rows_per_unit), n_units), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, rows_per_unit, groups,
                       gn_slab_rows(n_units, rows_per_unit), stats, gamma, beta,
                       silu, (bf16_t*)out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_layernorm(const void* x, int ldx, int M, int C, const float* gamma, const float* beta, float eps,
                             void* out, int ldo, void* stream) {
    T2V_REQUIRE(x && gamma && beta && out && M > 0, T2V_EINVAL, "t2v_layernorm: bad argument");
    T2V_REQUIRE(C % 8 == 0 && C <= 64 * 8 * LN_MAX && ldx % 8 == 0 && ldo % 8 == 0, T2V_ESHAPE, "t2v_layernorm: unsupported C");
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, M, C,
                       gamma, beta, eps, (bf16_t*)out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_softmax_rows(void* sm, long long rows, int n, int n_pad, int ld, void* stream) {
    T2V_REQUIRE(sm && rows > 0 && n > 0 && n_pad >= n, T2V_EINVAL, "t2v_softmax_rows: bad argument");
    T2V_REQUIRE(n_pad % 8 == 0 && ld % 8 == 0 && n_pad <= ld && n_pad <= 64 * 8 * SM_MAX, T2V_ESHAPE,
                "t2v_softmax_rows: unsupported row length");
    T2V_REQUIRE((rows + 3) / 4 < 2147483647LL, T2V_ESHAPE, "t2v_softmax_rows: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)sm,
                       rows, n, n_pad, ld);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
