// GroupNorm (two-phase, virtual-concat aware), LayerNorm and row softmax for gfx950.
// All three are HBM-bandwidth bound: 16-byte (8 x bf16) per-lane accesses, fp32 math,
// wave-shuffle (64-lane butterfly) reductions, no re-reads beyond what the algorithm needs.
#include "common.h"
#include <cstdlib>

namespace {

// Thread decomposition shared by the GroupNorm kernels: a block owns a slab of rows of ONE
// statistics unit; thread (cx, ry) owns channel chunks cx + j*tx (8 channels each) for rows
// ry + i*ty.  Each thread therefore sees the same channels on every row it touches, so the
// per-channel scale/shift (apply) or partial sums (stats) stay in registers.  A thread keeps
// GN_RPT rows (16-byte loads) in flight; a slab is ty*GN_RPT rows, so small tensors still
// spread over many blocks (these kernels are latency-bound below a few MB) and a call never
// has more than 1024 slabs per unit.
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
constexpr int GN_RPT = 8;          // rows in flight per thread
constexpr int GN_MAX_SLABS = 1024;
constexpr int GN_FUSE_SLABS = 128;  // up to this many slabs the apply kernel finishes the statistics itself
// slab limits: a finishing thread holds at most 64 partials (gn_finish_unit), with NT / (2*groups) threads per value
inline int gn_max_slabs(int groups) { const int m = 64 * (1024 / (2 * groups)); return m < GN_MAX_SLABS ? m : GN_MAX_SLABS; }
inline int gn_fuse_slabs(int groups) { const int m = 64 * (256 / (2 * groups)); return m < GN_FUSE_SLABS ? m : GN_FUSE_SLABS; }
inline int gn_slab_rows(int C, int rows_per_unit, int groups) {
    const GnGeom g = gn_geom(C);
    int s = g.ty * GN_RPT;
    const int cap = gn_max_slabs(groups);
    const int need = (rows_per_unit + cap - 1) / cap;
    if (s < need) s = (need + g.ty - 1) / g.ty * g.ty;
    return s;
}

__device__ __forceinline__ const bf16_t* gn_src(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int ld1,
                                               long long row, int c) {
    return c < c0 ? x0 + row * ld0 + c : x1 + row * ld1 + (c - c0);
}

// partial[unit][slab][value = 2*group + stat] = (sum, sumsq) of the slab's rows.  Deterministic: per-thread
// register sums -> LDS [row-lane][channel] -> one thread per value adds its group's entries in a fixed order.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int c1,
                                                         int ld1, int rows_per_unit, int groups, int slab_rows, float* partial) {
    extern __shared__ float sred[];  // [2][ty][C]
    const int C = c0 + c1, cpg = C / groups;
    const GnGeom g = gn_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int tid = threadIdx.x;
    const int cx = tid % g.tx, ry = tid / g.tx;
    const int r0 = slab * slab_rows;
    const int r1 = min(r0 + slab_rows, rows_per_unit);
    float* ssum = sred;
    float* ssq = sred + g.ty * C;
    if (ry < g.ty) {
#pragma unroll
        for (int j = 0; j < GN_MAX_CPT; ++j) {
            const int ci = cx + j * g.tx;
            if (j < g.cpt && ci < g.cpr) {
                float s[8], q[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
                for (int r = r0 + ry; r < r1; r += GN_RPT * g.ty) {
                    uint4 u[GN_RPT];
#pragma unroll
                    for (int t = 0; t < GN_RPT; ++t) {
                        const int rr = r + t * g.ty;
                        u[t] = rr < r1 ? *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, (long long)unit * rows_per_unit + rr, ci * 8)
                                       : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < GN_RPT; ++t) {
                        float f[8];
                        unpack8(u[t], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
                    }
                }
                *(float4*)(ssum + ry * C + ci * 8) = make_float4(s[0], s[1], s[2], s[3]);
                *(float4*)(ssum + ry * C + ci * 8 + 4) = make_float4(s[4], s[5], s[6], s[7]);
                *(float4*)(ssq + ry * C + ci * 8) = make_float4(q[0], q[1], q[2], q[3]);
                *(float4*)(ssq + ry * C + ci * 8 + 4) = make_float4(q[4], q[5], q[6], q[7]);
            }
        }
    }
    __syncthreads();
    // partial is [unit][slab][value]: the finishing pass reads a slab's values with one coalesced load per wave
    for (int i = tid; i < groups * 2; i += 256) {
        const int grp = i >> 1;
        const float* src = ((i & 1) ? ssq : ssum) + grp * cpg;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four independent chains: the LDS reads pipeline
        for (int y = 0; y < g.ty; ++y) {
            const float* row = src + y * C;
            int c = 0;
            for (; c + 4 <= cpg; c += 4) { a0 += row[c]; a1 += row[c + 1]; a2 += row[c + 2]; a3 += row[c + 3]; }
            for (; c < cpg; ++c) a0 += row[c];
        }
        partial[((long long)unit * nslab + slab) * groups * 2 + i] = (a0 + a1) + (a2 + a3);
    }
}

// Statistics of one unit from its slab partials, in a fixed order (block-wide helper, NT threads): thread = (part, value)
// walks slabs [part*chunk, ...), then the parts are added in order.  Leaves (mean, rstd) of group g in sm[g], sr[g].
// COHERENT: the partials were published by other workgroups of the SAME launch with agent-scope atomic stores; read them with
// agent-scope atomic loads (served by L2 / the fabric, never by this CU's L1), so no acquire fence is needed.
template <int NT, int MAXCH, bool COHERENT = false>
__device__ __forceinline__ void gn_finish_unit(const float* partial, int unit, int nslab, int groups, float inv_count, float eps,
                                               double* sh /*[NT]*/, float* sm, float* sr) {
    auto ld = [](const float* p) -> float {
#ifndef T2V_HOSTSIM
        if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        return *p;
    };
    const int tid = threadIdx.x;
    const int width = groups * 2, parts = NT / width;
    const int v = tid % width, part = tid / width;
    const float* base = partial + (long long)unit * nslab * width + v;
    const int chunk = (nslab + parts - 1) / parts;
    double acc = 0.0;
    if (COHERENT && chunk <= 16) {
        // device-scope loads are served past the L1 (and across XCDs past the L2): a fabric round trip each, so ALL of a thread's
        // (at most 16) loads go out before the first add
        float t[16];
        const int k0 = part * chunk;
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = (part < parts && e < chunk && k0 + e < nslab) ? ld(base + (long long)(k0 + e) * width) : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += (double)t[e];
    } else if (part < parts) {
        const int k1 = min(nslab, (part + 1) * chunk);
        int k = part * chunk;
        for (; k + 8 <= k1; k += 8) {  // 8 coalesced loads in flight, then a fixed-order add
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = ld(base + (long long)(k + e) * width);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (double)t[e];
        }
        for (; k < k1; ++k) acc += (double)ld(base + (long long)k * width);
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
        sm[tid] = (float)mean;
        sr[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
}

// partial[unit][blk][value = 2*group + stat] from the COLUMN STATISTICS the producing t2v_gemm launches left behind
// (t2v_gemm_desc::colstat_out: cs[slab of 32 rows][channel][2] = (sum, sumsq) of the bf16 outputs; a virtual concat has one
// array per part): the statistics pass of GroupNorm without reading the tensor (1/8 of its bytes).  Block (blk, unit) adds the
// unit's 32-row slabs [blk * slabs_per_blk, ...) per channel in slab order, then one thread per value adds its group's channels
// in channel order: deterministic, same output format as gn_partial_kernel.
__global__ __launch_bounds__(256) void gn_partial_cs_kernel(const float* cs0, int c0, const float* cs1, int c1, int slabs_per_unit,
                                                            int slabs_per_blk, int groups, float* partial) {
    extern __shared__ float sred[];  // [C][2]
    const int C = c0 + c1, cpg = C / groups;
    const int unit = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int s0 = blk * slabs_per_blk, s1 = min(s0 + slabs_per_blk, slabs_per_unit);
    const long long first = (long long)unit * slabs_per_unit;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float* src = c < c0 ? cs0 + 2 * c : cs1 + 2 * (c - c0);
        const long long ldc = c < c0 ? 2LL * c0 : 2LL * c1;
        float a = 0.f, q = 0.f;
        int sl = s0;
        for (; sl + 4 <= s1; sl += 4) {  // four loads in flight, added in slab order
            float2 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = *(const float2*)(src + (first + sl + e) * ldc);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a += v[e].x; q += v[e].y; }
        }
        for (; sl < s1; ++sl) {
            const float2 v = *(const float2*)(src + (first + sl) * ldc);
            a += v.x; q += v.y;
        }
        sred[2 * c] = a;
        sred[2 * c + 1] = q;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < groups * 2; i += 256) {
        const float* src = sred + (long long)(i >> 1) * cpg * 2 + (i & 1);
        float a0 = 0.f, a1 = 0.f;
        int c = 0;
        for (; c + 2 <= cpg; c += 2) { a0 += src[2 * c]; a1 += src[2 * c + 2]; }
        if (c < cpg) a0 += src[2 * c];
        partial[((long long)unit * nblk + blk) * groups * 2 + i] = a0 + a1;
    }
}

// coef[unit][0][c] = rstd * gamma[c], coef[unit][1][c] = beta[c] - mean * rstd * gamma[c] STRAIGHT from the producers' column statistics:
// block (group, unit) owns one group.  Thread (pair, lane) adds the (sum, sumsq) float4 of channel pair `pair` of the group over
// the slabs lane, lane + nlane, ... (a thread's loads are independent, eight in flight per trip; NT = 1024 threads where 256 would
// need more than eight loads each: one unit of 40 960 rows is 1 280 slabs), every wave adds its 64 double pairs by butterfly, thread 0
// adds the waves' partials in order, finishes (mean, rstd) in double (-> `stats` when given: the training engine keeps them) and the
// first cpg threads write the group's coefficients.
// The apply pass (gn_apply_kernel<1>) then starts on its rows at once instead of finishing 80 partials per block behind three barriers.
constexpr int GN_CSD_LOADS = 32;
template <int NT>
__global__ __launch_bounds__(NT) void gn_coef_cs_kernel(const float* cs0, int c0, const float* cs1, int c1, int slabs_per_unit, int groups,
                                                         float inv_count, float eps, const float* gamma, const float* beta, float* coef, float* stats) {
    __shared__ double shs[NT / 64], shq[NT / 64];
    __shared__ float smr[2];
    const int C = c0 + c1, cpg = C / groups, ncp = cpg >> 1, nlane = NT / ncp;
    const int unit = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x;
    const int pair = tid % ncp, lane = tid / ncp;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < nlane) {
        const int c = grp * cpg + 2 * pair;
        const float* src = c < c0 ? cs0 + 2 * c : cs1 + 2 * (c - c0);
        const long long ldc = c < c0 ? 2LL * c0 : 2LL * c1;
        src += (long long)unit * slabs_per_unit * ldc;
        int sl = lane;
        for (; sl + 7 * nlane < slabs_per_unit; sl += 8 * nlane) {   // eight loads in flight per trip, added in slab order
            float4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *(const float4*)(src + (long long)(sl + e * nlane) * ldc);
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
        }
        for (; sl < slabs_per_unit; sl += nlane) {
            const float4 v = *(const float4*)(src + (long long)sl * ldc);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    // wave sums by butterfly (fixed order), one partial per wave through LDS, thread 0 adds the waves in order
    double ds = (double)acc.x + (double)acc.z, dq = (double)acc.y + (double)acc.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
    if ((tid & 63) == 0) { shs[tid >> 6] = ds; shq[tid >> 6] = dq; }
    __syncthreads();
    if (tid == 0) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { ts += shs[w]; tq += shq[w]; }
        const double mean = ts * (double)inv_count;
        double var = tq * (double)inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        smr[0] = (float)mean;
        smr[1] = (float)(1.0 / sqrt(var + (double)eps));
        if (stats) {
            stats[((long long)unit * groups + grp) * 2] = smr[0];
            stats[((long long)unit * groups + grp) * 2 + 1] = smr[1];
        }
    }
    if (!coef) return;
    __syncthreads();
    if (tid < cpg) {
        const int c = grp * cpg + tid;
        const float a = smr[1] * gamma[c];
        coef[(long long)unit * 2 * C + c] = a;
        coef[(long long)unit * 2 * C + C + c] = beta[c] - smr[0] * a;
    }
}

// one block per unit: stats[unit][group] = (mean, rstd) and, when coef != null, the per-channel affine
// coef[unit][0][c] = rstd*gamma[c], coef[unit][1][c] = beta[c] - mean*rstd*gamma[c] the apply pass streams with
__global__ __launch_bounds__(1024) void gn_final_kernel(const float* partial, int nslab, int groups, int C, float inv_count, float eps,
                                                        const float* gamma, const float* beta, float* stats, float* coef) {
    __shared__ double sh[1024];
    __shared__ float sm[128], sr[128];
    const int unit = blockIdx.x, tid = threadIdx.x;
    gn_finish_unit<1024, 64>(partial, unit, nslab, groups, inv_count, eps, sh, sm, sr);
    if (stats && tid < groups) {
        stats[((long long)unit * groups + tid) * 2] = sm[tid];
        stats[((long long)unit * groups + tid) * 2 + 1] = sr[tid];
    }
    if (coef) {
        const int cpg = C / groups;
        for (int c = tid; c < C; c += 1024) {
            const int grp = c / cpg;
            const float a = sr[grp] * gamma[c];
            coef[(long long)unit * 2 * C + c] = a;
            coef[(long long)unit * 2 * C + C + c] = beta[c] - sm[grp] * a;
        }
    }
}

// (Round 5, measured and NOT kept: issuing the first batch of row loads before gn_finish_unit — so that the 2-3 us of statistics finish
// would overlap the rows' flight — made the 117 fused-statistics GroupNorms of a UNet step 3 % SLOWER in-graph, 1.880 vs 1.821 ms
// (profiles/r05_ops_ingraph_ab.csv): the in-order vmcnt wait for the partials then also waits for the 8 rows queued in front of them.)
// MODE 0: (mean, rstd) per group in `stats`.  MODE 1: per-channel affine in `coef` (gn_final_kernel ran).
// MODE 2: slab partials in `partial` (nslab <= GN_FUSE_SLABS): every block finishes the statistics of its unit itself,
// which saves the final launch where launch latency, not bandwidth, is what a small tensor pays for.
template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x0, int c0, int ld0, const bf16_t* x1, int c1,
                                                       int ld1, int rows_per_unit, int groups, int slab_rows, const float* stats,
                                                       const float* coef, const float* partial, int nslab_stats, float inv_count,
                                                       float eps, const float* gamma, const float* beta, int silu, bf16_t* out,
                                                       int ldo, const char* pf, long long pf_lines) {
    const int C = c0 + c1, cpg = C / groups;
    const GnGeom g = gn_geom(C);
    const int unit = blockIdx.y, slab = blockIdx.x;
    const int tid = threadIdx.x;
    // Prefetch for the NEXT launch (the conv whose weights `pf` points at, up to tens of MB that the 256 MB Infinity Cache has
    // not seen since the previous step): every block touches its share of the 128-byte lines with streaming (nt) loads issued
    // before its own work; the values are only looked at after it, so their latency costs nothing here and the conv's first
    // K steps find the weights on the memory side of the fabric instead of in HBM.
    unsigned pf_acc = 0;
    if (pf_lines > 0) {
        const long long nb = (long long)gridDim.x * gridDim.y, b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        const long long per = (pf_lines + nb - 1) / nb;
        const long long l1 = min(pf_lines, (b + 1) * per);
        for (long long l = b * per + tid; l < l1; l += 256) pf_acc |= __builtin_nontemporal_load((const unsigned*)(pf + (l << 7)));
    }
    const int cx = tid % g.tx, ry = tid / g.tx;
    __shared__ double sh[MODE == 2 ? 256 : 1];
    __shared__ float sm[MODE == 2 ? 128 : 1], sr[MODE == 2 ? 128 : 1];
    if (MODE == 2) gn_finish_unit<256, 64>(partial, unit, nslab_stats, groups, inv_count, eps, sh, sm, sr);
    if (ry >= g.ty) return;
    float sc[GN_MAX_CPT][8], sf[GN_MAX_CPT][8];
#pragma unroll
    for (int j = 0; j < GN_MAX_CPT; ++j) {
        const int ci = cx + j * g.tx;
        if (j < g.cpt && ci < g.cpr) {
            if (MODE == 1) {
                const float* cf = coef + (long long)unit * 2 * C + ci * 8;
                const float4 a0 = *(const float4*)cf, a1 = *(const float4*)(cf + 4);
                const float4 b0 = *(const float4*)(cf + C), b1 = *(const float4*)(cf + C + 4);
                sc[j][0] = a0.x; sc[j][1] = a0.y; sc[j][2] = a0.z; sc[j][3] = a0.w;
                sc[j][4] = a1.x; sc[j][5] = a1.y; sc[j][6] = a1.z; sc[j][7] = a1.w;
                sf[j][0] = b0.x; sf[j][1] = b0.y; sf[j][2] = b0.z; sf[j][3] = b0.w;
                sf[j][4] = b1.x; sf[j][5] = b1.y; sf[j][6] = b1.z; sf[j][7] = b1.w;
            } else {
                const float4 g0 = *(const float4*)(gamma + ci * 8), g1 = *(const float4*)(gamma + ci * 8 + 4);
                const float4 b0 = *(const float4*)(beta + ci * 8), b1 = *(const float4*)(beta + ci * 8 + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                const int grp0 = (ci * 8) / cpg;
                int grp = grp0, next = (grp0 + 1) * cpg - ci * 8;  // channels left in the current group
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e == next) { ++grp; next += cpg; }
                    float mean, rstd;
                    if (MODE == 2) { mean = sm[grp]; rstd = sr[grp]; }
                    else { const float* st = stats + ((long long)unit * groups + grp) * 2; mean = st[0]; rstd = st[1]; }
                    const float a = rstd * gm[e];
                    sc[j][e] = a;
                    sf[j][e] = bt[e] - mean * a;
                }
            }
        }
    }
    const int r0 = slab * slab_rows;
    const int r1 = min(r0 + slab_rows, rows_per_unit);
    for (int r = r0 + ry; r < r1; r += GN_RPT * g.ty) {
#pragma unroll
        for (int j = 0; j < GN_MAX_CPT; ++j) {
            const int ci = cx + j * g.tx;
            if (j < g.cpt && ci < g.cpr) {
                uint4 u[GN_RPT];
#pragma unroll
                for (int t = 0; t < GN_RPT; ++t) {
                    const int rr = r + t * g.ty;
                    if (rr < r1) u[t] = *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, (long long)unit * rows_per_unit + rr, ci * 8);
                }
#pragma unroll
                for (int t = 0; t < GN_RPT; ++t) {
                    const int rr = r + t * g.ty;
                    if (rr < r1) {
                        float f[8];
                        unpack8(u[t], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = f[e] * sc[j][e] + sf[j][e];
                            f[e] = silu ? silu_f(v) : v;
                        }
                        *(uint4*)(out + ((long long)unit * rows_per_unit + rr) * ldo + ci * 8) = pack8(f);
                    }
                }
            }
        }
    }
    if (pf_lines > 0 && pf_acc == 0x7fc1a55au && ldo < 0) out[0] = 0;   // (never true: keeps the prefetch loads alive)
}

#ifndef T2V_HOSTSIM
// ---- GroupNorm(+SiLU) in ONE launch (t2v_group_norm, tensors that fit the chip's registers) ---------------------------------------
// The three-launch form reads x twice and pays three kernel boundaries for tensors that are mostly a few MB.  Here every
// workgroup (1024 threads, at most one per CU so that all of them are resident) keeps its slice of one statistics unit IN
// REGISTERS — thread (cx, ry) owns the 16-byte chunk cx of rows ry, ry + ty, ... (up to MAXC of them, all loads issued before
// the first use) — reduces it to per-group (sum, sum of squares), publishes those 2*groups floats, meets the other workgroups
// of its unit at a barrier, finishes the unit's statistics from all the partials in a fixed order (deterministic: no float
// atomics) and normalises its registers straight into the output: x is read once and y written once.
// Barrier (per unit): the partials travel as agent-scope atomic stores / loads on both sides (a valid hand-off form of
// MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": no release / acquire fences, which
// would write back and invalidate whole caches); stores drained -> arrival counter; the last arriver resets the counter and
// bumps a generation word, the others poll the generation with relaxed agent-scope loads + s_sleep.  Every workgroup reads the generation
// BEFORE it arrives, and the bump needs all arrivals, so nobody can miss it.  The spin is bounded: if the workgroups are not
// co-resident after all (a shared GPU), the launch finishes with a wrong result and the error word set instead of hanging.
constexpr int GC_NT = 1024;
constexpr int GC_SPIN_LIMIT = 1 << 21;

template <int MAXC>
__global__ __launch_bounds__(GC_NT) void gn_coop_kernel(const bf16_t* __restrict__ x0, int c0, int ld0, const bf16_t* __restrict__ x1, int c1,
                                                        int ld1, int rows_per_unit, int groups, int wgs_per_unit, int rows_per_wg,
                                                        float inv_count, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu, float* partial, unsigned* sync_words,
                                                        bf16_t* __restrict__ out, int ldo) {
    extern __shared__ float sred[];  // [2][ty][C]
    __shared__ double sh[GC_NT];
    __shared__ float sm[128], sr[128];
    const int C = c0 + c1, cpg = C / groups, cpr = C / 8;
    const int tx = cpr, ty = GC_NT / tx;
    const int tid = threadIdx.x;
    const int unit = blockIdx.x / wgs_per_unit, wg = blockIdx.x - unit * wgs_per_unit;
    const int cx = tid % tx, ry = tid / tx;
    const bool active = ry < ty;
    const int r0 = wg * rows_per_wg;
    const int r1 = min(r0 + rows_per_wg, rows_per_unit);
    const long long row_base = (long long)unit * rows_per_unit;
    uint4 u[MAXC];
    float* ssum = sred;
    float* ssq = sred + ty * C;
    if (active) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int rr = r0 + ry + i * ty;
            u[i] = rr < r1 ? *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, row_base + rr, cx * 8) : make_uint4(0, 0, 0, 0);
        }
        float sv[8], qv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sv[e] = 0.f; qv[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            float f[8];
            unpack8(u[i], f);  // rows past the slice are zero chunks: they add nothing
#pragma unroll
            for (int e = 0; e < 8; ++e) { sv[e] += f[e]; qv[e] += f[e] * f[e]; }
        }
        *(float4*)(ssum + ry * C + cx * 8) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        *(float4*)(ssum + ry * C + cx * 8 + 4) = make_float4(sv[4], sv[5], sv[6], sv[7]);
        *(float4*)(ssq + ry * C + cx * 8) = make_float4(qv[0], qv[1], qv[2], qv[3]);
        *(float4*)(ssq + ry * C + cx * 8 + 4) = make_float4(qv[4], qv[5], qv[6], qv[7]);
    }
    __syncthreads();
    // per-group sums of this workgroup: thread = (part, value) adds the thread-rows y = part, part + parts, ... of its group's
    // channels, the parts are then added in order
    {
        const int width = groups * 2, parts = GC_NT / width;
        const int v = tid % width, part = tid / width;
        float a = 0.f;
        if (part < parts) {
            const float* src = ((v & 1) ? ssq : ssum) + (v >> 1) * cpg;
            for (int y = part; y < ty; y += parts) {
                const float* row = src + y * C;
                float b0 = 0.f, b1 = 0.f;
                int c = 0;
                for (; c + 2 <= cpg; c += 2) { b0 += row[c]; b1 += row[c + 1]; }
                if (c < cpg) b0 += row[c];
                a += b0 + b1;
            }
        }
        sh[tid] = (double)a;
        __syncthreads();
        if (tid < width) {
            double t = 0.0;
            for (int pz = 0; pz < parts; ++pz) t += sh[pz * width + tid];
            // write-through (agent-scope atomic) store, drained before the block barrier below: the partials are visible at the
            // device's coherence point when thread 0 arrives at the counter, with NO release fence (a release writes back every
            // dirty line of the XCD's L2, i.e. the producer kernel's whole output: 4-8 us measured on the first version)
            __hip_atomic_store(partial + ((long long)unit * wgs_per_unit + wg) * width + tid, (float)t, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0 && wgs_per_unit > 1) {
        unsigned* cnt = sync_words + 2 * unit;
        unsigned* gen = cnt + 1;
        const unsigned my_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)wgs_per_unit - 1u) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int it = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen && it < GC_SPIN_LIMIT) {
                __builtin_amdgcn_s_sleep(2);
                ++it;
            }
            if (it >= GC_SPIN_LIMIT) __hip_atomic_store(sync_words + 2 * 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    gn_finish_unit<GC_NT, 64, true>(partial, unit, wgs_per_unit, groups, inv_count, eps, sh, sm, sr);
    if (!active) return;
    float sc[8], sf[8];
    {
        const float4 g0 = *(const float4*)(gamma + cx * 8), g1 = *(const float4*)(gamma + cx * 8 + 4);
        const float4 b0 = *(const float4*)(beta + cx * 8), b1 = *(const float4*)(beta + cx * 8 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        int grp = (cx * 8) / cpg, next = (grp + 1) * cpg - cx * 8;  // channels left in the current group
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e == next) { ++grp; next += cpg; }
            const float a = sr[grp] * gm[e];
            sc[e] = a;
            sf[e] = bt[e] - sm[grp] * a;
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int rr = r0 + ry + i * ty;
        if (rr < r1) {
            float f[8];
            unpack8(u[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = f[e] * sc[e] + sf[e];
                f[e] = silu ? silu_f(v) : v;
            }
            *(uint4*)(out + (row_base + rr) * ldo + cx * 8) = pack8(f);
        }
    }
}
#endif  // T2V_HOSTSIM

// ---- LayerNorm: a wave owns ROWS rows, each held in registers (NJ 16-byte chunks per lane, C <= 64*8*NJ);
// all ROWS*NJ loads are issued before the first reduction so a wave keeps several rows in flight ----------
constexpr int LN_MAX = 8;
template <int NJ, int ROWS>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* x, int ldx, int M, int C, const float* gamma,
                                                        const float* beta, float eps, bf16_t* out, int ldo) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= M) return;
    const int cpr = C / 8;
    uint4 u[ROWS][NJ];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = lane + j * 64;
            u[r][j] = (ci < cpr && row0 + r < M) ? *(const uint4*)(x + (row0 + r) * ldx + ci * 8) : make_uint4(0, 0, 0, 0);
        }
    float gg[NJ][8], bb[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = lane + j * 64;
        if (ci < cpr) {
            const float4 g0 = *(const float4*)(gamma + ci * 8), g1 = *(const float4*)(gamma + ci * 8 + 4);
            const float4 b0 = *(const float4*)(beta + ci * 8), b1 = *(const float4*)(beta + ci * 8 + 4);
            gg[j][0] = g0.x; gg[j][1] = g0.y; gg[j][2] = g0.z; gg[j][3] = g0.w; gg[j][4] = g1.x; gg[j][5] = g1.y; gg[j][6] = g1.z; gg[j][7] = g1.w;
            bb[j][0] = b0.x; bb[j][1] = b0.y; bb[j][2] = b0.z; bb[j][3] = b0.w; bb[j][4] = b1.x; bb[j][5] = b1.y; bb[j][6] = b1.z; bb[j][7] = b1.w;
        }
    }
    const float inv_c = 1.0f / (float)C;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (row0 + r >= M) break;
        float v[NJ][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            unpack8(u[r][j], v[j]);  // chunks past the row end are zero and add nothing
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (lane + j * 64 < cpr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mean; q += dlt * dlt; }
            }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = lane + j * 64;
            if (ci < cpr) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * gg[j][e] + bb[j][e];
                *(uint4*)(out + (row0 + r) * ldo + ci * 8) = pack8(o);
            }
        }
    }
}

// ---- LayerNorm, row-group form (round 5): LPR lanes own one row — CPL 16-byte chunks each, chunk sub + k * LPR, so a group reads
// whole 16 * LPR-byte runs — and a wave holds 64 / LPR rows x RI iterations in registers.  The one-row-per-wave kernel above leaves
// 24 of 64 lanes idle at C = 320 (40 chunks) and pays two 6-stage butterflies per row; here every lane carries data at the UNet's
// widths (320 / 640 / 1280 = 8 / 16 / 32 lanes x 5 chunks) and a reduction is log2(LPR) stages for 64 / LPR rows at once.  Same
// two-pass arithmetic (mean, then centred variance), so results differ from the other kernel only by fp32 summation order.
template <int LPR, int CPL, int RI>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* x, int ldx, int M, int C, const float* gamma,
                                                             const float* beta, float eps, bf16_t* out, int ldo) {
    constexpr int RPW = 64 / LPR;   // rows per wave and iteration
    const int lane = threadIdx.x & 63, sub = lane % LPR, rl = lane / LPR;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RPW * RI) + rl;
    uint4 u[RI][CPL];
#pragma unroll
    for (int it = 0; it < RI; ++it)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const long long row = row0 + it * RPW;
            u[it][k] = row < M ? *(const uint4*)(x + row * ldx + (sub + k * LPR) * 8) : make_uint4(0, 0, 0, 0);
        }
    float gg[CPL][8], bb[CPL][8];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = (sub + k * LPR) * 8;
        const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
        const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
        gg[k][0] = g0.x; gg[k][1] = g0.y; gg[k][2] = g0.z; gg[k][3] = g0.w; gg[k][4] = g1.x; gg[k][5] = g1.y; gg[k][6] = g1.z; gg[k][7] = g1.w;
        bb[k][0] = b0.x; bb[k][1] = b0.y; bb[k][2] = b0.z; bb[k][3] = b0.w; bb[k][4] = b1.x; bb[k][5] = b1.y; bb[k][6] = b1.z; bb[k][7] = b1.w;
    }
    const float inv_c = 1.0f / (float)C;
#pragma unroll
    for (int it = 0; it < RI; ++it) {
        const long long row = row0 + it * RPW;
        float v[CPL][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            unpack8(u[it][k], v[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[k][e];
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = v[k][e] - mean; q += dlt * dlt; }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * inv_c + eps);
        if (row < M) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                float o8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (v[k][e] - mean) * rstd * gg[k][e] + bb[k][e];
                *(uint4*)(out + row * ldo + (sub + k * LPR) * 8) = pack8(o8);
            }
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
    T2V_REQUIRE(groups <= 128, T2V_ESHAPE, "groupnorm: too many groups");
    return T2V_OK;
}

}  // namespace

static int gn_nslab(int C, int rows_per_unit, int groups) {
    const int sr = gn_slab_rows(C, rows_per_unit, groups);
    return (rows_per_unit + sr - 1) / sr;
}

// upper bound over every channel count (a slab is >= GN_RPT rows and a unit has <= GN_MAX_SLABS slabs)
extern "C" long long t2v_gn_ws_floats(int n_units, int rows_per_unit, int groups) {
    long long nslab = (rows_per_unit + GN_RPT - 1) / GN_RPT;
    if (nslab > GN_MAX_SLABS) nslab = GN_MAX_SLABS;
    return (long long)n_units * nslab * groups * 2;
}
extern "C" long long t2v_group_norm_ws_floats(int n_units, int rows_per_unit, int groups, int channels) {
    return t2v_gn_ws_floats(n_units, rows_per_unit, groups) + 2LL * n_units * channels;
}

static int gn_launch_partial(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units, int rows_per_unit,
                             int groups, float* ws, hipStream_t s) {
    const int C = c0 + c1;
    const GnGeom gg = gn_geom(C);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(gn_nslab(C, rows_per_unit, groups), n_units), dim3(256), (size_t)2 * gg.ty * C * sizeof(float), s,
                       (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, rows_per_unit, groups, gn_slab_rows(C, rows_per_unit, groups), ws);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_gn_stats(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                            int rows_per_unit, int groups, float eps, float* ws, float* stats, void* stream) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups);
    if (rc) return rc;
    T2V_REQUIRE(ws && stats, T2V_EINVAL, "t2v_gn_stats: null workspace");
    if (!x1) { c1 = 0; ld1 = 0; }
    hipStream_t s = (hipStream_t)stream;
    rc = gn_launch_partial(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups, ws, s);
    if (rc) return rc;
    const int C = c0 + c1;
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    hipLaunchKernelGGL(gn_final_kernel, dim3(n_units), dim3(1024), 0, s, (const float*)ws, gn_nslab(C, rows_per_unit, groups), groups, C,
                       inv_count, eps, (const float*)nullptr, (const float*)nullptr, stats, (float*)nullptr);
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
    const int C = c0 + c1;
    hipLaunchKernelGGL(gn_apply_kernel<0>, dim3(gn_nslab(C, rows_per_unit, groups), n_units), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, rows_per_unit, groups, gn_slab_rows(C, rows_per_unit, groups),
                       stats, (const float*)nullptr, (const float*)nullptr, 0, 0.f, 0.f, gamma, beta, silu, (bf16_t*)out, ldo, (const char*)nullptr, 0LL);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

#ifndef T2V_HOSTSIM
// per-device state of the one-launch GroupNorm: 2 words per unit (arrival count, generation) + an error word, zeroed once
static unsigned* gn_coop_sync_words(int* n_cu) {
    static unsigned* words[64] = {nullptr};
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!words[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, (2 * 256 + 64) * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, (2 * 256 + 64) * sizeof(unsigned)) != hipSuccess) return nullptr;
        words[dev] = (unsigned*)p;
        cus[dev] = prop.multiProcessorCount;
    }
    *n_cu = cus[dev];
    return words[dev];
}
extern "C" int t2v_gn_coop_error(void) {  // 1 if a one-launch GroupNorm ever gave up waiting at its barrier on this device
    int n_cu = 0;
    unsigned* w = gn_coop_sync_words(&n_cu);
    unsigned v = 0;
    if (!w || hipMemcpy(&v, w + 2 * 256, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}
static int g_gn_coop = -1;
extern "C" int t2v_gn_coop_enable(int on) { g_gn_coop = on ? 1 : 0; return T2V_OK; }

// true (and launched) when the tensor fits the one-launch form
static bool gn_try_coop(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units, int rows_per_unit, int groups,
                        float eps, const float* gamma, const float* beta, int silu, float* ws, void* out, int ldo, hipStream_t s) {
    // OFF by default: measured on MI355X (profiles/r02_gn_one_launch_vs_three.csv) the one-launch form is SLOWER than the three
    // launches on every UNet shape (3.25-3.70 ms vs 2.67 ms per step in-graph): the tensors are L2 / Infinity-Cache resident
    // when GroupNorm runs, so the second read the three-launch form pays is cheap, while the device-scope hand-off (atomic
    // arrival + fabric-latency polls and partial loads) costs 8-14 us per call however small the tensor.
    if (g_gn_coop < 0) g_gn_coop = getenv("T2V_GN_COOP") ? atoi(getenv("T2V_GN_COOP")) : 0;
    if (!g_gn_coop) return false;
    const int C = c0 + c1, cpr = C / 8;
    if (cpr > GC_NT || cpr < 1 || n_units > 256 || 2 * groups > GC_NT) return false;
    int n_cu = 0;
    unsigned* sync_words = gn_coop_sync_words(&n_cu);
    if (!sync_words || n_cu < n_units) return false;
    const int ty = GC_NT / cpr;
    int w = n_cu / n_units;                                   // one workgroup per CU: all resident
    const int by_rows = (rows_per_unit + ty - 1) / ty;        // a workgroup wants at least one row per thread row
    if (w > by_rows) w = by_rows;
    const int ws_cap = (rows_per_unit + GN_RPT - 1) / GN_RPT;  // what t2v_gn_ws_floats sized the partial area for
    if (w > ws_cap) w = ws_cap;
    if (w > GN_MAX_SLABS) w = GN_MAX_SLABS;
    if (w < 1) w = 1;
    const int rows_per_wg = (rows_per_unit + w - 1) / w;
    w = (rows_per_unit + rows_per_wg - 1) / rows_per_wg;
    const int need = (rows_per_wg + ty - 1) / ty;             // chunks a thread keeps in registers
    if (need > 16) return false;
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    const size_t lds = (size_t)2 * ty * C * sizeof(float);
#define T2V_GC_LAUNCH(MAXC)                                                                                                       \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (!attr) { hipFuncSetAttribute((const void*)gn_coop_kernel<MAXC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; } \
        hipLaunchKernelGGL(gn_coop_kernel<MAXC>, dim3(n_units * w), dim3(GC_NT), lds, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, \
                           c1, ld1, rows_per_unit, groups, w, rows_per_wg, inv_count, eps, gamma, beta, silu, ws, sync_words,      \
                           (bf16_t*)out, ldo);                                                                                    \
    } while (0)
    if (need <= 2) T2V_GC_LAUNCH(2);
    else if (need <= 4) T2V_GC_LAUNCH(4);
    else if (need <= 8) T2V_GC_LAUNCH(8);
    else T2V_GC_LAUNCH(16);
#undef T2V_GC_LAUNCH
    return true;
}
#else  // host simulator: workgroups run one after the other, a spin barrier between them cannot be simulated
extern "C" int t2v_gn_coop_error(void) { return 0; }
extern "C" int t2v_gn_coop_enable(int) { return T2V_OK; }
#endif

// ---- GroupNorm(+SiLU) of a SMALL unit in one launch, one workgroup per (group, unit) --------------------------------------------------
// The norms that cannot take their statistics from a producer (split-K launches, the 5 x 8 level: 49 per UNet step, 1.6-6.5 MB each)
// paid two latency-bound launches.  A group of a unit is independent of every other one: block (group, unit) holds its
// rows_per_unit x cpg elements in registers (16-byte chunks, at most GG_MAXCH per thread, all loads issued first), reduces (sum, sumsq)
// — per-thread floats over <= 32 elements, then doubles by wave butterfly and one LDS hand-off, fixed order — and normalises its
// registers straight into the output: x is read once, no workspace, no second launch.  A group's row segment is cpg x 2 bytes
// (80 / 160 bytes at C = 1 280 / 2 560): neighbouring groups share lines in L2.
namespace {
constexpr int GG_MAXCH = 4;
template <int NT>
__global__ __launch_bounds__(NT) void gn_group_kernel(const bf16_t* __restrict__ x0, int c0, int ld0, const bf16_t* __restrict__ x1, int c1,
                                                      int ld1, int rows_per_unit, int groups, float inv_count, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                      bf16_t* __restrict__ out, int ldo) {
    __shared__ double shs[NT / 64], shq[NT / 64];
    __shared__ float smr[2];
    const int C = c0 + c1, cpg = C / groups, cpr = cpg >> 3;
    const int unit = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x;
    const int nchunk = rows_per_unit * cpr, ch0 = grp * cpg;
    const long long row0 = (long long)unit * rows_per_unit;
    uint4 u[GG_MAXCH];
    int rr[GG_MAXCH], cc[GG_MAXCH];
#pragma unroll
    for (int j = 0; j < GG_MAXCH; ++j) {
        const int idx = tid + j * NT;
        rr[j] = idx / cpr;
        cc[j] = ch0 + (idx - rr[j] * cpr) * 8;
        u[j] = make_uint4(0u, 0u, 0u, 0u);
        if (idx < nchunk) u[j] = *(const uint4*)gn_src(x0, c0, ld0, x1, ld1, row0 + rr[j], cc[j]);
    }
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < GG_MAXCH; ++j) {   // (chunks past the end are zeros: they add nothing)
        float f[8];
        unpack8(u[j], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a += f[e]; q = fmaf(f[e], f[e], q); }
    }
    double ds = (double)a, dq = (double)q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
    if ((tid & 63) == 0) { shs[tid >> 6] = ds; shq[tid >> 6] = dq; }
    __syncthreads();
    if (tid == 0) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { ts += shs[w]; tq += shq[w]; }
        const double mean = ts * (double)inv_count;
        double var = tq * (double)inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        smr[0] = (float)mean;
        smr[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = smr[0], rstd = smr[1];
#pragma unroll
    for (int j = 0; j < GG_MAXCH; ++j) {
        if (tid + j * NT >= nchunk) break;
        const float4 g0 = *(const float4*)(gamma + cc[j]), g1 = *(const float4*)(gamma + cc[j] + 4);
        const float4 b0 = *(const float4*)(beta + cc[j]), b1 = *(const float4*)(beta + cc[j] + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float f[8];
        unpack8(u[j], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = rstd * gm[e];
            const float v = f[e] * sc + (bt[e] - mean * sc);
            f[e] = silu ? silu_f(v) : v;
        }
        *(uint4*)(out + (row0 + rr[j]) * ldo + cc[j]) = pack8(f);
    }
}
// threads per block of that form, 0 = not eligible (channels per group not a multiple of 8, a group that straddles the two parts of a
// concat is fine — chunks never do, c0 % 8 == 0 —, more than GG_MAXCH chunks per thread at 1 024 threads, a prefetch hint to carry)
int gn_group_threads(int c0, int c1, int rows_per_unit, int groups, long long prefetch_bytes) {
    static const bool on = !(getenv("T2V_GN_GROUP") && getenv("T2V_GN_GROUP")[0] == '0');
    const int cpg = (c0 + c1) / groups;
    if (!on || prefetch_bytes > 0 || cpg % 8 || c0 % 8) return 0;
    const long long nchunk = (long long)rows_per_unit * (cpg >> 3);
    return nchunk <= 256 * GG_MAXCH ? 256 : nchunk <= 1024 * GG_MAXCH ? 1024 : 0;
}
}  // namespace

// GroupNorm(+SiLU) in one call: slab partial sums, then either [finish + per-channel affine, streaming apply] or, for
// tensors with few slabs, an apply pass whose blocks finish the statistics themselves (2 launches instead of 3).
extern "C" int t2v_group_norm(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                              int rows_per_unit, int groups, float eps, const float* gamma, const float* beta, int silu,
                              float* ws, void* out, int ldo, const void* prefetch, long long prefetch_bytes, void* stream) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups);
    if (rc) return rc;
    T2V_REQUIRE(ws && gamma && beta && out && ldo % 8 == 0, T2V_EINVAL, "t2v_group_norm: bad argument");
    if (!x1) { c1 = 0; ld1 = 0; }
    hipStream_t s = (hipStream_t)stream;
    if (const int nt = gn_group_threads(c0, c1, rows_per_unit, groups, prefetch_bytes)) {
        const float inv = 1.0f / ((float)rows_per_unit * (float)((c0 + c1) / groups));
        if (nt == 1024)
            hipLaunchKernelGGL(gn_group_kernel<1024>, dim3(groups, n_units), dim3(1024), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1,
                               rows_per_unit, groups, inv, eps, gamma, beta, silu, (bf16_t*)out, ldo);
        else
            hipLaunchKernelGGL(gn_group_kernel<256>, dim3(groups, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1,
                               rows_per_unit, groups, inv, eps, gamma, beta, silu, (bf16_t*)out, ldo);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
#ifndef T2V_HOSTSIM
    if (gn_try_coop(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups, eps, gamma, beta, silu, ws, out, ldo, s)) {
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
#endif
    const int C = c0 + c1, nslab = gn_nslab(C, rows_per_unit, groups), slab_rows = gn_slab_rows(C, rows_per_unit, groups);
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    // opt-in: measured neutral on MI355X (2.70 vs 2.67 ms of GroupNorm per UNet step in-graph, 23.69 vs 23.71 ms per step:
    // the coarser statistics pass of the 40960-row tensors costs what the dropped gn_final launches saved)
    static const int two_launch = getenv("T2V_GN_TWO_LAUNCH") ? atoi(getenv("T2V_GN_TWO_LAUNCH")) : 0;
    if (two_launch && nslab > gn_fuse_slabs(groups)) {
        // Many slabs (the 40960- and 10240-row tensors): the statistics pass walks COARSER slabs — at most gn_fuse_slabs of
        // them per unit, so that every block of the (fine-grained) apply pass can finish the statistics itself — instead of
        // paying a third, latency-bound launch (gn_final: 6.6 us + a kernel boundary, 77 times per UNet step).
        const GnGeom gg = gn_geom(C);
        const int cap = gn_fuse_slabs(groups);
        int stat_rows = (rows_per_unit + cap - 1) / cap;
        stat_rows = (stat_rows + gg.ty * GN_RPT - 1) / (gg.ty * GN_RPT) * (gg.ty * GN_RPT);
        const int stat_nslab = (rows_per_unit + stat_rows - 1) / stat_rows;
        hipLaunchKernelGGL(gn_partial_kernel, dim3(stat_nslab, n_units), dim3(256), (size_t)2 * gg.ty * C * sizeof(float), s,
                           (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1, rows_per_unit, groups, stat_rows, ws);
        T2V_CHECK_LAUNCH();
        hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1,
                           ld1, rows_per_unit, groups, slab_rows, (const float*)nullptr, (const float*)nullptr, (const float*)ws, stat_nslab,
                           inv_count, eps, gamma, beta, silu, (bf16_t*)out, ldo, (const char*)prefetch, prefetch_bytes >> 7);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
    rc = gn_launch_partial(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups, ws, s);
    if (rc) return rc;
    if (nslab <= gn_fuse_slabs(groups)) {
        hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1,
                           ld1, rows_per_unit, groups, slab_rows, (const float*)nullptr, (const float*)nullptr, (const float*)ws, nslab,
                           inv_count, eps, gamma, beta, silu, (bf16_t*)out, ldo, (const char*)prefetch, prefetch_bytes >> 7);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
    float* coef = ws + t2v_gn_ws_floats(n_units, rows_per_unit, groups);
    hipLaunchKernelGGL(gn_final_kernel, dim3(n_units), dim3(1024), 0, s, (const float*)ws, nslab, groups, C, inv_count, eps, gamma, beta,
                       (float*)nullptr, coef);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1, ld1,
                       rows_per_unit, groups, slab_rows, (const float*)nullptr, (const float*)coef, (const float*)nullptr, 0, 0.f, 0.f,
                       gamma, beta, silu, (bf16_t*)out, ldo, (const char*)prefetch, prefetch_bytes >> 7);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// GroupNorm(+SiLU) on the column statistics of the producing GEMMs: [statistics from cs: gn_partial_cs_kernel] + [apply pass whose
// blocks finish the unit's statistics themselves].  Two launches, the tensor is read once.
constexpr int GN_CS_BLOCKS = 80;  // statistics blocks per unit (<= gn_fuse_slabs: the apply pass finishes them per block)
// threads per block of the direct form for this shape, 0 = not eligible: odd channels per group or part width, too many slabs per thread,
// or a group's 16 ncp bytes per slab are less than a 128-byte line while the statistics are too large to sit in the caches — the VAE
// decoder's 256- and 128-channel levels have 42 / 84 MB of them per norm, read in 64- / 32-byte pieces 1-2 KB apart (measured: the
// decode 0.2 ms slower, profiles/r05_gn_direct_ab.txt); there the partial-sums form reads whole rows of the array.
static int gn_csd_threads(int c0, int c1, int groups, int slabs_per_unit, int n_units, const float* cs0, const float* cs1) {
    static const bool direct_on = !(getenv("T2V_GN_CS_DIRECT") && getenv("T2V_GN_CS_DIRECT")[0] == '0');
    const int C = c0 + c1, cpg = C / groups, ncp = cpg >> 1;
    if (!direct_on || cpg % 2 || c0 % 2 || ncp < 1 || ncp > 256 || (uintptr_t)cs0 % 16 || (cs1 && (uintptr_t)cs1 % 16)) return 0;
    if (ncp < 8 && (long long)n_units * slabs_per_unit * C * 8 > (8LL << 20)) return 0;
    const int lanes256 = 256 / ncp;
    const int nt = (slabs_per_unit + lanes256 - 1) / lanes256 > 8 ? 1024 : 256;
    return (slabs_per_unit + nt / ncp - 1) / (nt / ncp) <= GN_CSD_LOADS ? nt : 0;
}
static void gn_csd_launch(int nt, const float* cs0, int c0, const float* cs1, int c1, int n_units, int slabs_per_unit, int groups, float inv_count,
                          float eps, const float* gamma, const float* beta, float* coef, float* stats, hipStream_t s) {
    if (nt == 1024)
        hipLaunchKernelGGL(gn_coef_cs_kernel<1024>, dim3(groups, n_units), dim3(1024), 0, s, cs0, c0, cs1, c1, slabs_per_unit, groups,
                           inv_count, eps, gamma, beta, coef, stats);
    else
        hipLaunchKernelGGL(gn_coef_cs_kernel<256>, dim3(groups, n_units), dim3(256), 0, s, cs0, c0, cs1, c1, slabs_per_unit, groups,
                           inv_count, eps, gamma, beta, coef, stats);
}
extern "C" long long t2v_group_norm_cs_ws_floats(int n_units, int rows_per_unit, int groups) {
    (void)rows_per_unit;
    return (long long)n_units * GN_CS_BLOCKS * 2 * groups;
}
extern "C" int t2v_group_norm_cs(const float* cs0, const float* cs1, const void* x0, int c0, int ld0, const void* x1, int c1, int ld1,
                                 int n_units, int rows_per_unit, int groups, float eps, const float* gamma, const float* beta, int silu,
                                 float* ws, void* out, int ldo, const void* prefetch, long long prefetch_bytes, void* stream) {
    int rc = gn_check(x0, c0, ld0, x1, c1, ld1, n_units, rows_per_unit, groups);
    if (rc) return rc;
    T2V_REQUIRE(ws && gamma && beta && out && ldo % 8 == 0 && cs0, T2V_EINVAL, "t2v_group_norm_cs: bad argument");
    if (!x1) { c1 = 0; ld1 = 0; }
    T2V_REQUIRE(rows_per_unit % 32 == 0 && (c1 == 0 || cs1) && GN_CS_BLOCKS <= gn_fuse_slabs(groups), T2V_ESHAPE,
                "t2v_group_norm_cs: rows_per_unit must be a multiple of the 32-row statistics slab");
    T2V_REQUIRE((uintptr_t)cs0 % 8 == 0 && (!cs1 || (uintptr_t)cs1 % 8 == 0), T2V_ESHAPE, "t2v_group_norm_cs: unaligned statistics");
    hipStream_t s = (hipStream_t)stream;
    const int C = c0 + c1, nslab = gn_nslab(C, rows_per_unit, groups), slab_rows = gn_slab_rows(C, rows_per_unit, groups);
    const int slabs_per_unit = rows_per_unit / 32;
    const int slabs_per_blk = (slabs_per_unit + GN_CS_BLOCKS - 1) / GN_CS_BLOCKS;
    const int nblk = (slabs_per_unit + slabs_per_blk - 1) / slabs_per_blk;
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    // Direct form (round 5): one block per (group, unit) turns the column statistics into the per-channel affine, the apply pass
    // streams with it.  Needs an even number of channels per group with both parts' widths even (a float4 = two channels of one
    // part), at most GN_CSD_LOADS slabs per thread, and the coefficients in the workspace (2 C <= 2 * groups * GN_CS_BLOCKS floats per unit).
    const int nt = C <= groups * GN_CS_BLOCKS ? gn_csd_threads(c0, c1, groups, slabs_per_unit, n_units, cs0, cs1) : 0;
    if (nt) {
        gn_csd_launch(nt, cs0, c0, cs1, c1, n_units, slabs_per_unit, groups, inv_count, eps, gamma, beta, ws, nullptr, s);
        T2V_CHECK_LAUNCH();
        hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1,
                           ld1, rows_per_unit, groups, slab_rows, (const float*)nullptr, (const float*)ws, (const float*)nullptr, 0,
                           inv_count, eps, gamma, beta, silu, (bf16_t*)out, ldo, (const char*)prefetch, prefetch_bytes >> 7);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
    hipLaunchKernelGGL(gn_partial_cs_kernel, dim3(nblk, n_units), dim3(256), (size_t)2 * C * sizeof(float), s, cs0, c0, cs1, c1,
                       slabs_per_unit, slabs_per_blk, groups, ws);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(nslab, n_units), dim3(256), 0, s, (const bf16_t*)x0, c0, ld0, (const bf16_t*)x1, c1,
                       ld1, rows_per_unit, groups, slab_rows, (const float*)nullptr, (const float*)nullptr, (const float*)ws, nblk,
                       inv_count, eps, gamma, beta, silu, (bf16_t*)out, ldo, (const char*)prefetch, prefetch_bytes >> 7);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// The per-channel affine of a GroupNorm by itself — coef[unit][0][c] = rstd gamma[c], coef[unit][1][c] = beta[c] - mean rstd gamma[c] — from
// the producers' column statistics (the statistics launch of t2v_group_norm_cs's direct form, alone): for a consumer that applies the
// norm in its own load phase (t2v_linear_pr with t2v_gemm_desc::gn_coef: the transformers' proj_in).  _supported: 1 where the direct
// form takes the shape (even channels per group and part widths, <= GN_CSD_LOADS slabs per thread), else 0: use t2v_group_norm_cs.
extern "C" int t2v_gn_coef_cs_supported(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups) {
    if (!cs0 || n_units <= 0 || rows_per_unit <= 0 || rows_per_unit % 32 || groups <= 0 || c0 <= 0) return 0;
    if (!cs1) c1 = 0;
    if ((c0 + c1) % groups) return 0;
    return gn_csd_threads(c0, c1, groups, rows_per_unit / 32, n_units, cs0, cs1) ? 1 : 0;
}
extern "C" int t2v_gn_coef_cs(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups, float eps,
                              const float* gamma, const float* beta, float* coef, void* stream) {
    T2V_REQUIRE(cs0 && gamma && beta && coef && n_units > 0 && rows_per_unit > 0 && groups > 0 && groups <= 128 && c0 > 0, T2V_EINVAL,
                "t2v_gn_coef_cs: bad argument");
    if (!cs1) c1 = 0;
    const int C = c0 + c1;
    T2V_REQUIRE(rows_per_unit % 32 == 0 && C % groups == 0, T2V_ESHAPE, "t2v_gn_coef_cs: rows_per_unit a multiple of 32, channels of the groups");
    const int slabs_per_unit = rows_per_unit / 32;
    const int nt = gn_csd_threads(c0, c1, groups, slabs_per_unit, n_units, cs0, cs1);
    T2V_REQUIRE(nt, T2V_ESHAPE, "t2v_gn_coef_cs: the direct statistics form does not take this shape (ask t2v_gn_coef_cs_supported)");
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    gn_csd_launch(nt, cs0, c0, cs1, c1, n_units, slabs_per_unit, groups, inv_count, eps, gamma, beta, coef, nullptr, (hipStream_t)stream);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// (mean, rstd) per (unit, group) from the column statistics of the producing GEMMs — t2v_gn_stats without reading the tensor
// (the training engine keeps the statistics for the backward and normalises with t2v_gn_apply).  ws: t2v_group_norm_cs_ws_floats.
extern "C" int t2v_gn_stats_cs(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups,
                               float eps, float* ws, float* stats, void* stream) {
    T2V_REQUIRE(cs0 && ws && stats && n_units > 0 && rows_per_unit > 0 && groups > 0 && groups <= 128 && c0 > 0, T2V_EINVAL,
                "t2v_gn_stats_cs: bad argument");
    if (!cs1) c1 = 0;
    const int C = c0 + c1;
    T2V_REQUIRE(rows_per_unit % 32 == 0 && C % groups == 0 && C <= 4096, T2V_ESHAPE,
                "t2v_gn_stats_cs: rows_per_unit must be a multiple of the 32-row statistics slab, channels a multiple of the groups");
    T2V_REQUIRE((uintptr_t)cs0 % 8 == 0 && (!cs1 || (uintptr_t)cs1 % 8 == 0), T2V_ESHAPE, "t2v_gn_stats_cs: unaligned statistics");
    hipStream_t s = (hipStream_t)stream;
    const int slabs_per_unit = rows_per_unit / 32;
    const int slabs_per_blk = (slabs_per_unit + GN_CS_BLOCKS - 1) / GN_CS_BLOCKS;
    const int nblk = (slabs_per_unit + slabs_per_blk - 1) / slabs_per_blk;
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / groups));
    if (const int nt = gn_csd_threads(c0, c1, groups, slabs_per_unit, n_units, cs0, cs1)) {   // one launch: a block per (group, unit)
        gn_csd_launch(nt, cs0, c0, cs1, c1, n_units, slabs_per_unit, groups, inv_count, eps, nullptr, nullptr, nullptr, stats, s);
        T2V_CHECK_LAUNCH();
        return T2V_OK;
    }
    hipLaunchKernelGGL(gn_partial_cs_kernel, dim3(nblk, n_units), dim3(256), (size_t)2 * C * sizeof(float), s, cs0, c0, cs1, c1,
                       slabs_per_unit, slabs_per_blk, groups, ws);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_final_kernel, dim3(n_units), dim3(1024), 0, s, (const float*)ws, nblk, groups, C, inv_count, eps,
                       (const float*)nullptr, (const float*)nullptr, stats, (float*)nullptr);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_layernorm(const void* x, int ldx, int M, int C, const float* gamma, const float* beta, float eps,
                             void* out, int ldo, void* stream) {
    T2V_REQUIRE(x && gamma && beta && out && M > 0, T2V_EINVAL, "t2v_layernorm: bad argument");
    T2V_REQUIRE(C % 8 == 0 && C <= 64 * 8 * LN_MAX && ldx % 8 == 0 && ldo % 8 == 0, T2V_ESHAPE, "t2v_layernorm: unsupported C");
    // row-group form where C / 8 chunks split evenly over 8 / 16 / 32 / 64 lanes with at most 5 chunks per lane (the UNet's 320 /
    // 640 / 1280 and every power-of-two width from 64); T2V_LN_ROWGROUP=0: the one-row-per-wave kernels (round 1-4) for A/B
    static const bool rowgroup = !(getenv("T2V_LN_ROWGROUP") && getenv("T2V_LN_ROWGROUP")[0] == '0');
    const int cpr = C / 8;
    if (rowgroup) {
#define T2V_LNR_LAUNCH(LPR, CPL, RI)                                                                                                \
    do {                                                                                                                            \
        constexpr int rows_blk = 4 * (64 / LPR) * RI;                                                                               \
        hipLaunchKernelGGL((layernorm_rows_kernel<LPR, CPL, RI>), dim3((M + rows_blk - 1) / rows_blk), dim3(256), 0,               \
                           (hipStream_t)stream, (const bf16_t*)x, ldx, M, C, gamma, beta, eps, (bf16_t*)out, ldo);                  \
        T2V_CHECK_LAUNCH();                                                                                                         \
        return T2V_OK;                                                                                                              \
    } while (0)
        // two iterations per wave while that still leaves >= 2 workgroups per CU
#define T2V_LNR_PICK(LPR, CPL)                                                        \
    do {                                                                              \
        if ((long long)M >= 512LL * 8 * (64 / LPR)) T2V_LNR_LAUNCH(LPR, CPL, 2);      \
        else T2V_LNR_LAUNCH(LPR, CPL, 1);                                             \
    } while (0)
        for (int lpr = 8; lpr <= 64; lpr <<= 1) {
            if (cpr % lpr || cpr / lpr > 5) continue;
            const int cpl = cpr / lpr;
#define T2V_LNR_CASE(LPR)                                  \
    if (lpr == LPR) {                                      \
        if (cpl == 1) T2V_LNR_PICK(LPR, 1);                \
        else if (cpl == 2) T2V_LNR_PICK(LPR, 2);           \
        else if (cpl == 3) T2V_LNR_PICK(LPR, 3);           \
        else if (cpl == 4) T2V_LNR_PICK(LPR, 4);           \
        else T2V_LNR_PICK(LPR, 5);                         \
    }
            T2V_LNR_CASE(8) T2V_LNR_CASE(16) T2V_LNR_CASE(32) T2V_LNR_CASE(64)
#undef T2V_LNR_CASE
        }
#undef T2V_LNR_PICK
#undef T2V_LNR_LAUNCH
    }
    const int nj = (cpr + 63) / 64;
#define T2V_LN_LAUNCH(NJ, ROWS)                                                                                              \
    hipLaunchKernelGGL((layernorm_kernel<NJ, ROWS>), dim3((M + 4 * ROWS - 1) / (4 * ROWS)), dim3(256), 0, (hipStream_t)stream, \
                       (const bf16_t*)x, ldx, M, C, gamma, beta, eps, (bf16_t*)out, ldo)
    if (nj == 1) T2V_LN_LAUNCH(1, 4);
    else if (nj == 2) T2V_LN_LAUNCH(2, 4);
    else if (nj == 3) T2V_LN_LAUNCH(3, 2);
    else if (nj == 4) T2V_LN_LAUNCH(4, 2);
    else T2V_LN_LAUNCH(8, 1);
#undef T2V_LN_LAUNCH
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
