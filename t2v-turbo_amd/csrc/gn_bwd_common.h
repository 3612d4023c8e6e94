// Pieces of the GroupNorm backward shared by backward.hip (validated single-tensor path) and backward_unet.hip (two-part / wide).
#pragma once
#include "common.h"

namespace {

constexpr int GB_RPT = 4;  // rows in flight per thread (two input streams)

// per-channel constants of one 8-channel chunk
struct GbChan { float mu[8], rs[8], ga[8], be[8]; };
__device__ __forceinline__ void gb_load_chan(GbChan& k, const float* stats, const float* gamma, const float* beta, int unit,
                                             int groups, int cpg, int c0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e, grp = c / cpg;
        const float* st = stats + ((long long)unit * groups + grp) * 2;
        k.mu[e] = st[0]; k.rs[e] = st[1]; k.ga[e] = gamma[c]; k.be[e] = beta[c];
    }
}
// g = dL/d(xhat) of one element: dy * act'(xhat*gamma+beta) * gamma ; xh = normalised input
__device__ __forceinline__ void gb_elem(const GbChan& k, int e, float x, float dy, int silu, float& xh, float& g) {
    xh = (x - k.mu[e]) * k.rs[e];
    float d = dy;
    if (silu) {
        const float u = xh * k.ga[e] + k.be[e];
        const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        d *= sig * (1.0f + u * (1.0f - sig));
    }
    g = d * k.ga[e];
}

// bstats[unit][group] = (mean g, mean g*xhat) over the group's rows x channels.  1024 threads: thread = (part, value) walks its
// share of the slabs with eight coalesced loads in flight (a 256-thread block walking up to 256 slabs one dependent load at a time
// took 22 us per call, 166 calls per student step), then the parts are added in a fixed order.
constexpr int GB_FINAL_THREADS = 1024;
__global__ __launch_bounds__(GB_FINAL_THREADS) void gn_bwd_final_kernel(const float* partial, int nslab, int groups, float inv_count, float* bstats) {
    __shared__ double sh[GB_FINAL_THREADS];
    const int unit = blockIdx.x, tid = threadIdx.x;
    const int width = groups * 2, parts = GB_FINAL_THREADS / width;
    const int v = tid % width, part = tid / width;
    const float* base = partial + (long long)unit * nslab * width + v;
    const int chunk = (nslab + parts - 1) / parts;
    double acc = 0.0;
    if (part < parts) {
        const int k1 = min(nslab, (part + 1) * chunk);
        int k = part * chunk;
        for (; k + 8 <= k1; k += 8) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = base[(long long)(k + e) * width];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (double)t[e];
        }
        for (; k < k1; ++k) acc += (double)base[(long long)k * width];
    }
    sh[tid] = acc;
    __syncthreads();
    if (tid < width) {
        double t = 0.0;
        for (int pz = 0; pz < parts; ++pz) t += sh[pz * width + tid];
        bstats[(long long)unit * width + tid] = (float)(t * inv_count);
    }
}


}  // namespace
