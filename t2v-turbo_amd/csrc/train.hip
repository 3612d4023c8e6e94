// Kernels of the LoRA training path (weight gradients of the student UNet).  Own translation unit: what hipcc emits
// for a kernel depends on its neighbours, and everything else in the library is hardware-validated.
#include "common.h"

// out[i] = idx[i] >= 0 ? src[idx[i]] * alpha : 0      (accumulate = 0; out fp32 or bf16)
// out[i] += src[idx[i]] * alpha  where idx[i] >= 0     (accumulate = 1; untouched elsewhere)
// One launch re-lays EVERY LoRA tensor into its kernel packs (fp32 flat parameters -> bf16 [N][K] operands in four
// layouts, zero-padded to rank 64), and one launch carries every weight gradient from the GEMM output layout
// ([tap][r][cin]) back to the parameter layout ([r][cin][ky][kx]).  HBM-bound: 4 B index + 4 B source + 2..4 B out.
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, float alpha,
                                                      void* __restrict__ out, int out_bf16, int accumulate, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int j = idx[i];
        if (j < 0 && accumulate) continue;
        float v = j >= 0 ? src[j] * alpha : 0.f;
        if (out_bf16) {
            bf16_t* o = (bf16_t*)out;
            if (accumulate) v += bf2f(o[i]);
            o[i] = f2bf(v);
        } else {
            float* o = (float*)out;
            if (accumulate) v += o[i];
            o[i] = v;
        }
    }
}

extern "C" int t2v_gather_f32(const float* src, const int* idx, float alpha, void* out, int dt_out, int accumulate, long long n,
                              void* stream) {
    T2V_REQUIRE(src && idx && out && n > 0, T2V_EINVAL, "t2v_gather_f32: null pointer / empty");
    T2V_REQUIRE(dt_out == T2V_F32 || dt_out == T2V_BF16, T2V_EINVAL, "t2v_gather_f32: out dtype must be fp32 or bf16");
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride: 32 workgroups per CU
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, idx, alpha, out,
                       dt_out == T2V_BF16 ? 1 : 0, accumulate ? 1 : 0, n);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
