// The GEGLU gate's activation, shared by the GEMM epilogues that carry it (gemm.hip, linear_pr.hip).
#pragma once
#include "common.h"

// exact (erf) GELU without transcendentals: x * Phi(x), Phi(x) - 1/2 = xc * P(xc^2) with xc = clamp(x, +-4.5) and P the
// degree-9 weighted least-squares fit on Chebyshev nodes (1 - Phi(4.5) = 3.4e-6).  |error| < 4.2e-5 absolute and < 9e-4
// relative wherever |gelu| > 0.01, below half a bf16 ulp of the output; 12 packable FMAs instead of 11 + v_rcp + v_exp
// (quarter rate), which matters because the GEGLU epilogue's VALU time is as long as the MFMA time of its K = 320..1280 loop.
// -DT2V_GELU_ERF keeps the Abramowitz-Stegun 7.1.26 form (|erf error| < 1.5e-7).
__device__ __forceinline__ float fast_gelu(float x) {
#ifdef T2V_GELU_ERF
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    const float erf_v = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.0f + erf_v);
#else
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
#endif
}

// The same function on a PAIR of values: v_pk_fma_f32 / v_pk_mul_f32 work on two fp32 lanes per instruction at the scalar issue
// rate, so the polynomial costs 7.5 VALU instructions per element instead of 14 (bit-identical: a packed FMA is an FMA per
// component).  The coefficients sit in scalar register pairs (a packed instruction cannot carry a 32-bit literal).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t fast_gelu2(f32x2_t x) {
#if defined(T2V_GELU_ERF) || defined(T2V_HOSTSIM)
    f32x2_t r;
    r[0] = fast_gelu(x[0]);
    r[1] = fast_gelu(x[1]);
    return r;
#else
    f32x2_t xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f);
    const f32x2_t t = xc * xc;
    f32x2_t q = (f32x2_t)(-1.684528927e-12f);
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(1.983680165e-10f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(-1.041734787e-08f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(3.247777158e-07f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(-6.776206646e-06f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(1.014731897e-04f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(-1.141749439e-03f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(9.890335612e-03f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(-6.642110646e-02f));
    q = __builtin_elementwise_fma(q, t, (f32x2_t)(3.989264667e-01f));
    return x * __builtin_elementwise_fma(xc, q, (f32x2_t)(0.5f));
#endif
}
