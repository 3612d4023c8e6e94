// Internal helpers shared by the gfx950 kernels of libt2v_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "t2v_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

#define T2V_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (matches torch's .to(bfloat16)); NaN kept quiet
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2, round-to-nearest-even in ONE instruction (gfx950 v_cvt_pk_bf16_f32; there is
// no clang builtin for it, hence the asm)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}
// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroups are dealt to the 8 XCDs round-robin by their linear id (x fastest), and each XCD has its own L2: blocks that share an
// operand (the query tiles of one attention head, the output tiles of one token range) should have CONSECUTIVE ids on ONE XCD.
// xcd_contiguous renumbers the linear ids bijectively that way; xcd_block3 applies it to a 3-D grid.
__device__ __forceinline__ int xcd_contiguous(int bid, int nblocks) {
    const int xcd = bid & 7, pos = bid >> 3, q = nblocks >> 3, r = nblocks & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}
__device__ __forceinline__ void xcd_block3(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int id = xcd_contiguous(((int)blockIdx.z * gy + (int)blockIdx.y) * gx + (int)blockIdx.x, gx * gy * gz);
    bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
}

// counter-based dropout bits: one splitmix64 finaliser per QUAD of adjacent elements of the row-major [rows][ncols] matrix (flat
// index i = row * ncols + col: word = quad i >> 2, 16 bits per element, element i & 3 = bits [16 (i & 3), +16)), compared with
// thr16 = (p * 2^32) >> 16.  Shared by t2v_dropout_bf16 (train.hip) and the dropout / LoRA epilogues of t2v_gemm.  (Rounds 1-3 drew
// 32 bits per element, one word per pair: the mask arithmetic — six quarter-rate 32-bit multiplies per word — was 2/3 of the LoRA
// epilogue's cost; p is resolved to 2^-16 now, 0.1 -> 6553 / 65536.)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
constexpr uint64_t kDropQuadMul = 0xD1B54A32D192ED03ull;
__device__ __forceinline__ uint64_t dropout_key(uint64_t seed, uint32_t site) { return seed + (uint64_t)site * 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ uint64_t dropout_quad(uint64_t key, uint64_t quad) { return splitmix64(key + quad * kDropQuadMul); }
__device__ __forceinline__ bool dropout_keep16(uint64_t word, int e, uint32_t thr16) {   // e in 0..3
    const uint32_t half = (e & 2) ? (uint32_t)(word >> 32) : (uint32_t)word;
    return ((e & 1) ? (half >> 16) : (half & 0xffffu)) >= thr16;
}
// keep bits of 4 NQ consecutive elements starting at flat index 4 quad0: bit e of the result = element e kept
template <int NQ>
__device__ __forceinline__ uint32_t dropout_keep_mask(uint64_t key, uint64_t quad0, uint32_t thr16) {
    const uint64_t z0 = key + quad0 * kDropQuadMul;   // one 64-bit multiply for the run; the quads after it are constant adds
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const uint64_t w = splitmix64(z0 + (uint64_t)k * kDropQuadMul);
#pragma unroll
        for (int e = 0; e < 4; ++e) m |= dropout_keep16(w, e, thr16) ? (1u << (4 * k + e)) : 0u;
    }
    return m;
}

// host side
void t2v_set_error(const char* msg);
const void* t2v_zero_page();  // >= 256 B of device zeros (lazily allocated per process)
#define T2V_CHECK_LAUNCH()                                              \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) {                                        \
            t2v_set_error(hipGetErrorString(e__));                      \
            return T2V_EHIP;                                            \
        }                                                               \
    } while (0)
#define T2V_REQUIRE(cond, code, msg)        \
    do {                                    \
        if (!(cond)) {                      \
            t2v_set_error(msg);             \
            return code;                    \
        }                                   \
    } while (0)
