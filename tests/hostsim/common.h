// Host-simulator replacement of csrc/common.h (found first on the include path): the same helpers in plain C++.
// TEST INFRASTRUCTURE ONLY.  The bf16 conversions are the library's own code (round-to-nearest-even); pack2bf is the
// software equivalent of v_cvt_pk_bf16_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "t2v_hip.h"

typedef uint16_t bf16_t;
typedef hostsim::v8u16 bf16x8_t;   // 8 bf16 as raw bits (g++ 11 has no __bf16): the kernels only move them and feed the MFMA
typedef hostsim::v16f f32x16_t;
typedef float f32x4_t __attribute__((vector_size(16)));
#define T2V_WAVE 64

inline float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
inline bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
inline void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
inline uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}
inline float silu_f(float x) { return x * (1.0f / (1.0f + expf(-x))); }
inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
inline float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
inline float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-contiguous block renumbering (csrc/common.h): any bijection is correct; the simulator runs the blocks one after the other
inline int xcd_contiguous(int bid, int nblocks) {
    const int xcd = bid & 7, pos = bid >> 3, q = nblocks >> 3, r = nblocks & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}
inline void xcd_block3(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int id = xcd_contiguous(((int)blockIdx.z * gy + (int)blockIdx.y) * gx + (int)blockIdx.x, gx * gy * gz);
    bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
}

// counter-based dropout bits (csrc/common.h)
inline uint64_t splitmix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
constexpr uint64_t kDropQuadMul = 0xD1B54A32D192ED03ull;
inline uint64_t dropout_key(uint64_t seed, uint32_t site) { return seed + (uint64_t)site * 0x9E3779B97F4A7C15ull; }
inline uint64_t dropout_quad(uint64_t key, uint64_t quad) { return splitmix64(key + quad * kDropQuadMul); }
inline bool dropout_keep16(uint64_t word, int e, uint32_t thr16) {
    const uint32_t half = (e & 2) ? (uint32_t)(word >> 32) : (uint32_t)word;
    return ((e & 1) ? (half >> 16) : (half & 0xffffu)) >= thr16;
}
template <int NQ>
inline uint32_t dropout_keep_mask(uint64_t key, uint64_t quad0, uint32_t thr16) {
    const uint64_t z0 = key + quad0 * kDropQuadMul;
    uint32_t m = 0;
    for (int k = 0; k < NQ; ++k) {
        const uint64_t w = splitmix64(z0 + (uint64_t)k * kDropQuadMul);
        for (int e = 0; e < 4; ++e) m |= dropout_keep16(w, e, thr16) ? (1u << (4 * k + e)) : 0u;
    }
    return m;
}

#ifdef HOSTSIM_FULL  // the whole library: csrc/elementwise.hip brings its own definitions
void t2v_set_error(const char* msg);
const void* t2v_zero_page();
#else
inline std::string& t2v_err() { static std::string e; return e; }
inline void t2v_set_error(const char* msg) { t2v_err() = msg; }
inline const void* t2v_zero_page() { static char z[4096] = {0}; return z; }
#endif
#define T2V_CHECK_LAUNCH() do { } while (0)
#define T2V_REQUIRE(cond, code, msg) \
    do {                             \
        if (!(cond)) {               \
            t2v_set_error(msg);      \
            return code;             \
        }                            \
    } while (0)
