// Host SIMT simulator — TEST INFRASTRUCTURE ONLY (tests/hostsim/).  Stands in for <hip/hip_runtime.h> so that the library's
// kernel SOURCES that use only plain SIMT features (threadIdx / blockIdx, __shared__, __syncthreads, __shfl_xor, vector
// loads) compile with g++ and run on the CPU: every thread of a workgroup is a fiber (ucontext), workgroups run one after
// the other, __syncthreads and the 64-lane shuffles are rendezvous points between fibers.  No MFMA, no inline asm, no
// LDS-DMA: kernels using those (t2v_gemm, the attention forwards) are not simulated.  What it buys: the exact index
// arithmetic, reductions and LDS traffic of kernels that have not run on hardware yet are executed and compared with the
// emulated op backend in the CPU suite (tests/test_hostsim_kernels.py).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static  // workgroups run sequentially: one static instance is the workgroup's LDS

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

using std::max;
using std::min;
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#define __expf expf
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

namespace hostsim {

struct Sync { int count = 0; unsigned gen = 0; };
struct Fiber {
    ucontext_t ctx;
    bool done = false;
    dim3 tid;
    int flat = 0;
};
struct State {
    dim3 grid, block, bid;
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    std::vector<char> dyn;            // dynamic shared memory of the running workgroup
    Sync block_sync;
    std::vector<Sync> wave_sync;
    std::vector<uint64_t> slots;      // shuffle exchange, one per thread
    int cur = -1;
    ucontext_t sched;
    const std::function<void()>* body = nullptr;
};
inline State& st() { static State s; return s; }
constexpr size_t kStack = 128 * 1024;

inline int alive_in_block() { int n = 0; for (auto& f : st().fibers) n += !f.done; return n; }
inline int alive_in_wave(int w) {
    State& s = st();
    int n = 0;
    for (int i = w * 64; i < std::min<int>((w + 1) * 64, (int)s.fibers.size()); ++i) n += !s.fibers[i].done;
    return n;
}
inline void yield() { State& s = st(); swapcontext(&s.fibers[s.cur].ctx, &s.sched); }
inline void release_if_complete(Sync& sy, int alive) {
    if (sy.count > 0 && sy.count >= alive) { sy.count = 0; ++sy.gen; }
}
inline void arrive(Sync& sy, int alive) {
    ++sy.count;
    const unsigned g = sy.gen;
    if (sy.count >= alive) { sy.count = 0; ++sy.gen; return; }
    while (sy.gen == g) yield();
}
inline void trampoline() {
    State& s = st();
    (*s.body)();
    Fiber& f = s.fibers[s.cur];
    f.done = true;  // an exited thread no longer takes part in barriers / shuffles
    release_if_complete(s.block_sync, alive_in_block());
    release_if_complete(s.wave_sync[f.flat / 64], alive_in_wave(f.flat / 64));
    swapcontext(&f.ctx, &s.sched);
}
inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    State& s = st();
    const int n = (int)(block.x * block.y * block.z);
    s.grid = grid; s.block = block; s.body = &body;
    s.fibers.resize(n);
    s.stacks.resize((size_t)n * kStack);
    s.slots.assign(n, 0);
    s.dyn.assign(shmem + 64, 0);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.bid = dim3(bx, by, bz);
                s.block_sync = Sync();
                s.wave_sync.assign((n + 63) / 64, Sync());
                for (int i = 0; i < n; ++i) {
                    Fiber& f = s.fibers[i];
                    f.done = false;
                    f.flat = i;
                    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = s.stacks.data() + (size_t)i * kStack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &s.sched;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                int done = 0;
                long long rounds = 0;
                while (done < n) {
                    done = 0;
                    for (int i = 0; i < n; ++i) {
                        if (s.fibers[i].done) { ++done; continue; }
                        s.cur = i;
                        swapcontext(&s.sched, &s.fibers[i].ctx);
                        done += s.fibers[i].done;
                    }
                    if (++rounds > 2000000) { fprintf(stderr, "hostsim: workgroup does not terminate (barrier mismatch?)\n"); abort(); }
                }
            }
    s.cur = -1;
}
inline void* dyn_shared() { return st().dyn.data(); }
template <class T>
inline T shfl_xor(T v, int mask) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64, lane = me % 64;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    s.slots[me] = bits;
    arrive(s.wave_sync[w], alive_in_wave(w));
    const int partner = w * 64 + (lane ^ mask);
    T r = v;
    if (partner < (int)s.fibers.size()) memcpy(&r, &s.slots[partner], sizeof(T));
    arrive(s.wave_sync[w], alive_in_wave(w));
    return r;
}

inline float med3(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
// LDS-DMA (global_load_lds_dwordx{1,4}): every lane copies `size` bytes from ITS global pointer to the wave-uniform LDS base
// + lane * size (+ offset): the LDS image of one wave instruction is lane-linear.  Synchronous here (the s_waitcnt counters
// the kernels pair it with are no-ops on the host).
inline void glds(const void* gsrc, void* lds_wave_base, int size, int offset) {
    State& s = st();
    const int lane = s.fibers[s.cur].flat % 64;
    memcpy((char*)lds_wave_base + (size_t)lane * size + offset, gsrc, size);
}
inline void wave_rendezvous() {
    State& s = st();
    const int w = s.fibers[s.cur].flat / 64;
    arrive(s.wave_sync[w], alive_in_wave(w));
}
typedef uint16_t v8u16 __attribute__((vector_size(16)));
typedef float v16f __attribute__((vector_size(64)));
inline std::vector<v8u16>& mf_a() { static std::vector<v8u16> v; return v; }
inline std::vector<v8u16>& mf_b() { static std::vector<v8u16> v; return v; }
inline float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
// v_mfma_f32_32x32x16_bf16 as a wave collective: D = A (32 x 16) * B (16 x 32) + C.  Operand layout (CDNA4 ISA): lane l holds
// A[row = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][col = l & 31]; accumulator register r of lane l is
// D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].  Calibrated by running the hardware-validated tile
// configurations of t2v_gemm against the emulated backend (tests/test_hostsim_gemm.py).
inline v16f mfma_32x32x16_bf16(v8u16 a, v8u16 b, v16f c) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64, lane = me % 64;
    if (mf_a().size() < s.fibers.size()) { mf_a().resize(s.fibers.size()); mf_b().resize(s.fibers.size()); }
    mf_a()[me] = a;
    mf_b()[me] = b;
    wave_rendezvous();
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc += bf(mf_a()[w * 64 + row + 32 * (k >> 3)][k & 7]) * bf(mf_b()[w * 64 + col + 32 * (k >> 3)][k & 7]);
        c[r] = acc;
    }
    wave_rendezvous();
    return c;
}

}  // namespace hostsim

enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define __builtin_amdgcn_readfirstlane(x) (x)  /* only ever applied to wave-uniform values */
#define __builtin_amdgcn_fmed3f(a, b, c) hostsim::med3((a), (b), (c))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_wave_barrier() hostsim::wave_rendezvous()
#define __builtin_amdgcn_exp2f exp2f
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hostsim::glds((const void*)(g), (void*)(l), (size), (off))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hostsim::mfma_32x32x16_bf16((a), (b), (c))

#define threadIdx (hostsim::st().fibers[hostsim::st().cur].tid)
#define blockIdx (hostsim::st().bid)
#define blockDim (hostsim::st().block)
#define gridDim (hostsim::st().grid)
inline void __syncthreads() { hostsim::arrive(hostsim::st().block_sync, hostsim::alive_in_block()); }
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hostsim::shfl_xor(v, mask); }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hostsim::launch((grid), (block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })
