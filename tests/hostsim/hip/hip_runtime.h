// Host SIMT simulator — TEST INFRASTRUCTURE ONLY (tests/hostsim/).  Stands in for <hip/hip_runtime.h> so that the library's
// kernel SOURCES compile with g++ and run on the CPU: every thread of a workgroup is a fiber, workgroups run one after the
// other, __syncthreads and the 64-lane shuffles / votes are rendezvous points between fibers, the matrix instructions
// (v_mfma_f32_32x32x16_bf16, 16x16x32, 16x16x16) and the LDS-DMA are wave collectives with the CDNA4 lane layouts, and the
// points where a kernel relies on a wave running in lockstep (s_waitcnt lgkmcnt(0), compiler fences, wave barriers) are wave
// rendezvous.  What it buys: the exact index arithmetic, reductions, LDS traffic and MFMA operand placement of kernels that
// have not run on hardware yet are executed and compared with the emulated op backend in the CPU suite.  What it cannot
// show: asynchrony (DMA ring / counted waits), the memory model, the gfx950 code generator, speed.
//
// Context switches are a dozen instructions of x86-64 assembly (callee-saved registers + stack pointer): glibc's swapcontext
// makes a sigprocmask system call per switch, and one simulated MFMA is 256 switches.
#pragma once
#define T2V_HOSTSIM 1  // sources may leave out what the simulator cannot run (inter-workgroup spin barriers)

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
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }

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

#if !defined(__x86_64__)
#error "the host SIMT simulator's context switch is written for x86-64"
#endif
extern "C" void hostsim_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak hostsim_switch
    .type hostsim_switch, @function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hostsim_switch, .-hostsim_switch
)");

namespace hostsim {

struct Sync { int count = 0; unsigned gen = 0; };
struct Fiber {
    void* sp = nullptr;
    const Sync* wait = nullptr;  // blocked on this rendezvous until its generation moves past wait_gen (scheduler skips it)
    unsigned wait_gen = 0;
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
    void* sched_sp = nullptr;
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
inline void yield() { State& s = st(); hostsim_switch(&s.fibers[s.cur].sp, s.sched_sp); }
inline void release_if_complete(Sync& sy, int alive) {
    if (sy.count > 0 && sy.count >= alive) { sy.count = 0; ++sy.gen; }
}
inline void arrive(Sync& sy, int alive) {
    ++sy.count;
    const unsigned g = sy.gen;
    if (sy.count >= alive) { sy.count = 0; ++sy.gen; return; }
    Fiber& f = st().fibers[st().cur];
    f.wait = &sy;
    f.wait_gen = g;
    while (sy.gen == g) yield();
    f.wait = nullptr;
}
inline void trampoline() {
    State& s = st();
    (*s.body)();
    Fiber& f = s.fibers[s.cur];
    f.done = true;  // an exited thread no longer takes part in barriers / shuffles
    release_if_complete(s.block_sync, alive_in_block());
    release_if_complete(s.wave_sync[f.flat / 64], alive_in_wave(f.flat / 64));
    hostsim_switch(&f.sp, s.sched_sp);  // never resumed
    abort();
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
                    f.wait = nullptr;
                    f.flat = i;
                    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
                    // initial frame: six zeroed callee-saved registers below the entry address, which sits 16-byte aligned so
                    // that the entry sees the stack as after a call
                    uintptr_t top = ((uintptr_t)(s.stacks.data() + (size_t)(i + 1) * kStack)) & ~(uintptr_t)15;
                    void** sp = (void**)(top - 16);
                    sp[0] = (void*)trampoline;
                    for (int r = 1; r <= 6; ++r) sp[-r] = nullptr;
                    f.sp = (void*)(sp - 6);
                }
                int done = 0;
                long long rounds = 0;
                // HOSTSIM_ORDER = reverse | random: visit the runnable fibers in another order every round.  Results must not
                // depend on it — a missing barrier (one wave restaging an LDS tile another wave still reads) shows up as a
                // result that changes with the schedule, which the fixed round-robin order could hide.
                const char* ord = getenv("HOSTSIM_ORDER");
                const int mode = !ord ? 0 : (ord[0] == 'r' && ord[1] == 'e') ? 1 : 2;
                uint64_t lcg = 0x9E3779B97F4A7C15ull;
                while (done < n) {
                    done = 0;
                    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                    const int start = mode == 2 ? (int)((lcg >> 33) % n) : 0;
                    const int stride = mode == 2 ? ((int)((lcg >> 13) % n) | 1) : 1;   // odd stride: a permutation when n is a power of two
                    for (int k = 0; k < n; ++k) {
                        int i = mode == 1 ? n - 1 - k : mode == 2 ? (int)(((long long)start + (long long)k * stride) % n) : k;
                        if (s.fibers[i].done) continue;
                        if (s.fibers[i].wait && s.fibers[i].wait->gen == s.fibers[i].wait_gen) continue;  // still blocked
                        s.cur = i;
                        hostsim_switch(&s.sched_sp, s.fibers[i].sp);
                    }
                    for (int i = 0; i < n; ++i) done += s.fibers[i].done;  // (a random round may visit a fiber twice or not at all)
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

template <class T>
inline T shfl_idx(T v, int src) {   // ds_bpermute: every lane reads lane `src`'s value (src may differ per lane)
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    s.slots[me] = bits;
    arrive(s.wave_sync[w], alive_in_wave(w));
    const int partner = w * 64 + (src & 63);
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
// buffer_load_dwordx4 ... offen lds (raw buffer form): lane copies `size` bytes from base + voffset + soffset + offset, or ZEROS when
// that range leaves [0, num_records) — the hardware's out-of-range rule for raw buffers, which the kernels use instead of a zero page.
struct BufferRsrc { const char* base; unsigned num_records; };
inline BufferRsrc make_buffer_rsrc(const void* p, int stride, unsigned num_records, unsigned flags) {
    (void)stride; (void)flags;
    return BufferRsrc{(const char*)p, num_records};
}
inline void buffer_lds(BufferRsrc r, void* lds_wave_base, int size, unsigned voffset, unsigned soffset, int offset) {
    State& s = st();
    const int lane = s.fibers[s.cur].flat % 64;
    char* dst = (char*)lds_wave_base + (size_t)lane * size;
    const unsigned long long off = (unsigned long long)voffset + soffset + (unsigned)offset;
    if (off + (unsigned)size <= r.num_records) memcpy(dst, r.base + off, size);
    else memset(dst, 0, size);
}
inline void wave_rendezvous() {
    State& s = st();
    const int w = s.fibers[s.cur].flat / 64;
    arrive(s.wave_sync[w], alive_in_wave(w));
}
typedef uint16_t v8u16 __attribute__((vector_size(16)));
typedef float v16f __attribute__((vector_size(64)));
struct F8 { float v[8]; };
inline std::vector<F8>& mf_a() { static std::vector<F8> v; return v; }
inline std::vector<F8>& mf_b() { static std::vector<F8> v; return v; }
inline float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
inline void put8(std::vector<F8>& dst, int me, v8u16 x) { for (int j = 0; j < 8; ++j) dst[me].v[j] = bf(x[j]); }
// v_mfma_f32_32x32x16_bf16 as a wave collective: D = A (32 x 16) * B (16 x 32) + C.  Operand layout (CDNA4 ISA): lane l holds
// A[row = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][col = l & 31]; accumulator register r of lane l is
// D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].  Calibrated by running the hardware-validated tile
// configurations of t2v_gemm against the emulated backend (tests/test_hostsim_gemm.py).
inline v16f mfma_32x32x16_bf16(v8u16 a, v8u16 b, v16f c) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64, lane = me % 64;
    if (mf_a().size() < s.fibers.size()) { mf_a().resize(s.fibers.size()); mf_b().resize(s.fibers.size()); }
    put8(mf_a(), me, a);
    put8(mf_b(), me, b);
    wave_rendezvous();
    const int col = lane & 31, hi = lane >> 5;
    const F8* A = mf_a().data() + w * 64;
    const F8 &b0 = mf_b()[w * 64 + col], &b1 = mf_b()[w * 64 + col + 32];
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 8; ++k) acc += A[row].v[k] * b0.v[k];
        for (int k = 0; k < 8; ++k) acc += A[row + 32].v[k] * b1.v[k];
        c[r] = acc;
    }
    wave_rendezvous();
    return c;
}

typedef float v4f __attribute__((vector_size(16)));
typedef short v4s16 __attribute__((vector_size(8)));
inline std::vector<v4s16>& mf_a4() { static std::vector<v4s16> v; return v; }
inline std::vector<v4s16>& mf_b4() { static std::vector<v4s16> v; return v; }
// v_mfma_f32_16x16x32_bf16: lane l holds A[row = l & 15][k = 8 (l >> 4) .. +7], B[k = 8 (l >> 4) .. +7][col = l & 15];
// accumulator register r of lane l is D[row = 4 (l >> 4) + r][col = l & 15]  (calibrated by t2v_attn_temporal)
inline v4f mfma_16x16x32_bf16(v8u16 a, v8u16 b, v4f c) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64, lane = me % 64;
    if (mf_a().size() < s.fibers.size()) { mf_a().resize(s.fibers.size()); mf_b().resize(s.fibers.size()); }
    put8(mf_a(), me, a);
    put8(mf_b(), me, b);
    wave_rendezvous();
    const int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k)
            acc += mf_a()[w * 64 + row + 16 * (k >> 3)].v[k & 7] * mf_b()[w * 64 + col + 16 * (k >> 3)].v[k & 7];
        c[r] = acc;
    }
    wave_rendezvous();
    return c;
}
// v_mfma_f32_16x16x16_bf16 (_1k form): 4 bf16 per lane, k = 4 (l >> 4) .. +3; same accumulator layout
inline v4f mfma_16x16x16_bf16(v4s16 a, v4s16 b, v4f c) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64, lane = me % 64;
    if (mf_a4().size() < s.fibers.size()) { mf_a4().resize(s.fibers.size()); mf_b4().resize(s.fibers.size()); }
    mf_a4()[me] = a;
    mf_b4()[me] = b;
    wave_rendezvous();
    const int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc += bf((uint16_t)mf_a4()[w * 64 + row + 16 * (k >> 2)][k & 3]) * bf((uint16_t)mf_b4()[w * 64 + col + 16 * (k >> 2)][k & 3]);
        c[r] = acc;
    }
    wave_rendezvous();
    return c;
}

inline bool vote_any(bool pred) {
    State& s = st();
    const int me = s.fibers[s.cur].flat, w = me / 64;
    s.slots[me] = pred ? 1 : 0;
    wave_rendezvous();
    bool r = false;
    for (int i = w * 64; i < std::min<int>((w + 1) * 64, (int)s.fibers.size()); ++i) r = r || (!s.fibers[i].done && s.slots[i]);
    wave_rendezvous();
    return r;
}
// IEEE half <-> float (g++ 11 on x86 has no _Float16 in C++)
struct half_t {
    uint16_t bits;
    half_t() : bits(0) {}
    half_t(float f) {
        uint32_t x; memcpy(&x, &f, 4);
        const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
        if (x >= 0x7f800000u) { bits = (uint16_t)(sign | 0x7c00u | ((x & 0x7fffffu) ? 0x200u : 0)); return; }
        if (x >= 0x477ff000u) { bits = (uint16_t)(sign | 0x7c00u); return; }                       // overflow -> inf
        if (x < 0x33000001u) { bits = (uint16_t)sign; return; }                                    // underflow -> 0
        int e = (int)(x >> 23) - 127;
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        int shift = e < -14 ? 13 + (-14 - e) : 13;
        uint32_t h = e < -14 ? 0u : (uint32_t)(e + 15) << 10;
        uint32_t mant = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (e >= -14) mant &= 0x3ffu;
        h += mant;
        if (rem > half || (rem == half && (h & 1))) ++h;
        bits = (uint16_t)(sign | h);
    }
    operator float() const {
        const uint32_t sign = (uint32_t)(bits & 0x8000u) << 16, e = (bits >> 10) & 0x1f, m = bits & 0x3ffu;
        uint32_t x;
        if (e == 0) {
            if (m == 0) x = sign;
            else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++sh; } x = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13); }
        } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
        else x = sign | ((e + 112) << 23) | (m << 13);
        float f; memcpy(&f, &x, 4); return f;
    }
};

}  // namespace hostsim

#define _Float16 hostsim::half_t
#define __any(p) hostsim::vote_any((p))
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hostsim::mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, x, y, z) hostsim::mfma_16x16x16_bf16((a), (b), (c))
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
#define __amdgpu_buffer_rsrc_t hostsim::BufferRsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hostsim::make_buffer_rsrc((const void*)(p), (stride), (num), (flags))
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, size, voff, soff, off, aux) hostsim::buffer_lds((r), (void*)(l), (size), (voff), (soff), (off))

#define threadIdx (hostsim::st().fibers[hostsim::st().cur].tid)
#define blockIdx (hostsim::st().bid)
#define blockDim (hostsim::st().block)
#define gridDim (hostsim::st().grid)
inline void __syncthreads() { hostsim::arrive(hostsim::st().block_sync, hostsim::alive_in_block()); }
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hostsim::shfl_xor(v, mask); }
template <class T>
inline T __shfl(T v, int src, int width = 64) { (void)width; return hostsim::shfl_idx(v, src); }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hostsim::launch((grid), (block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })
