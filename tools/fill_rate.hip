// L2 -> CU fill-rate microbenchmark (gfx950): how fast can one CU pull an L2-resident operand panel
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, what t2v_gemm stages its operands with)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register-staged)
//   mode 2: global_load_dwordx4 -> VGPR only (upper bound of the vector-memory path)
// Every workgroup (4 or 8 waves) sweeps a window of a buffer small enough to live in the L2s / Infinity Cache, 16 KiB
// (16 wave-instructions of 1 KiB) at a time, like one ring slot of the GEMM.  Prints GB/s per CU and for the chip.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o /tmp/fill_rate && /tmp/fill_rate [window MiB] [iterations] [span KiB]
//   (span: each workgroup cycles inside its own span KiB of the window instead of sweeping all of it, e.g. 64)
//
// Why: every t2v_gemm shape of the UNet step moves its A and W tiles into LDS at 6 - 9.4 TB/s chip-wide
// (profiles/r01_gemm_ablation_by_shape.csv), and 9.8 TB/s = 256 CUs x 16 B/clk x 2.4 GHz is also what a DMA-only
// main loop reached.  If mode 1 is well above mode 0 here, the GEMM should stage through registers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

template <int N>
struct Slot { uint4 r[N]; };
template <int N>
__device__ __forceinline__ Slot<N> load_slot(const char* p, int wave, int lane) {
    Slot<N> s;
#pragma unroll
    for (int j = 0; j < N; ++j) s.r[j] = *(const uint4*)(p + (wave * N + j) * 1024 + lane * 16);
    return s;
}
template <int MODE, int N>
__device__ __forceinline__ void consume_slot(const Slot<N>& s, char* slot, int wave, int lane, unsigned& acc) {
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            *(uint4*)(slot + (wave * N + j) * 1024 + lane * 16) = s.r[j];
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) acc ^= s.r[j].x ^ s.r[j].w;
    }
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char* src, long long window, int iters, unsigned* sink, long long span) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 slots x 16 KiB per wave group
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PER_WAVE = 16 / NW >= 1 ? 16 / NW : 1;  // wave-instructions per 16 KiB slot per wave
    unsigned acc = 0;
    // slot `it` of this workgroup: a 16 KiB piece of the window (the buffer has 1 MiB of slack behind it)
    // span > 0: every workgroup cycles inside its own `span` bytes (what one CU can pull when nothing else limits it)
    auto src_of = [&](int it) __attribute__((always_inline)) {
        return span > 0 ? src + ((long long)blockIdx.x * span) % window + ((long long)it * 16384) % span
                        : src + (((long long)blockIdx.x + (long long)it * 61) * 16384) % window;
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            const char* p = src_of(it);
            char* slot = smem + (it & 1) * 16384;
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) {
                const int inst = wave * PER_WAVE + j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + inst * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
            }
            // keep one slot in flight, like a 2-deep ring
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        }
    } else {
        // register double buffer (loop unrolled by two so both sets stay in VGPRs): the loads of slot it+1 are in
        // flight while slot it is written / consumed, so as many bytes are outstanding per workgroup as in mode 0
        Slot<PER_WAVE> va = load_slot<PER_WAVE>(src_of(0), wave, lane), vb;
        for (int it = 0; it < iters; it += 2) {
            vb = load_slot<PER_WAVE>(src_of(it + 1), wave, lane);
            consume_slot<MODE, PER_WAVE>(va, smem, wave, lane, acc);
            va = load_slot<PER_WAVE>(src_of(it + 2), wave, lane);
            consume_slot<MODE, PER_WAVE>(vb, smem + 16384, wave, lane, acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) acc ^= *(const unsigned*)(smem + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;  // keep everything alive
}

template <int MODE, int NW>
double run(const char* buf, long long window, int blocks, int iters, unsigned* sink, long long span) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fill_kernel<MODE, NW>), dim3(blocks), dim3(NW * 64), 32768, 0, buf, window, iters, sink, span);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((fill_kernel<MODE, NW>), dim3(blocks), dim3(NW * 64), 32768, 0, buf, window, iters, sink, span);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * iters * 16384.0 / (ms * 1e-3) / 1e9;  // GB/s
}

int main(int argc, char** argv) {
    const long long window = (argc > 1 ? atoll(argv[1]) : 16) << 20;  // MiB; 16 MiB: half the aggregate L2, all of it in the Infinity Cache
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    const long long span = (argc > 3 ? atoll(argv[3]) : 0) << 10;  // KiB per workgroup (multiple of 16), 0 = sweep the whole window
    if (span < 0 || span > (1 << 20) || span % 16384 != 0 || window % 16384 != 0) {
        fprintf(stderr, "span must be a multiple of 16 KiB, at most 1024 KiB (the slack behind the window)\n");
        return 2;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    char* buf;
    unsigned* sink;
    CHECK(hipMalloc(&buf, window + (1 << 20)));
    CHECK(hipMemset(buf, 1, window + (1 << 20)));
    CHECK(hipMalloc(&sink, 4));
    printf("%s: %d CUs, window %lld MiB, %d x 16 KiB per workgroup, private span %lld KiB\n", prop.gcnArchName, cus, window >> 20, iters,
           span >> 10);
    printf("%-44s %10s %10s\n", "mode / waves per WG / WGs per CU", "GB/s/CU", "TB/s chip");
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int blocks = cus * per_cu;
        double r;
        r = run<0, 4>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "LDS-DMA dwordx4, 4 waves, 1 WG/CU" : "LDS-DMA dwordx4, 4 waves, 2 WG/CU", r / cus, r / 1e3);
        r = run<0, 8>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "LDS-DMA dwordx4, 8 waves, 1 WG/CU" : "LDS-DMA dwordx4, 8 waves, 2 WG/CU", r / cus, r / 1e3);
        r = run<1, 4>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "load + ds_write_b128, 4 waves, 1 WG/CU" : "load + ds_write_b128, 4 waves, 2 WG/CU", r / cus, r / 1e3);
        r = run<1, 8>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "load + ds_write_b128, 8 waves, 1 WG/CU" : "load + ds_write_b128, 8 waves, 2 WG/CU", r / cus, r / 1e3);
        r = run<2, 4>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "load to VGPR only, 4 waves, 1 WG/CU" : "load to VGPR only, 4 waves, 2 WG/CU", r / cus, r / 1e3);
        r = run<2, 8>(buf, window, blocks, iters, sink, span); printf("%-44s %10.1f %10.2f\n", per_cu == 1 ? "load to VGPR only, 8 waves, 1 WG/CU" : "load to VGPR only, 8 waves, 2 WG/CU", r / cus, r / 1e3);
    }
    return 0;
}
