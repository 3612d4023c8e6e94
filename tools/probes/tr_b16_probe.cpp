// What does ds_read_b64_tr_b16 deliver?  LDS holds lds[i] = i (16-bit); every lane reads 8 bytes at its own address
//   pattern 0: addr = 8 * lane (lane l points at elements 4l .. 4l+3)
//   pattern 1: addr = 32 * (lane & 15) + 8 * (lane >> 4) (16-element rows: lane l of a 16-lane group points at row l, the groups at
//              column blocks 0, 4, 8, 12)
// and the program prints the four 16-bit values every lane received.   hipcc --offload-arch=gfx950 -O2 tr_b16_probe.cpp -o tr_b16_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ void probe(int pattern, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int byte_addr = pattern == 0 ? 8 * lane : 32 * (lane & 15) + 8 * (lane >> 4);
    const uint32_t a = (uint32_t)(uintptr_t)lds + byte_addr;   // LDS aperture: the low 32 bits are the LDS byte address
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int pattern = 0; pattern < 2; ++pattern) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pattern, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pattern);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
