// Kernels of the LoRA training path (weight gradients of the student UNet).  Own translation unit: what hipcc emits
// for a kernel depends on its neighbours, and everything else in the library is hardware-validated.
// Not yet run on hardware; executed on the host SIMT simulator (tests/hostsim) bit for bit against tests/emu_ops.py.
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

// ---------------------------------------------------------------------------------------------------------------------
// Dropout with a counter-based mask: keep(row, col) is a pure function of (seed, site, row * ncols + col), so the backward
// regenerates the forward's mask instead of storing it (nn.Dropout in LoraInjected*.forward, utils/lora.py:45-50,124-129,
// and in TemporalConvBlock conv2..4, openaimodel3d.py:280-297).  One splitmix64 finaliser per QUAD of adjacent elements
// (16 bits each).  out = keep ? x / (1 - p) : 0   (+ resid).  `seed` lives in device memory: a replayed launch list
// sees the step's seed without being re-recorded.
// (splitmix64 / dropout_quad / dropout_keep_mask: common.h)

// Thread mapping: a block walks whole rows — thread (ty, tx) of a (256 / TX) x TX arrangement owns the W-element chunk tx, tx + TX, ...
// of rows ty, ty + 256 / TX, ... of the block's row range — so the (row, column) of a chunk comes from adds, not from a 64-bit
// division per chunk (which cost more VALU time than the mask itself: 14.3 us per launch at 40960 x 320 against a 6.5 us HBM floor).
template <bool VEC8>
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ resid, int ldr,
                                                      bf16_t* __restrict__ out, int ldo, long long rows, int ncols,
                                                      const uint64_t* __restrict__ seed_p, uint32_t site, uint32_t thr, float inv_keep,
                                                      int tx_n, int rows_per_block) {
    constexpr int W = VEC8 ? 8 : 2;
    const int per_row = ncols / W;
    const uint64_t key = dropout_key(*seed_p, site);
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n, ty_n = 256 / tx_n;
    if (ty >= ty_n) return;   // (256 is not a multiple of tx_n: the last few threads have no row)
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    const long long r_end = r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows;
    // RI rows per thread and pass: all their loads are issued before the first mask is drawn (one 16-byte load in flight per thread
    // left the kernel latency-bound at ~3 TB/s of traffic)
    constexpr int RI = 4;
    for (long long rb = r_begin + ty; rb < r_end; rb += (long long)RI * ty_n) {
        for (int cc = tx; cc < per_row; cc += tx_n) {
            const int c = cc * W;
            uint4 ux[RI], ur[RI];
#pragma unroll
            for (int k = 0; k < RI; ++k) {
                const long long r = rb + (long long)k * ty_n;
                ux[k] = ur[k] = make_uint4(0, 0, 0, 0);
                if (r < r_end) {
                    if constexpr (VEC8) {
                        ux[k] = *(const uint4*)(x + r * ldx + c);
                        if (resid) ur[k] = *(const uint4*)(resid + r * ldr + c);
                    } else {
                        ux[k].x = *(const uint32_t*)(x + r * ldx + c);
                        if (resid) ur[k].x = *(const uint32_t*)(resid + r * ldr + c);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < RI; ++k) {
                const long long r = rb + (long long)k * ty_n;
                if (r >= r_end) continue;
                const uint64_t i0 = (uint64_t)r * (uint64_t)ncols + (uint64_t)c;   // flat element index: even (W = 2) / a multiple of 8 (VEC8)
                float v[W], rs[W];
                if constexpr (VEC8) {
                    unpack8(ux[k], v);
                    unpack8(ur[k], rs);
                    const uint32_t keep = dropout_keep_mask<2>(key, i0 >> 2, thr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * inv_keep : 0.f;
                } else {
                    v[0] = __uint_as_float(ux[k].x << 16); v[1] = __uint_as_float(ux[k].x & 0xffff0000u);
                    rs[0] = __uint_as_float(ur[k].x << 16); rs[1] = __uint_as_float(ur[k].x & 0xffff0000u);
                    const uint64_t w = dropout_quad(key, i0 >> 2);
                    const int e0 = (int)(i0 & 2);
                    v[0] = dropout_keep16(w, e0, thr) ? v[0] * inv_keep : 0.f;
                    v[1] = dropout_keep16(w, e0 + 1, thr) ? v[1] * inv_keep : 0.f;
                }
                if (resid) {
#pragma unroll
                    for (int e = 0; e < W; ++e) v[e] += rs[e];
                }
                if constexpr (VEC8) *(uint4*)(out + r * ldo + c) = pack8(v);
                else *(uint32_t*)(out + r * ldo + c) = pack2bf(v[0], v[1]);
            }
        }
    }
}

extern "C" int t2v_dropout_bf16(const void* x, int ldx, const void* resid, int ldr, void* out, int ldo, long long rows, int ncols,
                                float p, const void* seed, unsigned site, void* stream) {
    T2V_REQUIRE(x && out && seed && rows > 0 && ncols > 0, T2V_EINVAL, "t2v_dropout_bf16: null pointer / empty");
    // (p >= 1 - 2^-16 would quantise to "one element in 65536 survives at scale 65536": refused, torch's p = 1 is all zeros)
    T2V_REQUIRE(p >= 0.f && p < 1.f - 1.f / 65536.f, T2V_EINVAL, "t2v_dropout_bf16: p must be in [0, 1 - 2^-16)");
    T2V_REQUIRE(ncols % 2 == 0 && ldx % 2 == 0 && ldo % 2 == 0 && (!resid || ldr % 2 == 0) && ldx >= ncols && ldo >= ncols, T2V_ESHAPE,
                "t2v_dropout_bf16: even column count / row strides");
    T2V_REQUIRE((uintptr_t)x % 4 == 0 && (uintptr_t)out % 4 == 0 && (!resid || (uintptr_t)resid % 4 == 0), T2V_ESHAPE,
                "t2v_dropout_bf16: 4-byte aligned rows");
    const double t = (double)p * 4294967296.0;
    const uint32_t thr = (t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t) >> 16;   // 16 bits per element
    const float inv_keep = 65536.0f / (65536.0f - (float)thr);   // the scale of the QUANTISED drop probability thr / 65536 (E[out] = x exactly)
    const bool vec8 = ncols % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && (!resid || ldr % 8 == 0) && (uintptr_t)x % 16 == 0 &&
                      (uintptr_t)out % 16 == 0 && (!resid || (uintptr_t)resid % 16 == 0);
    const int per_row = ncols / (vec8 ? 8 : 2);
    // threads across a row: the count that keeps most of the block's 256 threads busy (chunks per row need not divide anything:
    // 40 chunks -> 40 x 6 threads, 160 chunks -> 80 x 3 in two passes)
    int tx_n = 1;
    double best = 0.0;
    for (int cand = 1; cand <= 256 && cand <= per_row; ++cand) {
        const int passes = (per_row + cand - 1) / cand;
        const double eff = (double)per_row / ((double)passes * cand) * (double)((256 / cand) * cand) / 256.0;
        if (eff >= best) { best = eff; tx_n = cand; }
    }
    const int ty_n = 256 / tx_n;
    // about 2048 blocks (8 per CU) of whole rows; at least one pass of the block's row arrangement each
    long long rpb = (rows + 2047) / 2048;
    rpb = ((rpb + ty_n - 1) / ty_n) * ty_n;   // (whole row groups of the block; large tensors get >= 4 of them per thread, the kernel's register blocking)
    const long long blocks = (rows + rpb - 1) / rpb;
    if (vec8)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           (const bf16_t*)resid, ldr, (bf16_t*)out, ldo, rows, ncols, (const uint64_t*)seed, site, thr, inv_keep, tx_n, (int)rpb);
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           (const bf16_t*)resid, ldr, (bf16_t*)out, ldo, rows, ncols, (const uint64_t*)seed, site, thr, inv_keep, tx_n, (int)rpb);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// out[b][c][r] = in[b][r][c] for r < rows, and 0 for rows <= r < roundup(rows, 64): the K-contiguous, K-padded operand of a
// token-contracted weight-gradient GEMM in ONE pass (t2v_transpose_bf16 + a memset before it, with 2-byte accesses, was the
// first version).  64x64 tile through LDS, 16-byte global loads and stores on both sides: a thread loads two 8-element row
// chunks and stores two 8-element chunks of the transposed rows.  LDS rows are padded to 66 elements (33 words: the column
// walk of the read-out touches 8 different banks per chunk).
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ in, int ld_in, int rows, int cols,
                                                            bf16_t* __restrict__ out, int ld_out, long long in_stride, long long out_stride) {
    __shared__ bf16_t tile[64][66];
    const bf16_t* ib = in + blockIdx.z * in_stride;
    bf16_t* ob = out + blockIdx.z * out_stride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, r = q >> 3, ch = (q & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows && c0 + ch < cols) v = *(const uint4*)(ib + (long long)(r0 + r) * ld_in + c0 + ch);
        uint32_t* t = (uint32_t*)&tile[r][ch];  // (66-element rows: 4-byte aligned, not 16)
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, c = q >> 3, rh = (q & 7) * 8;
        if (c0 + c >= cols) continue;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rh + 2 * e][c] | ((uint32_t)tile[rh + 2 * e + 1][c] << 16);
        *(uint4*)(ob + (long long)(c0 + c) * ld_out + r0 + rh) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

extern "C" int t2v_transpose_pad_bf16(const void* in, int ld_in, int rows, int cols, void* out, int ld_out, int batch,
                                      long long in_stride, long long out_stride, void* stream) {
    T2V_REQUIRE(in && out && rows > 0 && cols > 0 && batch > 0 && batch <= 65535, T2V_EINVAL, "t2v_transpose_pad_bf16: bad argument");
    const int rows_pad = (rows + 63) / 64 * 64;
    T2V_REQUIRE(cols % 8 == 0 && ld_in % 8 == 0 && ld_in >= cols && ld_out % 8 == 0 && ld_out >= rows_pad && in_stride % 8 == 0 &&
                out_stride % 8 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, T2V_ESHAPE,
                "t2v_transpose_pad_bf16: 16-byte rows on both sides, ld_out >= roundup(rows, 64)");
    T2V_REQUIRE((rows + 63) / 64 <= 65535, T2V_ESHAPE, "t2v_transpose_pad_bf16: too many rows");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, ld_in, rows, cols, (bf16_t*)out, ld_out, in_stride, out_stride);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
