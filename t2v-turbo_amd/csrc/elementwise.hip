// Layout conversion, embeddings, small direct conv and the scheduler / consistency-distillation
// elementwise family for gfx950.  All bandwidth/latency bound; vectorised where the layout allows.
#include <math.h>
#include "common.h"

namespace {

__device__ __forceinline__ float load_as_f32(const void* p, int dt, long long i) {
    if (dt == T2V_F32) return ((const float*)p)[i];
    if (dt == T2V_BF16) return bf2f(((const bf16_t*)p)[i]);
    return (float)(((const _Float16*)p)[i]);
}
__device__ __forceinline__ void store_from_f32(void* p, int dt, long long i, float v) {
    if (dt == T2V_F32) ((float*)p)[i] = v;
    else if (dt == T2V_BF16) ((bf16_t*)p)[i] = f2bf(v);
    else ((_Float16*)p)[i] = (_Float16)v;
}

// (b,c,f,hw) -> rows ((b*f + fi)*hw + p), c columns.  One thread per (row): reads c strided
// planes (each coalesced across threads), writes c contiguous values.
__global__ void ncfhw_to_tokens_kernel(const void* x, int dt, int B, int C, int F, int HW, bf16_t* out, int ldo) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)B * F * HW;
    if (m >= M) return;
    const int p = (int)(m % HW);
    const long long bf = m / HW;
    const int f = (int)(bf % F);
    const long long b = bf / F;
    for (int c = 0; c < C; ++c)
        out[m * ldo + c] = f2bf(load_as_f32(x, dt, ((b * C + c) * F + f) * HW + p));
}
__global__ void tokens_to_ncfhw_kernel(const void* tok, int tok_f32, int ld, int B, int C, int F, int HW, void* out, int dt) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)B * F * HW;
    if (m >= M) return;
    const int p = (int)(m % HW);
    const long long bf = m / HW;
    const int f = (int)(bf % F);
    const long long b = bf / F;
    for (int c = 0; c < C; ++c) {
        const float v = tok_f32 ? ((const float*)tok)[m * ld + c] : bf2f(((const bf16_t*)tok)[m * ld + c]);
        store_from_f32(out, dt, ((b * C + c) * F + f) * HW + p, v);
    }
}

__global__ void timestep_embedding_kernel(const void* t, int t_is_f32, int n, int dim, int guidance_style, bf16_t* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (idx >= n * half) return;
    const int row = idx / half, i = idx % half;
    float tv = t_is_f32 ? ((const float*)t)[row] : (float)((const long long*)t)[row];
    float c, s;
    if (guidance_style) {  // sin||cos of (w*1000) * exp(-i*ln(1e4)/(half-1))
        const float freq = expf((float)i * -(logf(10000.0f) / (float)(half - 1)));
        const float a = tv * 1000.0f * freq;
        sincosf(a, &s, &c);
        out[row * dim + i] = f2bf(s);
        out[row * dim + half + i] = f2bf(c);
    } else {  // cos||sin of t * exp(-ln(1e4) * i / half)
        const float freq = expf(-logf(10000.0f) * (float)i / (float)half);
        const float a = tv * freq;
        sincosf(a, &s, &c);
        out[row * dim + i] = f2bf(c);
        out[row * dim + half + i] = f2bf(s);
    }
    if ((dim & 1) && i == 0) out[row * dim + dim - 1] = 0;
}

__global__ void silu_kernel(const bf16_t* x, bf16_t* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f2bf(silu_f(bf2f(x[i])));
}
__global__ void cast_kernel(const void* x, int dt_in, void* out, int dt_out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store_from_f32(out, dt_out, i, load_as_f32(x, dt_in, i));
}

struct Coef3 { float a[64], b[64], c[64]; };
__global__ void lincomb3_kernel(const float* x, const float* y, const float* z, Coef3 k, long long inner, long long total, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / inner);
    float v = k.a[b] * x[i];
    if (y) v += k.b[b] * y[i];
    if (z) v += k.c[b] * z[i];
    out[i] = v;
}
__global__ void lcm_step_kernel(const float* x, const void* eps, int eps_dt, const float* noise, float sa_t, float sb_t,
                                float c_skip, float c_out, float sa_p, float sb_p, long long n, float* prev, float* den) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i];
    const float e = load_as_f32(eps, eps_dt, i);
    const float x0 = (xv - sb_t * e) / sa_t;
    const float dn = c_out * x0 + c_skip * xv;
    den[i] = dn;
    prev[i] = noise ? sa_p * dn + sb_p * noise[i] : dn;
}

// direct 3x3 s1 p1 conv for cin <= 8: thread = (token, 8 output channels); weights fp32 in LDS, staged ONCE per block
// (a block walks its share of the tokens grid-stride: staging the 46 KiB of a 4 -> 320 conv for every 6 tokens was
// 9/10 of this kernel's time)
template <int CIN>
__global__ __launch_bounds__(256) void conv_small_kernel(const bf16_t* x, int n_img, int H, int W, const float* wgt,
                                                         const float* bias, int cout, bf16_t* out) {
    extern __shared__ float sw[];  // [9*CIN][cout + 4]: a thread reads its 8 output channels as two float4
    const int ldw = cout + 4;      // +4: the transposing writes below (consecutive k, same oc) spread over 16 banks, not 1
    for (int i = threadIdx.x; i < cout * 9 * CIN; i += 256) {
        const int oc = i / (9 * CIN), k = i - oc * 9 * CIN;
        sw[k * ldw + oc] = wgt[i];
    }
    __syncthreads();
    const int nch = cout / 8;
    const long long M = (long long)n_img * H * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < M * nch; e += (long long)gridDim.x * 256) {
    const long long m = e / nch;
    const int oc0 = (int)(e % nch) * 8;
    const int px = (int)(m % W), py = (int)((m / W) % H);
    const long long nbase = (m / ((long long)W * H)) * H * W;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bias ? bias[oc0 + o] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = py + ky - 1, ix = px + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const bf16_t* xp = x + (nbase + (long long)iy * W + ix) * CIN;
            float xv[CIN];
            if constexpr (CIN == 4) {
                const uint2 u = *(const uint2*)xp;
                xv[0] = __uint_as_float(u.x << 16); xv[1] = __uint_as_float(u.x & 0xffff0000u);
                xv[2] = __uint_as_float(u.y << 16); xv[3] = __uint_as_float(u.y & 0xffff0000u);
            } else {
                unpack8(*(const uint4*)xp, xv);
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float* wp = sw + ((ky * 3 + kx) * CIN + c) * ldw + oc0;
                const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
                acc[0] += xv[c] * w0.x; acc[1] += xv[c] * w0.y; acc[2] += xv[c] * w0.z; acc[3] += xv[c] * w0.w;
                acc[4] += xv[c] * w1.x; acc[5] += xv[c] * w1.y; acc[6] += xv[c] * w1.z; acc[7] += xv[c] * w1.w;
            }
        }
    *(uint4*)(out + m * cout + oc0) = pack8(acc);
    }
}

inline unsigned nblk(long long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

extern "C" int t2v_ncfhw_to_tokens(const void* x, int dt, int b, int c, int f, int hw, void* out, int ldo, void* stream) {
    T2V_REQUIRE(x && out && b > 0 && c > 0 && f > 0 && hw > 0 && ldo >= c && dt >= 0 && dt <= 2, T2V_EINVAL, "t2v_ncfhw_to_tokens");
    hipLaunchKernelGGL(ncfhw_to_tokens_kernel, dim3(nblk((long long)b * f * hw)), dim3(256), 0, (hipStream_t)stream, x, dt, b,
                       c, f, hw, (bf16_t*)out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_tokens_to_ncfhw(const void* tok, int tok_f32, int ld, int b, int c, int f, int hw, void* out, int dt,
                                   void* stream) {
    T2V_REQUIRE(tok && out && b > 0 && c > 0 && f > 0 && hw > 0 && ld >= c && dt >= 0 && dt <= 2, T2V_EINVAL, "t2v_tokens_to_ncfhw");
    hipLaunchKernelGGL(tokens_to_ncfhw_kernel, dim3(nblk((long long)b * f * hw)), dim3(256), 0, (hipStream_t)stream, tok,
                       tok_f32, ld, b, c, f, hw, out, dt);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_timestep_embedding(const void* t, int t_is_f32, int n, int dim, int guidance_style, void* out, void* stream) {
    T2V_REQUIRE(t && out && n > 0 && dim >= 4, T2V_EINVAL, "t2v_timestep_embedding");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblk((long long)n * (dim / 2))), dim3(256), 0, (hipStream_t)stream, t,
                       t_is_f32, n, dim, guidance_style, (bf16_t*)out);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_silu(const void* x, void* out, long long n, void* stream) {
    T2V_REQUIRE(x && out && n > 0, T2V_EINVAL, "t2v_silu");
    hipLaunchKernelGGL(silu_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, n);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
// Zero fill as a KERNEL (not hipMemsetAsync): every node of a captured launch list is then a kernel node, ordered like every
// other launch of the chain.  16-byte stores over the aligned body, byte stores for the (at most 15 + 15) edge bytes.
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, long long head, long long body16, long long nbytes) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    uint4* b = (uint4*)(p + head);
    for (long long i = i0; i < body16; i += step) b[i] = make_uint4(0, 0, 0, 0);
    const long long tail0 = head + body16 * 16;
    if (i0 < head) p[i0] = 0;
    if (i0 < nbytes - tail0) p[tail0 + i0] = 0;
}
extern "C" int t2v_fill_zero(void* p, long long nbytes, void* stream) {
    T2V_REQUIRE(p && nbytes > 0, T2V_EINVAL, "t2v_fill_zero");
    long long head = (16 - (long long)((uintptr_t)p & 15)) & 15;
    if (head > nbytes) head = nbytes;
    const long long body16 = (nbytes - head) / 16;
    long long blocks = (body16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, head, body16, nbytes);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_cast(const void* x, int dt_in, void* out, int dt_out, long long n, void* stream) {
    T2V_REQUIRE(x && out && n > 0 && dt_in >= 0 && dt_in <= 2 && dt_out >= 0 && dt_out <= 2, T2V_EINVAL, "t2v_cast");
    hipLaunchKernelGGL(cast_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, dt_in, out, dt_out, n);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_lincomb3(const float* x, const float* y, const float* z, const float* ca, const float* cb,
                            const float* cc, int nb, long long inner, float* out, void* stream) {
    T2V_REQUIRE(x && out && ca && nb > 0 && nb <= 64 && inner > 0, T2V_EINVAL, "t2v_lincomb3");
    T2V_REQUIRE((!y || cb) && (!z || cc), T2V_EINVAL, "t2v_lincomb3: missing coefficients");
    Coef3 k;
    for (int i = 0; i < nb; ++i) { k.a[i] = ca[i]; k.b[i] = cb ? cb[i] : 0.f; k.c[i] = cc ? cc[i] : 0.f; }
    hipLaunchKernelGGL(lincomb3_kernel, dim3(nblk(inner * nb)), dim3(256), 0, (hipStream_t)stream, x, y, z, k, inner,
                       inner * nb, out);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_lcm_step(const float* x, const void* eps, int eps_dt, const float* noise, float sa_t, float sb_t,
                            float c_skip, float c_out, float sa_p, float sb_p, long long n, float* prev, float* denoised,
                            void* stream) {
    T2V_REQUIRE(x && eps && prev && denoised && n > 0 && sa_t != 0.f, T2V_EINVAL, "t2v_lcm_step");
    hipLaunchKernelGGL(lcm_step_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, eps, eps_dt, noise, sa_t, sb_t,
                       c_skip, c_out, sa_p, sb_p, n, prev, denoised);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
#ifdef T2V_EXPERIMENTAL   // (4x slower than the MFMA tile on MI355X: in the T2V_EXPERIMENTAL=1 library only)
// ---- direct 3x3 s1 p1 conv for a tiny number of OUTPUT channels (the VAE decoder's conv_out: 128 -> 3 at 320x512, ae_modules.py:641) ------
// An MFMA tile is at least 64 channels wide: at cout = 3 the implicit-GEMM kernels multiply 0.77 TFLOP of padding per 16-frame decode
// (0.75 ms).  Here a thread owns FOUR consecutive output pixels of an image row and all COUT channels: per filter row and 8-channel
// chunk it fetches the six source pixels it needs (16-byte loads, neighbours shared through L1) and walks the three column taps; the
// fp32 weights sit in LDS as [tap][cin][4] so that one broadcast ds_read_b128 serves 4 pixels x COUT FMAs (12 FMAs per LDS read:
// the per-pixel form needs one read per 3 and is LDS-bound).  VALU-bound by design: 9 * cin * COUT FMAs per pixel.
template <int COUT>
__global__ __launch_bounds__(256, 3) void conv_small_cout_kernel(const bf16_t* __restrict__ x, int ldx, int n_img, int H, int W, int cin,
                                                              const float* __restrict__ wgt, const float* __restrict__ bias,
                                                              void* __restrict__ out, int ldo, int out_f32, int cout) {
    extern __shared__ float sw[];  // [9 * cin][4]
    for (int i = threadIdx.x; i < 9 * cin * 4; i += 256) {
        const int k = i >> 2, oc = i & 3;
        sw[i] = oc < cout ? wgt[(long long)oc * 9 * cin + k] : 0.f;
    }
    __syncthreads();
    const int qpr = W >> 2;                                   // quads per image row (host-checked: W % 4 == 0)
    const long long nquad = (long long)n_img * H * qpr;
    const int nchunk = cin >> 3;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (long long)gridDim.x * 256) {
        const int px0 = (int)(q % qpr) * 4;
        const long long row = q / qpr;                        // img * H + py
        const int py = (int)(row % H);
        float acc[4][COUT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int oc = 0; oc < COUT; ++oc) acc[t][oc] = (bias && oc < cout) ? bias[oc] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = py + ky - 1;
            if (iy < 0 || iy >= H) continue;
            const bf16_t* xr = x + ((row + ky - 1) * W) * (long long)ldx;   // source row (same image: iy is inside it)
#pragma unroll 1
            for (int c8 = 0; c8 < nchunk; ++c8) {
                uint4 u[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int ix = px0 + j - 1;
                    u[j] = make_uint4(0u, 0u, 0u, 0u);
                    if (ix >= 0 && ix < W) u[j] = *(const uint4*)(xr + (long long)ix * ldx + c8 * 8);
                }
                const float* wp = sw + (ky * 3 * cin + c8 * 8) * 4;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // channel c of the six pixels: a shift or a mask of the dword that holds it (unpacked on the way, not kept)
                    float xs[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const uint32_t w2 = (c >> 1) == 0 ? u[j].x : ((c >> 1) == 1 ? u[j].y : ((c >> 1) == 2 ? u[j].z : u[j].w));
                        xs[j] = __uint_as_float((c & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 w4 = *(const float4*)(wp + (kx * cin + c) * 4);
                        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int oc = 0; oc < COUT; ++oc) acc[t][oc] = fmaf(xs[t + kx], wv[oc], acc[t][oc]);
                    }
                }
            }
        }
        const long long m0 = row * W + px0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int oc = 0; oc < COUT; ++oc) {
                if (oc >= cout) continue;
                if (out_f32) ((float*)out)[(m0 + t) * ldo + oc] = acc[t][oc];
                else ((bf16_t*)out)[(m0 + t) * ldo + oc] = f2bf(acc[t][oc]);
            }
    }
}

extern "C" int t2v_conv3x3_small_cout_supported(int w, int cin, int cout) {
    return (w > 0 && w % 4 == 0 && cin > 0 && cin % 8 == 0 && cout >= 1 && cout <= 4 && 9 * cin * 16 <= 150 * 1024) ? 1 : 0;
}
extern "C" int t2v_conv3x3_small_cout(const void* x, int ldx, int n_img, int h, int w, int cin, const float* wgt, const float* bias,
                                      int cout, void* out, int ldo, int out_f32, void* stream) {
    T2V_REQUIRE(x && wgt && out && n_img > 0 && h > 0 && w > 0, T2V_EINVAL, "t2v_conv3x3_small_cout");
    T2V_REQUIRE(t2v_conv3x3_small_cout_supported(w, cin, cout) && ldx % 8 == 0 && ldx >= cin && ldo >= cout && (uintptr_t)x % 16 == 0,
                T2V_ESHAPE, "t2v_conv3x3_small_cout: w % 4, cin % 8, 1 <= cout <= 4, 16-byte aligned rows");
    const long long nquad = (long long)n_img * h * (w / 4);
    const int smem = 9 * cin * 16;
    long long blocks = (nquad + 255) / 256;
    if (blocks > 256LL * 8) blocks = 256LL * 8;   // grid-stride beyond 8 workgroups per CU
    hipStream_t s = (hipStream_t)stream;
#define T2V_CSC_LAUNCH(CO)                                                                                                              \
    do {                                                                                                                                \
        static bool set = false;                                                                                                        \
        if (!set) { hipFuncSetAttribute((const void*)conv_small_cout_kernel<CO>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set = true; } \
        hipLaunchKernelGGL(conv_small_cout_kernel<CO>, dim3((unsigned)blocks), dim3(256), smem, s, (const bf16_t*)x, ldx, n_img, h, w, cin, \
                           wgt, bias, out, ldo, out_f32, cout);                                                                         \
    } while (0)
    // (cout = 3 — the one shape the VAE has — compiles to MORE registers than 4 and spills; 1 and 2 have their own instantiations,
    // 3 runs the 4-channel code with a zero fourth filter and three stores)
    if (cout == 1) T2V_CSC_LAUNCH(1);
    else if (cout == 2) T2V_CSC_LAUNCH(2);
    else T2V_CSC_LAUNCH(4);
#undef T2V_CSC_LAUNCH
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
#endif   // T2V_EXPERIMENTAL

extern "C" int t2v_conv3x3_small_cin(const void* x, int n_img, int h, int w, int cin, const float* wgt, const float* bias,
                                     int cout, void* out, void* stream) {
    T2V_REQUIRE(x && wgt && out && n_img > 0 && h > 0 && w > 0, T2V_EINVAL, "t2v_conv3x3_small_cin");
    T2V_REQUIRE((cin == 4 || cin == 8) && cout % 8 == 0 && (cout + 4) * 9 * cin * 4 <= 150 * 1024, T2V_ESHAPE,
                "t2v_conv3x3_small_cin: cin must be 4 or 8, cout a multiple of 8");
    const long long work = (long long)n_img * h * w * (cout / 8);
    const int smem = (cout + 4) * 9 * cin * 4;
    // as many blocks as stay resident at once (LDS-limited), at least 8 tokens' worth of work each
    const int per_cu = smem > 0 ? (160 * 1024) / smem : 1;
    long long blocks = 256LL * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    if (blocks > (work + 255) / 256) blocks = (work + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (cin == 4) {
        static bool set4 = false;
        if (!set4) { hipFuncSetAttribute((const void*)conv_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set4 = true; }
        hipLaunchKernelGGL(conv_small_kernel<4>, dim3((unsigned)blocks), dim3(256), smem, s, (const bf16_t*)x, n_img, h, w, wgt, bias, cout, (bf16_t*)out);
    } else {
        static bool set8 = false;
        if (!set8) { hipFuncSetAttribute((const void*)conv_small_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set8 = true; }
        hipLaunchKernelGGL(conv_small_kernel<8>, dim3((unsigned)blocks), dim3(256), smem, s, (const bf16_t*)x, n_img, h, w, wgt, bias, cout, (bf16_t*)out);
    }
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// ---- optimizer / EMA over flat buffers ---------------------------------------------------------------
namespace {
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        float4 pp = *(float4*)(p + i), gg = *(const float4*)(g + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
        float* P = (float*)&pp; float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = G[e] * gscale;
            P[e] *= 1.0f - lr * wd;
            M[e] = b1 * M[e] + (1.0f - b1) * gr;
            V[e] = b2 * V[e] + (1.0f - b2) * gr * gr;
            P[e] -= (lr / bc1) * M[e] / (sqrtf(V[e]) / bc2_sqrt + eps);
        }
        *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
    } else {
        for (long long j = i; j < n; ++j) {
            const float gr = g[j] * gscale;
            float pj = p[j] * (1.0f - lr * wd);
            const float mj = b1 * m[j] + (1.0f - b1) * gr, vj = b2 * v[j] + (1.0f - b2) * gr * gr;
            pj -= (lr / bc1) * mj / (sqrtf(vj) / bc2_sqrt + eps);
            p[j] = pj; m[j] = mj; v[j] = vj;
        }
    }
}
__global__ void ema_kernel(float* t, const float* s, float rate, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = t[i] * rate + s[i] * (1.0f - rate);
}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, long long n, float* ws) {
    __shared__ float sh[4];
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a += x[i] * x[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* ws, int nblk, float* out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) a += (double)ws[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}
}  // namespace

extern "C" int t2v_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
    T2V_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, T2V_EINVAL, "t2v_adamw_step");
    T2V_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, T2V_ESHAPE,
                "t2v_adamw_step: buffers must be 16-byte aligned");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblk((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_ema_update(float* target, const float* src, float rate, long long n, void* stream) {
    T2V_REQUIRE(target && src && n > 0, T2V_EINVAL, "t2v_ema_update");
    hipLaunchKernelGGL(ema_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, target, src, rate, n);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
extern "C" int t2v_sumsq(const float* x, long long n, float* ws, float* out, void* stream) {
    T2V_REQUIRE(x && ws && out && n > 0, T2V_EINVAL, "t2v_sumsq");
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, ws);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, blocks, out);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

// ---- library state ---------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
void t2v_set_error(const char* msg) {
    size_t i = 0;
    for (; msg && msg[i] && i + 1 < sizeof(g_err); ++i) g_err[i] = msg[i];
    g_err[i] = 0;
}
static void* g_zero[64] = {nullptr};
const void* t2v_zero_page() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_zero[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
        g_zero[dev] = p;
    }
    return g_zero[dev];
}
extern "C" const char* t2v_last_error(void) { return g_err; }
extern "C" int t2v_version(void) { return 100; }
extern "C" int t2v_init(void) { return t2v_zero_page() ? T2V_OK : T2V_EHIP; }
