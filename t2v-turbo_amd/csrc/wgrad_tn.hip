// Token-contracted weight-gradient product  out[R][C] = alpha * sum_m a[m][r] * b[m][c]   (A^T B) for two TOKEN-MAJOR bf16 operands:
// the LoRA weight gradients dU = s dy^T t and dD = G^T x, and the per-clip column sums, without first transposing both operands to
// K-contiguous copies for t2v_gemm (1 954 t2v_transpose_pad_bf16 launches and 48 GB per student step in the launch census,
// profiles/r01_student_step_census.txt).
//
// 64 x 64 output tile per workgroup (4 waves, 32 x 32 each; 128 x 128 / 64 x 64 per wave where both extents reach 128: round 6, the
// base-weight gradients of full fine-tuning), the token range split over blockIdx.z; per step 64 tokens of both operands
// go to LDS row-major (16-byte global loads), and each lane gathers its MFMA fragments — 8 consecutive TOKENS of one column — with
// 2-byte LDS reads down a column (row pitch 66 elements = 33 words: the 32 lanes of a half-wave read 32 consecutive columns of one
// row).  fp32 partial tiles go to a workspace; a second kernel adds them in a fixed order (deterministic) and applies alpha.
// These products are bandwidth-bound (2 M R C FLOPs on M (R + C) 2 bytes: 1.7 GFLOP on 31 MB for dU at the 320-channel level), so
// what matters is keeping loads in flight and the partial slabs small: the global loads of token step i+1 are issued into
// registers BEFORE the LDS phase of step i (one workgroup then covers its own latency), the token range is split over about two
// workgroups per CU (the partial slabs were as large as the operands when every CU got four), and the reduce kernel spreads the
// splits over four thread groups per output (fixed assignment, fixed combine order: still deterministic).
// Round 4: the column gather is done by ds_read_b64_tr_b16 (two reads per operand and MFMA instead of sixteen 2-byte reads and their
// packing: the counters of the 2-byte version showed the LDS 40 % busy with 18 LDS and 17 VALU instructions per MFMA,
// profiles/r04_wgrad_pmc.csv).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int WT_TOK = 64;
// LDS token-row pitch of a 64 T-column operand tile: 96 / 160 elements = 48 / 80 words, both = 16 (mod 32) words past the tile, so that four
// consecutive token rows start 16 banks apart: see lds_read_tr16 / wgrad_tile
template <int T> struct WtGeom { static constexpr int TILE = 64 * T, PITCH = 64 * T + 32; };

// ds_read_b64_tr_b16 (gfx950): every lane supplies the LDS address of FOUR contiguous 16-bit elements (8-byte aligned); within each
// group of 16 lanes, lane l receives element (l & 3) of the values read by lanes (l >> 2), 4 + (l >> 2), 8 + (l >> 2), 12 + (l >> 2)
// (measured on MI355X: tools/probes/tr_b16_probe.cpp).  With lane i of a group pointing at [token t0 + (i >> 2)][column c0 + 4 (i & 3)]
// of a token-major tile, lane l gets tokens t0 .. t0 + 3 of column c0 + l: half an MFMA operand fragment.
typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const bf16_t* p, int lane) {
#ifdef T2V_HOSTSIM
    uint64_t mine;
    memcpy(&mine, p, 8);
    uint64_t out = 0;
    for (int j = 0; j < 4; ++j) {
        const uint64_t v = __shfl(mine, (lane & ~15) + 4 * j + ((lane & 15) >> 2), 64);
        out |= ((v >> (16 * (lane & 3))) & 0xffffull) << (16 * j);
    }
    return make_uint2((uint32_t)out, (uint32_t)(out >> 32));
#else
    (void)lane;
    const v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
    return *(const uint2*)&v;
#endif
}

// the first `n` (< 8, possibly <= 0) elements of an 8-element chunk, zeros behind them (the last column chunk of a ragged operand)
__device__ __forceinline__ uint4 ragged_chunk(const bf16_t* p, int n) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < n) w[e >> 1] |= (uint32_t)p[e] << (16 * (e & 1));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  The tiles of one token range share
// operand rows (every r-tile of dU reads the same rows of t; dD's tiles share G and x), so they should meet in ONE L2: `xcd_contiguous`
// renumbers the blocks so that consecutive ids run on the same XCD (common.h; the bijective form of csrc/gemm.hip).  Measured before it
// (profiles/r04_wgrad_pmc.csv): 94 % of the kernel's L2 requests missed, 419 MB per launch on 135 MB of operands, i.e. the kernel ran at
// the memory side's 6 TB/s on 3x the bytes.

// one 64 T x 64 T output tile (T = 1, 2) over the token range [m_begin, m_end): partial sums x `scale` into `slab` (fp32 rows of pitch
// `ld_slab`: a [R][C] partial slab with scale 1, or — one token split — the output itself with scale alpha).  Four waves,
// 32 T x 32 T each (T x T accumulator blocks).  T = 2 is for the base-weight gradients of full fine-tuning (R, C in the hundreds to
// tens of thousands: engine_full.py), where the 64 x 64 tile's 32 FLOP per operand byte out of L2 is the bound (~ 300 TFLOP/s measured,
// profiles/r06_full_finetune_kernel_stats.csv): twice the reuse, half the LDS reads per MFMA, a quarter of the partial-slab tiles.
template <int T>
__device__ __forceinline__ void wgrad_tile(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b, int ldb, int R, int C, int r0,
                                           int c0, long long m_begin, long long m_end, float* __restrict__ slab, long long ld_slab, float scale,
                                           bf16_t (*sa)[WtGeom<T>::PITCH], bf16_t (*sb)[WtGeom<T>::PITCH]) {
    constexpr int PITCH = WtGeom<T>::PITCH, CPR = 8 * T, NLD = 2 * T;   // 16-byte chunks per token row of a tile / per thread and operand
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31, i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int wr = (wave >> 1) * 32 * T, wc = (wave & 1) * 32 * T;  // this wave's sub-tile
    f32x16_t acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // 64 tokens x 64 T columns of each operand per step: 512 T 16-byte chunks per operand, 2 T per thread
    uint4 ua[NLD], ub[NLD];
    auto gload = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int qd = tid + 256 * i, row = qd / CPR, ch = (qd % CPR) * 8;
            const long long m = m0 + row;
            ua[i] = make_uint4(0, 0, 0, 0); ub[i] = make_uint4(0, 0, 0, 0);
            if (m < m_end) {
                // (a chunk wholly behind the operand's last column — every chunk of a 128-wide tile's tail at R = 320 — stays zero without
                // passing through the element-wise path)
                if (r0 + ch + 8 <= R) ua[i] = *(const uint4*)(a + m * lda + r0 + ch);
                else if (r0 + ch < R) ua[i] = ragged_chunk(a + m * lda + r0 + ch, R - (r0 + ch));
                if (c0 + ch + 8 <= C) ub[i] = *(const uint4*)(b + m * ldb + c0 + ch);
                else if (c0 + ch < C) ub[i] = ragged_chunk(b + m * ldb + c0 + ch, C - (c0 + ch));
            }
        }
    };
    if (m_begin < m_end) gload(m_begin);
    for (long long m0 = m_begin; m0 < m_end; m0 += WT_TOK) {
        __syncthreads();  // every wave is done reading the previous step's tiles
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int qd = tid + 256 * i, row = qd / CPR, ch = (qd % CPR) * 8;
            *(uint4*)&sa[row][ch] = ua[i];   // (192- / 320-byte token rows: 16-byte aligned)
            *(uint4*)&sb[row][ch] = ub[i];
        }
        if (m0 + WT_TOK < m_end) gload(m0 + WT_TOK);  // next step's loads fly under this step's LDS phase
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // 16 tokens per MFMA: this lane's 8 are ks*16 + 8*hi + 0..7
            // transpose reads: lane (i16, g16) of a half-wave points at [token t0 + (i16 >> 2)][column 16 g16 + 4 (i16 & 3)] and gets
            // tokens t0 .. t0 + 3 (second read: + 4 .. + 7) of column 16 g16 + i16 = l31.  Token rows of 48 / 80 words: the 32 lanes of a
            // half-wave (4 token rows x 2 column groups x 8 words) cover the 64 banks once, the other half-wave is the second pass.
            const int t0 = ks * 16 + 8 * hi;
            uint4 fa[T], fb[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const bf16_t* pa = &sa[t0 + (i16 >> 2)][wr + 32 * i + 16 * g16 + 4 * (i16 & 3)];
                const bf16_t* pb = &sb[t0 + (i16 >> 2)][wc + 32 * i + 16 * g16 + 4 * (i16 & 3)];
                const uint2 a0 = lds_read_tr16(pa, lane), a1 = lds_read_tr16(pa + 4 * PITCH, lane);
                const uint2 b0 = lds_read_tr16(pb, lane), b1 = lds_read_tr16(pb + 4 * PITCH, lane);
                fa[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                fb[i] = make_uint4(b0.x, b0.y, b1.x, b1.y);
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&fa[i], *(bf16x8_t*)&fb[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[row = (e & 3) + 8 (e >> 2) + 4 hi][col = l31] of each 32 x 32 block of the wave's sub-tile -> partial slab [R][C]
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int c = c0 + wc + 32 * j + l31;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = r0 + wr + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (r < R) slab[(long long)r * ld_slab + c] = acc[i][j][e] * scale;
                }
            }
        }
}

template <int T>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b, int ldb, long long M,
                                                       int R, int C, long long tok_per_split, float* __restrict__ ws, long long ld_ws, float scale,
                                                       int y_fast) {
    __shared__ __attribute__((aligned(16))) bf16_t sa[WT_TOK][WtGeom<T>::PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t sb[WT_TOK][WtGeom<T>::PITCH];
    // (linear id = x fastest: the order the dispatcher walks the grid in)
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    const int id = xcd_contiguous((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, nb);
    // Consecutive ids (one XCD, resident together, walking the token range at the same pace) should cover a near-SQUARE patch of the tile
    // grid: a patch of h x w tiles streams h + w operand column blocks for h w tiles.  With the row of tiles as the fast axis a patch
    // was 1 x ~100: every tile its own block of the wide operand, re-read from the memory side once per tile row (counters, 640 x 5 760
    // over 10 240 tokens: 1.17 GB fetched for 131 MB of operands, profiles/r06_wgrad_tile_pmc.csv) — so the SHORTER grid axis runs fastest.
    const int gx = gridDim.x, gy = gridDim.y;
    const int bx = y_fast ? (id / gy) % gx : id % gx, by = y_fast ? id % gy : (id / gx) % gy, bz = id / (gx * gy);
    const long long m_begin = (long long)bz * tok_per_split;
    const long long m_end = m_begin + tok_per_split < M ? m_begin + tok_per_split : M;
    wgrad_tile<T>(a, lda, b, ldb, R, C, by * WtGeom<T>::TILE, bx * WtGeom<T>::TILE, m_begin, m_end, ws + (long long)bz * R * C, ld_ws, scale, sa, sb);
}

// Up to T2V_WGRAD_GROUP_MAX independent products in ONE launch (the weight gradients of one LoRA group: dU of each of its leaves
// and dD, whose operands all exist once the rank-r gradient g is there): a workgroup finds its problem by a scan over the block
// prefix sums in the kernel argument, then works exactly like wgrad_tn_kernel.  One launch fills the chip where three or four
// small ones each had to split their token range a hundred ways to do so (fewer, longer workgroups: a fifth of the partial slabs).
struct WgradGroup {
    const bf16_t* a[T2V_WGRAD_GROUP_MAX];
    const bf16_t* b[T2V_WGRAD_GROUP_MAX];
    float* out[T2V_WGRAD_GROUP_MAX];
    long long M[T2V_WGRAD_GROUP_MAX], tok_per_split[T2V_WGRAD_GROUP_MAX], ws_off[T2V_WGRAD_GROUP_MAX];
    int lda[T2V_WGRAD_GROUP_MAX], ldb[T2V_WGRAD_GROUP_MAX], ldo[T2V_WGRAD_GROUP_MAX], R[T2V_WGRAD_GROUP_MAX], C[T2V_WGRAD_GROUP_MAX];
    int splits[T2V_WGRAD_GROUP_MAX], tiles_c[T2V_WGRAD_GROUP_MAX];
    float alpha[T2V_WGRAD_GROUP_MAX];
    int first_block[T2V_WGRAD_GROUP_MAX + 1];   // main kernel: tiles * splits per problem
    int first_rblock[T2V_WGRAD_GROUP_MAX + 1];  // reduce kernel: ceil(R * C / 64) per problem
    int n;
};

__global__ __launch_bounds__(256) void wgrad_tn_group_kernel(const WgradGroup g, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) bf16_t sa[WT_TOK][WtGeom<1>::PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t sb[WT_TOK][WtGeom<1>::PITCH];
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    int i = 0;
    while (i + 1 < g.n && bid >= g.first_block[i + 1]) ++i;   // block-uniform
    const int local = bid - g.first_block[i];
    const int tiles_c = g.tiles_c[i], tiles = tiles_c * ((g.R[i] + 63) / 64);
    const int split = local / tiles, tile = local - split * tiles;
    const long long m_begin = (long long)split * g.tok_per_split[i];
    const long long m_end = m_begin + g.tok_per_split[i] < g.M[i] ? m_begin + g.tok_per_split[i] : g.M[i];
    wgrad_tile<1>(g.a[i], g.lda[i], g.b[i], g.ldb[i], g.R[i], g.C[i], (tile / tiles_c) * 64, (tile % tiles_c) * 64, m_begin, m_end,
               ws + g.ws_off[i] + (long long)split * g.R[i] * g.C[i], g.C[i], 1.0f, sa, sb);
}

// out = alpha * sum over splits, fixed order: thread (o, g) adds the splits k = g, g + 4, g + 8, ... of output o (four loads in
// flight each), then the four group sums are combined in the order g = 0..3.
__global__ __launch_bounds__(256) void wgrad_tn_reduce_kernel(const float* __restrict__ ws, int splits, int R, int C, float alpha,
                                                              float* __restrict__ out, int ldo) {
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long RC = (long long)R * C, idx = (long long)blockIdx.x * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < RC) {
        const float* p = ws + idx;
        int k = g;
        for (; k + 12 < splits; k += 16) {
            const float t0 = p[(long long)k * RC], t1 = p[(long long)(k + 4) * RC], t2 = p[(long long)(k + 8) * RC], t3 = p[(long long)(k + 12) * RC];
            s0 += t0; s1 += t1; s2 += t2; s3 += t3;
        }
        for (; k < splits; k += 4) s0 += p[(long long)k * RC];
    }
    part[g][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && idx < RC) {
        const float s = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
        out[(idx / C) * ldo + idx % C] = s * alpha;
    }
}

// The same sum for four consecutive outputs per thread (C % 4 == 0, 16-byte aligned rows): the order of wgrad_tn_reduce_kernel — group g adds
// the splits k = g, g + 4, ... into four running sums, ((g0 + g1) + g2) + g3 — evaluated by ONE thread with float4 loads, so bit-identical
// to it.  For the megabyte-sized outputs of full fine-tuning, where the 64-output blocks of the first kernel were a launch of 10^5-10^6
// workgroups with one or two loads per thread (10.3 ms per step against 30 ms for the products, profiles/r06_full_finetune_kernel_stats.csv).
__global__ __launch_bounds__(256) void wgrad_tn_reduce4_kernel(const float* __restrict__ ws, int splits, int R, int C, float alpha,
                                                               float* __restrict__ out, int ldo) {
    const long long RC = (long long)R * C, idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= RC) return;
    const float* p = ws + idx;
    float4 part[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int k = g;
        for (; k + 12 < splits; k += 16) {
            const float4 t0 = *(const float4*)(p + (long long)k * RC), t1 = *(const float4*)(p + (long long)(k + 4) * RC);
            const float4 t2 = *(const float4*)(p + (long long)(k + 8) * RC), t3 = *(const float4*)(p + (long long)(k + 12) * RC);
            s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
            s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
            s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
            s3.x += t3.x; s3.y += t3.y; s3.z += t3.z; s3.w += t3.w;
        }
        for (; k < splits; k += 4) {
            const float4 t0 = *(const float4*)(p + (long long)k * RC);
            s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
        }
        part[g] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    }
    float4 o;
    o.x = (((part[0].x + part[1].x) + part[2].x) + part[3].x) * alpha;
    o.y = (((part[0].y + part[1].y) + part[2].y) + part[3].y) * alpha;
    o.z = (((part[0].z + part[1].z) + part[2].z) + part[3].z) * alpha;
    o.w = (((part[0].w + part[1].w) + part[2].w) + part[3].w) * alpha;
    *(float4*)(out + (idx / C) * ldo + idx % C) = o;
}

__global__ __launch_bounds__(256) void wgrad_tn_group_reduce_kernel(const WgradGroup g, const float* __restrict__ ws) {
    __shared__ float part[4][64];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.first_rblock[i + 1]) ++i;
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int R = g.R[i], C = g.C[i], splits = g.splits[i];
    const long long RC = (long long)R * C, idx = (long long)(blockIdx.x - g.first_rblock[i]) * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < RC) {   // the same fixed summation order as wgrad_tn_reduce_kernel
        const float* p = ws + g.ws_off[i] + idx;
        int k = grp;
        for (; k + 12 < splits; k += 16) {
            const float t0 = p[(long long)k * RC], t1 = p[(long long)(k + 4) * RC], t2 = p[(long long)(k + 8) * RC], t3 = p[(long long)(k + 12) * RC];
            s0 += t0; s1 += t1; s2 += t2; s3 += t3;
        }
        for (; k < splits; k += 4) s0 += p[(long long)k * RC];
    }
    part[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && idx < RC) {
        const float s = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
        g.out[i][(idx / C) * g.ldo[i] + idx % C] = s * g.alpha[i];
    }
}

}  // namespace

extern "C" int t2v_wgrad_tn(const void* a, int lda, const void* b, int ldb, long long M, int R, int C, float alpha, float* out, int ldo,
                            float* ws, long long ws_bytes, int splits, void* stream) {
    T2V_REQUIRE(a && b && out && ws && M > 0 && R > 0 && C > 0 && ldo >= C, T2V_EINVAL, "t2v_wgrad_tn: bad argument");
    T2V_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0, T2V_ESHAPE,
                "t2v_wgrad_tn: 16-byte aligned operand rows");
    const long long steps = (M + WT_TOK - 1) / WT_TOK;
    // 128 x 128 tiles where both output extents have at least one (the base-weight gradients of full fine-tuning); the LoRA products
    // (rank 64 on one side) and the per-clip column sums keep the 64 x 64 tile
    static const bool big_ok = getenv("T2V_WGRAD_TILE128") == nullptr || atoi(getenv("T2V_WGRAD_TILE128")) != 0;
    const long long pad128 = (long long)((R + 127) / 128 * 128) * ((C + 127) / 128 * 128);
    int tile = (big_ok && R >= 128 && C >= 128 && pad128 * 10 <= (long long)R * C * 11) ? 128 : 64;   // (not at 320: 384 x 384 of tiles for 320 x 320)
    if (tile == 128 && splits <= 0) {
        // ... and not where the large tiles would have to split a SHORT token range to fill the chip (640 tokens at the 5 x 8 level: 31 us
        // with 128 x 128 tiles in two splits of 5 steps + a reduction, 18 us with 64 x 64 tiles in one: profiles/r06_wgrad_tile128_by_shape.csv)
        const long long tiles128 = pad128 / (128 * 128), want = (512 + tiles128 - 1) / tiles128;
        if (want > 1 && steps / want < 8) tile = 64;
    }
    const long long tiles = (long long)((R + tile - 1) / tile) * ((C + tile - 1) / tile);
    if (splits <= 0) {  // about two workgroups per CU: each covers its own load latency, and the partial slabs stay small
        splits = (int)((512 + tiles - 1) / tiles);
        if (splits > 256) splits = 256;
    }
    if (splits > steps) splits = (int)steps;
    const long long slab = (long long)R * C * 4;
    if ((long long)splits * slab > ws_bytes) splits = (int)(ws_bytes / slab);
    if (splits < 2) splits = 1;   // (one split needs no workspace: an output larger than the workspace is written directly)
    long long tok_per_split = ((steps + splits - 1) / splits) * WT_TOK;
    splits = (int)((M + tok_per_split - 1) / tok_per_split);
    T2V_REQUIRE((R + 63) / 64 <= 65535 && splits <= 65535, T2V_ESHAPE, "t2v_wgrad_tn: grid");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((C + tile - 1) / tile, (R + tile - 1) / tile, splits);
    // ONE token split: the tiles go straight to the output (x alpha), no partial slab and no second kernel
    float* dst = splits == 1 ? out : ws;
    const long long ld_slab = splits == 1 ? ldo : C;
    const float scale = splits == 1 ? alpha : 1.0f;
    static const bool yfast_on = getenv("T2V_WGRAD_YFAST") == nullptr || atoi(getenv("T2V_WGRAD_YFAST")) != 0;
    const int y_fast = yfast_on && grid.y <= grid.x;
    if (tile == 128)
        hipLaunchKernelGGL(wgrad_tn_kernel<2>, grid, dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, M, R, C, tok_per_split, dst,
                           ld_slab, scale, y_fast);
    else
        hipLaunchKernelGGL(wgrad_tn_kernel<1>, grid, dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, M, R, C, tok_per_split, dst,
                           ld_slab, scale, y_fast);
    T2V_CHECK_LAUNCH();
    if (splits == 1) return T2V_OK;
    if (C % 4 == 0 && ldo % 4 == 0 && (uintptr_t)out % 16 == 0 && (long long)R * C >= (1 << 16))
        hipLaunchKernelGGL(wgrad_tn_reduce4_kernel, dim3((unsigned)(((long long)R * C / 4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, splits, R,
                           C, alpha, out, ldo);
    else
        hipLaunchKernelGGL(wgrad_tn_reduce_kernel, dim3((unsigned)(((long long)R * C + 63) / 64)), dim3(256), 0, s, (const float*)ws, splits, R, C,
                           alpha, out, ldo);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}

extern "C" int t2v_wgrad_tn_group(const t2v_wgrad_problem* p, int n, float* ws, long long ws_bytes, void* stream) {
    T2V_REQUIRE(p && ws && n >= 1 && n <= T2V_WGRAD_GROUP_MAX, T2V_EINVAL, "t2v_wgrad_tn_group: 1..8 problems");
    WgradGroup g;
    long long total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        T2V_REQUIRE(p[i].a && p[i].b && p[i].out && p[i].M > 0 && p[i].R > 0 && p[i].C > 0 && p[i].ldo >= p[i].C, T2V_EINVAL,
                    "t2v_wgrad_tn_group: bad problem");
        T2V_REQUIRE(p[i].lda % 8 == 0 && p[i].ldb % 8 == 0 && (uintptr_t)p[i].a % 16 == 0 && (uintptr_t)p[i].b % 16 == 0, T2V_ESHAPE,
                    "t2v_wgrad_tn_group: 16-byte aligned operand rows");
        total_tiles += (long long)((p[i].R + 63) / 64) * ((p[i].C + 63) / 64);
    }
    // about two and a half workgroups per CU over the whole group; every problem gets the same number of token splits (their
    // token counts are equal or close), shrunk until the partial slabs fit the workspace
    long long want = (640 + total_tiles - 1) / total_tiles;
    if (want > 256) want = 256;
    for (;;) {
        long long off = 0, blocks = 0, rblocks = 0;
        for (int i = 0; i < n; ++i) {
            const long long steps = (p[i].M + WT_TOK - 1) / WT_TOK;
            long long sp = want < steps ? want : steps;
            const long long tps = ((steps + sp - 1) / sp) * WT_TOK;
            sp = (p[i].M + tps - 1) / tps;
            const int tiles_c = (p[i].C + 63) / 64, tiles = tiles_c * ((p[i].R + 63) / 64);
            g.a[i] = (const bf16_t*)p[i].a; g.b[i] = (const bf16_t*)p[i].b; g.out[i] = p[i].out;
            g.M[i] = p[i].M; g.tok_per_split[i] = tps; g.ws_off[i] = off;
            g.lda[i] = p[i].lda; g.ldb[i] = p[i].ldb; g.ldo[i] = p[i].ldo; g.R[i] = p[i].R; g.C[i] = p[i].C;
            g.splits[i] = (int)sp; g.tiles_c[i] = tiles_c; g.alpha[i] = p[i].alpha;
            g.first_block[i] = (int)blocks; g.first_rblock[i] = (int)rblocks;
            blocks += (long long)tiles * sp;
            rblocks += ((long long)p[i].R * p[i].C + 63) / 64;
            off += sp * (long long)p[i].R * p[i].C;
        }
        g.first_block[n] = (int)blocks; g.first_rblock[n] = (int)rblocks;
        g.n = n;
        if (off * 4 <= ws_bytes) break;
        T2V_REQUIRE(want > 1, T2V_ESHAPE, "t2v_wgrad_tn_group: workspace smaller than the outputs");
        want = want / 2;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_tn_group_kernel, dim3((unsigned)g.first_block[n]), dim3(256), 0, s, g, ws);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_tn_group_reduce_kernel, dim3((unsigned)g.first_rblock[n]), dim3(256), 0, s, g, (const float*)ws);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
