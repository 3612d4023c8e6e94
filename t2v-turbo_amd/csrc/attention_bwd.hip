// Flash-style backward of the spatial self-attention (head dim 64): dQ, dK, dV from Q, K, V, dO and the forward's O, without ever
// materialising the [queries x keys] probability matrix (the GEMM-formulated backward of round 1 moves 1.05 GB per matrix and
// layer at the 2560-token level: ~70 GB per student step).  Two kernels with the forward kernel's
// structure (attention.hip: transposed scores, a lane owns one column, probabilities feed the next MFMA from registers):
//
//   attn_bwd_dq   a lane owns a QUERY.  Pass 1 over the key tiles: softmax statistics L = log2 sum_j exp2(c s_ij) (nothing is saved
//                 by the inference forward), D_i = sum_c dO_ic O_ic.  Pass 2: S^T = K Q^T, P^T = exp2(c S^T - L), dP^T = V dO^T,
//                 dS^T = P^T (dP^T - D), dQ^T += K^T dS^T.  Writes L and D for the second kernel.
//   attn_bwd_dkv  a lane owns a KEY; loop over query tiles: S = Q K^T, P = exp2(c S - L_i), dV^T += dO^T P, dP = dO V^T,
//                 dS = P (dP - D_i), dK^T += Q^T dS.
//
// Operands the caller provides next to the token-major rows: K^T, Q^T and dO^T per image ([heads*64][padded sequence], the
// layout of the forward's V^T buffer; t2v_transpose_pad_bf16 makes them).  Validated on MI355X (tests/test_gpu_unet_grad.py); also runs on the host SIMT
// simulator (tests/test_hostsim_attention_bwd.py) against the emulated backend.  The LDS tiles are double-buffered: the LDS-DMA of tile
// t+1 is issued before tile t's MFMAs and waited for (vmcnt(0) + barrier) after them (the first version waited right after issuing).
#include "common.h"

namespace {

constexpr int BT = 64;                 // rows per LDS tile (keys or queries)
constexpr int TILE_BYTES = BT * 128;   // [64][64] bf16, 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7

// -DT2V_ABWD_SINGLE (A/B builds): issue the loads of tile t+1 AFTER tile t's MFMAs and wait at once — the first version's exposure
#ifdef T2V_ABWD_SINGLE
#define ABWD_EARLY(x)
#define ABWD_LATE(x) x
#else
#define ABWD_EARLY(x) x
#define ABWD_LATE(x)
#endif
__device__ __forceinline__ void dma16b(const void* gsrc, char* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}
// LDS row r of a token-major tile holds token perm(r) = r with bits 2 and 3 swapped, so that the 8 accumulator registers a lane
// feeds into one K-step of the next MFMA are 8 CONSECUTIVE tokens = one 16-byte chunk of the transposed tile (attention.hip).
__device__ __forceinline__ int perm_row(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

struct TileLoader {
    int wave, lane, drow[2], dchunk[2];
    __device__ __forceinline__ void init(int w, int l) {
        wave = w; lane = l;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            drow[j] = (wave + 4 * j) * 8 + (lane >> 3);
            dchunk[j] = ((lane & 7) ^ ((drow[j] >> 1) & 7)) * 8;
        }
    }
    // rows = tokens tok0 + perm(r) of a token-major matrix (zero page past `seq`)
    __device__ __forceinline__ void rows(char* dst, const bf16_t* base, int ld, int tok0, int seq, const bf16_t* zero) const {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tok = tok0 + perm_row(drow[j]);
            dma16b(tok < seq ? base + (long long)tok * ld + dchunk[j] : zero, dst + (wave + 4 * j) * 1024);
        }
    }
    // rows = the 64 channels of a transposed matrix [64][ld_t], columns tok0 .. tok0 + 63 (the caller pads with zeros)
    __device__ __forceinline__ void cols(char* dst, const bf16_t* base_t, int ld_t, int tok0) const {
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16b(base_t + (long long)drow[j] * ld_t + tok0 + dchunk[j], dst + (wave + 4 * j) * 1024);
    }
};

__device__ __forceinline__ bf16x8_t frag(const char* tile, int row, int chunk, int swz) {
    return *(const bf16x8_t*)(tile + row * 128 + ((chunk ^ swz) << 4));
}
__device__ __forceinline__ bf16x8_t pack_b(const f32x16_t& s, int st) {  // 8 accumulator registers -> one B-operand fragment
    uint4 pu;
    pu.x = pack2bf(s[st * 8 + 0], s[st * 8 + 1]);
    pu.y = pack2bf(s[st * 8 + 2], s[st * 8 + 3]);
    pu.z = pack2bf(s[st * 8 + 4], s[st * 8 + 5]);
    pu.w = pack2bf(s[st * 8 + 6], s[st * 8 + 7]);
    return *(bf16x8_t*)&pu;
}

#ifndef T2V_ABWD_DQ_WPE
#define T2V_ABWD_DQ_WPE 3  // 163 registers, no scratch: three workgroups per CU (3 x 48 KiB of LDS); left to itself the compiler takes 208 for two
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(T2V_ABWD_DQ_WPE, T2V_ABWD_DQ_WPE))) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                          const bf16_t* __restrict__ v, int ldv, long long v_img_stride, long long v_head_stride,
                                                          const bf16_t* __restrict__ kt, int ld_kt, const bf16_t* __restrict__ dout, int ldo,
                                                          const bf16_t* __restrict__ o, int ldoo, float* __restrict__ l2, float* __restrict__ dsum,
                                                          int ld_stat, bf16_t* __restrict__ dq, int lddq, int seq_q, int seq_kv, int heads,
                                                          float scale, const bf16_t* zero) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 3 * TILE_BYTES];  // two stages of (K, V, K^T): tile t+1 lands under tile t's MFMAs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y, img = blockIdx.z;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = qi < seq_q;
    const float c2 = scale * 1.4426950408889634f;
    const int swz = (lane >> 1) & 7;
    TileLoader ld;
    ld.init(wave, lane);

    // Q and dO fragments (B operands: column = this lane's query, k = channels hi*8 + kk*16 .. +7) and D = sum_c dO O
    bf16x8_t qf[4], dof[4];
    float dloc = 0.f;
    {
        const long long row = (long long)img * seq_q + (q_ok ? qi : 0);
        const bf16_t* qp = q + row * ldq + head * 64 + hi * 8;
        const bf16_t* dp_ = dout + row * ldo + head * 64 + hi * 8;
        const bf16_t* op = o + row * ldoo + head * 64 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 uq = q_ok ? *(const uint4*)(qp + kk * 16) : make_uint4(0, 0, 0, 0);
            uint4 ud = q_ok ? *(const uint4*)(dp_ + kk * 16) : make_uint4(0, 0, 0, 0);
            uint4 uo = q_ok ? *(const uint4*)(op + kk * 16) : make_uint4(0, 0, 0, 0);
            qf[kk] = *(bf16x8_t*)&uq;
            dof[kk] = *(bf16x8_t*)&ud;
            float fd[8], fo[8];
            unpack8(ud, fd);
            unpack8(uo, fo);
#pragma unroll
            for (int e = 0; e < 8; ++e) dloc += fd[e] * fo[e];
        }
    }
    const float D = dloc + __shfl_xor(dloc, 32, 64);
    const bf16_t* kbase = k + (long long)img * seq_kv * ldk + head * 64;
    const bf16_t* vbase = v + img * v_img_stride + head * v_head_stride;
    const bf16_t* ktbase = kt + ((long long)img * heads + head) * 64 * ld_kt;
    const int ntile = (seq_kv + BT - 1) / BT;

    auto scores = [&](f32x16_t* s, const char* sk) {  // S^T = K Q^T for the staged key tile; register r of half h2 <-> key 8*hi + h2*32 + (r>>3)*16 + (r&7)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sk, h2 * 32 + l31, kk * 2 + hi, swz), qf[kk], s[h2], 0, 0, 0);
        }
    };
    // ---- pass 1: softmax statistics of this lane's query -----------------------------------------------------------------------
    float m_run = -INFINITY, l_run = 0.f;
    ld.rows(smem, kbase, ldk, 0, seq_kv, zero);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const char* sk = smem + (t & 1) * 3 * TILE_BYTES;
        ABWD_EARLY(if (t + 1 < ntile) ld.rows(smem + ((t + 1) & 1) * 3 * TILE_BYTES, kbase, ldk, (t + 1) * BT, seq_kv, zero);)
        f32x16_t s[2];
        scores(s, sk);
        const int key_base = t * BT + 8 * hi;
        float mloc = -INFINITY;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (key_base + h2 * 32 + (r >> 3) * 16 + (r & 7) >= seq_kv) s[h2][r] = -INFINITY;
                mloc = fmaxf(mloc, s[h2][r]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        float lsum = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum += __builtin_amdgcn_exp2f(fmaf(s[h2][r], c2, -m_new * c2));
        l_run = l_run * __builtin_amdgcn_exp2f((m_run - m_new) * c2) + lsum;
        m_run = m_new;
        ABWD_LATE(if (t + 1 < ntile) ld.rows(smem + ((t + 1) & 1) * 3 * TILE_BYTES, kbase, ldk, (t + 1) * BT, seq_kv, zero);)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t+1 has landed; every wave is done with tile t
        __syncthreads();
    }
    const float L = m_run * c2 + log2f(l_run + __shfl_xor(l_run, 32, 64));
    if (q_ok && hi == 0) {
        const long long si = ((long long)img * heads + head) * ld_stat + qi;
        l2[si] = L;
        dsum[si] = D;
    }
    // ---- pass 2: dQ^T += K^T dS^T -------------------------------------------------------------------------------------------------
    f32x16_t acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    auto stage = [&](int t) {
        char* b = smem + (t & 1) * 3 * TILE_BYTES;
        ld.rows(b, kbase, ldk, t * BT, seq_kv, zero);
        ld.rows(b + TILE_BYTES, vbase, ldv, t * BT, seq_kv, zero);
        ld.cols(b + 2 * TILE_BYTES, ktbase, ld_kt, t * BT);
    };
    stage(0);   // (pass 1 ended on a barrier: both stages are free)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const char* sk = smem + (t & 1) * 3 * TILE_BYTES;
        const char* sv = sk + TILE_BYTES;
        const char* skt = sk + 2 * TILE_BYTES;
        ABWD_EARLY(if (t + 1 < ntile) stage(t + 1);)
        f32x16_t s[2], dp[2];
        scores(s, sk);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[h2][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)  // dP^T = V dO^T
                dp[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sv, h2 * 32 + l31, kk * 2 + hi, swz), dof[kk], dp[h2], 0, 0, 0);
        }
        const int key_base = t * BT + 8 * hi;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = key_base + h2 * 32 + (r >> 3) * 16 + (r & 7) < seq_kv;
                const float p = __builtin_amdgcn_exp2f(fmaf(s[h2][r], c2, -L));
                s[h2][r] = ok ? p * (dp[h2][r] - D) : 0.f;  // dS^T
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t pb = pack_b(s[ks >> 1], ks & 1);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(skt, db * 32 + l31, ks * 2 + hi, swz), pb, acc[db], 0, 0, 0);
        }
        ABWD_LATE(if (t + 1 < ntile) stage(t + 1);)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (q_ok) {
        bf16_t* op = dq + ((long long)img * seq_q + qi) * lddq + head * 64 + 4 * hi;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(acc[db][g * 4 + 0] * scale, acc[db][g * 4 + 1] * scale);
                w.y = pack2bf(acc[db][g * 4 + 2] * scale, acc[db][g * 4 + 3] * scale);
                *(uint2*)(op + db * 32 + g * 8) = w;
            }
    }
}

#ifndef T2V_ABWD_DKV_WPE
#define T2V_ABWD_DKV_WPE 2  // waves per SIMD the register budget of the dK / dV kernel is sized for: 256 registers and 20 bytes of scratch per lane instead of 320 and none, and 26 % faster (1072 vs 1345 us at 16 x 2560 x 5 heads, profiles/r03_attn_bwd_ab.csv); -DT2V_ABWD_DKV_WPE=1 for A/B builds
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(T2V_ABWD_DKV_WPE, T2V_ABWD_DKV_WPE))) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                           const bf16_t* __restrict__ v, int ldv, long long v_img_stride, long long v_head_stride,
                                                           const bf16_t* __restrict__ qt, const bf16_t* __restrict__ dot, int ld_qt,
                                                           const bf16_t* __restrict__ dout, int ldo, const float* __restrict__ l2,
                                                           const float* __restrict__ dsum, int ld_stat, bf16_t* __restrict__ dk, int lddk,
                                                           bf16_t* __restrict__ dv, int lddv, int seq_q, int seq_kv, int heads, float scale,
                                                           const bf16_t* zero) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 4 * TILE_BYTES];  // two stages of (Q, dO, Q^T, dO^T)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y, img = blockIdx.z;
    const int kj = blockIdx.x * 128 + wave * 32 + l31;
    const bool k_ok = kj < seq_kv;
    const float c2 = scale * 1.4426950408889634f;
    const int swz = (lane >> 1) & 7;
    TileLoader ld;
    ld.init(wave, lane);
    bf16x8_t kf[4], vf[4];  // this lane's key as B operands (column = key, k = channels)
    {
        const bf16_t* kp = k + ((long long)img * seq_kv + (k_ok ? kj : 0)) * ldk + head * 64 + hi * 8;
        const bf16_t* vp = v + img * v_img_stride + head * v_head_stride + (long long)(k_ok ? kj : 0) * ldv + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 uk = k_ok ? *(const uint4*)(kp + kk * 16) : make_uint4(0, 0, 0, 0);
            uint4 uv = k_ok ? *(const uint4*)(vp + kk * 16) : make_uint4(0, 0, 0, 0);
            kf[kk] = *(bf16x8_t*)&uk;
            vf[kk] = *(bf16x8_t*)&uv;
        }
    }
    const bf16_t* qbase = q + (long long)img * seq_q * ldq + head * 64;
    const bf16_t* dobase = dout + (long long)img * seq_q * ldo + head * 64;
    const bf16_t* qtbase = qt + ((long long)img * heads + head) * 64 * ld_qt;
    const bf16_t* dotbase = dot + ((long long)img * heads + head) * 64 * ld_qt;
    const float* lrow = l2 + ((long long)img * heads + head) * ld_stat;
    const float* drow_ = dsum + ((long long)img * heads + head) * ld_stat;
    f32x16_t ak[2], av[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ak[0][r] = 0.f; ak[1][r] = 0.f; av[0][r] = 0.f; av[1][r] = 0.f; }
    const int ntile = (seq_q + BT - 1) / BT;
    auto stage = [&](int t) {
        char* b = smem + (t & 1) * 4 * TILE_BYTES;
        ld.rows(b, qbase, ldq, t * BT, seq_q, zero);
        ld.rows(b + TILE_BYTES, dobase, ldo, t * BT, seq_q, zero);
        ld.cols(b + 2 * TILE_BYTES, qtbase, ld_qt, t * BT);
        ld.cols(b + 3 * TILE_BYTES, dotbase, ld_qt, t * BT);
    };
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const char* sq = smem + (t & 1) * 4 * TILE_BYTES;
        const char* sdo = sq + TILE_BYTES;
        const char* sqt = sq + 2 * TILE_BYTES;
        const char* sdot = sq + 3 * TILE_BYTES;
        ABWD_EARLY(if (t + 1 < ntile) stage(t + 1);)
        f32x16_t s[2], dp[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[h2][r] = 0.f; dp[h2][r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {  // S = Q K^T and dP = dO V^T: rows = the tile's queries, column = this lane's key
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sq, h2 * 32 + l31, kk * 2 + hi, swz), kf[kk], s[h2], 0, 0, 0);
                dp[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sdo, h2 * 32 + l31, kk * 2 + hi, swz), vf[kk], dp[h2], 0, 0, 0);
            }
        }
        // register r of half h2 <-> query t*64 + 8*hi + h2*32 + (r>>3)*16 + (r&7): statistics of 8 consecutive queries per group
        const int q_base = t * BT + 8 * hi;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int qq0 = q_base + h2 * 32 + st * 16;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = st * 8 + e;
                    const bool ok = qq0 + e < seq_q;
                    const float Lq = ok ? lrow[qq0 + e] : 0.f, Dq = ok ? drow_[qq0 + e] : 0.f;
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[h2][r], c2, -Lq)) : 0.f;
                    s[h2][r] = p;                            // P
                    dp[h2][r] = ok ? p * (dp[h2][r] - Dq) : 0.f;  // dS
                }
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t pb = pack_b(s[ks >> 1], ks & 1), db_ = pack_b(dp[ks >> 1], ks & 1);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                av[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sdot, cb * 32 + l31, ks * 2 + hi, swz), pb, av[cb], 0, 0, 0);
                ak[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sqt, cb * 32 + l31, ks * 2 + hi, swz), db_, ak[cb], 0, 0, 0);
            }
        }
        ABWD_LATE(if (t + 1 < ntile) stage(t + 1);)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (k_ok) {
        bf16_t* pk_ = dk + ((long long)img * seq_kv + kj) * lddk + head * 64 + 4 * hi;
        bf16_t* pv_ = dv + ((long long)img * seq_kv + kj) * lddv + head * 64 + 4 * hi;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(ak[cb][g * 4 + 0] * scale, ak[cb][g * 4 + 1] * scale);
                w.y = pack2bf(ak[cb][g * 4 + 2] * scale, ak[cb][g * 4 + 3] * scale);
                *(uint2*)(pk_ + cb * 32 + g * 8) = w;
                w.x = pack2bf(av[cb][g * 4 + 0], av[cb][g * 4 + 1]);
                w.y = pack2bf(av[cb][g * 4 + 2], av[cb][g * 4 + 3]);
                *(uint2*)(pv_ + cb * 32 + g * 8) = w;
            }
    }
}

}  // namespace

extern "C" int t2v_attn_spatial_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, long long v_img_stride,
                                    long long v_head_stride, const void* kt, int ld_kt, const void* qt, const void* dot, int ld_qt,
                                    const void* dout, int ldo, const void* o, int ldoo, float* l2, float* dsum, int ld_stat, void* dq,
                                    int lddq, void* dk, int lddk, void* dv, int lddv, int n_img, int seq_q, int seq_kv, int heads,
                                    float scale, void* stream) {
    T2V_REQUIRE(q && k && v && kt && qt && dot && dout && o && l2 && dsum && dq && dk && dv, T2V_EINVAL, "t2v_attn_spatial_bwd: null pointer");
    T2V_REQUIRE(n_img > 0 && seq_q > 0 && seq_kv > 0 && heads > 0 && heads <= 65535 && n_img <= 65535, T2V_EINVAL, "t2v_attn_spatial_bwd: bad size");
    const int kvp = (seq_kv + 63) / 64 * 64, qp = (seq_q + 63) / 64 * 64;
    T2V_REQUIRE(ld_kt >= kvp && ld_qt >= qp && ld_stat >= seq_q, T2V_ESHAPE, "t2v_attn_spatial_bwd: transposed operands must be padded to 64");
    T2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ld_kt % 8 == 0 && ld_qt % 8 == 0 && ldo % 8 == 0 && ldoo % 8 == 0 &&
                lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 && v_img_stride % 8 == 0 && v_head_stride % 8 == 0, T2V_ESHAPE,
                "t2v_attn_spatial_bwd: row strides");
    const bf16_t* zero = (const bf16_t*)t2v_zero_page();
    T2V_REQUIRE(zero, T2V_EHIP, "t2v_attn_spatial_bwd: zero page");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((seq_q + 127) / 128, heads, n_img), dim3(256), 0, s, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)v, ldv, v_img_stride, v_head_stride, (const bf16_t*)kt, ld_kt, (const bf16_t*)dout, ldo, (const bf16_t*)o,
                       ldoo, l2, dsum, ld_stat, (bf16_t*)dq, lddq, seq_q, seq_kv, heads, scale, zero);
    T2V_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((seq_kv + 127) / 128, heads, n_img), dim3(256), 0, s, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)v, ldv, v_img_stride, v_head_stride, (const bf16_t*)qt, (const bf16_t*)dot, ld_qt, (const bf16_t*)dout, ldo,
                       (const float*)l2, (const float*)dsum, ld_stat, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, seq_q, seq_kv, heads, scale, zero);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
}
