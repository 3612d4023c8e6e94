/*
 * t2v_hip.h — C-ABI of libt2v_hip.so: the MI355X (gfx950) kernels under the
 * t2v-turbo denoise hot path (VideoCrafter2 3D-UNet forward + KL-VAE decoder).
 *
 * The reference (Ji4chenLi/t2v-turbo) is 100 % Python and has no FFI layer; what
 * this library replaces is the set of ATen/cuDNN/cuBLAS/xformers calls its hot
 * path issues.  Each entry point cites the reference call site(s) it stands in
 * for (paths relative to the reference root).
 *
 * Conventions
 *  - plain C: raw device pointers, ints, a stream handle; no torch types.
 *  - activations are bf16, token-major ("NHWC"): row m = ((b*F + f)*H + y)*W + x,
 *    C contiguous channels per row (row stride `ld*` in elements).
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library
 *    allocates nothing per call, never synchronises and launches only on the
 *    stream it is given (hipStream_t passed as void*; NULL = default stream).
 *  - every function returns 0 on success, a negative T2V_E* code on a bad
 *    argument / unsupported shape, and never throws across the boundary.
 */
#ifndef T2V_HIP_H
#define T2V_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define T2V_OK 0
#define T2V_EINVAL (-1)   /* bad argument */
#define T2V_ESHAPE (-2)   /* shape/alignment not supported by the kernel */
#define T2V_EHIP (-3)     /* HIP runtime error at launch */

/* element types of boundary tensors */
#define T2V_F32 0
#define T2V_BF16 1
#define T2V_F16 2

/* library version / build info; also forces lazy per-device state (16-byte zero page
 * used for conv zero padding) to exist before a stream capture starts. */
int t2v_version(void);
int t2v_init(void);
const char* t2v_last_error(void);

/* ---------------------------------------------------------------- GEMM / implicit-GEMM conv
 * out[M,N] = epilogue( alpha * gather(A)[M,K] * W[N,K]^T )
 *
 * Replaces: nn.Linear (lvdm/modules/attention.py:71-76,348,370,433,468,519,537-538;
 * openaimodel3d.py:403-430,172-178), nn.Conv2d 3x3 / 3x3-s2 / 1x1
 * (openaimodel3d.py:155-159,179-184,63-72,98-100,192-193,669; ae_modules.py:149-175,550-552),
 * nn.Conv3d (3,1,1) (openaimodel3d.py:274-296), nn.Conv1d k=1 (attention.py:425-431),
 * F.interpolate(nearest x2)+conv (openaimodel3d.py:104-111; ae_modules.py:118-121),
 * torch.cat skip concat feeding a conv (openaimodel3d.py:733), and the bmm pairs of the
 * VAE AttnBlock (ae_modules.py:59-69).
 */
#define T2V_GEMM_LINEAR 0      /* K = Cin; A row m is source row m                          */
#define T2V_GEMM_CONV3X3 1     /* 3x3, stride 1, pad 1; K = 9*Cin, tap-major (ky,kx) then c */
#define T2V_GEMM_CONV3X3_S2 2  /* 3x3, stride 2, pad 1                                      */
#define T2V_GEMM_CONV3X3_UP2 3 /* 3x3 s1 p1 over the nearest-x2 upsampled input (never materialised) */
#define T2V_GEMM_TCONV3 4      /* (3,1,1) temporal conv, pad (1,0,0); K = 3*Cin, tap = frame offset   */
#define T2V_GEMM_CONV3X3_S2_PAD01 5 /* 3x3 s2, pad right/bottom only (VAE encoder Downsample, ae_modules.py:98-102) */

#define T2V_ACT_NONE 0
#define T2V_ACT_GEGLU 1 /* out[m,j] = x*gelu(gate); W rows packed in 64-row groups [32 x | 32 gate]; out has N/2 cols */
#define T2V_ACT_SILU 2

typedef struct t2v_gemm_desc {
    /* A: bf16 activations; optional second source = virtual channel concat [a0 | a1] */
    const void* a0;
    const void* a1;
    int c0, c1;     /* channels per source (c1 = 0 without a1); c0, c1 multiples of 64 */
    int lda0, lda1; /* row strides, elements */
    int mode;       /* T2V_GEMM_* */
    int n_img, h_in, w_in; /* input grid for conv modes (n_img = B*F) */
    int frames;     /* TCONV3: frames per clip */
    int M, N;       /* output rows / cols (LINEAR: M = source rows) */
    /* W: bf16 [N][K], row stride ldw (elements), K = taps*(c0+c1) */
    const void* w;
    int ldw;
    /* batching: z in [0,batch): z0 = z / batch_inner, z1 = z % batch_inner; strides in elements */
    int batch, batch_inner;
    long long a_stride0, a_stride1, w_stride0, w_stride1, o_stride0, o_stride1;
    /* epilogue */
    float alpha;
    const float* bias;   /* [N] fp32 or NULL */
    const float* rowvec; /* fp32: += rowvec[(m / rowvec_div) * ld_rowvec + n], or NULL (time-embedding add) */
    int rowvec_div, ld_rowvec;
    const void* residual; /* bf16 [M][ldr] (batched with the o_strides), or NULL */
    int ldr;
    int act;     /* T2V_ACT_* */
    void* out;   /* bf16 (or fp32 if out_f32) [M][ldo] */
    int ldo;
    int out_f32;
    /* scheduling hints (0 = library heuristic): workgroup tile id and split-K factor.  Split-K needs a
     * caller workspace of >= split_k*batch*M*N*4 bytes; without one the library never splits. */
    int tile_cfg, split_k;
    void* ws;
    long long ws_bytes;
    /* dropout between the product and the residual (0 threshold = off): out = keep ? (alpha*acc + bias) / (1 - p) : 0, then
     * + rowvec + residual.  The mask is t2v_dropout_bf16's: one splitmix64 word per QUAD of adjacent elements of a
     * row-major [rows][drop_ncols] matrix in which this launch's output starts at column drop_col0 (both multiples of 4), 16 bits per
     * element against drop_thr >> 16 (drop_thr = p * 2^32): p is resolved to 2^-16, so the drop probability is p' = (drop_thr >> 16) / 65536
     * and the caller passes drop_inv_keep = 1 / (1 - p') = 65536 / (65536 - (drop_thr >> 16)) (E[out] = in exactly; p < 2^-16 keeps
     * everything at scale 1); the 64-bit seed is read from device memory.  LoraInjected*.forward's dropout(up(down(x))) * scale
     * (utils/lora.py:45-50) as the epilogue of the up-projection.  Not combined with GEGLU / split-K. */
    const void* drop_seed;
    unsigned drop_thr, drop_site;
    float drop_inv_keep;
    int drop_ncols, drop_col0;
    /* LayerNorm second output (NULL = off): ln_out[m][n] = (out[m][n] - mean_m) * rstd_m * ln_gamma[n] + ln_beta[n] over the N
     * columns of the row (fp32 statistics on the fp32 epilogue values, two-pass), bf16 [M][ld_ln_out].  The pre-LN residual
     * stream and its LayerNorm (BasicTransformerBlock._forward, attention.py:300-311) from one launch.  N == 320 only (the
     * 160x320 workgroup tile holds whole rows); no batch, no activation, bf16 out, alpha == 1. */
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    void* ln_out;
    int ld_ln_out;
    /* ---- normalisation statistics as by-products of the producing GEMM, LayerNorm folded into the consuming one ----------
     * (all NULL = off; honoured by the fast kernels only: ask t2v_gemm_fuse_supported first, t2v_gemm refuses otherwise)
     *
     * rowstat_out [M][ld_rowstat] fp32: for every output row and every 32-column block b the pair (sum, sum of squares) of the
     * fp32 epilogue values at floats [2b, 2b+1] — the row statistics the NEXT LayerNorm needs (attention.py:300-311), written
     * by the GEMM that produces its input instead of a pass over the tensor.  N % 32 == 0, ld_rowstat >= N/16, % 4 == 0.
     *
     * colstat_out [M/32][N][2] fp32: for every 32-row slab and every column (sum, sum of squares) of the bf16-ROUNDED outputs —
     * what GroupNorm (openaimodel3d.py:223-254, lvdm/basics.py:78-89) reduces per (unit, group); t2v_group_norm_cs finishes
     * them for any unit whose row count is a multiple of 32.  M % 32 == 0.
     *
     * lnf_stats: this launch CONSUMES LayerNorm(x) without x ever being normalised in memory: A holds the raw rows x [M][C],
     * W holds W diag(gamma) (bf16), lnf_s[n] = sum_c W'[n][c] (fp32, of the bf16-rounded W'), bias[n] = b[n] + sum_c W[n][c]
     * beta[c]; with (mean_m, rstd_m) from the producer's rowstat (lnf_nblk blocks of 32 columns, C = 32 lnf_nblk):
     *     out[m][n] = rstd_m * (acc[m][n] - mean_m * lnf_s[n]) + bias[n]      (then GEGLU / activation as usual)
     * LINEAR mode, no batch, no residual / rowvec / dropout, alpha == 1. */
    float* rowstat_out;
    int ld_rowstat;
    float* colstat_out;
    const float* lnf_stats;
    int lnf_ld, lnf_nblk;
    float lnf_eps;
    const float* lnf_s;
    /* ---- LoRA branch in the base leaf's epilogue (NULL = off; fast kernels only: ask t2v_gemm_fuse_supported first) ------------
     * lora_t [M][ld_lora_t] bf16: the rank-64 down-projections t_l = x (*) D_l of the leaves this launch's N columns belong to,
     * leaf l (columns [l * lora_n_leaf, (l + 1) * lora_n_leaf)) at columns [64 l, 64 l + 64);  lora_u [N][ld_lora_u] bf16: row n
     * = the up-projection row of output channel n (rank zero-padded to 64).  The launch then computes
     *     out = epilogue( acc + bias + rowvec + residual + lora_scale * dropout( t_l u_n^T ) )
     * i.e. LoraInjected*.forward (utils/lora.py:45-50,124-129,204-209) in ONE launch after the down-projection: the M x N
     * up-projection z is never written or re-read.  With lora_t set, the drop_* fields mask the LoRA product only (drop_thr = 0:
     * no dropout); the mask is t2v_dropout_bf16's over the [M][drop_ncols] matrix, so it is bit-identical to the three-launch form.
     * lora_n_leaf % 32 == 0, no GEGLU / batch / split-K, alpha == 1, bf16 out.
     * Validated on the host SIMT simulator (tests/test_hostsim_gemm_fuse.py) and, in the last seconds of round 3's GPU budget, on
     * MI355X (tests/test_gpu_gemm_fuse.py at the UNet's shapes; the gradient engine at tiny width against the reference's LoRA
     * gradients and with replayed train-mode masks).  NOT YET TIMED: the engines use it only with T2V_LORA_EPILOGUE=1. */
    const void* lora_t;
    int ld_lora_t;
    const void* lora_u;
    int ld_lora_u;
    int lora_n_leaf;
    float lora_scale;
    /* ---- LayerNorm of the INPUT rows, applied while the activation panel is filled (t2v_linear_pr only; 0 = off) ---------------------
     * ln_in != 0: the launch computes epilogue( LayerNorm(A)[M, K] x W^T ) with LayerNorm over the K = c0 columns of every A row
     * (ln_gamma / ln_beta fp32 [K], ln_eps; two-pass fp32 statistics, the normalised rows rounded to bf16 exactly as a t2v_layernorm
     * launch would have written them): norm1 / norm2 / norm3 of BasicTransformerBlock (attention.py:300-311) where their output feeds
     * ONE Linear — no LayerNorm launch, no normalised tensor in memory.  ln_out must be NULL; not combined with a residual.
     * t2v_gemm and t2v_conv_halo refuse a descriptor with ln_in set. */
    int ln_in;
    /* ---- GroupNorm applied to the INPUT rows while the activation panel is filled (t2v_linear_pr only; NULL = off) -----------------------
     * gn_coef fp32 [units][2][K]: the per-channel affine t2v_gn_coef_cs made of the producers' column statistics (coef[u][0][c] = rstd
     * gamma[c], coef[u][1][c] = beta[c] - mean rstd gamma[c]); unit of row m = m / gn_rows_per_unit.  The launch computes
     * epilogue( (A * coef[u][0] + coef[u][1]) x W^T ), the normalised rows rounded to bf16 as t2v_group_norm_cs would have written them —
     * SpatialTransformer / TemporalTransformer: x = proj_in(norm(x)) (attention.py:373-389,471-513) without the normalised tensor in
     * memory.  gn_rows_per_unit a multiple of the panel height (160 rows at K = 320, 96 at K = 640: ask t2v_linear_pr_supported), no
     * SiLU, plain epilogue (bias only), not combined with ln_in.  t2v_gemm and t2v_conv_halo refuse a descriptor with gn_coef set. */
    const float* gn_coef;
    int gn_rows_per_unit;
} t2v_gemm_desc;

int t2v_gemm(const t2v_gemm_desc* d, void* stream);
/* 1 if t2v_gemm would honour the descriptor's rowstat_out / colstat_out / lnf_stats fields (it resolves the tile and the
 * split-K factor exactly as the launch does: fast kernel, one K split, a tile that carries the fused epilogue), 0 if the
 * caller has to use the standalone normalisation kernels, negative on an invalid descriptor.  Launches nothing. */
int t2v_gemm_fuse_supported(const t2v_gemm_desc* d);
/* The tile id and the K-split count t2v_gemm would use for the descriptor (resolved exactly as the launch does; launches nothing).
 * The gradient engine asks with the PLAIN descriptor before it attaches a LoRA epilogue (lora_*: one K split only): where the plain
 * launch would split K, the three-launch form (up-projection launches + split base leaf) is faster. */
int t2v_gemm_plan(const t2v_gemm_desc* d, int* tile_cfg, int* splits);
/* 3x3 convolution (stride 1, pad 1) with the activation tile's halo slab RESIDENT in LDS across all nine filter taps
 * (csrc/conv_halo.hip): same descriptor as t2v_gemm (mode T2V_GEMM_CONV3X3, virtual concat allowed; no batch / split-K /
 * dropout / GEGLU / fp32 output; epilogue: bias, rowvec, residual, SiLU, colstat_out), EXCEPT that `w` holds the SLAB-MAJOR pack:
 * K order (32-channel sub-slab, tap, channel) = [N][C/32][9][32] instead of t2v_gemm's tap-major [N][9][C] — one
 * (sub-slab, tap) pair is 64 contiguous bytes of a weight row, a stage of pairs one contiguous run — with every row zero-padded
 * to t2v_conv_halo_pack_cols(C) elements (ldw >= that).  Replaces the same reference call sites as t2v_gemm's 3x3 mode
 * (openaimodel3d.py:155-159,179-184).  tile_cfg 40 / 41 / 42 / 43 force the 320x160 / 320x80 / 160x80 (16-wide grid) / 160x80
 * (32-wide grid) workgroup tile (tokens x channels), anything else lets the library choose by grid fill.
 * t2v_conv_halo_supported: 0 = not taken (use t2v_gemm with the tap-major pack), 1 = taken; negative = invalid descriptor.
 * Launches nothing. */
int t2v_conv_halo(const t2v_gemm_desc* d, void* stream);
int t2v_conv_halo_supported(const t2v_gemm_desc* d);
int t2v_conv_halo_pack_cols(int channels);
int t2v_conv_halo_force_config(int cfg);
int t2v_conv_halo_debug(int bits);   /* ablation bits; honoured by -DT2V_HALO_ABLATE tool builds only */
/* Short-K linear layers (K = 320 / 640) with the activation panel RESIDENT in LDS and the weights streamed into registers in MFMA
 * fragment order (csrc/linear_pr.hip): same descriptor as t2v_gemm (mode T2V_GEMM_LINEAR, one source, no batch / split-K / rowvec /
 * dropout / fused statistics / fp32 output; epilogue: bias, then GEGLU, or an optional residual), EXCEPT that `w` holds the
 * FRAGMENT pack of the [N][K] matrix t2v_gemm takes (for GEGLU: of its 64-row [32 value | 32 gate] interleave), N % 64 == 0:
 *     pack[chunk q = n / 64][step s = k / 16][block b = (n % 64) / 32][lane l][e = 0..7]  (bf16, N * K elements)
 *         = W[64 q + 32 b + 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3)][16 s + 8 (l >> 5) + e],   i = l & 31
 * i.e. every (chunk, step, block) is the 1 KiB v_mfma_f32_32x32x16_bf16 A operand of 32 output rows x 16 K, lane-major, with the
 * rows of a block permuted so that a lane's 16 accumulator registers are 16 consecutive output channels.
 * Replaces the same reference call sites as t2v_gemm's LINEAR mode at these widths: the GEGLU projection
 * (lvdm/modules/attention.py:516-523), q | k | v (:71-76), to_out (:164), proj_in / proj_out (:373-389,471-513).
 * t2v_linear_pr_supported: 0 = not taken (use t2v_gemm with the plain [N][K] matrix), 1 = taken; negative = invalid descriptor.
 * Launches nothing. */
int t2v_linear_pr(const t2v_gemm_desc* d, void* stream);
int t2v_linear_pr_supported(const t2v_gemm_desc* d);
int t2v_linear_pr_debug(int bits);         /* ablation bits; honoured by -DT2V_LPR_ABLATE tool builds only */
int t2v_linear_pr_force_split(int ny);     /* tuning hook: column splits (workgroup rows) for every following call, 0 = library rule */
/* tuning/test hooks: override tile id / split-K factor for every following call (0 = off) */
int t2v_gemm_force_config(int cfg);
#ifdef T2V_EXPERIMENTAL /* measured negative results: compiled and exported by a T2V_EXPERIMENTAL=1 build (libt2v_hip_exp.so) only */
/* t2v_gemm's second kernel family (csrc/gemm2.hip: static-schedule main loop, 80x80 wave tiles; tile ids 50 = 320x160, 51 = 160x160)
 * for LINEAR / TCONV3 launches with a plain epilogue (bias, residual, SiLU, colstat_out).  Measured slower than the tuned first-family
 * tiles on the UNet's shapes (operand-delivery bound), so it is OFF by default: t2v_gemm2_enable(1) (or T2V_GEMM2=1) lets the library
 * route long-K launches to it by its own rule, a forced tile id 50 / 51 puts an eligible launch on it whatever its K. */
int t2v_gemm2_enable(int on);
#endif
int t2v_gemm_force_split(int splits);
int t2v_gemm_num_configs(void);

#ifdef T2V_EXPERIMENTAL
/* The GEGLU feed-forward of a BasicTransformerBlock, LayerNorm and residual included, in ONE launch (measured slower than the three
 * launches it replaces: profiles/r03_ffn_fused_pmc.csv):
 *     out = x + W2 . [value . gelu(gate)] + b2,   [value | gate] = W1 . LayerNorm(x) + b1
 * (lvdm/modules/attention.py:300-311 `x = self.ff(self.norm3(x)) + x`, :516-542 FeedForward / GEGLU).  The 4C-wide hidden
 * activation never reaches memory.  x, out: bf16 [M][ld] (not in place); C = 320 (and 64 for tests): t2v_ffn_fused_supported.
 * The weights come PRE-PACKED in MFMA fragment order, LayerNorm's affine folded into W1 / b1 (W1 diag(gamma), b1 + W1 beta):
 *   w1p bf16 [C/8 chunks][4 row tiles: value a | gate a | value b | gate b][C/32 k-steps][64 lanes][8]:
 *       lane l = W1'[row][32 s + 8 (l >> 4) .. +8], row = (gate ? 4C : 0) + 32 chunk + 16 (b ? 1 : 0) + (l & 15)
 *   b1p fp32 [C/8][4][16] in the same row order
 *   w2p bf16 [C/8 chunks][C/16 row tiles][64 lanes][8]: lane l = W2[16 t + (l & 15)][32 chunk + h], h = 4 (l >> 4) + e for
 *       e < 4 (pair a), 16 + 4 (l >> 4) + e - 4 for e >= 4 (pair b)
 *   b2 fp32 [C] */
int t2v_ffn_fused_supported(int C);
int t2v_ffn_fused(const void* x, int ldx, int M, int C, const void* w1p, const float* b1p, const void* w2p, const float* b2,
                  float ln_eps, void* out, int ldo, void* stream);
#endif

/* direct 3x3 s1 p1 conv for tiny Cin (the 4-channel latent): x bf16 [M][cin] (cin <= 8),
 * w fp32 [cout][9][cin], bias fp32 [cout], out bf16 [M][cout].
 * Replaces input_blocks.0 (openaimodel3d.py:435) and Decoder.conv_in (ae_modules.py:550-552). */
int t2v_conv3x3_small_cin(const void* x, int n_img, int h, int w, int cin, const float* wgt,
                          const float* bias, int cout, void* out, void* stream);

#ifdef T2V_EXPERIMENTAL
/* direct 3x3 s1 p1 conv for a tiny number of OUTPUT channels (1 <= cout <= 4; measured 4x slower than the MFMA tile): x bf16 [M][cin] (row stride ldx, cin % 8 == 0),
 * w fp32 [cout][9][cin] (tap-major), bias fp32 [cout] or NULL, out [M][cout] fp32 (out_f32 != 0) or bf16 at row stride ldo; the image
 * width must be a multiple of 4 (a thread owns four pixels of a row).  Replaces Decoder.conv_out (ae_modules.py:641: 128 -> 3 channels
 * at 320x512) — on the MFMA tiles that conv multiplies a 64-/128-wide tile of padding.  t2v_conv3x3_small_cout_supported: 1 / 0. */
int t2v_conv3x3_small_cout_supported(int w, int cin, int cout);
int t2v_conv3x3_small_cout(const void* x, int ldx, int n_img, int h, int w, int cin, const float* wgt, const float* bias, int cout,
                           void* out, int ldo, int out_f32, void* stream);
#endif

/* ---------------------------------------------------------------- normalisation
 * GroupNorm(32) in two phases over token-major data with optional virtual concat.
 * A "unit" is the statistics extent: one frame (ResBlock / SpatialTransformer / VAE, 4-D input)
 * or all frames of a clip (TemporalConvBlock / TemporalTransformer, 5-D input):
 * rows_per_unit consecutive rows.  Replaces GroupNormSpecific / nn.GroupNorm (+ nn.SiLU)
 * (lvdm/basics.py:78-89; openaimodel3d.py:155-157,179-181,275-292,666-668;
 * attention.py:340-342,422-424; ae_modules.py:11-19).
 * ws: fp32 workspace of t2v_gn_ws_floats(...) floats.  stats out: [n_units][groups][2] = (mean, rstd). */
long long t2v_gn_ws_floats(int n_units, int rows_per_unit, int groups);
int t2v_gn_stats(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                 int rows_per_unit, int groups, float eps, float* ws, float* stats, void* stream);
int t2v_gn_apply(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units,
                 int rows_per_unit, int groups, const float* stats, const float* gamma,
                 const float* beta, int silu, void* out, int ldo, void* stream);
/* GroupNorm(+SiLU) in one call (what the engines use): statistics + normalise — ONE launch (a workgroup per (group, unit), the group's
 * slice of the unit in registers, ws untouched) where a group has a multiple of 8 channels and rows_per_unit * channels-per-group / 8
 * <= 4096 and no prefetch hint is given; else 2 launches for tensors with few row slabs, 3 otherwise.
 * ws: t2v_group_norm_ws_floats(n_units, rows_per_unit, groups, c0 + c1) floats.
 * prefetch / prefetch_bytes (NULL / 0 = off): a buffer the NEXT launch will stream — the weights of the conv that follows every
 * GroupNorm of the UNet — touched with streaming loads by the normalise pass so that it sits in the 256 MB Infinity Cache when
 * that launch starts (a UNet step walks 2.8 GB of weights: nothing survives from the previous step).  Hint only. */
long long t2v_group_norm_ws_floats(int n_units, int rows_per_unit, int groups, int channels);
int t2v_group_norm(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units, int rows_per_unit,
                   int groups, float eps, const float* gamma, const float* beta, int silu, float* ws, void* out, int ldo,
                   const void* prefetch, long long prefetch_bytes, void* stream);

/* GroupNorm(+SiLU) whose statistics pass reads the column statistics the producing t2v_gemm launches wrote
 * (t2v_gemm_desc::colstat_out) instead of the tensor: cs0 [rows/32][c0][2] for x0, cs1 [rows/32][c1][2] for the second part of
 * a virtual concat (NULL without one); rows_per_unit % 32 == 0.  Two launches; ws: t2v_group_norm_cs_ws_floats(...) floats.
 * Same result as t2v_group_norm up to the summation order of the statistics (fixed: deterministic).  The statistics launch has two
 * forms, chosen by shape: one block per (group, unit) that leaves the per-channel affine (rstd gamma, beta - mean rstd gamma) in ws and
 * an apply pass that streams with it (even channels per group, c0 even, cs0 / cs1 16-byte aligned, 2 (c0 + c1) <= the ws floats per
 * unit, at most 32 slabs per thread), or per-slab-block partial sums that every block of the apply pass finishes for itself. */
long long t2v_group_norm_cs_ws_floats(int n_units, int rows_per_unit, int groups);
int t2v_group_norm_cs(const float* cs0, const float* cs1, const void* x0, int c0, int ld0, const void* x1, int c1, int ld1,
                      int n_units, int rows_per_unit, int groups, float eps, const float* gamma, const float* beta, int silu,
                      float* ws, void* out, int ldo, const void* prefetch, long long prefetch_bytes, void* stream);
/* (mean, rstd) per (unit, group) — the `stats` of t2v_gn_stats — from the column statistics alone: the training engine keeps them
 * for the GroupNorm backward and normalises with t2v_gn_apply (openaimodel3d.py:223-254 under autograd).  cs1 / c1: second part of
 * a virtual concat (NULL / 0: none).  ws: t2v_group_norm_cs_ws_floats floats. */
int t2v_gn_stats_cs(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups, float eps,
                    float* ws, float* stats, void* stream);

/* LayerNorm over the channel dim, eps, affine; bf16 in/out (attention.py:279-281). */
int t2v_layernorm(const void* x, int ldx, int M, int C, const float* gamma, const float* beta,
                  float eps, void* out, int ldo, void* stream);

/* row softmax in place on bf16 [rows][ld]: cols [0,n) normalised, cols [n, n_pad) set to 0
 * (score matrix of the GEMM-formulated attention: attention.py:143, ae_modules.py:61). */
int t2v_softmax_rows(void* s, long long rows, int n, int n_pad, int ld, void* stream);

/* ---------------------------------------------------------------- attention
 * Fused spatial attention (flash style), head dim 64: softmax(Q K^T * scale) V per (image, head).
 * q: bf16 rows (img*seq_q + i), head h at columns [h*64, h*64+64); k likewise over seq_kv rows;
 * vt: V transposed per image: bf16 [img_kv][heads*64][ld_vt] (keys contiguous); vt_img_stride = elements
 * between consecutive kv images (0 = heads*64*ld_vt; larger when several layers' V^T share one buffer).
 * ld_vt >= seq_kv rounded up to 64; the padding columns are read (their probability is exactly 0) and
 * must hold finite values.
 * kv image of q image b is b / kv_div (text cross-attention shares K/V across the frames of a clip).
 * Replaces CrossAttention.forward / efficient_forward (attention.py:102-164,166-240 = xformers
 * memory_efficient_attention). */
int t2v_attn_spatial(const void* q, int ldq, const void* k, int ldk, const void* vt, int ld_vt,
                     long long vt_img_stride, void* out, int ldo, int n_img, int seq_q, int seq_kv,
                     int heads, int kv_div, float scale, void* stream);

/* Temporal self-attention, head dim 64, sequence = frames: for every (clip b, pixel p, head h)
 * softmax(Q K^T * scale) V over the F frames, reading rows ((b*F+f)*HW + p) directly from the
 * token-major q/k/v (no '(b h w) t c' rearrange copies, attention.py:475-511,102-164).
 * probs (optional, fp32 [(b*HW+p)*heads+h][F][F]) = CrossAttention.attention_probs
 * (attention.py:124-126). */
int t2v_attn_temporal(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                      void* out, int ldo, int n_clips, int frames, int hw, int heads, float scale,
                      float* probs, void* stream);

/* ---------------------------------------------------------------- layout / elementwise
 * (b,c,f,h,w) tensor of dtype `dt` <-> token-major bf16/fp32 rows (openaimodel3d.py:714,739). */
int t2v_ncfhw_to_tokens(const void* x, int dt, int b, int c, int f, int hw, void* out, int ldo,
                        void* stream);
int t2v_tokens_to_ncfhw(const void* tok, int tok_f32, int ld, int b, int c, int f, int hw,
                        void* out, int dt, void* stream);
/* sinusoidal embedding cos||sin of t (int64) -> bf16 [n][dim] (utils_diffusion.py:8-32);
 * flip_sin_cos=1, scale: sin||cos of t*scale with the (half-1) divisor = guidance embedding
 * (pipeline/t2v_turbo_vc2_pipeline.py:99-120) taking fp32 t. */
int t2v_timestep_embedding(const void* t, int t_is_f32, int n, int dim, int guidance_style,
                           void* out_bf16, void* stream);
int t2v_silu(const void* x, void* out, long long n, void* stream); /* bf16 -> bf16 */
/* stream-ordered memset(0) of a caller buffer (padding columns of the GEMM-formulated attention) */
int t2v_fill_zero(void* p, long long nbytes, void* stream);
int t2v_cast(const void* x, int dt_in, void* out, int dt_out, long long n, void* stream);
/* out = ca[b]*x + cb[b]*y + cc[b]*z over (b, inner) fp32 tensors; y/z may be NULL. Host coefficient
 * arrays of length nb (<= 64).  The scheduler / consistency-distillation elementwise family
 * (t2v_turbo_scheduler.py:437-462,470-495; utils/common_utils.py:87-133; ode_solver/ddim_solver.py:67-97). */
int t2v_lincomb3(const float* x, const float* y, const float* z, const float* ca, const float* cb,
                 const float* cc, int nb, long long inner, float* out, void* stream);
/* fused LCM step: x0=(x-sb_t*eps)/sa_t; den=c_out*x0+c_skip*x; prev=sa_p*den+sb_p*noise
 * (t2v_turbo_scheduler.py:437-462). eps may be bf16 or fp32; x/noise/prev/den fp32. */
int t2v_lcm_step(const float* x, const void* eps, int eps_dt, const float* noise, float sa_t,
                 float sb_t, float c_skip, float c_out, float sa_p, float sb_p, long long n,
                 float* prev, float* denoised, void* stream);

/* ---------------------------------------------------------------- diagnostics (tools/ only)
 * Ablation bits for tools/gemm_one.py and tools/attn_one.py.  They only act in libraries built with
 * -DT2V_GEMM_ABLATE / -DT2V_ATTN_ABLATE (runtime branches inside the K / KV loops cost ~20 %, so the product build
 * compiles them out and these calls just store the value). */
int t2v_gemm_debug(int bits);
int t2v_attn_debug(int bits);
#ifdef T2V_EXPERIMENTAL
/* which form of the spatial forward t2v_attn_spatial launches (tools / tests; the product default is 0 and the others are measured no
 * faster): 0 = 4 waves x 32 queries per workgroup, 8 = 8 waves x 32, 64 = 4 waves x 64 queries (two query sets per wave, phases offset)
 * on launches with >= 512 queries and keys, 65 = that form always.  Same arithmetic per query in every form: bit-identical outputs. */
int t2v_attn_spatial_form(int form);
#endif
/* t2v_group_norm has a ONE-launch form (registers hold the tensor, per-unit inter-workgroup barrier) for tensors that fit:
 * t2v_gn_coop_enable(1) / environment T2V_GN_COOP=1 selects it (default off: measured slower than the three-launch form on
 * MI355X for cache-resident tensors); t2v_gn_coop_error() returns 1 if a
 * barrier of the one-launch form ever timed out on the current device (workgroups not co-resident), -1 on a HIP error. */
int t2v_gn_coop_enable(int on);
int t2v_gn_coop_error(void);

/* ---------------------------------------------------------------- backward (dX) pieces of the VAE decoder
 * Reward-gradient branch (train_t2v_turbo_v1_lora.py:1047-1098: autograd through vae.decode, ae_modules.py:602-641).
 * Conv / linear data gradients are t2v_gemm launches on re-packed weights; these are the non-GEMM parts.
 * t2v_gn_bwd: dx = d/dx [act(GroupNorm(x))] . dy (+ resid), act = SiLU or identity; stats = (mean, rstd) per
 *   (unit, group) from t2v_gn_stats of the forward; ws: t2v_gn_bwd_ws_floats() floats.  x, dy, resid, dx bf16.
 * t2v_softmax_bwd_rows: dp <- p * (dp - sum_j p_j dp_j) per row (columns >= n set to 0).
 * t2v_transpose_bf16: out[b][c][r] = in[b][r][c].   t2v_sumpool2x2: adjoint of nearest-x2 upsampling (token-major). */
long long t2v_gn_bwd_ws_floats(int n_units, int rows_per_unit, int groups);
int t2v_gn_bwd(const void* x, int ldx, int C, int n_units, int rows_per_unit, int groups, const float* stats,
               const float* gamma, const float* beta, int silu, const void* dy, int ldy, const void* resid, int ldr,
               float* ws, void* dx, int ldo, void* stream);
int t2v_softmax_bwd_rows(const void* p, void* dp, long long rows, int n, int n_pad, int ld, void* stream);
int t2v_transpose_bf16(const void* in, int ld_in, int rows, int cols, void* out, int ld_out, int batch,
                       long long in_stride, long long out_stride, void* stream);
int t2v_sumpool2x2(const void* in, int n_img, int h, int w, int C, void* out, void* stream);

/* ---------------------------------------------------------------- backward (dX) pieces of the UNet
 * d(loss)/d(latents) through UNetModel.forward (openaimodel3d.py:672-740) for losses on the output and on the recorded
 * temporal attention probabilities: motion_prior_sample.py:59-84 (autograd.grad(loss, latents)) and the data half of the
 * student's backward (train_t2v_turbo_v1_lora.py:1190).  Convolution / linear / spatial-attention gradients are t2v_gemm
 * launches; these are the rest.  STATUS: written after the round's GPU budget was spent - the engine's dataflow is verified on
 * CPU against autograd with an emulation of exactly these semantics, and on MI355X against the emulation and autograd (tests/test_gpu_unet_grad.py).
 * t2v_gn_bwd2: t2v_gn_bwd over a virtual channel concat [x0 | x1] (the skip connections), up to 4096 channels.
 * t2v_layernorm_bwd: dx = d/dx LayerNorm(x) . dy (+ resid); statistics recomputed per row (attention.py:279-281).
 * t2v_geglu_fwd / _bwd: h = packed pre-activation [M][2*inner] in 64-column groups [32 value | 32 gate] (the layout
 *   T2V_ACT_GEGLU consumes); out = value * gelu(gate); dh in the same packed layout (attention.py:516-523).
 * t2v_scatter2x: out[n][2y][2x] = src[n][y][x], zero elsewhere, H in {2h-1, 2h}: the gradient of a 3x3 s2 p1 conv is the
 *   flipped 3x3 s1 p1 conv over this (openaimodel3d.py:63-72).
 * t2v_add_bf16: out = a + b over [M][C] with row strides (gradient fan-in at the skip connections).
 * t2v_attn_temporal_bwd: (dq, dk, dv) of t2v_attn_temporal from d(out) and, optionally, d(probs)
 *   (fp32 [(b*HW+p)*heads+h][F][F], the layout of the forward's probs output); frames <= 16. */
int t2v_gn_bwd2(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int n_units, int rows_per_unit, int groups,
                const float* stats, const float* gamma, const float* beta, int silu, const void* dy, int ldy, const void* resid,
                int ldr, float* ws, void* dx, int ldo, void* stream);
int t2v_layernorm_bwd(const void* x, int ldx, int M, int C, const float* gamma, float eps, const void* dy, int ldy,
                      const void* resid, int ldr, void* dx, int ldo, void* stream);
int t2v_geglu_fwd(const void* h, int ldh, long long M, int inner, void* out, int ldo, void* stream);
int t2v_geglu_bwd(const void* h, int ldh, const void* dy, int ldy, long long M, int inner, void* dh, int ldd, void* stream);
int t2v_scatter2x(const void* src, int n_img, int h, int w, int C, int H, int W, void* out, void* stream);
int t2v_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, long long M, int C, void* stream);
int t2v_attn_temporal_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                          const float* dprobs, void* dq, int ldq2, void* dk, int ldk2, void* dv, int ldv2, int n_clips, int frames,
                          int hw, int heads, float scale, void* stream);

/* ---------------------------------------------------------------- optimizer / EMA over flat fp32 buffers
 * Replace bitsandbytes AdamW8bit / torch AdamW on the LoRA tensors (train_t2v_turbo_v1_lora.py:765-803),
 * accelerator.clip_grad_norm_ (:1193) and update_ema (utils/common_utils.py:307-319).
 * t2v_adamw_step: decoupled weight decay, bias-corrected (torch.optim.AdamW semantics); grad_scale multiplies
 * the gradient first (fold the clip coefficient in).  t2v_sumsq: out[0] = sum(x^2), deterministic two-pass;
 * ws = workspace of >= 1024 floats. */
int t2v_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                   void* stream);
int t2v_ema_update(float* target, const float* src, float rate, long long n, void* stream);
int t2v_sumsq(const float* x, long long n, float* ws, float* out, void* stream);

/* ---------------------------------------------------------------- LoRA training path (utils/lora.py:45-50,124-129,204-209)
 * t2v_gather_f32: indexed re-layout of fp32 data.  accumulate = 0: out[i] = idx[i] >= 0 ? alpha*src[idx[i]] : 0;
 * accumulate = 1: out[i] += alpha*src[idx[i]] where idx[i] >= 0.  out is fp32 (T2V_F32) or bf16 (T2V_BF16).
 * One launch packs every lora_down / lora_up tensor (flat fp32 parameters) into the bf16 GEMM operand layouts, and one
 * launch scatters every LoRA weight gradient from the GEMM output layout into the flat gradient buffer the
 * all-reduce and the optimizer work on (what autograd's AccumulateGrad does per tensor in the reference,
 * train_t2v_turbo_v1_lora.py:1190). */
int t2v_gather_f32(const float* src, const int* idx, float alpha, void* out, int dt_out, int accumulate, long long n,
                   void* stream);
/* t2v_attn_spatial_bwd: flash-style backward of the spatial SELF-attention (head dim 64): dQ, dK, dV from token-major Q / K / V /
 * dO and the forward's O, the [queries x keys] probabilities recomputed tile by tile (never written to memory).  Two launches:
 * per query block (statistics L = log2-sum-exp and D = sum_c dO O into l2 / dsum, then dQ), per key block (dK, dV).
 * q, k, dout, o, dq, dk, dv: bf16 [n_img*seq][ld], head h at column 64 h.  v: bf16, row r of (img, head) at
 * v + img*v_img_stride + head*v_head_stride + r*ldv (token-major buffer or the per-head [keys][64] transpose of V^T).
 * kt, qt, dot: K^T, Q^T, dO^T per image, [n_img][heads*64][ld] with the sequence zero-padded to a multiple of 64
 * (t2v_transpose_pad_bf16).  l2, dsum: fp32 workspaces [n_img*heads][ld_stat], ld_stat >= seq_q.
 * Replaces the autograd backward of CrossAttention.forward for attn1 of the spatial transformers (attention.py:102-164). */
int t2v_attn_spatial_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, long long v_img_stride,
                         long long v_head_stride, const void* kt, int ld_kt, const void* qt, const void* dot, int ld_qt,
                         const void* dout, int ldo, const void* o, int ldoo, float* l2, float* dsum, int ld_stat, void* dq, int lddq,
                         void* dk, int lddk, void* dv, int lddv, int n_img, int seq_q, int seq_kv, int heads, float scale,
                         void* stream);
/* t2v_wgrad_tn: out[R][C] (fp32, row stride ldo) = alpha * a[:, :R]^T b[:, :C] for two TOKEN-MAJOR bf16 operands a [M][lda],
 * b [M][ldb] — the token-contracted LoRA weight gradients (dU = s dy^T t, dD = G^T x) and per-clip column sums without transposed
 * operand copies, and the base-weight gradients of full fine-tuning (dW = dy^T x, dy^T xcol).  The token range is split over
 * workgroups (splits = 0: library choice); fp32 partial tiles go through the caller's workspace ws (>= splits*R*C*4 bytes, the split
 * count shrinks to fit) and are added in a fixed order.  With ONE split — a product with enough output tiles to fill the chip, or an
 * output larger than the workspace — the tiles are written to `out` directly and ws is not touched.  Output tile 64 x 64, or
 * 128 x 128 where both extents reach 128 and pad to multiples of 128 within 10 % (T2V_WGRAD_TILE128=0: always 64 x 64). */
int t2v_wgrad_tn(const void* a, int lda, const void* b, int ldb, long long M, int R, int C, float alpha, float* out, int ldo, float* ws,
                 long long ws_bytes, int splits, void* stream);
/* t2v_wgrad_tn_group: up to T2V_WGRAD_GROUP_MAX such products in one launch pair (main + fixed-order reduce): the weight gradients
 * of ONE LoRA group — dU of each of its leaves and dD (utils/lora.py:45-50 under autograd) — whose operands all exist once the
 * rank-r gradient is there.  Same arithmetic and summation structure per product as t2v_wgrad_tn; the token splits are chosen for
 * the group as a whole.  ws: fp32 workspace for the partial slabs of all products. */
#define T2V_WGRAD_GROUP_MAX 8
typedef struct t2v_wgrad_problem {
    const void* a;      /* bf16 [M][lda], columns [0, R) */
    const void* b;      /* bf16 [M][ldb], columns [0, C) */
    float* out;         /* fp32 [R][ldo] = alpha * a^T b */
    long long M;
    int lda, ldb, ldo, R, C;
    float alpha;
} t2v_wgrad_problem;
int t2v_wgrad_tn_group(const t2v_wgrad_problem* problems, int n, float* ws, long long ws_bytes, void* stream);
/* t2v_gn_coef_cs: the per-channel affine of a GroupNorm alone, from the producers' column statistics (t2v_gemm_desc::colstat_out layout,
 * one array per part of a virtual concat): coef fp32 [n_units][2][c0 + c1] — for consumers that normalise in their own load phase
 * (t2v_gemm_desc::gn_coef).  t2v_gn_coef_cs_supported: 1 / 0 (else t2v_group_norm_cs, which writes the normalised tensor). */
int t2v_gn_coef_cs_supported(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups);
int t2v_gn_coef_cs(const float* cs0, int c0, const float* cs1, int c1, int n_units, int rows_per_unit, int groups, float eps,
                   const float* gamma, const float* beta, float* coef, void* stream);
/* ---- base-weight gradients for FULL fine-tuning (csrc/full_grad.hip; train_latent_t2v_turbo_v2.py:798-816,1262) ----------------------
 * t2v_im2col_bf16: the shifted-row matrix of a conv leaf in the K order of the forward's tap-major pack,
 *     out[m][tap * C + c] = x[src(m, tap)][c]   (0 outside the grid; x = virtual concat [x0 | x1], C = c0 + c1, channels % 8 == 0)
 * for mode = T2V_GEMM_CONV3X3 / _S2 / _S2_PAD01 / _UP2 (9 taps, (ky, kx) row-major) or T2V_GEMM_TCONV3 (3 taps = frame offsets; n_img =
 * clips x frames) on the (n_img, h, w) INPUT grid; out: bf16 [t2v_im2col_rows(mode, n_img, h, w)][ldo], ldo >= taps * C.  With it the
 * weight gradient of a conv leaf, dW[n][tap][c] = sum_m dy[m][n] out[m][tap * C + c], is one t2v_wgrad_tn(dy, out) product. */
long long t2v_im2col_rows(int mode, int n_img, int h, int w);
int t2v_im2col_bf16(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, int mode, int n_img, int h, int w, int frames, void* out,
                    int ldo, void* stream);
/* t2v_norm_affine_grad: per-channel sums over token rows, `sum_rows` consecutive rows per output row u (rows % sum_rows == 0):
 *     dgamma[u][c] = sum dz[m][c] xhat[m][c],   dbeta[u][c] = sum dz[m][c]
 * kind 0: xhat from GroupNorm statistics stats[row / rows_per_unit][group][2] = (mean, rstd) of the forward (lvdm/basics.py:78-89);
 * kind 1: xhat = LayerNorm's (x - mean_row) * rstd_row, recomputed per row with ln_eps (attention.py:300-311);
 * kind 2: no norm — dbeta = plain column sums of dy (bias gradients; per-clip sums = d(loss)/d(time embedding row); c0 = columns of dy).
 * silu != 0 (kinds 0, 1): the forward applied SiLU behind the norm (openaimodel3d.py:223-254), dz = dy * silu'(xhat gamma + beta); else
 * dz = dy.  x = [x0 | x1] bf16 (the norm's INPUT), dy bf16 [rows][ldy] over the C = c0 + c1 channels (C % 8 == 0; C <= 2560 for kinds 0 / 1); dgamma /
 * dbeta fp32 with row strides ld_dgamma / ld_dbeta (either may be NULL for kind 2 / when not wanted).  ws: fp32 workspace of
 * t2v_norm_affine_grad_ws_floats(rows, sum_rows, C) floats.  Two launches, fixed summation order (no float atomics). */
long long t2v_norm_affine_grad_ws_floats(long long rows, long long sum_rows, int channels);
int t2v_norm_affine_grad(const void* x0, int c0, int ld0, const void* x1, int c1, int ld1, long long rows, long long sum_rows, int kind,
                         int rows_per_unit, int groups, const float* stats, float ln_eps, const float* gamma, const float* beta, int silu,
                         const void* dy, int ldy, float* dgamma, int ld_dgamma, float* dbeta, int ld_dbeta, float* ws, void* stream);
/* t2v_repack_conv_f32: a conv leaf's fp32 parameter w [N][C][taps] (taps = 9: (ky, kx) row-major, or 3: frame offsets; <= 9) into one of
 * the bf16 packs the launch lists read, cast (round to nearest even) included:
 *     kind 0   out[n][t * C + c] = w[n][c][t]                   the tap-major forward pack (t2v_gemm's K order), ldo >= taps * C
 *     kind 1   out[c][(taps - 1 - t) * N + n] = w[n][c][t]      the data-gradient pack: the same conv over dy with channels and filters
 *                                                               swapped and the taps mirrored, ldo >= taps * N
 * Full fine-tuning re-makes every pack per optimizer step IN PLACE (the recorded launch lists keep their pointers); this replaces
 * torch's permute / flip / cast chain for the conv leaves (70 % of the UNet's parameters). */
int t2v_repack_conv_f32(const float* w, int N, int C, int taps, int kind, void* out, int ldo, void* stream);
/* t2v_transpose_pad_bf16: out[b][c][r] = in[b][r][c] for r < rows and 0 for rows <= r < roundup(rows, 64) — the K-contiguous,
 * K-padded operand of the token-contracted weight-gradient GEMMs (dU = dy^T t, dD = G^T x) in one pass; 16-byte accesses on both
 * sides: cols % 8 == 0, ld_in % 8 == 0, ld_out % 8 == 0 and >= roundup(rows, 64), 16-byte aligned bases, batch strides % 8. */
int t2v_transpose_pad_bf16(const void* in, int ld_in, int rows, int cols, void* out, int ld_out, int batch, long long in_stride,
                           long long out_stride, void* stream);
/* t2v_dropout_bf16: out[r][c] = keep(r, c) ? x[r][c] / (1 - p') : 0  (+ resid[r][c]) over rows x ncols bf16 (ncols even), p' = ((p * 2^32) >> 16) / 65536
 * (the probability the 16-bit mask really drops with: the scale keeps E[out] = x exactly), where
 * keep is a pure function of (*seed, site, i = r * ncols + c): word = splitmix64(seed + site * 0x9E3779B97F4A7C15 + (i >> 2) *
 * 0xD1B54A32D192ED03), element i keeps iff bits [16 (i & 3), +16) of the word >= (p * 2^32) >> 16 (p resolved to 2^-16).  The backward calls it again with the same (seed,
 * site) on the gradient.  seed: device pointer to one uint64 (a replayed launch list follows the step's seed).  In-place is
 * allowed (out == x).  Replaces nn.Dropout in LoraInjected*.forward (utils/lora.py:45-50,124-129) and TemporalConvBlock
 * (openaimodel3d.py:280-297); the random stream is not torch's (train-mode parity is statistical). */
int t2v_dropout_bf16(const void* x, int ldx, const void* resid, int ldr, void* out, int ldo, long long rows, int ncols, float p,
                     const void* seed, unsigned site, void* stream);

/* ---------------------------------------------------------------- recorded launch lists
 * The reference has no counterpart: its hot path is issued op by op from Python (and that is what bounds a small-batch step on
 * the host).  The engines of this library record a forward / backward once and replay it; t2v_replay walks such a recording —
 * a flat array of 64-bit words [function id, n, n argument slots] ... — inside the library, one host call per LIST instead of
 * one per launch.  Function ids come from t2v_replay_lookup(name) (-1: not a replayable entry point).  A slot holds an integer
 * or pointer value as is, a float as its 32 bits (low half), a double as its 64 bits; the stream parameter (always last in the
 * entry point's signature) is not stored: every launch goes to the stream given here.  Returns the first non-zero return code
 * and its launch index in *failed_index (may be NULL). */
int t2v_replay_lookup(const char* name);
int t2v_replay(const unsigned long long* prog, long long nwords, void* stream, int* failed_index);

#ifdef __cplusplus
}
#endif
#endif /* T2V_HIP_H */
