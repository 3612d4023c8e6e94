"""ctypes binding of libt2v_hip.so (include/t2v_hip.h) and the tensor-level HIP op backend.

There is NO fallback here: if the shared library is missing or a kernel rejects a shape, the
caller gets an exception.  torch is used only for device memory (tensors own the buffers) and for
the current stream.
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libt2v_hip.so")

# include/t2v_hip.h constants
F32, BF16, F16 = 0, 1, 2
GEMM_LINEAR, GEMM_CONV3X3, GEMM_CONV3X3_S2, GEMM_CONV3X3_UP2, GEMM_TCONV3, GEMM_CONV3X3_S2_PAD01 = range(6)
ACT_NONE, ACT_GEGLU, ACT_SILU = 0, 1, 2

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


WGRAD_GROUP_MAX = 8


class WgradProblem(C.Structure):
    """struct t2v_wgrad_problem, field for field."""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("M", C.c_longlong), ("lda", C.c_int), ("ldb", C.c_int),
                ("ldo", C.c_int), ("R", C.c_int), ("C", C.c_int), ("alpha", C.c_float)]


class GemmDesc(C.Structure):
    """struct t2v_gemm_desc"""
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("c0", C.c_int), ("c1", C.c_int),
        ("lda0", C.c_int), ("lda1", C.c_int), ("mode", C.c_int),
        ("n_img", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int), ("frames", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("w", C.c_void_p), ("ldw", C.c_int),
        ("batch", C.c_int), ("batch_inner", C.c_int),
        ("a_stride0", C.c_longlong), ("a_stride1", C.c_longlong),
        ("w_stride0", C.c_longlong), ("w_stride1", C.c_longlong),
        ("o_stride0", C.c_longlong), ("o_stride1", C.c_longlong),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("rowvec_div", C.c_int), ("ld_rowvec", C.c_int), ("residual", C.c_void_p), ("ldr", C.c_int),
        ("act", C.c_int), ("out", C.c_void_p), ("ldo", C.c_int), ("out_f32", C.c_int),
        ("tile_cfg", C.c_int), ("split_k", C.c_int), ("ws", C.c_void_p), ("ws_bytes", C.c_longlong),
        ("drop_seed", C.c_void_p), ("drop_thr", C.c_uint), ("drop_site", C.c_uint), ("drop_inv_keep", C.c_float),
        ("drop_ncols", C.c_int), ("drop_col0", C.c_int),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("ln_out", C.c_void_p), ("ld_ln_out", C.c_int),
        ("rowstat_out", C.c_void_p), ("ld_rowstat", C.c_int), ("colstat_out", C.c_void_p),
        ("lnf_stats", C.c_void_p), ("lnf_ld", C.c_int), ("lnf_nblk", C.c_int), ("lnf_eps", C.c_float), ("lnf_s", C.c_void_p),
        ("lora_t", C.c_void_p), ("ld_lora_t", C.c_int), ("lora_u", C.c_void_p), ("ld_lora_u", C.c_int), ("lora_n_leaf", C.c_int),
        ("lora_scale", C.c_float), ("ln_in", C.c_int), ("gn_coef", C.c_void_p), ("gn_rows_per_unit", C.c_int),
    ]


class NativeError(RuntimeError):
    pass


_lib = None


def conv_halo_pack_cols(channels):
    """Columns of a slab-major pack row for ``channels`` input channels: an even number of 4-pair weight stages (t2v_conv_halo_pack_cols)."""
    return ((channels // 32 * 9 + 7) // 8) * 8 * 32


def dropout_thr16(p):
    """The 16-bit keep threshold the device compares an element's mask bits with: (p * 2^32) >> 16 (csrc/common.h)."""
    t = float(p) * 4294967296.0
    return (0xFFFFFFFF if t >= 4294967295.0 else int(t)) >> 16


def dropout_inv_keep(p):
    """1 / (1 - p') with p' = thr16 / 65536, the drop probability the 16-bit mask really has (p = 0.1 -> 6553 / 65536): the scale
    that keeps E[dropout(x)] = x exactly.  p < 2^-16 quantises to 'keep everything' at scale 1; p must be < 1 - 2^-16 (the library
    refuses more: one element in 65536 would survive at scale 65536 where torch gives all zeros)."""
    if dropout_thr16(p) >= 0xFFFF:
        raise NativeError(f"dropout p = {p}: the 16-bit mask cannot represent p >= 1 - 2^-16")
    return 65536.0 / (65536.0 - dropout_thr16(p))


def pack_conv_slab(w_tap_major, taps=9):
    """[N, 9 * C] tap-major 3x3 conv weights (``Packer.conv``) -> the K order of t2v_conv_halo: (32-channel sub-slab, tap, channel),
    i.e. [N][C / 32][9][32] flattened — one (sub-slab, tap) pair is 64 contiguous bytes of every row — zero-padded to
    ``conv_halo_pack_cols`` columns (whole weight stages: the kernel multiplies the padding pairs by finite rows)."""
    n, k = w_tap_major.shape
    c = k // taps
    assert taps == 9 and c * taps == k and c % 32 == 0
    w = w_tap_major.reshape(n, taps, c // 32, 32).permute(0, 2, 1, 3).reshape(n, k)
    cols = conv_halo_pack_cols(c)
    out = w.new_zeros(n, cols)
    out[:, :k] = w
    return out


def unpack_conv_slab(w_slab_major, channels, taps=9):
    """Inverse of ``pack_conv_slab`` (drops the padding)."""
    n = w_slab_major.shape[0]
    k = taps * channels
    return w_slab_major[:, :k].reshape(n, channels // 32, taps, 32).permute(0, 2, 1, 3).reshape(n, k).contiguous()


def _lpr_row_perm(device=None):
    """MFMA D-row i of a 32-row block holds output channel 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3): a lane's 16 accumulator
    registers (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) are then the 16 consecutive channels 16 (lane >> 5) + r."""
    i = torch.arange(32, device=device)
    return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)


def pack_linear_pr(w):
    """[N, K] bf16 Linear weights (for GEGLU: the 64-row [32 value | 32 gate] interleave ``Packer.geglu`` makes) -> the FRAGMENT
    pack of t2v_linear_pr (include/t2v_hip.h): [chunk = n / 64][step = k / 16][block = (n % 64) / 32][lane][8], every (chunk, step,
    block) the 1 KiB A operand of v_mfma_f32_32x32x16_bf16, lane l = row (l & 31) of the permuted block, K 16 s + 8 (l >> 5) .. + 8.
    Returned as an [N, K] tensor (same bytes, other order) so that the descriptor plumbing is t2v_gemm's."""
    n, k = w.shape
    assert n % 64 == 0 and k % 16 == 0, "t2v_linear_pr: N % 64 == 0, K % 16 == 0"
    wb = w.reshape(n // 32, 32, k)[:, _lpr_row_perm(w.device), :]          # [block, i, K] with permuted rows
    wb = wb.reshape(n // 64, 2, 32, k // 16, 2, 8)                          # [q, b, i, s, hk, e]
    return wb.permute(0, 3, 1, 4, 2, 5).reshape(n, k).contiguous()         # [q, s, b, hk, i, e]: lane = 32 hk + i


def unpack_linear_pr(wp):
    """Inverse of ``pack_linear_pr``."""
    n, k = wp.shape
    wb = wp.reshape(n // 64, k // 16, 2, 2, 32, 8).permute(0, 2, 4, 1, 3, 5).reshape(n // 32, 32, k)   # [block, i, K]
    out = torch.empty_like(wb)
    out[:, _lpr_row_perm(wp.device), :] = wb
    return out.reshape(n, k).contiguous()


_SIGS = {
    "t2v_version": (C.c_int, []),
    "t2v_init": (C.c_int, []),
    "t2v_last_error": (C.c_char_p, []),
    "t2v_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "t2v_gemm_fuse_supported": (C.c_int, [C.POINTER(GemmDesc)]),
    "t2v_gemm_plan": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "t2v_conv_halo": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "t2v_conv_halo_supported": (C.c_int, [C.POINTER(GemmDesc)]),
    "t2v_conv_halo_force_config": (C.c_int, [C.c_int]),
    "t2v_conv_halo_debug": (C.c_int, [C.c_int]),
    "t2v_conv_halo_pack_cols": (C.c_int, [C.c_int]),
    "t2v_linear_pr": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "t2v_linear_pr_supported": (C.c_int, [C.POINTER(GemmDesc)]),
    "t2v_linear_pr_debug": (C.c_int, [C.c_int]),
    "t2v_linear_pr_force_split": (C.c_int, [C.c_int]),
    "t2v_replay_lookup": (C.c_int, [C.c_char_p]),
    "t2v_replay": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.POINTER(C.c_int)]),
    "t2v_gemm2_enable": (C.c_int, [C.c_int]),
    "t2v_gemm_force_config": (C.c_int, [C.c_int]),
    "t2v_gemm_force_split": (C.c_int, [C.c_int]),
    "t2v_gemm_num_configs": (C.c_int, []),
    "t2v_ffn_fused_supported": (C.c_int, [C.c_int]),
    "t2v_ffn_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_conv3x3_small_cin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p]),
    "t2v_conv3x3_small_cout_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "t2v_conv3x3_small_cout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "t2v_gn_ws_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "t2v_gn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2v_gn_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_void_p]),
    "t2v_group_norm_ws_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "t2v_group_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong,
                                 C.c_void_p]),
    "t2v_group_norm_cs_ws_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "t2v_gn_stats_cs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "t2v_group_norm_cs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_longlong, C.c_void_p]),
    "t2v_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_softmax_rows": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "t2v_attn_spatial": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_void_p]),
    "t2v_attn_temporal": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p]),
    "t2v_ncfhw_to_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    "t2v_tokens_to_ncfhw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.c_void_p]),
    "t2v_timestep_embedding": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "t2v_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "t2v_fill_zero": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "t2v_cast": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "t2v_lincomb3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_longlong, C.c_void_p, C.c_void_p]),
    "t2v_gemm_debug": (C.c_int, [C.c_int]),
    "t2v_attn_debug": (C.c_int, [C.c_int]),
    "t2v_attn_spatial_form": (C.c_int, [C.c_int]),
    "t2v_gn_coop_enable": (C.c_int, [C.c_int]),
    "t2v_gn_coop_error": (C.c_int, []),
    "t2v_gn_bwd_ws_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "t2v_gn_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_gn_bwd2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_void_p]),
    "t2v_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_geglu_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_geglu_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_scatter2x": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "t2v_add_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p]),
    "t2v_attn_temporal_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_void_p]),
    "t2v_softmax_bwd_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "t2v_transpose_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                     C.c_longlong, C.c_void_p]),
    "t2v_sumpool2x2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "t2v_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "t2v_ema_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_longlong, C.c_void_p]),
    "t2v_sumsq": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2v_gather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p]),
    "t2v_attn_spatial_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "t2v_wgrad_tn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "t2v_wgrad_tn_group": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]),
    "t2v_gn_coef_cs_supported": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "t2v_gn_coef_cs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "t2v_im2col_rows": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "t2v_im2col_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_repack_conv_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "t2v_norm_affine_grad_ws_floats": (C.c_longlong, [C.c_longlong, C.c_longlong, C.c_int]),
    "t2v_norm_affine_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "t2v_transpose_pad_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                         C.c_longlong, C.c_void_p]),
    "t2v_dropout_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_float,
                                   C.c_void_p, C.c_uint, C.c_void_p]),
    "t2v_lcm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
}
# Entry points of a T2V_EXPERIMENTAL=1 build only (libt2v_hip_exp.so: the measured negative results — the second t2v_gemm kernel family, the
# one-launch feed-forward, the alternative flash-attention forms, the direct small-Cout conv): bound when the loaded library has them.
EXPERIMENTAL = ("t2v_gemm2_enable", "t2v_ffn_fused_supported", "t2v_ffn_fused", "t2v_conv3x3_small_cout_supported", "t2v_conv3x3_small_cout",
                "t2v_attn_spatial_form")
EXPORTED = sorted(n for n in _SIGS if n not in EXPERIMENTAL)


def has_experimental(lib=None):
    """True if the loaded library is a T2V_EXPERIMENTAL=1 build (csrc/build.py)."""
    lib = lib if lib is not None else load()
    return hasattr(lib, "t2v_ffn_fused")


def load(path=None):
    """dlopen the library and attach signatures.  Raises NativeError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("T2V_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise NativeError(
            f"{path} not found: build it with `python __graft_entry__.py` / "
            f"`python t2v-turbo_amd/csrc/build.py` (hipcc, gfx950). There is no fallback path.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGS.items():
        if name in EXPERIMENTAL and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = _lib.t2v_last_error().decode(errors="replace") if _lib is not None else ""
        raise NativeError(f"{what} failed with code {rc}: {msg}")


def _p(t):
    return None if t is None else t.data_ptr()


def _row_stride(t):
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), "expected a row-major 2-D view"
    return t.stride(0)


def ffn_pack(w1, b1, w2, b2, gamma, beta, wdtype):
    """Weights of ``t2v_ffn_fused`` (include/t2v_hip.h): the GEGLU projection [8C, C] / [8C] (value rows, then gate rows), the
    output projection [C, 4C] / [C] and the affine of the LayerNorm in front of them -> (w1p, b1p, w2p, b2) in MFMA fragment
    order, the affine folded into the first projection."""
    C, inner = w2.shape[0], w2.shape[1]
    assert w1.shape == (2 * inner, C) and inner % 32 == 0 and C % 32 == 0
    dev = w1.device
    w1f = w1.float() * gamma.float()[None, :]
    b1f = (b1.float() if b1 is not None else torch.zeros(2 * inner, device=dev)) + w1.float() @ beta.float()
    nch, ks, rt = inner // 32, C // 32, C // 16
    lane = torch.arange(64, device=dev)
    col, kg = lane & 15, lane >> 4
    j = torch.arange(nch, device=dev)[:, None, None]
    tile = torch.arange(4, device=dev)[None, :, None]
    row0 = (tile % 2) * inner + 32 * j + 16 * (tile // 2)                      # [nch, 4, 1]: first row of a tile
    rows = row0 + col[None, None, :]                                           # [nch, 4, 64]
    e = torch.arange(8, device=dev)
    cols = 32 * torch.arange(ks, device=dev)[:, None, None] + 8 * kg[None, :, None] + e[None, None, :]   # [ks, 64, 8]
    w1p = w1f[rows[:, :, None, :, None], cols[None, None, :, :, :]].to(wdtype).contiguous()            # [nch, 4, ks, 64, 8]
    b1p = b1f[row0 + torch.arange(16, device=dev)[None, None, :]].float().contiguous()                   # [nch, 4, 16]
    orow = 16 * torch.arange(rt, device=dev)[:, None] + col[None, :]                                     # [rt, 64]
    h = torch.where(e[None, :] < 4, 4 * kg[:, None] + e[None, :], 16 + 4 * kg[:, None] + e[None, :] - 4)  # [64, 8]
    hid = 32 * torch.arange(nch, device=dev)[:, None, None] + h[None, :, :]                               # [nch, 64, 8]
    w2p = w2.float()[orow[None, :, :, None], hid[:, None, :, :]].to(wdtype).contiguous()                  # [nch, rt, 64, 8]
    b2f = (b2.float() if b2 is not None else torch.zeros(C, device=dev)).contiguous()
    return w1p, b1p, w2p, b2f


def on_tensor_device(fn):
    """Engine entry points: make the first CUDA tensor argument's GPU the current device for the launches inside.  Streams
    (``torch.cuda.current_stream()``), the library's zero page and the split-K workspace are all per CURRENT device: a model
    moved to a GPU that is not the current one (``pipe.to("cuda:1")``, a gloo rank that never called ``set_device``) would
    otherwise launch on device 0's stream against device-1 pointers."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        t = next((a for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(self, *args, **kwargs)

    return wrapped


def load_tune_table(path=None):
    """(mode, M, N, K, batch) -> (tile_cfg, split_k), produced on an MI355X by tools/tune_gemm.py."""
    import json
    path = path or os.environ.get("T2V_GEMM_TUNE_FILE") or os.path.join(_PKG, "gemm_tune.json")
    if os.environ.get("T2V_GEMM_TUNE", "1") == "0" or not os.path.exists(path):
        return {}
    with open(path) as f:
        rows = json.load(f)
    return {(r["mode"], r["M"], r["N"], r["K"], r["batch"]): (r["cfg"], r["split"]) for r in rows}


class TuneTable(dict):
    """The tile table with a NEAREST-SHAPE fallback.  gemm_tune.json holds exactly the shapes the tuner recorded (the bench latent
    16x40x64, the 16-frame VAE, the training step); any other frame count or resolution (app.py:342-348 offers 16-48 frames) gave
    every launch to the library heuristic.  What a tile choice depends on is mostly (mode, N, K) — the operand widths, the epilogue
    — and whether M fills the chip: a missing key takes the tile of the tuned entry with the same (mode, N, K, batch) whose M is
    nearest in ratio (tuned M between 0.3x and 2.5x of the asked one), and its K split only when M is within a third (else the library's own split rule).
    ``T2V_GEMM_TUNE_NEAREST=0``: exact keys only (rounds 1-4).  ``stats`` counts exact / nearest / miss lookups."""

    def __init__(self, table=(), nearest=None):
        super().__init__(table)
        self.nearest = (os.environ.get("T2V_GEMM_TUNE_NEAREST", "1") == "1") if nearest is None else nearest
        self.stats = {"exact": 0, "nearest": 0, "miss": 0}
        self._index = None

    def _near(self):
        if self._index is None or self._index[0] != len(self):
            idx = {}
            for (mode, M, N, K, batch), (cfg, split) in self.items():
                idx.setdefault((mode, N, K, batch), []).append((M, cfg, split))
            self._index = (len(self), idx)
        return self._index[1]

    def lookup(self, key):
        hit = dict.get(self, key)
        if hit is not None:
            self.stats["exact"] += 1
            return hit
        if self.nearest and len(self):
            import math
            mode, M, N, K, batch = key
            cands = self._near().get((mode, N, K, batch))
            if cands and M > 0:
                m_t, cfg, split = min(cands, key=lambda e: abs(math.log(e[0] / M)))
                ratio = m_t / M
                if 0.3 <= ratio <= 2.5:
                    self.stats["nearest"] += 1
                    return (cfg, split if 0.75 <= ratio <= 1.34 else 0)
        self.stats["miss"] += 1
        return None


class HipOps:
    """Tensor-level view of the C-ABI.  Every method launches on torch's current stream; while
    ``recording`` is a list, the raw (function, args) tuples are appended to it as well so that a
    whole forward can be replayed without Python-side tensor work (and captured into a hipGraph)."""

    act_dtype = torch.bfloat16
    is_native = True

    SPLITK_WS_BYTES = 96 << 20

    def __init__(self):
        self.lib = load()
        self.recording = None
        self._keep = []  # objects that must outlive recorded calls (descs, host arrays)
        self._ws = {}
        self.tune = TuneTable(load_tune_table())

    def workspace(self, device):
        """split-K partial-sum workspace shared by all GEMM launches of this backend (one stream)."""
        key = (device.type, device.index)
        if key not in self._ws:
            self._ws[key] = torch.empty(self.SPLITK_WS_BYTES, dtype=torch.uint8, device=device)
        return self._ws[key]

    # -- plumbing ------------------------------------------------------------------------------
    @staticmethod
    def stream():
        return torch.cuda.current_stream().cuda_stream

    def init(self):
        _check(self.lib.t2v_init(), "t2v_init")

    def _call(self, name, *args, keep=None):
        fn = getattr(self.lib, name)
        _check(fn(*args, self.stream()), name)
        if self.recording is not None:
            self.recording.append((fn, args, name))
            if keep is not None:
                self._keep.append(keep)

    def record_host_call(self, fn, args, name):
        """A host-side entry of a recorded list: ``fn(*args, stream)`` (returning 0) runs between the launches on either side
        of it on every replay (the gradient engine's all-reduce markers).  Not capturable into a hipGraph as anything but a no-op."""
        if self.recording is not None:
            self.recording.append((fn, args, name))

    # One host call per recorded LIST (t2v_replay, csrc/replay.hip) instead of one ctypes call per launch: the distillation step
    # issues ~8 800 launches, ~230 ms of Python per step, which paced the step on every box in round 3.  T2V_C_REPLAY=0: the loop.
    c_replay = os.environ.get("T2V_C_REPLAY", "1") == "1"

    def compile_recording(self, recording):
        """-> list of segments: ("c", uint64 array, n words, first launch index) for runs of C-ABI launches, ("py", fn, args, name)
        for host-side entries (the gradient exchange's markers).  The recording's argument objects stay referenced by it."""
        import struct
        segs, words, first = [], [], 0
        ids = getattr(self, "_replay_ids", None)
        if ids is None:
            ids = self._replay_ids = {}

        def flush(upto):
            nonlocal words, first
            if words:
                arr = (C.c_ulonglong * len(words))(*words)
                segs.append(("c", arr, len(words), first))
            words, first = [], upto

        for i, (fn, args, name) in enumerate(recording):
            if name not in ids:
                ids[name] = int(self.lib.t2v_replay_lookup(name.encode())) if name in _SIGS else -1
            fid = ids[name]
            if fid < 0:
                flush(i + 1)
                segs.append(("py", fn, args, name))
                continue
            types = _SIGS[name][1][:-1]
            assert len(types) == len(args), name
            words.append(fid)
            words.append(len(args))
            for t, a in zip(types, args):
                if t is C.c_float:
                    words.append(struct.unpack("<I", struct.pack("<f", float(a)))[0])
                elif t is C.c_double:
                    words.append(struct.unpack("<Q", struct.pack("<d", float(a)))[0])
                elif t in (C.c_int, C.c_longlong, C.c_uint, C.c_ulonglong):
                    words.append(int(a) & 0xFFFFFFFFFFFFFFFF)
                elif hasattr(a, "_obj"):            # C.byref(struct): the recording keeps the struct alive
                    words.append(C.addressof(a._obj))
                else:                               # raw pointer value (int) or None
                    words.append(int(a or 0) if not hasattr(a, "value") else int(a.value or 0))
        flush(len(recording))
        return segs

    # compiled launch lists kept (least recently used beyond this are dropped, one at a time).  A service that rotates through more
    # live plans than this recompiles a list per replay: T2V_MAX_PROGS sizes it (every engine keeps T2V_MAX_PLANS plans, default 4).
    max_progs = int(os.environ.get("T2V_MAX_PROGS", "16"))

    def replay(self, recording, stream, cache=True):
        """Issue a recorded launch list.  ``cache=False``: a one-shot list (the sub-lists an engine cuts for hipGraph capture) —
        compiled, replayed and forgotten.  Cached programs are an LRU keyed by the list's identity: a plan that was dropped
        elsewhere (weight change, plan eviction) ages out here instead of pinning its descriptors until 64 entries pile up, and
        eviction never throws away the programs of the plans that are still in use."""
        if not self.c_replay or not hasattr(self.lib, "t2v_replay"):
            for fn, args, name in recording:
                rc = fn(*args, stream)
                if rc != 0:
                    _check(rc, name)
            return
        if not cache:
            prog = self.compile_recording(recording)
        else:
            progs = getattr(self, "_progs", None)
            if progs is None:
                import collections
                progs = self._progs = collections.OrderedDict()
            key = id(recording)
            hit = progs.get(key)
            if hit is None or hit[0] is not recording or hit[1] != len(recording):
                hit = progs[key] = (recording, len(recording), self.compile_recording(recording))
            progs.move_to_end(key)
            while len(progs) > self.max_progs:
                progs.popitem(last=False)
            prog = hit[2]
        failed = C.c_int(-1)
        for seg in prog:
            if seg[0] == "c":
                rc = self.lib.t2v_replay(seg[1], seg[2], stream, C.byref(failed))
                if rc != 0:
                    _check(rc, recording[seg[3] + failed.value][2] if failed.value >= 0 else "t2v_replay")
            else:
                _, fn, args, name = seg
                rc = fn(*args, stream)
                if rc != 0:
                    _check(rc, name)

    # -- ops ----------------------------------------------------------------------------------------
    def gemm(self, a0, w, out, **kw):
        """``dropout``: (p, seed tensor [1] int64 on the device, site, ncols, col0) — the mask of ``dropout()`` over a
        [M, ncols] matrix whose columns col0 .. col0 + N are this launch's output, applied to alpha*acc + bias before the
        residual (include/t2v_hip.h).  ``rowstat`` / ``colstat``: fp32 tensors that receive the row / column statistics of the
        output for the next LayerNorm / GroupNorm; ``lnf``: (producer's rowstat [M, ld], eps, s [N] fp32) — this launch consumes a
        LayerNorm folded into its weights (t2v_gemm_desc::lnf_*); ``lora``: (t, u, columns per leaf, scale) — the rank-64 LoRA branch
        of the leaf(s) added in the epilogue, ``dropout`` then masks that product only (t2v_gemm_desc::lora_*).  Ask
        ``gemm_fuse_supported`` (same arguments) first."""
        self._call("t2v_gemm", C.byref(self._gemm_desc(a0, w, out, **kw)))  # the byref object holds a reference to d: a recording keeps its descriptors alive

    def conv_halo(self, a0, w, out, **kw):
        """3x3 / (3,1,1) conv on the halo-slab kernel (csrc/conv_halo.hip): same arguments as ``gemm`` except that ``w`` is the
        SLAB-MAJOR pack ``pack_conv_slab`` makes ([N][C/32][9][32], zero-padded to whole weight stages).  Ask ``conv_halo_supported`` first."""
        self._call("t2v_conv_halo", C.byref(self._gemm_desc(a0, w, out, tune_exact=True, **kw)))

    def conv_halo_supported(self, a0, w, out, **kw):
        """0: not taken by the halo kernel (use ``gemm`` with the tap-major pack); 1: taken.  (Only an EXACT tile-table entry counts
        here — a tuned K split keeps a conv on t2v_gemm, as in rounds 1-4; a nearest-shape guess made for t2v_gemm must not.)"""
        rc = self.lib.t2v_conv_halo_supported(C.byref(self._gemm_desc(a0, w, out, tune_exact=True, **kw)))
        if rc < 0:
            _check(rc, "t2v_conv_halo_supported")
        return rc

    def linear_pr(self, a0, wp, out, **kw):
        """A short-K Linear on the panel-resident kernel (csrc/linear_pr.hip): same arguments as ``gemm`` except that ``wp`` is the
        FRAGMENT pack ``pack_linear_pr`` makes of the [N, K] matrix.  Ask ``linear_pr_supported`` first."""
        self._call("t2v_linear_pr", C.byref(self._gemm_desc(a0, wp, out, **dict(kw, tile_cfg=-1))))   # (tile_cfg != 0: no tile-table lookup)

    def linear_pr_supported(self, a0, wp, out, **kw):
        """0: not taken (use ``gemm`` with the plain matrix); 1: taken."""
        rc = self.lib.t2v_linear_pr_supported(C.byref(self._gemm_desc(a0, wp, out, **dict(kw, tile_cfg=-1))))
        if rc < 0:
            _check(rc, "t2v_linear_pr_supported")
        return rc

    def gemm_fuse_supported(self, a0, w, out, **kw):
        """Would ``gemm`` with these arguments honour its rowstat / colstat / lnf request?  (Resolves tile and split-K as the
        launch does; launches nothing.)"""
        rc = self.lib.t2v_gemm_fuse_supported(C.byref(self._gemm_desc(a0, w, out, **kw)))
        if rc < 0:
            _check(rc, "t2v_gemm_fuse_supported")
        return rc == 1

    def gemm_plan(self, a0, w, out, **kw):
        """(tile id, K splits) ``gemm`` would use for these arguments; launches nothing."""
        cfg, splits = C.c_int(0), C.c_int(0)
        _check(self.lib.t2v_gemm_plan(C.byref(self._gemm_desc(a0, w, out, **kw)), C.byref(cfg), C.byref(splits)), "t2v_gemm_plan")
        return cfg.value, splits.value

    def _gemm_desc(self, a0, w, out, *, M, N, a1=None, mode=GEMM_LINEAR, n_img=0, h=0, wd=0, frames=0, bias=None,
                   rowvec=None, rowvec_div=0, residual=None, act=ACT_NONE, alpha=1.0, batch=1, batch_inner=1,
                   a_strides=(0, 0), w_strides=(0, 0), o_strides=(0, 0), tile_cfg=0, split_k=0, dropout=None, ln=None,
                   rowstat=None, colstat=None, lnf=None, lora=None, tune_exact=False, ln_in=None, gn_in=None):
        d = GemmDesc()
        d.a0, d.c0, d.lda0 = _p(a0), a0.shape[1], _row_stride(a0)
        if a1 is not None:
            d.a1, d.c1, d.lda1 = _p(a1), a1.shape[1], _row_stride(a1)
        d.mode, d.n_img, d.h_in, d.w_in, d.frames = mode, n_img, h, wd, frames
        d.M, d.N = M, N
        d.w, d.ldw = _p(w), _row_stride(w)
        d.batch, d.batch_inner = batch, batch_inner
        d.a_stride0, d.a_stride1 = a_strides
        d.w_stride0, d.w_stride1 = w_strides
        d.o_stride0, d.o_stride1 = o_strides
        d.alpha = alpha
        d.bias = _p(bias)
        if rowvec is not None:
            d.rowvec, d.rowvec_div, d.ld_rowvec = _p(rowvec), rowvec_div, _row_stride(rowvec)
        if residual is not None:
            d.residual, d.ldr = _p(residual), _row_stride(residual)
        d.act = act
        d.out, d.ldo = _p(out), _row_stride(out)
        d.out_f32 = 1 if out.dtype == torch.float32 else 0
        if dropout is not None and dropout[0] > 0:
            p_drop, seed_t, site, ncols, col0 = dropout
            t = float(p_drop) * 4294967296.0
            d.drop_thr = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
            d.drop_seed, d.drop_site, d.drop_inv_keep = _p(seed_t), int(site), dropout_inv_keep(p_drop)
            d.drop_ncols, d.drop_col0 = int(ncols), int(col0)
            split_k = 1
        taps = {GEMM_LINEAR: 1, GEMM_TCONV3: 3}.get(mode, 9)
        # a split_k argument without a tile id is a hint (the training engine's token-contracted weight gradients): a tuned entry wins
        tuned = None
        if tile_cfg == 0:
            key = (mode, M, N, taps * (d.c0 + d.c1), batch)
            tuned = self.tune.lookup(key) if (hasattr(self.tune, "lookup") and not tune_exact) else self.tune.get(key)
        # (a nearest-shape hit whose K-split plan is not trusted comes back with split 0: the caller's hint then stands)
        d.tile_cfg, d.split_k = (tuned[0], tuned[1] or split_k) if tuned else (tile_cfg, split_k)
        if d.drop_thr:
            d.split_k = 1
        if ln is not None:  # (gamma fp32 [N], beta fp32 [N], eps, out2 bf16 [M, N]): LayerNorm(out) as a second output (N == 320)
            gamma, beta, eps, out2 = ln
            d.ln_gamma, d.ln_beta, d.ln_eps, d.ln_out, d.ld_ln_out = _p(gamma), _p(beta), float(eps), _p(out2), _row_stride(out2)
        if ln_in is not None:   # (gamma fp32 [K], beta fp32 [K], eps): LayerNorm of the A rows in t2v_linear_pr's panel fill
            assert ln is None, "ln (second output) and ln_in (input rows) share the ln_* fields"
            gamma, beta, eps = ln_in
            assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == a0.shape[1] == beta.numel()
            d.ln_gamma, d.ln_beta, d.ln_eps, d.ln_in = _p(gamma), _p(beta), float(eps), 1
        if gn_in is not None:   # (coef fp32 [units, 2, K], rows per unit): GroupNorm affine of the A rows in t2v_linear_pr's panel fill
            coef, rpu = gn_in
            assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.numel() == (M // rpu) * 2 * a0.shape[1]
            d.gn_coef, d.gn_rows_per_unit = _p(coef), int(rpu)
        if rowstat is not None:   # fp32 [M, ld]: (sum, sumsq) per 32-column block of every output row
            assert rowstat.dtype == torch.float32 and rowstat.shape[0] == M
            d.rowstat_out, d.ld_rowstat = _p(rowstat), _row_stride(rowstat)
        if colstat is not None:   # fp32 [M / 32, N, 2]: (sum, sumsq) per column of every 32-row slab
            assert colstat.dtype == torch.float32 and colstat.is_contiguous() and colstat.numel() == (M // 32) * N * 2
            d.colstat_out = _p(colstat)
        if lnf is not None:
            stats, eps, s_vec = lnf
            assert stats.dtype == torch.float32 and s_vec.dtype == torch.float32 and stats.shape[0] == M
            d.lnf_stats, d.lnf_ld, d.lnf_nblk, d.lnf_eps, d.lnf_s = _p(stats), _row_stride(stats), a0.shape[1] // 32, float(eps), _p(s_vec)
        if lora is not None:   # (t [M, 64 leaves] bf16, u [N, 64] bf16, columns per leaf, scale): the LoRA branch in this launch's epilogue
            t, u, n_leaf, scale = lora
            assert t.dtype == torch.bfloat16 and u.dtype == torch.bfloat16 and t.shape[0] == M and u.shape[0] == N and u.shape[1] >= 64
            d.lora_t, d.ld_lora_t, d.lora_u, d.ld_lora_u = _p(t), _row_stride(t), _p(u), _row_stride(u)
            d.lora_n_leaf, d.lora_scale = int(n_leaf), float(scale)
        ws = self.workspace(a0.device)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        return d

    def ffn_fused_supported(self, C_):
        return hasattr(self.lib, "t2v_ffn_fused_supported") and bool(self.lib.t2v_ffn_fused_supported(int(C_)))

    def ffn_fused(self, x, w1p, b1p, w2p, b2, eps, out):
        """out = x + FF(LayerNorm(x)) (GEGLU feed-forward) in one launch; packed weights: ``ffn_pack`` (include/t2v_hip.h)."""
        self._call("t2v_ffn_fused", _p(x), _row_stride(x), x.shape[0], x.shape[1], _p(w1p), _p(b1p), _p(w2p), _p(b2), float(eps),
                   _p(out), _row_stride(out))

    def conv_small(self, x, n_img, h, w, wgt, bias, out):
        self._call("t2v_conv3x3_small_cin", _p(x), n_img, h, w, x.shape[1], _p(wgt), _p(bias), out.shape[1], _p(out))

    def conv_small_cout_supported(self, w, cin, cout):
        return hasattr(self.lib, "t2v_conv3x3_small_cout_supported") and self.lib.t2v_conv3x3_small_cout_supported(int(w), int(cin), int(cout)) == 1

    def conv_small_cout(self, x, n_img, h, w, wgt, bias, out):
        """Direct 3x3 conv to 1..4 output channels: x bf16 [M, cin], wgt fp32 [cout, 9 * cin] (tap-major), out [M, cout] fp32 or bf16."""
        assert x.dtype == torch.bfloat16 and wgt.dtype == torch.float32 and out.dtype in (torch.float32, torch.bfloat16)
        self._call("t2v_conv3x3_small_cout", _p(x), _row_stride(x), n_img, h, w, x.shape[1], _p(wgt), _p(bias), out.shape[1], _p(out),
                   _row_stride(out), 1 if out.dtype == torch.float32 else 0)

    def gn_ws_floats(self, n_units, rows_per_unit, groups=32):
        return int(self.lib.t2v_gn_ws_floats(n_units, rows_per_unit, groups))

    def gn_stats(self, x0, x1, n_units, rows_per_unit, eps, ws, stats, groups=32):
        self._call("t2v_gn_stats", _p(x0), x0.shape[1], _row_stride(x0), _p(x1),
                   0 if x1 is None else x1.shape[1], 0 if x1 is None else _row_stride(x1),
                   n_units, rows_per_unit, groups, eps, _p(ws), _p(stats))

    def gn_apply(self, x0, x1, n_units, rows_per_unit, stats, gamma, beta, silu, out, groups=32):
        self._call("t2v_gn_apply", _p(x0), x0.shape[1], _row_stride(x0), _p(x1),
                   0 if x1 is None else x1.shape[1], 0 if x1 is None else _row_stride(x1),
                   n_units, rows_per_unit, groups, _p(stats), _p(gamma), _p(beta), int(silu), _p(out), _row_stride(out))

    def group_norm_ws_floats(self, n_units, rows_per_unit, groups, channels):
        return int(self.lib.t2v_group_norm_ws_floats(n_units, rows_per_unit, groups, channels))

    @staticmethod
    def _pf(t):
        """(pointer, bytes) of a contiguous tensor the next launch will stream (``prefetch=`` of the GroupNorm ops), or (None, 0)."""
        return (None, 0) if t is None else (t.data_ptr(), t.numel() * t.element_size())

    def group_norm(self, x0, x1, n_units, rows_per_unit, eps, gamma, beta, silu, ws, out, groups=32, prefetch=None):
        self._call("t2v_group_norm", _p(x0), x0.shape[1], _row_stride(x0), _p(x1),
                   0 if x1 is None else x1.shape[1], 0 if x1 is None else _row_stride(x1),
                   n_units, rows_per_unit, groups, eps, _p(gamma), _p(beta), int(silu), _p(ws), _p(out), _row_stride(out),
                   *self._pf(prefetch))

    def group_norm_cs_ws_floats(self, n_units, rows_per_unit, groups):
        return int(self.lib.t2v_group_norm_cs_ws_floats(n_units, rows_per_unit, groups))

    def group_norm_cs(self, cs0, cs1, x0, x1, n_units, rows_per_unit, eps, gamma, beta, silu, ws, out, groups=32, prefetch=None):
        """GroupNorm(+SiLU) on the column statistics the producing GEMMs wrote (``gemm(colstat=...)``): no statistics pass."""
        self._call("t2v_group_norm_cs", _p(cs0), _p(cs1), _p(x0), x0.shape[1], _row_stride(x0), _p(x1),
                   0 if x1 is None else x1.shape[1], 0 if x1 is None else _row_stride(x1),
                   n_units, rows_per_unit, groups, eps, _p(gamma), _p(beta), int(silu), _p(ws), _p(out), _row_stride(out),
                   *self._pf(prefetch))

    def gn_stats_cs(self, cs0, cs1, c0, c1, n_units, rows_per_unit, eps, ws, stats, groups=32):
        """``gn_stats`` from the producers' column statistics (cs [rows / 32, C, 2] per part) instead of the tensor."""
        self._call("t2v_gn_stats_cs", _p(cs0), c0, _p(cs1), c1 if cs1 is not None else 0, n_units, rows_per_unit, groups, eps, _p(ws), _p(stats))

    def layernorm(self, x, gamma, beta, eps, out):
        self._call("t2v_layernorm", _p(x), _row_stride(x), x.shape[0], x.shape[1], _p(gamma), _p(beta), eps, _p(out),
                   _row_stride(out))

    def softmax_rows(self, s, rows, n, n_pad, ld):
        self._call("t2v_softmax_rows", _p(s), rows, n, n_pad, ld)

    def attn_spatial(self, q, k, vt, ld_vt, out, n_img, seq_q, seq_kv, heads, kv_div, scale, vt_img_stride=0):
        self._call("t2v_attn_spatial", _p(q), _row_stride(q), _p(k), _row_stride(k), _p(vt), ld_vt, vt_img_stride,
                   _p(out), _row_stride(out), n_img, seq_q, seq_kv, heads, kv_div, scale)

    def attn_temporal(self, q, k, v, out, n_clips, frames, hw, heads, scale, probs=None):
        self._call("t2v_attn_temporal", _p(q), _row_stride(q), _p(k), _row_stride(k), _p(v), _row_stride(v), _p(out),
                   _row_stride(out), n_clips, frames, hw, heads, scale, _p(probs))

    def ncfhw_to_tokens(self, x, out):
        b, c, f, h, w = x.shape
        assert x.is_contiguous()
        self._call("t2v_ncfhw_to_tokens", _p(x), _DT[x.dtype], b, c, f, h * w, _p(out), _row_stride(out))

    def tokens_to_ncfhw(self, tok, out):
        b, c, f, h, w = out.shape
        assert out.is_contiguous()
        self._call("t2v_tokens_to_ncfhw", _p(tok), 1 if tok.dtype == torch.float32 else 0, _row_stride(tok), b, c, f,
                   h * w, _p(out), _DT[out.dtype])

    def timestep_embedding(self, t, dim, guidance_style, out):
        assert t.dtype in (torch.int64, torch.float32) and t.is_contiguous()
        self._call("t2v_timestep_embedding", _p(t), 1 if t.dtype == torch.float32 else 0, t.numel(), dim,
                   int(guidance_style), _p(out))

    def silu(self, x, out):
        self._call("t2v_silu", _p(x), _p(out), x.numel())

    def fill_zero(self, t):
        assert t.is_contiguous()
        self._call("t2v_fill_zero", _p(t), t.numel() * t.element_size())

    def cast(self, x, out):
        assert x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel()
        self._call("t2v_cast", _p(x), _DT[x.dtype], _p(out), _DT[out.dtype], x.numel())

    def lincomb3(self, x, y, z, ca, cb, cc, out):
        nb = len(ca)
        arr = (C.c_float * nb)
        ha, hb, hc = arr(*ca), (arr(*cb) if cb is not None else None), (arr(*cc) if cc is not None else None)
        self._call("t2v_lincomb3", _p(x), _p(y), _p(z), C.cast(ha, C.c_void_p), C.cast(hb, C.c_void_p) if hb else None,
                   C.cast(hc, C.c_void_p) if hc else None, nb, x.numel() // nb, _p(out), keep=(ha, hb, hc))

    # -- backward pieces (VAE decoder dX) ---------------------------------------------------------------------
    def gn_bwd_ws_floats(self, n_units, rows_per_unit, groups=32):
        return int(self.lib.t2v_gn_bwd_ws_floats(n_units, rows_per_unit, groups))

    def gn_bwd(self, x, n_units, rows_per_unit, stats, gamma, beta, silu, dy, resid, ws, dx, groups=32, x1=None):
        if x1 is None and x.shape[1] <= 2048:
            self._call("t2v_gn_bwd", _p(x), _row_stride(x), x.shape[1], n_units, rows_per_unit, groups, _p(stats), _p(gamma),
                       _p(beta), int(silu), _p(dy), _row_stride(dy), _p(resid), 0 if resid is None else _row_stride(resid),
                       _p(ws), _p(dx), _row_stride(dx))
        else:  # virtual channel concat [x | x1]
            self._call("t2v_gn_bwd2", _p(x), x.shape[1], _row_stride(x), _p(x1), 0 if x1 is None else x1.shape[1],
                       0 if x1 is None else _row_stride(x1), n_units,
                       rows_per_unit, groups, _p(stats), _p(gamma), _p(beta), int(silu), _p(dy), _row_stride(dy), _p(resid),
                       0 if resid is None else _row_stride(resid), _p(ws), _p(dx), _row_stride(dx))

    # -- backward pieces (UNet dX; device kernels not yet validated on hardware, see include/t2v_hip.h) ------------------
    def layernorm_bwd(self, x, gamma, eps, dy, resid, dx):
        self._call("t2v_layernorm_bwd", _p(x), _row_stride(x), x.shape[0], x.shape[1], _p(gamma), eps, _p(dy), _row_stride(dy),
                   _p(resid), 0 if resid is None else _row_stride(resid), _p(dx), _row_stride(dx))

    def geglu_fwd(self, h, out):
        self._call("t2v_geglu_fwd", _p(h), _row_stride(h), h.shape[0], out.shape[1], _p(out), _row_stride(out))

    def geglu_bwd(self, h, dy, dh):
        self._call("t2v_geglu_bwd", _p(h), _row_stride(h), _p(dy), _row_stride(dy), h.shape[0], dy.shape[1], _p(dh), _row_stride(dh))

    def scatter2x(self, src, n_img, h, w, H, W, out):
        self._call("t2v_scatter2x", _p(src), n_img, h, w, src.shape[1], H, W, _p(out))

    def add(self, a, b, out):
        self._call("t2v_add_bf16", _p(a), _row_stride(a), _p(b), _row_stride(b), _p(out), _row_stride(out), a.shape[0], a.shape[1])

    def attn_temporal_bwd(self, q, k, v, do, dprobs, dq, dk, dv, n_clips, frames, hw, heads, scale):
        self._call("t2v_attn_temporal_bwd", _p(q), _row_stride(q), _p(k), _row_stride(k), _p(v), _row_stride(v), _p(do),
                   _row_stride(do), _p(dprobs), _p(dq), _row_stride(dq), _p(dk), _row_stride(dk), _p(dv), _row_stride(dv),
                   n_clips, frames, hw, heads, scale)

    def softmax_bwd_rows(self, p, dp, rows, n, n_pad, ld):
        self._call("t2v_softmax_bwd_rows", _p(p), _p(dp), rows, n, n_pad, ld)

    def transpose(self, src, rows, cols, out, batch=1, in_stride=0, out_stride=0):
        """out[b][c][r] = src[b][r][c]; row strides taken from the tensors, batch strides in elements."""
        self._call("t2v_transpose_bf16", _p(src), _row_stride(src), rows, cols, _p(out), _row_stride(out), batch, in_stride,
                   out_stride)

    def sumpool2x2(self, src, n_img, h, w, out):
        self._call("t2v_sumpool2x2", _p(src), n_img, h, w, src.shape[1], _p(out))

    def adamw_step(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        self._call("t2v_adamw_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2,
                   eps, weight_decay, step, grad_scale)

    def ema_update(self, target, src, rate):
        self._call("t2v_ema_update", _p(target), _p(src), rate, target.numel())

    def sumsq(self, x, ws, out):
        self._call("t2v_sumsq", _p(x), x.numel(), _p(ws), _p(out))

    def gather(self, src, idx, out, alpha=1.0, accumulate=False):
        """out.flat[i] = alpha * src.flat[idx[i]] (0 where idx < 0), or += with ``accumulate``; src fp32, idx int32."""
        assert src.dtype == torch.float32 and idx.dtype == torch.int32 and idx.numel() == out.numel()
        self._call("t2v_gather_f32", _p(src), _p(idx), alpha, _p(out), _DT[out.dtype], 1 if accumulate else 0, out.numel())

    def attn_spatial_bwd(self, q, k, v, v_img_stride, v_head_stride, kt, qt, dot, dout, o, l2, dsum, dq, dk, dv, n_img, seq, heads, scale):
        """Flash-style backward of the spatial self-attention (see include/t2v_hip.h); kt / qt / dot: [n_img*heads*64, padded seq]."""
        self._call("t2v_attn_spatial_bwd", _p(q), _row_stride(q), _p(k), _row_stride(k), _p(v), _row_stride(v), v_img_stride,
                   v_head_stride, _p(kt), _row_stride(kt), _p(qt), _p(dot), _row_stride(qt), _p(dout), _row_stride(dout), _p(o),
                   _row_stride(o), _p(l2), _p(dsum), _row_stride(l2), _p(dq), _row_stride(dq), _p(dk), _row_stride(dk), _p(dv),
                   _row_stride(dv), n_img, seq, seq, heads, scale)

    def gn_coef_cs_supported(self, cs0, cs1, c0, c1, n_units, rows_per_unit, groups=32):
        return self.lib.t2v_gn_coef_cs_supported(_p(cs0), c0, _p(cs1), c1, n_units, rows_per_unit, groups) == 1

    def gn_coef_cs(self, cs0, cs1, c0, c1, n_units, rows_per_unit, eps, gamma, beta, coef, groups=32):
        """coef fp32 [n_units, 2, c0 + c1]: the GroupNorm's per-channel affine from the producers' column statistics (t2v_gn_coef_cs)."""
        assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.numel() == n_units * 2 * (c0 + c1)
        self._call("t2v_gn_coef_cs", _p(cs0), c0, _p(cs1), c1, n_units, rows_per_unit, groups, float(eps), _p(gamma), _p(beta), _p(coef))

    # ---- base-weight gradients for full fine-tuning (csrc/full_grad.hip) -----------------------------------------------------------
    def im2col_rows(self, mode, n_img, h, w):
        return int(self.lib.t2v_im2col_rows(int(mode), int(n_img), int(h), int(w)))

    def im2col(self, x0, x1, mode, n_img, h, w, frames, out):
        """out[m][tap * C + c] = x[src(m, tap)][c] (bf16; x = [x0 | x1] on the (n_img, h, w) input grid) for a conv gather mode of ``gemm``."""
        self._call("t2v_im2col_bf16", _p(x0), x0.shape[1], _row_stride(x0), _p(x1), 0 if x1 is None else x1.shape[1],
                   0 if x1 is None else _row_stride(x1), int(mode), n_img, h, w, frames, _p(out), _row_stride(out))

    def repack_conv(self, w, out, kind):
        """out (bf16 pack, in place) from the fp32 conv parameter w [N, C, k...]: kind 0 = tap-major forward pack [N, taps * C],
        kind 1 = data-gradient pack [C, taps * N] with mirrored taps (t2v_repack_conv_f32)."""
        N, Cc = w.shape[0], w.shape[1]
        taps = w.numel() // (N * Cc)
        assert w.dtype == torch.float32 and w.is_contiguous() and out.dtype == torch.bfloat16
        assert tuple(out.shape) == ((N, taps * Cc) if kind == 0 else (Cc, taps * N))
        self._call("t2v_repack_conv_f32", _p(w), N, Cc, taps, kind, _p(out), _row_stride(out))

    def norm_affine_grad_ws_floats(self, rows, sum_rows, channels):
        return int(self.lib.t2v_norm_affine_grad_ws_floats(int(rows), int(sum_rows), int(channels)))

    def norm_affine_grad(self, x0, x1, dy, *, kind, sum_rows, ws, dgamma=None, dbeta=None, rows_per_unit=0, groups=0, stats=None, eps=0.0,
                         gamma=None, beta=None, silu=False):
        """Per-channel row sums (include/t2v_hip.h): ``kind`` 0 GroupNorm (``stats``), 1 LayerNorm (``eps``), 2 column sums of ``dy``;
        ``dgamma`` / ``dbeta``: fp32 [rows / sum_rows, C] (row slices of wider buffers allowed)."""
        rows = dy.shape[0]
        c0 = dy.shape[1] if kind == 2 else x0.shape[1]
        self._call("t2v_norm_affine_grad", _p(x0) if kind != 2 else None, c0, _row_stride(x0) if kind != 2 else 0, _p(x1),
                   0 if x1 is None else x1.shape[1], 0 if x1 is None else _row_stride(x1), rows, int(sum_rows), int(kind), int(rows_per_unit),
                   int(groups), _p(stats), float(eps), _p(gamma), _p(beta), 1 if silu else 0, _p(dy), _row_stride(dy), _p(dgamma),
                   0 if dgamma is None else _row_stride(dgamma), _p(dbeta), 0 if dbeta is None else _row_stride(dbeta), _p(ws))

    def wgrad_tn(self, a, b, out, alpha=1.0, splits=0):
        """out[R, C] (fp32) = alpha * a^T b for token-major bf16 a [M, R], b [M, C] (column slices allowed)."""
        assert out.dtype == torch.float32 and a.shape[0] == b.shape[0] and out.shape == (a.shape[1], b.shape[1])
        ws = self.workspace(a.device)
        self._call("t2v_wgrad_tn", _p(a), _row_stride(a), _p(b), _row_stride(b), a.shape[0], a.shape[1], b.shape[1], alpha, _p(out),
                   _row_stride(out), ws.data_ptr(), ws.numel(), splits)

    def wgrad_tn_group(self, problems):
        """``problems``: [(a, b, out, alpha)] with token-major bf16 a [M, R], b [M, C] and fp32 out [R, C] — the weight gradients
        of one LoRA group in ONE launch pair (t2v_wgrad_tn_group); more than 8 problems go out in chunks of 8."""
        for k in range(0, len(problems), WGRAD_GROUP_MAX):
            chunk = problems[k:k + WGRAD_GROUP_MAX]
            arr = (WgradProblem * len(chunk))()
            for d, (a, b, out, alpha) in zip(arr, chunk):
                assert out.dtype == torch.float32 and a.shape[0] == b.shape[0] and out.shape == (a.shape[1], b.shape[1])
                d.a, d.b, d.out, d.M = _p(a), _p(b), _p(out), a.shape[0]
                d.lda, d.ldb, d.ldo, d.R, d.C, d.alpha = _row_stride(a), _row_stride(b), _row_stride(out), a.shape[1], b.shape[1], alpha
            ws = self.workspace(chunk[0][0].device)
            self._call("t2v_wgrad_tn_group", C.cast(arr, C.c_void_p), len(chunk), ws.data_ptr(), ws.numel(), keep=arr)

    def transpose_pad(self, src, rows, cols, out, batch=1, in_stride=0, out_stride=0):
        """out[b][c][r] = src[b][r][c], zero for rows <= r < roundup(rows, 64) (16-byte accesses; see include/t2v_hip.h)."""
        self._call("t2v_transpose_pad_bf16", _p(src), _row_stride(src), rows, cols, _p(out), _row_stride(out), batch, in_stride,
                   out_stride)

    def dropout(self, x, resid, out, ncols, p, seed, site):
        """out[:, :ncols] = dropout(x[:, :ncols]) (+ resid); mask = f(seed[0], site, row * ncols + col); seed: int64 device tensor."""
        assert seed.dtype == torch.int64 and x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
        self._call("t2v_dropout_bf16", _p(x), _row_stride(x), _p(resid), 0 if resid is None else _row_stride(resid), _p(out),
                   _row_stride(out), x.shape[0], ncols, p, _p(seed), site)

    def lcm_step(self, x, eps, noise, sa_t, sb_t, c_skip, c_out, sa_p, sb_p, prev, denoised):
        self._call("t2v_lcm_step", _p(x), _p(eps), _DT[eps.dtype], _p(noise), sa_t, sb_t, c_skip, c_out, sa_p, sb_p,
                   x.numel(), _p(prev), _p(denoised))
