"""ctypes binding of libtag_hip.so (the C ABI declared in include/tag_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  ``import torch`` must precede the CDLL load so that the HIP runtime
PyTorch ships (SONAME libamdhip64.so.7) is the one the library binds to -- streams and device
pointers are then interchangeable with torch's.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_uint64, c_void_p

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TAG_HIP_LIB: a privately built library (profiling / ablation builds of tools/*.sh); the product is the in-tree one
LIB_PATH = os.environ.get("TAG_HIP_LIB") or os.path.join(_HERE, "libtag_hip.so")

# = TAG_ABI_VERSION of include/tag_hip.h: bumped whenever an EXISTING entry point changes its argument list, so that a stale
# libtag_hip.so (git-ignored, shipped separately) is refused instead of being called with shifted arguments
ABI_VERSION = 3

P = c_void_p
_SIGS = {
    "tag_abi_version": (c_int, []),
    "tag_build_id": (c_char_p, []),
    "tag_last_error": (c_char_p, []),
    "tag_device_cu_count": (c_int, []),
    "tag_mfma_probe": (c_int, [c_int, c_int, c_int, ctypes.c_uint, P, P]),
    "tag_mfma_probe_flop": (c_double, [c_int, c_int, c_int]),
    "tag_valu_probe": (c_int, [c_int, c_int, ctypes.c_uint, P, P]),
    "tag_stream_create_cu_mask": (c_int, [P, c_int, P]),
    "tag_logmel_forward": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P]),
    "tag_bn_stats_ws_bytes": (c_size_t, [c_long, c_int]),
    "tag_bn_stats": (c_int, [P, c_long, c_int, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, P]),
    "tag_bn_eval_affine": (c_int, [P, P, P, P, c_float, c_int, P, P, P]),
    "tag_affine_forward": (c_int, [P, c_long, c_int, P, P, P, P]),
    "tag_bn_param_grad": (c_int, [P, P, c_long, c_int, P, P, P, P, P, P]),
    "tag_pack_conv_weight": (c_int, [P, P, P, c_int, c_int, P]),
    "tag_conv3x3_stats_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "tag_conv3x3_x3_stats_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "tag_conv3x3_x3_bf16_stats_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "tag_conv_rows_enable": (c_int, [c_int]),
    "tag_wgrad_dma_enable": (c_int, [c_int]),
    "tag_bn_stats_from_partials_ws_bytes": (c_size_t, [c_int, c_int]),
    "tag_bn_stats_from_partials": (c_int, [P, c_int, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, P]),
    "tag_conv3x3_forward": (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_dgrad_bnsums": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "tag_conv3x3_wino_ok": (c_int, [c_int] * 5),
    "tag_pack_conv_weight_wino": (c_int, [P, P, P, c_int, c_int, P]),
    "tag_conv3x3_wino_ws_bytes": (c_size_t, [c_int] * 5),
    "tag_conv3x3_wino_stats_rows": (c_int, [c_int] * 4),
    "tag_conv3x3_wino_wgrad_ws_bytes": (c_size_t, [c_int] * 5),
    "tag_conv3x3_wino_wgrad_can_reuse_v": (c_int, [c_int] * 5),
    "tag_conv3x3_wino_wgrad": (c_int, [P, c_int, P, P, P, P] + [c_int] * 5 + [P, P, P]),
    "tag_conv3x3_wino_forward": (c_int, [P, P, c_int, P, P, P, P] + [c_int] * 5 + [P, P, P]),
    "tag_conv3x3_wino_forward_bnrelu_pool_eval": (c_int, [P, P, c_int, P, P, P, P, P] + [c_int] * 8 + [P, P]),
    "tag_conv3x3_wino_dgrad_poolsums": (c_int, [P] * 9 + [c_int] * 10 + [c_float, c_uint64, P, P]),
    "tag_conv3x3_wino_dgrad_bnsums": (c_int, [P] * 9 + [c_int] * 5 + [P, P]),
    "tag_conv3x3_forward_bnrelu_pool_eval": (c_int, [P, P, c_int, P, P, P, P, P] + [c_int] * 8 + [P]),
    "tag_conv3x3_dgrad_poolsums": (c_int, [P] * 9 + [c_int] * 10 + [c_float, c_uint64, P]),
    "tag_conv3x3_dgrad_poolsums_bf16_rows": (c_int, [c_int] * 5),
    "tag_conv3x3_dgrad_poolsums_bf16": (c_int, [P] * 9 + [c_int] * 10 + [c_float, c_uint64, P]),
    "tag_bnrelu_pool_backward_apply": (c_int, [P] * 10 + [c_int] * 7 + [c_float, c_uint64, c_int, P]),
    "tag_bnrelu_pool_backward_apply_bf16": (c_int, [P] * 10 + [c_int] * 7 + [c_float, c_uint64, c_int, P]),
    "tag_bn_grad_from_partials_ws_bytes": (c_size_t, [c_int, c_int]),
    "tag_bn_grad_from_partials": (c_int, [P, c_int, c_int, P, P, P, P]),
    "tag_bnrelu_backward_apply": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P]),
    "tag_conv3x3_x3_pack_bytes": (c_size_t, [c_int, c_int]),
    "tag_pack_conv_weight_x3": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "tag_conv3x3_forward_x3": (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_wgrad_x3_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "tag_conv3x3_wgrad_x3": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "tag_conv3x3_wgrad": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_c1_forward": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_c1_stats_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "tag_conv3x3_c1_forward_stats": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_c1_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "tag_conv3x3_c1_wgrad": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_c1_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_c1_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "tag_conv3x3_c1_backward": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_c1_backward_bnrelu": (c_int, [P] * 12 + [c_int, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "tag_waveform_f16_to_f32_padded": (c_int, [P, P, c_int, c_int, P, P, P]),
    "tag_bnact_pool_forward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_float, c_uint64, P]),
    "tag_bn_backward_ws_bytes": (c_size_t, [c_long, c_int]),
    "tag_bnrelu_pool_backward": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_float, c_uint64, c_int, P, P]),
    "tag_bnrelu_backward": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P, P]),
    "tag_bn_act_backward": (c_int, [P, c_int, P, P, P, P, P, P, P, c_long, c_int, c_int, P, P]),
    "tag_lppool_leaky_backward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, P]),
    "tag_dropout_mask": (c_int, [c_uint64, c_long, c_float, P, P]),
    "tag_dropout_mask_pooled": (c_int, [c_uint64, c_long, c_float, P, P]),
    "tag_mean_w_forward": (c_int, [P, c_long, c_int, c_int, c_float, c_uint64, P, P]),
    "tag_mean_w_backward": (c_int, [P, c_long, c_int, c_int, c_float, c_uint64, P, P]),
    "tag_gemm_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tag_gemm": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P]),
    "tag_gemm_bf16": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P]),
    "tag_colsum_ws_bytes": (c_size_t, [c_long, c_int]),
    "tag_colsum": (c_int, [P, c_int, c_long, c_int, P, P, P]),
    "tag_relu_backward": (c_int, [P, P, P, c_long, P]),
    "tag_gru_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tag_gru_timed_out": (c_int, [P]),
    "tag_gru_disable_xcd_fast": (c_int, []),
    "tag_gru_forward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "tag_gru_backward": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "tag_embed_mean_forward": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_embed_mean_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_embed_check_ids": (c_int, [P, c_long, c_int, P, P]),
    "tag_match_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_match_backward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_align_dot_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "tag_align_dot_dscore": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_l2norm_rows_forward": (c_int, [P, P, c_long, c_int, P]),
    "tag_l2norm_rows_backward": (c_int, [P, P, P, c_long, c_int, P]),
    "tag_roberta_embed_ln": (c_int, [P, P, P, P, P, P, c_float, P, c_int, c_int, c_int, c_int, P]),
    "tag_add_layernorm": (c_int, [P, P, P, P, c_float, P, c_long, c_int, P]),
    "tag_mha_small": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_addattn_forward": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_addattn_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "tag_addattn_backward": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "tag_mul": (c_int, [P, P, P, c_long, P]),
    "tag_gate_backward": (c_int, [P, P, P, P, c_int, P, c_long, P]),
    "tag_rowdot_sigmoid_forward": (c_int, [P, P, P, c_long, c_int, c_int, P]),
    "tag_rowdot_sigmoid_backward": (c_int, [P, P, P, P, P, c_long, c_int, c_int, P]),
    "tag_rowpair_forward": (c_int, [P, P, P, c_long, c_int, c_int, c_int, c_int, P]),
    "tag_rowpair_backward": (c_int, [P, P, P, P, P, c_long, c_int, c_int, c_int, c_int, P]),
    "tag_embed_tokens_backward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_match_group_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_match_group_backward": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_linear_softmax_pool_forward": (c_int, [P, P, P, c_long, c_int, c_int, P]),
    "tag_linear_softmax_pool_backward": (c_int, [P, P, P, P, c_long, c_int, c_int, P]),
    "tag_meanmean_pool_forward": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "tag_meanmean_pool_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "tag_maxmargin_forward": (c_int, [P, c_int, c_float, c_float, c_int, P, P]),
    "tag_maxmargin_backward": (c_int, [P, c_int, c_float, c_float, c_int, P, P, P]),
    "tag_frame_bce_forward": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, P]),
    "tag_frame_bce_backward": (c_int, [P, c_int, P, c_int, P, c_int, c_int, P, P, P]),
    "tag_segments": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, P, P, c_int, P]),
    "tag_conv3x3_c1_forward_stats_bf16": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_c1_backward_bf16": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_c1_backward_bnrelu_bf16": (c_int, [P] * 12 + [c_int, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "tag_conv3x3_forward_x3_bf16": (c_int, [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_conv3x3_wgrad_x3_bf16": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "tag_bnact_pool_forward_bf16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_float, c_uint64, P]),
    "tag_bnrelu_pool_backward_bf16": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                              c_float, c_uint64, c_int, P, P]),
    "tag_bnrelu_backward_bf16": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P, P]),
    "tag_conv3x3_dgrad_bnsums_bf16": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_bnrelu_backward_apply_bf16": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P]),
    "tag_mean_w_forward_bf16": (c_int, [P, c_long, c_int, c_int, c_float, c_uint64, P, P]),
    "tag_mean_w_backward_bf16": (c_int, [P, c_long, c_int, c_int, c_float, c_uint64, P, P]),
    "tag_mha_cross_forward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, P]),
    "tag_mha_cross_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "tag_mha_cross_backward": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, P, P]),
    "tag_resln_head_forward": (c_int, [P, P, P, P, P, P, P, P, P, c_long, c_int, c_float, c_float, c_uint64, P]),
    "tag_resln_head_backward": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_float, c_uint64, P]),
    "tag_sim_pool_forward": (c_int, [P, P, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_sim_pool_backward": (c_int, [P, P, P, P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "tag_attnpool_forward": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "tag_attnpool_backward": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "tag_upsample_linear_forward": (c_int, [P, P, c_long, c_int, c_int, P]),
    "tag_upsample_linear_backward": (c_int, [P, P, c_long, c_int, c_int, P]),
    "tag_group_expand_forward": (c_int, [P, P, c_long, c_int, c_long, P]),
    "tag_group_expand_backward": (c_int, [P, P, c_long, c_int, c_long, P]),
    "tag_sumsq_ws_bytes": (c_size_t, [c_long]),
    "tag_sumsq": (c_int, [P, c_long, P, P, P]),
    "tag_adam_step": (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_int, P, c_float, c_float, P]),
}

_lib = None


def load():
    """Load libtag_hip.so (once) and attach argtypes.  Raises RuntimeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C texttoaudiogrounding_amd/csrc`). There is no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    # version first: a stale library then says so, instead of failing on the first symbol it does not have yet
    try:
        lib.tag_abi_version.restype = c_int
        have = lib.tag_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise RuntimeError(f"libtag_hip.so ABI version {have} != {ABI_VERSION} expected by lib.py: the shared library is stale, "
                           "rebuild it (make -C texttoaudiogrounding_amd/csrc)")
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError(f"{LIB_PATH} does not export {name}: the shared library is stale, rebuild it "
                               "(make -C texttoaudiogrounding_amd/csrc)") from None
        fn.restype = res
        fn.argtypes = args
    # the binary attests the sources it was compiled from: a stale .so beside edited kernels is refused (a private build named by
    # TAG_HIP_LIB is compiled from the same sources with extra -D flags and passes; TAG_ALLOW_STALE_LIB=1 is the developer's override)
    built = lib.tag_build_id().decode()
    if os.path.isdir(os.path.join(_HERE, "csrc")) and built != csrc_sha256() and os.environ.get("TAG_ALLOW_STALE_LIB") != "1":
        raise RuntimeError(f"{LIB_PATH} was built from other kernel sources (build id {built[:12]} != csrc sha256 "
                           f"{csrc_sha256()[:12]}): rebuild it (make -C texttoaudiogrounding_amd/csrc)")
    # developer switches of the tool scripts: the launchers read no environment, the TAG_* variables are forwarded here
    for env, (opt, conv) in _ENV_OPTIONS.items():
        if env in os.environ:
            if lib.tag_set_option(opt.encode(), conv(os.environ[env])) != 0:
                raise RuntimeError(lib.tag_last_error().decode())
    _lib = lib
    return lib


#: environment variable of a tool script -> (option of tag_set_option, value conversion)
_ENV_OPTIONS = {
    "TAG_CONV_IMPL": ("conv_impl", int), "TAG_HALO_LDS_PAD": ("halo_lds_pad", int), "TAG_HALO_BN256": ("halo_bn256", int),
    "TAG_WGRAD_WGS": ("wgrad_wgs", int), "TAG_CONV_ROWS": ("conv_rows", int), "TAG_WGRAD_DMA": ("wgrad_dma", int),
    "TAG_X3_PRODUCTS": ("x3_products", int), "TAG_GEMM_BIG_MIN": ("gemm_big_min", int),
    "TAG_GRU_TILE": ("gru_tile4", lambda v: 0 if v == "16" else 1), "TAG_GRU_XCD": ("gru_xcd", int),
    "TAG_GRU_COOP": ("gru_coop", int), "TAG_MHA_MFMA": ("mha_mfma", int),
}


def build_id() -> str:
    """sha256 of the kernel sources the LOADED binary was compiled from (tag_build_id)."""
    return load().tag_build_id().decode()


def declared_symbols():
    return sorted(_SIGS)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on torch's current stream; raise on failure."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.tag_last_error().decode()}")


def query(name, *args):
    return getattr(load(), name)(*args)


def csrc_sha256() -> str:
    """sha256 over the kernel sources and their build flags (csrc/*.hip, csrc/*.h, csrc/Makefile; file names included, byte-sorted):
    stored in the _meta of the PMC traffic profiles (tools/pmc_traffic.py) so that bench.py can tell a profile collected on OTHER kernels from a current one."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + [os.path.join(src, "Makefile")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
