"""ctypes binding of libbidiff_hip.so (the C ABI declared in include/bidiff.h).

The library is built in-tree (``make -C csrc`` or ``__graft_entry__.build()``) and loaded from
``<package>/lib``. There is no CPU or PyTorch fallback: if the shared object is missing or a
call returns an error code, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbidiff_hip.so")

_lib = None
_weights_generation = 0


def weights_generation():
    return _weights_generation


def bump_weights_generation():
    """Called after parameters were updated in place by a kernel (torch's version counters do
    not see that): invalidates every cached bf16 weight copy."""
    global _weights_generation
    _weights_generation += 1


c_i64p = C.c_void_p
VP = C.c_void_p


class LayerWeights(C.Structure):
    _fields_ = [
        ("ln1_g", VP), ("ln1_b", VP),
        ("w_qkv", VP), ("b_qkv", VP),
        ("w_o", VP), ("b_o", VP),
        ("ln2_g", VP), ("ln2_b", VP),
        ("w_fc1", VP), ("b_fc1", VP),
        ("w_fc2", VP), ("b_fc2", VP),
        ("w_qkv_ln", VP), ("b_qkv_ln", VP),
        ("w_fc1_ln", VP), ("b_fc1_ln", VP),
        ("ffn_pack", VP),
        ("attn_pack", VP), ("b_q_attn", VP), ("b_o_fold", VP),
        ("cs_qkv_ln", VP), ("cs_fc1_ln", VP),
    ]


class WgradJob(C.Structure):
    _fields_ = [("dY", VP), ("X", VP), ("dW", VP * 3), ("db", VP * 3),
                ("ldy", C.c_int), ("ldx", C.c_int), ("n_seg", C.c_int), ("seg_rows", C.c_int), ("K", C.c_int)]


class BackboneDesc(C.Structure):
    _fields_ = [
        ("d", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("ffn", C.c_int),
        ("vocab", C.c_int), ("vocab_pad", C.c_int), ("max_pos", C.c_int), ("n_timesteps", C.c_int),
        ("ln_eps", C.c_float),
        ("tok_emb", VP), ("pos_emb", VP), ("time_emb", VP),
        ("lnf_g", VP), ("lnf_b", VP),
        ("w_out", VP),
        ("layers", C.POINTER(LayerWeights)),
        ("attn_width", C.c_int),
    ]


class Dropout(C.Structure):
    """bd_dropout: keep decisions are a function of (seed, step, site, row, column)."""
    _fields_ = [("p", C.c_float), ("step", C.c_uint), ("seed", C.c_ulonglong)]


class CastJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int),
                ("transpose", C.c_int), ("pad_axis", C.c_int), ("group", C.c_int), ("group_pad", C.c_int),
                ("scale", C.c_float)]


class StepArgs(C.Structure):
    _fields_ = [
        ("ids", VP), ("fixed", VP), ("attn", VP), ("keymask", VP), ("has_eos", VP),
        ("logits", VP), ("hidden", VP), ("wout", VP), ("u_sel", VP), ("u_tok", VP),
        ("ktable", VP), ("tok_flags", VP), ("contrib", VP), ("ring_slot", VP),
        ("counted_ids", VP), ("bond_ids", VP), ("non_special", VP),
        ("n_counted", C.c_int), ("n_bond", C.c_int), ("n_ns", C.c_int), ("R", C.c_int),
        ("B", C.c_int), ("L", C.c_int), ("V", C.c_int), ("d", C.c_int),
        ("mask_id", C.c_int), ("pad_id", C.c_int), ("bos_id", C.c_int), ("eos_id", C.c_int),
        ("open_id", C.c_int), ("close_id", C.c_int), ("boost_id", C.c_int), ("count_id", C.c_int),
        ("variant", C.c_int), ("use_constraints", C.c_int), ("t", C.c_int),
        ("temperature", C.c_float),
        ("seed", C.c_ulonglong),
        ("seq_offset", C.c_longlong),
        ("row_logits", VP), ("row_pos", VP), ("row_cap", C.c_int), ("rows_chosen", C.c_int),
    ]


class FixArgs(C.Structure):
    _fields_ = [
        ("ids", VP), ("logits", VP), ("attn", VP), ("keymask", VP), ("has_eos", VP),
        ("tok_flags", VP), ("ring_ids", VP),
        ("excl", C.c_int * 8),
        ("n_excl", C.c_int), ("R", C.c_int),
        ("B", C.c_int), ("L", C.c_int), ("V", C.c_int),
        ("mask_id", C.c_int), ("pad_id", C.c_int), ("bos_id", C.c_int), ("eos_id", C.c_int),
        ("open_id", C.c_int), ("close_id", C.c_int),
        ("count_id", C.c_int), ("target", C.c_int),
        ("do_fix", C.c_int), ("smiles_syntax", C.c_int), ("natural_eos", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/bidiff.h declares
SIGNATURES = {
    "bd_version": (C.c_char_p, []),
    "bd_last_error": (C.c_char_p, []),
    "bd_device_available": (C.c_int, []),
    "bd_dropout_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, C.c_uint, VP]),
    "bd_dropout_f32_to_bf16": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, C.c_uint, VP]),
    "bd_collate_batch": (C.c_int, [VP, VP, VP, C.c_longlong, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP]),
    "bd_pool_hidden": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_embed_drop": (C.c_int, [VP, VP, VP, VP, VP, VP] + [C.c_int] * 6 + [VP, C.c_uint, VP]),
    "bd_gemm_bf16_drop": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    VP, C.c_uint, VP]),
    "bd_attention_fwd_drop": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_uint, VP]),
    "bd_attention_bwd_drop": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP,
                                        C.c_uint, VP]),
    "bd_embed": (C.c_int, [VP, VP, VP, VP, VP, VP] + [C.c_int] * 6 + [VP]),
    "bd_layernorm": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_float, VP]),
    "bd_gemm_f32": (C.c_int, [VP, C.c_longlong, VP, C.c_longlong, VP, VP, VP, C.c_longlong, C.c_int, C.c_int, C.c_int,
                              C.c_int, VP]),
    "bd_attention_f32": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_bf16": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_bf16_aux": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_backbone_forward_rows_unfused_supported": (C.c_int, [C.POINTER(BackboneDesc), C.c_int, C.c_int]),
    "bd_backbone_rows_workspace_bytes": (C.c_longlong, [C.POINTER(BackboneDesc), C.c_int, C.c_int]),
    "bd_backbone_forward_rows_unfused": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, C.c_int, C.c_int, VP, C.c_int, VP,
                                                   C.c_longlong, VP, C.c_longlong, VP, VP]),
    "bd_gemm_bf16_tile256": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_deferred_ln_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bd_gemm_resid_stats_part_bytes": (C.c_longlong, [C.c_int, C.c_int]),
    "bd_gemm_bf16_resid_stats": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_float, VP]),
    "bd_gemm_bf16_deferred_ln": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_masked_ce_fwd": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, VP]),
    "bd_masked_ce_bwd": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, VP]),
    "bd_layernorm_bwd": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, VP]),
    "bd_layernorm_bwd_cast": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, VP, VP, C.c_uint, VP]),
    "bd_layernorm_bwd_parts": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_uint, VP]),
    "bd_layernorm_parts_finish": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_decode_ids": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, C.c_int, VP]),
    "bd_gemm_wgrad_bf16": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_wgrad_group": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_wgrad_segments": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_gemm_bf16_batched": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_float, VP]),
    "bd_muon_workspace_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "bd_muon_step": (C.c_int, [VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                               C.c_int, C.c_int, VP, C.c_float, VP]),
    "bd_colsum_bf16": (C.c_int, [VP, C.c_int, VP, C.c_int, C.c_int, VP]),
    "bd_attention_bwd": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_attention_bwd_static_capacity": (C.c_longlong, []),
    "bd_attention_bwd_static": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_build_rowmap": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP]),
    "bd_pack_targets": (C.c_int, [VP, VP, VP, VP, C.c_int, VP, VP, VP]),
    "bd_masked_ce_finish": (C.c_int, [VP, VP, C.c_int, VP]),
    "bd_masked_ce_fwd_parts": (C.c_int, [VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_embed_rows": (C.c_int, [VP, VP, VP, VP, VP, VP, VP] + [C.c_int] * 6 + [VP, C.c_uint, VP]),
    "bd_attention_fwd_varlen": (C.c_int, [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_uint, VP]),
    "bd_attention_bwd_varlen": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP,
                                          C.c_uint, VP]),
    "bd_embed_bwd_varlen": (C.c_int, [VP, VP, VP, VP, VP, VP, VP] + [C.c_int] * 7 + [VP]),
    "bd_embed_bwd": (C.c_int, [VP, VP, VP, VP, VP, VP] + [C.c_int] * 7 + [VP]),
    "bd_sumsq_f32": (C.c_int, [VP, C.c_longlong, VP, VP]),
    "bd_adamw_step": (C.c_int, [VP, VP, VP, VP, C.c_longlong] + [C.c_float] * 7 + [VP, C.c_float, VP]),
    "bd_adamw_step_guarded": (C.c_int, [VP, VP, VP, VP, C.c_longlong, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float,
                                        VP, VP, C.c_float, VP]),
    "bd_step_guard_update": (C.c_int, [VP, VP, VP]),
    "bd_muon_step_guarded": (C.c_int, [VP, VP, VP, VP, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_int, C.c_int, VP, C.c_float, VP]),
    "bd_multi_cast_bf16": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_multi_unpad_add_f32": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP]),
    "bd_cast_f32_to_bf16": (C.c_int, [VP, VP, C.c_longlong, VP]),
    "bd_gemm_panel_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bd_gemm_panel": (C.c_int, [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, VP]),
    "bd_ffn_fused_supported": (C.c_int, [C.c_int, C.c_int]),
    "bd_ffn_pack_bytes": (C.c_long, [C.c_int, C.c_int]),
    "bd_ffn_pack_weights": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP]),
    "bd_ffn_fused": (C.c_int, [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_float, VP]),
    "bd_attn_block_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bd_attn_pack_bytes": (C.c_long, [C.c_int]),
    "bd_attn_pack_bytes_heads": (C.c_long, [C.c_int, C.c_int]),
    "bd_attn_block_width": (C.c_int, [C.c_int, C.c_int]),
    "bd_attn_pack_weights": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP]),
    "bd_attn_block_fused": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, VP]),
    "bd_attention_fwd": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "bd_mask_to_bytes": (C.c_int, [VP, VP, C.c_longlong, VP]),
    "bd_backbone_workspace_bytes": (C.c_longlong, [C.POINTER(BackboneDesc), C.c_int, C.c_int]),
    "bd_backbone_forward": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, C.c_int, C.c_int, VP,
                                      C.c_longlong, VP, VP, VP]),
    "bd_backbone_mega_supported": (C.c_int, [C.POINTER(BackboneDesc), C.c_int]),
    "bd_backbone_mega": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, C.c_int, C.c_int, VP, VP, VP]),
    "bd_backbone_mega_ragged_supported": (C.c_int, [C.POINTER(BackboneDesc), C.c_int]),
    "bd_backbone_mega_ragged": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP]),
    "bd_backbone_forward_ragged": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, VP, C.c_int, C.c_int, C.c_int,
                                             VP, C.c_longlong, VP, VP, VP]),
    "bd_backbone_forward_rows_supported": (C.c_int, [C.POINTER(BackboneDesc), C.c_int]),
    "bd_backbone_forward_rows": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, VP, C.c_int, C.c_int, C.c_int,
                                           VP, VP, C.c_int, VP, C.c_longlong, VP, VP]),
    "bd_backbone_forward_layers": (C.c_int, [C.POINTER(BackboneDesc), VP, VP, VP, C.c_int, C.c_int, VP,
                                             C.c_longlong, C.c_int, VP, VP]),
    "bd_corrupt": (C.c_int, [VP, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_ulonglong, C.c_longlong, VP]),
    "bd_sample_step": (C.c_int, [C.POINTER(StepArgs), VP]),
    "bd_sample_choose_rows": (C.c_int, [C.POINTER(StepArgs), VP, VP, C.c_int, VP]),
    "bd_sample_fixup": (C.c_int, [C.POINTER(FixArgs), VP]),
    "bd_host_gumbel": (None, [VP, VP, C.c_longlong]),
    "bd_host_philox": (None, [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]),
    "bd_host_dropout_factors": (C.c_int, [VP, C.c_uint, C.c_longlong, C.c_longlong, VP]),
    "bd_host_uniform_select": (C.c_float, [C.c_ulonglong, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bd_host_uniform_token": (C.c_float, [C.c_ulonglong, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bd_clock_stamps_set": (C.c_int, [VP, C.c_int]),
    "bd_prof_enable": (C.c_int, [C.c_int]),
    "bd_prof_read": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "bd_prof_read_kind": (C.c_int, [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
}


def lib():
    """Load (once) and return the shared library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"bidiff: {LIB_PATH} not found - build it with `make -C "
                f"{os.path.join(os.path.dirname(_HERE), 'csrc')}` (there is no fallback path)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().bd_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"bidiff {what} failed ({rc}): {msg}")


def ptr(t):
    """Device (or host) address of a tensor, or None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"bidiff {what}: tensors must live on a ROCm device (got {t.device}); "
            "this package has no CPU execution path")
