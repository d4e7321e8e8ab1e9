"""ctypes binding of libsplice_hip.so (the C ABI of include/splice_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails,
a RuntimeError is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C splice_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsplice_hip.so")

_lib = None


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int), ("resid_mod", C.c_int),
        ("out_f32", C.c_void_p), ("ldo", C.c_int),
        ("out_bf", C.c_void_p), ("ldbf", C.c_int),
        ("out_bf_t", C.c_void_p), ("ldt", C.c_int),
        ("out_pre", C.c_void_p), ("ldp", C.c_int), ("pre_row_lo", C.c_int),
        ("aux", C.c_void_p), ("ldaux", C.c_int),
        ("out_f32_cols", C.c_void_p), ("ld_cols", C.c_int), ("col_lo", C.c_int), ("col_hi", C.c_int),
        ("alpha", C.c_float), ("ksplit", C.c_int), ("slab_stride", C.c_longlong),
        ("rd_other", C.c_void_p), ("ld_rd", C.c_int), ("rd_rows", C.c_int), ("rowdot", C.c_void_p),
        ("row_scale", C.c_void_p), ("col_scale", C.c_void_p),
        ("out_f8", C.c_void_p), ("ld8", C.c_int),
        ("out_f8_t", C.c_void_p), ("ldt8", C.c_int),
    ]


class GenArch(C.Structure):
    _fields_ = [("n_scales", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
                ("down", C.c_int * 6), ("up", C.c_int * 6), ("skip", C.c_int * 6),
                ("filter_down", C.c_int * 6), ("filter_up", C.c_int * 6), ("filter_skip", C.c_int), ("reflect", C.c_int)]


class StepConfig(C.Structure):
    _fields_ = [
        ("crop_h", C.c_int), ("crop_w", C.c_int), ("vit_h", C.c_int), ("vit_w", C.c_int),
        ("ent_h", C.c_int), ("ent_w", C.c_int), ("ent_vit_h", C.c_int), ("ent_vit_w", C.c_int),
        ("lambda_global_cls", C.c_float), ("lambda_global_ssim", C.c_float), ("lambda_global_identity", C.c_float),
        ("lambda_entire_cls", C.c_float), ("lambda_entire_ssim", C.c_float),
        ("entire_every", C.c_int), ("cls_warmup", C.c_int),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("pairs", C.c_int), ("arena_stride", C.c_longlong), ("fp8_selfsim", C.c_int), ("top_cls_only", C.c_int), ("n_crops", C.c_int), ("n_crops_b", C.c_int),
    ]


EPI_BIAS, EPI_RESID, EPI_OUT_F32, EPI_OUT_BF, EPI_OUT_T = 1, 2, 4, 8, 16
EPI_GELU, EPI_GELU_GRAD, EPI_COLS_F32, EPI_ALPHA, EPI_ROWDOT, EPI_SCALE_RC, EPI_OUT_F8, EPI_OUT_F8T = 32, 64, 128, 256, 512, 1024, 2048, 4096

_vp, _i, _f, _sz, _u = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint

_SIGNATURES = {
    "splice_version": ([], C.c_int),
    "splice_dev_switches": ([], C.c_int),
    "splice_last_error": ([], C.c_char_p),
    "splice_gemm_nt_bf16": ([_u, _vp, _i, _vp, _i, _i, _i, _i, C.POINTER(GemmEpilogue), _vp], _i),
    "splice_gemm_nt_fp8": ([_u, _vp, _i, _vp, _i, _i, _i, _i, C.POINTER(GemmEpilogue), _vp], _i),
    "splice_quantize_rows_fp8": ([_vp, _i, _vp, _i, _vp, _i, _i, _vp], _i),
    "splice_gemm_splitk_slabs": ([_i, _i], _i),
    "splice_gemm_force_tile": ([_i], _i),
    "splice_attention_variant": ([_i], _i),
    "splice_attention_qfold": ([_i], _i),
    "splice_attention_bwd_variant": ([_i], _i),
    "splice_layernorm_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp], _i),
    "splice_layernorm_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "splice_attention_fwd": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "splice_attention_fwd_fp8": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "splice_attention_bwd": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "splice_attention_probs": ([_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "splice_augment_structure": ([_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_float), _f, _vp], _i),
    "splice_keys_selfsim_ws_bytes": ([_i, _i], _sz),
    "splice_keys_selfsim_fwd": ([_vp, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "splice_keys_selfsim_bwd": ([_vp, _vp, _i, _i, _f, _vp, _i, _i, _vp, _vp], _i),
    "splice_mse": ([_vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp], _i),
    "splice_patchify": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "splice_unpatchify": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "splice_cast_f32_bf16": ([_vp, _vp, _sz, _vp], _i),
    "splice_cast_bf16_f32": ([_vp, _vp, _sz, _vp], _i),
    "splice_transpose_f32_bf16": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "splice_resize_bilinear_fwd": ([_vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "splice_resize_bilinear_bwd": ([_vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    # ViT engine
    "splice_vit_create": ([_i, _i, _i, _i, C.POINTER(_vp)], _i),
    "splice_vit_destroy": ([_vp], None),
    "splice_vit_set_param": ([_vp, C.c_char_p, _vp, C.c_longlong, _vp], _i),
    "splice_vit_params_complete": ([_vp], _i),
    "splice_vit_enable_fp8": ([_vp, _vp], _i),
    "splice_vit_ctx_create": ([_vp, _i, _i, _i, _vp, _i, _vp, C.POINTER(_vp)], _i),
    "splice_vit_ctx_destroy": ([_vp], None),
    "splice_vit_ctx_info": ([_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _i),
    "splice_vit_ctx_set_top_cls_only": ([_vp, _i], _i),
    "splice_vit_ctx_set_fp8": ([_vp, _i], _i),
    "splice_vit_forward": ([_vp, _vp, _i, _vp], _i),
    "splice_vit_forward_ex": ([_vp, _vp, _i, _i, _vp], _i),
    "splice_vit_forward_passes": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "splice_vit_get_tensor": ([_vp, _i, _i, C.POINTER(_vp)], _i),
    "splice_vit_qscale": ([_vp], C.c_float),
    "splice_vit_read_tensor": ([_vp, _i, _i, _vp, _sz, _vp], _i),
    "splice_vit_backward": ([_vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp, _i, _vp], _i),
    # generator engine
    "splice_gen_create": ([C.POINTER(_vp)], _i),
    "splice_gen_create_arch": ([C.POINTER(GenArch), C.POINTER(_vp)], _i),
    "splice_gen_destroy": ([_vp], None),
    "splice_gen_param_count": ([_vp], C.c_longlong),
    "splice_gen_num_tensors": ([_vp], _i),
    "splice_gen_tensor_info": ([_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], _i),
    "splice_gen_plan_create": ([_vp, _i, _i, _i, _i, C.POINTER(_vp)], _i),
    "splice_gen_plan_destroy": ([_vp], None),
    "splice_gen_forward": ([_vp, _vp, _vp, _vp, _vp], _i),
    "splice_gen_forward_borrowed": ([_vp, _vp, _vp, _vp, _vp], _i),
    "splice_gen_backward": ([_vp, _vp, _vp, _vp, _i, _vp], _i),
    "splice_adam_step": ([_vp, _vp, _vp, _vp, C.c_longlong, _f, _f, _f, _f, _i, _i, _vp], _i),
    "splice_prof_begin": ([_i], _i),
    "splice_prof_end": ([C.POINTER(_f), C.POINTER(_i)], _i),
    "splice_prof_end_ex": ([C.POINTER(_f), C.POINTER(_i), C.POINTER(_i)], _i),
    "splice_prof_end_detail": ([C.POINTER(_f), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i], _i),
    "splice_prof_active": ([], _i),
    "splice_step_use_graph": ([_vp, _i], _i),
    "splice_step_graph_stats": ([_vp, C.POINTER(C.c_longlong)], _i),
    "splice_step_use_overlap": ([_vp, _i], _i),
    "splice_vit_ctx_dims": ([_vp] + [C.POINTER(_i)] * 7, _i),
    "splice_gen_plan_dims": ([_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_longlong)], _i),
    # fused step
    "splice_step_create": ([C.POINTER(StepConfig), _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)], _i),
    "splice_step_destroy": ([_vp], None),
    "splice_step_run": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "splice_step_output": ([_vp, _i, C.POINTER(_vp)], _i),
    "splice_step_set_crops": ([_vp, _i, _i, _i, _i], _i),
    "splice_step_set_running_stats": ([_vp, _vp, C.c_longlong], _i),
    "splice_step_set_mode": ([_vp, _i, _i], _i),
    "splice_step_set_phases": ([_vp, _i, _vp], _i),
    "splice_gen_buffer_count": ([_vp], C.c_longlong),
    "splice_gen_num_buffers": ([_vp], _i),
    "splice_gen_buffer_info": ([_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], _i),
    "splice_gen_running_stats_update": ([C.POINTER(_vp), _i, _vp, C.c_longlong, _f, _vp], _i),
    "splice_gen_plan_resize": ([_vp, _i, _i], _i),
    "splice_gen_plan_set_arena_stride": ([_vp, C.c_longlong], _i),
    "splice_gen_plan_set_batch_stats": ([_vp, _i], _i),
}


def exported_symbols():
    """Names include/splice_hip.h declares (checked against the .so by the CPU tests)."""
    return sorted(_SIGNATURES)


def register(name, argtypes, restype):
    _SIGNATURES[name] = (argtypes, restype)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built "
                "(run __graft_entry__.build() or `make -C splice_amd/csrc`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().splice_last_error().decode(errors="replace")
        raise RuntimeError(f"libsplice_hip {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
