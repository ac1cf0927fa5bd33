"""ctypes binding of libtrimodal_hip.so (the C ABI declared in include/trimodal_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TG_LIB_PATH: another build of the same library -- same-box A/B timing against an older build, tools/ab_bench.sh; the lab library of the
# ablation tools.  Unset in every test and in bench.py's default run.)
LIB_PATH = os.environ.get("TG_LIB_PATH") or os.path.join(_HERE, "libtrimodal_hip.so")


class Window(C.Structure):
    """struct tg_window (include/trimodal_hip.h)."""
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("row_stride", C.c_int64), ("rows_in", C.c_int32),
                ("rows_out", C.c_int32), ("row_step", C.c_int32), ("shift", C.c_int32), ("dil", C.c_int32),
                ("cw", C.c_int32), ("K", C.c_int32)]


P, I32, I64, F32, U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
WP = C.POINTER(Window)


class NtProblem(C.Structure):
    """struct tg_gemm_nt_problem"""
    _fields_ = [("A", Window), ("Bw", P), ("ldb", I64), ("b_seg_k", I32), ("reserved", I32), ("b_seg_stride", I64), ("bias", P), ("C", P),
                ("c_batch_stride", I64), ("c_row_stride", I64), ("c_rows_out", I32), ("M", I32), ("N", I32), ("act_slope", F32),
                ("accumulate", I32), ("out_scale", P), ("b_planes", P), ("b_plane_stride", I64), ("b_rows", I32), ("b_row0", I32),
                ("gate", P), ("res", P), ("C2", P), ("res_slope", F32), ("drop_site", U32), ("drop_state", P), ("drop_index0", I64),
                ("drop_p", F32), ("reserved4", I32), ("b_planes_kind", I32), ("reserved5", I32), ("b_inv_scale", P), ("a_row_scale", P), ("a_rowmax", P), ("c_rowmax", P), ("c2_rowmax", P), ("a_rowmax_rows", I32), ("reserved6", I32)]


class TnProblem(C.Structure):
    """struct tg_gemm_tn_problem"""
    _fields_ = [("dY", P), ("ldy", I64), ("A", Window), ("dW", P), ("ldw", I64), ("M", I32), ("N", I32), ("out_kw", I32), ("reserved", I32),
                ("dbias", P), ("ws", P), ("ws_floats", I64), ("y_colmax", P), ("a_colmax", P)]


class AeStepArgs(C.Structure):
    """struct tg_ae_step_args"""
    _fields_ = [("x", P), ("params", P), ("grads", P), ("off", I32 * 44), ("running_mean", P * 8), ("running_var", P * 8),
                ("num_batches_tracked", P * 8), ("ws", P), ("ws_bytes", I64), ("loss", P), ("recon", P), ("feat", P), ("step", P), ("B", I32),
                ("bn_eps", F32), ("momentum", F32), ("last_phase", I32), ("adam_m", P), ("adam_v", P), ("lr", F32), ("beta1", F32), ("beta2", F32), ("adam_eps", F32)]


MAX_GROUP = 8

# name -> argtypes (all return int); mirrors include/trimodal_hip.h one to one
SIGNATURES = {
    "tg_gemm_nt": [WP, P, I64, P, P, I64, I64, I32, I32, I32, F32, I32, P],
    "tg_gemm_nt_group": [C.POINTER(NtProblem), I32, P],
    "tg_gemm_tn_group": [C.POINTER(TnProblem), I32, P],
    "tg_ae_train_step": [C.POINTER(AeStepArgs), P],
    "tg_split3_planes": [P, I64, I32, I32, P, I32, I64, P],
    "tg_split2h_planes": [P, I64, I32, I32, P, I32, I64, P, P],
    "tg_split2h_planes_tcat": [P, P, I32, I32, P, I32, I64, P, P],
    "tg_win_row_absmax": [WP, I32, P, P],
    "tg_h2_row_scales": [WP, I32, P, P, P],
    "tg_absmax_rows_cols": [P, I64, I32, I32, I32, P, P, P],
    "tg_gemm_tn": [P, I64, WP, P, I64, I32, I32, I32, P, P, I64, P],
    "tg_colsum": [P, I64, I32, I32, P, I32, P],
    "tg_gru_forward": [P, I64, P, P, P, P, P, P, I64, I32, I32, I32, P],
    "tg_gru_backward": [P, P, P, I64, P, P, P, P, I64, P, I32, I32, I32, P],
    "tg_gru_h64_forward": [P, I64, P, P, P, P, P, P, I64, P, P, I32, I32, P],
    "tg_gru_h64_backward": [P, P, P, P, I64, P, P, P, P, I64, I32, I32, P],
    "tg_gru_forward_cluster": [P, I64, P, P, P, P, P, P, I64, P, P, P, I64, I32, I32, I32, P],
    "tg_gru_forward_cluster_rows": [P, I64, P, P, P, P, P, P, I64, P, P, P, I64, I32, I32, I32, I32, I32, P],
    "tg_gru_forward_vec": [P, I64, P, P, P, P, P, P, I64, I32, I32, I32, P],
    "tg_gru_backward_cluster": [P, P, P, P, I64, P, P, P, P, I64, P, I64, I32, I32, I32, P],
    "tg_gru_backward_cluster_stats": [P, P, P, P, I64, P, P, P, P, I64, P, I64, I32, I32, I32, P, I64, P, P, P],
    "tg_bn_train_stats": [P, I32, I32, I32, P, P, P, P, P, P, F32, F32, I32, P],
    "tg_bn_eval_stats": [P, P, I32, F32, P, P, P],
    "tg_bn_train_fused": [P, P, I32, I32, I32, P, P, P, P, P, P, P, F32, F32, F32, I32, P],
    "tg_bn_apply": [P, P, I32, I32, I32, P, P, P, P, F32, P],
    "tg_bn2_train": [P, P, I32, I32, I32, P, I64, P, P, P, P, P, P, P, F32, F32, F32, I32, P],
    "tg_bn2_backward": [P, P, P, I32, I32, I32, P, P, P, P, F32, P, I64, P, P, P],
    "tg_bn_backward": [P, P, P, I32, I32, P, P, P, P, F32, P, P, P, P],
    "tg_wav_front_stats": [P, I64, I32, I32, P, P, I32, I32, I32, P, I64, P, P, P, P, P, P, F32, F32, I32, P],
    "tg_wav_front_apply": [P, I64, I32, I32, P, P, I32, I32, I32, P, P, P, P, F32, P, P, P],
    "tg_wav_front_backward": [P, P, P, I64, I32, I32, P, P, I32, I32, I32, P, P, P, P, F32, P, I64, P, P, P, P, P],
    "tg_wav_front_backward_fused": [P, I32, P, P, P, I64, I32, I32, P, P, I32, I32, I32, P, P, P, P, F32, P, I64, P, P, P, P, P],
    "tg_speaker_fwd": [P, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, I32, P, I64, I32, P, U32, P],
    "tg_speaker_bwd": [P, P, P, P, P, P, P, P, I32, P, P, P, P, P, P, P, P, P, P, I32, P],
    "tg_wav_conv2_wgrad": [P, P, I32, I32, I32, P, I64, P, P, P],
    "tg_out_mlp_compose": [P, P, P, P, I32, I32, I32, I32, P, P, P, P],
    "tg_out_mlp_param_grads": [P, P, P, P, P, I32, I32, I32, I32, P, P, P, P, P],
    "tg_zero": [P, I64, P],
    "tg_permute3_batch": [P, I32, I32, P],
    "tg_add_relu": [P, P, P, I64, P],
    "tg_act_mask_bwd": [P, P, P, F32, P, I64, P],
    "tg_act_mask_bwd2": [P, P, P, P, F32, P, P, I64, P],
    "tg_embed_gather_drop": [P, P, P, I32, I32, I32, F32, P, U32, P],
    "tg_d_preconv_bwd": [P] * 26 + [I32, P, I64, I32, I32, P],
    "tg_d_preconv_fwd": [P] * 27 + [I64, I32, I32, F32, F32, P],
    "tg_iter_head": [P, P, P, P, P, P, I64, I32, I32, I32, I32, I32, P, P, P, P, I32, P, U32, P, P, P],
    "tg_act_mask_bwd_drop": [P, P, F32, P, U32, I64, F32, P, I64, P],
    "tg_act_mask_bwd2_drop": [P, P, P, F32, P, U32, I64, F32, P, P, I64, P],
    "tg_mul": [P, P, P, I64, P],
    "tg_axpy": [P, P, F32, I32, I64, P],
    "tg_copy2d": [P, I64, P, I64, I32, I32, I32, P],
    "tg_repeat_rows": [P, I64, P, I64, I32, I32, I32, P],
    "tg_sum_rows": [P, I64, P, I64, I32, I32, I32, I32, P],
    "tg_add_halves": [P, P, I32, I32, P],
    "tg_sum_parts": [P, I64, I32, P, I64, P],
    "tg_narrow8_pair": [P, P, P, P, P, I32, I32, P],
    "tg_dup_halves": [P, P, I32, I32, P],
    "tg_make_pre_seq": [P, P, I32, I32, I32, I32, P],
    "tg_embed_gather": [P, P, P, I32, I32, I32, P],
    "tg_assemble_batch": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, P, P, P, P, P],
    "tg_embed_scatter_add": [P, P, P, I32, I32, I32, P],
    "tg_permute3": [P, P, I32, I32, I32, I32, I32, I32, P],
    "tg_conv_dgrad_pack": [P, P, I32, I32, I32, I32, P],
    "tg_weight_norm_fwd": [P, P, P, I32, I32, I32, P],
    "tg_weight_norm_fwd_batch": [I32, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), I32, I32, I32, P],
    "tg_weight_norm_bwd": [P, P, P, P, P, I32, I32, I32, P],
    "tg_weight_norm_bwd_batch": [I32, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), I32, I32, I32, P],
    "tg_rng_advance": [P, P],
    "tg_iter_begin": [P, P, P, P, P],
    "tg_dropout_mask": [P, I64, F32, P, U32, P],
    "tg_dropout_apply": [P, P, P, I64, F32, P, U32, P],
    "tg_normal": [P, I64, P, U32, P],
    "tg_randperm": [P, I32, P, U32, P],
    "tg_gather_i64": [P, P, P, I32, P],
    "tg_reparam_fwd": [P, P, P, P, I64, P],
    "tg_reparam_bwd": [P, P, P, P, P, I64, P],
    "tg_gan_d_loss": [P, P, I32, P, P, P, P],
    "tg_gan_g_loss": [P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, F32, F32, I32, P, P, P, P, P, P, P],
    "tg_d_head_fwd": [P, P, P, P, P, P, P, P, I32, I32, I32, P],
    "tg_d_head_bwd": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, P],
    "tg_d_head_step": [P] * 17 + [I32, I32, F32, F32, I32, I32, P],
    "tg_l1_mean": [P, P, I64, P, P],
    "tg_sigmoid": [P, P, I64, P],
    "tg_sigmoid_bwd": [P, P, P, I64, P],
    "tg_window_blend": [P, P, I32, I32, I32, I32, P],
    "tg_pose_metrics": [P, P, P, I32, I32, I32, P, P],
    "tg_ae_loss": [P, P, I32, I32, I32, P, P, P],
    "tg_counter_inc": [P, P],
    "tg_adam_step": [P, P, P, P, I64, F32, F32, F32, F32, P, P],
}

ABI_VERSION = 8
_lib = None


def load():
    """Load the shared library once; raise with a build hint if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gesture-generation-from-trimodal-context_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.tg_version.restype = C.c_int
    lib.tg_last_error.restype = C.c_char_p
    lib.tg_gemm_tn_ws_floats.restype = C.c_int64
    lib.tg_gemm_tn_ws_floats.argtypes = [I32, I32, I32]
    lib.tg_gemm_nt_family.restype = C.c_int32
    lib.tg_gemm_nt_family.argtypes = [C.POINTER(NtProblem)]
    lib.tg_gemm_nt_ext_supported.restype = C.c_int32
    lib.tg_gemm_nt_ext_supported.argtypes = [C.POINTER(NtProblem)]
    lib.tg_gemm_nt_kernel_plan.restype = C.c_int32
    lib.tg_gemm_nt_kernel_plan.argtypes = [C.POINTER(NtProblem), I32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.tg_gemm_tn_kernel_plan.restype = C.c_int32
    lib.tg_gemm_tn_kernel_plan.argtypes = [C.POINTER(TnProblem), I32]
    lib.tg_bn_fused_supported.restype = C.c_int32
    lib.tg_bn_fused_supported.argtypes = [I32, I32, I32]
    lib.tg_set_math_mode.restype = C.c_int
    lib.tg_set_math_mode.argtypes = [I32]
    lib.tg_get_math_mode.restype = C.c_int
    lib.tg_set_deterministic.restype = C.c_int
    lib.tg_set_deterministic.argtypes = [I32]
    lib.tg_get_deterministic.restype = C.c_int
    lib.tg_set_tn_workgroup_cap.restype = C.c_int
    lib.tg_set_tn_workgroup_cap.argtypes = [I32]
    lib.tg_get_tn_workgroup_cap.restype = C.c_int
    lib.tg_set_nt_mover_waves.restype = C.c_int
    lib.tg_set_nt_mover_waves.argtypes = [I32]
    lib.tg_gru_cluster_fused_dropout.restype = C.c_int32
    lib.tg_gru_cluster_supported.restype = C.c_int32
    lib.tg_gru_cluster_supported.argtypes = [I32, I32]
    lib.tg_gru_cluster_ws_bytes.restype = C.c_int64
    lib.tg_gru_cluster_ws_bytes.argtypes = [I32, I32]
    lib.tg_gru_vec_supported.restype = C.c_int32
    lib.tg_gru_vec_supported.argtypes = [I32, I32]
    lib.tg_gru_vec_ws_bytes.restype = C.c_int64
    lib.tg_gru_vec_ws_bytes.argtypes = [I32]
    lib.tg_gru_vec_ws_header_bytes.restype = C.c_int32
    lib.tg_gru_vec_ws_header_bytes.argtypes = []
    lib.tg_gru_cluster_bwd_supported.restype = C.c_int32
    lib.tg_gru_cluster_bwd_supported.argtypes = [I32, I32]
    lib.tg_gru_cluster_bwd_ws_bytes.restype = C.c_int64
    lib.tg_gru_cluster_bwd_ws_bytes.argtypes = [I32, I32]
    lib.tg_speaker_bwd_max_rows.restype = C.c_int32
    lib.tg_bn2_supported.restype = C.c_int32
    lib.tg_bn2_supported.argtypes = [I32, I32]
    lib.tg_bn2_ws_doubles.restype = C.c_int64
    lib.tg_bn2_ws_doubles.argtypes = [I32, I32, I32]
    lib.tg_d_preconv_fwd_supported.restype = C.c_int32
    lib.tg_d_preconv_fwd_supported.argtypes = [I32, I32]
    lib.tg_d_preconv_ws_bytes.restype = C.c_int64
    lib.tg_d_preconv_ws_bytes.argtypes = [I32]
    lib.tg_ae_step_supported.restype = C.c_int32
    lib.tg_ae_step_supported.argtypes = [I32]
    lib.tg_ae_step_ws_bytes.restype = C.c_int64
    lib.tg_ae_step_ws_bytes.argtypes = [I32]
    for q, at in (("tg_wav_conv2_wgrad_ws_floats", []), ("tg_wav_front_ws_doubles", []), ("tg_wav_front_fstat_doubles", []), ("tg_wav_front_gate_words", [I32, I32])):
        getattr(lib, q).restype = C.c_int64
        getattr(lib, q).argtypes = at
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: intended
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.tg_version() != ABI_VERSION:
        raise RuntimeError(f"libtrimodal_hip.so ABI {lib.tg_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.tg_last_error().decode()}")
