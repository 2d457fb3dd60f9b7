"""ctypes binding of libeodiff.so (C ABI declared in include/eodiff.h).

The library is the ONLY compute path of this package: there is no eager / CPU fallback.  If the
shared object is missing or a call fails, an exception is raised -- nothing is silently rerouted.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EOD_LIBRARY: another build of the same library, for same-box A/B measurements of a kernel change -- tools only)
LIB_PATH = os.environ.get("EOD_LIBRARY") or os.path.join(_HERE, "lib", "libeodiff.so")

EOD_F32, EOD_F16 = 0, 1
ATTN_OUT_PRESPLIT, ATTN_IN_PRESPLIT, ATTN_EXACT_F32 = 1, 2, 4  # eod_attention_fwd_nat flags
(OP_CONV, OP_GEMM, OP_GN_PARTIAL, OP_GN_FINALIZE, OP_GN_APPLY, OP_SOFTMAX, OP_TEMB, OP_TO_NHWC, OP_TO_NCHW,
 OP_POOL, OP_ATTN, OP_TRANSPOSE, OP_ATTN_NAT, OP_DROPOUT, OP_ACT_BOUND, OP_BOUND_AFFINE) = range(1, 17)

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ConvDesc(C.Structure):
    _fields_ = [("x", vp), ("x2", vp), ("w", vp), ("bias", vp), ("cbias", vp), ("res", vp), ("y", vp),
                ("cbias_stride", i64), ("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("C0", i32), ("C1", i32),
                ("Cout", i32), ("ksize", i32), ("stride", i32), ("pad", i32), ("upsample", i32), ("pad_tl", i32),
                ("Ho", i32), ("Wo", i32), ("out_nchw_f32", i32), ("alpha", f32), ("stats", vp), ("stats_slots", i32),
                ("gn_silu", i32), ("gn_scale_shift", vp), ("workspace", vp), ("workspace_bytes", i64),
                ("w_tapmajor", i32), ("w_split", i32), ("w_scale", vp), ("a_bound", vp),
                ("skip_x", vp), ("skip_x2", vp), ("skip_w", vp), ("skip_C0", i32), ("skip_C1", i32), ("skip_bound", vp), ("x_presplit", i32), ("y_presplit_bound", vp)]


class GemmDesc(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("bias", vp), ("res", vp), ("c", vp), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("sa0", i64), ("sa1", i64), ("sb0", i64), ("sb1", i64), ("sc0", i64), ("sc1", i64), ("dtype", i32),
                ("M", i32), ("N", i32), ("K", i32), ("nb0", i32), ("nb1", i32), ("bias_mode", i32), ("c_f32", i32),
                ("alpha", f32), ("x3", i32), ("a_bound", vp), ("b_bound", vp)]


class TembDesc(C.Structure):
    _fields_ = [("t", vp), ("freqs", vp), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("label_emb", vp), ("y", vp),
                ("wcat", vp), ("bcat", vp), ("h1", vp), ("emb", vp), ("out", vp), ("N", i32), ("D", i32), ("E", i32),
                ("J", i32), ("t_f32", i32), ("_pad", i32)]


class AttnDesc(C.Structure):
    _fields_ = [("qk", vp), ("vT", vp), ("out", vp), ("lse", vp), ("ld_qk", i64), ("ldt", i64), ("dtype", i32), ("N", i32), ("T", i32),
                ("C", i32), ("heads", i32), ("d", i32), ("dpad", i32), ("k_off", i32)]


class PackJob(C.Structure):
    _fields_ = [("w", vp), ("dst", vp), ("kind", i32), ("Cout", i32), ("Cin", i32), ("taps", i32), ("ci0", i32), ("nci", i32),
                ("cpad", i32), ("_pad", i32)]


PACK_CHUNK = 4096
WSCALE_ROWS = 4  # float slots of a split-fp16 weight scale buffer in front of its per-row exponents (csrc/misc.hip: EOD_WSCALE_ROWS)


class SmallDesc(C.Structure):
    _fields_ = [("p", vp * 8), ("l", i64 * 4), ("i", i32 * 10), ("f", f32 * 2)]


class _OpU(C.Union):
    _fields_ = [("conv", ConvDesc), ("gemm", GemmDesc), ("temb", TembDesc), ("attn", AttnDesc), ("small", SmallDesc)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("_pad", i32), ("u", _OpU)]


# every symbol include/eodiff.h declares: (name, restype, argtypes)
SYMBOLS = {
    "eod_last_error": (C.c_char_p, []),
    "eod_version": (i32, []),
    "eod_set_option": (i32, [C.c_char_p, i32]),
    "eod_get_option": (i32, [C.c_char_p]),
    "eod_struct_size": (i32, [i32]),
    "eod_conv2d_igemm": (i32, [C.POINTER(ConvDesc), vp]),
    "eod_gemm_nt": (i32, [C.POINTER(GemmDesc), vp]),
    "eod_pack_conv_weight": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "eod_pack_conv_weight_tapmajor": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "eod_pack_conv_weight_tapmajor_split": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "eod_conv_tapmajor_ldk": (i32, [i32, i32]),
    # training path (csrc/train.hip)
    "eod_pack_conv_weight_dgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eod_pack_jobs": (i32, [vp, vp, vp, i32, i32, vp]),
    "eod_transpose_gather": (i32, [vp, i32, i32, i32, i32, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eod_rowsum_segments": (i32, [vp, i32, i32, i64, i32, i64, f32, vp, i64, vp]),
    "eod_colsum": (i32, [vp, i32, i32, vp, vp]),
    "eod_channel_sums_finish": (i32, [vp, i32, i32, i32, i32, f32, vp, vp, i64, vp, vp]),
    "eod_wgrad_reduce": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]),
    "eod_conv3x3_wgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "eod_wgrad_up4_map": (i32, [vp, i32, i32, vp, vp]),
    "eod_conv1x1_wgrad": (i32, [vp, vp, i32, i64, i32, i32, i32, vp, i32, i32, vp]),
    "eod_gn_mean_rstd": (i32, [vp, i32, i32, vp, i32, i32, i32, i64, i32, f32, vp, vp]),
    "eod_gn_bwd_partial": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp]),
    "eod_gn_bwd_finalize": (i32, [vp, i32, i32, i32, i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp]),
    "eod_gn_bwd_params": (i32, [vp, i32, i32, f32, vp, vp, vp]),
    "eod_gn_bwd_apply": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "eod_gn_bwd_apply_slabs": (i32, [i32, i32, i32, i32]),
    "eod_add": (i32, [vp, vp, vp, i32, i64, vp]),
    "eod_dropout": (i32, [vp, vp, i32, i64, f32, C.c_uint64, C.c_uint32, C.c_uint32, vp]),
    "eod_rowdot": (i32, [vp, vp, i32, i64, i64, i64, i64, i64, i64, i32, vp, vp]),
    "eod_scale_f32": (i32, [vp, i64, f32, vp]),
    "eod_attention_fwd_nat": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "eod_weight_l1max": (i32, [vp, i32, i32, vp, vp, vp]),
    "eod_bound_affine": (i32, [vp, vp, vp, i32, vp]),
    "eod_attention_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eod_gemm_tn": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, f32, i32, i32, i64, i64, i64, i64, i64, i64, vp]),
    "eod_softmax_bwd_rows": (i32, [vp, i64, vp, i64, vp, i32, i64, i32, vp]),
    "eod_linear_bwd_small": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "eod_temb_pre1": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "eod_embedding_bwd": (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
    "eod_mse_loss": (i32, [vp, vp, i64, vp, vp, vp, i32, vp]),
    "eod_adamw_step": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i32, vp]),
    "eod_adamw_step_guarded": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i32, vp, vp, i32, vp]),
    "eod_ema_update": (i32, [vp, vp, i64, f64, vp]),
    "eod_pack_rows": (i32, [vp, i64, vp, vp, i64, i32, i32, i32, vp]),
    "eod_nchw_to_nhwc": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "eod_nhwc_to_nchw": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "eod_gn_partial": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp]),
    "eod_gn_finalize": (i32, [vp, i32, i32, vp, i32, i32, i32, i64, i32, f32, vp, vp, vp, i64, vp, vp, vp, vp]),
    "eod_act_bound": (i32, [vp, i32, i32, i64, vp, i32, i32, vp, i32, i32, vp, i32, vp]),
    "eod_conv_stats_slots": (i32, [C.POINTER(ConvDesc)]),
    "eod_conv_gn_fusable": (i32, [C.POINTER(ConvDesc)]),
    "eod_conv_split_ok": (i32, [C.POINTER(ConvDesc)]),
    "eod_conv_skip_ok": (i32, [C.POINTER(ConvDesc)]),
    "eod_conv_up4_ok": (i32, [C.POINTER(ConvDesc)]),
    "eod_conv_up4_weights": (i32, [vp, vp, i32, i32, vp]),
    "eod_conv_up4_bwd_ok": (i32, [C.POINTER(ConvDesc)]),
    "eod_pack_conv_weight_split": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "eod_pack_conv_weight_split_pair": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eod_conv_workspace_size": (i64, [C.POINTER(ConvDesc)]),
    "eod_gn_apply": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp]),
    "eod_attention_fwd": (i32, [C.POINTER(AttnDesc), vp]),
    "eod_softmax_rows": (i32, [vp, i64, vp, i64, i32, i64, i32, vp]),
    "eod_time_embed": (i32, [C.POINTER(TembDesc), vp]),
    "eod_timestep_embedding": (i32, [vp, vp, vp, i32, i32, vp]),
    "eod_q_sample": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, vp]),
    "eod_repaint_mix": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "eod_ddpm_step": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "eod_ddim_step": (i32, [vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, i64, vp]),
    "eod_cfg_combine": (i32, [vp, vp, f32, vp, i64, vp]),
    "eod_ldm_p_sample": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "eod_repaint_cond": (i32, [vp, vp, vp, i32, i32, i64, i32, vp]),
    "eod_postprocess": (i32, [vp, vp, i64, i32, vp]),
    "eod_masked_preview": (i32, [vp, vp, vp, i32, i32, i64, f32, vp]),
    "eod_randn_philox": (i32, [vp, i32, i64, C.c_uint64, i64, i32, i32, vp]),
    "eod_program_run": (i32, [C.POINTER(Op), i32, vp]),
    "eod_timer_create": (vp, [i32, i32]),
    "eod_timer_destroy": (None, [vp]),
    "eod_timer_read": (i32, [vp, C.POINTER(f32)]),
    "eod_timer_set_mask": (i32, [vp, vp, i32]),
    "eod_program_run_timed": (i32, [C.POINTER(Op), i32, vp, vp]),
    "eod_resample2x": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
}

_lib = None


class EodError(RuntimeError):
    pass


ABI_VERSION = 104  # EOD_ABI_VERSION of the include/eodiff.h this file mirrors


def lib():
    """Load libeodiff.so once.  Raises if it is missing: the HIP extension is mandatory."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EodError(
            f"libeodiff.so not found at {LIB_PATH}. Build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C eo_diffusion_amd/csrc). There is no fallback path.")
    # libeodiff.so must share ONE HIP runtime with PyTorch (device pointers and streams cross the boundary): torch
    # bundles its own libamdhip64 under the same SONAME, so import torch first and let the loader reuse that copy.
    # (Loading /opt/rocm's runtime first ends in "no ROCm-capable device is detected" at the first launch.)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if L.eod_version() != ABI_VERSION and os.environ.get("EOD_ABI_ANY", "0") != "1":  # (EOD_ABI_ANY=1: A/B tooling against an older build)
        raise EodError(f"ABI mismatch: {LIB_PATH} was built for revision {L.eod_version()} of include/eodiff.h, this binding mirrors {ABI_VERSION}")
    for kind, st in ((1, ConvDesc), (2, GemmDesc), (3, TembDesc), (4, SmallDesc), (5, Op), (6, AttnDesc)):
        if L.eod_struct_size(kind) != C.sizeof(st):
            raise EodError(f"ABI mismatch: struct kind {kind}: C {L.eod_struct_size(kind)} vs ctypes {C.sizeof(st)}")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().eod_last_error()
        raise EodError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def dtype_id(torch_dtype):
    import torch
    if torch_dtype == torch.float16:
        return EOD_F16
    if torch_dtype == torch.float32:
        return EOD_F32
    raise EodError(f"unsupported storage dtype {torch_dtype}")


def ptr(t):
    return 0 if t is None else t.data_ptr()
