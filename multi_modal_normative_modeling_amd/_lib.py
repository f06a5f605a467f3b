"""ctypes binding of libnmhip.so (include/nmhip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, the
caller gets an exception.  ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NM_MAX_MOD = 8
NM_MAX_EXP = 4
NM_MAX_HID = 8
NM_MAX_CLS = 5
NM_MAX_CLS_WIDTH = 512
NM_MAX_CLASSES = 4
NM_BATCH = 256
NM_MAX_WIDTH = 127
NM_MAX_LATENT = 64
NM_WIDE_MAX_WIDTH = 4096
NM_WIDE_MAX_LATENT = 128
NM_LOSS_STRIDE = 16

NM_COMBINE = {"poe": 0, "gpoe": 1, "moe": 2, "mopoe": 3, "poe2v": 4}

NM_F_BACKWARD = 1
NM_F_ADAM = 2
NM_F_GRADS = 4
NM_F_EXPORT = 8
NM_F_PROFILE = 16
NM_F_ZGIVEN = 32
NM_F_TRACE = 64
NM_LOSS_TC = 11
NM_LOSS_REG = 12
NM_LOSS_CE = 13
NM_LOSS_CONTRAST = 14
NM_F_BNSTATS = 256
NM_F_SPLIT = 512
NM_F_FAULT_INJECT = 1024
NM_METRICS_MAX_N = 8192
NM_METRICS_STRIDE = 8

LIB_NAME = os.environ.get("NMHIP_LIB_NAME", "libnmhip.so")     # diagnostic builds (tools/ablate.py) override the name
LIB_PATH = Path(__file__).resolve().parent / LIB_NAME


class NmModality(C.Structure):
    _fields_ = [
        ("D", C.c_int32), ("Kx", C.c_int32), ("x_pitch", C.c_int32), ("Cz", C.c_int32),
        ("x_f32", C.c_void_p), ("xb", C.c_void_p), ("cz", C.c_void_p),
        ("enc_w", C.c_int64 * NM_MAX_HID), ("enc_b", C.c_int64 * NM_MAX_HID),
        ("mu_w", C.c_int64), ("mu_b", C.c_int64), ("lv_w", C.c_int64), ("lv_b", C.c_int64),
        ("logvar_out", C.c_int64),
        ("dec_w", C.c_int64 * NM_MAX_HID), ("dec_b", C.c_int64 * NM_MAX_HID),
        ("out_w", C.c_int64), ("out_b", C.c_int64),
        ("alpha", C.c_int64),
        ("enc_s", C.c_int64 * NM_MAX_HID), ("heads_s", C.c_int64), ("dec_s", C.c_int64 * NM_MAX_HID), ("out_s", C.c_int64),
        ("out_loc", C.c_void_p), ("out_sqerr", C.c_void_p), ("out_rowdev", C.c_void_p),
        ("dloc_extra", C.c_void_p), ("dloc_rowcoef", C.c_void_p),
    ]


class NmJob(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("M_enc", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("Z", C.c_int32),
        ("H", C.c_int32 * NM_MAX_HID),
        ("combine", C.c_int32), ("single_bypass", C.c_int32), ("n_rows", C.c_int32), ("non_linear", C.c_int32),
        ("act_slope", C.c_float), ("out_kind", C.c_int32), ("n_private", C.c_int32),
        ("var_floor", C.c_float), ("tc_weight", C.c_float), ("w_off", C.c_int64), ("dephase", C.c_int32), ("shared_cov", C.c_int32), ("wide", C.c_int32),
        ("loss_cap", C.c_int32), ("eps_cap", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("adam_off", C.c_int64), ("lr_table", C.c_void_p), ("lr_cap", C.c_int32),
        ("kl_weight", C.c_float), ("ll_weight", C.c_float),
        ("params", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("grads", C.c_void_p),
        ("eps", C.c_void_p), ("seed", C.c_uint64),
        ("loss_log", C.c_void_p), ("wsh", C.c_void_p), ("workspace", C.c_void_p), ("workspace_stride", C.c_int64),
        ("out_mu", C.c_void_p), ("out_logvar", C.c_void_p), ("out_z", C.c_void_p), ("dz_extra", C.c_void_p),
        ("reg_head", C.c_int32), ("reg_lambda", C.c_float), ("reg_w", C.c_int64 * 3), ("reg_b", C.c_int64 * 3),
        ("reg_s", C.c_int64), ("reg_resid", C.c_void_p), ("reg_dres", C.c_void_p),
        ("fi_target", C.c_void_p), ("out_fi_pred", C.c_void_p),
        ("cls_layers", C.c_int32), ("cls_classes", C.c_int32), ("cls_width", C.c_int32 * NM_MAX_CLS),
        ("cls_train", C.c_int32), ("cls_use_mu", C.c_int32),
        ("cls_w", C.c_int64 * (NM_MAX_CLS + 1)), ("cls_b", C.c_int64 * (NM_MAX_CLS + 1)),
        ("cls_bn_w", C.c_int64 * NM_MAX_CLS), ("cls_bn_b", C.c_int64 * NM_MAX_CLS),
        ("cls_bn_mean", C.c_int64 * NM_MAX_CLS), ("cls_bn_var", C.c_int64 * NM_MAX_CLS),
        ("cls_dropout", C.c_float), ("cls_margin", C.c_float), ("cls_w_ce", C.c_float), ("cls_w_contrast", C.c_float),
        ("labels", C.c_void_p), ("out_logits", C.c_void_p), ("dz_out", C.c_void_p),
        ("rowcoef_out", C.c_void_p * NM_MAX_MOD),
        ("gpart", C.c_void_p), ("gpart_stride", C.c_int64), ("n_params", C.c_int64),
        ("mod", NmModality * NM_MAX_MOD),
    ]


class NmError(RuntimeError):
    pass


_lib = None


def load():
    """Load libnmhip.so from the package directory; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NmError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.nm_version.restype = C.c_int
    lib.nm_status_string.restype = C.c_char_p
    lib.nm_status_string.argtypes = [C.c_int]
    lib.nm_abi_sizes.argtypes = [C.POINTER(i64), C.POINTER(i64)]
    lib.nm_workspace_bytes.restype = i64
    lib.nm_fill_shadow.restype = i64
    lib.nm_fill_shadow.argtypes = [C.POINTER(NmJob)]
    lib.nm_sync_shadow.argtypes = [vp, i32, vp]
    lib.nm_xb_elems.restype = i64
    lib.nm_xb_elems.argtypes = [i32, i32]
    lib.nm_workspace_bytes.argtypes = [C.POINTER(NmJob)]
    lib.nm_workspace_offset.restype = i64
    lib.nm_workspace_offset.argtypes = [C.POINTER(NmJob), i32]
    lib.nm_validate_job.argtypes = [C.POINTER(NmJob)]
    for name in ("nm_launch", "nm_launch_scalar_tr"):
        getattr(lib, name).argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.nm_launch_split.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.nm_launch_wide.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.nm_split_errors.argtypes = [vp, i32, vp, i32, vp]
    lib.nm_launch_rowsplit.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.nm_rowsplit_ok.argtypes = [C.POINTER(NmJob)]
    lib.nm_sync_reset.argtypes = [vp, i32, vp]
    lib.nm_trace_read_rs.argtypes = [C.POINTER(C.c_ulonglong), i32]
    lib.nm_devpass.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.nm_trace_read_dv.argtypes = [C.POINTER(C.c_ulonglong), i32]
    lib.nm_devpass_ok.argtypes = [C.POINTER(NmJob)]
    lib.nm_combine_latent.argtypes = [vp, vp, i32, i64, i32, vp, i32, i32, i32, f32, vp, vp, vp]
    lib.nm_total_correlation.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.nm_train_steps.argtypes = [vp, i32, i32, i32, vp]
    lib.nm_train_steps_persistent.argtypes = [vp, i32, i32, i32, vp]
    lib.nm_deviation.argtypes = [vp, i32, i32, i32, vp]
    lib.nm_grads.argtypes = [vp, i32, i32, vp]
    lib.nm_forward.argtypes = [vp, i32, i32, i32, vp]
    lib.nm_head_regression.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.nm_head_classifier.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    lib.nm_train_steps_head.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.nm_train_steps_head.restype = i32
    lib.nm_posthoc_metrics.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.nm_confusion_metrics.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.nm_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp]
    lib.nm_pack_table.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp]
    lib.nm_prep_scaler_fit.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp]
    lib.nm_prep_onehot.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, vp, vp]
    lib.nm_pack_table_raw.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp]
    lib.nm_test_gemm.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
    lib.nm_prof_read.argtypes = [C.POINTER(C.c_ulonglong), i32]
    lib.nm_trace_read.argtypes = [C.POINTER(C.c_ulonglong), i32]
    lib.nm_wgtimes_read.argtypes = [C.POINTER(C.c_ulonglong)]
    sj, sm = i64(0), i64(0)
    lib.nm_abi_sizes(C.byref(sj), C.byref(sm))
    if sj.value != C.sizeof(NmJob) or sm.value != C.sizeof(NmModality):
        raise NmError(f"ABI mismatch: C sizeof(nm_job_t)={sj.value}, sizeof(nm_modality_t)={sm.value}; "
                      f"ctypes {C.sizeof(NmJob)}, {C.sizeof(NmModality)}")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "nm_version", "nm_status_string", "nm_abi_sizes", "nm_workspace_bytes", "nm_validate_job", "nm_launch",
    "nm_launch_scalar_tr", "nm_train_steps", "nm_grads", "nm_forward", "nm_adam_step", "nm_pack_table",
    "nm_test_gemm", "nm_prof_read", "nm_trace_read", "nm_wgtimes_read", "nm_head_regression", "nm_head_classifier", "nm_train_steps_head", "nm_train_steps_persistent", "nm_deviation", "nm_posthoc_metrics", "nm_confusion_metrics",
    "nm_fill_shadow", "nm_sync_shadow", "nm_xb_elems", "nm_launch_split", "nm_launch_wide", "nm_split_errors", "nm_combine_latent", "nm_total_correlation",
    "nm_prep_scaler_fit", "nm_prep_onehot", "nm_pack_table_raw",
    "nm_launch_rowsplit", "nm_rowsplit_ok", "nm_sync_reset", "nm_trace_read_rs", "nm_devpass", "nm_devpass_ok", "nm_trace_read_dv", "nm_workspace_offset",
]


def check(status: int, what: str = "nmhip"):
    if status != 0:
        msg = load().nm_status_string(status).decode()
        raise NmError(f"{what} failed with status {status}: {msg}")
