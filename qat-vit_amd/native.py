"""ctypes binding of libqatvit.so (C ABI: include/qatvit.h).

There is no CPU implementation behind this module: if the shared library is
missing or a call fails, the caller gets a RuntimeError - never a silent
PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqatvit.so")
CSRC = os.path.join(_HERE, "csrc")
_lib = None

# name -> (restype, argtypes); every symbol include/qatvit.h declares
SIGNATURES = {
    "qatvit_abi_version": (c_int, []),
    "qatvit_last_error": (c_char_p, []),
    "qatvit_target_arch": (c_char_p, []),
    "qatvit_fq_workspace_bytes": (c_int64, [c_int64]),
    "qatvit_fq_forward": (c_int, [c_void_p] * 9 + [c_float, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "qatvit_fq_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "qatvit_ln_forward": (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_float, c_void_p]),
    "qatvit_ln_backward": (c_int, [c_void_p] * 8 + [c_int64, c_int64, c_void_p]),
    "qatvit_kd_ce_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "qatvit_optim_grad_norm": (c_int, [c_void_p] * 4 + [c_int32, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "qatvit_optim_adamw": (c_int, [c_void_p] * 7 + [c_int32, c_int64] + [c_double] * 5 + [c_int64, c_void_p, c_void_p]),
    "qatvit_gemm_nt": (c_int, [c_void_p] * 4 + [c_int32] * 6 + [c_void_p] * 6),
    "qatvit_gemm_nt_f16": (c_int, [c_void_p] * 4 + [c_int32] * 6 + [c_void_p] * 6),
    "qatvit_gemm_nt_i8_minmax": (c_int, [c_void_p] * 4 + [c_int32] * 6 + [c_void_p] * 6),
    "qatvit_w8_fragment_order": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "qatvit_i8_strip": (c_int, [c_int32] + [c_void_p] * 4 + [c_int32] * 5 + [c_void_p] * 6 + [c_int32, c_int32, c_void_p, c_void_p, c_int32] + [c_void_p] * 4),
    "qatvit_gemm_nt_codes": (c_int, [c_void_p] * 4 + [c_int32] * 6 + [c_void_p] * 6),
    "qatvit_gemm_nt_i8": (c_int, [c_void_p] * 4 + [c_int32, c_void_p] + [c_int32] * 6 + [c_void_p] * 6),
    "qatvit_gemm_tn_scratch_bytes": (c_int64, []),
    "qatvit_gemm_tn": (c_int, [c_void_p] * 5 + [c_int32] * 6 + [c_void_p] * 4 + [c_int32] * 3 + [c_void_p] * 3 + [c_int64, c_void_p]),
    "qatvit_gemm_tn_codes": (c_int, [c_void_p] * 5 + [c_int32] * 6 + [c_void_p] * 4 + [c_int32] * 3 + [c_void_p] * 3 + [c_int64, c_void_p]),
    "qatvit_gemm_nt_dy16": (c_int, [c_void_p] * 3 + [c_int32] * 6 + [c_void_p] * 3),
    "qatvit_gemm_tn_dy16": (c_int, [c_void_p] * 6 + [c_int32] * 6 + [c_void_p] * 5 + [c_int32] * 3 + [c_void_p] * 3 + [c_int64, c_void_p]),
    "qatvit_gemm_tn_q8_dy16": (c_int, [c_void_p] * 3 + [c_int32, c_void_p] + [c_int32] * 6 + [c_void_p] * 4 + [c_int32] * 3 + [c_void_p] * 3 + [c_int64, c_void_p]),
    "qatvit_gemm_tn_stream_scratch_bytes": (c_int64, []),
    "qatvit_gemm_tn_stream_dy16": (c_int, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    "qatvit_attn_padded_tokens": (c_int32, [c_int32]),
    "qatvit_attn_forward": (c_int, [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p] * 4),
    "qatvit_attn_forward_f16": (c_int, [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p] * 9),
    "qatvit_attn_backward": (c_int, [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p] * 11),
    "qatvit_student_num_params": (c_int32, [c_void_p]),
    "qatvit_student_num_act_fq": (c_int32, [c_void_p]),
    "qatvit_student_num_weight_fq": (c_int32, [c_void_p]),
    "qatvit_student_workspace_bytes": (c_int64, [c_void_p]),
    "qatvit_student_init": (c_int, [c_void_p, c_void_p, c_void_p]),
    "qatvit_student_forward": (c_int, [c_void_p] * 8),
    "qatvit_student_backward": (c_int, [c_void_p] * 7 + [c_int32, c_int32, c_void_p]),
    "qatvit_student_forward_stages": (c_int, [c_void_p] * 7 + [c_int32, c_int32, c_int32, c_void_p]),
    "qatvit_student_forward_part": (c_int, [c_void_p] * 5 + [c_int32, c_int32, c_int32, c_void_p]),
    "qatvit_student_backward_stages": (c_int, [c_void_p] * 7 + [c_int32, c_int32, c_int32, c_void_p]),
    "qatvit_student_tensor_offset": (c_int64, [c_void_p, c_char_p, c_int32]),
    "qatvit_student_dy16_supported": (c_int32, [c_void_p]),
    "qatvit_student_dy16_to_pair": (c_int, [c_void_p, c_void_p, c_void_p]),
    "qatvit_student_dy16_set_mirror": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "qatvit_teacher_workspace_bytes": (c_int64, [c_void_p]),
    "qatvit_teacher_forward": (c_int, [c_void_p] * 8),
    "qatvit_teacher_forward_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "qatvit_infer_workspace_bytes": (c_int64, [c_void_p]),
    "qatvit_infer_prepare": (c_int, [c_void_p] * 6),
    "qatvit_infer_forward": (c_int, [c_void_p] * 8),
    "qatvit_profile_start": (c_int, [c_void_p, c_int32, c_int32]),
    "qatvit_profile_stop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
}


class Cfg(ctypes.Structure):
    """struct qatvit_cfg (include/qatvit.h)."""
    _fields_ = [(n, c_int32) for n in ("batch", "img_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads", "mlp_hidden",
                                       "num_classes", "act_qmin", "act_qmax", "w_qmin", "w_qmax", "w_per_channel")] + \
               [("averaging_const", c_float), ("ln_eps", c_float)]


class TNItem(ctypes.Structure):
    """struct qatvit_tn_item (include/qatvit.h)."""
    _fields_ = [(n, c_void_p) for n in ("P", "Q", "lut", "s1", "s2", "C", "W", "w_scale", "w_zp", "dbias", "row_div")] + [(n, c_int32) for n in ("N", "Kw", "ldp", "ldq", "ldc")]


class FQ(ctypes.Structure):
    """struct qatvit_fq (include/qatvit.h)."""
    _fields_ = [(n, c_void_p) for n in ("min_val", "max_val", "scale", "zero_point", "observer_on", "fake_quant_on")]


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libqatvit.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libqatvit.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X path has no fallback. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library drift
            fn.restype, fn.argtypes = res, args
        if L.qatvit_abi_version() != 4:
            raise RuntimeError("libqatvit.so ABI version mismatch")
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {lib().qatvit_last_error().decode()}")


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
