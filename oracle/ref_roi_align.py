"""ctypes access to oracle/_ref/libroi_align_ref.so = the REFERENCE's own CPU ROIAlign forward
(common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp compiled unmodified from /root/reference by oracle/build_ref.sh).

TEST INFRASTRUCTURE ONLY: pins oracle/roi_align_oracle.py (and through it the HIP kernels) to the reference binary on
arbitrary inputs.  The library exists where build_ref.sh ran (the build container) and travels to the GPU box inside the
repo snapshot; `available()` is False elsewhere and the tests that need it skip.
"""
import ctypes
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libroi_align_ref.so")
_h = None


def available():
    return os.path.isfile(LIB)


def _lib():
    global _h
    if _h is None:
        import torch  # noqa: F401  (libtorch / libc10 must be loaded first)
        _h = ctypes.CDLL(LIB)
        _h.ref_roi_align_forward_f32.restype = ctypes.c_int
    return _h


def roi_align_forward(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    """inp [B,C,H,W] fp32, rois [K,5] fp32 -> [K,C,ph,pw] fp32 computed by the reference's compiled kernel."""
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.empty((K, C, ph, pw), dtype=np.float32)
    f = ctypes.POINTER(ctypes.c_float)
    rc = _lib().ref_roi_align_forward_f32(inp.ctypes.data_as(f), B, C, H, W, rois.ctypes.data_as(f), K, ctypes.c_float(spatial_scale),
                                          ph, pw, sampling_ratio, out.ctypes.data_as(f))
    if rc != 0:
        raise RuntimeError("reference ROIAlign_forward_cpu failed")
    return out
