"""CPU oracle for ROIAlign forward / backward.

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header for the import rules).

Restates, in plain numpy loops (small cases only):
  * forward  -- common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp:17-111 (sample pre-calculation) and
                :113-219 (ROIAlignForward_cpu_kernel); identical math to cuda/ROIAlign_cuda.cu:15-122
  * backward -- cuda/ROIAlign_cuda.cu:125-254 (bilinear_interpolate_gradient + RoIAlignBackwardFeature);
                the reference has NO CPU backward (ROIAlign.h:44 raises "Not implemented on the CPU").

Pinning: forward is pinned to the known answer of the reference's own fixture
(common/lib/roi_pooling/debug.py:10-11 inputs -> aligned[0,0] = [[15,18,21],[42,45,48],[69,72,75]],
produced by the reference's compiled CPU kernel) in tests/test_roi_align_oracle.py.
Backward has no runnable reference on CPU: it is pinned to the forward by a finite-difference /
adjoint identity test  <dOut, fwd(x)> == <bwd(dOut), x>  (fwd is linear in x) -- "parity pinned
through the forward".
"""
import math

import numpy as np


def _sample(y, x, height, width):
    """-> list of (flat_index, weight); empty when the sample lies outside [-1, size]."""
    if y < -1.0 or y > height or x < -1.0 or x > width:
        return []
    y = max(y, 0.0)
    x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= height - 1:
        y_high = y_low = height - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= width - 1:
        x_high = x_low = width - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return [(y_low * width + x_low, hy * hx), (y_low * width + x_high, hy * lx),
            (y_high * width + x_low, ly * hx), (y_high * width + x_high, ly * lx)]


def _roi_samples(roi, spatial_scale, height, width, ph_n, pw_n, sampling_ratio, dtype):
    """Per output bin: list of (flat_index, weight/count) over the sampling grid."""
    f = dtype
    batch = int(roi[0])
    start_w, start_h = f(roi[1]) * f(spatial_scale), f(roi[2]) * f(spatial_scale)
    end_w, end_h = f(roi[3]) * f(spatial_scale), f(roi[4]) * f(spatial_scale)
    roi_w = max(end_w - start_w, f(1.0))
    roi_h = max(end_h - start_h, f(1.0))
    bin_h, bin_w = roi_h / f(ph_n), roi_w / f(pw_n)
    grid_h = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / ph_n))
    grid_w = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pw_n))
    count = grid_h * grid_w
    bins = []
    for ph in range(ph_n):
        for pw in range(pw_n):
            s = []
            for iy in range(grid_h):
                y = start_h + f(ph) * bin_h + f(iy + 0.5) * bin_h / f(grid_h)
                for ix in range(grid_w):
                    x = start_w + f(pw) * bin_w + f(ix + 0.5) * bin_w / f(grid_w)
                    s += [(i, w / count) for i, w in _sample(float(y), float(x), height, width)]
            bins.append(s)
    return batch, bins


def roi_align_forward(inp, rois, spatial_scale, ph_n, pw_n, sampling_ratio, dtype=np.float32):
    """inp [B,C,H,W], rois [K,5] -> [K,C,ph,pw]"""
    inp = np.asarray(inp, dtype=np.float64)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, ph_n, pw_n), dtype=np.float64)
    flat = inp.reshape(B, C, H * W)
    for n in range(K):
        batch, bins = _roi_samples(rois[n], spatial_scale, H, W, ph_n, pw_n, sampling_ratio, dtype)
        for b, s in enumerate(bins):
            for i, w in s:
                out[n, :, b // pw_n, b % pw_n] += w * flat[batch, :, i]
    return out


def roi_align_backward(grad_out, rois, spatial_scale, ph_n, pw_n, batch_size, C, H, W, sampling_ratio,
                       dtype=np.float32):
    """grad_out [K,C,ph,pw] -> grad_input [B,C,H,W] (scatter of the forward's weights)."""
    grad_out = np.asarray(grad_out, dtype=np.float64)
    gin = np.zeros((batch_size, C, H * W), dtype=np.float64)
    for n in range(rois.shape[0]):
        batch, bins = _roi_samples(rois[n], spatial_scale, H, W, ph_n, pw_n, sampling_ratio, dtype)
        for b, s in enumerate(bins):
            for i, w in s:
                gin[batch, :, i] += w * grad_out[n, :, b // pw_n, b % pw_n]
    return gin.reshape(batch_size, C, H, W)
