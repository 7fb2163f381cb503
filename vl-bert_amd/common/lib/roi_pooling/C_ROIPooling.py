"""Drop-in for the reference's pybind extension `common.lib.roi_pooling.C_ROIPooling`
(vision.cpp:6-11): same function names, argument order and tensor contract, backed by the HIP kernels
of libvlbert_hip.so through the C ABI (vlb_roi_align_fwd / vlb_roi_align_bwd).

    roi_align_forward(input[B,C,H,W], rois[K,5], spatial_scale, pooled_h, pooled_w, sampling_ratio) -> [K,C,ph,pw]
    roi_align_backward(grad[K,C,ph,pw], rois, spatial_scale, pooled_h, pooled_w, B, C, H, W, sampling_ratio) -> [B,C,H,W]

Contract kept from ROIAlign.h:11-45 / ROIAlign_cuda.cu:256-330: outputs freshly allocated by the callee,
inputs borrowed and made contiguous, fp32, enqueued on the current stream without synchronising, CPU
tensors rejected with RuntimeError ("Not compiled with ... support" in the reference).
ROIPool is not provided: the reference imports it (common/fast_rcnn.py:10) but never instantiates it.
"""
import torch

from .... import ops


def _check(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a GPU tensor (the MI355X build has no CPU ROIAlign)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (the reference wrapper casts with .float(), roi_align.py:69)" % name)


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    _check(input, "input")
    _check(rois, "rois")
    input, rois = input.contiguous(), rois.contiguous()
    out = torch.empty((rois.shape[0], input.shape[1], pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    return ops.roi_align_fwd(input, rois, out, spatial_scale, sampling_ratio)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    _check(grad, "grad")
    _check(rois, "rois")
    grad, rois = grad.contiguous(), rois.contiguous()
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    return ops.roi_align_bwd(grad, rois, gin, spatial_scale, sampling_ratio)


def roi_pool_forward(*a, **k):
    raise RuntimeError("ROIPool is dead code in the reference (never constructed) and is not provided")


roi_pool_backward = roi_pool_forward
