"""`ROIAlign` module / `roi_align` function with the reference's API (common/lib/roi_pooling/roi_align.py:11-70: same names, argument
order and fp32 output), running the HIP kernels behind `C_ROIPooling` (vlb_roi_align_fwd / vlb_roi_align_bwd, NCHW fp32)."""
import torch
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import C_ROIPooling


class _ROIAlign(torch.autograd.Function):
    """One autograd node: the pooling geometry and the feature-map shape travel to backward as a plain tuple, the RoIs as the only
    saved tensor.  Not twice differentiable (the gradient kernel has no autograd formula), like the reference's."""

    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio):
        bins_h, bins_w = _pair(output_size)
        ctx.geom = (float(spatial_scale), int(bins_h), int(bins_w), int(sampling_ratio)) + tuple(int(d) for d in input.shape)
        ctx.save_for_backward(rois)
        return C_ROIPooling.roi_align_forward(input, rois, *ctx.geom[:4])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        scale, bins_h, bins_w, ratio, n, c, h, w = ctx.geom
        (rois,) = ctx.saved_tensors
        d_input = C_ROIPooling.roi_align_backward(grad_output, rois, scale, bins_h, bins_w, n, c, h, w, ratio)
        return d_input, None, None, None, None        # no gradient w.r.t. the RoIs or the geometry


roi_align = _ROIAlign.apply


class ROIAlign(torch.nn.Module):
    """input [B,C,H,W], rois [k,5] = (image index, x1, y1, x2, y2) in image pixels -> [k,C,bins_h,bins_w]; both are cast to fp32 first,
    as in the reference (`output_size` e.g. (14, 14), `spatial_scale` e.g. 1/16, `sampling_ratio` samples per bin side, <= 0: adaptive)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio=1):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        return roi_align(input.float(), rois.float(), self.output_size, self.spatial_scale, self.sampling_ratio)

    def extra_repr(self):
        return "output_size=%s, spatial_scale=%s, sampling_ratio=%s" % (self.output_size, self.spatial_scale, self.sampling_ratio)
