"""`ROIAlign` module / `roi_align` function with the reference's API
(common/lib/roi_pooling/roi_align.py:11-70), running the HIP kernels behind C_ROIPooling."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import C_ROIPooling


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(rois)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        return C_ROIPooling.roi_align_forward(input, rois, spatial_scale, ctx.output_size[0], ctx.output_size[1], sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        grad_input = C_ROIPooling.roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                                     bs, ch, h, w, ctx.sampling_ratio)
        return grad_input, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=1):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        """input [B,C,H,W], rois [k,5] (im_index, x1, y1, x2, y2) -> [k,C,ph,pw] (fp32, like the reference)."""
        return roi_align(input.float(), rois.float(), self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
