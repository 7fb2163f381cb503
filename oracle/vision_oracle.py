"""CPU oracle for the end-to-end vision path.

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header for the import rules): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates in plain torch-CPU fp32 (NCHW, like the reference):
  * `bottleneck`       -- common/backbone/resnet/resnet.py:75-118 (Bottleneck: 1x1 -> 3x3 -> 1x1, BN after each,
                          residual add, ReLU; caffe-style stride_in_1x1 :79; padding = dilation :89)
  * `backbone`         -- resnet.py:175-199 (ResNet.forward: 7x7/2 conv, BN, ReLU, 3x3/2 max-pool, layer1..layer3 -> 'body4')
  * `roi_head`         -- common/fast_rcnn.py:74-84 (layer4 built by _make_layer with stride 1 / dilation 2 when
                          IMAGE_C5_DILATED, AvgPool2d(14), Flattener)
  * `roi_align`        -- common/lib/roi_pooling/roi_align.py:11-44 as a torch.autograd.Function over oracle/roi_align_oracle.py
  * `e2e_features`     -- common/fast_rcnn.py:144-156 (backbone -> rois -> ROIAlign(14x14, 1/16, sampling_ratio=1) -> head)
BatchNorm runs in eval mode with frozen affine parameters (common/fast_rcnn.py:88-100,122-126): y = (x-mean)/sqrt(var+1e-5)*g+b.
Parameters use the reference's state-dict names and OIHW layout (`init_vision_params` = torchvision ResNet key names:
conv1, bn1, layer1..layer4; FastRCNN maps layer1-3 under `backbone.` and layer4 under `roi_head_feature_extractor.`,
common/fast_rcnn.py:111-118).

Pinning: tests/golden/vision_small.npz is produced by oracle/make_golden.py by running the REFERENCE's own
FastRCNN(e2e) module on CPU with these parameters; tests/test_oracle_golden.py checks this restatement against it.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import roi_align_oracle as RA

BN_EPS = 1e-5
MODEL_LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}   # resnet.py model_layers


def block_specs(num_layers=101, stride_in_1x1=True, c5_dilated=True):
    """[(state-dict prefix, inplanes, planes, stride, dilation, has_downsample, stride_in_1x1)] for layer1..layer4
    (resnet.py:160-173 _make_layer; layer4 as FastRCNN builds the RoI head, common/fast_rcnn.py:74-78)."""
    blocks = MODEL_LAYERS[num_layers]
    specs = []
    inplanes = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
        stride = 1 if li == 0 else 2
        dilation = 1
        if li == 3 and c5_dilated:
            stride, dilation = 1, 2
        for b in range(nb):
            first = b == 0
            specs.append(("layer%d.%d." % (li + 1, b), inplanes, planes, stride if first else 1, dilation,
                          first and (stride != 1 or inplanes != planes * 4), stride_in_1x1 if first else False))
            inplanes = planes * 4
    return specs


def init_vision_params(seed, num_layers=101, randomize_bn=True):
    """Deterministic torchvision-style ResNet state dict (fp32).  Convolutions ~ N(0, sqrt(2/fan_out)) as resnet.py:153-155;
    BatchNorm statistics / affine parameters randomised (a pretrained checkpoint has non-trivial ones; identity BN would hide
    folding bugs)."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()

    def conv(name, o, i, k):
        P[name + ".weight"] = torch.randn(o, i, k, k, generator=g) * (2.0 / (o * k * k)) ** 0.5

    def bn(name, c):
        if randomize_bn:
            # the BN closing a residual branch (bn3) and the stem get small gains so activations stay O(1) over 33 blocks
            lo, span = (0.1, 0.2) if (name.endswith("bn3") or name == "bn1") else (0.5, 1.0)
            P[name + ".weight"] = lo + span * torch.rand(c, generator=g)
            P[name + ".bias"] = 0.2 * torch.randn(c, generator=g)
            P[name + ".running_mean"] = 0.2 * torch.randn(c, generator=g)
            P[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        else:
            P[name + ".weight"], P[name + ".bias"] = torch.ones(c), torch.zeros(c)
            P[name + ".running_mean"], P[name + ".running_var"] = torch.zeros(c), torch.ones(c)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    for prefix, inpl, planes, stride, dil, ds, s1 in block_specs(num_layers):
        conv(prefix + "conv1", planes, inpl, 1)
        bn(prefix + "bn1", planes)
        conv(prefix + "conv2", planes, planes, 3)
        bn(prefix + "bn2", planes)
        conv(prefix + "conv3", planes * 4, planes, 1)
        bn(prefix + "bn3", planes * 4)
        if ds:
            conv(prefix + "downsample.0", planes * 4, inpl, 1)
            bn(prefix + "downsample.1", planes * 4)
    return P


def frozen_names(P, frozen_stages=(1, 2)):
    """Names that receive no gradient: every BatchNorm tensor (IMAGE_FROZEN_BN) and the stages in
    IMAGE_FROZEN_BACKBONE_STAGES (resnet.py:201-222: stage 1 = conv1/bn1, stage s = layer(s-1))."""
    out = set()
    for k in P:
        if ".bn" in k or k.startswith("bn1") or "downsample.1" in k:
            out.add(k)
        if 1 in frozen_stages and k.startswith("conv1"):
            out.add(k)
        for s in frozen_stages:
            if s > 1 and k.startswith("layer%d." % (s - 1)):
                out.add(k)
    return out


def _bn(x, P, name):
    return F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"], P[name + ".bias"],
                        training=False, eps=BN_EPS)


def bottleneck(x, P, prefix, stride, dilation, has_downsample, stride_in_1x1):
    """resnet.py:98-118"""
    s1 = stride if stride_in_1x1 else 1
    s3 = 1 if stride_in_1x1 else stride
    out = F.relu(_bn(F.conv2d(x, P[prefix + "conv1.weight"], stride=s1), P, prefix + "bn1"))
    out = F.relu(_bn(F.conv2d(out, P[prefix + "conv2.weight"], stride=s3, padding=dilation, dilation=dilation), P, prefix + "bn2"))
    out = _bn(F.conv2d(out, P[prefix + "conv3.weight"]), P, prefix + "bn3")
    residual = x
    if has_downsample:
        residual = _bn(F.conv2d(x, P[prefix + "downsample.0.weight"], stride=stride), P, prefix + "downsample.1")
    return F.relu(out + residual)


def backbone(img, P, num_layers=101):
    """resnet.py:175-199 up to 'body4' (layer3 output, stride 16)."""
    x = F.relu(_bn(F.conv2d(img, P["conv1.weight"], stride=2, padding=3), P, "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for prefix, inpl, planes, stride, dil, ds, s1 in block_specs(num_layers):
        if prefix.startswith("layer4"):
            break
        x = bottleneck(x, P, prefix, stride, dil, ds, s1)
    return x


def roi_head(roi_feats, P, num_layers=101, segms=None):
    """common/fast_rcnn.py:80-84: layer4 (dilated, stride 1) -> AvgPool2d(14) -> flatten; with `segms` [K,14,14] the layer4 output is
    multiplied by the object mask first (:152-156, the VCR call)."""
    x = roi_feats
    for prefix, inpl, planes, stride, dil, ds, s1 in block_specs(num_layers):
        if prefix.startswith("layer4"):
            x = bottleneck(x, P, prefix, stride, dil, ds, s1)
    if segms is not None:
        x = x * segms[:, None].to(x.dtype)
    return F.avg_pool2d(x, x.shape[-1], stride=1).flatten(1)


class _RoiAlign(torch.autograd.Function):
    """common/lib/roi_pooling/roi_align.py:11-44 over the numpy oracle."""

    @staticmethod
    def forward(ctx, feat, rois, out_size, scale, sampling_ratio):
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(feat.shape), out_size, scale, sampling_ratio)
        out = RA.roi_align_forward(feat.detach().numpy().astype(np.float32), rois.numpy().astype(np.float32), scale, out_size,
                                   out_size, sampling_ratio)
        return torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32))

    @staticmethod
    def backward(ctx, grad):
        rois, = ctx.saved_tensors
        (b, c, h, w), out_size, scale, sr = ctx.cfg
        g = RA.roi_align_backward(grad.contiguous().numpy().astype(np.float32), rois.numpy().astype(np.float32), scale, out_size,
                                  out_size, b, c, h, w, sr)
        return torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)), None, None, None, None


def roi_align(feat, rois, out_size=14, scale=1.0 / 16, sampling_ratio=1):
    return _RoiAlign.apply(feat, rois, out_size, scale, sampling_ratio)


def mask_raw_pixels(image, boxes, mvrc_ops):
    """pretrain/data/datasets/conceptual_captions.py:201-206 (= coco_captions.py:240-244), per sample of a collated batch:
        for mvrc_op, box in zip(mvrc_ops, boxes):
            if mvrc_op == 1:
                x1, y1, x2, y2 = box
                image[:, int(y1):(int(y2)+1), int(x1):(int(x2)+1)] = 0
    image [B,3,H,W] (modified in place, returned), boxes [B,R,>=4], mvrc_ops [B,R].  Pinned: tests/test_data_cpu.py applies it to the
    unmasked samples of tests/fixtures/cc_tiny and compares with the images the reference's own ConceptualCaptionsDataset produced with
    MASK_RAW_PIXELS (tests/golden/data/cc_tiny.npz, oracle/make_data_golden.py) -- bit for bit."""
    for b in range(image.shape[0]):
        for mvrc_op, box in zip(mvrc_ops[b].tolist(), boxes[b]):
            if mvrc_op == 1:
                x1, y1, x2, y2 = [float(v) for v in box[:4]]
                image[b, :, int(y1):(int(y2) + 1), int(x1):(int(x2) + 1)] = 0
    return image


def rois_from_boxes(boxes):
    """common/fast_rcnn.py:136,145-149: (batch index, x1, y1, x2, y2) of every valid box, batch-major."""
    box_mask = boxes[:, :, 0] > -1.5
    inds = box_mask.nonzero()
    return torch.cat((inds[:, 0, None].to(boxes.dtype), boxes[inds[:, 0], inds[:, 1]][:, :4]), 1), inds


def e2e_features(img, boxes, P, num_layers=101, segms=None):
    """-> (post_roialign [K, 2048] for the valid boxes in batch-major order, body4).  segms: [B, R, 14, 14] or None."""
    body4 = backbone(img, P, num_layers)
    rois, inds = rois_from_boxes(boxes)
    pooled = roi_align(body4, rois)
    sel = segms[inds[:, 0], inds[:, 1]] if segms is not None else None
    return roi_head(pooled, P, num_layers, sel), body4


def split_state_dict(P):
    """torchvision-style dict -> FastRCNN names (common/fast_rcnn.py:55-56,74-78,111-118)."""
    out = OrderedDict()
    for k, v in P.items():
        if k.startswith("layer4."):
            out["roi_head_feature_extractor." + k[len("layer4."):]] = v
        else:
            out["backbone." + k] = v
    return out
