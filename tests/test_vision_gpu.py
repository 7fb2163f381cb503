"""-m gpu parity tests of the end-to-end vision path: every kernel of csrc/vision.hip and the
convolution epilogues of csrc/gemm.hip against plain torch-CPU fp32 statements, then the whole VisionStack (ResNet trunk ->
ROIAlign -> dilated layer4 head -> avg-pool, forward + hand-scheduled backward) against oracle/vision_oracle.py on the fixture
the REFERENCE's FastRCNN e2e module produced (tests/golden/vision/vision_small.npz), then one e2e pretraining step of the engine.

Tolerances: bf16 activations (2^-8 per rounding) through up to 19 Bottlenecks: features within 2e-2 of the tensor scale;
weight gradients by relative Frobenius error per tensor (ReLU sign flips of bf16-vs-fp32 pre-activations near 0 add gradient
noise that is not a kernel error, cf. tests/test_engine_gpu.py).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import roi_align_oracle as RA
from oracle import vision_oracle as VO
from oracle import vlbert_oracle as O
from tests.gpu_util import bf, dev, drop_scale, drop_thr, keep_mask, pkg, report, to_gpu_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    return pkg("ops")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf(torch.randn(*shape, generator=g) * scale)


def nhwc(x):     # [N,C,H,W] -> rows [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def rel_fro(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (515, 64, 64), (1000, 2048, 512)])
def test_gemm_conv_epilogues(ops, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res, aux = rnd(M, N, seed=4), rnd(M, N, seed=5)
    aux[::3] = 0.0                                   # exact zeros are "not > 0"
    a, b, r, x = to_gpu_bf16(A), to_gpu_bf16(B), to_gpu_bf16(res), to_gpu_bf16(aux)
    C = torch.zeros((M, N), dtype=torch.bfloat16, device=dev())
    ops.gemm_nt(a, b, C, bias=bias.to(dev()), res=r, act=ops.ACT_RES_RELU)
    report("gemm relu(bias+res) %dx%dx%d" % (M, N, K), C, torch.relu(A @ B.t() + bias + res), 1e-3, 1e-2)
    ops.gemm_nt(a, b, C, res=r, act=ops.ACT_RELU_MASK, aux=x)
    report("gemm (acc+res)*(aux>0)", C, (A @ B.t() + res) * (aux > 0), 1e-3, 1e-2)
    ops.gemm_nt(a, b, C, act=ops.ACT_RELU_MASK, aux=x)
    report("gemm acc*(aux>0)", C, (A @ B.t()) * (aux > 0), 1e-3, 1e-2)


@pytest.mark.parametrize("k,stride,pad,dil", [(3, 1, 1, 1), (3, 1, 2, 2), (3, 2, 1, 1), (1, 1, 0, 1)])
def test_im2col_nhwc(ops, k, stride, pad, dil):
    N, C, H, W = 2, 16, 9, 13
    x = rnd(N, C, H, W, seed=6)
    OH, OW = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    col = torch.full((N * OH * OW, k * k * C + 8), 7.0, dtype=torch.bfloat16, device=dev())
    ops.im2col_nhwc(to_gpu_bf16(nhwc(x)), col[:, :k * k * C], N, H, W, C, k, stride, pad, dil)
    ref = F.unfold(x, k, dilation=dil, padding=pad, stride=stride)            # [N, C*k*k, L], channel-major
    ref = ref.view(N, C, k * k, OH * OW).permute(0, 3, 2, 1).reshape(N * OH * OW, k * k * C)
    report("im2col k%d s%d p%d d%d" % (k, stride, pad, dil), col[:, :k * k * C], ref, 0, 0)
    assert float((col[:, k * k * C:].float() - 7.0).abs().max()) == 0.0


def test_im2col_image_and_stem_conv(ops):
    N, H, W = 2, 37, 50
    img = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(7)) * 50
    OH, OW = ops.conv_out_size(H, 7, 2, 3, 1), ops.conv_out_size(W, 7, 2, 3, 1)
    col = torch.zeros((N * OH * OW, 192), dtype=torch.bfloat16, device=dev())
    ops.im2col_image(img.to(dev()), col)
    ref = F.unfold(img, 7, padding=3, stride=2).view(N, 3, 49, OH * OW).permute(0, 3, 2, 1).reshape(N * OH * OW, 147)
    report("im2col image", col[:, :147], bf(ref), 0, 0)
    assert float(col[:, 147:].float().abs().max()) == 0.0
    # stem conv through the prepared weight
    w = torch.randn(64, 3, 7, 7, generator=torch.Generator().manual_seed(8)) * 0.05
    bn = [0.5 + torch.rand(64), torch.randn(64) * 0.1, torch.randn(64) * 0.1, 0.5 + torch.rand(64)]
    wf = torch.zeros((64, 192), dtype=torch.bfloat16, device=dev())
    scale, shift = torch.zeros(64, device=dev()), torch.zeros(64, device=dev())
    ops.conv_weight_prepare(w.permute(0, 2, 3, 1).reshape(64, 49, 3).contiguous().to(dev()), [t.to(dev()) for t in bn], wf, None, scale, shift)
    y = torch.zeros((N * OH * OW, 64), dtype=torch.bfloat16, device=dev())
    ops.gemm_nt(col, wf, y, bias=shift, act=ops.ACT_RELU)
    ref = torch.relu(F.batch_norm(F.conv2d(bf(img), w, stride=2, padding=3), bn[2], bn[3], bn[0], bn[1], False, eps=1e-5))
    report("stem conv+bn+relu", y, nhwc(ref), 1e-2, 1e-2)


def test_conv_weight_prepare_and_finalize(ops):
    O, I, k = 24, 16, 3
    g = torch.Generator().manual_seed(9)
    w = torch.randn(O, I, k, k, generator=g)
    bn = [0.5 + torch.rand(O, generator=g), torch.randn(O, generator=g), torch.randn(O, generator=g), 0.5 + torch.rand(O, generator=g)]
    w_ohwi = w.permute(0, 2, 3, 1).reshape(O, k * k, I).contiguous()
    wf = torch.zeros((O, k * k * I), dtype=torch.bfloat16, device=dev())
    wb = torch.zeros((I, k * k * O), dtype=torch.bfloat16, device=dev())
    scale, shift = torch.zeros(O, device=dev()), torch.zeros(O, device=dev())
    ops.conv_weight_prepare(w_ohwi.to(dev()), [t.to(dev()) for t in bn], wf, wb, scale, shift)
    s = bn[0] / torch.sqrt(bn[3] + 1e-5)
    report("bn scale", scale, s, 1e-6, 1e-6)
    report("bn shift", shift, bn[1] - bn[2] * s, 1e-6, 1e-6)
    folded = w_ohwi * s[:, None, None]
    report("wf", wf, folded.reshape(O, -1), 0, 4e-3)
    ref_wb = folded.flip(1).permute(2, 1, 0).reshape(I, k * k * O)             # [i, mirrored tap, o]
    report("wb", wb, ref_wb, 0, 4e-3)
    dwf = torch.randn(O, k * k * I + 8, generator=g)
    gacc = torch.randn(O, k * k * I, generator=g)
    gg = gacc.clone().to(dev())
    ops.conv_wgrad_finalize(dwf.to(dev()), scale, gg, accumulate=True)
    report("wgrad finalize", gg, gacc + dwf[:, :k * k * I] * s[:, None], 1e-6, 1e-6)


def test_pool_and_resample(ops):
    N, C, H, W = 2, 24, 11, 14
    x = rnd(N, C, H, W, seed=10)
    xg = to_gpu_bf16(nhwc(x))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.zeros((N * OH * OW, C), dtype=torch.bfloat16, device=dev())
    ops.maxpool3x3s2_nhwc(xg, y, N, H, W, C)
    report("maxpool", y, nhwc(F.max_pool2d(x, 3, 2, 1)), 0, 0)
    ops.subsample2_nhwc(xg, y, N, H, W, C)
    report("subsample2", y, nhwc(x[:, :, ::2, ::2]), 0, 0)
    dx = torch.full((N * H * W, C), 3.0, dtype=torch.bfloat16, device=dev())
    ops.upsample2_zero_nhwc(y, dx, N, H, W, C)
    ref = torch.zeros_like(x)
    ref[:, :, ::2, ::2] = x[:, :, ::2, ::2]
    report("upsample2_zero", dx, nhwc(ref), 0, 0)


@pytest.mark.parametrize("sr,C,H,W", [(1, 16, 9, 12), (2, 16, 9, 12), (1, 64, 9, 12), (2, 64, 9, 12), (1, 32, 24, 30), (0, 32, 24, 30)])
def test_roi_align_nhwc(ops, sr, C, H, W):
    """C % 32 == 0 takes the LDS-window backward (whole-map boxes on the 24x30 map exceed the LDS budget -> direct atomics);
    other widths the wave-per-bin kernel.  sr = 0: adaptive sampling grid."""
    N, R = 2, 3
    feat = rnd(N, C, H, W, seed=11)
    X, Y = 16.0 * W - 1, 16.0 * H - 1
    boxes = torch.tensor([[[10.0, 20.0, 150.0, 120.0, 9.0], [0.0, 0.0, X, Y, 9.0], [60.5, 30.25, 70.0, 35.0, 9.0]],
                          [[100.0, 8.0, 180.0, 140.0, 9.0], [-20.0, -10.0, 40.0, 50.0, 9.0], [-2.0, -2.0, -2.0, -2.0, 9.0]]])
    mask = boxes[:, :, 0] > -1.5
    rois = torch.cat((mask.nonzero()[:, :1].float(), boxes[mask][:, :4]), 1).numpy()
    ph = 4
    out = torch.zeros((N * R * ph * ph, C), dtype=torch.bfloat16, device=dev())
    bx = boxes.view(N * R, 5).contiguous().to(dev())
    ops.roi_align_nhwc_fwd(to_gpu_bf16(nhwc(feat)), bx, R, out, N, H, W, C, pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=sr)
    ref = torch.from_numpy(RA.roi_align_forward(feat.numpy(), rois, 1.0 / 16, ph, ph, sr).astype(np.float32))   # [K,C,ph,pw]
    got = out.float().cpu().view(N * R, ph, ph, C).permute(0, 3, 1, 2)
    report("roi_align nhwc fwd sr%d C%d %dx%d" % (sr, C, H, W), got[mask.view(-1)], ref, 1e-3, 1e-2)
    assert float(got[~mask.view(-1)].abs().max()) == 0.0
    dout = rnd(N * R, C, ph, ph, seed=12)
    dfeat = torch.full((N * H * W, C), 5.0, device=dev())
    ops.roi_align_nhwc_bwd(to_gpu_bf16(dout.permute(0, 2, 3, 1).reshape(-1, C)), bx, R, dfeat, N, H, W, C, pooled=ph,
                           spatial_scale=1.0 / 16, sampling_ratio=sr)
    refb = torch.from_numpy(RA.roi_align_backward(dout[mask.view(-1)].numpy(), rois, 1.0 / 16, ph, ph, N, C, H, W, sr).astype(np.float32))
    report("roi_align nhwc bwd sr%d C%d %dx%d" % (sr, C, H, W), dfeat, nhwc(refb), 1e-4, 1e-4)
    # the gather form (per-RoI separable weight tables, one owner per output element): fp32 output, and the bf16 output with the
    # ReLU mask of the producing activation folded in (what VisionStack consumes)
    ws = ops.roi_align_gather_workspace(N * R, H, W, ph, dev())
    dout_g = to_gpu_bf16(dout.permute(0, 2, 3, 1).reshape(-1, C))
    g32 = torch.full((N * H * W, C), 7.0, device=dev())
    gbf = torch.full((N * H * W, C), 7.0, dtype=torch.bfloat16, device=dev())
    act = rnd(N * H * W, C, seed=13)
    ops.roi_align_nhwc_bwd_gather(dout_g, bx, R, ws, N, H, W, C, act=to_gpu_bf16(act), dx_bf16=gbf, dx_f32=g32, pooled=ph,
                                  spatial_scale=1.0 / 16, sampling_ratio=sr)
    report("roi_align nhwc bwd gather sr%d C%d %dx%d" % (sr, C, H, W), g32, nhwc(refb), 1e-4, 1e-4)
    report("roi_align nhwc bwd gather bf16 + relu mask", gbf, nhwc(refb) * (bf(act) > 0), 1e-4, 4e-3)


def test_roi_align_nhwc_backward_gather_matches_scatter_many_boxes(ops):
    """More than 64 box slots per image (the gather kernel walks them in chunks of a wave), ragged padding, C not a multiple of 256."""
    N, R, C, H, W, ph = 2, 70, 40, 12, 16, 7
    g = torch.Generator().manual_seed(5)
    x1 = torch.rand(N, R, generator=g) * (16 * W - 40) - 10
    y1 = torch.rand(N, R, generator=g) * (16 * H - 40) - 10
    boxes = torch.stack((x1, y1, x1 + 4 + torch.rand(N, R, generator=g) * 150, y1 + 4 + torch.rand(N, R, generator=g) * 120,
                         torch.zeros(N, R)), -1)
    boxes[0, 50:] = -2.0
    boxes[1, 67:] = -2.0
    bx = boxes.view(N * R, 5).contiguous().to(dev())
    dout = to_gpu_bf16(torch.randn(N * R * ph * ph, C, generator=g))
    for sr in (1, 2, 0):
        ref = torch.zeros((N * H * W, C), device=dev())
        ops.roi_align_nhwc_bwd(dout, bx, R, ref, N, H, W, C, pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=sr)
        got = torch.full((N * H * W, C), 3.0, device=dev())
        ops.roi_align_nhwc_bwd_gather(dout, bx, R, ops.roi_align_gather_workspace(N * R, H, W, ph, dev()), N, H, W, C, dx_f32=got,
                                      pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=sr)
        report("roi_align bwd gather vs atomic scatter, 70 slots, sr %d" % sr, got, ref.cpu(), 1e-4, 1e-4)


@pytest.mark.parametrize("form", ["atomic", "gather"])
@pytest.mark.parametrize("C,H,W", [(32, 9, 12), (64, 38, 63)])
def test_roi_align_nhwc_backward_is_the_jacobian_transpose_of_the_hip_forward(ops, C, H, W, form):
    """The reference has no CPU ROIAlign backward (ROIAlign.h:44), so the backward oracle is only pinned through the adjoint of its own
    forward.  This test closes the loop on the DEVICE kernels themselves: ROIAlign is linear in the feature map, so the forward of a
    one-hot map IS a Jacobian column (exact finite difference), and the LDS-window backward (C % 32 == 0; 14x14 bins, the pre-training
    geometry at 38x63) must return <dout, that column> at the hot element -- for hot elements inside, at the border of and outside the
    RoI windows -- and the full adjoint identity <dout, fwd(x)> = <bwd(dout), x> must hold for a random map."""
    N, R, ph = 2, 3, 14
    X, Y = 16.0 * W - 1, 16.0 * H - 1
    boxes = torch.tensor([[[10.0, 20.0, 0.45 * X, 0.8 * Y, 9.0], [0.0, 0.0, X, Y, 9.0], [60.5, 30.25, 70.0, 35.0, 9.0]],
                          [[0.3 * X, 8.0, 0.9 * X, 0.9 * Y, 9.0], [-20.0, -10.0, 40.0, 50.0, 9.0], [-2.0, -2.0, -2.0, -2.0, 9.0]]])
    bx = boxes.view(N * R, 5).contiguous().to(dev())
    g = torch.Generator().manual_seed(21)
    dout = torch.randn(N * R * ph * ph, C, generator=g)
    dout_g = to_gpu_bf16(dout)
    dout_r = dout_g.float().cpu()
    dfeat = torch.zeros((N * H * W, C), device=dev())
    if form == "atomic":
        ops.roi_align_nhwc_bwd(dout_g, bx, R, dfeat, N, H, W, C, pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=1)
    else:
        ops.roi_align_nhwc_bwd_gather(dout_g, bx, R, ops.roi_align_gather_workspace(N * R, H, W, ph, dev()), N, H, W, C, dx_f32=dfeat,
                                      pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=1)
    torch.cuda.synchronize()
    dfeat_c = dfeat.cpu()
    out = torch.zeros((N * R * ph * ph, C), dtype=torch.bfloat16, device=dev())
    hot = [(0, 1, 1, 3), (0, H // 2, W // 3, 0), (0, H - 1, W - 1, C - 1), (1, 2, W // 2, 7), (1, H - 2, 1, C // 2), (1, 0, 0, 1),
           (0, 2, 4, 5), (1, H // 2, W - 2, 9)]
    worst = 0.0
    for n, y, x, c in hot:
        feat = torch.zeros((N * H * W, C))
        feat[(n * H + y) * W + x, c] = 1.0
        ops.roi_align_nhwc_fwd(to_gpu_bf16(feat), bx, R, out, N, H, W, C, pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=1)
        col = out.float().cpu()                                    # Jacobian column (bilinear weights / samples per bin, bf16)
        want = float((col * dout_r).sum())
        got = float(dfeat_c[(n * H + y) * W + x, c])
        scale = float((col.abs() * dout_r.abs()).sum()) + 1e-6
        worst = max(worst, abs(got - want) / scale)
        assert abs(got - want) <= 1e-2 * scale, (n, y, x, c, got, want)
    feat = torch.randn(N * H * W, C, generator=g)
    feat_g = to_gpu_bf16(feat)
    ops.roi_align_nhwc_fwd(feat_g, bx, R, out, N, H, W, C, pooled=ph, spatial_scale=1.0 / 16, sampling_ratio=1)
    lhs = float((out.float().cpu().double() * dout_r.double()).sum())
    rhs = float((dfeat_c.double() * feat_g.float().cpu().double()).sum())
    print("roi_align nhwc %dx%dx%d: one-hot Jacobian columns worst rel %.2e; adjoint identity %.6f vs %.6f" % (H, W, C, worst, lhs, rhs))
    # the forward output is rounded to bf16 (2^-9 relative per element): the identity holds up to that rounding noise on its terms
    noise = float(((out.float().cpu().double() * dout_r.double()) ** 2).sum().sqrt()) * 2.0 ** -8
    assert abs(lhs - rhs) <= 4.0 * noise + 1e-4 * abs(lhs), (lhs, rhs, noise)


def test_avgpool_rows_and_relu_mask(ops):
    K, P, C, ld = 5, 9, 32, 4 + 32
    y = torch.relu(rnd(K * P, C, seed=13))
    boxes = torch.full((K, ld), -1.0)
    boxes[:, 0] = torch.tensor([1.0, 2.0, -2.0, 0.0, 5.0])
    bx = boxes.to(dev())
    ops.avgpool_rows_fwd(to_gpu_bf16(y), bx, 4, K, P, C, pad_col=0)
    ref = y.view(K, P, C).mean(1)
    ref[boxes[:, 0] <= -1.5] = 0
    report("avgpool fwd", bx[:, 4:], ref, 1e-6, 1e-5)
    assert torch.equal(bx[:, :4].cpu(), boxes[:, :4])
    d = rnd(K, C, seed=14)
    seed, tag, p = 4321, 77, 0.25
    seed_t = torch.tensor([seed], dtype=torch.int32, device=dev())
    dz = torch.zeros((K * P, C), dtype=torch.bfloat16, device=dev())
    ops.avgpool_rows_bwd(to_gpu_bf16(d), to_gpu_bf16(y), bx, dz, K, P, C, drop_p=p, seed=seed_t, tag=tag, drop_row_elems=2 * C, drop_col0=C)
    idx = (np.arange(K)[:, None] * 2 * C + C + np.arange(C)[None, :])
    keep = torch.from_numpy(keep_mask(seed, tag, idx, drop_thr(p)))
    g = d * keep * drop_scale(drop_thr(p)) / P
    g[boxes[:, 0] <= -1.5] = 0
    ref = g[:, None, :].expand(K, P, C).reshape(K * P, C) * (y > 0)
    report("avgpool bwd (+dropout mask, padded boxes)", dz, ref, 1e-6, 1e-2)
    g32 = torch.randn(K * P, C)
    out = torch.zeros((K * P, C), dtype=torch.bfloat16, device=dev())
    ops.relu_mask_cast(g32.to(dev()), to_gpu_bf16(y), out)
    report("relu_mask_cast", out, g32 * (y > 0), 0, 4e-3)


@pytest.mark.parametrize("N,I,O,H,W,dil", [(3, 128, 256, 38, 63, 1), (40, 512, 512, 14, 14, 2)])
def test_conv3x3_implicit_equals_explicit_at_size(ops, N, I, O, H, W, dil):
    """layer2 / RoI-head sized convolutions (multiple tiles, persistent workgroups, tile tails): bit-identical to im2col + GEMM."""
    M = N * H * W
    x, w = to_gpu_bf16(rnd(M, I, seed=40)), to_gpu_bf16(rnd(O, 9 * I, seed=41, scale=0.05))
    bias = torch.randn(O, generator=torch.Generator().manual_seed(42)).to(dev())
    zero16 = torch.zeros(64, dtype=torch.bfloat16, device=dev())
    col = torch.zeros((M, 9 * I), dtype=torch.bfloat16, device=dev())
    y1 = torch.zeros((M, O), dtype=torch.bfloat16, device=dev())
    y2 = torch.zeros_like(y1)
    ops.im2col_nhwc(x, col, N, H, W, I, 3, 1, dil, dil)
    ops.gemm_nt(col, w, y1, bias=bias, act=ops.ACT_RELU)
    ops.conv3x3_nhwc(x, w, y2, N, H, W, I, dil, zero16, bias=bias, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2), "max diff %g" % float((y1.float() - y2.float()).abs().max())
    # weight gradient: TN GEMM over the im2col image vs the gather inside the TN kernel (same split-K plan -> same bits)
    dy = to_gpu_bf16(rnd(M, O, seed=43))
    ws = torch.zeros(max(ops.wgrad_workspace_floats(O, 9 * I, M), 4), device=dev())
    g1 = torch.zeros((O, 9 * I), device=dev())
    g2 = torch.full((O, 9 * I), 3.0, device=dev())
    ops.wgrad_tn(dy, col, g1, workspace=ws, accumulate=False)
    ops.conv3x3_wgrad_tn(dy, x, g2, N, H, W, I, dil, workspace=ws, accumulate=False)
    torch.cuda.synchronize()
    assert torch.equal(g1, g2), "wgrad max diff %g (scale %g)" % (float((g1 - g2).abs().max()), float(g1.abs().max()))
    ops.conv3x3_wgrad_tn(dy, x, g2, N, H, W, I, dil, workspace=None, accumulate=True)      # single pass, accumulate
    report("conv3x3 implicit wgrad, accumulate, no workspace", g2, 2 * g1, 1e-3, 1e-4)


@pytest.mark.parametrize("dil", [1, 2])
def test_conv3x3_forward_dgrad_wgrad_vs_autograd(ops, dil):
    """One folded-BN 3x3 convolution end to end: im2col + NT GEMM forward, mirrored-tap dgrad, TN wgrad + finalize,
    against F.conv2d / batch_norm autograd on the same bf16-rounded operands."""
    N, I, O, H, W = 2, 64, 128, 14, 11
    g = torch.Generator().manual_seed(30 + dil)
    x = rnd(N, I, H, W, seed=31)
    w = torch.randn(O, I, 3, 3, generator=g) * 0.05
    bn = [0.5 + torch.rand(O, generator=g), torch.randn(O, generator=g) * 0.1, torch.randn(O, generator=g) * 0.1, 0.5 + torch.rand(O, generator=g)]
    dy = rnd(N, O, H, W, seed=32)
    M = N * H * W
    wf = torch.zeros((O, 9 * I), dtype=torch.bfloat16, device=dev())
    wb = torch.zeros((I, 9 * O), dtype=torch.bfloat16, device=dev())
    scale, shift = torch.zeros(O, device=dev()), torch.zeros(O, device=dev())
    w_ohwi = w.permute(0, 2, 3, 1).reshape(O, 9, I).contiguous().to(dev())
    ops.conv_weight_prepare(w_ohwi, [t.to(dev()) for t in bn], wf, wb, scale, shift)
    xg, dyg = to_gpu_bf16(nhwc(x)), to_gpu_bf16(nhwc(dy))
    col = torch.zeros((M, 9 * I), dtype=torch.bfloat16, device=dev())
    y = torch.zeros((M, O), dtype=torch.bfloat16, device=dev())
    ops.im2col_nhwc(xg, col, N, H, W, I, 3, 1, dil, dil)
    ops.gemm_nt(col, wf, y, bias=shift)
    # reference with the SAME folded bf16 weight the device uses (isolates the kernels from the folding rounding)
    wq = wf.float().cpu().view(O, 3, 3, I).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wq, padding=dil, dilation=dil) + shift.cpu().view(1, O, 1, 1)
    report("conv3x3 d%d forward" % dil, y, nhwc(yr), 1e-3, 1e-2)
    # implicit GEMM (gather in the LDS-DMA address generator): same numbers without the im2col image
    zero16 = torch.zeros(64, dtype=torch.bfloat16, device=dev())
    y2 = torch.zeros_like(y)
    ops.conv3x3_nhwc(xg, wf, y2, N, H, W, I, dil, zero16, bias=shift)
    assert torch.equal(y2, y), "implicit and explicit forward differ"
    ops.conv3x3_nhwc(xg, wf, y2, N, H, W, I, dil, zero16, bias=shift, act=ops.ACT_RELU)
    report("conv3x3 d%d implicit forward + relu" % dil, y2, torch.relu(nhwc(yr)), 1e-3, 1e-2)
    (yr * dy).sum().backward()
    dcol = torch.zeros((M, 9 * O), dtype=torch.bfloat16, device=dev())
    dx = torch.zeros((M, I), dtype=torch.bfloat16, device=dev())
    ops.im2col_nhwc(dyg, dcol, N, H, W, O, 3, 1, dil, dil)
    ops.gemm_nt(dcol, wb, dx)
    report("conv3x3 d%d dgrad (mirrored taps)" % dil, dx, nhwc(xr.grad), 1e-3, 1e-2)
    dx2 = torch.zeros_like(dx)
    ops.conv3x3_nhwc(dyg, wb, dx2, N, H, W, O, dil, zero16)
    assert torch.equal(dx2, dx), "implicit and explicit dgrad differ"
    aux = rnd(M, I, seed=33)
    ops.conv3x3_nhwc(dyg, wb, dx2, N, H, W, O, dil, zero16, act=ops.ACT_RELU_MASK, aux=to_gpu_bf16(aux))
    report("conv3x3 d%d implicit dgrad x (aux>0)" % dil, dx2, nhwc(xr.grad) * (aux > 0), 1e-3, 1e-2)
    dwf = torch.zeros((O, 9 * I), device=dev())
    ws = torch.zeros(max(ops.wgrad_workspace_floats(O, 9 * I, M), 4), device=dev())
    ops.wgrad_tn(dyg, col, dwf, workspace=ws, accumulate=False)
    report("conv3x3 d%d wgrad (folded)" % dil, dwf.view(O, 3, 3, I).permute(0, 3, 1, 2), wq.grad, 1e-3, 2e-3)
    g32 = torch.zeros((O, 9 * I), device=dev())
    ops.conv_wgrad_finalize(dwf, scale, g32, accumulate=True)
    report("conv3x3 d%d wgrad of the master weight" % dil, g32.view(O, 3, 3, I).permute(0, 3, 1, 2), wq.grad * scale.cpu().view(O, 1, 1, 1), 1e-3, 2e-3)


# ------------------------------------------------------------------------------------------------------------------
def _vision_fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vision", "vision_small.npz"), allow_pickle=False)
    nl = int(z["num_layers"])
    P = VO.init_vision_params(int(z["seed"]), nl)
    return z, nl, P


def _prefixed(P):
    return {"image_feature_extractor." + k: v for k, v in VO.split_state_dict(P).items()}


def test_vision_stack_matches_reference_golden():
    """VisionStack forward + backward against the fixture of the reference's FastRCNN e2e module and the oracle's gradients."""
    V = pkg("vision")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"])
    N, R = boxes4.shape[:2]
    vs = V.VisionStack(N, img.shape[2], img.shape[3], R, device=dev(), num_layers=nl)
    vs.load_state_dict({k: v.to(dev()) for k, v in _prefixed(P).items()})
    boxes = torch.zeros((N, R, 4 + 2048), device=dev())
    boxes[:, :, :4] = boxes4.to(dev())
    vs.forward(img.to(dev()), boxes)
    torch.cuda.synchronize()
    raw = torch.from_numpy(z["obj_reps_raw"])
    report("e2e post_roialign vs reference", boxes[:, :, 4:], raw, 2e-2, 2e-2)
    mask = boxes4[:, :, 0] > -1.5
    assert float(boxes[:, :, 4:].cpu()[~mask].abs().max()) == 0.0
    body4 = vs.body4.float().cpu().view(N, vs.H3, vs.W3, -1).permute(0, 3, 1, 2).reshape(-1)[::97][:512]
    report("body4 samples vs reference", body4, torch.from_numpy(z["body4_sample"]), 2e-2, 2e-2)
    # backward of <post_roialign, Wr>
    Wr = torch.from_numpy(z["Wr"])
    vs.zero_grad()
    vs.backward(to_gpu_bf16(Wr.view(N * R, -1)), boxes)
    torch.cuda.synchronize()
    got = vs.grads()
    want_norm = dict(zip([str(k) for k in z["grad_names"]], z["grad_norms"]))
    frozen = VO.frozen_names(P)
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    feats, _ = VO.e2e_features(img, boxes4, Po, nl)
    (feats * Wr[mask]).sum().backward()
    names = dict(zip(VO.split_state_dict(P).keys(), P.keys()))
    worst = (0.0, "")
    errs, by_layer = [], {}
    for name, g in got.items():
        short = name[len("image_feature_extractor."):]
        ref = Po[names[short]].grad
        e = rel_fro(g, ref)
        errs.append(e)
        by_layer.setdefault(short.rsplit(".", 3)[0] if short.startswith("backbone") else "roi_head." + short.split(".")[1], []).append(e)
        worst = max(worst, (e, short))
        assert abs(float(ref.double().norm()) - want_norm[short]) <= 1e-3 * want_norm[short]      # oracle == reference (pinned on CPU too)
    print("vision stack: %d weight gradients, median rel-fro %.3e, worst %.3e (%s)" % (len(errs), float(np.median(errs)), worst[0], worst[1]))
    for k in by_layer:       # the error grows with the number of bf16 Bottlenecks the gradient has crossed (head -> layer2)
        print("   %-22s max rel-fro %.3e" % (k, max(by_layer[k])))
    assert set(n[len("image_feature_extractor."):] for n in got) == set(want_norm)
    # measured on MI355X (bf16 build): median 3.9e-2, worst 8.1e-2 (layer2.0.conv1: the gradient that crossed the most bf16 Bottlenecks) --
    # bounds = measured + ~25 %; the global gradient norm is within 2e-3 (asserted by the engine-level e2e tests)
    assert np.median(errs) < 5e-2 and worst[0] < 0.10, worst
    # VCR call form: object masks inside the RoI head -- forward against the reference fixture, masked-pool backward against the oracle
    segms = torch.from_numpy(z["segms"])
    vs.forward(img.to(dev()), boxes, segms.to(dev()))
    report("e2e post_roialign with segms vs reference", boxes[:, :, 4:], torch.from_numpy(z["obj_reps_raw_segms"]), 2e-2, 2e-2)
    vs.zero_grad()
    vs.backward(to_gpu_bf16(Wr.view(N * R, -1)), boxes)
    torch.cuda.synchronize()
    Ps = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    feats_s, _ = VO.e2e_features(img, boxes4, Ps, nl, segms=segms)
    (feats_s * Wr[mask]).sum().backward()
    gs = vs.grads()
    for short, ref_name in (("roi_head_feature_extractor.2.conv3.weight", "layer4.2.conv3.weight"),
                            ("roi_head_feature_extractor.0.conv1.weight", "layer4.0.conv1.weight"),
                            ("backbone.layer3.5.conv2.weight", "layer3.5.conv2.weight")):
        e = rel_fro(gs["image_feature_extractor." + short], Ps[ref_name].grad)
        print("   with segms: d %s rel-fro %.3e" % (short, e))
        assert e < (3e-2 if "roi_head" in short else 8e-2), short


def _vision_grad_errors(eng, Po, names):
    """per-tensor rel-Frobenius errors of the trainable convolution gradients, the rel-Frobenius error of their concatenation and
    the relative difference of the global gradient norm over the vision parameters."""
    errs, num, gn, rn = [], 0.0, 0.0, 0.0
    for name, g in eng.vision.grads().items():
        ref = Po[names[name[len("image_feature_extractor."):]]].grad
        errs.append((rel_fro(g, ref), name))
        a, b = g.double().cpu().reshape(-1), ref.double().cpu().reshape(-1)
        num += float((a - b).pow(2).sum())
        gn += float(a.pow(2).sum())
        rn += float(b.pow(2).sum())
    return errs, (num / rn) ** 0.5, abs(gn ** 0.5 - rn ** 0.5) / rn ** 0.5


@pytest.mark.parametrize("empty_sample", [False, True])
def test_engine_e2e_step_vs_oracle(empty_sample):
    """One e2e pretraining step (image -> CNN -> VL-BERT -> losses -> gradients) of the engine against the composed oracle.
    empty_sample: the second image has no valid box at all (every RoI slot padded)."""
    E, syn = pkg("engine"), pkg("synthetic")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"]).clone()
    if empty_sample:
        boxes4[1] = -2.0
    B, R = boxes4.shape[:2]
    T = 12
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=21)
    batch = list(syn.make_batch(B, T, R, seed=22, ragged=False))
    batch[0] = torch.cat((boxes4, torch.zeros(B, R, 2048)), -1)          # boxes of the fixture (one padded), features unused
    batch[1] = torch.from_numpy(z["im_info"])
    pad = boxes4[:, :, 0] <= -1.5
    batch[5][pad] = 0                                                     # mvrc_ops / labels of the padded box
    batch[6][pad] = 0
    mc = E.ModelConfig(num_hidden_layers=1, e2e=True, image_num_layers=nl)
    eng = E.PretrainEngine(mc, B, T, R, device="cuda:0", train=False, keep_logits=True, image_size=tuple(img.shape[2:]))
    sd = {k: v.to(dev()) for k, v in params.items()}
    sd.update({k: v.to(dev()) for k, v in _prefixed(P).items()})
    eng.load_state_dict(sd)
    eng.set_batch(*[t.to(dev()) for t in batch], image=img.to(dev()))
    eng.zero_grad()
    eng.forward(False)
    eng.backward(False)
    torch.cuda.synchronize()
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    out, loss = O.pretrain_forward(leaves, cfg, *batch, train=False, image=img, vision_params=Po, image_num_layers=nl)
    loss.backward()
    lv = eng.loss_values()
    print("e2e losses: hip mlm %.5f mvrc %.5f | oracle mlm %.5f mvrc %.5f" % (lv["mlm_loss"], lv["mvrc_loss"], float(out["mlm_loss"].detach()),
                                                                              float(out["mvrc_loss"].detach())))
    assert abs(lv["mlm_loss"] - float(out["mlm_loss"])) < 2e-2 * max(1.0, abs(float(out["mlm_loss"])))
    assert abs(lv["mvrc_loss"] - float(out["mvrc_loss"])) < 2e-2 * max(1.0, abs(float(out["mvrc_loss"].detach())))
    names = dict(zip(VO.split_state_dict(P).keys(), P.keys()))
    errs, glob, dnorm = _vision_grad_errors(eng, Po, names)
    print("e2e engine: conv weight gradients median rel-fro %.3e worst %.3e (%s); all vision parameters: rel-fro %.3e, grad-norm "
          "difference %.3e" % (float(np.median([e for e, _ in errs])), *max(errs), glob, dnorm))
    assert np.median([e for e, _ in errs]) < 5.5e-2 and max(errs)[0] < 0.11      # measured: median 3.0-4.5e-2, worst 6.3-9.0e-2 (+ ~20 %)
    assert dnorm < 1e-2, dnorm          # global gradient norm over the vision parameters (what the clip and the step size see)
    assert leaves["object_mask_visual_embedding.weight"].grad is None or float(leaves["object_mask_visual_embedding.weight"].grad.abs().sum()) == 0.0
    assert float(eng.g32["object_mask_visual_embedding.weight"].abs().sum()) == 0.0
    for k in ("image_feature_extractor.obj_downsample.1.weight", "vlbert.encoder.layer.0.output.dense.weight"):
        e = rel_fro(eng.g32[k], leaves[k].grad)
        print("  %s rel-fro %.3e" % (k, e))
        assert e < 5e-2, k
    # one optimizer step moves the trainable convolutions and leaves the frozen stages / BatchNorm alone
    before = eng.vision.state_dict()
    eng.optimizer_step()
    torch.cuda.synchronize()
    after = eng.vision.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    trainable = set(eng.vision.grads())
    assert set(moved) == trainable, (set(moved) ^ trainable)


def test_raw_pixel_masking_of_masked_regions_vs_oracle(ops):
    """vlb_mask_image_boxes_f32 = the dataset's MASK_RAW_PIXELS step (pretrain/data/datasets/conceptual_captions.py:201-206) on a
    collated batch: bit-exact against oracle.vision_oracle.mask_raw_pixels -- fractional corners (int() truncation), boxes touching
    and leaving the image, degenerate and padded (-2) boxes, ops other than 1 -- and through engine.set_batch(mask_raw_pixels=True),
    whose e2e step then matches the oracle fed the masked image."""
    g = torch.Generator().manual_seed(5)
    B, R, H, W = 3, 7, 37, 53
    img = torch.randn(B, 3, H, W, generator=g) * 40 + 3
    boxes = torch.zeros(B, R, 6)
    x1 = torch.rand(B, R, generator=g) * (W - 10)
    y1 = torch.rand(B, R, generator=g) * (H - 10)
    boxes[..., 0], boxes[..., 1] = x1, y1
    boxes[..., 2] = x1 + torch.rand(B, R, generator=g) * 20
    boxes[..., 3] = y1 + torch.rand(B, R, generator=g) * 20
    boxes[0, 0, :4] = torch.tensor([0.0, 0.0, W - 1.0, H - 1.0])          # the whole image
    boxes[0, 1, :4] = torch.tensor([W - 3.5, H - 2.2, W + 9.0, H + 4.0])    # leaves the image
    boxes[1, 0, :4] = torch.tensor([4.9, 5.9, 4.1, 5.2])                    # x2 < x1 after int(): still one pixel column (stop = int()+1)
    boxes[1, 1, :4] = torch.tensor([-2.0, -2.0, -2.0, -2.0])                # a padded slot that (wrongly) carries op 1: python slice semantics
    boxes[2, 2, :4] = torch.tensor([10.0, 8.0, 10.0, 8.0])                  # one pixel
    opsm = (torch.rand(B, R, generator=g) < 0.5).long()
    opsm[0, 0], opsm[0, 1], opsm[1, 0], opsm[1, 1], opsm[2, 2], opsm[2, 3] = 0, 1, 1, 1, 1, 2
    ref = VO.mask_raw_pixels(img.clone(), boxes, opsm)
    got = img.clone().to(dev())
    ops.mask_image_boxes(got, boxes.to(dev()), opsm.to(dev()))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref), "masked image differs from conceptual_captions.py:201-206"
    assert int((ref == 0).sum()) > 0 and not torch.equal(ref, img)
    opsm[0, 0] = 1                                                           # the whole image masked
    got = img.clone().to(dev())
    ops.mask_image_boxes(got, boxes.to(dev()), opsm.to(dev()))
    assert float(got[0].abs().sum()) == 0.0 and torch.equal(got.cpu(), VO.mask_raw_pixels(img.clone(), boxes, opsm))


def test_engine_e2e_step_with_raw_pixel_masking_vs_oracle():
    """The e2e step on a batch whose masked regions (mvrc_op 1) have their PIXELS zeroed by set_batch(mask_raw_pixels=True) -- what the
    reference trains on with MASK_RAW_PIXELS -- against the composed oracle on the image masked by the oracle's own statement."""
    E, syn = pkg("engine"), pkg("synthetic")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"]).clone()
    B, R = boxes4.shape[:2]
    T = 12
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=23)
    batch = list(syn.make_batch(B, T, R, seed=24, ragged=False))
    batch[0] = torch.cat((boxes4, torch.zeros(B, R, 2048)), -1)
    batch[1] = torch.from_numpy(z["im_info"])
    pad = boxes4[:, :, 0] <= -1.5
    batch[5][pad] = 0
    batch[6][pad] = 0
    assert int((batch[5] == 1).sum()) >= 2
    mc = E.ModelConfig(num_hidden_layers=1, e2e=True, image_num_layers=nl)
    eng = E.PretrainEngine(mc, B, T, R, device="cuda:0", train=False, keep_logits=True, image_size=tuple(img.shape[2:]))
    sd = {k: v.to(dev()) for k, v in params.items()}
    sd.update({k: v.to(dev()) for k, v in _prefixed(P).items()})
    eng.load_state_dict(sd)
    eng.set_batch(*[t.to(dev()) for t in batch], image=img.to(dev()), mask_raw_pixels=True)
    masked = VO.mask_raw_pixels(img.clone(), boxes4, batch[5])
    assert torch.equal(eng.in_image.cpu(), masked) and not torch.equal(masked, img)
    eng.zero_grad()
    eng.forward(False)
    eng.backward(False)
    torch.cuda.synchronize()
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    out, loss = O.pretrain_forward(leaves, cfg, *batch, train=False, image=masked, vision_params=Po, image_num_layers=nl)
    _, loss_unmasked = O.pretrain_forward(params, cfg, *batch, train=False, image=img, vision_params=P, image_num_layers=nl)
    lv = eng.loss_values()
    print("e2e masked pixels: hip mlm %.5f mvrc %.5f | oracle mlm %.5f mvrc %.5f (oracle loss on the unmasked image %.5f)" %
          (lv["mlm_loss"], lv["mvrc_loss"], float(out["mlm_loss"]), float(out["mvrc_loss"]), float(loss_unmasked)))
    assert abs(lv["mlm_loss"] - float(out["mlm_loss"])) < 2e-2 * max(1.0, abs(float(out["mlm_loss"])))
    assert abs(lv["mvrc_loss"] - float(out["mvrc_loss"])) < 2e-2 * max(1.0, abs(float(out["mvrc_loss"])))
    with pytest.raises(ValueError):
        pre = E.PretrainEngine(E.ModelConfig(num_hidden_layers=1), B, T, R, device="cuda:0", train=False)
        pre.set_batch(*[t.to(dev()) for t in batch], mask_raw_pixels=True)


def test_engine_multitask_e2e_step_vs_oracle():
    """multitask x e2e (cfgs/pretrain/base_e2e_16x16G_fp16.yaml: MODULE ResNetVLBERTForPretrainingMultitask with
    IMAGE_FEAT_PRECOMPUTED false): caption samples take their region features from the CNN, the text-only samples never touch it
    and see aux_text_visual_embedding; three losses (mlm_wvc, mlm_aux, mvrc) and all gradients against the composed oracle
    (resnet_vlbert_for_pretraining_multitask.py:96-290)."""
    E, syn = pkg("engine"), pkg("synthetic")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"]).clone()
    B, R = boxes4.shape[:2]
    T, Ba = 12, 3
    cfg = O.VLBertConfig(num_hidden_layers=1)
    cfg.multitask = True
    params = O.init_params(cfg, seed=31)
    batch = list(syn.make_batch(B, T, R, seed=32, ragged=False))
    batch[0] = torch.cat((boxes4, torch.zeros(B, R, 2048)), -1)
    batch[1] = torch.from_numpy(z["im_info"])
    pad = boxes4[:, :, 0] <= -1.5
    batch[5][pad] = 0
    batch[6][pad] = 0
    aux_text, aux_lab = syn.make_aux_text(Ba, T, seed=33)
    mc = E.ModelConfig(num_hidden_layers=1, e2e=True, multitask=True, image_num_layers=nl)
    eng = E.PretrainEngine(mc, B, T, R, device="cuda:0", train=False, keep_logits=True, image_size=tuple(img.shape[2:]), B_aux=Ba)
    sd = {k: v.to(dev()) for k, v in params.items()}
    sd.update({k: v.to(dev()) for k, v in _prefixed(P).items()})
    eng.load_state_dict(sd)
    eng.set_batch(*[t.to(dev()) for t in batch], aux_text=aux_text.to(dev()), aux_mlm_labels=aux_lab.to(dev()), image=img.to(dev()))
    eng.zero_grad()
    eng.forward(False)
    eng.backward(False)
    torch.cuda.synchronize()
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    out, loss = O.pretrain_multitask_forward(leaves, cfg, *batch, aux_text, aux_lab, train=False, image=img, vision_params=Po,
                                             image_num_layers=nl)
    loss.backward()
    lv = eng.loss_values()
    for k in ("mlm_loss_wvc", "mlm_loss_aux", "mvrc_loss"):
        print("multitask e2e %s: hip %.5f oracle %.5f" % (k, lv[k], float(out[k].detach())))
        assert abs(lv[k] - float(out[k])) < 2e-2 * max(1.0, abs(float(out[k].detach()))), k
    names = dict(zip(VO.split_state_dict(P).keys(), P.keys()))
    errs, glob, dnorm = _vision_grad_errors(eng, Po, names)
    print("multitask e2e: conv weight gradients median rel-fro %.3e worst %.3e (%s); all vision parameters: rel-fro %.3e, "
          "grad-norm difference %.3e" % (float(np.median([e for e, _ in errs])), *max(errs), glob, dnorm))
    assert np.median([e for e, _ in errs]) < 5.5e-2 and max(errs)[0] < 0.11      # measured: median 3.0-4.5e-2, worst 6.3-9.0e-2 (+ ~20 %)
    assert dnorm < 1e-2, dnorm
    for k in ("aux_text_visual_embedding.weight", "image_feature_extractor.obj_downsample.1.weight",
              "vlbert.encoder.layer.0.output.dense.weight", "vlbert.word_embeddings.weight"):
        e = rel_fro(eng.g32[k], leaves[k].grad)
        print("  %s rel-fro %.3e" % (k, e))
        assert e < 5e-2, k


def test_train_end2end_runs_the_shipped_e2e_multitask_config():
    """cfgs/pretrain/base_e2e_16x16G_fp16.yaml (BASELINE config 3) unmodified through the reference-style entry point: MODULE
    ResNetVLBERTForPretrainingMultitask, 8 image-caption samples (600x1000 images through ResNet-101 / ROIAlign / layer4) + 8
    text-only samples per GPU, triangle schedule with 16000 warm-up steps -- two optimizer steps on synthetic batches."""
    tr = pkg("pretrain.train_end2end")
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs", "pretrain", "base_e2e_16x16G_fp16.yaml")
    eng = tr.main(["--cfg", cfg, "--steps", "2"])
    torch.cuda.synchronize()
    lv = eng.loss_values()
    assert eng.B == 8 and eng.Ba == 8 and eng.vision is not None and eng.in_image.shape[2:] == (600, 1000)
    assert np.isfinite(lv["loss"]) and lv["mlm_loss_wvc"] > 0 and lv["mlm_loss_aux"] > 0 and lv["mvrc_loss"] > 0
    assert float(eng.adam[5]) == 2.0
    base = 1.0e-7 * 16                                           # TRAIN.LR x (8 + 8) x world 1
    assert abs(float(eng.adam[0]) - base * O.warmup_linear_lr(2, 16000, 100000)) < 1e-6 * base


def test_dropin_module_e2e_training_loop_contract():
    """The `ResNetVLBERTForPretraining` mirror in the e2e configuration (IMAGE_FEAT_PRECOMPUTED false): reference-layout
    checkpoint in, `outputs, loss = net(image, boxes, ...)`, `loss.backward()`, torch optimizer step, reference-layout checkpoint out."""
    from tests.test_engine_gpu import _module_config
    M = pkg("pretrain.modules.resnet_vlbert_for_pretraining")
    syn = pkg("synthetic")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"])
    B, R, T = boxes4.shape[0], boxes4.shape[1], 12
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=21)
    conf = _module_config(cfg)
    conf["NETWORK"].update(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_NUM_LAYERS=nl, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2], IMAGE_FROZEN_BN=True,
                           IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True)
    net = M.ResNetVLBERTForPretraining(conf, device="cuda:0")
    ref_sd = dict(params)
    ref_sd.update(_prefixed(P))
    net.load_state_dict({k: v for k, v in ref_sd.items()})
    net.eval()
    batch = list(syn.make_batch(B, T, R, seed=22, ragged=False))
    batch[0] = boxes4.clone()                                             # e2e collate: boxes [B,R,4]
    batch[1] = torch.from_numpy(z["im_info"])
    pad = boxes4[:, :, 0] <= -1.5
    batch[5][pad] = 0
    batch[6][pad] = 0
    outputs, loss = net(img.to(dev()), *[t.to(dev()) for t in batch])
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    obatch = list(batch)
    obatch[0] = torch.cat((boxes4, torch.zeros(B, R, 2048)), -1)
    out, oloss = O.pretrain_forward(leaves, cfg, *obatch, train=False, image=img, vision_params=Po, image_num_layers=nl)
    print("e2e module loss: hip %.5f oracle %.5f" % (float(loss), float(oloss)))
    assert abs(float(loss) - float(oloss)) < 2e-2 * abs(float(oloss))
    report("e2e module mlm_logits", outputs["mlm_logits"], out["mlm_logits"].detach(), 5e-2, 2e-2)
    loss.backward()
    oloss.backward()
    name = "image_feature_extractor.roi_head_feature_extractor.2.conv2.weight"
    g = dict(net.named_parameters())[name].grad                           # engine layout [O,KH,KW,I]
    e = rel_fro(g.permute(0, 3, 1, 2), Po["layer4.2.conv2.weight"].grad)
    print("e2e module: d %s rel-fro %.3e" % (name, e))
    assert e < 5e-2
    trainable = {n for n, p in net.named_parameters() if p.requires_grad}
    assert "image_feature_extractor.backbone.layer1.0.conv1.weight" not in trainable      # frozen stage: a buffer
    assert "image_feature_extractor.backbone.layer2.0.conv1.weight" in trainable
    # checkpoint round trip in the reference's layout
    sd = net.state_dict()
    for k, v in _prefixed(P).items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
        assert torch.equal(sd[k].cpu(), v), k
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    before = sd[name].clone()
    opt.step()
    _, loss2 = net(img.to(dev()), *[t.to(dev()) for t in batch])          # weights re-synchronised (bf16 / folded copies) on use
    assert not torch.equal(net.state_dict()[name], before)
    assert torch.isfinite(loss2)


def test_fast_rcnn_mirror_image_branch_vs_oracle():
    """`common.fast_rcnn.FastRCNN` mirror with IMAGE_FEAT_PRECOMPUTED false (the class the VQA / VCR wrappers construct with images):
    reference-layout checkpoint in, forward(images, boxes[B,R,4], box_mask, im_info) -> obj_reps_raw / obj_reps, autograd backward
    into the RoI head, the trainable trunk stages and obj_downsample."""
    F_ = pkg("common.fast_rcnn")
    z, nl, P = _vision_fixture()
    img, boxes4, im_info = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"]), torch.from_numpy(z["im_info"])
    B, R = boxes4.shape[:2]

    class A(dict):
        __getattr__ = dict.__getitem__
    conf = A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_SEMANTIC=False, IMAGE_NUM_LAYERS=nl, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2],
                       IMAGE_FROZEN_BN=True, IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True, OUTPUT_CONV5=False))
    net = F_.FastRCNN(conf, average_pool=True, final_dim=768, device="cuda:0")
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=21)
    sd = {k: v for k, v in VO.split_state_dict(P).items()}
    for k in ("weight", "bias"):
        sd["obj_downsample.1." + k] = params["image_feature_extractor.obj_downsample.1." + k]
    net.load_state_dict(sd)
    net.eval()
    mask = boxes4[:, :, 0] > -1.5
    out = net(img.to(dev()), boxes4.to(dev()), mask.to(dev()), im_info.to(dev()))
    report("FastRCNN e2e obj_reps_raw vs reference fixture", out["obj_reps_raw"], torch.from_numpy(z["obj_reps_raw"]), 2e-2, 2e-2)
    frozen = VO.frozen_names(P)
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    feats, _ = VO.e2e_features(img, boxes4, Po, nl)
    full = torch.zeros(B, R, 4 + 2048)
    full[:, :, :4] = boxes4
    full = torch.cat((full[:, :, :4], feats.new_zeros(B, R, 2048).masked_scatter(mask[:, :, None], feats)), -1)
    want = O.fast_rcnn_precomputed(leaves, cfg, full, mask, im_info, train=False)
    report("FastRCNN e2e obj_reps vs oracle", out["obj_reps"], want.detach(), 2e-2, 2e-2)
    W = torch.randn(want.shape, generator=torch.Generator().manual_seed(5)) / want.numel() ** 0.5
    (out["obj_reps"] * W.to(dev())).sum().backward()
    (want * W).sum().backward()
    got = dict(net.named_parameters())
    e1 = rel_fro(got["obj_downsample.1.weight"].grad, leaves["image_feature_extractor.obj_downsample.1.weight"].grad)
    e2 = rel_fro(got["roi_head_feature_extractor.1.conv2.weight"].grad.permute(0, 3, 1, 2), Po["layer4.1.conv2.weight"].grad)
    e3 = rel_fro(got["backbone.layer3.2.conv1.weight"].grad.permute(0, 3, 1, 2), Po["layer3.2.conv1.weight"].grad)
    print("FastRCNN e2e grads rel-fro: obj_downsample %.3e, head conv2 %.3e, layer3 conv1 %.3e" % (e1, e2, e3))
    assert e1 < 6e-2 and e2 < 6e-2 and e3 < 0.12      # (ReLU sign flips of bf16-vs-fp32 pre-activations: cf. the FastRCNN test in test_engine_gpu.py)
    assert "backbone.layer1.0.conv1.weight" not in got and "backbone.bn1.weight" not in got          # frozen: buffers
    back = net.state_dict()
    for k, v in VO.split_state_dict(P).items():
        assert torch.equal(back[k].cpu(), v), k
    # VCR call form: `segms` object masks inside the RoI head (common/fast_rcnn.py:152-156), against the reference fixture + oracle grads
    segms = torch.from_numpy(z["segms"])
    for q in net.parameters():
        q.grad = None
    out_s = net(img.to(dev()), boxes4.to(dev()), mask.to(dev()), im_info.to(dev()), segms=segms.to(dev()))
    report("FastRCNN e2e obj_reps_raw with segms vs reference fixture", out_s["obj_reps_raw"], torch.from_numpy(z["obj_reps_raw_segms"]), 2e-2, 2e-2)
    Po2 = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    leaves2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    feats_s, _ = VO.e2e_features(img, boxes4, Po2, nl, segms=segms)
    full_s = torch.cat((boxes4, feats_s.new_zeros(B, R, 2048).masked_scatter(mask[:, :, None], feats_s)), -1)
    want_s = O.fast_rcnn_precomputed(leaves2, cfg, full_s, mask, im_info, train=False)
    (out_s["obj_reps"] * W.to(dev())).sum().backward()
    (want_s * W).sum().backward()
    e4 = rel_fro(dict(net.named_parameters())["roi_head_feature_extractor.2.conv3.weight"].grad.permute(0, 3, 1, 2), Po2["layer4.2.conv3.weight"].grad)
    print("FastRCNN e2e with segms: d head conv3 rel-fro %.3e" % e4)
    assert e4 < 0.12      # (through obj_downsample's bf16 ReLU flips; the sharp check of the masked pool backward is in the stack test)


@pytest.mark.parametrize("Hi,Wi", [(75, 101), (130, 66)])
def test_vision_stack_odd_image_sizes_vs_oracle(Hi, Wi):
    """Image sizes that are not multiples of the strides (odd stem / pool / stride-2 output sizes, body4 of 5x7 / 9x5): forward and one
    weight gradient per stage against oracle/vision_oracle.py (itself pinned to the reference module)."""
    V = pkg("vision")
    nl = 50
    P = VO.init_vision_params(3, nl)
    g = torch.Generator().manual_seed(4)
    N, R = 2, 2
    img = torch.randn(N, 3, Hi, Wi, generator=g) * 50
    boxes4 = torch.tensor([[[3.0, 4.0, Wi - 5.0, Hi - 6.0], [Wi * 0.3, Hi * 0.2, Wi * 0.9, Hi * 0.7]],
                           [[0.0, 0.0, Wi - 1.0, Hi - 1.0], [-2.0, -2.0, -2.0, -2.0]]])
    vs = V.VisionStack(N, Hi, Wi, R, device=dev(), num_layers=nl)
    vs.load_state_dict({k: v.to(dev()) for k, v in _prefixed(P).items()})
    boxes = torch.zeros((N, R, 4 + 2048), device=dev())
    boxes[:, :, :4] = boxes4.to(dev())
    vs.forward(img.to(dev()), boxes)
    frozen = VO.frozen_names(P)
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    feats, body4 = VO.e2e_features(img, boxes4, Po, nl)
    mask = boxes4[:, :, 0] > -1.5
    assert (vs.H3, vs.W3) == tuple(body4.shape[2:])
    report("odd %dx%d body4" % (Hi, Wi), vs.body4.float().cpu().view(N, vs.H3, vs.W3, -1).permute(0, 3, 1, 2), body4.detach(), 2e-2, 2e-2)
    report("odd %dx%d post_roialign" % (Hi, Wi), boxes[:, :, 4:].cpu()[mask], feats.detach(), 2e-2, 2e-2)
    W = torch.randn(N, R, 2048, generator=g) / 64.0
    vs.zero_grad()
    vs.backward(to_gpu_bf16(W.view(N * R, -1)), boxes)
    torch.cuda.synchronize()
    (feats * W[mask]).sum().backward()
    got = vs.grads()
    for short, ref_name in (("roi_head_feature_extractor.0.conv2.weight", "layer4.0.conv2.weight"),
                            ("backbone.layer3.0.downsample.0.weight", "layer3.0.downsample.0.weight"),
                            ("backbone.layer2.0.conv1.weight", "layer2.0.conv1.weight")):
        e = rel_fro(got["image_feature_extractor." + short], Po[ref_name].grad)
        print("   odd %dx%d d %s rel-fro %.3e" % (Hi, Wi, short, e))
        assert e < 0.15, short


def test_resnet101_forward_at_headline_image_size_vs_oracle():
    """The e2e configuration's real shape (BASELINE config 3): ONE 600x1000 image through the full ResNet-101 trunk (conv1 ... layer3,
    stride 16 -> body4 38x63x1024), ROIAlign 14x14 and the dilated layer4 head, forward only, against oracle/vision_oracle.py
    (torch fp32 on the host).  bf16 activations through 101 convolutions: rel-Frobenius bounds, measured values printed."""
    V = pkg("vision")
    nl, Hi, Wi = 101, 600, 1000
    P = VO.init_vision_params(5, nl)
    g = torch.Generator().manual_seed(6)
    N, R = 1, 4
    img = torch.randn(N, 3, Hi, Wi, generator=g) * 50
    boxes4 = torch.tensor([[[0.0, 0.0, Wi - 1.0, Hi - 1.0], [120.5, 80.25, 431.0, 377.5], [700.0, 300.0, 990.0, 590.0],
                            [-2.0, -2.0, -2.0, -2.0]]])
    vs = V.VisionStack(N, Hi, Wi, R, device=dev(), num_layers=nl)
    vs.load_state_dict({k: v.to(dev()) for k, v in _prefixed(P).items()})
    boxes = torch.zeros((N, R, 4 + 2048), device=dev())
    boxes[:, :, :4] = boxes4.to(dev())
    vs.forward(img.to(dev()), boxes)
    torch.cuda.synchronize()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        feats, body4 = VO.e2e_features(img, boxes4, P, nl)
    mask = boxes4[:, :, 0] > -1.5
    assert (vs.H3, vs.W3) == tuple(body4.shape[2:]) == (38, 63)
    e_body = rel_fro(vs.body4.float().cpu().view(N, vs.H3, vs.W3, -1).permute(0, 3, 1, 2), body4)
    e_feat = rel_fro(boxes[:, :, 4:].cpu()[mask], feats)
    print("ResNet-101 600x1000: body4 rel-fro %.3e, post-ROIAlign layer4 features rel-fro %.3e (|feats| max %.3f)" %
          (e_body, e_feat, float(feats.abs().max())))
    assert e_body < 2e-2 and e_feat < 1e-2          # measured on MI355X: 9.1e-3 / 3.0e-3
    assert float(boxes[0, 3, 4:].abs().max()) == 0.0          # the padded box slot stays zero


def test_e2e_checkpoint_optimizer_state_in_the_reference_index_order(tmp_path):
    """Round-4 ADVICE (medium): for the e2e model the reference's optimizer indexes EVERY named parameter -- frozen backbone stages and
    frozen BatchNorm weights / biases too (pretrain/function/train.py:139-142), with no state entry for them.  The engine's checkpoint
    (a) writes that index table (len = the reference module's parameter count, golden e2e_fastrcnn_50 + the VL-BERT part), state only
    on the trainable tensors, moments of the convolutions in [O,I,KH,KW]; (b) a file in the REFERENCE's form (no 'param_names', one
    group) loads into a fresh engine and gives back the identical Adam state and step count; (c) a several-group file without names is
    refused instead of being zipped against the wrong tensors."""
    import json
    E, C, syn = pkg("engine"), pkg("common.checkpoint"), pkg("synthetic")
    z, nl, P = _vision_fixture()
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"]).clone()
    B, R, T = boxes4.shape[0], boxes4.shape[1], 12
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=21)
    batch = list(syn.make_batch(B, T, R, seed=22, ragged=False))
    batch[0] = torch.cat((boxes4, torch.zeros(B, R, 2048)), -1)
    batch[1] = torch.from_numpy(z["im_info"])
    pad = boxes4[:, :, 0] <= -1.5
    batch[5][pad] = 0
    batch[6][pad] = 0
    mc = E.ModelConfig(num_hidden_layers=1, e2e=True, image_num_layers=nl)

    def fresh():
        eng = E.PretrainEngine(mc, B, T, R, device="cuda:0", train=False, lr=1e-3, image_size=tuple(img.shape[2:]))
        sd = {k: v.to(dev()) for k, v in params.items()}
        sd.update({k: v.to(dev()) for k, v in _prefixed(P).items()})
        eng.load_state_dict(sd)
        eng.set_batch(*[t.to(dev()) for t in batch], image=img.to(dev()))
        eng.sync_weights()
        return eng
    eng = fresh()
    for _ in range(2):
        eng.zero_grad(); eng.forward(False); eng.backward(False); eng.optimizer_step()
    torch.cuda.synchronize()
    path = C.save_checkpoint(eng, str(tmp_path / "e2e"), 0)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    osd = ck["optimizer"]
    with open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint", "param_order.json")) as f:
        gold = json.load(f)
    vis = ["image_feature_extractor." + n for n in gold["e2e_fastrcnn_%d" % nl]]
    names = osd["param_names"]
    assert names[:len(vis)] == vis and osd["param_groups"][0]["params"] == list(range(len(names)))
    trainable = {"image_feature_extractor." + n for n in gold["e2e_fastrcnn_%d_trainable" % nl]}
    for i, n in enumerate(names):
        if n.startswith("image_feature_extractor."):
            assert (i in osd["state"]) == (n in trainable), n
    i = names.index("image_feature_extractor.roi_head_feature_extractor.0.conv2.weight")
    assert tuple(osd["state"][i]["exp_avg"].shape) == tuple(ck["state_dict"][names[i]].shape) and osd["state"][i]["exp_avg"].shape[2:] == (3, 3)
    assert float(osd["state"][i]["exp_avg"].abs().max()) > 0 and int(osd["state"][i]["step"]) == 2
    # (b) the reference's form of the same file
    ref_form = dict(ck)
    ref_form["optimizer"] = {"state": osd["state"], "param_groups": osd["param_groups"]}
    torch.save(ref_form, str(tmp_path / "ref-0000.model"))
    eng2 = fresh()
    C.load_checkpoint(eng2, str(tmp_path / "ref-0000.model"))
    torch.cuda.synchronize()
    assert torch.equal(eng2.P.m, eng.P.m) and torch.equal(eng2.P.v, eng.P.v) and torch.equal(eng2.P.master, eng.P.master)
    assert float(eng2.adam[5]) == 2.0
    # (c) several groups, no names: refused
    bad = {"state": osd["state"], "param_groups": [dict(osd["param_groups"][0], params=[0]), dict(osd["param_groups"][0], params=list(range(1, len(names))))]}
    with pytest.raises(ValueError, match="param_groups"):
        C.load_optimizer_state_dict(eng2, bad)
