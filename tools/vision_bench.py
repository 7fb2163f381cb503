"""Times the e2e vision path (VisionStack forward / backward) on one MI355X at the reference's e2e shape
(cfgs/pretrain/base_e2e_16x16G_fp16.yaml: ResNet-101, 8 images of 600x1000 per GPU, 36 boxes) with random weights.
Usage: python tools/vision_bench.py [N] [H] [W] [R] [layers]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("vl-bert_amd.vision")


def flops(vs):
    """multiply-add = 2 FLOP; forward of every convolution, and which of them also run dgrad / wgrad."""
    fwd = bwd = 0
    need_dx = vs._dgrad_set()
    stem = vs.convs["backbone.conv1"]
    fwd += 2 * vs.N * vs.H1 * vs.W1 * 147 * stem.O
    for b in vs.blocks:
        k, M, P, C = b["key"], b["M"], b["planes"], b["inplanes"]
        per = {"conv1": 2 * M * C * P, "conv2": 2 * M * 9 * P * P, "conv3": 2 * M * P * 4 * P}
        if b["downsample"]:
            per["downsample.0"] = 2 * M * C * 4 * P
        fwd += sum(per.values())
        if b["trainable"]:
            for n, f in per.items():
                bwd += f                      # wgrad
                if k + n in need_dx:
                    bwd += f                  # dgrad
    return fwd, bwd


def main():
    a = [int(x) for x in sys.argv[1:]]
    N, H, W, R, layers = (a + [8, 600, 1000, 36, 101][len(a):])[:5]
    dev = torch.device("cuda:0")
    vs = V.VisionStack(N, H, W, R, device=dev, num_layers=layers)
    g = torch.Generator().manual_seed(0)
    for c in vs.convs.values():
        c.w32.copy_((torch.randn(c.w32.shape, generator=g) * (2.0 / (c.O * c.taps)) ** 0.5).to(dev))
        if c.key.endswith("conv3") or c.key == "backbone.conv1":
            c.bn[0].fill_(0.2)
    vs.refresh_weights()
    img = (torch.randn(N, 3, H, W, generator=g) * 50).to(dev)
    boxes = torch.zeros((N, R, 4 + 2048), device=dev)
    x1 = torch.rand(N, R, generator=g) * (W - 200)
    y1 = torch.rand(N, R, generator=g) * (H - 200)
    boxes[:, :, 0], boxes[:, :, 1] = x1.to(dev), y1.to(dev)
    boxes[:, :, 2] = (x1 + 20 + torch.rand(N, R, generator=g) * 180).to(dev)
    boxes[:, :, 3] = (y1 + 20 + torch.rand(N, R, generator=g) * 180).to(dev)
    d_feat = (torch.randn(N * R, 2048, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    f_fwd, f_bwd = flops(vs)
    print("alloc %.1f GB; forward %.2f TFLOP, backward %.2f TFLOP per step (%d images)" %
          (torch.cuda.memory_allocated() / 2**30, f_fwd / 1e12, f_bwd / 1e12, N))
    for _ in range(2):
        vs.forward(img, boxes)
        vs.backward(d_feat, boxes)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    steps = 5
    tf = tb = 0.0
    for _ in range(steps):
        e[0].record()
        vs.forward(img, boxes)
        e[1].record()
        vs.backward(d_feat, boxes)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    tf, tb = tf / steps, tb / steps
    print("forward %.2f ms (%.0f TFLOP/s)  backward %.2f ms (%.0f TFLOP/s)  -> %.1f images/s for the vision path alone" %
          (tf, f_fwd / tf / 1e9, tb, f_bwd / tb / 1e9, N / (tf + tb) * 1e3))
    print("features finite:", bool(torch.isfinite(boxes).all()), " |feat| max %.3f" % float(boxes[:, :, 4:].abs().max()))


if __name__ == "__main__":
    main()
