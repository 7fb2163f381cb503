"""ROIAlign backward at the e2e geometry (8 images x 36 boxes, 14 x 14 bins x 1024 channels onto 38 x 63): atomic scatter (+ memset +
ReLU-mask / cast pass) vs the gather form.  python tools/roi_bwd_bench.py [synthetic|large]"""
import importlib
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
D = torch.device("cuda:0")
N, R, C, H, W, P = 8, 36, 1024, 38, 63, 14
Wi, Hi = 1000.0, 600.0
g = torch.Generator().manual_seed(3)
kind = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
x1 = torch.rand(N, R, generator=g) * (Wi - 200)
y1 = torch.rand(N, R, generator=g) * (Hi - 200)
if kind == "large":      # detector-like boxes: 10-60 % of the image per side
    w = Wi * (0.1 + 0.5 * torch.rand(N, R, generator=g)); h = Hi * (0.1 + 0.5 * torch.rand(N, R, generator=g))
    x1 = torch.rand(N, R, generator=g) * (Wi - w); y1 = torch.rand(N, R, generator=g) * (Hi - h)
else:                    # bench.py's synthetic boxes
    w = 30 + torch.rand(N, R, generator=g) * 160; h = 30 + torch.rand(N, R, generator=g) * 160
boxes = torch.stack((x1, y1, x1 + w, y1 + h), -1)
boxes[:, 0] = torch.tensor([0.0, 0.0, Wi - 1, Hi - 1])
boxes[:, 30:] = -2.0      # ragged: 30 valid boxes per image
bx = boxes.view(N * R, 4).contiguous().to(D)
dout = torch.randn(N * R * P * P, C, generator=g).to(torch.bfloat16).to(D)
act = torch.randn(N * H * W, C, generator=g).to(torch.bfloat16).to(D)
d32 = torch.zeros((N * H * W, C), device=D)
o1 = torch.zeros((N * H * W, C), dtype=torch.bfloat16, device=D)
o2 = torch.zeros((N * H * W, C), dtype=torch.bfloat16, device=D)
ws = ops.roi_align_gather_workspace(N * R, H, W, P, D)


def scatter():
    ops.roi_align_nhwc_bwd(dout, bx, R, d32, N, H, W, C, pooled=P)
    ops.relu_mask_cast(d32, act, o1)


def gather():
    ops.roi_align_nhwc_bwd_gather(dout, bx, R, ws, N, H, W, C, act=act, dx_bf16=o2, pooled=P)


for name, fn in (("scatter + mask pass", scatter), ("gather", gather)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s boxes, %-20s %8.1f us" % (kind, name, e0.elapsed_time(e1) * 100), flush=True)
err = float((o1.float() - o2.float()).abs().max())
print("max |scatter - gather| %.3e (scale %.3e)" % (err, float(o1.float().abs().max())))
