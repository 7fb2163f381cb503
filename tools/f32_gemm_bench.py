"""Per-shape throughput of vlb_gemm_nt_f32 (the fp32 encoder's GEMM: fp32 operands, products split over 3 bf16 MFMAs) at the shapes of
one VL-BERT-large micro-batch of BASELINE config 4 (16 samples x 229 positions = 3664 rows).  Development tool; run on the GPU box."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")
dev = torch.device("cuda:0")


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, H, I, S, Sp, Bt, nh = 3664, 1024, 4096, 229, 256, 16, 16
rnd = lambda *s: torch.randn(s, device=dev)
print("%-44s %9s %9s" % ("shape", "us", "TFLOP/s"))
for name, m, n, k in (("QKV fwd", M, 3 * H, H), ("attn-out fwd / dgrad", M, H, H), ("FFN1 fwd", M, I, H), ("FFN2 fwd", M, H, I),
                      ("QKV dgrad", M, H, 3 * H), ("FFN1 dgrad", M, H, I), ("FFN2 dgrad", M, I, H)):
    A, B, C = rnd(m, k), rnd(n, k), rnd(m, n)
    us = timed(lambda: ops.gemm_nt_f32(A, k, B, k, C, n, m, n, k))
    print("%-44s %9.1f %9.1f" % ("%s %dx%dx%d" % (name, m, n, k), us, 2.0 * m * n * k / us / 1e6))
Mp = 3680
for name, m, n in (("wgrad QKV", 3 * H, H), ("wgrad attn-out", H, H), ("wgrad FFN1", I, H), ("wgrad FFN2", H, I)):
    A, B, C = rnd(m, Mp), rnd(n, Mp), rnd(m, n)
    for sk in (1, 2, 3, 4, 8):
        us = timed(lambda: ops.gemm_nt_f32(A, Mp, B, Mp, C, n, m, n, Mp, atomic=True, splitk=sk))
        print("%-44s %9.1f %9.1f" % ("%s %dx%dx%d split %d (accumulate)" % (name, m, n, Mp, sk), us, 2.0 * m * n * Mp / us / 1e6))
qkv = rnd(M + Sp, 3 * H)
sc = rnd(Bt * nh * S, Sp)
vt = rnd(Bt * nh * 64, Sp)
ctx = rnd(M, H)
us = timed(lambda: ops.gemm_nt_f32(qkv, 3 * H, (qkv, H), 3 * H, sc, Sp, S, Sp, 64, batch=(Bt, nh), sA=(S * 3 * H, 64), sB=(S * 3 * H, 64),
                                   sC=(nh * S * Sp, S * Sp), alpha=0.125))
print("%-44s %9.1f %9.1f" % ("scores 256 x (229x256x64)", us, 2.0 * S * Sp * 64 * Bt * nh / us / 1e6))
us = timed(lambda: ops.gemm_nt_f32(sc, Sp, vt, Sp, ctx, H, S, 64, Sp, batch=(Bt, nh), sA=(nh * S * Sp, S * Sp), sB=(nh * 64 * Sp, 64 * Sp),
                                   sC=(S * H, 64)))
print("%-44s %9.1f %9.1f" % ("P.V 256 x (229x64x256)", us, 2.0 * S * 64 * Sp * Bt * nh / us / 1e6))
tt = rnd(Bt * nh * S, Sp)
us = timed(lambda: ops.transpose_f32(sc, Sp, tt, Sp, S, S, Sp, batch=(Bt, nh), sS=(nh * S * Sp, S * Sp), sD=(nh * S * Sp, S * Sp)))
print("%-44s %9.1f" % ("transpose 256 x [229,229]", us))
x, t = rnd(M, I), rnd(I, Mp)
us = timed(lambda: ops.transpose_f32(x, I, t, Mp, M, I, Mp))
print("%-44s %9.1f   (%.1f TB/s)" % ("transpose [3664,4096] -> [4096,3680]", us, 2.0 * M * I * 4 / us / 1e6))
