"""Stream-K vs whole-tile kernels on the N = 768 GEMMs of the encoder layer at the per-GPU batches of the strong-scaling run.
    python tools/sk_bench.py [batches ...]        (default 32 64 128)
Per shape: the launcher's choice with stream-K off (option nt_sk 0: ring / 128x128 / large-tile core) against stream-K forced
(nt_sk 2), microseconds per launch over 60 launches (HIP events), TFLOP/s, and the largest difference between the two results.
Epilogues as the step uses them: forward GEMMs with bias + dropout + LayerNorm-residual -> fp16, data gradients with + residual."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("vl-bert_amd._lib")
ops = importlib.import_module("vl-bert_amd.ops")


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    batches = [int(x) for x in sys.argv[1:]] or [32, 64, 128]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    H = 768
    print("%-22s %6s %5s %5s | %9s %8s | %9s %8s | %7s %s" % ("gemm", "M", "N", "K", "off us", "TF/s", "stream-K", "TF/s", "speedup", "max|diff|"))
    for Bt in batches:
        M = Bt * 101
        for name, K, form in (("attn-out fwd", 768, "ln"), ("ffn2 fwd", 3072, "ln"), ("out dgrad", 768, "plain"), ("qkv dgrad", 2304, "res"),
                              ("ffn1 dgrad", 3072, "res")):
            A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(ops.BF16)
            W = (torch.randn(H, K, device=dev, generator=g) * 0.05).to(ops.BF16)
            bias = torch.randn(H, device=dev, generator=g) * 0.1
            res16 = torch.randn(M, H, device=dev, generator=g).to(ops.BF16)
            z = torch.randn(M, H, device=dev, generator=g).half()
            st = torch.stack((z.float().mean(1), 1.0 / (z.float().var(1, unbiased=False) + 1e-12).sqrt()), 1).contiguous()
            gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
            seed = torch.tensor([1234], dtype=torch.int32, device=dev)
            if form == "ln":
                C = torch.empty(M, H, dtype=torch.float16, device=dev)
                fn = lambda: ops.gemm_nt(A, W, C, bias=bias, res=z, res_ln=(st, gam, bet), drop_p=0.1, seed=seed, tag=3)
            elif form == "res":
                C = torch.empty(M, H, dtype=ops.BF16, device=dev)
                fn = lambda: ops.gemm_nt(A, W, C, res=res16)
            else:
                C = torch.empty(M, H, dtype=ops.BF16, device=dev)
                fn = lambda: ops.gemm_nt(A, W, C)
            out = {}
            for mode in (0, 2):
                L.gemm_set_option("nt_sk", mode)
                t = timed(fn)
                out[mode] = (t, C.float().clone())
            L.gemm_set_option("nt_sk", 0)
            # the whole-tile ring kernel forced to its variants (p8 off so that the ring is what runs): 3 = 128x64 / 4 waves, 2 = 128x128 / 8 waves
            ring = {}
            L.gemm_set_option("p8_mode", 0)
            for rm in (3, 2):
                L.gemm_set_option("nt_ring", rm)
                ring[rm] = timed(fn)
            L.gemm_set_option("nt_ring", 1)
            L.gemm_set_option("p8_mode", 1)
            fl = 2.0 * M * H * K
            d = float((out[0][1] - out[2][1]).abs().max())
            print("%-22s %6d %5d %5d | %9.1f %8.1f | %9.1f %8.1f | %6.2fx %.3e | ring 128x64 %6.1f  128x128/8w %6.1f" %
                  ("b%d %s" % (Bt, name), M, H, K, out[0][0], fl / out[0][0] * 1e-6, out[2][0], fl / out[2][0] * 1e-6, out[0][0] / out[2][0], d,
                   ring[3], ring[2]))
    print("stream-K hand-off timeouts:", L.gemm_sk_timeouts())


if __name__ == "__main__":
    main()
