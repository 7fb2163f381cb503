#!/usr/bin/env python
"""How much do the persistent one-workgroup-per-CU GEMM kernels lose when a few CUs are held by another kernel (as an RCCL collective
would do during the backward of a data-parallel step)?  Runs the bench step with a spin kernel of N workgroups resident on a second
stream (tools/contention_probe.hip), for the default workgroup caps and for p8_wgs / tn8_wgs = 256 - N.  Development tool."""
import ctypes
import importlib
import os
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(tempfile.gettempdir(), "contention_probe.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "contention_probe.hip"),
                "-o", so], check=True)
probe = ctypes.CDLL(so)
probe.contention_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
engine = importlib.import_module("vl-bert_amd.engine")
syn = importlib.import_module("vl-bert_amd.synthetic")
lib = importlib.import_module("vl-bert_amd._lib")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = engine.ModelConfig(num_hidden_layers=12)
eng = engine.PretrainEngine(cfg, B, 64, 36, device="cuda:0", train=True, lr=1e-4, weight_decay=1e-4, max_grad_norm=10.0)
eng.init_random(seed=0)
eng.set_batch(*[t.cuda() for t in syn.make_batch(B, 64, 36, seed=100)])
eng.sync_weights()
side = torch.cuda.Stream()
sink = torch.zeros(1, dtype=torch.int32, device="cuda:0")


def run(nspin, steps=12, spin_us=300):
    """step time on the MAIN stream (events), with the spinners covering roughly the first half of every step"""
    for _ in range(3):
        eng.train_step()
    torch.cuda.synchronize()
    base = getattr(run, "base_ms", None)
    nlaunch = max(1, int((base or 12.0) * 0.5e3 / spin_us))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        if nspin:
            side.wait_stream(torch.cuda.current_stream())      # spinners start with the step
            for _ in range(nlaunch):
                probe.contention_spin(nspin, spin_us, sink.data_ptr(), side.cuda_stream)
        eng.train_step()
    e1.record()
    e1.synchronize()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    if not nspin and base is None:
        run.base_ms = ms
    return ms


for nspin in (0, 8, 16, 32, 64):
    lib.gemm_set_option("p8_wgs", 256)
    lib.gemm_set_option("tn8_wgs", 256)
    a = run(nspin)
    torch.cuda.synchronize()
    cap = max(8, (256 - nspin) & ~7)
    lib.gemm_set_option("p8_wgs", cap)
    lib.gemm_set_option("tn8_wgs", cap)
    b = run(nspin)
    torch.cuda.synchronize()
    print("batch %d, %2d CUs held by a spin kernel: step %.2f ms with 256 GEMM workgroups | %.2f ms with %d" % (B, nspin, a, b, cap), flush=True)
