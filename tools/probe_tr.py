#!/usr/bin/env python
"""Probe the lane/element semantics of ds_read_b64_tr_b16 on gfx950 (development tool)."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np, torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), "probe_tr.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "probe_tr.hip"), "-o", so], check=True)
probe = ctypes.CDLL(so)
probe.probe_tr_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

def run(addr):
    d = "cuda:0"
    n = 8192
    inp = torch.arange(n, dtype=torch.int32).to(torch.int16).to(d)
    a = torch.tensor(addr, dtype=torch.int32, device=d)
    out = torch.zeros(256, dtype=torch.int16, device=d)
    assert probe.probe_tr_read(inp.data_ptr(), n, a.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.int64).reshape(64, 4) & 0xFFFF

# A: linear addresses, lane l -> byte 8*l
o = run([8 * l for l in range(64)])
pred = np.array([[(l & 15) + 16 * j + (l >> 4) * 64 for j in range(4)] for l in range(64)])
print("A linear: matches guide formula:", bool((o == pred).all()))
print(o[:20])
# B: random per-lane 8-byte aligned addresses -> model: lane i, elem j <- lane (4j + i//4) of the same 16-group, element i%4
rng = np.random.RandomState(0)
addr = (rng.randint(0, 1024, size=64) * 8).tolist()
o = run(addr)
pred = np.zeros((64, 4), dtype=np.int64)
for l in range(64):
    g, i = l >> 4, l & 15
    for j in range(4):
        src = 16 * g + 4 * j + i // 4
        pred[l, j] = addr[src] // 2 + (i % 4)
print("B scattered: matches (src lane = 4j + i//4, elem i%4):", bool((o == pred).all()))
if not (o == pred).all():
    print(o[:16]); print(pred[:16]); print(addr[:16])
