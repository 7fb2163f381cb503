#!/usr/bin/env python
"""Host-side launch cost of one training step (eager): time to ENQUEUE steps while the device is still busy with earlier ones.
Development tool; run on the GPU box."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
engine = importlib.import_module("vl-bert_amd.engine")
syn = importlib.import_module("vl-bert_amd.synthetic")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = engine.PretrainEngine(engine.ModelConfig(num_hidden_layers=12), B, 64, 36, device="cuda:0", train=True, lr=1e-4, weight_decay=1e-4,
                            max_grad_norm=10.0)
eng.init_random(seed=0)
eng.set_batch(*[t.cuda() for t in syn.make_batch(B, 64, 36, seed=100)])
eng.sync_weights()
for _ in range(3):
    eng.train_step()
torch.cuda.synchronize()
# a long device-side head start so that the host never waits for queue space
a = torch.zeros((8192, 8192), dtype=torch.bfloat16, device="cuda:0")
for _ in range(40):
    torch.mm(a, a)
t0 = time.perf_counter()
for _ in range(2):
    eng.train_step()
host = (time.perf_counter() - t0) / 2
torch.cuda.synchronize()
print("batch %d: host time to enqueue one step %.2f ms" % (B, host * 1e3))
for _ in range(40):
    torch.mm(a, a)
pr = cProfile.Profile()
pr.enable()
eng.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
