"""Data-parallel step on the GPU box (-m gpu): two engine ranks SHARE the one MI355X over gloo (RCCL refuses two ranks per device; the
engine's torch.distributed calls are the same), through tools/dp2_check.py -- bucket hooks inside backward, reduce-scatter /
all-reduce of the gradient buckets, sharded clip + AdamW (vlb_sumsq_ranges_det / vlb_adamw_step_ranges), weight all-gather under the
next forward, bit-identical replicas after two optimizer steps.  Replaces pretrain/function/train.py:89-90 + common/trainer.py:139-153."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, tool="dp2_check.py"):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", tool)] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    print(r.stderr[-3000:])
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "dp_test_%s.log" % "_".join([tool[:-3]] + [a.strip("-") for a in extra])), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("parameters identical to rank 0 after 2 D") == 2 and "identical to rank 0 after 2 DP steps: False" not in r.stdout \
        and "after 2 DDP steps: False" not in r.stdout


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_two_ranks_on_one_gpu_precomputed(mode):
    _run(["--mode", mode])


def test_two_ranks_on_one_gpu_e2e_sharded():
    _run(["--e2e", "--mode", "sharded"])


def test_module_mirror_under_distributed_data_parallel():
    """parallel.DistributedDataParallel around the VQA module mirror (vqa/function/train.py:327): gradients = the hand-averaged local
    gradients, identical replicas after two FusedAdamW steps with the fused clip, no_sync() semantics."""
    _run([], tool="dp2_mirror_check.py")
