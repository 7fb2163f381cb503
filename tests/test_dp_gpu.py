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


def _rank_report(stdout, stderr, limit=70):
    """What the ranks themselves said: every `[rankN]:` traceback line torchrun relays (the failing rank's exception, not the
    launcher's boilerplate that follows it) and the last result lines of the tool -- this is what goes into the assertion message."""
    import re
    tb = [l for l in stderr.splitlines() if re.match(r"^\[rank\d+\]:", l)]
    # the rank that failed FIRST is the one whose error is not a consequence of a peer going away
    own = [l for l in tb if "Connection closed by peer" not in l and "SIGTERM" not in l]
    said = [l for l in stdout.splitlines() if re.match(r"^rank \d+", l)]
    return "\n".join(["---- rank tracebacks ----"] + (own or tb)[:limit] + ["---- rank output (tail) ----"] + said[-24:])


def _launch(cmd, env, tag, timeout=600):
    """Run one multi-process tool; full per-run log under gpurun_out/ (merged back from the GPU box), rank-level report on failure."""
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "dp_test_%s.log" % tag), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    report = _rank_report(r.stdout, r.stderr)
    print(report)
    assert r.returncode == 0, "%s exited %d\n%s" % (tag, r.returncode, report)
    return r


def _run(extra, tool="dp2_check.py", nproc=2, env_extra=None, marker="parameters identical to rank 0 after 2 D"):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    env.update(env_extra or {})
    path = os.path.join(ROOT, tool) if os.sep in tool else os.path.join(ROOT, "tools", tool)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), path] + extra
    r = _launch(cmd, env, "_".join([os.path.basename(tool)[:-3]] + [a.strip("-") for a in extra] + ["n%d" % nproc]))
    report = _rank_report(r.stdout, r.stderr)
    assert r.stdout.count(marker) == nproc and "identical to rank 0 after 2 DP steps: False" not in r.stdout \
        and "after 2 DDP steps: False" not in r.stdout, report
    return r.stdout


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_two_ranks_on_one_gpu_precomputed(mode):
    out = _run(["--mode", mode])
    if mode == "sharded":
        # the fp32-read tensors (biases, LayerNorm parameters) are replicated with the weight gather, checked BEFORE any master gather,
        # and a 5-step sharded trajectory equals the all-reduce one (round-3 ADVICE: stale biases on the non-owner ranks)
        assert out.count("fp32-read tensors identical to rank 0 before the master gather: True") == 2, out[-2000:]
        assert out.count("steps sharded vs allreduce") == 2, out[-2000:]


def test_two_ranks_on_one_gpu_e2e_sharded():
    _run(["--e2e", "--mode", "sharded"])


def test_module_mirror_under_distributed_data_parallel():
    """parallel.DistributedDataParallel around the VQA module mirror (vqa/function/train.py:327): gradients = the hand-averaged local
    gradients, identical replicas after two FusedAdamW steps with the fused clip, no_sync() semantics."""
    _run([], tool="dp2_mirror_check.py")


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_two_ranks_against_the_oracle(mode):
    """The oracle anchor of the multi-rank path (tests/dp2_oracle_check.py): two ranks, different batches, two optimizer steps --
    the reduced gradient against the mean of the oracle's per-rank gradients (pretrain/function/train.py:89-90), the weights against
    the oracle's clip + AdamW (common/trainer.py:139-145, common/nlp/bert/optimization.py:155-185), replicas identical."""
    out = _run(["--mode", mode], tool=os.path.join("tests", "dp2_oracle_check.py"), marker="ORACLE-ANCHORED DP OK=True")
    assert out.count("collectives emulated") == 2


@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_rccl_world_of_one_runs_the_native_exchange(mode):
    """RCCL on the 1-GPU box: one rank over `nccl` with VLB_DP_FORCE_EXCHANGE=1 -- every collective is the identity, but the code the
    8-GPU run depends on EXECUTES: native reduce_scatter_tensor / all_gather_into_tensor / all_reduce on the communicator's stream,
    the wire casts, the sharded clip + AdamW, the weight gather under the next forward, and engine.make_step_graph()'s hipGraph
    segments cut around the RCCL calls -- the same checks as the two-rank gloo run, plus the oracle anchor."""
    env = {"VLB_DP_FORCE_EXCHANGE": "1"}
    out = _run(["--mode", mode, "--backend", "nccl"], nproc=1, env_extra=env)
    assert out.count("backend nccl") == 1 and "collectives native" in out and "graph replay:" in out
    out = _run(["--mode", mode, "--backend", "nccl"], tool=os.path.join("tests", "dp2_oracle_check.py"), nproc=1, env_extra=env,
               marker="ORACLE-ANCHORED DP OK=True")
    assert "collectives native" in out


_COMM_SCRIPT = r"""
import ctypes, importlib, sys
import torch
sys.path.insert(0, %r)
L = importlib.import_module("vl-bert_amd._lib")
ops = importlib.import_module("vl-bert_amd.ops")
lib = L.load()
torch.cuda.set_device(0)
uid = ctypes.create_string_buffer(128)
assert lib.vlb_comm_unique_id(uid) == 0, lib.vlb_last_error()
comm = ctypes.c_void_p()
assert lib.vlb_comm_init(0, 1, uid, ctypes.byref(comm)) == 0, lib.vlb_last_error()
st = torch.cuda.current_stream().cuda_stream
g = torch.randn(1 << 20, device="cuda")
ref = g.clone()
assert lib.vlb_comm_allreduce_bucket(comm, g.data_ptr(), g.numel(), 0, st) == 0, lib.vlb_last_error()
w = torch.empty(g.numel(), dtype=ops.BF16, device="cuda")
ops.cast_f32_bf16(g, w)                                   # the 16-bit wire image of the bucket
wref = w.clone()
assert lib.vlb_comm_allreduce_bucket(comm, w.data_ptr(), w.numel(), 1, st) == 0, lib.vlb_last_error()
shard = torch.empty_like(w)
assert lib.vlb_comm_reduce_scatter_bucket(comm, w.data_ptr(), shard.data_ptr(), w.numel(), 1, st) == 0, lib.vlb_last_error()
full = torch.empty_like(w)
assert lib.vlb_comm_allgather_bucket(comm, shard.data_ptr(), full.data_ptr(), w.numel(), 1, st) == 0, lib.vlb_last_error()
torch.cuda.synchronize()
assert torch.equal(g, ref) and torch.equal(w, wref) and torch.equal(shard, wref) and torch.equal(full, wref)
assert lib.vlb_comm_allreduce_bucket(comm, g.data_ptr(), -1, 0, st) == -1 and b"bad count" in lib.vlb_last_error()
assert lib.vlb_comm_allreduce_bucket(None, g.data_ptr(), 4, 0, st) == -1
assert lib.vlb_comm_finalize(comm) == 0, lib.vlb_last_error()
assert lib.vlb_comm_finalize(None) == 0
print("VLB_COMM_OK")
"""


def test_c_abi_rccl_exchange_single_rank():
    """vlb_comm_* (include/vlbert_hip.h): the C host's RCCL binding -- unique id, communicator, in-place all-reduce of an
    fp32 bucket and of its 16-bit wire image, the reduce-scatter / all-gather halves of the sharded optimizer, argument errors,
    finalize -- with the one rank a 1-GPU box allows (a world of one is the identity; RCCL refuses two ranks on one device).  Runs in
    its own process: a communicator bound to the RCCL copy torch loaded."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _COMM_SCRIPT % ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:])
    print(r.stderr[-3000:])
    assert r.returncode == 0 and "VLB_COMM_OK" in r.stdout, r.stderr[-2000:]


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpus() < 2, reason="needs >= 2 GPUs: the N > 1 RCCL path (this box has %d)" % _gpus())
@pytest.mark.parametrize("mode", ["sharded", "allreduce"])
def test_two_ranks_over_rccl_one_gpu_each(mode):
    """First contact with RCCL at N > 1 wherever the suite runs on a multi-GPU box (auto-skipped on the 1-GPU boxes): one rank per
    GPU over `nccl`, native reduce-scatter / all-gather, the same checks as the shared-GPU gloo run -- reduced slices = summed local
    gradients, fp32-read tensors replicated, bit-identical replicas, sharded trajectory = all-reduce trajectory, graph replay."""
    out = _run(["--mode", mode, "--backend", "nccl"])
    assert out.count("backend nccl") == 2 and "collectives native" in out


@pytest.mark.skipif(_gpus() < 2, reason="needs >= 2 GPUs")
def test_bench_two_gpus_over_rccl_reports_the_exchange():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU): finishes with rc 0 and a JSON line that
    names RCCL, 2 ranks and the per-rank exposed-communication figures."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{")))
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["collective_backend"] == "RCCL (nccl)"
    assert not d["config"]["ranks_share_devices"]
    assert d["comm"] is not None and len(d["comm"]["exposed_comm_ms_per_rank"]) == 2
    assert d["value"] > 0 and d["config"]["global_batch"] == 256


def test_bench_multi_gpu_path_runs_over_rccl_in_a_world_of_one():
    """The code `bench.py --gpus N` will run on the 8-GPU node, executed here: VLB_DP_FORCE_EXCHANGE=1 makes a single rank form the RCCL
    communicator (probe all-reduce included), build the gradient buckets and the sharded optimizer, capture the step as hipGraph segments
    cut at the collectives, time it, and run the exposed-communication measurement -- every collective the identity.  rc 0, one JSON
    line that names RCCL, the segmented graph and the per-rank communication figures; the loss is finite."""
    import json
    import math
    env = dict(os.environ, VLB_DP_FORCE_EXCHANGE="1", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--global-batch", "32",
                        "--no-cpu-baseline", "--no-clock-probe"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2500:])
    print(r.stderr[-2500:])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{")))
    c = d["config"]
    assert c["forced_exchange"] is True and c["ranks"] == 1 and c["collective_backend"] == "RCCL (nccl)" and not c["ranks_share_devices"]
    assert c["dp_exchange"].startswith("sharded optimizer") and c["grad_wire_dtype"] == "bfloat16"
    assert isinstance(c["hipgraph"], str) and "graph segment" in c["hipgraph"], c["hipgraph"]
    assert d["comm"] is not None and len(d["comm"]["exposed_comm_ms_per_rank"]) == 1
    assert d["value"] > 0 and math.isfinite(d["loss"]) and d["roofline"]["achieved"] > 0


def _forced_bench(inject, extra=()):
    env = dict(os.environ, VLB_DP_FORCE_EXCHANGE="1", OMP_NUM_THREADS="4", VLB_BENCH_INJECT_FAIL=inject)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--global-batch", "32",
                           "--no-cpu-baseline", "--no-clock-probe"] + list(extra), env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


def test_bench_ladder_reaches_the_all_reduce_rung_and_names_it():
    """First-contact insurance of `bench.py --gpus N` (round 6), on the real engine over RCCL at world 1 (VLB_DP_FORCE_EXCHANGE): with the
    sharded + graph rung and the sharded eager rung made to fail after their warm-up steps (VLB_BENCH_INJECT_FAIL), the SAME invocation
    tears both engines down, runs the bucketed all-reduce eagerly with an fp32 wire and still prints its JSON line -- naming the rung that
    produced the number and the ones it gave up (config.dp_ladder)."""
    import json
    import math
    r = _forced_bench("sharded + segmented hipGraph;sharded, eager")
    print(r.stdout[-1500:])
    print(r.stderr[-2500:])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{")))
    c = d["config"]
    assert c["dp_ladder"]["rung"] == "all-reduce, eager, fp32 wire"
    assert [f["rung"] for f in c["dp_ladder"]["given_up"]] == ["sharded + segmented hipGraph", "sharded, eager"]
    assert all("injected failure" in f["reason"] for f in c["dp_ladder"]["given_up"])
    assert c["dp_exchange"].startswith("bucketed all-reduce") and c["grad_wire_dtype"] == "float32" and c["hipgraph"] is False
    assert d["value"] > 0 and math.isfinite(d["loss"])
    assert "given up" in r.stderr


def test_bench_ladder_exhausted_fails_loudly():
    """All three rungs failing: non-zero exit code and ONE line saying which rungs failed and why."""
    r = _forced_bench("sharded + segmented hipGraph;sharded, eager;all-reduce, eager, fp32 wire")
    print(r.stderr[-1500:])
    assert r.returncode != 0
    assert "bench.py: FAILED:" in r.stderr and "every configuration of the gradient exchange failed" in r.stderr


def test_bench_fails_loudly_when_the_communicator_cannot_form():
    """A rank whose peers never show up must end with a non-zero exit code and ONE line saying why -- not hang the driver's SCALE run:
    WORLD_SIZE = 2 with a single process, short rendezvous time-out."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               VLB_BENCH_COLLECTIVE_TIMEOUT_S="8", VLB_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    print(r.stderr[-1500:])
    assert r.returncode != 0
    assert "bench.py: FAILED:" in r.stderr and "communicator" in r.stderr
