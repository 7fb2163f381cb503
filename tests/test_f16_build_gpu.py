"""The fp16 build of the library (libvlbert_hip_f16.so: the same kernels with IEEE fp16 as the 16-bit activation / working-weight /
gradient type + a static loss scale -- the reference's own mixed precision, Apex O2 `TRAIN.FP16: true` with
`FP16_LOSS_SCALE`, pretrain/function/train.py:345-352) against the reference fixtures and the fp32 oracle.

One precision per process (vl-bert_amd/_lib.py), so the engine parity tests are re-run in a child process with VLB_PRECISION=f16.
What this buys over bf16: 3 more mantissa bits on every GEMM operand.  BASELINE.json configs 4-5 (VL-BERT-large, 24 x 1024) name fp32
and "mixed precision" -- not bf16 --: at that depth bf16 operand rounding alone reaches 1.1e-2 on the logits (measured,
test_engine_large_24_layers_s229_vs_oracle), the fp16 build stays an order of magnitude inside the 1e-2 bar."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ("golden or c1_shape or optimizer_step_matches or headline_c2 or large_24_layers or large_4_layers or batch_32 or degenerate or "
          "no_valid_mvrc or multitask_matches or maximum_sequence or mlm_head_compaction or dropout_training")


def test_engine_parity_suite_on_the_fp16_build():
    env = dict(os.environ, VLB_PRECISION="f16")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_engine_gpu.py"), "-m", "gpu", "-q", "-x", "-s", "-k", SELECT]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-6000:]
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "f16_suite.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    print(tail)
    print(r.stderr[-2000:])
    assert r.returncode == 0, tail[-3000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 20, r.stdout[-500:]
    # the depth-24 error the fp16 build reaches (printed by check_against_oracle): an order of magnitude inside the bar
    f = re.search(r"large 24-layer S=229 logits relative Frobenius error: mlm (\S+)\s+mvrc (\S+)", r.stdout)
    assert f, "24-layer test did not report"
    print("fp16 build, 24 x 1024: logits rel-Frobenius error mlm %s mvrc %s" % (f.group(1), f.group(2)))
    assert float(f.group(1)) <= 4e-3 and float(f.group(2)) <= 4e-3
