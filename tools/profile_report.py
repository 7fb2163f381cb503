#!/usr/bin/env python
"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/make_profiles.sh into the text summaries kept under profiles/.

    python tools/profile_report.py gpurun_out profiles r01 [steps_in_run]

Writes  profiles/<tag>_kernel_stats.txt   per-kernel time of the traced bench run (+ the bench JSON line of that run)
        profiles/<tag>_gemm_pmc.txt       SQ counters per GEMM kernel family, averaged per launch
        profiles/<tag>_hbm_traffic.txt    FETCH_SIZE / WRITE_SIZE per kernel family
        profiles/<tag>_gemm_traffic.json  HBM bytes per GEMM launch (read by bench.py for roofline.traffic)
FETCH_SIZE is doubled for the kernels that stream 16 B/lane (MI355X_MICROARCH.md, HBM: on gfx950 the counter tallies a
128-B request as 64 B); WRITE_SIZE is taken as reported (uncalibrated per the same guide).
"""
import glob
import json
import os
import re
import sqlite3
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
steps = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0     # 1 warm-up + 3 timed + 1 instrumented step


def db(name):
    return sqlite3.connect(glob.glob(os.path.join(src, name, "*.db"))[0])


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return re.sub(r"^void ", "", n)[:64]


def family(n):
    if "gemm_nt_p8" in n:
        return "gemm_nt_p8_kernel"            # large-tile 8-phase NT core (gemm_p8.hip): every epilogue / tile height
    if "gemm_tn8" in n:
        return "gemm_tn8_kernel"              # large-tile TN core (weight gradients, grouped per layer)
    if "gemm_nt_bf16" in n or "gemm_nt_256" in n or "gemm_nt_ring" in n:
        return "gemm_nt_bf16_kernel"          # (incl. the 256x256-tile instantiation used by the decoder)
    if "gemm_tn_bf16" in n:
        return "gemm_tn_bf16_kernel"
    return short(n)


# ---------------------------------------------------------------- kernel time
def kernel_stats(trace, out_name, cmd_note, steps=steps):
    cur = db(trace).cursor()
    rows = cur.execute("select name, end - start from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    line = ""
    for l in open(os.path.join(src, trace + ".log"), errors="replace"):
        if '"metric"' in l:
            line = l.strip()
    with open(os.path.join(dst, out_name), "w") as f:
        f.write(cmd_note)
        f.write("# (MI355X; %d steps in the process: 1 warm-up + 3 timed + 1 instrumented; the fill / copy kernels are mostly one-time\n"
                "# buffer setup; weight-gradient streams serialised so that per-kernel durations are not inflated by overlap --\n"
                "# the default bench run overlaps them)\n" % steps)
        f.write("# the Cijk_* kernel (60 launches in ONE step) is not part of the step: it is the torch.mm head start that bench.py queues in\n"
                "# front of its instrumented step so that the per-launch HIP events never include a wait for the host (bench.py --head-start)\n")
        f.write("# summarised from the rocpd sqlite output by tools/profile_report.py; bench line of the same (profiled) run:\n# %s\n" % line)
        f.write("%-66s %7s %10s %10s %9s %9s %9s %7s\n" % ("kernel", "calls", "total_ms", "ms/step", "avg_us", "min_us", "max_us", "share"))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write("%-66s %7d %10.3f %10.3f %9.2f %9.2f %9.2f %6.1f%%\n" % (k, a[0], a[1] / 1e6, a[1] / 1e6 / steps, a[1] / a[0] / 1e3,
                                                                          a[2] / 1e3, a[3] / 1e3, 100 * a[1] / total))
        f.write("%-66s %7s %10.3f %10.3f\n" % ("TOTAL kernel time", "", total / 1e6, total / 1e6 / steps))


if glob.glob(os.path.join(src, "final_trace", "*.db")):
    kernel_stats("final_trace", tag + "_kernel_stats.txt",
                 "# VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-phase-times\n")
if glob.glob(os.path.join(src, "final_e2e_trace", "*.db")):
    kernel_stats("final_e2e_trace", tag + "_e2e_kernel_stats.txt",
                 "# VLB_WGRAD_STREAM=0 VLB_VISION_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --e2e --steps 3 --warmup 1 "
                 "--no-graph --no-cpu-baseline --no-phase-times\n# (config C3: 8 images of 600x1000, ResNet-101 trunk + ROIAlign + dilated layer4 head + the VL-BERT step)\n")


if glob.glob(os.path.join(src, "final_large_trace", "*.db")):
    kernel_stats("final_large_trace", tag + "_large_kernel_stats.txt",
                 "# VLB_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --large --steps 3 --warmup 1 --no-graph --no-cpu-baseline "
                 "--no-phase-times\n# (VL-BERT-large shape of BASELINE configs 4-5 through the pretraining step: 24 x 1024, 128 text + 100 regions, batch 64)\n")
if glob.glob(os.path.join(src, "final_vcr_trace", "*.db")):
    kernel_stats("final_vcr_trace", tag + "_vcr_kernel_stats.txt",
                 "# rocprofv3 --kernel-trace --stats -- python bench.py --vcr --steps 2 --warmup 1 --no-cpu-baseline\n# (BASELINE config 5 through the "
                 "module mirror: 24 x 1024 encoder, 4 samples x 4 answer choices x 256 positions + 4 images of 600x1000 per micro-batch,\n# 4 micro-batches "
                 "per optimizer step; 1 warm-up + 2 timed + 1 instrumented optimizer step = 4 steps in the process; weight-gradient streams NOT serialised)\n",
                 steps=4.0)
if glob.glob(os.path.join(src, "final_vqa_trace", "*.db")):
    kernel_stats("final_vqa_trace", tag + "_vqa_kernel_stats.txt",
                 "# rocprofv3 --kernel-trace --stats -- python bench.py --vqa --steps 2 --warmup 1 --no-cpu-baseline\n# (BASELINE config 4's shape through "
                 "the module mirror, bf16: 24 x 1024 encoder, 16 samples x 229 positions per micro-batch, precomputed region features,\n# 4 micro-batches "
                 "per optimizer step, AdamW; 1 warm-up + 2 timed + 1 instrumented optimizer step = 4 steps in the process; weight-gradient streams NOT serialised)\n",
                 steps=4.0)
if len(sys.argv) > 5 and sys.argv[5] == "extra-only":
    sys.exit(0)


# ---------------------------------------------------------------- PMC helpers
def pmc(name):
    cur = db(name).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
    disp = {}
    for did, kn, cn, v, dur in rows:
        d = disp.setdefault(did, {"k": kn, "dur": dur, "c": {}})
        d["c"][cn] = d["c"].get(cn, 0.0) + v
    fam = {}
    for d in disp.values():
        a = fam.setdefault(family(d["k"]), {"n": 0, "dur": 0.0, "c": {}})
        a["n"] += 1; a["dur"] += d["dur"]
        for k, v in d["c"].items():
            a["c"][k] = a["c"].get(k, 0.0) + v
    return fam


def family_e2e(n):
    conv = ", true>" in n
    if "gemm_nt_p8" in n:
        return "gemm_nt_p8_kernel"
    if "gemm_tn8" in n:
        return "gemm_tn8_kernel"
    if "gemm_nt_bf16" in n or "gemm_nt_256" in n or "gemm_nt_ring" in n:
        return "gemm_nt_bf16_kernel<..,CONV>" if conv else "gemm_nt_bf16_kernel"
    if "gemm_tn_bf16" in n:
        return "gemm_tn_bf16_kernel<..,CONV>" if conv else "gemm_tn_bf16_kernel"
    return short(n)


def write_sq(sq_dir, out_name, what, gemm_keys):
    sq = pmc(sq_dir)
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc SQ_* (one separate pass) over %s; per kernel family, AVERAGE PER LAUNCH.\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and\n"
                "# SQ_LDS_* count cycles summed over SIMDs / CUs (MI355X_MICROARCH.md, PMC slots).\n" % what)
        names = sorted({k for a in sq.values() for k in a["c"]})
        f.write("%-34s %6s %9s " % ("kernel family", "calls", "avg_us") + " ".join("%24s" % n for n in names) + "\n")
        for k, a in sorted(sq.items(), key=lambda kv: -kv[1]["dur"])[:12]:
            f.write("%-34s %6d %9.1f " % (k[:34], a["n"], a["dur"] / a["n"] / 1e3) + " ".join("%24.4g" % (a["c"].get(n, 0) / a["n"]) for n in names) + "\n")
        for k in gemm_keys:
            if k in sq:
                c = sq[k]["c"]
                f.write("# %s: MFMA pipe busy %.0f %% of SIMD-cycles at the nominal 2.4 GHz (DVFS runs lower, so this is a lower bound) ; of the wave-cycles "
                        "%.0f %% parked on s_waitcnt/barrier (WAIT_ANY), %.0f %% issue-stalled (WAIT_INST_ANY), %.0f %% issuing ; "
                        "LDS bank-conflict cycles / LDS active = %.2f %%\n" % (
                            k, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / sq[k]["n"] / (1024 * sq[k]["dur"] / sq[k]["n"] * 2.4),
                            100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                            100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))


def write_traffic(fetch_dir, write_dir, out_name, what, top=14):
    fe, wr = pmc(fetch_dir), pmc(write_dir)
    traffic = {}
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE over %s.\n"
                "# Counters are in KB per dispatch; 'read' = 2 x FETCH_SIZE for the kernels that stream 16 B/lane (gfx950 correction,\n"
                "# MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported.  MB per launch (average), achieved GB/s = total / avg duration, MB per step.\n" % what)
        f.write("%-40s %6s %10s %10s %10s %10s %12s\n" % ("kernel family", "calls", "read MB", "write MB", "total MB", "GB/s", "MB per step"))
        for k, a in sorted(fe.items(), key=lambda kv: -kv[1]["dur"])[:top]:
            rd = 2.0 * a["c"].get("FETCH_SIZE", 0.0) / a["n"] / 1e3
            w = wr.get(k, {"c": {}, "n": 1})
            wm = w["c"].get("WRITE_SIZE", 0.0) / max(w["n"], 1) / 1e3
            gbs = (rd + wm) / 1e3 / (a["dur"] / a["n"] / 1e9) if a["dur"] > 0 else 0.0
            f.write("%-40s %6d %10.2f %10.2f %10.2f %10.0f %12.1f\n" % (k[:40], a["n"], rd, wm, rd + wm, gbs, (rd + wm) * a["n"] / steps))
            traffic[k] = {"launches_per_step": a["n"] / steps, "read_MB_per_launch": rd, "write_MB_per_launch": wm}
    return traffic


def gemm_sources_sha():
    """= bench.gemm_sources_sha (kept in step by tests/test_host_logic_cpu.py): content hash of the GEMM kernel sources."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vl-bert_amd", "csrc")
    for n in ("gemm.hip", "gemm_p8.hip", "gemm_tn8.hip", "gemm_params.h", "vlb_common.h"):
        with open(os.path.join(d, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


GEMM_FAMILIES = ("gemm_nt_p8_kernel", "gemm_tn8_kernel", "gemm_nt_bf16_kernel", "gemm_tn_bf16_kernel")
write_sq("final_sq", tag + "_gemm_pmc.txt", "the same bench command", GEMM_FAMILIES)
traffic = write_traffic("final_fetch", "final_write", tag + "_hbm_traffic.txt", "the same bench command")
g = [traffic[k] for k in GEMM_FAMILIES if k in traffic]
n = sum(t["launches_per_step"] for t in g)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/make_profiles.sh + tools/profile_report.py",
       "correction": "read = 2 x FETCH_SIZE (gfx950, 16 B/lane streaming reads); WRITE_SIZE as reported",
       "workload": "bench.py default (global batch 256, 1 GPU)",
       "commit": os.environ.get("VLB_COMMIT", "unknown"),
       "gemm_sources_sha": gemm_sources_sha(),      # bench.py quotes this file only when its own hash of the same files matches
       "gemm_launches_per_step": n,
       "gemm_hbm_GB_per_launch": sum((t["read_MB_per_launch"] + t["write_MB_per_launch"]) * t["launches_per_step"] for t in g) / n / 1e3,
       "families": traffic}
json.dump(out, open(os.path.join(dst, tag + "_gemm_traffic.json"), "w"), indent=1)
print("wrote", dst, tag, "GEMM GB/launch", out["gemm_hbm_GB_per_launch"])
if glob.glob(os.path.join(src, "final_e2e_sq", "*.db")):        # the e2e configuration: convolution / ROIAlign kernels
    family = family_e2e
    write_sq("final_e2e_sq", tag + "_e2e_pmc.txt", "python bench.py --e2e (config C3)",
             ("gemm_nt_bf16_kernel<..,CONV>", "gemm_tn_bf16_kernel<..,CONV>", "gemm_nt_p8_kernel", "gemm_tn8_kernel", "gemm_nt_bf16_kernel",
              "gemm_tn_bf16_kernel"))
    write_traffic("final_e2e_fetch", "final_e2e_write", tag + "_e2e_hbm_traffic.txt", "python bench.py --e2e (config C3)", top=18)
