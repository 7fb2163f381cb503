#!/usr/bin/env python
"""What shader clock does the MI355X sustain under the step's GEMMs?  (Run on the GPU box.)

    python tools/clock_probe.py [seconds]

Keeps the device busy with the headline FFN GEMMs (vlb_gemm_nt_bf16, M = 25856) for a few seconds while a thread samples the shader
clock (sysfs pp_dpm_sclk, `rocm-smi --showclocks` as fall-back) and the socket power, then the same for an HBM-bound kernel (AdamW)
and for an idle device.  The MFMA peak in bench.py / DESIGN.md (2.5 PFLOP/s dense bf16) is quoted at the 2.4 GHz boost clock; the
fraction of it a kernel CAN reach is bounded by the clock the chip holds under that kernel's power draw."""
import glob
import importlib
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("vl-bert_amd.ops")


def sclk_sysfs():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f).read().splitlines():
                if "*" in line:
                    m = re.search(r"(\d+)\s*[Mm]hz", line)
                    if m:
                        out.append(int(m.group(1)))
        except OSError:
            pass
    return max(out) if out else None


def power_sysfs():
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            return int(open(f).read()) / 1e6
        except (OSError, ValueError):
            pass
    return None


def sclk_smi():
    try:
        t = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        m = re.findall(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*[Mm]hz", t)
        return max(int(x) for x in m) if m else None
    except Exception:
        return None


def sample_while(fn, seconds, label):
    stop, clk, pw = [False], [], []
    use_smi = sclk_sysfs() is None

    def poll():
        while not stop[0]:
            c = sclk_smi() if use_smi else sclk_sysfs()
            if c:
                clk.append(c)
            p = power_sysfs()
            if p:
                pw.append(p)
            time.sleep(0.05)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    el = time.time() - t0
    stop[0] = True
    th.join()
    clk2 = clk[len(clk) // 4:] or clk          # skip the ramp
    pw2 = pw[len(pw) // 4:] or pw
    print("%-34s %5d launches in %.1f s | sclk MHz: min %s avg %s max %s (%d samples, %s) | power W: avg %s max %s" %
          (label, n, el, min(clk2) if clk2 else None, int(sum(clk2) / len(clk2)) if clk2 else None, max(clk2) if clk2 else None, len(clk2),
           "rocm-smi" if use_smi else "sysfs", int(sum(pw2) / len(pw2)) if pw2 else None, int(max(pw2)) if pw2 else None), flush=True)
    return n, el, (sum(clk2) / len(clk2) if clk2 else None)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    D = "cuda:0"
    M, N, K = 25856, 768, 3072
    g = torch.Generator().manual_seed(0)
    A = (torch.rand((M, K), generator=g) * 2 - 1).to(torch.bfloat16).to(D)
    B = ((torch.rand((N, K), generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).to(D)
    C = torch.empty((M, N), dtype=torch.bfloat16, device=D)
    bias = torch.zeros(N, device=D)
    print("idle: sclk %s MHz, power %s W" % (sclk_sysfs() or sclk_smi(), power_sysfs()))
    n, el, clk = sample_while(lambda: ops.gemm_nt(A, B, C, bias=bias), secs, "p8 GEMM 25856x768x3072 (FFN2 fwd)")
    tf = n * 2.0 * M * N * K / el / 1e12
    if clk:
        print("   -> %.0f TFLOP/s incl. launch gaps = %.3f of the 2.5 PFLOP/s boost-clock peak, %.3f of the peak at the sustained %.0f MHz"
              % (tf, tf / 2500.0, tf / (2500.0 * clk / 2400.0), clk))
    # in-kernel evidence, independent of what sysfs reports: every workgroup of the large-tile kernel stamps the shader-clock counter
    # (s_memtime) and the constant 100 MHz real-time counter (s_memrealtime) at entry and exit (p8_ablate = 4, table passed as `pre`)
    lib = importlib.import_module("vl-bert_amd._lib")
    table = torch.zeros((256, 32), dtype=torch.bfloat16, device=D)           # 256 workgroups x 8 x int64 (a 2-D bf16 view: the `pre` argument)
    for _ in range(200):                                                      # bring the device to its sustained state first
        ops.gemm_nt(A, B, C, bias=bias)
    lib.gemm_set_option("p8_ablate", 4)
    ops.gemm_nt(A, B, C, bias=bias, pre=table)
    lib.gemm_set_option("p8_ablate", 0)
    torch.cuda.synchronize()
    t = table.view(torch.int64).view(256, 8).cpu()
    t = t[t[:, 3] > 0]
    if t.numel():
        cyc, rt = (t[:, 2] - t[:, 0]).double(), (t[:, 3] - t[:, 1]).double()
        mhz = cyc / rt * 100.0
        print("in-kernel (FFN2 shape, %d workgroups): kernel residency %.1f us, effective shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz: "
              "min %.0f median %.0f max %.0f MHz" % (t.shape[0], float(rt.median()) / 100.0, float(mhz.min()), float(mhz.median()), float(mhz.max())), flush=True)
    A2 = (torch.rand((8192, 8192), generator=g) * 2 - 1).to(torch.bfloat16).to(D)
    C2 = torch.empty((8192, 8192), dtype=torch.bfloat16, device=D)
    n, el, clk = sample_while(lambda: ops.gemm_nt(A2, A2, C2), secs, "p8 GEMM 8192^3")
    tf = n * 2.0 * 8192 ** 3 / el / 1e12
    if clk:
        print("   -> %.0f TFLOP/s = %.3f of the boost-clock peak, %.3f of the peak at the sustained %.0f MHz" % (tf, tf / 2500.0, tf / (2500.0 * clk / 2400.0), clk))
    n, el, clk = sample_while(lambda: torch.mm(A2, A2.t(), out=C2), secs, "hipBLASLt torch.mm 8192^3 (yardstick)")
    tf = n * 2.0 * 8192 ** 3 / el / 1e12
    if clk:
        print("   -> %.0f TFLOP/s = %.3f of the boost-clock peak, %.3f of the peak at the sustained %.0f MHz" % (tf, tf / 2500.0, tf / (2500.0 * clk / 2400.0), clk))
    P = torch.randn(64 << 20, device=D)
    Q = torch.empty_like(P)
    sample_while(lambda: Q.copy_(P), secs, "HBM copy 256 MB (memory-bound)")


if __name__ == "__main__":
    main()
