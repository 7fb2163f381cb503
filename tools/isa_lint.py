#!/usr/bin/env python
"""Static checks on the gfx950 ISA hipcc emits for the library's kernels -- no GPU needed (hipcc cross-compiles).

Why: the last step-time gain of round 3 (-3.8 %) came from READING ISA, not from a profiler.  hipcc compiles a global load that sits
under a lane condition (`if (s < S) v = *p`) as a branch around the load with `s_waitcnt vmcnt(0)` right behind it, so every such load
becomes its own dependent HBM round trip; the attention backward's prologue was ten of them ("Serialised prologue
loads").  The same file-level view also shows register counts (occupancy contracts), spills, and scratch traffic or a drained
LDS-DMA queue inside an MFMA main loop.

    python tools/isa_lint.py [file.hip ...] [-D...]      per-kernel table (default: every vl-bert_amd/csrc/*.hip except the 6-minute gemm_p8.hip)

Used as a library by tests/test_isa_cpu.py, which pins the outcomes the hot kernels depend on.
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vl-bert_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only"]   # build.sh's code-generation flags

_LOAD = re.compile(r"^\s+(global|buffer|flat)_load_")
_WAIT = re.compile(r"^\s+s_waitcnt.*vmcnt\((\d+)\)")


def available():
    return os.path.isfile(HIPCC)


def compile_isa(path, defines=()):
    """ISA text of one .hip source (cached in the temp directory by content hash of the source, its local headers and the flags)."""
    h = hashlib.sha1()
    for p in [path] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "vlbert_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + list(defines)).encode())
    out = os.path.join(tempfile.gettempdir(), "vlb_isa_%s_%s.s" % (os.path.basename(path), h.hexdigest()[:16]))
    if not os.path.isfile(out):
        subprocess.run([HIPCC] + FLAGS + list(defines) + [path, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


def kernels(isa):
    """{mangled name: dict(body=[lines up to s_endpgm], vgpr=, agpr=, spill=, scratch=)} for every kernel of an ISA file."""
    meta = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?"
                         r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", isa):
        meta[m.group(2)] = dict(agpr=int(m.group(1)), scratch=int(m.group(3)), vgpr=int(m.group(4)), spill=int(m.group(5)))
    out = {}
    parts = re.split(r"\n(_Z\w+):\s", isa)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0].split("\n")
        if name in meta:
            out[name] = dict(meta[name], body=body)
    return out


def _vregs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def landing_register_violations(body):
    """Asynchronous LDS reads issued from inline asm WITHOUT a wait in the same statement (`ds_read*` inside an ASMSTART / ASMEND block)
    land in their destination registers some time later; the kernel retires them with an explicit `s_waitcnt lgkmcnt(..)` asm statement
    that lists those registers as in / out operands.  hipcc believes the values are there at once: if it copies, spills or reuses a
    landing register between the read and the wait, the copy holds stale data (seen once: a tied operand copied IN FRONT of the wait).
    Returns [(line index, instruction, register)] for every instruction outside asm blocks that touches a landing register while its
    read is in flight; [] is the contract."""
    queue, bad, in_asm = [], [], False      # in-flight asm reads in issue order: [set of destination registers]
    for i, l in enumerate(body):
        t = l.strip()
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        m = re.search(r"lgkmcnt\((\d+)\)", t) if t.startswith("s_waitcnt") else None
        if m:      # LDS operations complete in issue order: all but the N youngest reads have landed (other lgkm operations in flight can
            n = int(m.group(1))      # only make the real wait longer than this model assumes -- never shorter)
            queue = queue[len(queue) - n:] if n else []
            continue
        if in_asm:
            m = re.match(r"ds_read\w*\s+(v\[\d+:\d+\]|v\d+)", t)
            if m:
                queue.append(_vregs(m.group(1)))
            continue
        touched = _vregs(t)
        for regs in queue:
            if touched & regs:
                bad.append((i, t, min(touched & regs)))
                break
    return bad


def find(ks, pattern):
    """The one kernel whose mangled name matches `pattern` (regex)."""
    hits = [n for n in ks if re.search(pattern, n)]
    if len(hits) != 1:
        raise KeyError("pattern %r matches %d kernels: %s" % (pattern, len(hits), hits[:5]))
    return ks[hits[0]]


def is_load(line):
    return bool(_LOAD.match(line)) and " lds" not in line


def serialized_loads(body):
    """Waits `vmcnt(0)` with exactly ONE ordinary load issued since the previous vmcnt wait: each is a dependent round trip."""
    n, since = 0, 0
    for l in body:
        if is_load(l):
            since += 1
        m = _WAIT.match(l)
        if m:
            if int(m.group(1)) == 0 and since == 1:
                n += 1
            since = 0
    return n


def loads_before_first_wait(body, start=0):
    """Ordinary loads issued from line `start` on before the first vmcnt wait: the size of the kernel's first burst."""
    n = 0
    for l in body[start:]:
        if is_load(l):
            n += 1
        elif _WAIT.match(l):
            break
    return n


def first_line(body, needle, start=0):
    for i in range(start, len(body)):
        if needle in body[i]:
            return i
    return -1


def mfma_region(body):
    """(first, last) line index of the kernel's MFMAs, or None."""
    idx = [i for i, l in enumerate(body) if "v_mfma_" in l]
    return (idx[0], idx[-1]) if idx else None


def region_counts(body, lo, hi):
    seg = body[lo:hi + 1]
    return dict(scratch=sum(1 for l in seg if re.match(r"^\s+scratch_", l)),
                vmcnt0=sum(1 for l in seg if (_WAIT.match(l) and int(_WAIT.match(l).group(1)) == 0)),
                mfma=sum(1 for l in seg if "v_mfma_" in l),
                barriers=sum(1 for l in seg if re.match(r"^\s+s_barrier", l)),
                branches=sum(1 for l in seg if re.match(r"^\s+s_cbranch", l)),
                lds_dma=sum(1 for l in seg if re.match(r"^\s+buffer_load_dwordx4 .* lds", l)))


def report(path, defines=()):
    ks = kernels(compile_isa(path, defines))
    rows = []
    for name, k in sorted(ks.items()):
        b = k["body"]
        reg = mfma_region(b)
        inner = region_counts(b, *reg) if reg else None
        rows.append((name, k["vgpr"], k["agpr"], k["spill"], sum(1 for l in b if is_load(l)), serialized_loads(b), loads_before_first_wait(b), inner))
    return rows


def main(argv):
    defines = [a for a in argv if a.startswith("-D")]
    files = [a for a in argv if not a.startswith("-")]
    if not files:
        files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and f != "gemm_p8.hip")
    print("%-78s %5s %5s %5s %6s %10s %11s  %s" % ("kernel", "vgpr", "agpr", "spill", "loads", "serialised", "first burst", "MFMA region: mfma / barriers / vmcnt(0) / scratch"))
    for f in files:
        print("# " + os.path.relpath(f, ROOT))
        for name, vg, ag, sp, nl, ser, fb, inner in report(f, defines):
            tail = "%d / %d / %d / %d" % (inner["mfma"], inner["barriers"], inner["vmcnt0"], inner["scratch"]) if inner else "-"
            print("%-78s %5d %5d %5d %6d %10d %11d  %s" % (name[:78], vg, ag, sp, nl, ser, fb, tail))


if __name__ == "__main__":
    if not available():
        sys.exit("hipcc not found at %s" % HIPCC)
    main(sys.argv[1:])
