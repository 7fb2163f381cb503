"""Helpers shared by the `-m gpu` parity tests."""
import importlib
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def pkg(name):
    return importlib.import_module("vl-bert_amd." + name)


def dev():
    return torch.device("cuda:0")


def act_dtype():
    """The library's 16-bit type: torch.bfloat16, or torch.float16 when the suite runs against the fp16 build (VLB_PRECISION=f16)."""
    return pkg("ops").BF16


def bf(t):
    """Round an fp32 CPU tensor to the library's 16-bit type (kept in fp32)."""
    return t.to(act_dtype()).float()


def to_gpu_bf16(t):
    return t.to(act_dtype()).to(dev())


def report(name, got, ref, atol, rtol):
    """max |got-ref| <= atol + rtol*max|ref|  (tensor-scale relative tolerance, bf16 friendly)."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = ref.abs().max().item() if ref.numel() else 0.0
    bad = not np.isfinite(err) or err > atol + rtol * scale
    line = "%-44s max_err %.3e  ref_max %.3e  tol %.3e  %s" % (name, err, scale, atol + rtol * scale, "FAIL" if bad else "ok")
    print(line)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert not bad, line


# ---- numpy re-statement of the device dropout RNG (vl-bert_amd/csrc/vlb_common.h) -------------
def _hash32(x):
    x = x.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x


def drop_thr(p):
    return 0 if p <= 0 else min(int(p * 65536.0 + 0.5), 65535)


def drop_scale(thr):
    return 65536.0 / (65536.0 - thr) if thr else 1.0


def _pair_hash(pair_idx, key):
    """vlb_pair_hash (vlb_common.h)"""
    m = np.uint64(0xFFFFFFFF)
    return _hash32(((pair_idx.astype(np.uint64) * np.uint64(0x9E3779B1)) + np.uint64(key)) & m)


def keep_mask(seed, tag, idx, thr):
    """idx: numpy integer array of element indices -> bool keep mask."""
    m = np.uint64(0xFFFFFFFF)
    idx = idx.astype(np.uint64)
    key = _hash32(np.array([(seed ^ ((tag * 0x85EBCA6B + 0x632BE5AB) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint64))[0]
    h = _pair_hash(idx >> np.uint64(1), key)
    bits = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return bits >= np.uint64(thr)
