"""CPU study behind vlb_pair_hash (vl-bert_amd/csrc/vlb_common.h): candidate dropout-mask hashes built from 24-bit multiplies (full-rate
VALU on CDNA) against the 32-bit-multiply hash they replace.  Metrics: avalanche bias, chi^2 of the two 16-bit fields over sequential
pair indices, keep-mask statistics of row-major [M, 768] and [., 101, 101] layouts (row / column count variance against binomial, lag
correlations), cross-key correlation, pair indices beyond 2^24.  H_E is the one shipped.   python tools/hash_quality.py"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF); M24 = np.uint64(0xFFFFFF)
def u(x): return np.uint64(x)
def mul24(a, b): return ((a & M24) * (u(b) & M24)) & M32
def mad24(a, b, c): return (mul24(a, b) + c) & M32
def hash32(x):
    x = x & M32
    x ^= x >> u(16); x = (x * u(0x7feb352d)) & M32; x ^= x >> u(15); x = (x * u(0x846ca68b)) & M32; x ^= x >> u(16)
    return x
def H_old(x, key): return hash32((x * u(0x9E3779B1) + key) & M32)

def H_A(x, key):      # 3 + 2 + 1 + 2 + 1 + 2 = 11 ops
    y = mad24(x, 0xB5297B, mad24(x >> u(24), 0x68E31D, key))
    y ^= y >> u(15)
    y = mul24(y, 0xD168AB) ^ (y & u(0xFF000000))   # placeholder, counts 2
    y ^= y >> u(13)
    y = mul24(y, 0xA4D3B5)
    y ^= y >> u(16)
    return y
def H_B(x, key):      # cheaper: 3 + 2 + 1 + 2 = 8 ops
    y = mad24(x, 0xB5297B, mad24(x >> u(24), 0x68E31D, key))
    y ^= y >> u(14)
    y = mul24(y, 0xD168AB)
    y ^= y >> u(15)
    return y
def H_C(x, key):      # 3 + 2 + 1 + 2 + 1 + 2 = 11, pure: two more rounds
    y = mad24(x, 0xB5297B, mad24(x >> u(24), 0x68E31D, key))
    y ^= y >> u(14)
    y = mul24(y, 0xD168AB)
    y ^= y >> u(13)
    y = mul24(y, 0xA4D3B5)
    y ^= y >> u(16)
    return y

def avalanche(H, nbits_in=28, n=200000, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 1 << nbits_in, n, dtype=np.uint64)
    key = u(rng.integers(0, 1 << 32))
    h0 = H(x, key)
    worst = 0; mat = np.zeros((nbits_in, 32))
    for i in range(nbits_in):
        d = h0 ^ H(x ^ u(1 << i), key)
        for o in range(32):
            mat[i, o] = ((d >> u(o)) & u(1)).mean()
    return np.abs(mat - 0.5).max(), np.abs(mat - 0.5).mean(), mat

def chi2_fields(H, n=1 << 22, seed=1):
    rng = np.random.default_rng(seed)
    key = u(rng.integers(0, 1 << 32))
    x = np.arange(n, dtype=np.uint64)
    h = H(x, key)
    out = []
    for f in (h & u(0xFFFF), h >> u(16)):
        cnt = np.bincount(f.astype(np.int64), minlength=65536)
        e = n / 65536
        out.append(((cnt - e) ** 2 / e).sum() / 65535)     # ~1 for uniform
    return out

def keep_stats(H, rows=4096, cols=768, thr=6554, seed=2):
    """2-D dropout mask: row / column keep counts against binomial variance; lag correlations"""
    rng = np.random.default_rng(seed)
    key = u(rng.integers(0, 1 << 32))
    idx = np.arange(rows * cols, dtype=np.uint64)
    h = H(idx >> u(1), key)
    bits = np.where((idx & u(1)) == 1, h >> u(16), h & u(0xFFFF))
    keep = (bits >= u(thr)).reshape(rows, cols).astype(np.float64)
    p = 1 - thr / 65536
    res = {"frac": keep.mean() - p}
    res["row_var_ratio"] = keep.sum(1).var() / (cols * p * (1 - p))
    res["col_var_ratio"] = keep.sum(0).var() / (rows * p * (1 - p))
    k = keep - p
    for lag in (1, 2, 3, 4, 8, 64, cols, cols + 1, 2 * cols):
        f = k.reshape(-1)
        res["lag%d" % lag] = (f[:-lag] * f[lag:]).mean() / (p * (1 - p))
    return res

def H_D(x, key):      # 11 ops: (shift, mad, mad) + (xorshift 2 + mul 1) x2 + xorshift 2
    y = mad24(x >> u(8), 0x68E31D, mad24(x, 0xB5297B, key))
    y ^= y >> u(14)
    y = mul24(y, 0xD168AB)
    y ^= y >> u(13)
    y = mul24(y, 0xA4D3B5)
    y ^= y >> u(16)
    return y
def H_E(x, key):      # like D but odd multipliers chosen with good bit patterns (from hash32 constants, truncated to 24 bits)
    y = mad24(x >> u(8), 0x6ca68b, mad24(x, 0xeb352d, key))
    y ^= y >> u(15)
    y = mul24(y, 0xca68b5 | 1)
    y ^= y >> u(12)
    y = mul24(y, 0x5352d7)
    y ^= y >> u(16)
    return y
def cross_key(H, n=1 << 21, thr=6554):
    rng = np.random.default_rng(5)
    x = np.arange(n, dtype=np.uint64)
    p = 1 - thr / 65536
    out = []
    for dk in (1, 2, 0x9E3779B1, 0x85ebca6b, 12345):
        k1 = u(rng.integers(0, 1 << 32)); k2 = (k1 + u(dk)) & M32
        a = ((H(x, k1) & u(0xFFFF)) >= u(thr)).astype(float) - p
        b = ((H(x, k2) & u(0xFFFF)) >= u(thr)).astype(float) - p
        out.append((a * b).mean() / (p * (1 - p)))
    return out
def big_idx(H, thr=6554):
    """pair indices beyond 2^24 (large config attention): keep fraction and lag correlation"""
    x = np.arange((1 << 24) - (1 << 20), (1 << 24) + (1 << 21), dtype=np.uint64)
    key = u(0x1234567)
    h = H(x, key)
    p = 1 - thr / 65536
    k = ((h >> u(16)) >= u(thr)).astype(float) - p
    return k.mean(), (k[:-1] * k[1:]).mean() / (p * (1 - p)), (k[:-(1 << 20)] * k[(1 << 20):]).mean() / (p * (1 - p))
for name in ("H_old", "H_C", "H_D", "H_E"):
    H = globals()[name]
    mx, mean, _ = avalanche(H, nbits_in=30, n=100000)
    print(name, "avalanche30 max %.4f mean %.4f" % (mx, mean), "chi2", ["%.2f" % c for c in chi2_fields(H)],
          "crosskey", ["%.4f" % c for c in cross_key(H)], "big", ["%.5f" % c for c in big_idx(H)])
    print("    S=101 layout", {k: round(float(v), 4) for k, v in keep_stats(H, rows=3072 * 8, cols=101).items() if "var" in k or k in ("lag1", "lag101", "frac")})
