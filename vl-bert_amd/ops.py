"""Thin tensor-level wrappers over the C ABI (vl-bert_amd/_lib.py -> libvlbert_hip.so).

PyTorch is used for device memory and streams only: every function checks dtype/device,
passes `tensor.data_ptr()` and the current HIP stream, and returns.  No arithmetic is done by
torch here and there is no eager fallback -- a missing library or a CPU tensor raises.
"""
import torch

from . import _lib

BF16 = _lib.act_torch_dtype()      # the library's 16-bit type: torch.bfloat16, or torch.float16 for the VLB_PRECISION=f16 build (_lib.py)
ACT_NONE, ACT_GELU, ACT_RELU, ACT_DGELU, ACT_GELU_D, ACT_MULAUX, ACT_TANH = 0, 1, 2, 3, 4, 5, 6
OUT_BF16, OUT_F32, OUT_F32_ATOMIC = 0, 1, 2


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw-handle query is ~10x cheaper than building a Stream object:
    a step makes ~300 calls)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vl-bert_amd ops need GPU tensors (got %s); there is no CPU path" % t.device)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def _ld(t):
    """Leading dimension (elements) of a 2-D row-major view."""
    if t is None:
        return 0
    assert t.dim() == 2 and t.stride(1) == 1, "expected a 2-D tensor with unit inner stride"
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


F16 = torch.float16


def gemm_nt(A, B, C, bias=None, act=ACT_NONE, aux=None, pre=None, res=None, drop_p=0.0, seed=None, tag=0,
            out_mode=OUT_BF16, splitk=0, K=None, res_ln=None):
    """C[M,N] (+)= A[M,K] B[N,K]^T with the fused epilogue of vlb_gemm_nt_bf16.
    C of dtype float16 (16-bit output mode): the result is stored as IEEE fp16 (the encoder's pre-LayerNorm sums).
    res_ln = (stats [M,2], gamma [N], beta [N]): `res` holds fp16 pre-LayerNorm rows and the residual added is that LayerNorm's
    output re-materialised in fp32 (vlb_gemm_nt_bf16_ex)."""
    M, N = C.shape
    K = A.shape[1] if K is None else K
    assert A.shape[0] == M and B.shape[0] == N and B.shape[1] >= K and A.shape[1] >= K
    if res_ln is not None or C.dtype == F16:
        assert out_mode == OUT_BF16
        st, g, b = res_ln if res_ln is not None else (None, None, None)
        if res_ln is not None:
            assert res is not None and res.dtype == F16 and st.shape[0] >= M
        _lib.call("vlb_gemm_nt_bf16_ex", _p(A, BF16), _ld(A), _p(B, BF16), _ld(B), _p(C), _ld(C), M, N, K,
                  _p(bias, torch.float32), act, _p(aux, BF16), _ld(aux), _p(pre, BF16), _ld(pre), _p(res), _ld(res),
                  _p(st, torch.float32), _p(g, torch.float32), _p(b, torch.float32), 1 if C.dtype == F16 else 0,
                  float(drop_p), _p(seed), int(tag), out_mode, splitk, _stream())
        return C
    cdt = BF16 if out_mode == OUT_BF16 else torch.float32
    _lib.call("vlb_gemm_nt_bf16", _p(A, BF16), _ld(A), _p(B, BF16), _ld(B), _p(C, cdt), _ld(C), M, N, K,
              _p(bias, torch.float32), act, _p(aux, BF16), _ld(aux), _p(pre, BF16), _ld(pre), _p(res, BF16), _ld(res),
              float(drop_p), _p(seed), int(tag), out_mode, splitk, _stream())
    return C


def wgrad_nt(A, B, C, workspace=None):
    """C[M,N] (fp32) += A[M,K] B[N,K]^T  (slab split-K, no atomics)."""
    M, N = C.shape
    K = A.shape[1]
    assert A.shape[0] == M and B.shape[0] == N and B.shape[1] == K
    _lib.call("vlb_wgrad_nt_bf16", _p(A, BF16), _ld(A), _p(B, BF16), _ld(B), _p(C, torch.float32), _ld(C), M, N, K,
              _p(workspace, torch.float32), workspace.numel() if workspace is not None else 0, _stream())
    return C


def wgrad_tn(dy, x, C, colsum=None, workspace=None, accumulate=True):
    """C[Mo,No] (fp32) (+)= dy[R,Mo]^T x[R,No]  (+ colsum[Mo] += dy.sum(0)); operands as the passes left them.
    accumulate=False overwrites C (colsum is always accumulated)."""
    Mo, No = C.shape
    R = dy.shape[0]
    assert dy.shape[1] == Mo and x.shape[1] == No and x.shape[0] == R
    _lib.call("vlb_wgrad_tn_bf16", _p(dy, BF16), _ld(dy), _p(x, BF16), _ld(x), _p(C, torch.float32), _ld(C), R, Mo, No,
              _p(colsum, torch.float32), _p(workspace, torch.float32), workspace.numel() if workspace is not None else 0,
              int(bool(accumulate)), _stream())
    return C


class ZeroRanges:
    """Zeroes a fixed set of [lo, hi) ranges of one fp32 buffer in one launch (vlb_zero_ranges_f32)."""

    def __init__(self, base, ranges):
        self.base = base
        rows, starts, total = [], [0], 0
        for lo, hi in ranges:
            if hi <= lo:
                continue
            rows.append([lo, hi - lo])
            total += (hi - lo + 1023) // 1024
            starts.append(total)
        self.n, self.total = len(rows), total
        self.ranges = torch.tensor(rows, dtype=torch.int64).to(base.device) if rows else None
        self.starts = torch.tensor(starts, dtype=torch.int32).to(base.device)

    def run(self):
        if self.n:
            _lib.call("vlb_zero_ranges_f32", _p(self.base, torch.float32), self.ranges.data_ptr(), self.starts.data_ptr(), self.n,
                      self.total, _stream())


class CopyRanges:
    """dst[dst_start + i] = src[src_start + i] over a fixed table of (src_start, dst_start, length) rows in one launch
    (vlb_copy_ranges_f32): pack / unpack of the fp32-read parameters the sharded data-parallel optimizer replicates."""

    def __init__(self, rows, device):
        rows = [[int(a), int(b), int(n)] for a, b, n in rows if n > 0]
        starts, total = [0], 0
        for _, _, n in rows:
            total += (n + 1023) // 1024
            starts.append(total)
        self.rows, self.n, self.total = rows, len(rows), total
        self.ranges = torch.tensor(rows, dtype=torch.int64).reshape(-1, 3).to(device) if rows else None
        self.starts = torch.tensor(starts, dtype=torch.int32).to(device)

    def run(self, src, dst):
        if self.n:
            _lib.call("vlb_copy_ranges_f32", _p(src, torch.float32), _p(dst, torch.float32), self.ranges.data_ptr(), self.starts.data_ptr(),
                      self.n, self.total, _stream())


def wgrad_tn_group(items, workspace=None, accumulate=True):
    """items: up to 4 (dy [R,Mo], x [R,No], C [Mo,No] fp32, colsum [Mo] | None) over the same R rows -> one grouped launch
    (vlb_wgrad_tn_group_bf16)."""
    import ctypes
    n = len(items)
    R = items[0][0].shape[0]
    P, Lg, I = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    for dy, x, C, cs in items:
        assert dy.shape[0] == R and x.shape[0] == R and C.shape == (dy.shape[1], x.shape[1])
    A = P(*[_p(t[0], BF16) for t in items]); lda = Lg(*[_ld(t[0]) for t in items])
    B = P(*[_p(t[1], BF16) for t in items]); ldb = Lg(*[_ld(t[1]) for t in items])
    Cs = P(*[_p(t[2], torch.float32) for t in items]); ldc = Lg(*[_ld(t[2]) for t in items])
    Mo = I(*[t[0].shape[1] for t in items]); No = I(*[t[1].shape[1] for t in items])
    cs = P(*[_p(t[3], torch.float32) for t in items])
    _lib.call("vlb_wgrad_tn_group_bf16", n, A, lda, B, ldb, Cs, ldc, R, Mo, No, cs, _p(workspace, torch.float32),
              workspace.numel() if workspace is not None else 0, 1 if accumulate else 0, _stream())


class WgradTable:
    """Any number of weight gradients  C_i (+)= rowscale_i . (dy_i^T x_i)  in ONE launch of the large-tile core (vlb_wgrad_tn_table_*):
    items = [(dy [R_i, Mo] bf16, x [R_i, No] bf16, C [Mo, No] fp32, colsum [Mo] | None, rowscale [Mo] | None)], R_i % 128 == 0 (the
    operands' zero pad rows included).  Built once for fixed buffers; .ok is False when a product is outside what the kernel covers."""

    def __init__(self, items, device, accumulate=True):
        import ctypes
        lib = _lib.load()
        n = len(items)
        self.keep = list(items)
        P, Lg, I = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
        for dy, x, C, cs, rs in items:
            assert dy.shape[0] == x.shape[0] and C.shape == (dy.shape[1], x.shape[1]) and dy.dtype == BF16 and x.dtype == BF16
        A = P(*[_p(t[0], BF16) for t in items]); lda = Lg(*[_ld(t[0]) for t in items])
        B = P(*[_p(t[1], BF16) for t in items]); ldb = Lg(*[_ld(t[1]) for t in items])
        Cs = P(*[_p(t[2], torch.float32) for t in items]); ldc = Lg(*[_ld(t[2]) for t in items])
        R = I(*[t[0].shape[0] for t in items])
        Mo = I(*[t[0].shape[1] for t in items]); No = I(*[t[1].shape[1] for t in items])
        cs = P(*[_p(t[3], torch.float32) for t in items]); rs = P(*[_p(t[4], torch.float32) for t in items])
        nbytes = int(lib.vlb_wgrad_tn_table_desc_bytes()) * n
        host = ctypes.create_string_buffer(nbytes)
        rc = lib.vlb_wgrad_tn_table_pack(n, A, lda, B, ldb, Cs, ldc, R, Mo, No, cs, rs, 1 if accumulate else 0, host, nbytes)
        if rc < 0:
            raise RuntimeError("vlb_wgrad_tn_table_pack failed (%d): %s" % (rc, lib.vlb_last_error().decode()))
        self.ok, self.n, self.nitems = rc > 0, n, rc
        self.flops = sum(2.0 * t[0].shape[0] * t[0].shape[1] * t[1].shape[1] for t in items)
        self.nbytes = sum(t[0].numel() * 2 + t[1].numel() * 2 + t[2].numel() * 4 for t in items)
        self.desc = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(device) if self.ok else None

    def run(self):
        if not self.ok:
            raise RuntimeError("WgradTable: a product is outside what the table kernel covers (row counts must be multiples of 128)")
        _lib.call("vlb_wgrad_tn_table_launch", self.desc.data_ptr(), self.n, self.nitems, _stream())


def gemm_nt_splitk(A, B, C, workspace=None):
    """C (bf16) = A B^T with slab split-K when the output has too few tiles to fill the chip (long-K dgrad)."""
    M, K = A.shape
    N = B.shape[0]
    _lib.call("vlb_gemm_nt_bf16_splitk", _p(A, BF16), _ld(A), _p(B, BF16), _ld(B), _p(C, BF16), _ld(C), M, N, K,
              _p(workspace, torch.float32), workspace.numel() if workspace is not None else 0, _stream())
    return C


class TransposeBatch:
    """Descriptor table for vlb_transpose_batched_bf16: built once for fixed (src, dst) buffer pairs, replayed every step."""

    def __init__(self, pairs, device):
        desc, starts, total = [], [0], 0
        self.keep = list(pairs)                      # the tensors own the memory the raw pointers refer to
        for src, dst in pairs:
            R, C = src.shape
            assert dst.shape[0] == C and dst.shape[1] >= R and src.dtype == BF16 and dst.dtype == BF16
            desc.append([src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst), R, C])
            total += ((R + 63) // 64) * ((C + 63) // 64)
            starts.append(total)
        self.n, self.total = len(desc), total
        self.desc = torch.tensor(desc, dtype=torch.int64).to(device)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(device)

    def run(self):
        _lib.call("vlb_transpose_batched_bf16", self.desc.data_ptr(), self.starts.data_ptr(), self.n, self.total, _stream())


def ln_bwd_workspace_floats(H):
    return int(_lib.load().vlb_layernorm_bwd_workspace_floats(H))


def wgrad_workspace_floats(M, N, K):
    return int(_lib.load().vlb_wgrad_workspace_floats(M, N, K))


def transpose(x, out, colsum=None):
    """out[c, r] = x[r, c] (out may have a padded leading dimension)."""
    R, Cc = x.shape
    _lib.call("vlb_transpose_bf16", _p(x, BF16), _ld(x), _p(out, BF16), _ld(out), R, Cc, _p(colsum, torch.float32), _stream())
    return out


def layernorm_fwd(x, gamma, beta, y, stats=None, eps=1e-12, rows=None, ldx=None):
    """`rows`/`ldx` override the view-derived values (ldx=0 broadcasts one input row to every output row)."""
    H = x.shape[1]
    rows = x.shape[0] if rows is None else rows
    assert x.dtype in (BF16, F16)       # fp16: the encoder's pre-LayerNorm sums (gemm_nt with a float16 C)
    _lib.call("vlb_layernorm_fwd", _p(x), _ld(x) if ldx is None else ldx, _p(gamma, torch.float32), _p(beta, torch.float32), _p(y, BF16),
              _ld(y), _p(stats, torch.float32), rows, H, float(eps), 1 if x.dtype == F16 else 0, _stream())
    return y


def layernorm_bwd(dy, x, stats, gamma, dx=None, dx_drop=None, drop_p=0.0, seed=None, tag=0, dx_acc=None, dgamma=None,
                  dbeta=None, rows=None, ldx=None, ldacc=None, workspace=None, defer=False):
    """workspace: fp32 scratch tensor of ln_bwd_workspace_floats(H) elements, or None (direct atomics).
    defer=True (vlb_layernorm_bwd_deferred): the parameter-gradient partial sums stay in `workspace`; returns the number of partial
    vectors for ln_param_finalize_batch (0: they were added directly)."""
    H = x.shape[1]
    rows = x.shape[0] if rows is None else rows
    dy_f32 = 1 if dy.dtype == torch.float32 else 0
    assert x.dtype in (BF16, F16)
    _lib.call("vlb_layernorm_bwd_deferred" if defer else "vlb_layernorm_bwd", _p(dy), _ld(dy), dy_f32, _p(x),
              _ld(x) if ldx is None else ldx, _p(stats, torch.float32),
              _p(gamma, torch.float32), _p(dx, BF16), _ld(dx), _p(dx_drop, BF16), _ld(dx_drop), float(drop_p), _p(seed),
              int(tag), _p(dx_acc, torch.float32), _ld(dx_acc) if ldacc is None else ldacc, _p(dgamma, torch.float32),
              _p(dbeta, torch.float32), _p(workspace, torch.float32), rows, H, 1 if x.dtype == F16 else 0, _stream())
    return int(_lib.load().vlb_layernorm_bwd_slabs(rows)) if defer else None


def ln_param_finalize_batch(entries, H):
    """entries: [(workspace, n partial vectors, dgamma, dbeta)], at most 32 -> one launch (vlb_ln_param_finalize_batch)."""
    import ctypes
    n = len(entries)
    if not n:
        return
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    _lib.call("vlb_ln_param_finalize_batch", n, P(*[_p(e[0], torch.float32) for e in entries]), I(*[int(e[1]) for e in entries]),
              P(*[_p(e[2], torch.float32) for e in entries]), P(*[_p(e[3], torch.float32) for e in entries]), int(H), _stream())


def attention_fwd(qkv, mask, ctx, lse, B, S, H, nh, drop_p=0.0, seed=None, tag=0):
    _lib.call("vlb_attention_fwd", _p(qkv, BF16), _p(mask, torch.float32), _p(ctx, BF16), _p(lse, torch.float32), B, S, H, nh,
              float(drop_p), _p(seed), int(tag), _stream())
    return ctx


def attention_bwd(qkv, mask, ctx, lse, dctx, dqkv, B, S, H, nh, drop_p=0.0, seed=None, tag=0):
    _lib.call("vlb_attention_bwd", _p(qkv, BF16), _p(mask, torch.float32), _p(ctx, BF16), _p(lse, torch.float32),
              _p(dctx, BF16), _p(dqkv, BF16), B, S, H, nh, float(drop_p), _p(seed), int(tag), _stream())
    return dqkv


def seq_layout(text_mask, obj_mask, S):
    """masks: uint8/bool [B,T], [B,R] -> dict of layout tensors (all device int32 / fp32)."""
    B, T = text_mask.shape
    R = obj_mask.shape[1]
    dev = text_mask.device
    tm = text_mask.to(torch.uint8).contiguous()
    om = obj_mask.to(torch.uint8).contiguous()
    out = dict(code=torch.empty((B, S), dtype=torch.int32, device=dev), text_len=torch.empty((B,), dtype=torch.int32, device=dev),
               nobj=torch.empty((B,), dtype=torch.int32, device=dev), text_rows=torch.empty((B, T), dtype=torch.int32, device=dev),
               obj_rows=torch.empty((B, R), dtype=torch.int32, device=dev), attn_mask=torch.empty((B, S), dtype=torch.float32, device=dev))
    seq_layout_into(tm, om, S, out)
    return out


def seq_layout_into(tm, om, S, out):
    B, T = tm.shape
    R = om.shape[1]
    _lib.call("vlb_seq_layout", _p(tm, torch.uint8), _p(om, torch.uint8), B, T, R, S, _p(out["code"]), _p(out["text_len"]),
              _p(out["nobj"]), _p(out["text_rows"]), _p(out["obj_rows"]), _p(out["attn_mask"]), _stream())


def obj_prep_fwd(boxes, im_info, mvrc_ops, mask_emb, out, drop_p=0.0, seed=None, tag=0):
    B, R, ldb = boxes.shape
    assert boxes.is_contiguous() and out.shape[-1] == 4096
    # im_info rows are (width, height, ...): 5 columns from the pre-training / VCR datasets, 4 from VQA's (vqa/data/datasets/vqa.py:217)
    assert im_info.dim() == 2 and im_info.shape[0] == B and im_info.shape[1] >= 2 and im_info.stride(1) == 1, tuple(im_info.shape)
    _lib.call("vlb_obj_prep_fwd", _p(boxes, torch.float32), ldb, _p(im_info, torch.float32), im_info.stride(0), _p(mvrc_ops, torch.int64),
              _p(mask_emb, torch.float32), _p(out, BF16), B, R, float(drop_p), _p(seed), int(tag), _stream())
    return out


def zero_padded_rows(x, boxes):
    """x bf16 [B*R, H]: rows whose box (boxes fp32 [B,R,ld], x1 <= -1.5) is padding become 0."""
    B, R, ldb = boxes.shape
    _lib.call("vlb_zero_padded_rows_bf16", _p(x, BF16), _ld(x), _p(boxes, torch.float32), ldb, B * R, x.shape[1], _stream())
    return x


def masked_colsum(src, sel, dst, drop_p=0.0, seed=None, tag=0, row_elems=0, col_off=0):
    rows, C = src.shape
    _lib.call("vlb_masked_colsum", _p(src, BF16), _ld(src), _p(sel, torch.int64), rows, C, _p(dst, torch.float32), float(drop_p),
              _p(seed), int(tag), int(row_elems), int(col_off), _stream())


def embed_fwd(lay, text_ids, text_type, word_emb, pos_emb, type_emb, end_emb, text_vis, tv_strides, obj_vis, ov_strides,
              obj_ling, ol_strides, obj_ling_idx, gamma, beta, pre, stats, out, B, T, R, S, H, eps=1e-12, drop_p=0.0,
              seed=None, tag=0):
    _lib.call("vlb_embed_fwd", _p(lay["code"]), _p(lay["text_len"]), _p(text_ids, torch.int64), _p(text_type, torch.int64),
              _p(word_emb, BF16), _p(pos_emb, BF16), _p(type_emb, BF16), _p(end_emb, BF16),
              _p(text_vis, BF16), tv_strides[0], tv_strides[1], _p(obj_vis, BF16), ov_strides[0], ov_strides[1],
              _p(obj_ling, BF16), ol_strides[0], ol_strides[1], _p(obj_ling_idx, torch.int64),
              _p(gamma, torch.float32), _p(beta, torch.float32), _p(pre, BF16), _p(stats, torch.float32), _p(out, BF16),
              B, T, R, S, H, word_emb.shape[0], pos_emb.shape[0], float(eps), float(drop_p), _p(seed), int(tag), _stream())
    return out


def embed_bwd(dy, pre, stats, gamma, lay, text_ids, text_type, obj_ling_idx, d_word, d_pos, d_type, d_end, d_gamma, d_beta,
              d_text_vis, dtv_strides, d_obj_vis, dov_strides, d_obj_ling, dol_strides, B, T, R, S, H, drop_p=0.0, seed=None,
              tag=0, text_vis_zeroed=False):
    _lib.call("vlb_embed_bwd", _p(dy, BF16), _p(pre, BF16), _p(stats, torch.float32), _p(gamma, torch.float32), _p(lay["code"]),
              _p(lay["text_len"]), _p(text_ids, torch.int64), _p(text_type, torch.int64), _p(obj_ling_idx, torch.int64),
              _p(d_word, torch.float32), _p(d_pos, torch.float32), _p(d_type, torch.float32), _p(d_end, torch.float32),
              _p(d_gamma, torch.float32), _p(d_beta, torch.float32),
              _p(d_text_vis, torch.float32), dtv_strides[0], dtv_strides[1], _p(d_obj_vis, torch.float32), dov_strides[0],
              dov_strides[1], _p(d_obj_ling, torch.float32), dol_strides[0], dol_strides[1],
              B, T, R, S, H, d_word.shape[0], d_pos.shape[0], float(drop_p), _p(seed), int(tag), int(bool(text_vis_zeroed)),
              _stream())


def gather_rows(src, idx, out):
    n, H = out.shape
    _lib.call("vlb_gather_rows", _p(src, BF16), _p(idx, torch.int32), _p(out, BF16), n, H, _stream())
    return out


def head_grad_combine(d_text, d_obj, code, dx, B, T, R, S, H):
    _lib.call("vlb_head_grad_combine", _p(d_text, BF16), _p(d_obj, BF16), _p(code, torch.int32), _p(dx, BF16), B, T, R, S, H,
              _stream())
    return dx


def relu_bwd_cast(g, y, out):
    _lib.call("vlb_relu_bwd_cast", _p(g, torch.float32), _p(y, BF16), _p(out, BF16), g.numel(), _stream())
    return out


def dgelu_mul(dg, u, out):
    _lib.call("vlb_dgelu_mul", _p(dg, BF16), _p(u, BF16), _p(out, BF16), dg.numel(), _stream())
    return out


def tanh_bwd(dy, y, out):
    _lib.call("vlb_tanh_bwd", _p(dy, BF16), _p(y, BF16), _p(out, BF16), dy.numel(), _stream())
    return out


def mul_bf16(a, b, out):
    _lib.call("vlb_mul_bf16", _p(a, BF16), _p(b, BF16), _p(out, BF16), a.numel(), _stream())
    return out


def ce_fwd_bwd(logits, V, labels, counts, loss_out, gscale=1.0, logits_copy=None):
    rows = logits.shape[0]
    _lib.call("vlb_ce_fwd_bwd", _p(logits, BF16), _ld(logits), rows, V, _p(labels, torch.int64), _p(counts, torch.float32),
              float(gscale), _p(loss_out, torch.float32), _p(logits_copy, BF16), _ld(logits_copy), _stream())


def mlm_compact(labels, src_rows, n_split, V, sel_pos, sel_src, labels_c, count0, count1, overflow):
    """Lists the labelled positions (vlb_mlm_compact); capacity = sel_pos.numel()."""
    _lib.call("vlb_mlm_compact", _p(labels, torch.int64), _p(src_rows, torch.int32), labels.numel(), int(n_split), int(V), sel_pos.numel(),
              _p(sel_pos, torch.int32), _p(sel_src, torch.int32), _p(labels_c, torch.int64), _p(count0, torch.float32),
              _p(count1, torch.float32), _p(overflow, torch.int32), _stream())


def ce_fwd_bwd_compact(logits, V, labels_c, count0, count1, loss_out0, loss_out1, gscale=1.0):
    _lib.call("vlb_ce_fwd_bwd_compact", _p(logits, BF16), _ld(logits), logits.shape[0], V, _p(labels_c, torch.int64),
              _p(count0, torch.float32), _p(count1, torch.float32), float(gscale), _p(loss_out0, torch.float32),
              _p(loss_out1, torch.float32), _stream())


def scatter_rows(src, idx, out):
    _lib.call("vlb_scatter_rows", _p(src, BF16), _p(idx, torch.int32), _p(out, BF16), idx.numel(), src.shape[1], _stream())
    return out


def soft_ce_fwd_bwd(logits, C, target, tsum, counts, loss_out, gscale=1.0, logits_copy=None):
    rows = logits.shape[0]
    _lib.call("vlb_soft_ce_fwd_bwd", _p(logits, BF16), _ld(logits), rows, C, _p(target, torch.float32), _ld(target),
              _p(tsum, torch.float32), _p(counts, torch.float32), float(gscale), _p(loss_out, torch.float32),
              _p(logits_copy, BF16), _ld(logits_copy), _stream())


def bce_logits_fwd_bwd(logits, A, label, loss_out, gscale=1.0, logits_copy=None, pos_weight=1.0):
    """logits bf16 [rows, >=A] <- gscale * w * (sigmoid(x) - y) / rows in place; loss_out += BCE-with-logits * A (reference convention);
    w = pos_weight on the positive labels (VCR), 1 elsewhere."""
    rows = logits.shape[0]
    _lib.call("vlb_bce_logits_fwd_bwd", _p(logits, BF16), _ld(logits), rows, A, _p(label, torch.float32), _ld(label), float(gscale),
              float(pos_weight), _p(loss_out, torch.float32), _p(logits_copy, BF16), _ld(logits_copy), _stream())


def dropout_bf16(x, y, drop_p, seed, tag):
    _lib.call("vlb_dropout_bf16", _p(x, BF16), _p(y, BF16), x.numel(), float(drop_p), _p(seed), int(tag), _stream())
    return y


def sumsq(g, out):
    _lib.call("vlb_sumsq_f32", _p(g, torch.float32), g.numel(), _p(out, torch.float32), _stream())


def sumsq_det(g, partials, out):
    """out += sum(g^2) with a fixed summation order (same bits on every data-parallel rank); g fp32, or the bf16 wire image."""
    if g.dtype == BF16:
        _lib.call("vlb_sumsq_bf16_det", _p(g, BF16), g.numel(), _p(partials, torch.float32), partials.numel(), _p(out, torch.float32),
                  _stream())
        return
    _lib.call("vlb_sumsq_f32_det", _p(g, torch.float32), g.numel(), _p(partials, torch.float32), partials.numel(), _p(out, torch.float32),
              _stream())


def adamw_step(p, g, m, v, p16, state, grad_scale=1.0):
    if g.dtype == BF16:
        _lib.call("vlb_adamw_step_gbf16", _p(p, torch.float32), _p(g, BF16), _p(m, torch.float32), _p(v, torch.float32),
                  _p(p16, BF16), p.numel(), _p(state, torch.float32), float(grad_scale), _stream())
        return
    _lib.call("vlb_adamw_step", _p(p, torch.float32), _p(g, torch.float32), _p(m, torch.float32), _p(v, torch.float32),
              _p(p16, BF16), p.numel(), _p(state, torch.float32), float(grad_scale), _stream())


def sgd_momentum_step(p, g, buf, lr, momentum=0.9, weight_decay=0.0, p16=None, sumsq=None, max_norm=0.0, grad_scale=1.0):
    """torch.optim.SGD(lr, momentum, weight_decay) on flat fp32 tensors (vlb_sgd_momentum_step); sumsq: device scalar for the clip."""
    _lib.call("vlb_sgd_momentum_step", _p(p, torch.float32), _p(g, torch.float32), _p(buf, torch.float32), _p(p16, BF16), p.numel(),
              float(lr), float(momentum), float(weight_decay), _p(sumsq, torch.float32), float(max_norm), float(grad_scale), _stream())


LR_CONSTANT, LR_WARMUP_CONSTANT, LR_WARMUP_LINEAR = 0, 1, 2


def lr_schedule_step(state, kind, base_lr, warmup_steps=0, t_total=0):
    """state[0] = base_lr * lambda(steps_taken + 1) on the device (common/nlp/bert/optimization.py:27-62)."""
    _lib.call("vlb_lr_schedule_step", _p(state, torch.float32), int(kind), float(base_lr), float(warmup_steps), float(t_total),
              _stream())


def scale_f32(x, alpha):
    _lib.call("vlb_scale_f32", _p(x, torch.float32), x.numel(), float(alpha), _stream())
    return x


def cast_f32_bf16(src, dst):
    _lib.call("vlb_cast_f32_bf16", _p(src, torch.float32), _p(dst, BF16), src.numel(), _stream())
    return dst


def cast_bf16_f32(src, dst):
    _lib.call("vlb_cast_bf16_f32", _p(src, BF16), _p(dst, torch.float32), src.numel(), _stream())
    return dst


def rank_seed(base):
    """Odd per-rank dropout seed for the nn.Module mirrors: data-parallel replicas must not share dropout masks."""
    import torch.distributed as dist
    r = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return (int(base) | 1) + 2 * r


def rng_advance(seed):
    _lib.call("vlb_rng_advance", _p(seed), _stream())


def roi_align_fwd(inp, rois, out, spatial_scale, sampling_ratio):
    K, C, ph, pw = out.shape
    _lib.call("vlb_roi_align_fwd", _p(inp, torch.float32), _p(rois, torch.float32), _p(out, torch.float32), K, C, inp.shape[2],
              inp.shape[3], ph, pw, float(spatial_scale), int(sampling_ratio), _stream())
    return out


def roi_align_bwd(grad_out, rois, grad_in, spatial_scale, sampling_ratio):
    K, C, ph, pw = grad_out.shape
    Bn, _, Hh, Ww = grad_in.shape
    _lib.call("vlb_roi_align_bwd", _p(grad_out, torch.float32), _p(rois, torch.float32), _p(grad_in, torch.float32), K, Bn, C, Hh,
              Ww, ph, pw, float(spatial_scale), int(sampling_ratio), _stream())
    return grad_in


# ---------------------------------------------------------------------------------------------------------------
# end-to-end vision path (NHWC bf16): see csrc/vision.hip
# ---------------------------------------------------------------------------------------------------------------
ACT_RES_RELU, ACT_RELU_MASK = 7, 8     # relu(acc + bias + res) ; (acc [+ res]) where aux > 0


def conv_out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_weight_prepare(w, bn, wf, wb=None, scale=None, shift=None, eps=1e-5):
    """w fp32 [O, taps, I] (+ bn = (gamma, beta, mean, var) or None) -> wf bf16 [O, kf], wb bf16 [I, taps*O], scale/shift [O]."""
    O, taps, I = w.shape
    g, b, m, v = bn if bn is not None else (None, None, None, None)
    _lib.call("vlb_conv_weight_prepare", _p(w, torch.float32), _p(g, torch.float32), _p(b, torch.float32), _p(m, torch.float32),
              _p(v, torch.float32), float(eps), _p(wf, BF16), _p(wb, BF16), _p(scale, torch.float32), _p(shift, torch.float32),
              O, I, taps, wf.shape[1], _stream())


class ConvPrepareBatch:
    """vlb_conv_weight_prepare for a fixed list of convolutions in one launch: items = [(w [O,taps,I], bn 4-tuple or None, wf, wb,
    scale, shift)], built once, replayed after every optimizer step."""

    def __init__(self, items, device, eps=1e-5):
        self.keep = list(items)
        desc, starts, total = [], [], 0
        ptr = lambda t: 0 if t is None else t.data_ptr()
        for w, bn, wf, wb, scale, shift in items:
            O, taps, I = w.shape
            g, b, m, v = bn if bn is not None else (None, None, None, None)
            desc.append([ptr(w), ptr(g), ptr(b), ptr(m), ptr(v), ptr(wf), ptr(wb), ptr(scale), ptr(shift), O, I, taps, wf.shape[1]])
            starts.append(total)
            total += (O * taps * I + 1023) // 1024
        self.n, self.total, self.eps = len(desc), total, eps
        self.desc = torch.tensor(desc, dtype=torch.int64).to(device) if desc else None
        self.starts = torch.tensor(starts, dtype=torch.int32).to(device) if desc else None

    def run(self):
        if self.n:
            _lib.call("vlb_conv_weight_prepare_batched", self.desc.data_ptr(), self.starts.data_ptr(), self.n, self.total, float(self.eps),
                      _stream())


def conv_wgrad_finalize(dwf, scale, g, accumulate=True):
    """g fp32 [O, kreal] (+)= scale[o] * dwf[O, kf][:, :kreal]"""
    O, kreal = g.shape
    _lib.call("vlb_conv_wgrad_finalize", _p(dwf, torch.float32), _p(scale, torch.float32), _p(g, torch.float32), O, kreal,
              dwf.shape[1], int(bool(accumulate)), _stream())


def conv3x3_nhwc(x, w, y, N, H, W, C, dil, zero16, bias=None, act=ACT_NONE, aux=None):
    """y[N*H*W, O] = epi(implicit im2col(x) . w^T): 3x3, stride 1, padding = dilation; w bf16 [O, 9C] tap-major."""
    O = w.shape[0]
    assert x.shape[0] == N * H * W and x.shape[1] == C and x.is_contiguous() and y.shape[0] == x.shape[0] and y.shape[1] == O
    _lib.call("vlb_conv3x3_nhwc_bf16", _p(x, BF16), N, H, W, C, dil, _p(w, BF16), _ld(w), _p(y, BF16), _ld(y), O,
              _p(bias, torch.float32), act, _p(aux, BF16), _ld(aux), _p(zero16, BF16), _stream())
    return y


def conv3x3_wgrad_tn(dy, x, dW, N, H, W, C, dil, workspace=None, accumulate=True, rowscale=None):
    """dW fp32 [O, 9C] (+)= rowscale[o] * dy[N*H*W, O]^T . im2col(x) with the gather inside the TN GEMM (C % 128 == 0)."""
    O = dW.shape[0]
    assert dy.shape[0] == N * H * W and dy.shape[1] == O and x.shape[0] == dy.shape[0] and x.shape[1] == C and x.is_contiguous()
    _lib.call("vlb_conv3x3_wgrad_tn_bf16", _p(dy, BF16), _ld(dy), _p(x, BF16), N, H, W, C, dil, _p(dW, torch.float32), _ld(dW), O,
              _p(rowscale, torch.float32), _p(workspace, torch.float32), workspace.numel() if workspace is not None else 0,
              int(bool(accumulate)), _stream())
    return dW


def wgrad_tn_rowscale(dy, x, C, rowscale, workspace, accumulate=True):
    """C[Mo,No] (fp32) (+)= rowscale[m] * (dy[R,Mo]^T x[R,No])"""
    Mo, No = C.shape
    R = dy.shape[0]
    assert dy.shape[1] == Mo and x.shape[1] == No and x.shape[0] == R
    _lib.call("vlb_wgrad_tn_rowscale_bf16", _p(dy, BF16), _ld(dy), _p(x, BF16), _ld(x), _p(C, torch.float32), _ld(C), R, Mo, No,
              _p(rowscale, torch.float32), _p(workspace, torch.float32), workspace.numel(), int(bool(accumulate)), _stream())
    return C


def im2col_nhwc(x, col, N, H, W, C, k, stride, pad, dil):
    _lib.call("vlb_im2col_nhwc_bf16", _p(x, BF16), _p(col, BF16), _ld(col), N, H, W, C, k, k, stride, pad, dil, _stream())
    return col


def im2col_image(img, col, k=7, stride=2, pad=3):
    N, Cin, H, W = img.shape
    _lib.call("vlb_im2col_image_f32", _p(img, torch.float32), _p(col, BF16), col.shape[1], N, Cin, H, W, k, k, stride, pad, _stream())
    return col


def mask_image_boxes(image, boxes, mvrc_ops):
    """image [N,3,H,W] fp32 (in place): zero the pixels of every box whose mvrc_op is 1 (conceptual_captions.py:201-206)."""
    N, C, H, W = image.shape
    assert boxes.shape[0] == N and mvrc_ops.shape == boxes.shape[:2] and mvrc_ops.dtype == torch.int64 and image.is_contiguous()
    assert boxes.stride(2) == 1 and boxes.stride(0) == boxes.shape[1] * boxes.stride(1) and mvrc_ops.is_contiguous()
    _lib.call("vlb_mask_image_boxes_f32", _p(image, torch.float32), N, C, H, W, _p(boxes, torch.float32), boxes.stride(1), boxes.shape[1],
              mvrc_ops.data_ptr(), _stream())


def maxpool3x3s2_nhwc(x, y, N, H, W, C):
    _lib.call("vlb_maxpool3x3s2_nhwc", _p(x, BF16), _p(y, BF16), N, H, W, C, _stream())
    return y


def subsample2_nhwc(x, y, N, H, W, C):
    _lib.call("vlb_subsample2_nhwc", _p(x, BF16), _p(y, BF16), N, H, W, C, _stream())
    return y


def upsample2_zero_nhwc(dy, dx, N, H, W, C):
    _lib.call("vlb_upsample2_zero_nhwc", _p(dy, BF16), _p(dx, BF16), N, H, W, C, _stream())
    return dx


def roi_align_nhwc_fwd(feat, boxes, boxes_per_image, out, N, H, W, C, pooled=14, spatial_scale=1.0 / 16, sampling_ratio=1):
    """feat bf16 [N*H*W, C]; boxes fp32 [K, ld] (x1,y1,x2,y2 first; x1 <= -1.5 = padding); out bf16 [K*pooled*pooled, C]."""
    K = boxes.shape[0]
    _lib.call("vlb_roi_align_nhwc_fwd", _p(feat, BF16), _p(boxes, torch.float32), _ld(boxes), boxes_per_image, _p(out, BF16), K, C,
              H, W, pooled, pooled, float(spatial_scale), sampling_ratio, _stream())
    return out


def roi_align_nhwc_bwd(dout, boxes, boxes_per_image, dfeat, N, H, W, C, pooled=14, spatial_scale=1.0 / 16, sampling_ratio=1):
    K = boxes.shape[0]
    _lib.call("vlb_roi_align_nhwc_bwd", _p(dout, BF16), _p(boxes, torch.float32), _ld(boxes), boxes_per_image,
              _p(dfeat, torch.float32), K, N, C, H, W, pooled, pooled, float(spatial_scale), sampling_ratio, _stream())
    return dfeat


def roi_align_gather_workspace(K, H, W, pooled, device):
    n = _lib.load().vlb_roi_align_gather_workspace_bytes(K, H, W, pooled, pooled)
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def roi_align_nhwc_bwd_gather(dout, boxes, boxes_per_image, workspace, N, H, W, C, act=None, dx_bf16=None, dx_f32=None, pooled=14,
                              spatial_scale=1.0 / 16, sampling_ratio=1):
    """ROIAlign backward as a gather per feature pixel (vlb_roi_align_nhwc_bwd_gather): dx_bf16 (masked by act > 0) and / or dx_f32."""
    assert dx_bf16 is not None or dx_f32 is not None
    _lib.call("vlb_roi_align_nhwc_bwd_gather", _p(dout, BF16), _p(boxes, torch.float32), _ld(boxes), boxes_per_image,
              _p(act, BF16) if act is not None else None, _p(dx_bf16, BF16) if dx_bf16 is not None else None,
              _p(dx_f32, torch.float32) if dx_f32 is not None else None, _p(workspace, torch.uint8), workspace.numel(), N, C, H, W,
              pooled, pooled, float(spatial_scale), sampling_ratio, _stream())
    return dx_bf16 if dx_bf16 is not None else dx_f32


def relu_mask_cast(g, y, dz):
    _lib.call("vlb_relu_mask_cast", _p(g, torch.float32), _p(y, BF16), _p(dz, BF16), g.numel(), _stream())
    return dz


def avgpool_rows_fwd(y, out, col0, K, P, C, pad_col=-1, segm=None):
    """y bf16 [K*P, C] -> out fp32 [K, ld][:, col0:col0+C] = mean over the P pixels of y (* segm [K,P]) (0 where out[:, pad_col] <= -1.5)"""
    _lib.call("vlb_avgpool_rows_fwd", _p(y, BF16), _p(out, torch.float32), _ld(out), col0, pad_col, K, P, C, _p(segm, torch.float32), _stream())


def avgpool_rows_bwd(dfeat, y, boxes, dz, K, P, C, drop_p=0.0, seed=None, tag=0, drop_row_elems=0, drop_col0=0, segm=None):
    _lib.call("vlb_avgpool_rows_bwd", _p(dfeat, BF16), _ld(dfeat), _p(y, BF16), _p(boxes, torch.float32), _ld(boxes) if boxes is not None else 0,
              _p(dz, BF16), K, P, C, float(drop_p), _p(seed), int(tag), int(drop_row_elems), int(drop_col0), _p(segm, torch.float32), _stream())
    return dz


class ShardRanges:
    """Device table of the slices one data-parallel rank owns (parallel.GradBuckets, mode "sharded"): rows (offset into the flat
    parameter buffers, offset into the compact reduced-gradient / bf16-weight images, length).  Built once; sumsq() and adamw() are one
    launch each over all slices (vlb_sumsq_ranges_det / vlb_adamw_step_ranges)."""

    def __init__(self, rows, device, max_blocks=2048):
        rows = [r for r in rows if r[2] > 0]
        total = sum(r[2] for r in rows)
        chunk = 4096
        while sum((r[2] + chunk - 1) // chunk for r in rows) > max_blocks:
            chunk *= 2
        starts, nb = [0], 0
        for r in rows:
            nb += (r[2] + chunk - 1) // chunk
            starts.append(nb)
        self.rows, self.n, self.total_blocks, self.chunk, self.total = rows, len(rows), nb, chunk, total
        self.ranges = torch.tensor(rows, dtype=torch.int64).reshape(-1, 3).to(device)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(device)

    def sumsq(self, g, partials, out):
        _lib.call("vlb_sumsq_ranges_det", _p(g), 1 if g.dtype == BF16 else 0, self.ranges.data_ptr(), self.starts.data_ptr(), self.n,
                  self.total_blocks, self.chunk, _p(partials, torch.float32), partials.numel(), _p(out, torch.float32), _stream())

    def adamw(self, p, g, m, v, p16c, state, grad_scale=1.0):
        _lib.call("vlb_adamw_step_ranges", _p(p, torch.float32), _p(g), 1 if g.dtype == BF16 else 0, _p(m, torch.float32),
                  _p(v, torch.float32), _p(p16c, BF16), self.ranges.data_ptr(), self.starts.data_ptr(), self.n, self.total_blocks,
                  self.chunk, _p(state, torch.float32), float(grad_scale), _stream())


# ---------------------------------------------------------------------------------------------------------------
# fp32 encoder path (csrc/f32_path.hip): raw-pointer wrappers -- operands are sub-matrices of larger fp32 buffers addressed by
# (tensor, element offset), batched with two stride levels
# ---------------------------------------------------------------------------------------------------------------
F32 = torch.float32


def _pf(t, off=0):
    if t is None:
        return None
    if isinstance(t, tuple):
        t, off = t
    if not t.is_cuda or t.dtype != F32:
        raise RuntimeError("fp32 path: expected a float32 GPU tensor (got %s on %s)" % (t.dtype, t.device))
    return t.data_ptr() + 4 * int(off)


def gemm_nt_f32(A, lda, B, ldb, C, ldc, M, N, K, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, sBias1=0, alpha=1.0, epi=0,
                aux=None, ldaux=0, pre=None, ldpre=0, res=None, ldres=0, drop_p=0.0, seed=None, tag=0, atomic=False, splitk=1):
    """C (+)= epilogue(alpha * A . B^T) in fp32 (vlb_gemm_nt_f32).  A / B / C / bias / aux / pre / res: tensor or (tensor, element offset)."""
    _lib.call("vlb_gemm_nt_f32", _pf(A), int(lda), _pf(B), int(ldb), _pf(C), int(ldc), int(M), int(N), int(K), int(batch[0]), int(batch[1]),
              int(sA[0]), int(sA[1]), int(sB[0]), int(sB[1]), int(sC[0]), int(sC[1]), _pf(bias), int(sBias1), float(alpha), int(epi),
              _pf(aux), int(ldaux), _pf(pre), int(ldpre), _pf(res), int(ldres), float(drop_p), _p(seed), int(tag), 1 if atomic else 0,
              int(splitk), _stream())


def gemm_tn_f32(A, lda, B, ldb, C, ldc, R, Mo, No, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0, atomic=False, splitk=1, colsum=None):
    """C[Mo, No] (+)= alpha * A[R, Mo]^T . B[R, No] in fp32 (vlb_gemm_tn_f32: reduction over the rows of two row-major operands).
    A / B / C: tensor or (tensor, element offset); colsum [Mo] += column sums of A (unbatched)."""
    _lib.call("vlb_gemm_tn_f32", _pf(A), int(lda), _pf(B), int(ldb), _pf(C), int(ldc), int(R), int(Mo), int(No), int(batch[0]), int(batch[1]),
              int(sA[0]), int(sA[1]), int(sB[0]), int(sB[1]), int(sC[0]), int(sC[1]), float(alpha), 1 if atomic else 0, int(splitk), _pf(colsum),
              _stream())


def transpose_f32(src, lds, dst, ldd, R, C, Rp, batch=(1, 1), sS=(0, 0), sD=(0, 0), colsum=None):
    _lib.call("vlb_transpose_f32", _pf(src), int(lds), _pf(dst), int(ldd), int(R), int(C), int(Rp), int(batch[0]), int(batch[1]), int(sS[0]),
              int(sS[1]), int(sD[0]), int(sD[1]), _pf(colsum), _stream())


def layernorm_f32_fwd(x, gamma, beta, y, stats, eps=1e-12):
    rows, H = x.shape
    _lib.call("vlb_layernorm_f32_fwd", _pf(x), _ld(x), _pf(gamma), _pf(beta), _pf(y), _ld(y), _pf(stats), rows, H, float(eps), _stream())


def layernorm_f32_bwd(dy, x, stats, gamma, dx=None, dx_drop=None, drop_p=0.0, seed=None, tag=0, dgamma=None, dbeta=None):
    rows, H = x.shape
    _lib.call("vlb_layernorm_f32_bwd", _pf(dy), _ld(dy), _pf(x), _ld(x), _pf(stats), _pf(gamma), _pf(dx), _ld(dx), _pf(dx_drop), _ld(dx_drop),
              float(drop_p), _p(seed), int(tag), _pf(dgamma), _pf(dbeta), rows, H, _stream())


def softmax_f32_fwd(s, mask01, rows_per_sample, p, pd, rows, S, Sp, drop_p=0.0, seed=None, tag=0):
    _lib.call("vlb_softmax_f32_fwd", _pf(s), _pf(mask01), int(rows_per_sample), _pf(p), _pf(pd), int(rows), int(S), int(Sp), float(drop_p),
              _p(seed), int(tag), _stream())


def softmax_f32_bwd(p, dpd, rows, S, Sp, drop_p=0.0, seed=None, tag=0):
    _lib.call("vlb_softmax_f32_bwd", _pf(p), _pf(dpd), int(rows), int(S), int(Sp), float(drop_p), _p(seed), int(tag), _stream())
