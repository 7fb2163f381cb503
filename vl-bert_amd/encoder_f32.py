"""fp32 compute path of the VL-BERT encoder (the reference's `TRAIN.FP16: false` configurations: cfgs/pretrain/base_prec_4x16G_fp32.yaml,
cfgs/vqa/large_4x16G_fp32.yaml; north_star's 1e-3 fp32 tolerance).

Replaces, layer by layer, external/pytorch_pretrained_bert/modeling.py:268-397 (BertSelfAttention / BertSelfOutput / BertIntermediate /
BertOutput) and its autograd backward with every tensor in fp32: activations, the residual stream, LayerNorm, softmax, the weights
(the engine's fp32 MASTER weights are the GEMM operands -- no 16-bit working copy exists for the encoder) and the gradients.
The kernels are csrc/f32_path.hip: one batched NT GEMM whose fp32 products run on the bf16 matrix cores by operand splitting
(x = h + m, three MFMAs per product, fp32 accumulation: 2^-16 relative), LayerNorm, masked softmax with dropout, a batched transpose.

Where the precision boundary lies: the embedding side and the heads stay on the engine's 16-bit kernels (run them on the fp16 build,
VLB_PRECISION=f16: 2^-12 per rounding, a handful of roundings in total); `forward` takes the embedding output X[0] up to fp32,
`backward` hands the 16-bit d X[0] back.  The 24 (12) layers in between -- where rounding error accumulates with depth, 10 operand
roundings per layer in a 16-bit build -- see none.  Measured: tests/test_f32_encoder_gpu.py.

Attention is composed from the batched GEMM: scores = Q.K^T / 8 per (sample, head) -> softmax (+ key mask, + dropout) -> P.V with V
transposed per head; the backward is the five products of the same shapes.  The probabilities are kept per layer ([B, h, S, Sp] fp32:
54 MB per layer at 16 x 229 -- 1.3 GB for 24 layers, nothing against 288 GB).
Weight gradients: dW += dY^T.X as an NT product of the two transposed operands (zero-padded to a multiple of 32 rows), accumulated into
the engine's flat gradient by a plain read-modify-write (one K slice; two slices with fp32 atomics only for the 1024 x 1024 outputs
that would otherwise occupy a quarter of the chip); bias gradients are the column sums taken by the transpose of dY.  (VLB_F32_TN=1: the
same products, and dV = Pd^T.dO / dK = dS^T.Q, straight from the row-major tensors through vlb_gemm_tn_f32 -- correct, not faster yet.)
"""
import os

import torch

from . import ops

F32 = torch.float32


def _ru(x, m):
    return (x + m - 1) // m * m


class EncoderF32:
    def __init__(self, eng):
        cfg = eng.cfg
        self.eng = eng
        d = eng.dev
        H, I, L, nh = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads
        M, S, Bt = eng.M, eng.S, eng.Bt
        self.H, self.I, self.L, self.nh, self.M, self.S, self.Bt = H, I, L, nh, M, S, Bt
        self.Sp = _ru(S, 32)                     # keys / queries padded to the GEMM's K granularity
        self.Mp = _ru(M, 32)
        if self.Sp > 256:
            raise ValueError("fp32 encoder: sequences of at most 256 positions")
        Sp = self.Sp
        z = lambda *s: torch.zeros(s, dtype=F32, device=d)
        slack = Sp                               # score GEMMs read Sp key rows per sample: the last sample reads past M rows
        self.X = [z(M, H) for _ in range(L + 1)]
        self.QKV = [z(M + slack, 3 * H) for _ in range(L)]
        self.P = [z(Bt * nh * S, Sp) for _ in range(L)]
        self.Pd = None                           # dropout(P) per layer, allocated on the first training forward
        self.CTX = [z(M, H) for _ in range(L)]
        self.Z1, self.ST1, self.Y1 = [z(M, H) for _ in range(L)], [z(M, 2) for _ in range(L)], [z(M, H) for _ in range(L)]
        self.G, self.dG = [z(M, I) for _ in range(L)], [z(M, I) for _ in range(L)]
        self.Z2, self.ST2 = [z(M, H) for _ in range(L)], [z(M, 2) for _ in range(L)]
        # temporaries
        self.Sbuf = z(Bt * nh * S, Sp)           # scores / dP / dS
        self.Tt = z(Bt * nh * S, Sp)             # transposed [S, S] matrices (Pd^T, dS^T)
        self.hT = [z(Bt * nh * 64, Sp) for _ in range(2)]      # per-head transposes of Q / K / V / dO ([64, Sp])
        self.dQKV = z(M + slack, 3 * H)
        self.dZ, self.dD, self.dY1, self.dCTX = z(M, H), z(M, H), z(M, H), z(M, H)
        self.dU = z(M, I)
        self.dx = [z(M, H), z(M, H)]
        self.tG, self.tA = z(max(3 * H, I), self.Mp), z(max(H, I), self.Mp)
        # transposed weights for the data gradients (dX = dY . W as an NT product against W^T), refreshed with the weights
        self.wT = []
        for l in range(L):
            self.wT.append(dict(qkv=z(H, 3 * H), ao=z(H, H), f1=z(H, I), f2=z(I, H)))
        self._fresh = False
        # VLB_F32_TN=1: row-reduction ("TN") products straight from the row-major tensors (vlb_gemm_tn_f32) for the weight gradients,
        # dV = Pd^T dO and dK = dS^T Q instead of transposed fp32 copies + the NT kernel.  Same gradients (tested); OFF by default: at the
        # VQA-large micro-batch (3 664 rows) the step is 193.7 ms with it and 190.9 without -- the 128 x 128-only TN kernel loses more
        # on the under-filled 1024 x 1024 outputs than the 23 ms of transposes it removes (DESIGN.md, fp32 path)
        self.tn = os.environ.get("VLB_F32_TN", "0") == "1"

    # -- parameters ---------------------------------------------------------------------------------------------------
    def _w(self, l):
        e, p = self.eng, "vlbert.encoder.layer.%d." % l
        H = self.H
        W, G = e.P.master, e.P.grad
        v = lambda buf, name, shape, span=1: e.P.view(buf, p + name, shape, span=span)
        return dict(
            wqkv=v(W, "attention.self.query.weight", (3 * H, H), 3), bqkv=v(W, "attention.self.query.bias", (3 * H,), 3),
            wo=e.w32[p + "attention.output.dense.weight"], bo=e.w32[p + "attention.output.dense.bias"],
            g1=e.w32[p + "attention.output.LayerNorm.weight"], be1=e.w32[p + "attention.output.LayerNorm.bias"],
            w1=e.w32[p + "intermediate.dense.weight"], b1=e.w32[p + "intermediate.dense.bias"],
            w2=e.w32[p + "output.dense.weight"], b2=e.w32[p + "output.dense.bias"],
            g2=e.w32[p + "output.LayerNorm.weight"], be2=e.w32[p + "output.LayerNorm.bias"],
            gwqkv=v(G, "attention.self.query.weight", (3 * H, H), 3), gbqkv=v(G, "attention.self.query.bias", (3 * H,), 3),
            gwo=e.g32[p + "attention.output.dense.weight"], gbo=e.g32[p + "attention.output.dense.bias"],
            gg1=e.g32[p + "attention.output.LayerNorm.weight"], gbe1=e.g32[p + "attention.output.LayerNorm.bias"],
            gw1=e.g32[p + "intermediate.dense.weight"], gb1=e.g32[p + "intermediate.dense.bias"],
            gw2=e.g32[p + "output.dense.weight"], gb2=e.g32[p + "output.dense.bias"],
            gg2=e.g32[p + "output.LayerNorm.weight"], gbe2=e.g32[p + "output.LayerNorm.bias"])

    def refresh(self):
        """W^T copies of the current fp32 master weights (after load / every optimizer step)."""
        H, I = self.H, self.I
        for l in range(self.L):
            w, t = self._w(l), self.wT[l]
            ops.transpose_f32(w["wqkv"], H, t["qkv"], 3 * H, 3 * H, H, 3 * H)
            ops.transpose_f32(w["wo"], H, t["ao"], H, H, H, H)
            ops.transpose_f32(w["w1"], H, t["f1"], I, I, H, I)
            ops.transpose_f32(w["w2"], I, t["f2"], H, H, I, H)
        self._fresh = True

    # -- helpers ------------------------------------------------------------------------------------------------------
    def _linear(self, x, K, w, N, out, **kw):
        ops.gemm_nt_f32(x, K, w, K, out, N, self.M, N, K, **kw)

    def _head_T(self, src, col0, lds, dst):
        """dst[(b, h)] [64, Sp] = the [S, 64] head slices of src (columns col0 + 64 h) transposed, zero-padded to Sp."""
        S, Sp, nh = self.S, self.Sp, self.nh
        ops.transpose_f32((src, col0), lds, dst, Sp, S, 64, Sp, batch=(self.Bt, nh), sS=(S * lds, 64), sD=(nh * 64 * Sp, 64 * Sp))

    def _ss_T(self, src, dst):
        """dst[(b, h)] [S, Sp] = src[(b, h)] [S, :S]^T, zero-padded to Sp columns."""
        S, Sp, nh = self.S, self.Sp, self.nh
        ops.transpose_f32(src, Sp, dst, Sp, S, S, Sp, batch=(self.Bt, nh), sS=(nh * S * Sp, S * Sp), sD=(nh * S * Sp, S * Sp))

    def _wgrad(self, dy, N, x, K, gw, gb):
        """gw[N, K] += dy[M, N]^T . x[M, K] ; gb[N] += column sums of dy."""
        M, Mp = self.M, self.Mp
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        if self.tn:      # K slices only where the output has too few tiles to occupy the chip (fp32 atomics cost more than idle CUs)
            ops.gemm_tn_f32(dy, N, x, K, gw, K, M, N, K, atomic=True, splitk=1 if tiles >= 128 else 2, colsum=gb)
            return
        ops.transpose_f32(dy, N, self.tG, Mp, M, N, Mp, colsum=gb)
        ops.transpose_f32(x, K, self.tA, Mp, M, K, Mp)
        # K slices only where the output has too few tiles to occupy the chip: fp32 atomics cost more than idle CUs (measured,
        # tools/f32_gemm_bench.py: 4096 x 1024 x 3680 -- 131 us with one slice (plain read-modify-write), 186 / 226 / 268 us with 2 / 3 / 4)
        ops.gemm_nt_f32(self.tG, Mp, self.tA, Mp, gw, K, N, K, Mp, atomic=True, splitk=1 if tiles >= 128 else 2)

    # -- forward ------------------------------------------------------------------------------------------------------
    def forward(self, p_h, p_a):
        """engine.X[0] (16-bit embedding output) -> 24 / 12 fp32 layers -> engine.X[L] (16-bit, for the heads)."""
        e = self.eng
        H, I, L, nh, M, S, Sp, Bt = self.H, self.I, self.L, self.nh, self.M, self.S, self.Sp, self.Bt
        if not self._fresh:
            self.refresh()
        if p_a > 0 and self.Pd is None:
            self.Pd = [torch.zeros((Bt * nh * S, Sp), dtype=F32, device=e.dev) for _ in range(L)]
        seed, mask = e.seed, e.lay["attn_mask"]
        ops.cast_bf16_f32(e.X[0], self.X[0])
        bh = (Bt, nh)
        for l in range(L):
            w = self._w(l)
            x, qkv = self.X[l], self.QKV[l]
            self._linear(x, H, w["wqkv"], 3 * H, qkv, bias=w["bqkv"])
            # scores = Q . K^T / sqrt(64) per (sample, head)
            ops.gemm_nt_f32(qkv, 3 * H, (qkv, H), 3 * H, self.Sbuf, Sp, S, Sp, 64, batch=bh, sA=(S * 3 * H, 64), sB=(S * 3 * H, 64),
                            sC=(nh * S * Sp, S * Sp), alpha=0.125)
            pd = self.Pd[l] if p_a > 0 else self.P[l]
            ops.softmax_f32_fwd(self.Sbuf, mask, nh * S, self.P[l], pd, Bt * nh * S, S, Sp, drop_p=p_a, seed=seed, tag=l * 8 + 0)
            self._head_T(qkv, 2 * H, 3 * H, self.hT[0])                                             # V^T per head
            ops.gemm_nt_f32(pd, Sp, self.hT[0], Sp, self.CTX[l], H, S, 64, Sp, batch=bh, sA=(nh * S * Sp, S * Sp),
                            sB=(nh * 64 * Sp, 64 * Sp), sC=(S * H, 64))
            self._linear(self.CTX[l], H, w["wo"], H, self.Z1[l], bias=w["bo"], drop_p=p_h, seed=seed, tag=l * 8 + 1, res=x, ldres=H)
            ops.layernorm_f32_fwd(self.Z1[l], w["g1"], w["be1"], self.Y1[l], self.ST1[l])
            self._linear(self.Y1[l], H, w["w1"], I, self.G[l], bias=w["b1"], epi=1, pre=self.dG[l], ldpre=I)
            self._linear(self.G[l], I, w["w2"], H, self.Z2[l], bias=w["b2"], drop_p=p_h, seed=seed, tag=l * 8 + 2, res=self.Y1[l], ldres=H)
            ops.layernorm_f32_fwd(self.Z2[l], w["g2"], w["be2"], self.X[l + 1], self.ST2[l])
        ops.cast_f32_bf16(self.X[L], e.X[L])

    # -- backward -----------------------------------------------------------------------------------------------------
    def backward(self, dx16, p_h, p_a, on_layer_done=None, will_launch=None):
        """dx16: 16-bit d X[L] [M, H] -> weight gradients ACCUMULATED into the engine's flat fp32 gradient; returns the 16-bit d X[0]
        (written over dx16's buffer)."""
        e = self.eng
        H, I, L, nh, M, S, Sp, Bt = self.H, self.I, self.L, self.nh, self.M, self.S, self.Sp, self.Bt
        seed = e.seed
        bh = (Bt, nh)
        dx = self.dx[0]
        ops.cast_bf16_f32(dx16, dx)
        for l in reversed(range(L)):
            w, t = self._w(l), self.wT[l]
            dx_next = self.dx[1] if dx is self.dx[0] else self.dx[0]
            qkv = self.QKV[l]
            # LayerNorm 2 -> dZ2 (residual branch) and its dropout-masked copy into output.dense
            ops.layernorm_f32_bwd(dx, self.Z2[l], self.ST2[l], w["g2"], dx=self.dZ, dx_drop=self.dD, drop_p=p_h, seed=seed, tag=l * 8 + 2,
                                  dgamma=w["gg2"], dbeta=w["gbe2"])
            self._wgrad(self.dD, H, self.G[l], I, w["gw2"], w["gb2"])
            self._linear(self.dD, H, t["f2"], I, self.dU, epi=3, aux=self.dG[l], ldaux=I)               # dU = (dD2 . W2) x GELU'
            self._wgrad(self.dU, I, self.Y1[l], H, w["gw1"], w["gb1"])
            self._linear(self.dU, I, t["f1"], H, self.dY1, res=self.dZ, ldres=H)                          # dY1 = dU . W1 + dZ2
            # LayerNorm 1
            ops.layernorm_f32_bwd(self.dY1, self.Z1[l], self.ST1[l], w["g1"], dx=self.dZ, dx_drop=self.dD, drop_p=p_h, seed=seed,
                                  tag=l * 8 + 1, dgamma=w["gg1"], dbeta=w["gbe1"])
            self._wgrad(self.dD, H, self.CTX[l], H, w["gwo"], w["gbo"])
            self._linear(self.dD, H, t["ao"], H, self.dCTX)
            # attention backward: dPd = dO . V^T ; dV = Pd^T . dO ; dS = softmax'(dP) ; dQ = dS . K / 8 ; dK = dS^T . Q / 8
            pd = self.Pd[l] if p_a > 0 else self.P[l]
            ops.gemm_nt_f32(self.dCTX, H, (qkv, 2 * H), 3 * H, self.Sbuf, Sp, S, Sp, 64, batch=bh, sA=(S * H, 64), sB=(S * 3 * H, 64),
                            sC=(nh * S * Sp, S * Sp))
            if self.tn:      # dV = Pd^T . dO per (sample, head): the rows of Pd and of dO are the reduction
                ops.gemm_tn_f32(pd, Sp, self.dCTX, H, (self.dQKV, 2 * H), 3 * H, S, S, 64, batch=bh, sA=(nh * S * Sp, S * Sp), sB=(S * H, 64),
                                sC=(S * 3 * H, 64))
            else:
                self._ss_T(pd, self.Tt)
                self._head_T(self.dCTX, 0, H, self.hT[0])                                               # dO^T per head
                ops.gemm_nt_f32(self.Tt, Sp, self.hT[0], Sp, (self.dQKV, 2 * H), 3 * H, S, 64, Sp, batch=bh, sA=(nh * S * Sp, S * Sp),
                                sB=(nh * 64 * Sp, 64 * Sp), sC=(S * 3 * H, 64))
            ops.softmax_f32_bwd(self.P[l], self.Sbuf, Bt * nh * S, S, Sp, drop_p=p_a, seed=seed, tag=l * 8 + 0)
            self._head_T(qkv, H, 3 * H, self.hT[0])                                                     # K^T per head
            ops.gemm_nt_f32(self.Sbuf, Sp, self.hT[0], Sp, self.dQKV, 3 * H, S, 64, Sp, batch=bh, sA=(nh * S * Sp, S * Sp),
                            sB=(nh * 64 * Sp, 64 * Sp), sC=(S * 3 * H, 64), alpha=0.125)
            if self.tn:      # dK = dS^T . Q / 8
                ops.gemm_tn_f32(self.Sbuf, Sp, qkv, 3 * H, (self.dQKV, H), 3 * H, S, S, 64, batch=bh, sA=(nh * S * Sp, S * Sp),
                                sB=(S * 3 * H, 64), sC=(S * 3 * H, 64), alpha=0.125)
            else:
                self._ss_T(self.Sbuf, self.Tt)
                self._head_T(qkv, 0, 3 * H, self.hT[1])                                                 # Q^T per head
                ops.gemm_nt_f32(self.Tt, Sp, self.hT[1], Sp, (self.dQKV, H), 3 * H, S, 64, Sp, batch=bh, sA=(nh * S * Sp, S * Sp),
                                sB=(nh * 64 * Sp, 64 * Sp), sC=(S * 3 * H, 64), alpha=0.125)
            self._wgrad(self.dQKV, 3 * H, self.X[l], H, w["gwqkv"], w["gbqkv"])
            self._linear(self.dQKV, 3 * H, t["qkv"], H, dx_next, res=self.dZ, ldres=H)                   # dX_l = dQKV . Wqkv + dZ1
            dx = dx_next
            if on_layer_done and (will_launch is None or will_launch(l)):
                on_layer_done(l)
        ops.cast_f32_bf16(dx, dx16)
        return dx16
