"""fp32 compute mode of the encoder (csrc/f32_path.hip + vl-bert_amd/encoder_f32.py; the reference's `TRAIN.FP16: false` configurations,
BASELINE.json config 4 "VL-BERT-large VQA fine-tune ... fp32"), -m gpu:
  * the kernels against fp64 torch statements of the same ops (the fp32 products run on the bf16 matrix cores by operand splitting:
    the bar is fp32-class, 2e-5 relative, not bf16-class);
  * the engine with `encoder_fp32=True` against the fp32 oracle at north_star's fp32 tolerance, 1e-3 on logits and gradient norm --
    at 4 and at 24 layers of the large model, on the fp16 build of the 16-bit front / back end (child process, VLB_PRECISION=f16)."""
import math
import os
import re
import subprocess
import sys

import pytest
import torch

from oracle import vlbert_oracle as O
from tests.gpu_util import dev, pkg, report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 200, 96), (1000, 1024, 1024), (229, 256, 64)])
def test_gemm_f32_split_products_are_fp32_class(M, N, K):
    ops = pkg("ops")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    res, aux = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    Ag, Bg, bg, rg, ag = (t.to(dev()) for t in (A, B, bias, res, aux))
    acc = A.double() @ B.double().t()
    C = torch.full((M, N), 7.0, device=dev())
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K)
    e = _rel(C, acc)
    print("gemm_f32 %dx%dx%d plain: max rel err %.2e" % (M, N, K, e))
    assert e < 2e-5
    # the same product with bf16 operands would sit at 4e-3: make sure the split is really taken
    e16 = _rel(A.bfloat16().double() @ B.bfloat16().double().t(), acc)
    assert e16 > 50 * e
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, bias=bg, alpha=0.5)
    assert _rel(C, 0.5 * acc + bias.double()) < 2e-5
    pre = torch.zeros((M, N), device=dev())
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, bias=bg, epi=1, pre=pre, ldpre=N)
    u = acc + bias.double()
    cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2)))
    assert _rel(C, u * cdf) < 2e-5 and _rel(pre, cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)) < 5e-5      # (erf by A&S 7.1.26)
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, epi=3, aux=ag, ldaux=N, res=rg, ldres=N)
    assert _rel(C, acc * aux.double() + res.double()) < 2e-5
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, bias=bg, epi=2)
    assert _rel(C, torch.relu(u)) < 2e-5
    # accumulate with split K (weight gradients)
    C.fill_(1.0)
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, atomic=True, splitk=3)
    assert _rel(C, acc + 1.0) < 2e-5
    # dropout + residual: the keep pattern must be the library's counter RNG at element m * N + n
    import numpy as np
    from tests.gpu_util import drop_scale, drop_thr, keep_mask
    seed = torch.tensor([991], dtype=torch.int32, device=dev())
    ops.gemm_nt_f32(Ag, K, Bg, K, C, N, M, N, K, bias=bg, drop_p=0.1, seed=seed, tag=5, res=rg, ldres=N)
    thr = drop_thr(0.1)
    keep = torch.from_numpy(keep_mask(991, 5, np.arange(M * N, dtype=np.int64), thr).reshape(M, N))
    assert _rel(C, torch.where(keep, u * drop_scale(thr), torch.zeros_like(u)) + res.double()) < 2e-5


def test_gemm_f32_batched_strided_heads_and_transpose():
    """The attention products: operands are [S, 64] head slices of a [B*S, 3H] buffer (two batch levels), results go to / come from
    per-(sample, head) [S, Sp] matrices; the batched transpose pads the reduction dimension with zeros and sums columns."""
    ops = pkg("ops")
    Bt, nh, S, Sp, H = 2, 3, 45, 64, 192
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(Bt * S + Sp, 3 * H, generator=g)
    qg = qkv.to(dev())
    sc = torch.zeros((Bt * nh * S, Sp), device=dev())
    ops.gemm_nt_f32(qg, 3 * H, (qg, H), 3 * H, sc, Sp, S, Sp, 64, batch=(Bt, nh), sA=(S * 3 * H, 64), sB=(S * 3 * H, 64), sC=(nh * S * Sp, S * Sp),
                    alpha=0.125)
    for b in range(Bt):
        for h in range(nh):
            q = qkv[b * S:(b + 1) * S, h * 64:(h + 1) * 64].double()
            k = qkv[b * S:(b + 1) * S, H + h * 64:H + (h + 1) * 64].double()
            got = sc[(b * nh + h) * S:(b * nh + h + 1) * S, :S]
            assert _rel(got, 0.125 * q @ k.t()) < 2e-5, (b, h)
    vt = torch.full((Bt * nh * 64, Sp), 9.0, device=dev())
    ops.transpose_f32((qg, 2 * H), 3 * H, vt, Sp, S, 64, Sp, batch=(Bt, nh), sS=(S * 3 * H, 64), sD=(nh * 64 * Sp, 64 * Sp))
    for b in range(Bt):
        for h in range(nh):
            v = qkv[b * S:(b + 1) * S, 2 * H + h * 64:2 * H + (h + 1) * 64]
            blk = vt[(b * nh + h) * 64:(b * nh + h + 1) * 64].cpu()
            assert torch.equal(blk[:, :S], v.t()) and float(blk[:, S:].abs().max()) == 0.0
    x = torch.randn(70, 40, generator=g)
    t = torch.zeros((40, 96), device=dev())
    cs = torch.ones(40, device=dev())
    ops.transpose_f32(x.to(dev()), 40, t, 96, 70, 40, 96, colsum=cs)
    assert torch.equal(t[:, :70].cpu(), x.t()) and float(t[:, 70:].abs().max()) == 0.0
    assert _rel(cs, x.double().sum(0) + 1.0) < 1e-6


@pytest.mark.parametrize("R,Mo,No", [(64, 128, 128), (229, 229, 64), (3664, 1024, 256), (1000, 200, 136), (33, 8, 4)])
def test_gemm_tn_f32_row_reduction_products(R, Mo, No):
    """vlb_gemm_tn_f32: C = alpha A^T B with the reduction over the ROWS of two row-major fp32 operands (weight gradients, P^T dO, dS^T Q):
    fp32-class against fp64 torch -- overwrite, accumulate with one and several K slices, column sums of A, R not a multiple of 32,
    Mo not a multiple of 8 (operand columns beyond Mo hold garbage that must only reach masked outputs)."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(R + Mo + No)
    lda, ldb = (Mo + 7) // 8 * 8 + 8, (No + 7) // 8 * 8
    A = torch.randn(R, lda, generator=g) * 3.0            # (columns >= Mo: garbage the kernel may read)
    B = torch.randn(R, ldb, generator=g) * 0.05
    A[:, Mo:] = 1e30
    Ag, Bg = A.to(dev()), B.to(dev())
    ref = A[:, :Mo].double().t() @ B[:, :No].double()
    ldc = No + 4
    C = torch.full((Mo + 3, ldc), 7.0, device=dev())
    ops.gemm_tn_f32(Ag, lda, Bg, ldb, C, ldc, R, Mo, No)
    e = _rel(C[:Mo, :No], ref)
    print("gemm_tn_f32 R=%d %dx%d plain: max rel err %.2e" % (R, Mo, No, e))
    assert e < 2e-5
    assert float((C[Mo:] - 7.0).abs().max()) == 0.0 and float((C[:, No:] - 7.0).abs().max()) == 0.0      # nothing outside the product is touched
    e16 = _rel(A[:, :Mo].bfloat16().double().t() @ B[:, :No].bfloat16().double(), ref)
    assert e16 > 20 * e or R < 64
    cs = torch.ones(Mo, device=dev())
    C.fill_(1.0)
    ops.gemm_tn_f32(Ag, lda, Bg, ldb, C, ldc, R, Mo, No, alpha=0.5, atomic=True, colsum=cs)
    assert _rel(C[:Mo, :No], 0.5 * ref + 1.0) < 2e-5
    assert _rel(cs, A[:, :Mo].double().sum(0) + 1.0) < 1e-5
    C.fill_(-2.0)
    cs.zero_()
    ops.gemm_tn_f32(Ag, lda, Bg, ldb, C, ldc, R, Mo, No, atomic=True, splitk=3, colsum=cs)
    assert _rel(C[:Mo, :No], ref - 2.0) < 2e-5 and _rel(cs, A[:, :Mo].double().sum(0)) < 1e-5


def test_gemm_tn_f32_batched_attention_products():
    """dV = Pd^T dO and dK = dS^T Q / 8 per (sample, head) straight from the [S, Sp] probability matrices and the [B*S, 3H] / [B*S, H] buffers."""
    ops = pkg("ops")
    Bt, nh, S, Sp, H = 2, 3, 45, 64, 192
    g = torch.Generator().manual_seed(5)
    pd = torch.randn(Bt * nh * S, Sp, generator=g)
    do = torch.randn(Bt * S, H, generator=g)
    qkv = torch.randn(Bt * S + Sp, 3 * H, generator=g)
    dqkv = torch.full((Bt * S + Sp, 3 * H), 5.0, device=dev())
    ops.gemm_tn_f32(pd.to(dev()), Sp, do.to(dev()), H, (dqkv, 2 * H), 3 * H, S, S, 64, batch=(Bt, nh), sA=(nh * S * Sp, S * Sp), sB=(S * H, 64),
                    sC=(S * 3 * H, 64))
    ops.gemm_tn_f32(pd.to(dev()), Sp, qkv.to(dev()), 3 * H, (dqkv, H), 3 * H, S, S, 64, batch=(Bt, nh), sA=(nh * S * Sp, S * Sp),
                    sB=(S * 3 * H, 64), sC=(S * 3 * H, 64), alpha=0.125)
    for b in range(Bt):
        for h in range(nh):
            P = pd[(b * nh + h) * S:(b * nh + h + 1) * S, :S].double()
            dv = P.t() @ do[b * S:(b + 1) * S, h * 64:(h + 1) * 64].double()
            dk = 0.125 * P.t() @ qkv[b * S:(b + 1) * S, h * 64:(h + 1) * 64].double()
            assert _rel(dqkv[b * S:(b + 1) * S, 2 * H + h * 64:2 * H + (h + 1) * 64], dv) < 2e-5, (b, h)
            assert _rel(dqkv[b * S:(b + 1) * S, H + h * 64:H + (h + 1) * 64], dk) < 2e-5, (b, h)
    assert float((dqkv[:, :H] - 5.0).abs().max()) == 0.0 and float((dqkv[Bt * S:] - 5.0).abs().max()) == 0.0


def test_fp32_encoder_row_reduction_form_matches_the_transposed_form():
    """The backward with vlb_gemm_tn_f32 (VLB_F32_TN=1) against the default transposes + NT products: same gradients to fp32 summation order."""
    E, syn = pkg("engine"), pkg("synthetic")
    grads = []
    for tn in (True, False):
        mc = E.ModelConfig(num_hidden_layers=2)
        eng = E.PretrainEngine(mc, 3, 32, 10, device="cuda:0", train=False, seed=5, encoder_fp32=True)
        eng.enc32.tn = tn
        eng.init_random(seed=1, visual_ln_init=1.0)
        eng.set_batch(*[t.to(dev()) for t in syn.make_batch(3, 32, 10, seed=9, ragged=True)])
        eng.zero_grad()
        eng.forward(False)
        eng.backward(False)
        torch.cuda.synchronize()
        grads.append(eng.P.grad.clone())
    scale = float(grads[1].abs().max())
    assert scale > 0 and float((grads[0] - grads[1]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("rows,H", [(37, 768), (200, 1024), (9, 64)])
def test_layernorm_f32_fwd_bwd(rows, H):
    ops = pkg("ops")
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, H, generator=g) * 1.7 - 0.2)
    gam, bet = 1.0 + 0.2 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    dy = torch.randn(rows, H, generator=g)
    xr = x.double().clone().requires_grad_(True)
    mu = xr.mean(1, keepdim=True)
    yref = (xr - mu) / torch.sqrt(((xr - mu) ** 2).mean(1, keepdim=True) + 1e-12) * gam.double() + bet.double()
    yref.backward(dy.double())
    xg, gg, bg, dyg = (t.to(dev()) for t in (x, gam, bet, dy))
    y, st = torch.zeros((rows, H), device=dev()), torch.zeros((rows, 2), device=dev())
    ops.layernorm_f32_fwd(xg, gg, bg, y, st)
    assert _rel(y, yref.detach()) < 1e-5
    dx, dxd = torch.zeros((rows, H), device=dev()), torch.zeros((rows, H), device=dev())
    dgam, dbet = torch.zeros(H, device=dev()), torch.zeros(H, device=dev())
    ops.layernorm_f32_bwd(dyg, xg, st, gg, dx=dx, dx_drop=dxd, dgamma=dgam, dbeta=dbet)
    assert _rel(dx, xr.grad) < 1e-5 and torch.equal(dx, dxd)
    xh = (x.double() - x.double().mean(1, keepdim=True)) / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-12)
    assert _rel(dgam, (dy.double() * xh).sum(0)) < 1e-5 and _rel(dbet, dy.double().sum(0)) < 1e-5


def test_softmax_f32_mask_dropout_fwd_bwd():
    ops = pkg("ops")
    import numpy as np
    from tests.gpu_util import drop_scale, drop_thr, keep_mask
    Bt, nh, S, Sp = 2, 2, 37, 64
    rows = Bt * nh * S
    g = torch.Generator().manual_seed(8)
    s = torch.randn(rows, Sp, generator=g) * 2
    mask = torch.ones(Bt, S)
    mask[0, 30:] = 0
    mask[1, 20:] = 0
    add = ((1 - mask) * -10000.0)[:, None, None, :].expand(Bt, nh, S, S).reshape(rows, S).double()
    pref = torch.softmax(s[:, :S].double() + add, -1)
    p, pd = torch.full((rows, Sp), 5.0, device=dev()), torch.full((rows, Sp), 5.0, device=dev())
    seed = torch.tensor([313], dtype=torch.int32, device=dev())
    ops.softmax_f32_fwd(s.to(dev()), mask.to(dev()), nh * S, p, pd, rows, S, Sp, drop_p=0.1, seed=seed, tag=16)
    assert _rel(p[:, :S], pref) < 1e-6 and float(p[:, S:].abs().max()) == 0.0
    thr = drop_thr(0.1)
    keep = torch.from_numpy(keep_mask(313, 16, np.arange(rows * Sp, dtype=np.int64), thr).reshape(rows, Sp))[:, :S]
    assert _rel(pd[:, :S], torch.where(keep, pref * drop_scale(thr), torch.zeros_like(pref))) < 1e-6
    dpd = torch.randn(rows, Sp, generator=g)
    dg = dpd.to(dev())
    ops.softmax_f32_bwd(p, dg, rows, S, Sp, drop_p=0.1, seed=seed, tag=16)
    dp = torch.where(keep, dpd[:, :S].double() * drop_scale(thr), torch.zeros_like(pref))
    ref = pref * (dp - (dp * pref).sum(-1, keepdim=True))
    assert _rel(dg[:, :S], ref) < 1e-5 and float(dg[:, S:].abs().max()) == 0.0


def _engine_case(layers, large, B, T, R, seed, tag):
    from tests.test_engine_gpu import _per_layer_report, check_against_oracle
    syn = pkg("synthetic")
    kw = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096) if large else {}
    cfg = O.VLBertConfig(num_hidden_layers=layers, **kw)
    params = O.init_params(cfg, seed=seed)
    batch = syn.make_batch(B, T, R, seed=seed + 1, ragged=True)
    f16 = pkg("ops").BF16 == torch.float16
    # fp32 bar of the north star on the fp16 build (its 16-bit front / back end rounds at 2^-12); on the bf16 build those few
    # roundings are 2^-9 each and the same code sits at ~6e-3
    bar = 1e-3 if f16 else 8e-3
    eng = check_against_oracle(tag, cfg, params, batch, grad_tol=2e-2 if f16 else 6e-2, logit_rtol=bar, logit_fro_tol=bar,
                               engine_kw=dict(encoder_fp32=True), loss_tol=1e-3, norm_tol=1e-3 if f16 else 5e-3)
    rows = _per_layer_report(tag, eng, eng.oracle_result[2], eng.oracle_result[3], layers)
    assert max(e for _, e in rows) <= (2e-3 if f16 else 2e-2), rows
    return eng


def test_engine_fp32_encoder_base_2_layers_vs_oracle():
    """In-process (whatever build the suite runs on): the fp32 encoder inside the engine, 2 base layers."""
    _engine_case(2, False, 3, 32, 10, 301, "fp32-encoder base 2-layer")


def test_engine_fp32_encoder_training_step_runs():
    """Dropout on, two optimizer steps: finite, deterministic under the same seed (masks are regenerated from the counter RNG in the
    fp32 LayerNorm / softmax backward)."""
    E, syn = pkg("engine"), pkg("synthetic")
    outs = []
    for _ in range(2):
        mc = E.ModelConfig(num_hidden_layers=2)
        eng = E.PretrainEngine(mc, 3, 32, 10, device="cuda:0", train=True, seed=5, encoder_fp32=True)
        eng.init_random(seed=1, visual_ln_init=1.0)
        eng.set_batch(*[t.to(dev()) for t in syn.make_batch(3, 32, 10, seed=9, ragged=True)])
        for _ in range(2):
            eng.train_step()
        torch.cuda.synchronize()
        lv = eng.loss_values()
        assert math.isfinite(lv["loss"])
        outs.append((lv["loss"], eng.P.master.clone()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-4 * abs(outs[0][0])
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-4


def test_engine_fp32_encoder_large_4_and_24_layers_on_the_fp16_build():
    """north_star's fp32 tolerance (1e-3 on logits and gradient norm) at BASELINE config 4's model: VL-BERT-large, 128 + 100
    positions, 4 and 24 layers, fp32 encoder + fp16 front / back end -- in a child process with VLB_PRECISION=f16."""
    env = dict(os.environ, VLB_PRECISION="f16")
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-s", "-k", "child_case"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    try:
        with open(os.path.join(ROOT, "gpurun_out", "f32_encoder_f16_build.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    print(r.stdout[-5000:])
    print(r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    for name in ("large 4-layer", "large 24-layer"):
        f = re.search(r"fp32-encoder %s logits relative Frobenius error: mlm (\S+)\s+mvrc (\S+)" % name, r.stdout)
        assert f, name
        print("fp32 encoder, %s: logits rel-Frobenius error mlm %s mvrc %s" % (name, f.group(1), f.group(2)))
        assert float(f.group(1)) <= 1e-3 and float(f.group(2)) <= 1e-3


@pytest.mark.skipif(os.environ.get("VLB_PRECISION", "bf16").lower() not in ("f16", "fp16"), reason="child of the test above (fp16 build)")
@pytest.mark.parametrize("layers", [4, 24])
def test_child_case_large(layers):
    _engine_case(layers, True, 2, 128, 100, 310 + layers, "fp32-encoder large %d-layer" % layers)


def test_vqa_and_module_api_mirrors_on_the_fp32_encoder():
    """The module API mirrors (common.visual_linguistic_bert, the VQA wrapper of BASELINE config 4) with VLB_ENCODER_FP32=1 on the fp16
    build: the same reference-fixture / oracle tests as the 16-bit builds run, through the fp32 encoder (child process)."""
    env = dict(os.environ, VLB_PRECISION="f16", VLB_ENCODER_FP32="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_engine_gpu.py"), "-m", "gpu", "-q", "-x", "-s", "-k",
           "vqa_module_mirror or module_api_hidden_states or module_api_pooler or module_api_pretraining_heads"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    try:
        with open(os.path.join(ROOT, "gpurun_out", "f32_encoder_mirrors.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    print(r.stdout[-4000:])
    print(r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 5


def test_module_api_inspection_forms_all_layers_and_attention_probs():
    """VisualLinguisticBert.forward(output_all_encoded_layers=True / output_attention_probs=True) (reference :131-171, its viz/ scripts):
    every layer's output and every layer's attention probabilities through the fp32 encoder path.  Checks: the last layer agrees with
    the default call form and with the oracle; probabilities are a masked softmax (rows sum to 1, padded keys get 0); layer 0's
    probabilities equal softmax(Q K^T / 8 + mask) recomputed in fp64 from the layer's own input and weights; the separate
    text / object form places the objects as the reference's masked scatter does."""
    from tests.test_engine_gpu import _core_fixture, _module_config
    VL = pkg("common.visual_linguistic_bert")
    z, cfg, params, ins = _core_fixture()
    net = VL.VisualLinguisticBert(_module_config(cfg)["NETWORK"]["VLBERT"])
    net.load_state_dict({k: v for k, v in params.items() if not k.startswith(("mlm_head.", "mvrc_head."))})
    net.eval()
    layers, pooled, probs = net(*ins, output_all_encoded_layers=True, output_attention_probs=True)
    L, nh, H = cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size
    assert len(layers) == L and len(probs) == L and pooled is None
    B, n = layers[0].shape[0], layers[0].shape[1]
    assert probs[0].shape == (B, nh, n, n)
    seq, _ = net(*ins, output_all_encoded_layers=False)
    report("inspection: last layer vs the default call form", layers[-1], seq[:, :n], 2e-3, 1.5e-2)
    p = O.init_params(cfg, seed=int(z["pseed"]))
    rt, ro, _, _ = O.vlbert_forward(p, cfg, *[t.cpu() for t in ins], False)
    T = ins[0].shape[1]
    tm = ins[3].cpu().unsqueeze(-1).float()
    report("inspection: last layer (text part) vs oracle", layers[-1][:, :T].cpu() * tm[:, :min(T, n)], (rt * tm)[:, :n], 2e-3, 1.5e-2)
    # masked softmax properties
    eng = net._engines[("inspect", B, T, ins[4].shape[1])]
    valid = eng.lay["attn_mask"].view(B, eng.S)[:, :n].bool()                       # [B, n] keys that may be attended
    for l in range(L):
        pr = probs[l]
        assert float((pr.sum(-1) - 1.0).abs().max()) < 1e-5
        assert float((pr * (~valid)[:, None, None, :]).abs().max()) == 0.0
    # layer 0 recomputed from its input
    x0 = eng.enc32.X[0].double().view(B, eng.S, H)[:, :n]
    wq, wk = p["vlbert.encoder.layer.0.attention.self.query.weight"].double().to(dev()), p["vlbert.encoder.layer.0.attention.self.key.weight"].double().to(dev())
    bq, bk = p["vlbert.encoder.layer.0.attention.self.query.bias"].double().to(dev()), p["vlbert.encoder.layer.0.attention.self.key.bias"].double().to(dev())
    q = (x0 @ wq.t() + bq).view(B, n, nh, 64).transpose(1, 2)
    k = (x0 @ wk.t() + bk).view(B, n, nh, 64).transpose(1, 2)
    sc = q @ k.transpose(-1, -2) / 8.0 + ((~valid).double() * -10000.0)[:, None, None, :]
    ref = torch.softmax(sc, -1)
    err = float((probs[0].double() - ref).abs().max())
    print("inspection: layer-0 attention probabilities vs fp64 recomputation: max abs err %.2e" % err)
    assert err < 2e-5
    # separate text / object form, last layer only + pooled None
    t_out, o_out, pooled = net(*ins, output_all_encoded_layers=True, output_text_and_object_separately=True)
    assert len(t_out) == L and t_out[-1].shape[1] == T and o_out[-1].shape[1] == ins[4].shape[1]
    report("inspection: object rows vs oracle", o_out[-1], ro, 2e-3, 1.5e-2)
