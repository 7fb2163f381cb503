"""End-to-end parity (-m gpu): the HIP training step (vl-bert_amd/engine.py through the C ABI) against
(a) the golden fixtures produced by the REAL reference and (b) the CPU oracle on identical synthetic
batches.  Tolerance = north_star's bf16 bound: 1e-2 on logits (relative to the tensor's scale) and on
the global gradient norm; per-parameter gradients are checked in relative Frobenius norm."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import vlbert_oracle as O
from tests.gpu_util import bf, dev, pkg, report

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
SAMPLE = 4096


def make_engine(cfg, B, T, R, **kw):
    E = pkg("engine")
    mc = E.ModelConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                       num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                       vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings,
                       visual_region_classes=cfg.visual_region_classes,
                       hidden_dropout_prob=cfg.hidden_dropout_prob,
                       attention_probs_dropout_prob=cfg.attention_probs_dropout_prob,
                       obj_downsample_dropout=cfg.obj_downsample_dropout, multitask=getattr(cfg, "multitask", False),
                       with_pooler=cfg.with_pooler, with_rel_loss=cfg.with_rel_loss)
    return E.PretrainEngine(mc, B, T, R, device="cuda:0", keep_logits=True, **kw)


def rel_fro(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / max(b.norm(), 1e-12))


def check_against_oracle(tag, cfg, params, batch, grad_tol=5e-2, logit_rtol=1e-2, logit_fro_tol=None, oracle=None, engine_kw=None,
                         loss_tol=1e-2, norm_tol=1e-2):
    """logit_rtol: bound on max|err| / max|ref| (+ 2e-3 absolute); logit_fro_tol: additional bound on the relative Frobenius error.
    oracle: a precomputed O.loss_and_grads(params, cfg, batch, train=False) result (several engine configurations against one
    oracle evaluation); the result used is left in eng.oracle_result."""
    B, T, R = batch[2].shape[0], batch[2].shape[1], batch[0].shape[1]
    eng = make_engine(cfg, B, T, R, train=False, **(engine_kw or {}))
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    eng.set_batch(*[t.to(dev()) for t in batch])
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    torch.cuda.synchronize()
    if oracle is None:
        oracle = O.loss_and_grads(params, cfg, batch, train=False)
    outputs, loss, grads, norm = oracle
    eng.oracle_result = oracle
    lv = eng.loss_values()
    V, C = cfg.vocab_size, cfg.visual_region_classes
    if logit_fro_tol is not None:
        fro = rel_fro(eng.mlm_logits_copy[:, :V].view(B, T, V), outputs["mlm_logits"].detach())
        fro2 = rel_fro(eng.mvrc_logits_copy[:, :C].view(B, R, C)[:, :int((batch[0][:, :, 0] > -1.5).sum(1).max())],
                       outputs["mvrc_logits"].detach()[:, :int((batch[0][:, :, 0] > -1.5).sum(1).max())])
        line = "%s logits relative Frobenius error: mlm %.3e  mvrc %.3e  (bound %.1e)" % (tag, fro, fro2, logit_fro_tol)
        print(line)
        try:
            from tests.gpu_util import REPORT
            with open(REPORT, "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
        assert fro <= logit_fro_tol and fro2 <= logit_fro_tol, line
    report(tag + " mlm_logits", eng.mlm_logits_copy[:, :V].view(B, T, V), outputs["mlm_logits"], 2e-3, logit_rtol)
    max_len = int((batch[0][:, :, 0] > -1.5).sum(1).max())
    report(tag + " mvrc_logits", eng.mvrc_logits_copy[:, :C].view(B, R, C)[:, :max_len], outputs["mvrc_logits"][:, :max_len], 2e-3, logit_rtol)
    report(tag + " encoder output", eng.X[-1].view(B, eng.S, -1)[:, :outputs["sequence_output"].shape[1]] *
           eng.lay["attn_mask"].view(B, eng.S, 1)[:, :outputs["sequence_output"].shape[1]].to(pkg("ops").BF16),
           outputs["sequence_output"] * (eng.lay["attn_mask"].cpu().view(B, eng.S, 1)[:, :outputs["sequence_output"].shape[1]]), 2e-3, 1.5e-2)
    for k in ("mlm_loss", "mvrc_loss") + (("relationship_loss",) if cfg.with_rel_loss else ()):
        ref = float(outputs[k])
        print("%s %s: hip %.6f oracle %.6f" % (tag, k, lv[k], ref))
        assert abs(lv[k] - ref) <= loss_tol * max(1.0, abs(ref)), (k, lv[k], ref)
    if cfg.with_rel_loss:
        report(tag + " relationship_logits", eng.rel_logits_copy[:, :2], outputs["relationship_logits"], 2e-3, 1e-2)
    gn = eng.grad_norm()
    print("%s grad_norm: hip %.6f oracle %.6f rel %.3e" % (tag, gn, norm, abs(gn - norm) / norm))
    assert abs(gn - norm) <= norm_tol * norm
    worst = []
    for name, g in eng.grads().items():
        ref = grads[name]
        if float(ref.norm()) < 1e-6 * norm:
            continue
        worst.append((rel_fro(g, ref), name))
    worst.sort(reverse=True)
    for e, n in worst[:8]:
        print("   rel-fro grad err %.3e  %s" % (e, n))
    assert worst[0][0] <= grad_tol, worst[:5]
    return eng


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_engine_matches_reference_golden(path):
    z = np.load(path, allow_pickle=False)
    kw = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        kw[str(k)] = bool(v) if str(k).startswith("with_") or str(k) == "multitask" else int(v)
    if kw.get("multitask"):
        pytest.skip("covered by test_engine_multitask_matches_reference")
    cfg = O.VLBertConfig(**kw)
    params = O.init_params(cfg, seed=int(z["pseed"]))
    batch = tuple(torch.from_numpy(z["in_" + k]) for k in
                  ("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels"))
    name = os.path.basename(path)[:-4]
    eng = check_against_oracle(name, cfg, params, batch)
    # and directly against what the reference itself produced
    B, T, R = int(z["B"]), int(z["T"]), int(z["R"])
    V = cfg.vocab_size
    report(name + " mlm_logits vs REFERENCE", eng.mlm_logits_copy[:, :V].view(B, T, V), torch.from_numpy(z["mlm_logits"]), 2e-3, 1e-2)
    if cfg.with_rel_loss:
        report(name + " relationship_logits vs REFERENCE", eng.rel_logits_copy[:, :2], torch.from_numpy(z["relationship_logits"]), 2e-3, 1e-2)
    lv = eng.loss_values()
    assert abs(lv["loss"] - float(z["loss"])) <= 1e-2 * float(z["loss"])
    assert abs(eng.grad_norm() - float(z["grad_norm"])) <= 1e-2 * float(z["grad_norm"])
    for n in z["names"]:
        n = str(n)
        g = eng.g32[n].detach().double().cpu().reshape(-1) / eng.loss_scale
        stride = max(1, g.numel() // SAMPLE)
        smp = g[::stride][:SAMPLE].float().numpy()
        ref = z["g_smp/" + n]
        if np.linalg.norm(ref) < 1e-6 * float(z["grad_norm"]):
            continue
        err = np.linalg.norm(smp - ref) / max(np.linalg.norm(ref), 1e-12)
        assert err <= 4e-2, (n, err)


def test_engine_matches_reference_at_the_benched_dimensions():
    """The HIP engine DIRECTLY against what the real reference produced at BASELINE.json's headline model (12 layers, H = 768, 12 heads,
    V = 30522, C = 1601, 64 text + 36 regions; batch 2, digest fixture tests/golden/c2/c2_headline.npz from oracle/make_golden.py c2):
    losses and gradient norm at north_star's 1e-2, the stored 4096-sample strides of both logit tensors in relative Frobenius norm, the
    256-sample stride of every parameter gradient -- the same statement test_engine_matches_reference_golden makes for the toy shapes."""
    from tests.test_oracle_golden import c2_digest, load_c2_case
    z, cfg, params, batch = load_c2_case()
    B, T, R = int(z["B"]), int(z["T"]), int(z["R"])
    eng = make_engine(cfg, B, T, R, train=False)
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    eng.set_batch(*[t.to(dev()) for t in batch])
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    torch.cuda.synchronize()
    V, C = cfg.vocab_size, cfg.visual_region_classes
    lv = eng.loss_values()
    for k in ("mlm_loss", "mvrc_loss", "loss"):
        print("c2 dims %s: hip %.6f REFERENCE %.6f" % (k, lv[k], float(z[k])))
        assert abs(lv[k] - float(z[k])) <= 1e-2 * max(1.0, abs(float(z[k]))), k
    gn = eng.grad_norm()
    print("c2 dims grad_norm: hip %.6f REFERENCE %.6f" % (gn, float(z["grad_norm"])))
    assert abs(gn - float(z["grad_norm"])) <= 1e-2 * float(z["grad_norm"])
    shape = tuple(int(v) for v in z["mvrc_logits_shape"])          # (B, max objects, C): the reference trims to the longest box list
    for k, got in (("mlm_logits", eng.mlm_logits_copy[:, :V].view(B, T, V)), ("mvrc_logits", eng.mvrc_logits_copy[:, :C].view(B, R, C)[:, :shape[1]])):
        assert tuple(got.shape) == tuple(int(v) for v in z[k + "_shape"]), k
        smp = c2_digest(got.float().cpu(), 4096)[1]
        ref = z[k + "_smp"]
        err = np.linalg.norm(smp - ref) / np.linalg.norm(ref)
        print("c2 dims %s sample rel-fro vs REFERENCE %.3e, max |err| %.3e of max |ref| %.3e" % (k, err, np.abs(smp - ref).max(), np.abs(ref).max()))
        assert err <= 1e-2, (k, err)
        assert np.abs(smp - ref).max() <= 2e-3 + 1e-2 * np.abs(ref).max(), k
    worst = []
    for n in z["names"]:
        n = str(n)
        ref = z["g_smp/" + n]
        if float(z["g_stat/" + n][0]) < 1e-6 * float(z["grad_norm"]) or np.linalg.norm(ref) < 1e-9:
            continue
        smp = c2_digest(eng.g32[n].detach().cpu() / eng.loss_scale, 256)[1]
        worst.append((float(np.linalg.norm(smp - ref) / np.linalg.norm(ref)), n))
    worst.sort(reverse=True)
    for e, n in worst[:6]:
        print("   rel-fro (256 samples) grad err %.3e  %s" % (e, n))
    assert worst[0][0] <= 8e-2, worst[:5]          # (256-sample strides at batch 2: noisier than the full-tensor 5e-2 of the oracle checks)


def test_engine_multitask_matches_reference():
    """ResNetVLBERTForPretrainingMultitask: B image-caption samples + B_aux text-only samples in one
    encoder pass, three losses; against the oracle and the real reference's fixture."""
    path = os.path.join(os.path.dirname(__file__), "golden", "multitask_small.npz")
    z = np.load(path, allow_pickle=False)
    kw = {str(k): (bool(v) if str(k) == "multitask" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg = O.VLBertConfig(**kw)
    params = O.init_params(cfg, seed=int(z["pseed"]))
    batch = tuple(torch.from_numpy(z["in_" + k]) for k in
                  ("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels",
                   "aux_text", "aux_mlm_labels"))
    B, T, R = int(z["B"]), int(z["T"]), int(z["R"])
    Ba, Ta = [int(x) for x in z["aux_shape"]]
    Tm = max(T, Ta)
    eng = make_engine(cfg, B, Tm, R, train=False, B_aux=Ba)
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    eng.set_batch(*[t.to(dev()) for t in batch])
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    torch.cuda.synchronize()
    outputs, loss, grads, norm = O.loss_and_grads(params, cfg, batch, train=False)
    V, C = cfg.vocab_size, cfg.visual_region_classes
    logits = eng.mlm_logits_copy[:, :V].view(B + Ba, Tm, V)
    report("multitask mlm_logits_wvc vs REFERENCE", logits[:B, :z["mlm_logits_wvc"].shape[1]], torch.from_numpy(z["mlm_logits_wvc"]), 2e-3, 1e-2)
    report("multitask mlm_logits_aux vs REFERENCE", logits[B:, :z["mlm_logits_aux"].shape[1]], torch.from_numpy(z["mlm_logits_aux"]), 2e-3, 1e-2)
    max_len = z["mvrc_logits"].shape[1]
    report("multitask mvrc_logits vs REFERENCE", eng.mvrc_logits_copy[:, :C].view(B, R, C)[:, :max_len], torch.from_numpy(z["mvrc_logits"]), 2e-3, 1e-2)
    lv = eng.loss_values()
    for k in ("mlm_loss_wvc", "mlm_loss_aux", "mvrc_loss"):
        print("multitask %s: hip %.6f reference %.6f oracle %.6f" % (k, lv[k], float(z[k]), float(outputs[k])))
        assert abs(lv[k] - float(z[k])) <= 1e-2 * max(1.0, abs(float(z[k])))
    assert abs(lv["loss"] - float(z["loss"])) <= 1e-2 * float(z["loss"])
    gn = eng.grad_norm()
    print("multitask grad_norm: hip %.6f reference %.6f" % (gn, float(z["grad_norm"])))
    assert abs(gn - float(z["grad_norm"])) <= 1e-2 * float(z["grad_norm"])
    worst = []
    for name, g in eng.grads().items():
        ref = grads[name]
        if float(ref.norm()) < 1e-6 * norm:
            continue
        worst.append((rel_fro(g, ref), name))
    worst.sort(reverse=True)
    for e, n in worst[:6]:
        print("   rel-fro grad err %.3e  %s" % (e, n))
    assert worst[0][0] <= 5e-2, worst[:5]
    assert rel_fro(eng.grads()["aux_text_visual_embedding.weight"], grads["aux_text_visual_embedding.weight"]) <= 5e-2


def test_engine_c1_shape_vs_oracle():
    """BASELINE.json configs[0] shape: VL-BERT-base 2-layer, 32 text + 10 regions, batch 4 (ragged)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    params = O.init_params(cfg, seed=2)
    batch = syn.make_batch(4, 32, 10, seed=7, ragged=True)
    check_against_oracle("C1", cfg, params, batch)


@pytest.mark.parametrize("schedule", [None, "triangle"])
def test_engine_optimizer_step_matches_oracle(schedule):
    """clip + AdamW + bf16 refresh on the engine's own gradients (gradient parity is checked above; Adam's first
    step is sign-like, so feeding the ORACLE gradients instead would measure bf16 sign flips, not the optimizer)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=4)
    batch = syn.make_batch(2, 16, 6, seed=8, ragged=True)
    lr, wd, max_norm = 1e-3, 1e-2, 1.0
    # "triangle": WarmupLinearSchedule evaluated on the device from the step counter (pretrain/function/train.py:316-320)
    warm, total = 3, 10
    eng = make_engine(cfg, 2, 16, 6, train=False, lr=lr, weight_decay=wd, max_grad_norm=max_norm,
                      lr_schedule=schedule, warmup_steps=warm, t_total=total)
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    eng.set_batch(*[t.to(dev()) for t in batch])
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    ref = {k: p.clone() for k, p in params.items()}
    for step in (1, 2):
        eng.zero_grad()
        eng.forward(train=False)
        eng.backward(train=False)
        torch.cuda.synchronize()
        grads = {k: g.cpu() for k, g in eng.grads().items()}
        coef = O.clip_coef(eng.grad_norm(), max_norm)
        eng.optimizer_step()
        torch.cuda.synchronize()
        worst = (0.0, "")
        for n in ref:
            lr_k = lr * (O.warmup_linear_lr(step, warm, total) if schedule else 1.0)
            O.adamw_step(ref[n], grads[n] * coef, m[n], v[n], step, lr_k, eps=1e-6, weight_decay=wd)
            err = float((eng.w32[n].cpu() - ref[n]).abs().max())
            worst = max(worst, (err, n))
            assert torch.equal(eng.w16[n].cpu(), eng.w32[n].cpu().to(pkg("ops").BF16)), n
        print("optimizer step %d: worst |p_hip - p_oracle| = %.3e (%s)" % (step, worst[0], worst[1]))
        assert worst[0] < 2e-6, worst
    assert float(eng.adam[5]) == 2.0
    # transposed weight copies were refreshed
    n = "vlbert.encoder.layer.0.output.dense.weight"
    assert torch.equal(eng.wT[n].cpu(), eng.w16[n].cpu().t().contiguous())


def test_dropout_training_step_runs_and_is_deterministic():
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    params = O.init_params(cfg, seed=2)
    batch = syn.make_batch(4, 32, 10, seed=7, ragged=False)
    vals = []
    for _ in range(2):
        eng = make_engine(cfg, 4, 32, 10, train=True, seed=99)
        eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
        eng.set_batch(*[t.to(dev()) for t in batch])
        eng.zero_grad()
        eng.forward(True)
        eng.backward(True)
        torch.cuda.synchronize()
        lv = eng.loss_values()
        vals.append((lv["loss"], eng.grad_norm()))
        assert np.isfinite(lv["loss"]) and np.isfinite(vals[-1][1])
    print("dropout step:", vals)
    assert abs(vals[0][0] - vals[1][0]) < 1e-4 and abs(vals[0][1] - vals[1][1]) < 1e-3 * vals[0][1]
    eval_loss = O.loss_and_grads(params, cfg, batch, train=False)[1]
    assert abs(vals[0][0] - float(eval_loss)) < 0.5   # dropout perturbs, not destroys


def _module_config(cfg):
    class A(dict):
        __getattr__ = dict.__getitem__
    return A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=True, IMAGE_SEMANTIC=False, WITH_REL_LOSS=False, WITH_MLM_LOSS=True,
                       WITH_MVRC_LOSS=True,
                       VLBERT=A(hidden_size=cfg.hidden_size, visual_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                                vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings,
                                type_vocab_size=3, visual_region_classes=cfg.visual_region_classes, visual_ln=True,
                                with_pooler=False, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                                initializer_range=0.02, visual_scale_text_init=0.0, visual_scale_object_init=0.0)))


def test_dropin_multitask_module_matches_reference():
    """ResNetVLBERTForPretrainingMultitask mirror: forward(image, ..., *aux) contract, output keys, losses, autograd."""
    M = pkg("pretrain.modules")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "multitask_small.npz"), allow_pickle=False)
    kw = {str(k): (bool(v) if str(k) == "multitask" else int(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg = O.VLBertConfig(**kw)
    net = M.ResNetVLBERTForPretrainingMultitask(_module_config(cfg))
    ref_names = set(str(n) for n in z["names"])
    assert "aux_text_visual_embedding.weight" in ref_names
    assert set(n for n, _ in net.named_parameters()) == ref_names
    params = O.init_params(cfg, seed=int(z["pseed"]))
    net.load_state_dict(params)
    net.eval()
    batch = tuple(torch.from_numpy(z["in_" + k]).to(dev()) for k in
                  ("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels",
                   "aux_text", "aux_mlm_labels"))
    outputs, loss = net(None, *batch)
    for k in ("mlm_logits_wvc", "mlm_logits_aux", "mvrc_logits"):
        report("multitask module %s vs REFERENCE" % k, outputs[k], torch.from_numpy(z[k]), 2e-3, 1e-2)
    for k in ("mlm_loss_wvc", "mlm_loss_aux", "mvrc_loss"):
        assert abs(float(outputs[k]) - float(z[k])) <= 1e-2 * max(1.0, abs(float(z[k]))), k
    assert abs(float(loss) - float(z["loss"])) <= 1e-2 * float(z["loss"])
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1e9)
    assert abs(float(total) - float(z["grad_norm"])) <= 1e-2 * float(z["grad_norm"])


def test_dropin_module_matches_reference_training_loop_contract():
    """The nn.Module mirror: reference constructor/forward signature, state-dict keys, autograd + torch optimizer."""
    M = pkg("pretrain.modules")
    syn = pkg("synthetic")
    z = np.load(GOLDEN[[os.path.basename(p) for p in GOLDEN].index("ragged_small.npz")], allow_pickle=False)
    kw = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg = O.VLBertConfig(**kw)

    config = _module_config(cfg)
    net = M.ResNetVLBERTForPretraining(config)
    # state-dict contract: exactly the reference's parameter names (+ the tied decoder alias)
    ref_names = set(str(n) for n in z["names"])
    sd = net.state_dict()
    assert set(sd) == ref_names | {"vlbert.mlm_head.predictions.decoder.weight"}
    assert set(n for n, _ in net.named_parameters()) == ref_names
    params = O.init_params(cfg, seed=int(z["pseed"]))
    net.load_state_dict({**params, "vlbert.mlm_head.predictions.decoder.weight": params["vlbert.word_embeddings.weight"]})
    net.eval()
    batch = tuple(torch.from_numpy(z["in_" + k]).to(dev()) for k in
                  ("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels"))
    outputs, loss = net(None, *batch)
    report("module mlm_logits vs REFERENCE", outputs["mlm_logits"], torch.from_numpy(z["mlm_logits"]), 2e-3, 1e-2)
    report("module mvrc_logits vs REFERENCE", outputs["mvrc_logits"], torch.from_numpy(z["mvrc_logits"]), 2e-3, 1e-2)
    assert abs(float(loss) - float(z["loss"])) <= 1e-2 * float(z["loss"])
    (loss / 2).backward()                                   # upstream scale, as with gradient accumulation
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1e9)
    assert abs(float(total) - 0.5 * float(z["grad_norm"])) <= 1e-2 * 0.5 * float(z["grad_norm"])
    # a torch optimizer steps the flat storage; the next forward sees the new weights
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.step()
    opt.zero_grad()
    outputs2, loss2 = net(None, *batch)
    assert float(loss2) < float(loss)
    loss2.backward()
    assert all(p.grad is not None for p in net.parameters())


def test_dropin_module_train_mode_backward_uses_forward_dropout_masks():
    """net.train(); loss.backward() of the nn.Module mirrors: dropout masks are regenerated from the device seed, so the backward
    must see the seed its forward used (the mirrors used to advance it in between).  Gradients of the mirror's autograd path must
    equal engine.forward / engine.backward run directly under the same seed; the next training forward draws fresh masks."""
    M = pkg("pretrain.modules")
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    net = M.ResNetVLBERTForPretraining(_module_config(cfg))
    net.load_state_dict(O.init_params(cfg, seed=81))
    batch = tuple(t.to(dev()) for t in syn.make_batch(4, 32, 10, seed=82, ragged=True))
    net.train()
    losses = []
    for it in range(2):
        net.zero_grad()
        outputs, loss = net(None, *batch)
        loss.backward()
        torch.cuda.synchronize()
        got = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        eng = next(iter(net._engines.values()))
        seed_now = int(eng.seed.cpu())
        eng.P.grad.zero_()
        eng._fresh_grads = False
        eng.forward(train=True)
        eng.backward(train=True)
        torch.cuda.synchronize()
        assert int(eng.seed.cpu()) == seed_now
        assert abs(float(loss) - eng.loss_values()["loss"]) < 1e-5 * max(1.0, abs(float(loss)))
        worst = max((rel_fro(got[n], eng.g32[n]), n) for n in got if float(eng.g32[n].norm()) > 0)
        print("train-mode mirror vs engine (same seed), iteration %d: worst rel-fro %.3e (%s)" % ((it,) + worst))
        assert worst[0] < 1e-4, worst
        losses.append(float(loss))
    assert losses[0] != losses[1]          # second training forward advanced the seed: different masks
    # module-API mirror (the VQA wrapper's net.vlbert): same property through _CoreFn
    VL = pkg("common.visual_linguistic_bert")
    z, ccfg, params, ins = _core_fixture()
    core = VL.VisualLinguisticBertForPretraining(_module_config(ccfg)["NETWORK"]["VLBERT"], with_rel_head=False)
    core.load_state_dict(params)
    core.train()
    tv = ins[2].clone().requires_grad_(True)
    ovl = ins[4].clone().requires_grad_(True)
    _, mlm, mvrc = core(ins[0], ins[1], tv, ins[3], ovl, ins[5], output_all_encoded_layers=False, output_text_and_object_separately=True)
    g = torch.Generator().manual_seed(9)
    c0 = torch.randn(mlm.shape, generator=g).to(dev()) * 1e-2
    c1 = torch.randn(mvrc.shape, generator=g).to(dev()) * 1e-2
    ((mlm * c0).sum() + (mvrc * c1).sum()).backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().clone() for n, p in core.named_parameters()}
    eng = next(iter(core._engines.values()))
    eng.P.grad.zero_()
    eng._fresh_grads = False
    eng.forward_core(train=True)
    pad = torch.nn.functional.pad           # the mirror rounds (T, R) up to its shape buckets: cotangents of the extra positions are 0
    eng.backward_core(pad(c0, (0, 0, 0, eng.T - c0.shape[1])), pad(c1, (0, 0, 0, eng.R - c1.shape[1])), None, train=True)
    torch.cuda.synchronize()
    named = eng.P.named(eng.P.grad)
    worst = max((rel_fro(got[n], named["vlbert." + n]), n) for n in got if float(named["vlbert." + n].norm()) > 0)
    print("train-mode module-API mirror vs engine (same seed): worst rel-fro %.3e (%s)" % worst)
    assert worst[0] < 1e-4, worst


def test_engine_accepts_batches_narrower_than_its_buffers():
    """set_batch with a collated batch that has fewer text columns and fewer box slots than the engine was built for (the collator
    pads to each batch's own maxima): the engine completes them with the collator's padding markers; losses and every gradient equal
    those of an engine built for the batch's exact shape."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    params = O.init_params(cfg, seed=61)
    B, T, R = 3, 24, 9
    batch = [t.to(dev()) for t in syn.make_batch(B, T, R, seed=62, ragged=True)]
    res = []
    for Te, Re in ((T, R), (T + 8, R + 3)):
        eng = make_engine(cfg, B, Te, Re, train=False)
        eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
        if (Te, Re) != (T, R):        # stale content in the extra slots must not leak: fill them with a full-width batch first
            eng.set_batch(*[t.to(dev()) for t in syn.make_batch(B, Te, Re, seed=63)])
        eng.set_batch(*batch)
        eng.zero_grad()
        eng.forward(train=False)
        eng.backward(train=False)
        torch.cuda.synchronize()
        res.append((eng.loss_values(), {k: v.detach().clone() for k, v in eng.g32.items()}))
    (l0, g0), (l1, g1) = res
    for k in ("loss", "mlm_loss", "mvrc_loss"):
        assert abs(l0[k] - l1[k]) <= 1e-3 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    worst = max((rel_fro(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 0)
    print("wider engine vs exact-shape engine: worst gradient rel-fro %.3e (%s)" % worst)
    assert worst[0] < 2e-2, worst


def test_mirror_shape_buckets_do_not_change_results(monkeypatch):
    """The collators pad each batch to its own longest text / largest box count, so the mirrors see a new (T, R) almost every batch; they
    round both up to a bucket (masked positions, outputs sliced back) and keep a bounded number of engines.  Exact shapes
    (VLB_MIRROR_BUCKETS=1,1,8) and bucketed shapes (default 8,4,8 -> T 11 -> 16, R 6 -> 8) must agree: outputs, and gradients w.r.t.
    parameters and both embedding inputs; the engine cache stays within its cap over many shapes."""
    VL = pkg("common.visual_linguistic_bert")
    z, ccfg, params, ins = _core_fixture()
    res = {}
    for mode in ("1,1,8", "8,4,8"):
        monkeypatch.setenv("VLB_MIRROR_BUCKETS", mode)
        core = VL.VisualLinguisticBertForPretraining(_module_config(ccfg)["NETWORK"]["VLBERT"], with_rel_head=False)
        core.load_state_dict(params)
        core.eval()
        tv, ovl = ins[2].clone().requires_grad_(True), ins[4].clone().requires_grad_(True)
        _, mlm, mvrc = core(ins[0], ins[1], tv, ins[3], ovl, ins[5], output_all_encoded_layers=False, output_text_and_object_separately=True)
        g = torch.Generator().manual_seed(9)
        c0 = torch.randn(mlm.shape, generator=g).to(dev()) * 1e-2
        c1 = torch.randn(mvrc.shape, generator=g).to(dev()) * 1e-2
        ((mlm * c0).sum() + (mvrc * c1).sum()).backward()
        torch.cuda.synchronize()
        key = next(iter(core._engines))
        res[mode] = (mlm.detach(), mvrc.detach(), tv.grad.clone(), ovl.grad.clone(),
                     {n: p.grad.detach().clone() for n, p in core.named_parameters()}, key)
    exact, buck = res["1,1,8"], res["8,4,8"]
    assert exact[5][1:3] == (11, 6) and buck[5][1:3] == (16, 8), (exact[5], buck[5])
    report("bucketed vs exact mlm logits", buck[0], exact[0], 2e-2, 1e-2)
    report("bucketed vs exact mvrc logits", buck[1], exact[1], 2e-2, 1e-2)
    assert rel_fro(buck[2], exact[2]) < 2e-2 and rel_fro(buck[3], exact[3]) < 2e-2
    worst = max((rel_fro(buck[4][n], exact[4][n]), n) for n in exact[4] if float(exact[4][n].norm()) > 0)
    print("bucketed vs exact shapes: worst parameter-gradient rel-fro %.3e (%s)" % worst)
    assert worst[0] < 2e-2, worst
    # bounded cache: 12 distinct text lengths -> at most 8 engines (here even fewer buckets)
    monkeypatch.setenv("VLB_MIRROR_BUCKETS", "1,1,3")
    core = VL.VisualLinguisticBert(_module_config(ccfg)["NETWORK"]["VLBERT"])
    core.eval()
    H = ccfg.hidden_size
    for T in range(4, 16):
        ids = torch.randint(5, 100, (2, T), device=dev())
        core(ids, torch.zeros_like(ids), torch.zeros(2, T, H, device=dev()), torch.ones(2, T, dtype=torch.bool, device=dev()),
             torch.zeros(2, 3, 2 * H, device=dev()), torch.ones(2, 3, dtype=torch.bool, device=dev()), output_all_encoded_layers=False,
             output_text_and_object_separately=True)
    assert len(core._engines) == 3, len(core._engines)


def _core_fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "core", "core_small.npz"), allow_pickle=False)
    cfg = O.VLBertConfig(**{str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])})
    params = {k[len("vlbert."):]: v for k, v in O.init_params(cfg, seed=int(z["pseed"])).items() if k.startswith("vlbert.")}
    ins = [torch.from_numpy(z["in_" + k]).to(dev()) for k in ("text_ids", "text_type", "text_vis", "text_mask", "obj_vl", "obj_mask")]
    return z, cfg, params, ins


def test_module_api_pretraining_heads_match_reference():
    """common.visual_linguistic_bert.VisualLinguisticBertForPretraining mirror: logits, gradients of the two embedding
    inputs and of the parameters against the fixture produced by the reference module itself."""
    VL = pkg("common.visual_linguistic_bert")
    z, cfg, params, ins = _core_fixture()
    vcfg = _module_config(cfg)["NETWORK"]["VLBERT"]
    net = VL.VisualLinguisticBertForPretraining(vcfg, with_rel_head=False)
    assert set(n for n, _ in net.named_parameters()) == set(str(n) for n in z["names"])
    net.load_state_dict(params)
    net.eval()
    tv = ins[2].clone().requires_grad_(True)
    ovl = ins[4].clone().requires_grad_(True)
    rel, mlm, mvrc = net(ins[0], ins[1], tv, ins[3], ovl, ins[5])
    assert rel is None
    tm, om = torch.from_numpy(z["in_text_mask"]), torch.from_numpy(z["in_obj_mask"])
    report("module-API mlm_logits vs REFERENCE (valid text positions)", mlm.cpu() * tm.unsqueeze(-1),
           torch.from_numpy(z["mlm_logits"]) * tm.unsqueeze(-1), 2e-3, 1e-2)
    report("module-API mvrc_logits vs REFERENCE", mvrc, torch.from_numpy(z["mvrc_logits"]), 2e-3, 1e-2)
    obj = (mlm * torch.from_numpy(z["w_mlm"]).to(dev())).sum() + (mvrc * torch.from_numpy(z["w_mvrc"]).to(dev())).sum()
    assert abs(float(obj) - float(z["objective"])) <= 2e-2 * max(1.0, abs(float(z["objective"])))
    obj.backward()
    e_tv = rel_fro(tv.grad, torch.from_numpy(z["d_text_vis"]))
    e_ovl = rel_fro(ovl.grad, torch.from_numpy(z["d_obj_vl"]))
    print("module-API input gradients: rel-fro err text_visual %.3e object_vl %.3e" % (e_tv, e_ovl))
    assert e_tv <= 3e-2 and e_ovl <= 3e-2
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1e9)
    print("module-API grad_norm hip %.5f reference %.5f" % (float(total), float(z["grad_norm"])))
    assert abs(float(total) - float(z["grad_norm"])) <= 1e-2 * float(z["grad_norm"])
    named = dict(net.named_parameters())
    for n in z["names"]:
        n = str(n)
        ref = z["g_smp/" + n]
        if np.linalg.norm(ref) < 1e-6 * float(z["grad_norm"]):
            continue
        g = named[n].grad.detach().double().cpu().reshape(-1)
        smp = g[::max(1, g.numel() // SAMPLE)][:SAMPLE].float().numpy()
        err = np.linalg.norm(smp - ref) / max(np.linalg.norm(ref), 1e-12)
        assert err <= 5e-2, (n, err)


def test_module_api_hidden_states_match_oracle():
    """VisualLinguisticBert mirror (no heads): text / object hidden states and the input gradients for a random cotangent."""
    VL = pkg("common.visual_linguistic_bert")
    z, cfg, params, ins = _core_fixture()
    net = VL.VisualLinguisticBert(_module_config(cfg)["NETWORK"]["VLBERT"])
    net.load_state_dict({k: v for k, v in params.items() if not k.startswith(("mlm_head.", "mvrc_head."))})
    net.eval()
    tv = ins[2].clone().requires_grad_(True)
    ovl = ins[4].clone().requires_grad_(True)
    text_out, obj_out, pooled = net(ins[0], ins[1], tv, ins[3], ovl, ins[5], output_all_encoded_layers=False,
                                    output_text_and_object_separately=True)
    assert pooled is None
    p = {k: v.clone().requires_grad_(True) for k, v in O.init_params(cfg, seed=int(z["pseed"])).items()}
    tvc, ovlc = ins[2].cpu().clone().requires_grad_(True), ins[4].cpu().clone().requires_grad_(True)
    rt, ro, _, _ = O.vlbert_forward(p, cfg, ins[0].cpu(), ins[1].cpu(), tvc, ins[3].cpu(), ovlc, ins[5].cpu(), False)
    tm = ins[3].cpu().unsqueeze(-1).float()
    report("module-API text_out vs oracle (valid positions)", text_out.cpu() * tm, rt * tm, 2e-3, 1.5e-2)
    report("module-API object_out vs oracle", obj_out, ro, 2e-3, 1.5e-2)
    g = torch.Generator().manual_seed(5)
    ct, co = torch.randn(rt.shape, generator=g) * tm, torch.randn(ro.shape, generator=g) * ins[5].cpu().unsqueeze(-1).float()
    ((rt * ct).sum() + (ro * co).sum()).backward()
    ((text_out * ct.to(dev())).sum() + (obj_out * co.to(dev())).sum()).backward()
    e_tv, e_ovl = rel_fro(tv.grad, tvc.grad), rel_fro(ovl.grad, ovlc.grad)
    print("module-API (hidden) input gradients: rel-fro err text_visual %.3e object_vl %.3e" % (e_tv, e_ovl))
    assert e_tv <= 3e-2 and e_ovl <= 3e-2
    gw = dict(net.named_parameters())["encoder.layer.0.intermediate.dense.weight"].grad
    assert rel_fro(gw, p["vlbert.encoder.layer.0.intermediate.dense.weight"].grad) <= 5e-2


def _language_checkpoint(params, heads):
    """`params` (the mirror's names) re-keyed the way a language-only BERT checkpoint names them (TF-style gamma / beta)."""
    tf = lambda k: k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
    ck = {"optimizer.state": torch.zeros(1)}
    for k, v in params.items():
        if k.startswith(("encoder.", "pooler.")):
            ck["bert." + tf(k)] = v.clone()
        elif k in ("word_embeddings.weight", "position_embeddings.weight"):
            ck["bert.embeddings." + k] = v.clone()
        elif k == "token_type_embeddings.weight":
            ck["bert.embeddings." + k] = v[:2].clone()
        elif k.startswith("embedding_LayerNorm."):
            ck["bert.embeddings." + tf(k[len("embedding_"):])] = v.clone()
        elif heads and k.startswith("mlm_head.predictions."):
            ck["cls.predictions." + tf(k[len("mlm_head.predictions."):])] = v.clone()
    if heads:
        ck["cls.predictions.decoder.weight"] = params["word_embeddings.weight"].clone()
    return ck


def test_module_api_language_pretrained_initialisation(tmp_path, capsys):
    """language_pretrained_model_path of the mirrors (common/visual_linguistic_bert.py:76-78,243-309,382-469; key mapping pinned
    against the reference's loader on CPU, tests/test_host_logic_cpu.py): the checkpoint's tensors land in the flat master buffer, the
    rest keeps its initialisation, and the engine computes with them; the pre-training wrapper picks BERT_PRETRAINED-<epoch>.model."""
    VL = pkg("common.visual_linguistic_bert")
    z, cfg, params, ins = _core_fixture()
    vcfg = _module_config(cfg)["NETWORK"]["VLBERT"]
    for heads in (False, True):
        ck = _language_checkpoint(params, heads)
        path = str(tmp_path / ("lang%d.bin" % heads))
        torch.save(ck, path)
        torch.manual_seed(3)
        net = (VL.VisualLinguisticBertForPretraining(vcfg, path, with_rel_head=False) if heads else VL.VisualLinguisticBert(vcfg, path))
        assert "Unexpected keys: ['optimizer.state']" in capsys.readouterr().out
        named = dict(net.named_parameters())
        loaded = [k for k in params if k.startswith(("encoder.", "embedding_LayerNorm.")) or k in ("word_embeddings.weight", "position_embeddings.weight")
                  or (heads and k.startswith("mlm_head.predictions."))]
        assert len(loaded) > 30
        for k in loaded:
            assert torch.equal(named[k].detach().cpu(), params[k]), k
        assert torch.equal(named["token_type_embeddings.weight"][:2].detach().cpu(), params["token_type_embeddings.weight"][:2])
        assert float(named["visual_ln_text.weight"].abs().max()) == 0.0            # visual_scale_text_init: untouched by the checkpoint
        twin = (VL.VisualLinguisticBertForPretraining(vcfg, with_rel_head=False) if heads else VL.VisualLinguisticBert(vcfg))
        twin.load_state_dict(net.state_dict())
        net.eval(), twin.eval()
        a = net(*ins) if heads else net(*ins, output_all_encoded_layers=False, output_text_and_object_separately=True)
        b = twin(*ins) if heads else twin(*ins, output_all_encoded_layers=False, output_text_and_object_separately=True)
        for x, y in zip(a, b):
            assert (x is None and y is None) or torch.equal(x, y)
    # the wrapper's checkpoint choice + from_scratch
    M = pkg("pretrain.modules")
    mc = _module_config(cfg)
    os.replace(str(tmp_path / "lang1.bin"), str(tmp_path / "bert-0003.model"))
    mc["NETWORK"].update(BERT_PRETRAINED=str(tmp_path / "bert"), BERT_PRETRAINED_EPOCH=3, BERT_MODEL_NAME="bert-base-uncased")
    wrap = M.ResNetVLBERTForPretraining(mc)
    wn = dict(wrap.named_parameters())
    for k in ("encoder.layer.1.output.dense.weight", "word_embeddings.weight", "mlm_head.predictions.transform.LayerNorm.bias"):
        assert torch.equal(wn["vlbert." + k].detach().cpu(), params[k]), k
    mc["NETWORK"]["VLBERT"]["from_scratch"] = True
    wrap2 = M.ResNetVLBERTForPretraining(mc)
    assert not torch.equal(dict(wrap2.named_parameters())["vlbert.word_embeddings.weight"].detach().cpu(), params["word_embeddings.weight"])


def test_module_api_pooler_and_relationship_head_vs_oracle():
    """with_pooler + with_rel_head through the module API: pooled output (base class) and relationship logits
    (pretraining class) and their gradients against the oracle (whose pooler / relationship head are pinned by the
    reference's full_rel fixture)."""
    VL = pkg("common.visual_linguistic_bert")
    z, cfg0, _, ins = _core_fixture()
    cfg = O.VLBertConfig(**{**{str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}, "with_pooler": True, "with_rel_loss": True})
    full = O.init_params(cfg, seed=5)
    params = {k[len("vlbert."):]: v for k, v in full.items() if k.startswith("vlbert.")}
    vcfg = _module_config(cfg)["NETWORK"]["VLBERT"]
    vcfg["with_pooler"] = True
    net = VL.VisualLinguisticBertForPretraining(vcfg, with_rel_head=True)
    net.load_state_dict(params)
    net.eval()
    tv, ovl = ins[2].clone().requires_grad_(True), ins[4].clone().requires_grad_(True)
    rel, mlm, mvrc = net(ins[0], ins[1], tv, ins[3], ovl, ins[5])
    p = {k: v.clone().requires_grad_(True) for k, v in full.items()}
    tvc, ovlc = ins[2].cpu().clone().requires_grad_(True), ins[4].cpu().clone().requires_grad_(True)
    rt, ro, pooled, _ = O.vlbert_forward(p, cfg, ins[0].cpu(), ins[1].cpu(), tvc, ins[3].cpu(), ovlc, ins[5].cpu(), False)
    rel_ref = O.linear(pooled, p, "vlbert.relationsip_head.caption_image_relationship")
    report("module-API relationship_logits vs oracle", rel, rel_ref, 2e-3, 1e-2)
    g = torch.Generator().manual_seed(6)
    cr = torch.randn(rel_ref.shape, generator=g)
    (rel_ref * cr).sum().backward()
    (rel * cr.to(dev())).sum().backward()
    e_tv, e_ovl = rel_fro(tv.grad, tvc.grad), rel_fro(ovl.grad, ovlc.grad)
    print("module-API (relationship head only) input gradients: rel-fro err text_visual %.3e object_vl %.3e" % (e_tv, e_ovl))
    assert e_tv <= 3e-2 and e_ovl <= 3e-2
    named = dict(net.named_parameters())
    for n in ("pooler.dense.weight", "pooler.dense.bias", "relationsip_head.caption_image_relationship.weight",
              "encoder.layer.1.output.dense.weight"):
        assert rel_fro(named[n].grad, p["vlbert." + n].grad) <= 5e-2, n
    # base class: pooled output
    base = VL.VisualLinguisticBert(vcfg)
    base.load_state_dict({k: v for k, v in params.items() if not k.startswith(("mlm_head.", "mvrc_head.", "relationsip_head."))})
    base.eval()
    _, _, pooled_hip = base(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], output_all_encoded_layers=False,
                            output_text_and_object_separately=True)
    report("module-API pooled_output vs oracle", pooled_hip, pooled, 2e-3, 1e-2)


def test_module_api_vqa_style_composition_vs_oracle():
    """The call pattern of vqa/modules/resnet_vlbert_for_vqa.py:169-245 on the mirrors: FastRCNN (precomputed branch) ->
    text-visual = obj_reps[:, 0], object_vl = [obj_reps || linguistic embedding] -> VisualLinguisticBert(..., packed
    sequence out) -> hidden state at the answer position.  Values and gradients (downsample weights through the
    FastRCNN node, the torch-side linguistic embedding through d(object_vl), an encoder weight) against the oracle."""
    VL, FR = pkg("common.visual_linguistic_bert"), pkg("common.fast_rcnn")
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=512,
                         max_position_embeddings=64, visual_region_classes=50)
    B, T, R, H = 3, 9, 5, cfg.hidden_size
    full = O.init_params(cfg, seed=17)
    boxes, im_info, text, _, _, _, _ = syn.make_batch(B, T, R, vocab_size=cfg.vocab_size, region_classes=50, seed=4, ragged=True)
    box_mask, text_mask = boxes[:, :, 0] > -1.5, text > 0
    ans_pos = (text_mask.sum(1) - 2).clamp(min=1)
    cot = torch.randn((B, H), generator=torch.Generator().manual_seed(8))

    class A(dict):
        __getattr__ = dict.__getitem__
    frcnn = FR.FastRCNN(A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=True, IMAGE_SEMANTIC=False)), final_dim=H)
    frcnn.load_state_dict({"obj_downsample.1.weight": full["image_feature_extractor.obj_downsample.1.weight"],
                           "obj_downsample.1.bias": full["image_feature_extractor.obj_downsample.1.bias"]})
    vlbert = VL.VisualLinguisticBert(_module_config(cfg)["NETWORK"]["VLBERT"])
    vlbert.load_state_dict({k[len("vlbert."):]: v for k, v in full.items()
                            if k.startswith("vlbert.") and not k.startswith(("vlbert.mlm_head.", "vlbert.mvrc_head."))})
    frcnn.eval(); vlbert.eval()
    ling = full["object_linguistic_embeddings.weight"].clone().to(dev()).requires_grad_(True)

    def compose(fr, vb, ling_w, to):
        obj = fr(None, to(boxes), to(box_mask), to(im_info))["obj_reps"]
        text_vis = obj[:, 0:1].expand(B, T, H)
        ovl = torch.cat((obj, ling_w.view(1, 1, H).expand(B, R, H)), -1)
        seq, pooled = vb(to(text), torch.zeros_like(to(text)), text_vis, to(text_mask), ovl, to(box_mask),
                         output_all_encoded_layers=False)
        return seq[torch.arange(B), to(ans_pos)]

    hm = compose(frcnn, vlbert, ling, lambda t: t.to(dev()))
    p = {k: v.clone().requires_grad_(True) for k, v in full.items()}

    def fr_ref(images, bx, bm, ii):
        return {"obj_reps": O.fast_rcnn_precomputed(p, cfg, bx, bm, ii, False)}

    def vb_ref(ids, types, tv, tm, ovl, om, output_all_encoded_layers=False):
        return O.vlbert_forward(p, cfg, ids, types, tv, tm, ovl, om, False)[3], None
    hm_ref = compose(fr_ref, vb_ref, p["object_linguistic_embeddings.weight"], lambda t: t)
    report("VQA-style hidden state at the answer position vs oracle", hm, hm_ref, 2e-3, 1.5e-2)
    (hm_ref * cot).sum().backward()
    (hm * cot.to(dev())).sum().backward()
    checks = [("obj_downsample.1.weight", getattr(frcnn.obj_downsample, "1").weight.grad, p["image_feature_extractor.obj_downsample.1.weight"].grad),
              ("obj_downsample.1.bias", getattr(frcnn.obj_downsample, "1").bias.grad, p["image_feature_extractor.obj_downsample.1.bias"].grad),
              ("object_linguistic_embeddings", ling.grad, p["object_linguistic_embeddings.weight"].grad),
              ("encoder.layer.0.attention.self.query.weight", dict(vlbert.named_parameters())["encoder.layer.0.attention.self.query.weight"].grad,
               p["vlbert.encoder.layer.0.attention.self.query.weight"].grad)]
    for name, g, ref in checks:
        e = rel_fro(g, ref)
        print("   VQA-style composition: rel-fro grad err %.3e  %s" % (e, name))
        assert e <= 5e-2, name


def test_fast_rcnn_mirror_im_info_row_width():
    """The datasets disagree on im_info's width: (w, h, 1, 1, index) for pre-training / VCR (conceptual_captions.py:138, vcr.py:377),
    (w, h, 1, 1) for VQA (vqa/data/datasets/vqa.py:217).  The coordinate embedding divides by THAT image's (w, h)
    (common/utils/bbox.py:33-65), so the row stride must be honoured: per-image sizes, 4 and 5 columns, against the oracle."""
    FR, syn = pkg("common.fast_rcnn"), pkg("synthetic")
    H, B, T, R = 128, 5, 6, 4
    cfg = O.VLBertConfig(hidden_size=H, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=512,
                         max_position_embeddings=64, visual_region_classes=50)
    full = O.init_params(cfg, seed=23)
    boxes, im_info, _, _, _, _, _ = syn.make_batch(B, T, R, vocab_size=512, region_classes=50, seed=11, ragged=True)
    im_info = im_info.clone()
    im_info[:, 0] = torch.tensor([1000.0, 640.0, 800.0, 1333.0, 500.0])      # a different size for every image
    im_info[:, 1] = torch.tensor([600.0, 480.0, 800.0, 750.0, 375.0])
    box_mask = boxes[:, :, 0] > -1.5

    class A(dict):
        __getattr__ = dict.__getitem__
    fr = FR.FastRCNN(A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=True, IMAGE_SEMANTIC=False)), final_dim=H)
    fr.load_state_dict({"obj_downsample.1.weight": full["image_feature_extractor.obj_downsample.1.weight"],
                        "obj_downsample.1.bias": full["image_feature_extractor.obj_downsample.1.bias"]})
    fr.eval()
    ref = O.fast_rcnn_precomputed(full, cfg, boxes, box_mask, im_info, False)
    for width in (5, 4, 2):
        info = im_info[:, :width].contiguous() if width <= im_info.shape[1] else im_info
        out = fr(None, boxes.to(dev()), box_mask.to(dev()), info.to(dev()))
        report("FastRCNN mirror obj_reps, im_info [B,%d]" % width, out["obj_reps"], ref, 2e-3, 1e-2)


def test_fast_rcnn_mirror_mask_embedding_gradient():
    """FastRCNN mirror with mvrc_ops / mask_visual_embed (common/fast_rcnn.py:170-172): output, obj_reps_raw and the
    gradient that flows back into the mask embedding."""
    FR, syn = pkg("common.fast_rcnn"), pkg("synthetic")
    H, B, T, R = 128, 3, 6, 7
    cfg = O.VLBertConfig(hidden_size=H, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=512,
                         max_position_embeddings=64, visual_region_classes=50)
    full = O.init_params(cfg, seed=19)
    boxes, im_info, _, _, _, mvrc_ops, _ = syn.make_batch(B, T, R, vocab_size=512, region_classes=50, seed=9, ragged=True)
    box_mask = boxes[:, :, 0] > -1.5

    class A(dict):
        __getattr__ = dict.__getitem__
    fr = FR.FastRCNN(A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=True, IMAGE_SEMANTIC=False)), final_dim=H)
    fr.load_state_dict({"obj_downsample.1.weight": full["image_feature_extractor.obj_downsample.1.weight"],
                        "obj_downsample.1.bias": full["image_feature_extractor.obj_downsample.1.bias"]})
    fr.eval()
    emb = bf(full["object_mask_visual_embedding.weight"].clone())
    emb_g = emb.clone().to(dev()).requires_grad_(True)
    out = fr(None, boxes.to(dev()), box_mask.to(dev()), im_info.to(dev()), mvrc_ops=mvrc_ops.to(dev()), mask_visual_embed=emb_g)
    p = {k: v.clone().requires_grad_(True) for k, v in full.items()}
    emb_r = emb.clone().requires_grad_(True)
    bx = boxes.clone()
    feats = torch.where((mvrc_ops == 1).unsqueeze(-1), emb_r.view(1, 1, -1).expand(B, R, -1), bx[:, :, 4:])
    bx = torch.cat((bx[:, :, :4], feats), -1)
    ref = O.fast_rcnn_precomputed(p, cfg, bx, box_mask, im_info, False)
    report("FastRCNN mirror obj_reps vs oracle", out["obj_reps"], ref, 2e-3, 1e-2)
    raw_ref = feats * box_mask.unsqueeze(-1)
    report("FastRCNN mirror obj_reps_raw", out["obj_reps_raw"], raw_ref.detach(), 1e-6, 1e-6)
    # cotangent only on units whose ReLU state cannot differ between bf16 and fp32 arithmetic (|pre-activation| near 0
    # flips ~1 % of the units; with 4 masked regions x 128 units that alone is a 10 % gradient difference)
    hip = out["obj_reps"].detach().cpu()
    safe = (ref.detach() > 0.05) | ((ref.detach() == 0) & (hip == 0))
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * safe
    (ref * cot).sum().backward()
    (out["obj_reps"] * cot.to(dev())).sum().backward()
    e = rel_fro(emb_g.grad, emb_r.grad)
    print("FastRCNN mirror: rel-fro err of d(mask_visual_embed) %.3e" % e)
    assert e <= 3e-2
    assert rel_fro(getattr(fr.obj_downsample, "1").weight.grad, p["image_feature_extractor.obj_downsample.1.weight"].grad) <= 3e-2


def test_side_stream_weight_gradients_match_serial_and_bucket_hook_order():
    """The weight gradients run on a second stream; (a) the result equals the serialised schedule, (b) the
    data-parallel hook sees complete gradients: at every on_layer_done the bucket it would all-reduce is final."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=3)
    params = O.init_params(cfg, seed=12)
    batch = syn.make_batch(8, 64, 36, seed=13, ragged=False)
    grads = []
    for side in (True, False):
        eng = make_engine(cfg, 8, 64, 36, train=True, seed=5)
        if not side:
            eng.side = None
        eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
        eng.set_batch(*[t.to(dev()) for t in batch])
        eng.zero_grad()
        eng.forward(True)
        seen = {}

        def hook(what, eng=eng, seen=seen):
            torch.cuda.current_stream().synchronize()          # what an all-reduce launched here would read
            if isinstance(what, int):
                n = "vlbert.encoder.layer.%d.intermediate.dense.weight" % what
                seen[what] = eng.g32[n].detach().clone()
        eng.backward(True, on_layer_done=hook)
        torch.cuda.synchronize()
        for l, g in seen.items():
            assert torch.equal(g, eng.g32["vlbert.encoder.layer.%d.intermediate.dense.weight" % l]), "layer %d gradient changed after its hook" % l
        grads.append({k: v.clone() for k, v in eng.grads().items()})
    worst = max((rel_fro(grads[0][k], grads[1][k]), k) for k in grads[0] if float(grads[1][k].norm()) > 0)
    print("side-stream vs serial weight gradients: worst rel-fro difference %.3e (%s)" % worst)
    assert worst[0] < 1e-5, worst


def test_engine_maximum_sequence_length_vs_oracle():
    """S = T + R + 1 = 128: the longest packed sequence the fused attention kernel accepts (ragged batch)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=1, max_position_embeddings=128)
    params = O.init_params(cfg, seed=31)
    batch = syn.make_batch(3, 91, 36, seed=32, ragged=True)
    check_against_oracle("S=128", cfg, params, batch)
    E = pkg("engine")
    with pytest.raises(ValueError):
        E.PretrainEngine(E.ModelConfig(num_hidden_layers=1), 1, 156, 100, device="cuda:0")


def test_engine_long_sequence_128_text_100_regions_vs_oracle():
    """S = 128 + 100 + 1 = 229 (the VL-BERT-large VQA / VCR sequence shape of BASELINE.json configs 4-5) through the
    8-key-block attention instantiation, ragged batch, 2 layers."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2, max_position_embeddings=512)
    params = O.init_params(cfg, seed=33)
    batch = syn.make_batch(2, 128, 100, seed=34, ragged=True)
    check_against_oracle("S=229", cfg, params, batch)


def test_engine_large_model_width_vs_oracle():
    """VL-BERT-large layer shape (H = 1024, 16 heads, I = 4096): exercises the wider LayerNorm / embedding register tilings."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=1, vocab_size=2048,
                         visual_region_classes=100)
    params = O.init_params(cfg, seed=33)
    batch = syn.make_batch(2, 24, 9, vocab_size=2048, region_classes=100, seed=34, ragged=True)
    # obj_downsample gradients: a ReLU unit whose pre-activation is within bf16 rounding of 0 flips state against the fp32
    # oracle; each flip changes a full gradient row, so ~1 % flipped units is a 10 % Frobenius difference whatever the batch
    check_against_oracle("large-width", cfg, params, batch, grad_tol=0.12)


def test_gradient_accumulation_over_micro_batches():
    """zero_grad -> (forward, backward) x 2 on different batches: the first backward overwrites the Linear weight gradients,
    the second accumulates; the sum must equal the two single-batch gradients added (common/trainer.py:117-153 semantics)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    params = O.init_params(cfg, seed=41)
    b1 = syn.make_batch(4, 32, 10, seed=42, ragged=True)
    b2 = syn.make_batch(4, 32, 10, seed=43, ragged=True)
    eng = make_engine(cfg, 4, 32, 10, train=False)
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    single = []
    for b in (b1, b2):
        eng.set_batch(*[t.to(dev()) for t in b])
        eng.zero_grad()
        eng.forward(False)
        eng.backward(False)
        torch.cuda.synchronize()
        single.append({k: v.clone() for k, v in eng.grads().items()})
    eng.zero_grad()
    for b in (b1, b2):
        eng.set_batch(*[t.to(dev()) for t in b])
        eng.forward(False)
        eng.backward(False)
    torch.cuda.synchronize()
    acc = eng.grads()
    worst = max((rel_fro(acc[k], single[0][k] + single[1][k]), k) for k in acc if float((single[0][k] + single[1][k]).norm()) > 0)
    print("gradient accumulation: worst rel-fro difference %.3e (%s)" % worst)
    assert worst[0] < 1e-4, worst


def test_train_end2end_entry_point_runs_reference_style_config():
    """python -m vl-bert_amd.pretrain.train_end2end --cfg <reference-style yaml>: 3 optimizer steps of 2 accumulated micro-batches
    (C1-sized: 2 layers, batch 4, 32 + 10), lr following the triangle schedule evaluated on the device."""
    tr = pkg("pretrain.train_end2end")
    cfg = os.path.join(os.path.dirname(__file__), "fixtures", "pretrain_small.yaml")
    eng = tr.main(["--cfg", cfg, "--steps", "3", "--steps-per-epoch", "20", "--text-len", "32", "--regions", "10"])
    torch.cuda.synchronize()
    lv = eng.loss_values()
    assert np.isfinite(lv["loss"]) and lv["mlm_loss"] > 0 and lv["mvrc_loss"] > 0
    base = 1.0e-5 * 4 * 2                                        # TRAIN.LR x batch x accumulate (world 1)
    assert float(eng.adam[5]) == 3.0                             # optimizer steps, not micro-batches
    assert abs(float(eng.adam[0]) - base * O.warmup_linear_lr(3, 4, 10)) < 1e-6 * base


def test_checkpoint_written_by_the_reference_resumes_on_the_engine(tmp_path):
    """tests/golden/checkpoint/ref_small-0000.model was written by the REFERENCE's `Checkpoint` callback after 2 steps of the reference's
    AdamW (oracle/make_checkpoint_golden.py).  The engine must (a) load it -- weights by name, Adam moments by the reference's parameter
    index order, step counter --, (b) continue the trajectory the reference itself continued after reloading that file (its next two
    losses, stored with the fixture), and (c) write a file with the same structure (keys, index order, shapes, group fields), which
    (d) it reads back to the identical state.  Replaces common/callbacks/epoch_end_callbacks/checkpoint.py:12-21 +
    common/utils/load.py:20-54."""
    E, C = pkg("engine"), pkg("common.checkpoint")
    gold = os.path.join(os.path.dirname(__file__), "golden", "checkpoint")
    z = np.load(os.path.join(gold, "ref_small_batch.npz"))
    kw = {k: (int(v) if float(v).is_integer() else float(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    batch = [torch.from_numpy(z["in_%d" % i]) for i in range(7)]
    B, T, R = batch[2].shape[0], batch[2].shape[1], batch[0].shape[1]
    eng = E.PretrainEngine(E.ModelConfig(**kw), B, T, R, device="cuda:0", train=False, lr=2e-3, weight_decay=1e-4, max_grad_norm=10.0)
    path = os.path.join(gold, "ref_small-0000.model")
    ck = C.load_checkpoint(eng, path)
    assert float(eng.adam[5]) == 2.0
    names = C.reference_param_order(eng.P.shapes)
    m = eng.P.named(eng.P.m)
    for i in (0, 5, len(names) - 1):
        assert torch.equal(m[names[i]].cpu(), ck["optimizer"]["state"][i]["exp_avg"]), names[i]
    eng.set_batch(*[t.to(dev()) for t in batch])
    got = []
    for _ in range(2):
        eng.zero_grad(); eng.forward(train=False); eng.backward(train=False); eng.optimizer_step()
        got.append(eng.loss_values()["loss"])
    ref = [float(x) for x in z["losses_after"]]
    print("resumed trajectory: engine %s | reference %s (before the checkpoint: %s)" % (got, ref, list(z["losses_before"])))
    for a, b in zip(got, ref):
        assert abs(a - b) <= 1e-2 * abs(b), (got, ref)
    probe = str(z["probe_name"])
    torch.cuda.synchronize()
    report("resumed AdamW: %s after 2 more steps" % probe, eng.w32[probe], torch.from_numpy(z["probe_after"]), 3e-4, 2e-3)
    # (c) + (d)
    out = C.save_checkpoint(eng, str(tmp_path / "mine"), 1)
    assert out.endswith("mine-0001.model")
    mine = torch.load(out, map_location="cpu", weights_only=False)
    assert list(mine.keys())[:2] == ["state_dict", "optimizer"] and set(mine["state_dict"]) == set(ck["state_dict"])
    for k, t in ck["state_dict"].items():
        assert tuple(mine["state_dict"][k].shape) == tuple(t.shape) and mine["state_dict"][k].dtype == t.dtype, k
    g0, g1 = ck["optimizer"]["param_groups"][0], mine["optimizer"]["param_groups"][0]
    assert g1["params"] == g0["params"] and set(g0) <= set(g1) and g1["betas"] == g0["betas"] and g1["correct_bias"] is True
    assert abs(g1["eps"] - g0["eps"]) < 1e-12 and abs(g1["weight_decay"] - g0["weight_decay"]) < 1e-9
    for i, st in ck["optimizer"]["state"].items():
        assert set(mine["optimizer"]["state"][i]) == set(st) and mine["optimizer"]["state"][i]["step"] == 4
        assert tuple(mine["optimizer"]["state"][i]["exp_avg_sq"].shape) == tuple(st["exp_avg_sq"].shape)
    eng2 = E.PretrainEngine(E.ModelConfig(**kw), B, T, R, device="cuda:0", train=False, lr=2e-3, weight_decay=1e-4, max_grad_norm=10.0)
    C.load_checkpoint(eng2, out)
    assert torch.equal(eng2.P.master, eng.P.master) and torch.equal(eng2.P.m, eng.P.m) and torch.equal(eng2.P.v, eng.P.v)
    assert torch.equal(eng2.adam[1:6], eng.adam[1:6])


def test_train_end2end_writes_epoch_checkpoints_and_auto_resumes(tmp_path):
    """--model-dir: `{prefix}-{epoch:04d}.model` after every epoch, TRAIN.AUTO_RESUME picks the newest one up and the run continues at
    the right optimizer step (the LR schedule follows the restored step counter)."""
    tr = pkg("pretrain.train_end2end")
    with open(os.path.join(os.path.dirname(__file__), "fixtures", "pretrain_small.yaml")) as f:
        text = f.read().replace("END_EPOCH: 1", "END_EPOCH: 10")
    cfg = str(tmp_path / "pretrain_small_10_epochs.yaml")
    with open(cfg, "w") as f:
        f.write(text)
    common = ["--cfg", cfg, "--steps-per-epoch", "4", "--text-len", "32", "--regions", "10", "--model-dir", str(tmp_path / "ckpt")]
    eng = tr.main(common + ["--steps", "4"])            # accumulate 2 -> 2 optimizer steps per epoch -> epochs 0 and 1
    torch.cuda.synchronize()
    files = sorted(f for _, _, fs in os.walk(tmp_path / "ckpt") for f in fs)
    assert [f[-11:] for f in files] == ["-0000.model", "-0001.model"], files
    w_end = eng.P.master.clone()
    eng2 = tr.main(common + ["--steps", "1"])           # resumes from epoch 2 = optimizer step 4
    torch.cuda.synchronize()
    assert float(eng2.adam[5]) == 5.0
    assert not torch.equal(eng2.P.master, w_end)
    base = 1.0e-5 * 4 * 2
    assert abs(float(eng2.adam[0]) - base * O.warmup_linear_lr(5, 4, 20)) < 1e-6 * base


def test_train_end2end_reads_the_reference_data_formats(tmp_path):
    """--data: annotation jsonl + base64 detector records -> dataset -> sampler -> collator -> engine.set_batch -> fused step, through the
    entry point.  (The readers themselves are pinned bit-exactly to the reference's dataset classes on the CPU: tests/test_data_cpu.py.)"""
    import json
    from tests.test_data_cpu import write_config, write_dataset
    tr = pkg("pretrain.train_end2end")
    R = pkg("pretrain.data.records")
    root = write_dataset(str(tmp_path / "cc"))
    cfg = write_config(str(tmp_path / "cfg.yaml"), root, batch=2, seq_len=32)
    eng = tr.main(["--cfg", cfg, "--data", "--steps", "4"])        # 3 batches per epoch: the 4th step opens the next epoch
    torch.cuda.synchronize()
    lv = eng.loss_values()
    assert float(eng.adam[5]) == 4.0 and np.isfinite(lv["loss"]) and lv["mlm_loss"] >= 0 and (eng.T, eng.R) == (32, 32)
    rows = []
    for i in range(6):
        with open(os.path.join(root, "train_frcnn", "%04d.json" % i)) as f:
            rows.append(R.decode_detector_record(json.load(f))["features"])
    rows = torch.from_numpy(np.concatenate(rows)).to(eng.in_boxes.device)
    boxes, ops_ = eng.in_boxes.float(), eng.in_mvrc_ops
    valid = boxes[:, :, 0] > -1.5
    assert valid[:, 0].all() and int(valid.sum()) >= 2 * 5 and bool((eng.in_text[:, 0] == 5).all())      # whole-image slot; [CLS] = id 5 of the fixture vocabulary
    whole = boxes[:, 0, :4].cpu()          # the whole-image box [0, 0, 499, 374] of a 500 x 375 image after Resize(600, 1000) = x 1.6 (flipped or not)
    assert torch.allclose(whole[:, 2] - whole[:, 0], torch.full((2,), 499 * 1.6), atol=1e-3) and torch.allclose(whole[:, 3], torch.full((2,), 374 * 1.6), atol=1e-3)
    assert bool((whole[:, 1] == 0).all()) and bool(((whole[:, 0] == 0) | ((whole[:, 0] - 0.6).abs() < 1e-3)).all())
    for b in range(boxes.shape[0]):
        for k in range(1, boxes.shape[1]):
            if valid[b, k]:
                assert bool((rows == boxes[b, k, 4:]).all(dim=1).any()), (b, k)
    assert ((valid.sum(1) + (eng.in_text > 0).sum(1)) <= 32).all()


def test_train_end2end_multitask_reads_both_data_sets(tmp_path):
    """--data with a DATASET list: image-caption batches from the detector records side by side with text-only batches from the corpus
    (MultiTaskDataLoader), through the multitask engine."""
    from tests.test_data_cpu import write_dataset, write_multitask_config
    tr = pkg("pretrain.train_end2end")
    root = write_dataset(str(tmp_path / "cc"))
    cfg = write_multitask_config(str(tmp_path / "cfg.yaml"), root, batches=(2, 3), seq_len=32)
    eng = tr.main(["--cfg", cfg, "--data", "--steps", "4"])
    torch.cuda.synchronize()
    lv = eng.loss_values()
    assert float(eng.adam[5]) == 4.0 and np.isfinite(lv["loss"]) and (eng.B, eng.Ba) == (2, 3)
    assert bool((eng.in_text[:, 0] > 0).all()) and bool((eng.in_text[2:, :12] > 0).all())      # text-only rows: at least MIN_SEQ_LEN tokens
    assert int((eng.in_mlm_labels[2:] >= 0).sum()) >= 0 and bool((eng.in_boxes[:, 0, 0] > -1.5).all())


def _vqa_config(cfg, classifier, answers, hidden):
    conf = _module_config(cfg)
    conf["NETWORK"].update(BLIND=False, NO_GROUNDING=False, ENABLE_CNN_REG_LOSS=False, CLASSIFIER_TYPE=classifier, CLASSIFIER_DROPOUT=0.1,
                           CLASSIFIER_HIDDEN_SIZE=hidden, IMAGE_FINAL_DIM=cfg.hidden_size)
    conf["NETWORK"]["VLBERT"].update(object_word_embed_mode=2)
    conf["DATASET"] = type(conf)(ANSWER_VOCAB_SIZE=answers)
    return conf


@pytest.mark.parametrize("classifier", ["2fc", "mlm", "1fc"])
def test_vqa_module_mirror_vs_reference_fixture_and_oracle(classifier):
    """vqa ResNetVLBERT mirror (FastRCNN + VisualLinguisticBert mirrors + the HIP classifier / BCE node): "2fc" against the fixture
    produced by the reference's own VQA module, "mlm" (the shipped cfgs/vqa classifier) against the oracle; logits, loss, gradients
    of the classifier, the encoder and obj_downsample; reference-named checkpoint in and out; inference_forward."""
    from oracle import vqa_oracle as VQ
    from tests.test_oracle_golden import load_vqa_case
    M = pkg("vqa.modules.resnet_vlbert_for_vqa")
    z, cfg, params, batch = load_vqa_case()
    A, hidden = int(z["answer_vocab"]), int(z["classifier_hidden"])
    if classifier != "2fc":
        params = VQ.init_vqa_params(cfg, int(z["pseed"]), A, classifier)
    net = M.ResNetVLBERT(_vqa_config(cfg, classifier, A, hidden), device="cuda:0")
    assert set(net.state_dict()) == set(params), set(net.state_dict()) ^ set(params)
    net.load_state_dict({k: v for k, v in params.items()})
    net.train()
    for m in (net.vlbert, net.image_feature_extractor):          # deterministic comparison: every dropout off (as in the fixture)
        m.eval()
    net.cls_drop = 0.0
    boxes, im_info, question, label = [t.to(dev()) for t in batch]
    outputs, loss = net.train_forward(None, boxes, im_info, question, label)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out, oloss = VQ.vqa_forward(leaves, cfg, *batch, classifier=classifier, classifier_dropout=0.0, train=False)
    if classifier == "2fc":
        report("vqa logits vs reference fixture", outputs["label_logits"], torch.from_numpy(z["logits"]), 2e-2, 2e-2)
        assert abs(float(loss.detach()) - float(z["loss"])) < 1e-2 * float(z["loss"])
    report("vqa logits (%s) vs oracle" % classifier, outputs["label_logits"], out["label_logits"].detach(), 2e-2, 2e-2)
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-2 * float(oloss.detach())
    loss.backward()
    oloss.backward()
    got = dict(net.named_parameters())
    names = ["vlbert.encoder.layer.1.output.dense.weight", "image_feature_extractor.obj_downsample.1.weight", "vlbert.word_embeddings.weight",
             "object_linguistic_embeddings.weight"]
    names += {"2fc": ["final_mlp.1.weight", "final_mlp.4.weight", "final_mlp.4.bias"], "1fc": ["final_mlp.1.weight", "final_mlp.1.bias"],
              "mlm": ["final_mlp.0.dense.weight", "final_mlp.0.LayerNorm.weight", "final_mlp.2.weight", "final_mlp.2.bias"]}[classifier]
    for k in names:
        e = rel_fro(got[k].grad, leaves[k].grad)
        print("  vqa %s d %s rel-fro %.3e" % (classifier, k, e))
        # obj_downsample: ReLU sign flips of bf16-vs-fp32 pre-activations near 0 (see test_fast_rcnn_mirror_mask_embedding_gradient)
        assert e < (0.12 if "obj_downsample" in k else 5e-2), k
    net.eval()
    inf = net(None, boxes, im_info, question)
    report("vqa inference_forward logits", inf["label_logits"], out["label_logits"].detach(), 2e-2, 2e-2)
    # classifier dropout on: runs, finite, different from the deterministic loss
    net.train()
    net.cls_drop = 0.5
    _, l2 = net.train_forward(None, boxes, im_info, question, label)
    l2.backward()
    assert torch.isfinite(l2) and abs(float(l2) - float(loss)) > 0


def test_engine_degenerate_samples_vs_oracle():
    """Edge cases of the ragged layout: a sample with NO valid box (all rows padded), a sample with the shortest possible text
    ([CLS] x [SEP]) and a sample with a single labelled token; every loss / gradient against the oracle."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    params = O.init_params(cfg, seed=51)
    B, T, R = 4, 16, 6
    batch = list(syn.make_batch(B, T, R, seed=52, ragged=True))
    boxes, im_info, text, rel, mlm_labels, mvrc_ops, mvrc_labels = batch
    boxes[1] = -2.0                                    # sample 1: no objects at all
    mvrc_ops[1] = 0
    mvrc_labels[1] = 0
    text[2] = 0                                        # sample 2: [CLS] tok [SEP]
    text[2, 0], text[2, 1], text[2, 2] = 101, 103, 102
    mlm_labels[2] = -1
    mlm_labels[2, 1] = 2000
    mlm_labels[3] = -1                                 # sample 3: exactly one labelled token
    mlm_labels[3, 1] = int(text[3, 1]) if int(text[3, 1]) != 103 else 2001
    # 20 valid boxes in the whole batch: the obj_downsample gradients are dominated by bf16-vs-fp32 ReLU sign flips (the engine's own
    # pieces are mutually consistent, cf. test_fast_rcnn_mirror_mask_embedding_gradient) -> looser per-tensor tolerance; logits,
    # losses and the gradient norm keep the standard bars
    check_against_oracle("degenerate samples", cfg, params, tuple(batch), grad_tol=0.12)


def test_engine_no_valid_mvrc_rows_and_single_sample_vs_oracle():
    """soft_cross_entropy returns 0 when no row has a valid soft label (common/utils/misc.py:139-140) -- the device-side count must not
    divide by zero; and batch size 1."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=1)
    params = O.init_params(cfg, seed=61)
    batch = list(syn.make_batch(1, 20, 7, seed=62, ragged=True))
    batch[5].zero_()                       # no masked regions ...
    batch[6].zero_()                       # ... and no soft labels at all
    eng = check_against_oracle("B=1, no MVRC labels", cfg, params, tuple(batch), grad_tol=0.12)
    assert eng.loss_values()["mvrc_loss"] == 0.0
    assert float(eng.g32["vlbert.mvrc_head.region_cls_pred.weight"].abs().max()) == 0.0


def _per_layer_report(tag, eng, grads, norm, L):
    """rel-Frobenius gradient error aggregated per encoder layer index (depth growth of the bf16 error is visible in the log)."""
    rows = []
    for l in range(L):
        p = "vlbert.encoder.layer.%d." % l
        num = den = 0.0
        for name, g in eng.g32.items():
            if name.startswith(p):
                d = (g.detach().double().cpu() / eng.loss_scale - grads[name].double()).norm() ** 2
                num += float(d)
                den += float(grads[name].double().norm() ** 2)
        rows.append((l, (num / max(den, 1e-300)) ** 0.5))
    line = "%s per-layer rel-fro grad err: %s" % (tag, "  ".join("L%d %.2e" % r for r in rows))
    print(line)
    try:
        from tests.gpu_util import REPORT
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return rows


def test_engine_headline_c2_12_layers_vs_oracle():
    """BASELINE.json configs[1] EXACTLY as bench.py times it: VL-BERT-base, 12 layers, H = 768, 64 text + 36 regions (S = 101),
    V = 30522, C = 1601 -- ragged batch of 6, eval mode, against oracle.loss_and_grads.  Bars (north_star's bf16 bound): losses and
    the global gradient norm within 1e-2; logits within 1e-2 of the tensor scale in the MAX norm over all 11.7 M logits and within
    1e-2 in relative Frobenius norm (measured on MI355X: 8.2e-3 / 7.6e-3 with the fp16 + LayerNorm-residual stream; 1.4e-2 / 1.3e-2
    with a bf16 residual stream; rounding the WEIGHTS to bf16 alone -- everything else fp32, computed with the oracle -- costs
    5.0e-3 / 5.2e-3 at this depth, DESIGN.md "precision"); per-tensor rel-Frobenius gradient error bounded, per-layer error
    printed so the growth of the bf16 error with depth is visible."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=12)
    params = O.init_params(cfg, seed=71)
    batch = syn.make_batch(6, 64, 36, seed=72, ragged=True)
    eng = check_against_oracle("C2 12-layer", cfg, params, batch, grad_tol=6e-2, logit_rtol=1e-2, logit_fro_tol=1e-2)
    _, _, grads, norm = eng.oracle_result
    rows = _per_layer_report("C2 12-layer", eng, grads, norm, 12)
    assert max(e for _, e in rows) <= 4e-2, rows


def test_engine_headline_c2_full_length_batch_vs_oracle():
    """Same configuration with the full-length (non-ragged) batch layout of the bench workload (every sample 64 + 36)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=12)
    params = O.init_params(cfg, seed=73)
    batch = syn.make_batch(4, 64, 36, seed=74, ragged=False)
    check_against_oracle("C2 12-layer full-length", cfg, params, batch, grad_tol=6e-2, logit_rtol=1e-2, logit_fro_tol=1e-2)


def test_engine_large_4_layers_s229_vs_oracle():
    """VL-BERT-large (H = 1024, 16 heads, I = 4096) at 4 layers with the 128 text + 100 regions sequence (S = 229) of
    BASELINE.json configs 4-5, full vocabulary, ragged batch of 2."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=4)
    params = O.init_params(cfg, seed=75)
    batch = syn.make_batch(2, 128, 100, seed=76, ragged=True)
    eng = check_against_oracle("large 4-layer S=229", cfg, params, batch, grad_tol=0.12, logit_rtol=1e-2, logit_fro_tol=1e-2)
    _, _, grads, norm = eng.oracle_result
    rows = _per_layer_report("large 4-layer S=229", eng, grads, norm, 4)
    assert max(e for _, e in rows) <= 4e-2, rows


def test_engine_large_24_layers_s229_vs_oracle():
    """VL-BERT-large at its FULL depth (24 layers x 1024, 16 heads, FFN 4096, 128 text + 100 regions, S = 229, full vocabulary, ragged
    batch of 2) -- the shape of BASELINE.json configs 4-5 -- against the fp32 oracle.

    Bars: losses and the global gradient norm within 1e-2 in either build.  Logits: within 1e-2 (max norm and relative Frobenius) in
    the fp16 build -- the precision configs 4-5 actually name is fp32 / "mixed precision" = the reference's Apex fp16, i.e.
    VLB_PRECISION=f16 here (tests/test_f16_build_gpu.py runs this test on it: ~1.5e-3).  In the default bf16 build the SAME kernels
    reach 1.10e-2 / 1.01e-2 at this depth (measured on MI355X): that is the rounding of the GEMM operands themselves -- 10 bf16
    roundings per layer at 2^-9/sqrt(3) each, growing with sqrt(depth): 7.6e-3 at 12 x 768, x sqrt(2) here; rounding only the WEIGHTS
    to bf16 in the fp32 oracle already costs 8.9e-3 / 8.3e-3 (fp16: 1.0e-3) -- not of the residual stream (fp16 + fp32 LayerNorm re-materialisation) and not of
    any kernel, so no kernel change moves it; the bf16 bound asserted here (1.25e-2) documents that physics, it is not the
    north star's bar, which the headline bf16 configuration (12 layers) meets at 1e-2 in its own tests."""
    syn = pkg("synthetic")
    f16 = pkg("ops").BF16 == torch.float16
    cfg = O.VLBertConfig(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24)
    params = O.init_params(cfg, seed=77)
    batch = syn.make_batch(2, 128, 100, seed=78, ragged=True)
    bar = 1e-2 if f16 else 1.25e-2
    # per-tensor gradients: the worst tensors are the query / key projections of the top layers (their gradient passes the softmax
    # Jacobian, a difference of nearly equal terms): 0.14 in bf16 at this depth (0.08 at 4 layers), 0.02 in the fp16 build
    eng = check_against_oracle("large 24-layer S=229", cfg, params, batch, grad_tol=0.12 if f16 else 0.16, logit_rtol=bar, logit_fro_tol=bar)
    _, _, grads, norm = eng.oracle_result
    rows = _per_layer_report("large 24-layer S=229", eng, grads, norm, 24)
    assert max(e for _, e in rows) <= 5e-2, rows


@pytest.fixture
def gemm_options():
    """Run-time GEMM kernel selection (vlb_gemm_set_option), restored to the library defaults afterwards."""
    lib = pkg("_lib")
    yield lib
    lib.gemm_set_option("p8_min_tiles", 160)
    lib.gemm_set_option("p8_mode", 1)
    lib.gemm_set_option("nt_ring", 1)


@pytest.mark.parametrize("tile", [3, 4, 5])
def test_engine_headline_c2_12_layers_through_large_tile_kernels(gemm_options, tile):
    """The headline configuration (12 layers, 64 + 36, V = 30522) with every eligible GEMM FORCED onto the kernels bench.py times
    at batch 256 -- `gemm_nt_p8_kernel<FMH, EPI>` with 192- / 256- / 320-row tiles (all fused epilogues incl. the LayerNorm-residual
    fp16 stream) and the grouped large-tile weight gradient `gemm_tn8_kernel` (the engine's row-padded operands reach it at any
    batch) -- against the fp32 oracle, same 1e-2 bars as the default-selection test.  At B = 6 the launcher's own choice would be
    the 128x128 kernels, so without this test the benched kernels were only compared per-op and engine-vs-engine at 2 layers."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=12)
    params = O.init_params(cfg, seed=71)
    batch = syn.make_batch(6, 64, 36, seed=72, ragged=True)
    cache = test_engine_headline_c2_12_layers_through_large_tile_kernels.__dict__.setdefault("oracle", {})
    if "r" not in cache:
        cache["r"] = O.loss_and_grads(params, cfg, batch, train=False)
    gemm_options.gemm_set_option("p8_min_tiles", 1)
    gemm_options.gemm_set_option("p8_mode", tile)
    eng = check_against_oracle("C2 12-layer via p8<%d>" % tile, cfg, params, batch, grad_tol=6e-2, logit_rtol=1e-2, logit_fro_tol=1e-2,
                               oracle=cache["r"])
    rows = _per_layer_report("C2 12-layer via p8<%d>" % tile, eng, cache["r"][2], cache["r"][3], 12)
    assert max(e for _, e in rows) <= 4e-2, rows


def test_engine_c2_per_gpu_batch_32_step_vs_oracle():
    """One rank of the 8-GPU strong-scaling run: 12 layers, 32 full-length samples (M = 3232 rows) with the launcher's OWN kernel
    selection at that size -- the 192-row large tile for the N = 3072 GEMMs, `gemm_nt_ring_kernel` (EPI -1 / 4 / 0) for the N = 768
    ones, the grouped weight gradient at R = 3328 -- against the oracle (1e-2 bars)."""
    syn = pkg("synthetic")
    cfg = O.VLBertConfig(num_hidden_layers=12)
    params = O.init_params(cfg, seed=79)
    batch = syn.make_batch(32, 64, 36, seed=80, ragged=False)
    check_against_oracle("C2 12-layer B=32", cfg, params, batch, grad_tol=6e-2, logit_rtol=1e-2, logit_fro_tol=1e-2)


def test_engine_headline_c2_batch_256_as_benched_vs_oracle():
    """The headline configuration AT THE BATCH bench.py TIMES: 12 layers, 256 full-length samples of 64 text + 36 regions (M = 25856
    rows), the engine built as bench.py builds it (MLM head on the labelled rows only, no logits copy), the library's OWN kernel
    selection at that size -- 243-1212-tile launches of `gemm_nt_p8_kernel` with mid-stream wave-private drains, the grouped
    `gemm_tn8_kernel` weight gradients with K = 25856 -- one forward + backward against the fp32 oracle (eval mode: the oracle cannot
    regenerate the counter-RNG masks; the dropout instantiations at this shape are compared per op in
    tests/test_ops_gpu.py::test_gemm_headline_shapes_with_the_launchers_own_selection).  Bars: north_star's bf16 bound 1e-2 on the
    losses, the global gradient norm and the encoder output; per-tensor / per-layer gradient errors as in the batch-6 test.
    (The oracle's forward + backward of 256 samples takes ~1-2 min on the box's host cores.)"""
    syn = pkg("synthetic")
    E = pkg("engine")
    cfg = O.VLBertConfig(num_hidden_layers=12)
    params = O.init_params(cfg, seed=81)
    B, T, R = 256, 64, 36
    batch = syn.make_batch(B, T, R, seed=82, ragged=False)
    eng = E.PretrainEngine(E.ModelConfig(num_hidden_layers=12), B, T, R, device="cuda:0", train=False)
    assert eng.mlm_cap is not None, "bench.py's engine runs the MLM head on the labelled rows"
    eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
    eng.set_batch(*[t.to(dev()) for t in batch])
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    torch.cuda.synchronize()
    lv = eng.loss_values()
    outputs, loss, grads, norm = O.loss_and_grads(params, cfg, batch, train=False)
    tag = "C2 12-layer B=256"
    S = eng.S
    seq = outputs["sequence_output"].detach()
    got = eng.X[-1].view(B, S, -1)[:, :seq.shape[1]]
    fro = rel_fro(got, seq)
    print("%s encoder output relative Frobenius error %.3e" % (tag, fro))
    assert fro <= 1e-2, fro
    report(tag + " encoder output", got, seq, 2e-3, 1.5e-2)
    for k in ("mlm_loss", "mvrc_loss"):
        ref = float(outputs[k])
        print("%s %s: hip %.6f oracle %.6f" % (tag, k, lv[k], ref))
        assert abs(lv[k] - ref) <= 1e-2 * max(1.0, abs(ref)), (k, lv[k], ref)
    gn = eng.grad_norm()
    print("%s grad_norm: hip %.6f oracle %.6f rel %.3e" % (tag, gn, norm, abs(gn - norm) / norm))
    assert abs(gn - norm) <= 1e-2 * norm
    worst = sorted(((rel_fro(g, grads[name]), name) for name, g in eng.grads().items() if float(grads[name].norm()) >= 1e-6 * norm),
                   reverse=True)
    for e, n in worst[:8]:
        print("   rel-fro grad err %.3e  %s" % (e, n))
    assert worst[0][0] <= 6e-2, worst[:5]
    rows = _per_layer_report(tag, eng, grads, norm, 12)
    assert max(e for _, e in rows) <= 4e-2, rows


def test_mlm_head_compaction_matches_full_path():
    """MLM head on the labelled rows only (engine default) vs the full head (keep_logits engine): identical losses, gradients equal
    up to fp32 summation order; multitask layout (caption + text-only groups with their own means); capacity overflow is loud."""
    syn = pkg("synthetic")
    E = pkg("engine")
    cfg = O.VLBertConfig(num_hidden_layers=2)
    cfg.multitask = True
    params = O.init_params(cfg, seed=95)
    B, T, R, Ba = 6, 64, 12, 4
    batch = syn.make_batch(B, T, R, seed=96, ragged=True)
    aux = syn.make_aux_text(Ba, T, seed=97)
    res = []
    for keep in (True, False):
        mc = E.ModelConfig(num_hidden_layers=2, multitask=True)
        eng = E.PretrainEngine(mc, B, T, R, device="cuda:0", keep_logits=keep, train=False, B_aux=Ba)
        assert (eng.mlm_cap is None) == keep
        eng.load_state_dict({k: v.to(dev()) for k, v in params.items()})
        eng.set_batch(*[t.to(dev()) for t in batch], aux_text=aux[0].to(dev()), aux_mlm_labels=aux[1].to(dev()))
        eng.zero_grad()
        eng.forward(train=False)
        eng.backward(train=False)
        torch.cuda.synchronize()
        res.append((eng.loss_values(), eng.grad_norm(), {k: v.cpu() for k, v in eng.grads().items()}, eng))
    full, comp = res
    for k in ("mlm_loss_wvc", "mlm_loss_aux", "mvrc_loss", "loss"):
        print("compaction %s: full %.7f compact %.7f" % (k, full[0][k], comp[0][k]))
        assert abs(full[0][k] - comp[0][k]) <= 1e-5 * max(1.0, abs(full[0][k])), k
    assert abs(full[1] - comp[1]) <= 1e-5 * full[1]
    # per tensor: |difference| against max(|tensor|, 1e-3 of the global norm) -- gradients that are ~0 by construction (the key bias:
    # softmax is invariant to it) consist of fp32 summation-order noise in BOTH paths and carry no signal
    worst = max((float((comp[2][n].double() - g.double()).norm()) / max(float(g.double().norm()), 1e-3 * full[1]), n) for n, g in full[2].items())
    print("compaction: worst per-tensor gradient difference %.3e (%s)" % worst)
    # not bit-identical: the split-K partition of the decoder dgrad / wgrad depends on the row count, so fp32 sums are taken in a
    # different order and a few bf16 roundings of d(hidden) flip by one ulp (2^-8) -- the difference stays an order of magnitude
    # below the engine-vs-fp32-oracle error (1e-2 class)
    assert worst[0] < 5e-3, worst
    n_lab = int((batch[4] >= 0).sum()) + int((aux[1] >= 0).sum())
    assert int(comp[3].counts[0] + comp[3].counts[2]) == n_lab and n_lab < comp[3].mlm_cap
    # oracle agreement of the compact engine (multitask forward)
    bt = tuple(batch) + tuple(aux)
    outputs, loss, grads, norm = O.loss_and_grads(params, cfg, bt, train=False)
    assert abs(comp[0]["loss"] - float(loss)) <= 1e-2 * float(loss) and abs(comp[1] - norm) <= 1e-2 * norm
    # capacity overflow: every position labelled.  Device-resident labels -> the flag trips loss_values(); host labels -> full path.
    eng = comp[3]
    lab = batch[4].clone()
    lab[:] = 2000
    b2 = list(batch)
    b2[4] = lab
    eng.set_batch(*[t.to(dev()) for t in b2], aux_text=aux[0].to(dev()), aux_mlm_labels=aux[1].to(dev()))
    eng.zero_grad()
    eng.forward(train=False)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="mlm_cap"):
        eng.loss_values()
    eng.mlm_overflow.zero_()
    eng.set_batch(*[t.to(dev()) if i != 4 else t for i, t in enumerate(b2)], aux_text=aux[0].to(dev()), aux_mlm_labels=aux[1])
    assert not eng._mlm_compact_now
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    torch.cuda.synchronize()
    o2, l2, g2, n2 = O.loss_and_grads(params, cfg, tuple(b2) + tuple(aux), train=False)
    assert abs(eng.loss_values()["loss"] - float(l2)) <= 1e-2 * float(l2)
