"""VCR fine-tuning wrapper (BASELINE config 5) on the GPU (-m gpu): the `ResNetVLBERT` mirror of
vcr/modules/resnet_vlbert_for_vcr.py against the fixture produced by the reference's own module and against oracle/vcr_oracle.py, and
the fused SGD-momentum step (vcr/function/train.py:124-128) against torch.optim.SGD semantics."""
import os

import numpy as np
import pytest
import torch

from oracle import vlbert_oracle as O
from tests.gpu_util import dev, pkg, report

pytestmark = pytest.mark.gpu


def rel_fro(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def _vcr_config(cfg, num_layers, pos_w, classifier="1fc", sigmoid=True):
    class A(dict):
        __getattr__ = dict.__getitem__
    return A(NETWORK=A(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_SEMANTIC=False, IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True,
                       IMAGE_NUM_LAYERS=num_layers, OUTPUT_CONV5=False, IMAGE_FROZEN_BN=True, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2],
                       IMAGE_FINAL_DIM=cfg.hidden_size, BLIND=False, NO_GROUNDING=False, NO_OBJ_ATTENTION=False, ANSWER_FIRST=False,
                       QA_ONE_SENT=False, FOR_MASK_VL_MODELING_PRETRAIN=False, ENABLE_CNN_REG_LOSS=True, CNN_LOSS_TOP=True,
                       CNN_REG_DROPOUT=0.0, CNN_LOSS_WEIGHT=1.0, ANS_LOSS_WEIGHT=1.0, CLASSIFIER_TYPE=classifier,
                       CLASSIFIER_HIDDEN_SIZE=64, CLASSIFIER_DROPOUT=0.1, CLASSIFIER_SIGMOID=sigmoid,
                       CLASSIFIER_SIGMOID_LOSS_POSITIVE_WEIGHT=pos_w,
                       VLBERT=A(hidden_size=cfg.hidden_size, visual_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                                num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                                vocab_size=cfg.vocab_size, max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=3,
                                visual_region_classes=cfg.visual_region_classes, visual_ln=True, with_pooler=True,
                                hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, initializer_range=0.02,
                                visual_scale_text_init=0.0, visual_scale_object_init=0.0, object_word_embed_mode=2)))


def test_vcr_module_mirror_vs_reference_fixture_and_oracle():
    """logits [B,4], answer loss, regulariser loss and total loss against the REFERENCE fixture; gradients of the classifier, the
    regulariser head, the encoder, the pooler, obj_downsample and the trainable convolutions against the oracle (which reproduces the
    reference's gradient norms exactly, tests/test_oracle_golden.py); reference-named checkpoint in and out (`vlbert._module.*`,
    `image_feature_extractor.head.0.*` aliases); inference_forward; a train-mode step with every dropout on."""
    from oracle import vcr_oracle as VC
    from oracle import vision_oracle as VO
    from tests.test_oracle_golden import load_vcr_case
    M = pkg("vcr.modules.resnet_vlbert_for_vcr")
    z, cfg, params, P, batch = load_vcr_case()
    nl, pos_w = int(z["num_layers"]), float(z["positive_weight"])
    net = M.ResNetVLBERT(_vcr_config(cfg, nl, pos_w), device="cuda:0")
    sd = {("vlbert._module." + k[len("vlbert."):]) if k.startswith("vlbert.") else k: v for k, v in params.items()}
    sd.update({"image_feature_extractor." + k: v for k, v in VO.split_state_dict(P).items()})
    own = net.state_dict()
    extra = {k for k in own if ".head.0." in k}
    assert extra and set(own) - extra == set(sd), (set(own) - extra) ^ set(sd)
    net.load_state_dict(sd)
    back = net.state_dict()
    for k in ("image_feature_extractor.backbone.layer3.0.conv2.weight", "vlbert._module.pooler.dense.weight", "cnn_loss_reg.2.weight"):
        assert torch.equal(back[k].cpu(), sd[k]), k                     # reference layout out again ([O,I,KH,KW] convolutions)
    net.train()
    for m in (net.vlbert, net.image_feature_extractor):                 # deterministic comparison: every dropout off (as in the fixture)
        m.eval()
    net.cls_drop = 0.0
    gb = {k: v.to(dev()) for k, v in batch.items()}
    args = (gb["image"], gb["boxes"], gb["masks"], gb["question"], None, gb["answer_choices"], None, gb["answer_label"], gb["im_info"])
    outputs, loss = net.train_forward(*args)
    report("vcr logits vs REFERENCE fixture", outputs["label_logits"], torch.from_numpy(z["logits"]), 3e-2, 3e-2)
    for name, got, ref in (("ans_loss", outputs["ans_loss"], z["ans_loss"]), ("cnn_regularization_loss", outputs["cnn_regularization_loss"],
                                                                              z["cnn_reg_loss"]), ("loss", loss, z["loss"])):
        print("vcr %s: hip %.5f reference %.5f" % (name, float(got.detach()), float(ref)))
        assert abs(float(got.detach()) - float(ref)) < 2e-2 * max(1.0, abs(float(ref))), name
    loss.backward()
    torch.cuda.synchronize()
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    out, oloss = VC.vcr_forward(leaves, cfg, vision_params=Po, image_num_layers=nl, classifier="1fc", classifier_dropout=0.0, sigmoid=True,
                                positive_weight=pos_w, cnn_reg_top=True, train=False, **batch)
    oloss.backward()
    got = dict(net.named_parameters())
    for k in ("final_mlp.1.weight", "final_mlp.1.bias", "cnn_loss_reg.0.dense.weight", "cnn_loss_reg.2.weight", "cnn_loss_reg.2.bias",
              "vlbert._module.pooler.dense.weight", "vlbert._module.encoder.layer.1.output.dense.weight",
              "vlbert._module.word_embeddings.weight", "object_linguistic_embeddings.weight",
              "image_feature_extractor.obj_downsample.1.weight"):
        ref = leaves[k.replace("vlbert._module.", "vlbert.")].grad
        e = rel_fro(got[k].grad, ref)
        print("  vcr d %s rel-fro %.3e" % (k, e))
        assert e < (0.12 if "obj_downsample" in k else 6e-2), k
    names = dict(zip(VO.split_state_dict(P).keys(), P.keys()))
    for short in ("roi_head_feature_extractor.2.conv3.weight", "backbone.layer3.5.conv2.weight"):
        g = got["image_feature_extractor." + short].grad.permute(0, 3, 1, 2)          # mirror stores [O,KH,KW,I]
        e = rel_fro(g, Po[names[short]].grad)
        print("  vcr d %s rel-fro %.3e" % (short, e))
        assert e < 0.15, short
    net.eval()
    inf = net(*args[:7], gb["im_info"])
    report("vcr inference_forward logits", inf["label_logits"], out["label_logits"].detach(), 3e-2, 3e-2)
    # every dropout on (hidden / attention / obj_downsample / classifier / regulariser): runs, finite, a different loss
    net.train()
    net.cls_drop, net.reg_drop = 0.5, 0.3
    net.zero_grad()
    _, l2 = net.train_forward(*args)
    l2.backward()
    assert torch.isfinite(l2) and abs(float(l2) - float(loss)) > 0
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


def test_vcr_softmax_loss_and_2fc_classifier_vs_oracle():
    """CLASSIFIER_SIGMOID false (softmax CE over the 4 choices, :358) with the "2fc" classifier (config default) against the oracle."""
    from oracle import vcr_oracle as VC
    from oracle import vision_oracle as VO
    from tests.test_oracle_golden import load_vcr_case
    M = pkg("vcr.modules.resnet_vlbert_for_vcr")
    z, cfg, _, P, batch = load_vcr_case()
    nl = int(z["num_layers"])
    params = VC.init_vcr_params(cfg, 77, classifier="2fc", hidden=64, embed_mode=2, cnn_reg_top=True)
    net = M.ResNetVLBERT(_vcr_config(cfg, nl, 1.0, classifier="2fc", sigmoid=False), device="cuda:0")
    sd = {("vlbert._module." + k[len("vlbert."):]) if k.startswith("vlbert.") else k: v for k, v in params.items()}
    sd.update({"image_feature_extractor." + k: v for k, v in VO.split_state_dict(P).items()})
    net.load_state_dict(sd)
    net.train()
    for m in (net.vlbert, net.image_feature_extractor):
        m.eval()
    net.cls_drop = 0.0
    gb = {k: v.to(dev()) for k, v in batch.items()}
    outputs, loss = net.train_forward(gb["image"], gb["boxes"], gb["masks"], gb["question"], None, gb["answer_choices"], None,
                                      gb["answer_label"], gb["im_info"])
    out, oloss = VC.vcr_forward(params, cfg, vision_params=P, image_num_layers=nl, classifier="2fc", classifier_dropout=0.0, sigmoid=False,
                                cnn_reg_top=True, train=False, **batch)
    report("vcr 2fc logits vs oracle", outputs["label_logits"], out["label_logits"].detach(), 3e-2, 3e-2)
    assert abs(float(loss.detach()) - float(oloss.detach())) < 2e-2 * max(1.0, abs(float(oloss.detach())))
    assert "positive_fraction" not in outputs


@pytest.mark.parametrize("n", [1, 1000, 4096 * 3 + 5])
def test_sgd_momentum_kernel_matches_torch_sgd(n):
    """vlb_sgd_momentum_step against torch.optim.SGD(lr, momentum, weight_decay) on the host, three steps, with and without the
    device-side gradient-norm clip; the bf16 working copy is the rounded parameter."""
    ops = pkg("ops")
    g0 = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g0)
    grads = [torch.randn(n, generator=g0) * (3.0 if i == 1 else 0.3) for i in range(3)]
    lr, mom, wd, max_norm = 0.05, 0.9, 1e-2, 1.0
    for clip in (False, True):
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.SGD([ref], lr=lr, momentum=mom, weight_decay=wd)
        p, buf = p0.clone().to(dev()), torch.zeros(n, device=dev())
        p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev())
        for g in grads:
            ref.grad = g.clone()
            if clip:
                torch.nn.utils.clip_grad_norm_([ref], max_norm)
            opt.step()
            gg = g.to(dev())
            ss = torch.zeros(1, device=dev())
            ops.sumsq(gg, ss)
            ops.sgd_momentum_step(p, gg, buf, lr, mom, wd, p16=p16, sumsq=ss if clip else None, max_norm=max_norm if clip else 0.0)
        torch.cuda.synchronize()
        err = float((p.cpu() - ref.detach()).abs().max())
        print("sgd n %d clip %s: max |p - torch| %.3e" % (n, clip, err))
        assert err < 2e-6 * max(1.0, float(ref.detach().abs().max()))
        assert torch.equal(p16.float().cpu(), p.cpu().to(torch.bfloat16).float())


@pytest.mark.parametrize("kind", ["sgd", "adamw"])
def test_fused_clip_grad_norm_matches_torch_clip_then_step(kind):
    """optim.clip_grad_norm_(params, max_norm, optimizer) -- the trainer's clip (common/trainer.py:139-145) fused into the NEXT step's
    kernel (device-side norm, coefficient applied as the gradients are read) -- against torch.nn.utils.clip_grad_norm_ followed by the
    torch statement of the same optimizer, two steps (one clipping, one not), several flat runs; also with an fp16-style loss scale
    folded out through grad_scale.  The gradients themselves must stay untouched."""
    OPT = pkg("optim")
    g0 = torch.Generator().manual_seed(11)
    shapes = [(300, 64), (64,), (1000,), (17, 5)]
    for scale in (1.0, 128.0):
        flat = torch.zeros(sum(int(np.prod(sh)) for sh in shapes) + 64, device=dev())
        ps, off = [], 0
        for i, sh in enumerate(shapes):
            n = int(np.prod(sh))
            if i == 2:
                off += 64                                    # a gap: two flat runs
            q = torch.nn.Parameter(flat[off:off + n].view(sh))
            q.data.copy_(torch.randn(sh, generator=g0))
            ps.append(q)
            off += n
        gflat = torch.zeros_like(flat)
        off = 0
        for i, (q, sh) in enumerate(zip(ps, shapes)):
            n = int(np.prod(sh))
            if i == 2:
                off += 64
            q.grad = gflat[off:off + n].view(sh)
            off += n
        ref = [torch.nn.Parameter(q.detach().cpu().clone()) for q in ps]
        if kind == "sgd":
            opt = OPT.FusedSGD(ps, lr=0.05, momentum=0.9, weight_decay=1e-2)
            ropt = torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=1e-2)
        else:
            opt = OPT.FusedAdamW(ps, lr=1e-2, eps=1e-6, weight_decay=1e-2)
            m = [torch.zeros_like(r) for r in ref]
            v = [torch.zeros_like(r) for r in ref]
        max_norm = 1.0
        for step, mag in ((1, 3.0), (2, 0.01)):               # step 1 clips (norm ~ 100), step 2 does not
            grads = [torch.randn(sh, generator=g0) * mag for sh in shapes]
            for q, g in zip(ps, grads):
                q.grad.copy_(g * scale)
            kept = [q.grad.clone() for q in ps]
            total = OPT.clip_grad_norm_(ps, max_norm, opt, grad_scale=1.0 / scale)
            for r, g in zip(ref, grads):
                r.grad = g.clone()
            tnorm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            assert abs(float(total) - float(tnorm)) <= 1e-5 * float(tnorm)
            opt.step()
            assert all(torch.equal(q.grad, k) for q, k in zip(ps, kept))           # fused: nothing rescaled in place
            if kind == "sgd":
                ropt.step()
            else:
                for r, mm, vv in zip(ref, m, v):
                    O.adamw_step(r.data, r.grad, mm, vv, step, 1e-2, eps=1e-6, weight_decay=1e-2)
            torch.cuda.synchronize()
            err = max(float((q.detach().cpu() - r.detach()).abs().max()) for q, r in zip(ps, ref))
            print("fused clip + %s, loss scale %g, step %d: norm %.4f, max |p - torch| %.3e" % (kind, scale, step, float(total), err))
            assert err < 5e-6
        opt.step()                                             # a step without a clip call: no stale coefficient
        assert opt._clip is None


def test_fused_sgd_optimizer_on_a_module_mirror():
    """FusedSGD (the VCR trainer's optimiser) on the VisualLinguisticBert mirror's parameters, consecutive slices of one flat buffer:
    whole runs updated per launch, identical to torch.optim.SGD on a copy; the mirror picks the new weights up (its bf16 copies are
    keyed on the parameters' version counters)."""
    VL = pkg("common.visual_linguistic_bert")
    OPT = pkg("optim")
    cfg = O.VLBertConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                         max_position_embeddings=64, with_pooler=True)
    conf = _vcr_config(cfg, 50, 1.0)["NETWORK"]["VLBERT"]
    core = VL.VisualLinguisticBert(conf, device="cuda:0")
    g = torch.Generator().manual_seed(5)
    B, T, R, H = 2, 8, 3, cfg.hidden_size
    ids = torch.randint(5, 300, (B, T), generator=g).to(dev())
    tv = torch.randn(B, T, H, generator=g).to(dev()).requires_grad_(True)
    ovl = torch.randn(B, R, 2 * H, generator=g).to(dev()).requires_grad_(True)
    tmask = torch.ones(B, T, dtype=torch.bool, device=dev())
    omask = torch.ones(B, R, dtype=torch.bool, device=dev())
    core.eval()

    def run():
        t, o, pooled = core(ids, torch.zeros_like(ids), tv, tmask, ovl, omask, output_all_encoded_layers=False,
                            output_text_and_object_separately=True)
        return t.float().sum() * 1e-2 + pooled.float().sum()
    core.zero_grad()
    l0 = run()
    l0.backward()
    torch.cuda.synchronize()
    params = [p for p in core.parameters() if p.grad is not None]
    before = [p.detach().clone() for p in params]
    gcopy = [p.grad.detach().clone() for p in params]
    opt = OPT.FusedSGD(core.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    opt.step()
    opt.step()
    torch.cuda.synchronize()
    runs = opt._runs[0]
    print("FusedSGD: %d parameters in %d launches" % (len(params), len(runs)))
    assert len(runs) <= len(params) // 3          # (alignment gaps of the flat layout split the range into a few runs)
    worst = 0.0
    for p, b, gr in zip(params, before, gcopy):
        r = torch.nn.Parameter(b.clone())
        o2 = torch.optim.SGD([r], lr=0.1, momentum=0.9, weight_decay=1e-3)
        for _ in range(2):
            r.grad = gr.clone()
            o2.step()
        worst = max(worst, float((p.detach() - r.detach()).abs().max()))
    assert worst < 1e-5, worst
    l1 = run()
    assert abs(float(l1) - float(l0)) > 1e-4            # the next forward ran on the updated weights
    assert "momentum_buffer" in opt.state[params[0]] and opt.state_dict()["param_groups"][0]["momentum"] == 0.9


def _reference_adamw_step(p, g, st, lr, betas, eps, wd):
    """common/nlp/bert/optimization.py:150-185 restated on fp32 tensors (test-local, double-free: the same operation order)."""
    st["step"] += 1
    st["m"].mul_(betas[0]).add_(g, alpha=1.0 - betas[0])
    st["v"].mul_(betas[1]).addcmul_(g, g, value=1.0 - betas[1])
    denom = st["v"].sqrt().add_(eps)
    step_size = lr * (1.0 - betas[1] ** st["step"]) ** 0.5 / (1.0 - betas[0] ** st["step"])
    p.addcdiv_(st["m"], denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


def test_fused_adamw_optimizer_matches_the_reference_adamw():
    """FusedAdamW (pre-training / VQA optimiser, common/nlp/bert/optimization.py:107-187) on the VisualLinguisticBert mirror's flat
    parameter runs and on separately allocated tensors: three steps with a learning-rate change in between (an LR scheduler writing
    param_groups[...]['lr']) against the reference's update rule restated in torch; state keys as the reference's."""
    VL = pkg("common.visual_linguistic_bert")
    OPT = pkg("optim")
    cfg = O.VLBertConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                         max_position_embeddings=64, with_pooler=True)
    core = VL.VisualLinguisticBert(_vcr_config(cfg, 50, 1.0)["NETWORK"]["VLBERT"], device="cuda:0")
    g = torch.Generator().manual_seed(9)
    core._prepare_grads()
    extra = [torch.nn.Parameter(torch.randn(33, 7, generator=g).to(dev())), torch.nn.Parameter(torch.randn(5, generator=g).to(dev()))]
    params = list(core.parameters()) + extra
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    betas, eps, wd = (0.9, 0.999), 1e-6, 1e-2
    opt = OPT.FusedAdamW([{"params": list(core.parameters())}, {"params": extra, "weight_decay": 0.0}], lr=1e-2, betas=betas, eps=eps,
                         weight_decay=wd)
    ref = [(p.detach().clone(), {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}) for p in params]
    for it, lr in enumerate((1e-2, 1e-2, 3e-3)):
        for group in opt.param_groups:
            group["lr"] = lr
        grads = []
        for p in params:
            gr = torch.randn(p.shape, generator=g).to(dev()) * 0.1
            p.grad.copy_(gr)
            grads.append(gr)
        opt.step()
        for (rp, st), gr, p in zip(ref, grads, params):
            _reference_adamw_step(rp, gr, st, lr, betas, eps, wd if not any(p is e for e in extra) else 0.0)
    torch.cuda.synchronize()
    worst = max(float((p.detach() - rp).abs().max()) for p, (rp, _) in zip(params, ref))
    worst_m = max(float((opt.state[p]["exp_avg"] - st["m"]).abs().max()) for p, (_, st) in zip(params, ref))
    worst_v = max(float((opt.state[p]["exp_avg_sq"] - st["v"]).abs().max()) for p, (_, st) in zip(params, ref))
    print("FusedAdamW: %d parameters in %d + %d launches; max |dp| %.2e |dm| %.2e |dv| %.2e"
          % (len(params), len(opt._runs[0]), len(opt._runs[1]), worst, worst_m, worst_v))
    assert worst < 1e-5 and worst_m < 1e-7 and worst_v < 1e-8, (worst, worst_m, worst_v)
    assert len(opt._runs[0]) <= len(list(core.parameters())) // 3 and len(opt._runs[1]) == 2
    assert all(opt.state[p]["step"] == 3 for p in params)
    assert set(opt.state[extra[0]]) >= {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("flag", ["BLIND", "NO_GROUNDING", "NO_OBJ_ATTENTION", "ANSWER_FIRST", "QA_ONE_SENT"])
def test_vcr_ablation_switches_run_and_change_what_they_should(flag):
    """The ablation switches of the reference's VCR forward (vcr/modules/resnet_vlbert_for_vcr.py:253-330) through the mirror: each is
    input plumbing in front of the same encoder (token layouts pinned on CPU against the reference's functions,
    tests/test_host_logic_cpu.py).  Here: the step runs (finite loss, gradients), the logits DIFFER from the default configuration, and
    the visual switches act as specified -- BLIND is independent of the image and of the boxes' classes, NO_OBJ_ATTENTION / BLIND hand
    the encoder an empty object mask."""
    from tests.test_oracle_golden import load_vcr_case
    from oracle import vision_oracle as VO
    M = pkg("vcr.modules.resnet_vlbert_for_vcr")
    z, cfg, params, P, batch = load_vcr_case()
    nl, pos_w = int(z["num_layers"]), float(z["positive_weight"])
    sd = {("vlbert._module." + k[len("vlbert."):]) if k.startswith("vlbert.") else k: v for k, v in params.items()}
    sd.update({"image_feature_extractor." + k: v for k, v in VO.split_state_dict(P).items()})
    gb = {k: v.to(dev()) for k, v in batch.items()}
    args = lambda b: (b["image"], b["boxes"], b["masks"], b["question"], None, b["answer_choices"], None, b["answer_label"], b["im_info"])

    def build(**kw):
        c = _vcr_config(cfg, nl, pos_w)
        c["NETWORK"].update(kw)
        if kw.get("BLIND"):
            c["NETWORK"]["ENABLE_CNN_REG_LOSS"] = False
        net = M.ResNetVLBERT(c, device="cuda:0")
        net.load_state_dict(sd, strict=False)
        net.train()
        for m in (net.vlbert, net.image_feature_extractor):
            m.eval()
        net.cls_drop = 0.0
        return net
    base = build()
    ref_logits = base.train_forward(*args(gb))[0]["label_logits"].detach().clone()
    net = build(**{flag: True})
    outputs, loss = net.train_forward(*args(gb))
    loss.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(outputs["label_logits"]).all())
    gsum = sum(float(p.grad.abs().sum()) for p in net.parameters() if p.grad is not None)
    assert gsum > 0 and gsum == gsum
    diff = float((outputs["label_logits"].detach() - ref_logits).abs().max())
    print("vcr %s: loss %.4f, max |logits - default| %.3e" % (flag, float(loss), diff))
    assert diff > 1e-4
    if flag == "BLIND":                       # no pixel and no detector class reaches the logits
        b2 = dict(gb)
        b2["image"] = gb["image"] * 0.0 + 7.0
        boxes = gb["boxes"].clone()
        boxes[:, :, -1] = (boxes[:, :, -1] + 3).clamp(max=80) * (boxes[:, :, -1] >= 0) + boxes[:, :, -1] * (boxes[:, :, -1] < 0)
        b2["boxes"] = boxes
        again = net.train_forward(*args(b2))[0]["label_logits"].detach()
        assert torch.equal(again, outputs["label_logits"].detach())


def test_finetune_entry_points_run_reference_style_configs():
    """`python -m vl-bert_amd.vqa.train_end2end` / `vcr.train_end2end` (the reference's vqa/train_end2end.py, vcr/train_end2end.py command
    lines) on reference-style YAMLs at test size: 3 optimizer steps of 2 accumulated micro-batches each through the module mirrors --
    VQA: FusedAdamW + triangle schedule + clip 1.0 in bf16; VCR: FusedSGD + warm-up multi-step schedule + clip 10 with TRAIN.FP16 ->
    the fp16 build and its static loss scale (run in a child process: one build of the library per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = os.path.join(root, "tests", "fixtures")
    for task, extra, lr3 in (("vqa", [], 1.0e-5 * 2 * 2 * (2.0 / 3.0)), ("vcr", ["--compute", "cfg"], 7.0e-5 * 2 * 2 * 0.5)):
        cmd = [sys.executable, "-c", "import importlib,sys; sys.path.insert(0, %r); m = importlib.import_module('vl-bert_amd.%s.train_end2end'); "
               "net, opt, loss = m.main(%r); import math; assert math.isfinite(loss), loss; print('LR %%.9e LOSS %%.5f' %% (opt.param_groups[0]['lr'], loss))"
               % (root, task, ["--cfg", os.path.join(fx, task + "_small.yaml"), "--steps", "3"] + extra)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        print(r.stdout[-1500:])
        print(r.stderr[-1500:])
        assert r.returncode == 0, r.stderr[-1500:]
        line = next(l for l in r.stdout.splitlines() if l.startswith("LR "))
        assert abs(float(line.split()[1]) - lr3) < 1e-6 * lr3, (line, lr3)      # the schedule's value at the LAST step run (k = 2)
        assert "step 3" in r.stdout
