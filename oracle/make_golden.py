"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference) on CPU.  Run in the build container only:

    python oracle/make_golden.py

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header).  The fixtures pin
the oracle restatement: the reference ships no golden vectors of its own
(SURVEY.md §4), so "the reference code executed on CPU" is the ground truth.

What is run: pretrain.modules.ResNetVLBERTForPretraining (precomputed-feature
configuration), eval() mode so dropout is off (SURVEY.md §8c pitfall ii), with every
parameter randomised by oracle.init_params(randomize_all=True) and loaded through
the reference's own load_state_dict, on seeded ragged batches from
vl-bert_amd/synthetic.py.  Stored: the batch, parameter checksums, logits, losses,
the encoder output, per-parameter gradient digests (norm, sum, strided sample),
the global grad norm, and parameters after 3 steps of the reference's AdamW
(common/nlp/bert/optimization.py:107-187).
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.vlbert_oracle import VLBertConfig, init_params  # noqa: E402

synthetic = importlib.import_module("vl-bert_amd.synthetic")

SAMPLE = 4096

CASES = {
    # ragged lengths, 2 heads, MLM + MVRC (the north-star loss set)
    "ragged_small": dict(
        cfg=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                 vocab_size=512, max_position_embeddings=64, visual_region_classes=50),
        B=3, T=12, R=5, ragged=True, seed=11, pseed=3),
    # full lengths, pooler + relationship head on, 3 layers
    "full_rel": dict(
        cfg=dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=384,
                 vocab_size=512, max_position_embeddings=64, visual_region_classes=40,
                 with_pooler=True, with_rel_loss=True),
        B=2, T=8, R=4, ragged=False, seed=5, pseed=7),
    # multitask wrapper: 2 auxiliary text-only samples appended (aux text longer than the caption length)
    "multitask_small": dict(
        cfg=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                 vocab_size=512, max_position_embeddings=64, visual_region_classes=50, multitask=True),
        B=3, T=10, R=5, ragged=True, seed=31, pseed=13, aux=(2, 14)),
    # single head (H=64), longer ragged sequences
    "ragged_1head": dict(
        cfg=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128,
                 vocab_size=300, max_position_embeddings=96, visual_region_classes=20),
        B=4, T=40, R=17, ragged=True, seed=23, pseed=9),
}


def digest(t):
    t = t.detach().double().reshape(-1)
    stride = max(1, t.numel() // SAMPLE)
    return np.array([t.norm().item(), t.sum().item()]), t[::stride][:SAMPLE].float().numpy()


def run_case(name, spec):
    RefModel, RefAdamW = ref_import.import_reference()
    cfg = VLBertConfig(**spec["cfg"])
    if cfg.multitask:
        RefModel = ref_import.import_reference_multitask()
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_%s" % name),
                                          cfg.vocab_size)
    torch.manual_seed(0)
    model = RefModel(ref_import.make_reference_config(cfg, vocab_dir))
    params = init_params(cfg, seed=spec["pseed"])
    sd = dict(params)
    sd["vlbert.mlm_head.predictions.decoder.weight"] = sd["vlbert.word_embeddings.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    model.eval()

    batch = synthetic.make_batch(spec["B"], spec["T"], spec["R"], vocab_size=cfg.vocab_size,
                                 region_classes=cfg.visual_region_classes, seed=spec["seed"],
                                 ragged=spec["ragged"])
    boxes, im_info, text, rel, mlm_labels, mvrc_ops, mvrc_labels = [t.clone() for t in batch]
    aux = ()
    if cfg.multitask:
        aux = synthetic.make_aux_text(spec["aux"][0], spec["aux"][1], vocab_size=cfg.vocab_size, seed=spec["seed"])
    outputs, loss = model(None, boxes, im_info, text, rel, mlm_labels, mvrc_ops, mvrc_labels, *[t.clone() for t in aux])
    model.zero_grad()
    loss.backward()

    out = {"cfg_keys": np.array(list(spec["cfg"].keys())),
           "cfg_vals": np.array([float(v) for v in spec["cfg"].values()]),
           "B": spec["B"], "T": spec["T"], "R": spec["R"], "ragged": spec["ragged"],
           "seed": spec["seed"], "pseed": spec["pseed"]}
    for k, t in zip(("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels"),
                    batch):
        out["in_" + k] = t.numpy()
    if cfg.multitask:
        out["in_aux_text"], out["in_aux_mlm_labels"] = aux[0].numpy(), aux[1].numpy()
        out["aux_shape"] = np.array(spec["aux"])
        out["mlm_logits_wvc"] = outputs["mlm_logits_wvc"].detach().numpy()
        out["mlm_logits_aux"] = outputs["mlm_logits_aux"].detach().numpy()
        out["mvrc_logits"] = outputs["mvrc_logits"].detach().numpy()
        for k in ("mlm_loss_wvc", "mlm_loss_aux", "mvrc_loss"):
            out[k] = float(outputs[k])
    else:
        out["mlm_logits"] = outputs["mlm_logits"].detach().numpy()
        out["mvrc_logits"] = outputs["mvrc_logits"].detach().numpy()
        if outputs["relationship_logits"] is not None:
            out["relationship_logits"] = outputs["relationship_logits"].detach().numpy()
        for k in ("relationship_loss", "mlm_loss", "mvrc_loss"):
            out[k] = float(outputs[k])
    out["loss"] = float(loss)

    named = dict(model.named_parameters())   # tied decoder.weight is deduplicated by named_parameters
    total = 0.0
    names = []
    for n in sorted(params.keys()):
        p = named[n]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        total += float((g.double() ** 2).sum())
        st, smp = digest(g)
        out["g_stat/" + n], out["g_smp/" + n] = st, smp
        st, smp = digest(params[n])
        out["p_stat/" + n] = st
        names.append(n)
    out["names"] = np.array(names)
    out["grad_norm"] = total ** 0.5

    # three steps of the reference AdamW on the (fixed) gradients above
    opt = RefAdamW([named[n] for n in names], lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-2,
                   correct_bias=True)
    for _ in range(3):
        opt.step()
    for n in names:
        out["adamw_smp/" + n] = digest(named[n])[1]

    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: loss %.6f grad_norm %.6f -> %s (%.1f KB)" % (name, out["loss"], out["grad_norm"], path, os.path.getsize(path) / 1024))


def run_core_case(name="core_small"):
    """Module-API fixture: the reference's common.visual_linguistic_bert.VisualLinguisticBertForPretraining driven with
    random per-token text-visual embeddings and [visual || linguistic] object embeddings (ragged masks); scalar
    objective = <mlm_logits, Wm> + <mvrc_logits, Wv> with fixed random weights, so d(logits) is known to the test."""
    ref_import.import_reference()
    from common.visual_linguistic_bert import VisualLinguisticBertForPretraining as RefCore
    cfg = VLBertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                       vocab_size=512, max_position_embeddings=64, visual_region_classes=50)
    B, T, R, H = 3, 11, 6, cfg.hidden_size
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_%s" % name), cfg.vocab_size)
    vcfg = ref_import.make_reference_config(cfg, vocab_dir).NETWORK.VLBERT
    torch.manual_seed(0)
    model = RefCore(vcfg, language_pretrained_model_path=None, with_rel_head=False, with_mlm_head=True, with_mvrc_head=True)
    params = init_params(cfg, seed=21)
    sd = {k[len("vlbert."):]: v for k, v in params.items() if k.startswith("vlbert.")}
    sd["mlm_head.predictions.decoder.weight"] = sd["word_embeddings.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model.eval()
    g = torch.Generator().manual_seed(77)
    tlen = torch.tensor([T, 7, 9])
    nobj = torch.tensor([4, R, 5])
    text_mask = torch.arange(T).unsqueeze(0) < tlen.unsqueeze(1)
    obj_mask = torch.arange(R).unsqueeze(0) < nobj.unsqueeze(1)
    text_ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g) * text_mask
    text_type = torch.randint(0, 2, (B, T), generator=g) * text_mask
    # inputs are rounded to bf16-representable values: the HIP path takes them as bf16
    text_vis = torch.randn((B, T, H), generator=g).bfloat16().float().requires_grad_(True)
    obj_vl = torch.randn((B, R, 2 * H), generator=g).bfloat16().float().requires_grad_(True)
    wm = torch.randn((B, T, cfg.vocab_size), generator=g) * 0.05 * text_mask.unsqueeze(-1)
    wv = torch.randn((B, R, cfg.visual_region_classes), generator=g) * 0.05 * obj_mask.unsqueeze(-1)
    _, mlm_logits, mvrc_logits = model(text_ids, text_type, text_vis, text_mask, obj_vl, obj_mask)
    obj = (mlm_logits * wm).sum() + (mvrc_logits * wv).sum()
    model.zero_grad()
    obj.backward()
    out = {"B": B, "T": T, "R": R, "pseed": 21,
           "cfg_keys": np.array(["hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "vocab_size",
                                 "max_position_embeddings", "visual_region_classes"]),
           "cfg_vals": np.array([128.0, 2, 2, 256, 512, 64, 50]),
           "in_text_ids": text_ids.numpy(), "in_text_type": text_type.numpy(), "in_text_vis": text_vis.detach().numpy(),
           "in_text_mask": text_mask.numpy(), "in_obj_vl": obj_vl.detach().numpy(), "in_obj_mask": obj_mask.numpy(),
           "w_mlm": wm.numpy(), "w_mvrc": wv.numpy(),
           "mlm_logits": mlm_logits.detach().numpy(), "mvrc_logits": mvrc_logits.detach().numpy(), "objective": float(obj),
           "d_text_vis": text_vis.grad.numpy(), "d_obj_vl": obj_vl.grad.numpy()}
    named = dict(model.named_parameters())
    total, names = 0.0, []
    for n in sorted(named):
        gr = named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])
        total += float((gr.double() ** 2).sum())
        out["g_stat/" + n], out["g_smp/" + n] = digest(gr)
        names.append(n)
    out["names"] = np.array(names)
    out["grad_norm"] = total ** 0.5
    path = os.path.join(ROOT, "tests", "golden", "core", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)
    print("%s: objective %.6f grad_norm %.6f -> %s (%.1f KB)" % (name, out["objective"], out["grad_norm"], path, os.path.getsize(path) / 1024))


def run_vision_case(name="vision_small", num_layers=50):
    """e2e fixture: the reference's own FastRCNN module (common/fast_rcnn.py, IMAGE_FEAT_PRECOMPUTED false: ResNet trunk ->
    ROIAlign -> dilated layer4 head -> avg-pool) on a small image batch with one padded box, parameters from
    oracle/vision_oracle.init_vision_params (handed to the reference through its own `torch.load(pretrained_model_path)` call).
    Objective <obj_reps_raw, Wr>; stores the pooled features, body4 statistics and per-parameter gradient norms."""
    from oracle import vision_oracle as VO
    ref_import.import_reference()
    ref_import.install_roi_align_oracle()
    import common.lib.roi_pooling as rp
    rp.C_ROIPooling = sys.modules["common.lib.roi_pooling.C_ROIPooling"]
    import common.lib.roi_pooling.roi_align as ra_mod
    ra_mod.C_ROIPooling = rp.C_ROIPooling
    from common.fast_rcnn import FastRCNN as RefFastRCNN
    seed = 7
    P = VO.init_vision_params(seed, num_layers)
    E = ref_import._EasyDict
    cfg = E(dict(NETWORK=dict(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_SEMANTIC=False, IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True,
                              IMAGE_NUM_LAYERS=num_layers, IMAGE_PRETRAINED="oracle_init", IMAGE_PRETRAINED_EPOCH=0,
                              OUTPUT_CONV5=False, IMAGE_FROZEN_BN=True, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2])))
    real_load = torch.load
    torch.load = lambda path, *a, **k: dict(P) if str(path).startswith("oracle_init") else real_load(path, *a, **k)
    try:
        model = RefFastRCNN(cfg, average_pool=True, final_dim=768, enable_cnn_reg_loss=False)
        model.init_weight()
    finally:
        torch.load = real_load
    model.train()
    model.bn_eval()                      # common/fast_rcnn.py:122-126, called by the trainer every iteration
    model.obj_downsample[0].p = 0.0      # the objective below does not involve obj_downsample; keep the run deterministic
    g = torch.Generator().manual_seed(seed + 1)
    B, R, Hi, Wi = 2, 3, 96, 128
    img = torch.randn(B, 3, Hi, Wi, generator=g) * 50.0
    boxes = torch.tensor([[[4.0, 6.0, 90.0, 80.0], [30.5, 10.25, 120.0, 60.0], [0.0, 0.0, 127.0, 95.0]],
                          [[10.0, 20.0, 50.0, 70.0], [64.0, 8.0, 100.0, 40.0], [-2.0, -2.0, -2.0, -2.0]]])
    im_info = torch.tensor([[Wi, Hi, 1.0, 1.0, 0.0], [Wi, Hi, 1.0, 1.0, 1.0]])
    box_mask = boxes[:, :, 0] > -1.5
    out = model(images=img, boxes=boxes.clone(), box_mask=box_mask, im_info=im_info, classes=None, segms=None, mvrc_ops=None,
                mask_visual_embed=None)
    raw = out["obj_reps_raw"]                                    # [B, R, 2048], zero rows for padded boxes
    Wr = torch.randn(raw.shape, generator=g) / raw.numel() ** 0.5
    (raw * Wr).sum().backward()
    names = VO.split_state_dict(P)
    ref_params = dict(model.named_parameters())
    frozen = {("roi_head_feature_extractor." + k[7:]) if k.startswith("layer4.") else ("backbone." + k) for k in VO.frozen_names(P)}
    gnorm, gsample = {}, {}
    for k in names:
        if k not in ref_params:
            continue                                              # BN running statistics are buffers
        p = ref_params[k]
        assert (p.grad is None or float(p.grad.abs().sum()) == 0.0) == (k in frozen), k
        if k not in frozen:
            gnorm[k] = float(p.grad.double().norm())
            gsample[k] = p.grad.reshape(-1)[:: max(1, p.grad.numel() // 64)][:64].numpy().copy()
    # the restatement, same inputs
    Po = {k: v.clone().requires_grad_(k not in VO.frozen_names(P)) for k, v in P.items()}
    feats, body4 = VO.e2e_features(img, boxes, Po, num_layers)
    err = float((feats - raw[box_mask]).abs().max())
    print("%s: restatement vs reference |d feats|max = %.3e (|feats| max %.3f)" % (name, err, float(raw.abs().max())))
    assert err < 1e-4
    # the VCR call form: per-pixel object masks multiplied into the RoI-head output before the pool (common/fast_rcnn.py:152-156)
    segms = (torch.rand(B, R, 14, 14, generator=g) < 0.6).float()
    with torch.no_grad():
        out_s = model(images=img, boxes=boxes.clone(), box_mask=box_mask, im_info=im_info, classes=None, segms=segms, mvrc_ops=None,
                      mask_visual_embed=None)
        feats_s, _ = VO.e2e_features(img, boxes, Po, num_layers, segms=segms)
    raw_s = out_s["obj_reps_raw"]
    err_s = float((feats_s - raw_s[box_mask]).abs().max())
    print("%s: segms variant restatement vs reference |d feats|max = %.3e" % (name, err_s))
    assert err_s < 1e-4
    path = os.path.join(ROOT, "tests", "golden", "vision", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    keys = sorted(gnorm)
    np.savez_compressed(path, seed=seed, num_layers=num_layers, img=img.numpy(), boxes=boxes.numpy(), im_info=im_info.numpy(),
                        Wr=Wr.numpy(), obj_reps_raw=raw.detach().numpy(), segms=segms.numpy(), obj_reps_raw_segms=raw_s.numpy(),
                        body4_mean=float(body4.mean()),
                        body4_abs_mean=float(body4.abs().mean()), body4_sample=body4.detach().reshape(-1)[::97][:512].numpy(),
                        grad_names=np.array(keys), grad_norms=np.array([gnorm[k] for k in keys]),
                        grad_samples=np.stack([gsample[k] if gsample[k].size == 64 else np.resize(gsample[k], 64) for k in keys]))
    print("%s -> %s (%.1f KB), %d trainable tensors" % (name, path, os.path.getsize(path) / 1024, len(keys)))


def run_vqa_case(name="vqa_small"):
    """VQA fixture: the reference's own vqa.modules.resnet_vlbert_for_vqa.ResNetVLBERT (precomputed features, "2fc" classifier) in
    eval-free train_forward with every dropout at p = 0, parameters from oracle/vqa_oracle.init_vqa_params; stores inputs, logits,
    loss and gradient digests; checks the restatement against it."""
    from oracle import vqa_oracle as VQ
    ref_import.import_reference()
    from vqa.modules.resnet_vlbert_for_vqa import ResNetVLBERT as RefVQA
    cfg = VLBertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                       vocab_size=512, max_position_embeddings=64, visual_region_classes=50, hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0, obj_downsample_dropout=0.0)
    A, hidden = 37, 96
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_%s" % name), cfg.vocab_size)
    rc = ref_import.make_reference_config(cfg, vocab_dir)
    E = ref_import._EasyDict
    rc.NETWORK.update(dict(BLIND=False, NO_GROUNDING=False, ENABLE_CNN_REG_LOSS=False, CLASSIFIER_TYPE="2fc", CLASSIFIER_DROPOUT=0.0,
                           CLASSIFIER_HIDDEN_SIZE=hidden, CLASSIFIER_PRETRAINED=False))
    for k, v in dict(BLIND=False, NO_GROUNDING=False, ENABLE_CNN_REG_LOSS=False, CLASSIFIER_TYPE="2fc", CLASSIFIER_DROPOUT=0.0,
                     CLASSIFIER_HIDDEN_SIZE=hidden).items():
        setattr(rc.NETWORK, k, v)
    rc.DATASET = E(dict(ANSWER_VOCAB_SIZE=A))
    torch.manual_seed(0)
    model = RefVQA(rc)
    pseed = 9
    params = VQ.init_vqa_params(cfg, pseed, A, "2fc", hidden)
    sd = model.state_dict()
    missing = [k for k in sd if k not in params and not k.endswith("num_batches_tracked")]
    assert not missing, missing
    model.load_state_dict({k: params[k] for k in sd}, strict=True)
    model.train()
    model.image_feature_extractor.obj_downsample[0].p = 0.0
    B, Lq, R = 3, 9, 6
    syn = importlib.import_module("vl-bert_amd.synthetic")
    batch = syn.make_batch(B, 8, R, vocab_size=cfg.vocab_size, region_classes=cfg.visual_region_classes, seed=13, ragged=True)
    boxes = batch[0]
    # im_info as the VQA dataset emits it: FOUR columns (w, h, 1, 1) (vqa/data/datasets/vqa.py:217; pre-training / VCR rows have a fifth),
    # and a different size per image, so that a consumer reading it with the wrong row stride cannot match this fixture
    im_info = torch.tensor([[640.0, 480.0, 1.0, 1.0], [600.0, 600.0, 1.0, 1.0], [500.0, 375.0, 1.0, 1.0]])
    g = torch.Generator().manual_seed(14)
    question = torch.randint(200, cfg.vocab_size, (B, Lq), generator=g)
    qlen = torch.tensor([Lq, 5, 7])
    question[torch.arange(Lq)[None, :] >= qlen[:, None]] = 0
    label = (torch.rand(B, A, generator=g) < 0.1).float() * torch.rand(B, A, generator=g)       # soft VQA scores
    outputs, loss = model(None, boxes.clone(), im_info, question, label)
    loss.backward()
    ref_grads = {k: v.grad.detach() for k, v in model.named_parameters() if v.grad is not None}
    out2, loss2 = VQ.vqa_forward({k: v.clone().requires_grad_(True) for k, v in params.items()}, cfg, boxes, im_info, question, label,
                                 classifier="2fc", classifier_dropout=0.0, train=False)
    err = float((out2["label_logits"] - outputs["label_logits"]).abs().max())
    print("%s: restatement vs reference |d logits|max %.3e, loss %.6f vs %.6f" % (name, err, float(loss2), float(loss)))
    assert err < 1e-4 and abs(float(loss2) - float(loss)) < 1e-5
    keys = sorted(ref_grads)
    path = os.path.join(ROOT, "tests", "golden", "vqa", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, pseed=pseed, answer_vocab=A, classifier_hidden=hidden, boxes=boxes.numpy(), im_info=im_info.numpy(),
                        question=question.numpy(), label=label.numpy(), logits=outputs["label_logits"].detach().numpy(),
                        loss=float(loss), grad_names=np.array(keys),
                        grad_norms=np.array([float(ref_grads[k].double().norm()) for k in keys]),
                        grad_total_norm=float(torch.sqrt(sum((g_.double() ** 2).sum() for g_ in ref_grads.values()))))
    print("%s -> %s (%.1f KB), %d gradient tensors" % (name, path, os.path.getsize(path) / 1024, len(keys)))


def run_vcr_case(name="vcr_small", num_layers=50):
    """VCR fixture: the reference's own vcr.modules.resnet_vlbert_for_vcr.ResNetVLBERT in the configuration of the shipped
    cfgs/vcr/*.yaml (images through the FastRCNN image branch with object masks, 4 answer choices folded by TimeDistributed, pooler,
    "1fc" classifier, sigmoid BCE, ENABLE_CNN_REG_LOSS + CNN_LOSS_TOP), every dropout at p = 0, parameters from
    oracle/vcr_oracle.init_vcr_params + oracle/vision_oracle.init_vision_params; stores inputs, logits, losses and gradient norms;
    checks the restatement (oracle/vcr_oracle.py) against it."""
    from oracle import vcr_oracle as VC
    from oracle import vision_oracle as VO
    ref_import.import_reference()
    ref_import.install_roi_align_oracle()
    import common.lib.roi_pooling as rp
    rp.C_ROIPooling = sys.modules["common.lib.roi_pooling.C_ROIPooling"]
    import common.lib.roi_pooling.roi_align as ra_mod
    ra_mod.C_ROIPooling = rp.C_ROIPooling
    from vcr.modules.resnet_vlbert_for_vcr import ResNetVLBERT as RefVCR
    cfg = VLBertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                       vocab_size=512, max_position_embeddings=64, visual_region_classes=50, hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0, obj_downsample_dropout=0.0, with_pooler=True)
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_%s" % name), cfg.vocab_size)
    rc = ref_import.make_reference_config(cfg, vocab_dir)
    pos_w = 2.0
    for k, v in dict(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_SEMANTIC=False, IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True,
                     IMAGE_NUM_LAYERS=num_layers, IMAGE_PRETRAINED="oracle_init", IMAGE_PRETRAINED_EPOCH=0, OUTPUT_CONV5=False,
                     IMAGE_FROZEN_BN=True, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2], BLIND=False, NO_GROUNDING=False, NO_OBJ_ATTENTION=False,
                     ANSWER_FIRST=False, QA_ONE_SENT=False, FOR_MASK_VL_MODELING_PRETRAIN=False, ENABLE_CNN_REG_LOSS=True,
                     CNN_LOSS_TOP=True, CNN_REG_DROPOUT=0.0, CNN_LOSS_WEIGHT=1.0, ANS_LOSS_WEIGHT=1.0, CLASSIFIER_TYPE="1fc",
                     CLASSIFIER_HIDDEN_SIZE=64, CLASSIFIER_DROPOUT=0.0, CLASSIFIER_SIGMOID=True,
                     CLASSIFIER_SIGMOID_LOSS_POSITIVE_WEIGHT=pos_w, REPLACE_OBJECT_CHANGE_LABEL=False).items():
        setattr(rc.NETWORK, k, v)
        rc.NETWORK[k] = v
    vseed, pseed = 17, 19
    P = VO.init_vision_params(vseed, num_layers)
    real_load = torch.load
    torch.load = lambda path, *a, **k: dict(P) if str(path).startswith("oracle_init") else real_load(path, *a, **k)
    try:
        torch.manual_seed(0)
        model = RefVCR(rc)
    finally:
        torch.load = real_load
    params = VC.init_vcr_params(cfg, pseed, classifier="1fc", embed_mode=2, cnn_reg_top=True)
    ref_sd = {}
    for k, v in params.items():
        ref_sd[("vlbert._module." + k[len("vlbert."):]) if k.startswith("vlbert.") else k] = v
    for k, v in VO.split_state_dict(P).items():
        ref_sd["image_feature_extractor." + k] = v
    sd = model.state_dict()
    # (`image_feature_extractor.head.0.*` are aliases: `head` = Sequential(roi_head_feature_extractor, pool, flatten), fast_rcnn.py:80-84)
    missing = [k for k in sd if k not in ref_sd and not k.endswith("num_batches_tracked") and ".head.0." not in k]
    assert not missing, missing
    model.load_state_dict({k: ref_sd[k] for k in sd if k in ref_sd}, strict=False)
    model.train()                                   # (ResNetVLBERT.train() puts the BatchNorms in eval mode, :100-104)
    model.image_feature_extractor.obj_downsample[0].p = 0.0
    g = torch.Generator().manual_seed(23)
    B, C, R, Hi, Wi, Lq, La = 2, 4, 3, 96, 128, 6, 5
    img = torch.randn(B, 3, Hi, Wi, generator=g) * 50.0
    boxes = torch.tensor([[[0.0, 0.0, 127.0, 95.0, 0.0], [30.5, 10.25, 120.0, 60.0, 17.0], [4.0, 6.0, 90.0, 80.0, 80.0]],
                          [[0.0, 0.0, 127.0, 95.0, 0.0], [64.0, 8.0, 100.0, 40.0, 3.0], [-1.0, -1.0, -1.0, -1.0, -1.0]]])
    masks = (torch.rand(B, R, 14, 14, generator=g) < 0.7).float()
    im_info = torch.tensor([[Wi, Hi, 1.0, 1.0, 0.0], [Wi, Hi, 1.0, 1.0, 1.0]])
    nvalid = torch.tensor([3, 2])
    qlen = torch.tensor([Lq, 4])
    question = torch.zeros((B, Lq, 2), dtype=torch.int64)
    question[:, :, 0] = torch.randint(200, cfg.vocab_size, (B, Lq), generator=g)
    question[:, :, 1] = torch.randint(-1, 2, (B, Lq), generator=g)                  # -1: no object (clamped to 0 = the image box)
    question[torch.arange(Lq)[None, :] >= qlen[:, None]] = 0
    answers = torch.zeros((B, C, La, 2), dtype=torch.int64)
    answers[..., 0] = torch.randint(200, cfg.vocab_size, (B, C, La), generator=g)
    answers[..., 1] = torch.randint(-1, 2, (B, C, La), generator=g)
    alen = torch.randint(2, La + 1, (B, C), generator=g)
    alen[0, 1] = La
    answers[torch.arange(La)[None, None, :] >= alen[:, :, None]] = 0
    answers[..., 1] = torch.minimum(answers[..., 1], (nvalid - 1)[:, None, None])
    question[:, :, 1] = torch.minimum(question[:, :, 1], (nvalid - 1)[:, None])
    label = torch.tensor([2, 0])
    outputs, loss = model(img, boxes.clone(), masks, question, None, answers, None, label, im_info)
    loss.backward()
    ref_grads = {}
    for k, v in model.named_parameters():
        if v.grad is not None and float(v.grad.abs().sum()) > 0:
            ref_grads[k.replace("vlbert._module.", "vlbert.")] = v.grad.detach()
    frozen = VO.frozen_names(P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Po = {k: v.clone().requires_grad_(k not in frozen) for k, v in P.items()}
    out2, loss2 = VC.vcr_forward(leaves, cfg, img, boxes, masks, question, answers, label, im_info, Po, num_layers, classifier="1fc",
                                 classifier_dropout=0.0, sigmoid=True, positive_weight=pos_w, cnn_reg_top=True, train=False)
    loss2.backward()
    err = float((out2["label_logits"] - outputs["label_logits"]).abs().max())
    print("%s: restatement vs reference |d logits|max %.3e, loss %.6f vs %.6f, cnn reg %.6f vs %.6f" %
          (name, err, float(loss2), float(loss), float(out2["cnn_regularization_loss"]), float(outputs["cnn_regularization_loss"])))
    assert err < 1e-4 and abs(float(loss2) - float(loss)) < 1e-5
    # per tensor, against max(|tensor|, 1e-6 of the global norm): tensors whose gradient is zero by construction (the key bias: softmax
    # is invariant to it) hold rounding noise on both sides -- the reference side now runs the reference's COMPILED ROIAlign forward,
    # whose last-bit differences from the restatement re-draw that noise
    total = float(torch.sqrt(sum((g_.double() ** 2).sum() for g_ in ref_grads.values())))
    worst = (0.0, "")
    for k, gref in ref_grads.items():
        if k in leaves:
            worst = max(worst, (float((leaves[k].grad - gref).norm() / max(float(gref.norm()), 1e-6 * total)), k))
    print("%s: restatement gradients, worst rel-fro vs reference %.3e (%s)" % (name, worst[0], worst[1]))
    assert worst[0] < 1e-3
    keys = sorted(ref_grads)
    path = os.path.join(ROOT, "tests", "golden", "vcr", name + ".npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, pseed=pseed, vseed=vseed, num_layers=num_layers, positive_weight=pos_w, img=img.numpy(), boxes=boxes.numpy(),
                        masks=masks.numpy(), im_info=im_info.numpy(), question=question.numpy(), answers=answers.numpy(),
                        label=label.numpy(), logits=outputs["label_logits"].detach().numpy(), loss=float(loss),
                        ans_loss=float(outputs["ans_loss"]), cnn_reg_loss=float(outputs["cnn_regularization_loss"]),
                        grad_names=np.array(keys), grad_norms=np.array([float(ref_grads[k].double().norm()) for k in keys]),
                        grad_total_norm=float(torch.sqrt(sum((g_.double() ** 2).sum() for g_ in ref_grads.values()))))
    print("%s -> %s (%.1f KB), %d gradient tensors" % (name, path, os.path.getsize(path) / 1024, len(keys)))


def run_c2_case(name="c2_headline"):
    """The reference at the BENCHED dimensions (BASELINE.json configs[1]: VL-BERT-base, 12 layers, H = 768, 12 heads, vocabulary
    30522, 1601 region classes, 64 text + 36 regions), batch 2, in DIGEST form so that the fixture stays under 1 MB: the inputs are
    regenerated from the seed by vl-bert_amd/synthetic.py (their digests are stored), logits as (norm, sum) + a 4096-sample stride,
    every parameter gradient as (norm, sum) + a 256-sample stride, losses, global gradient norm.  Pins the oracle -- and, through
    tests/test_engine_gpu.py, the HIP engine -- to /root/reference/pretrain/modules/resnet_vlbert_for_pretraining.py:93-216 run at the size
    the metric is quoted on, not only at the toy shapes above."""
    import time
    RefModel, RefAdamW = ref_import.import_reference()
    spec = dict(cfg=dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
                         max_position_embeddings=512, visual_region_classes=1601),
                B=2, T=64, R=36, ragged=True, seed=2026, pseed=17)
    cfg = VLBertConfig(**spec["cfg"])
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_%s" % name), cfg.vocab_size)
    torch.manual_seed(0)
    model = RefModel(ref_import.make_reference_config(cfg, vocab_dir))
    params = init_params(cfg, seed=spec["pseed"])
    sd = dict(params)
    sd["vlbert.mlm_head.predictions.decoder.weight"] = sd["vlbert.word_embeddings.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model.eval()
    batch = synthetic.make_batch(spec["B"], spec["T"], spec["R"], vocab_size=cfg.vocab_size, region_classes=cfg.visual_region_classes,
                                 seed=spec["seed"], ragged=spec["ragged"])
    t0 = time.time()
    outputs, loss = model(None, *[t.clone() for t in batch])
    model.zero_grad()
    loss.backward()
    dt = time.time() - t0
    out = {"cfg_keys": np.array(list(spec["cfg"].keys())), "cfg_vals": np.array([float(v) for v in spec["cfg"].values()]),
           "B": spec["B"], "T": spec["T"], "R": spec["R"], "ragged": spec["ragged"], "seed": spec["seed"], "pseed": spec["pseed"],
           "reference_fwd_bwd_seconds": dt, "reference_threads": torch.get_num_threads()}
    for k, t in zip(("boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels"), batch):
        out["in_stat/" + k] = digest(t)[0]
    for k in ("mlm_logits", "mvrc_logits"):
        out[k + "_shape"] = np.array(outputs[k].shape)
        out[k + "_stat"], out[k + "_smp"] = digest(outputs[k])
    for k in ("mlm_loss", "mvrc_loss"):
        out[k] = float(outputs[k])
    out["loss"] = float(loss)
    named = dict(model.named_parameters())
    total, names = 0.0, []
    global SAMPLE
    keep = SAMPLE
    SAMPLE = 256
    for n in sorted(params.keys()):
        g = named[n].grad if named[n].grad is not None else torch.zeros_like(named[n])
        total += float((g.double() ** 2).sum())
        out["g_stat/" + n], out["g_smp/" + n] = digest(g)
        out["p_stat/" + n] = digest(params[n])[0]
        names.append(n)
    SAMPLE = keep
    out["names"] = np.array(names)
    out["grad_norm"] = total ** 0.5
    os.makedirs(os.path.join(ROOT, "tests", "golden", "c2"), exist_ok=True)
    path = os.path.join(ROOT, "tests", "golden", "c2", name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: loss %.6f grad_norm %.6f, reference forward + backward %.1f s on %d threads -> %s (%.1f KB)"
          % (name, out["loss"], out["grad_norm"], dt, torch.get_num_threads(), path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "core":
        run_core_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "vision":
        run_vision_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "vqa":
        run_vqa_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "vcr":
        run_vcr_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "c2":
        run_c2_case()
    else:
        for name, spec in CASES.items():
            run_case(name, spec)
        run_core_case()
