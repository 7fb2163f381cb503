"""Host-side logic that needs no GPU: the mirrors' shape buckets and bounded caches, the optimisers' flat-run detection, and that the
product modules refuse to run without the HIP library / a GPU (no CPU fallback)."""
import importlib
from collections import OrderedDict

import pytest
import torch


def pkg(name):
    return importlib.import_module("vl-bert_amd." + name)


def test_shape_buckets_round_up_and_respect_the_sequence_limit(monkeypatch):
    VL = pkg("common.visual_linguistic_bert")
    monkeypatch.delenv("VLB_MIRROR_BUCKETS", raising=False)
    assert VL.shape_buckets() == (8, 4, 8)
    assert VL.bucketed(11, 6) == (16, 8)
    assert VL.bucketed(64, 36) == (64, 36)                  # the pre-training shape is already on the grid
    assert VL.bucketed(128, 100) == (128, 100)              # VQA large
    assert VL.bucketed(200, 55) == (200, 55)                # VCR: 200 + 56 + 1 would exceed the 256-position kernels -> exact shape
    monkeypatch.setenv("VLB_MIRROR_BUCKETS", "1,1,3")
    assert VL.shape_buckets() == (1, 1, 3) and VL.bucketed(11, 6) == (11, 6)
    monkeypatch.setenv("VLB_MIRROR_BUCKETS", "16,8,2")
    assert VL.bucketed(17, 9) == (32, 16)


def test_lru_cache_keeps_the_most_recently_used(monkeypatch):
    VL = pkg("common.visual_linguistic_bert")
    monkeypatch.setenv("VLB_MIRROR_BUCKETS", "8,4,3")
    cache, made = OrderedDict(), []

    def get(k):
        return VL.lru_get(cache, k, lambda: made.append(k) or ("engine", k))
    for k in (1, 2, 3):
        get(k)
    assert get(1) == ("engine", 1) and made == [1, 2, 3]    # hit: nothing rebuilt, 1 becomes the newest
    get(4)                                                  # evicts 2, the least recently used
    assert list(cache) == [3, 1, 4]
    get(2)
    assert made == [1, 2, 3, 4, 2] and list(cache) == [1, 4, 2]


def test_flat_runs_group_parameters_that_are_consecutive_in_memory():
    OPT = pkg("optim")
    flat_p, flat_g = torch.zeros(100), torch.zeros(100)
    views = [(0, 10), (10, 30), (30, 34), (40, 60), (60, 100)]      # a 6-element alignment gap after the third tensor
    params = []
    for a, b in views:
        p = torch.nn.Parameter(flat_p[a:b])
        p.grad = flat_g[a:b]
        params.append(p)
    lone = torch.nn.Parameter(torch.zeros(7))
    lone.grad = torch.zeros(7)
    params.append(lone)
    # (first, last + 1, elements spanned): the zero padding is part of the run
    assert OPT._flat_runs(params) == [(0, 5, 100), (5, 6, 7)]
    # a gap that holds somebody else's data is not padding: the run stops there (parameter side and gradient side alike)
    flat_p[36] = 1.0
    assert OPT._flat_runs(params) == [(0, 3, 34), (3, 5, 60), (5, 6, 7)]
    flat_p[36] = 0.0
    flat_g[35] = 1.0
    assert OPT._flat_runs(params) == [(0, 3, 34), (3, 5, 60), (5, 6, 7)]
    flat_g[35] = 0.0
    # a gradient that is NOT consecutive splits the run even when the parameters are
    params[1].grad = torch.zeros(20)
    assert OPT._flat_runs(params) == [(0, 1, 10), (1, 2, 20), (2, 5, 70), (5, 6, 7)]


def test_fused_optimisers_refuse_cpu_tensors_and_unsupported_modes():
    OPT = pkg("optim")
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        OPT.FusedAdamW([p], lr=1e-3).step()
    with pytest.raises(RuntimeError, match="no CPU path"):
        OPT.FusedSGD([p], lr=1e-3, momentum=0.9).step()
    with pytest.raises(NotImplementedError):
        OPT.FusedAdamW([p], correct_bias=False)
    with pytest.raises(NotImplementedError):
        OPT.FusedSGD([p], lr=1e-3, nesterov=True, momentum=0.9)
    with pytest.raises(ValueError):
        OPT.FusedAdamW([p], betas=(1.0, 0.999))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_module_mirrors_fail_loudly_without_a_gpu():
    VL = pkg("common.visual_linguistic_bert")
    conf = dict(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                max_position_embeddings=64, type_vocab_size=3, visual_ln=True, with_pooler=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VL.VisualLinguisticBert(conf)
    ops = pkg("ops")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.cast_f32_bf16(torch.zeros(8), torch.zeros(8, dtype=torch.bfloat16))


def _vcr_text_case():
    import torch
    q = torch.tensor([[11, 12, 13, 0], [21, 22, 0, 0]])
    qt = torch.tensor([[1, -1, 2, 0], [0, 3, 0, 0]])
    a = torch.tensor([[[31, 32, 0], [33, 0, 0]], [[41, 42, 43], [44, 45, 0]]])
    at = torch.tensor([[[2, -1, 0], [1, 0, 0]], [[-1, 0, 1], [2, 2, 0]]])
    return q, qt, q > 0, a, at, a > 0


def test_vcr_text_layouts_of_the_ablation_switches():
    """ResNetVLBERT._prepare_text (vcr mirror): the three token layouts of the reference's forward -- prepare_text_from_qa,
    _qa_onesent (QA_ONE_SENT) and _aq (ANSWER_FIRST), vcr/modules/resnet_vlbert_for_vcr.py:136-224 -- on a hand-checked case."""
    import importlib
    import torch
    M = importlib.import_module("vl-bert_amd.vcr.modules.resnet_vlbert_for_vcr")
    q, qt, qm, a, at, am = _vcr_text_case()
    C = a.shape[1]
    f = M.ResNetVLBERT._prepare_text
    ids, types, tags, mask = f(q, qt[:, None].expand(-1, C, -1), qm, a, at, am, order="qa")
    assert ids[0, 0].tolist() == [101, 11, 12, 13, 102, 31, 32, 102] and types[0, 0].tolist() == [0, 0, 0, 0, 0, 1, 1, 1]
    assert tags[0, 0].tolist() == [0, 1, -1, 2, 0, 2, -1, 0] and mask[0, 0].tolist() == [True] * 8
    assert ids[1, 1].tolist() == [101, 21, 22, 102, 44, 45, 102, 0] and mask[1, 1].tolist() == [True] * 7 + [False]
    ids, types, tags, mask = f(q, qt[:, None].expand(-1, C, -1), qm, a, at, am, order="qa_onesent")
    assert ids[0, 0].tolist() == [101, 11, 12, 13, 31, 32, 102] and int(types.sum()) == 0
    assert ids[1, 0].tolist() == [101, 21, 22, 41, 42, 43, 102] and tags[1, 0].tolist() == [0, 0, 3, -1, 0, 1, 0]
    ids, types, tags, mask = f(q, qt[:, None].expand(-1, C, -1), qm, a, at, am, order="aq")
    assert ids[0, 0].tolist() == [101, 31, 32, 102, 11, 12, 13, 102] and types[0, 0].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    assert ids[0, 1].tolist() == [101, 33, 102, 11, 12, 13, 102, 0] and mask[0, 1].tolist() == [True] * 7 + [False]


def test_vcr_text_layouts_match_the_reference_functions():
    """... and against the reference's own three functions where the reference tree is present (this container)."""
    import importlib
    import os
    import pytest
    import torch
    if not os.path.isdir(os.environ.get("VLBERT_REFERENCE_ROOT", "/root/reference")):
        pytest.skip("reference tree not present")
    from oracle import ref_import
    ref_import.import_reference()
    ref_import.install_roi_align_oracle()
    from vcr.modules.resnet_vlbert_for_vcr import ResNetVLBERT as Ref
    M = importlib.import_module("vl-bert_amd.vcr.modules.resnet_vlbert_for_vcr")

    class Stub:
        class tokenizer:
            @staticmethod
            def convert_tokens_to_ids(toks):
                return [101, 102]
    q, qt, qm, a, at, am = _vcr_text_case()
    C = a.shape[1]
    qtr = qt.repeat(1, C).view(qt.shape[0], C, -1)
    for order, fn in (("qa", Ref.prepare_text_from_qa), ("qa_onesent", Ref.prepare_text_from_qa_onesent), ("aq", Ref.prepare_text_from_aq)):
        r_ids, r_types, r_tags, r_mask = fn(Stub(), q, qtr, qm, a, at, am)
        ids, types, tags, mask = M.ResNetVLBERT._prepare_text(q, qtr, qm, a, at, am, order=order)
        assert torch.equal(ids, r_ids) and torch.equal(types, r_types) and torch.equal(tags, r_tags) and torch.equal(mask, r_mask.bool()), order


# ---- language-only BERT / RoBERTa initialisation (common/language_pretrained.py) ---------------------------------------------
def _fake_language_checkpoint(ref_sd, style, with_heads, seed=0):
    """A checkpoint in the naming of a language-only model, with tensors of the shapes `ref_sd` (a reference VisualLinguisticBert*
    state dict) expects: `bert.` + TF-style gamma / beta LayerNorm names, or `roberta.` + a one-row token-type table + `lm_head.*`."""
    import torch
    g = torch.Generator().manual_seed(seed)
    rnd = lambda t: torch.randn(t.shape, generator=g)
    prefix = "bert." if style == "bert" else "roberta."
    ck = {}
    for k, v in ref_sd.items():
        if k.startswith("encoder.") or k.startswith("pooler."):
            kk = k
            if style == "bert" and "LayerNorm" in k:
                kk = k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
            ck[prefix + kk] = rnd(v)
        elif k in ("word_embeddings.weight", "position_embeddings.weight"):
            ck[prefix + "embeddings." + k] = rnd(v)
        elif k == "token_type_embeddings.weight":
            ck[prefix + "embeddings." + k] = rnd(v[:2] if style == "bert" else v[:1])
        elif k.startswith("embedding_LayerNorm."):
            ck[prefix + "embeddings.LayerNorm." + ("gamma" if k.endswith("weight") else "beta")] = rnd(v)
    ck[prefix + "embeddings.bogus.weight"] = torch.zeros(3)            # unexpected under embeddings.
    ck[prefix + "something.else"] = torch.zeros(2)                     # base class: unexpected; pretraining class: silently dropped
    ck["optimizer.step"] = torch.zeros(1)                              # no bert. / roberta. prefix
    if with_heads:
        for k, v in ref_sd.items():
            if k.startswith("mlm_head.predictions."):
                k_ = k[len("mlm_head.predictions."):]
                if style == "bert":
                    ck["cls.predictions." + k_.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")] = rnd(v)
                else:
                    ck["lm_head." + k_.replace("transform.", "").replace("LayerNorm", "layer_norm")] = rnd(v)
            elif k.startswith("relationsip_head.caption_image_relationship."):
                ck["cls.seq_relationship." + k[len("relationsip_head.caption_image_relationship."):]] = rnd(v)
    return ck


def test_language_pretrained_plan_matches_the_reference_loader(tmp_path):
    """common/language_pretrained.plan + apply against the reference's own load_language_pretrained_model (both classes, BERT- and
    RoBERTa-style checkpoints) run on CPU in this container: same final parameters, same 'unexpected keys' list."""
    import importlib
    import os
    import pytest
    import torch
    if not os.path.isdir(os.environ.get("VLBERT_REFERENCE_ROOT", "/root/reference")):
        pytest.skip("reference tree not present")
    from oracle import ref_import
    ref_import.import_reference()
    from common.visual_linguistic_bert import VisualLinguisticBert as RefBase, VisualLinguisticBertForPretraining as RefPre
    lp = importlib.import_module("vl-bert_amd.common.language_pretrained")
    E = ref_import._EasyDict
    for with_heads in (False, True):
        for style in ("bert", "roberta"):
            for with_pooler in (False, True):
                cfg = E(dict(hidden_size=32, visual_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64, hidden_act="gelu",
                             hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=40, type_vocab_size=3,
                             vocab_size=50, initializer_range=0.02, visual_scale_text_init=0.0, visual_scale_object_init=0.0, visual_ln=True,
                             with_pooler=with_pooler, visual_region_classes=7, position_padding_idx=-1, obj_pos_id_relative=True,
                             word_embedding_frozen=False, pos_embedding_frozen=False))
                torch.manual_seed(1)
                kw = dict(with_rel_head=with_pooler) if with_heads else {}
                ref = (RefPre if with_heads else RefBase)(cfg, **kw)
                before = {k: v.clone() for k, v in ref.state_dict().items()}
                ck = _fake_language_checkpoint(before, style, with_heads, seed=3)
                path = str(tmp_path / ("ck_%d_%s_%d.bin" % (with_heads, style, with_pooler)))
                torch.save(ck, path)
                import io
                import contextlib
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    ref.load_language_pretrained_model(path)
                after = ref.state_dict()
                mine = {k: v.clone() for k, v in before.items()}
                assign, unexpected = lp.plan(torch.load(path), list(mine), with_pooler, pretraining=with_heads,
                                             with_rel_head=bool(kw.get("with_rel_head", False)), with_mlm_head=True)
                lp.apply(assign, mine)
                if with_heads:          # the reference's decoder is the tied word-embedding Parameter
                    mine["mlm_head.predictions.decoder.weight"] = mine["word_embeddings.weight"]
                for k in after:
                    assert torch.equal(after[k], mine[k]), (with_heads, style, with_pooler, k)
                assert any(not torch.equal(after[k], before[k]) for k in after)
                assert buf.getvalue().strip() == "Warnings: Unexpected keys: {}.".format(unexpected), (buf.getvalue(), unexpected)


def test_language_pretrained_strict_parts_and_path_resolution(tmp_path):
    import importlib
    import pytest
    import torch
    lp = importlib.import_module("vl-bert_amd.common.language_pretrained")
    own = ["word_embeddings.weight", "embedding_LayerNorm.weight", "embedding_LayerNorm.bias", "encoder.layer.0.a.weight", "encoder.layer.0.a.bias"]
    ck = {"bert.embeddings.LayerNorm.gamma": torch.ones(4), "bert.embeddings.LayerNorm.beta": torch.zeros(4),
          "bert.encoder.layer.0.a.weight": torch.ones(2, 2)}
    with pytest.raises(RuntimeError, match="Missing key"):           # the encoder is loaded strictly: its bias is absent
        lp.plan(ck, own, with_pooler=False)
    ck["bert.encoder.layer.0.a.bias"] = torch.zeros(2)
    assign, unexpected = lp.plan(ck, own, with_pooler=False)
    assert unexpected == [] and [a[0] for a in assign] == ["embedding_LayerNorm.weight", "embedding_LayerNorm.bias", "encoder.layer.0.a.weight",
                                                            "encoder.layer.0.a.bias"]
    with pytest.raises(RuntimeError, match="expects"):               # a table of another size cannot replace a flat-buffer view
        lp.apply([("word_embeddings.weight", torch.zeros(5, 4), None)], {"word_embeddings.weight": torch.zeros(6, 4)})
    # the wrappers' checkpoint choice
    assert lp.resolve_path({"BERT_PRETRAINED": "", "BERT_MODEL_NAME": "bert-base-uncased"}) is None
    assert lp.resolve_path({"BERT_PRETRAINED": "/x/bert", "BERT_PRETRAINED_EPOCH": 7, "BERT_MODEL_NAME": str(tmp_path)}) == "/x/bert-0007.model"
    assert lp.resolve_path({"BERT_PRETRAINED": "", "BERT_MODEL_NAME": str(tmp_path)}) is None
    (tmp_path / "pytorch_model.bin").write_bytes(b"")
    assert lp.resolve_path({"BERT_PRETRAINED": "", "BERT_MODEL_NAME": str(tmp_path)}) == str(tmp_path / "pytorch_model.bin")
    sd, keys = lp.mlm_transform_state_dict({"cls.predictions.transform.LayerNorm.gamma": 1, "cls.predictions.bias": 2, "bert.x": 3})
    assert sd == {"LayerNorm.weight": 1} and keys == ["cls.predictions.transform.LayerNorm.gamma"]


def test_bench_other_configs_summarises_children_and_survives_failures(monkeypatch):
    """bench.other_configs(): the child bench lines of BASELINE configs 3 / 4 / 5 are condensed into the default line; a failing or
    hanging child becomes an {"error": ...} entry and never an exception (the headline line must still print)."""
    import importlib.util
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vlb_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    class R:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err

    def fake_run(cmd, **kw):
        calls.append(cmd)
        assert kw.get("timeout") and kw.get("capture_output")
        if "--e2e" in cmd:
            line = json.dumps({"metric": "samples/sec e2e", "value": 740.0, "unit": "samples/s", "ms_per_step": 21.6, "dtype": "bf16",
                               "config": {"workload": "C3"}, "roofline": {"achieved": 320.4, "frac": 0.128, "peak": 2500.0}})
            return R(0, "some warning on stdout\n" + line + "\n")
        raise subprocess.TimeoutExpired(cmd, kw["timeout"])
    monkeypatch.setattr(subprocess, "run", fake_run)
    out = bench.other_configs()
    assert set(out) == {"config3_e2e", "config4_vqa_fp32", "config5_vcr_fp16"} and len(calls) == 3
    assert "--precision" in calls[2] and "f16" in calls[2] and "--vcr" in calls[2]          # config 5 at its named ("mixed") precision
    c3 = out["config3_e2e"]
    assert c3["value"] == 740.0 and c3["ms_per_step"] == 21.6 and c3["gemm_frac_of_peak"] == 0.128 and c3["workload"] == "C3"
    assert c3["cmd"].startswith("python bench.py --e2e") and "--no-cpu-baseline" in c3["cmd"]
    assert "error" in out["config4_vqa_fp32"] and "TimeoutExpired" in out["config4_vqa_fp32"]["error"]
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: R(1, "", "boom"))
    out = bench.other_configs()
    assert all("error" in v and "boom" in v["error"] for v in out.values())


def test_checkpoint_parameter_order_is_the_reference_models():
    """The index order of the optimizer state in a checkpoint (common/checkpoint.reference_param_order) against the reference's own
    `named_parameters()` order, recorded from the reference's modules by oracle/make_checkpoint_golden.py (plain, pooler + relationship
    head, multitask), and against the parameter order inside the checkpoint file the reference's Checkpoint callback wrote."""
    import json
    import os
    E = importlib.import_module("vl-bert_amd.engine")
    C = importlib.import_module("vl-bert_amd.common.checkpoint")
    gold = os.path.join(os.path.dirname(__file__), "golden", "checkpoint")
    with open(os.path.join(gold, "param_order.json")) as f:
        order = json.load(f)
    small = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128, vocab_size=300,
                 max_position_embeddings=64, visual_region_classes=50)
    for tag, kw in (("plain", {}), ("pooler_rel", dict(with_pooler=True, with_rel_loss=True)), ("multitask", dict(multitask=True))):
        shapes = E.param_layout(E.ModelConfig(**small, **kw))
        assert C.reference_param_order(shapes) == order[tag], tag
        assert C.reference_param_order(reversed(list(shapes))) == order[tag], tag          # independent of the input order
    ck = torch.load(os.path.join(gold, "ref_small-0000.model"), map_location="cpu", weights_only=False)
    shapes = E.param_layout(E.ModelConfig(**small))
    names = C.reference_param_order(shapes)
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(len(names)))
    for i, n in enumerate(names):
        assert tuple(ck["optimizer"]["state"][i]["exp_avg"].shape) == tuple(shapes[n]), n
    assert set(ck["state_dict"]) == set(shapes) | {E.TIED_DECODER_KEY}
    assert C.checkpoint_path("out/vl-bert", 3) == "out/vl-bert-0003.model"
    # e2e (IMAGE_FEAT_PRECOMPUTED false): the reference indexes EVERY parameter of the image branch, frozen stages and frozen BatchNorm
    # weights / biases included (round-4 ADVICE: the engine only knew its trainable convolutions) -- the structural table against the
    # real FastRCNN module's named_parameters(), and the engine's trainable set against the module's requires_grad set
    V = importlib.import_module("vl-bert_amd.vision")
    for nl in (50, 101):
        vis = C.reference_vision_param_names(nl)
        assert vis == order["e2e_fastrcnn_%d" % nl], nl
        shapes = E.param_layout(E.ModelConfig(**small, e2e=True, image_num_layers=nl))
        full = C.reference_param_order(shapes, e2e_num_layers=nl)
        assert full[:len(vis)] == ["image_feature_extractor." + n for n in vis]
        assert full[len(vis):] == [n for n in order["plain"] if not n.startswith("image_feature_extractor.")]
        assert set(shapes) <= set(full) and len(set(full)) == len(full)
        trainable = {"image_feature_extractor." + n for n in order["e2e_fastrcnn_%d_trainable" % nl]}
        assert {n for n in shapes if n.startswith("image_feature_extractor.")} == trainable, nl


def test_finetune_entry_points_resolve_reference_style_configs_and_schedules():
    """vqa / vcr train_end2end --dry-run on reference-style YAMLs (lr = LR x world x batch x accumulate, optimiser, clip, compute from
    TRAIN.FP16) and the two LR schedules against torch LambdaLR-free restatements -- and against the reference's own scheduler classes
    where the reference tree is present (build container)."""
    import os
    import sys
    F = importlib.import_module("vl-bert_amd.common.finetune_entry")
    fx = os.path.join(os.path.dirname(__file__), "fixtures")
    r = F.main("vqa", ["--cfg", os.path.join(fx, "vqa_small.yaml"), "--dry-run", "--compute", "cfg"])
    assert r["optimizer"] == "AdamW" and abs(r["lr"] - 1.0e-5 * 2 * 2) < 1e-12 and r["compute"] == "fp32" and r["clip_grad_norm"] == 1.0
    r = F.main("vcr", ["--cfg", os.path.join(fx, "vcr_small.yaml"), "--dry-run", "--compute", "cfg"])
    assert r["optimizer"] == "SGD" and abs(r["lr"] - 7.0e-5 * 2 * 2) < 1e-12 and r["compute"] == "fp16" and r["loss_scale"] == 128.0
    cfg = F.load_config("vcr", os.path.join(fx, "vcr_small.yaml"))
    assert cfg.NETWORK.VLBERT.with_pooler is True and cfg.NETWORK.CNN_LOSS_TOP is True
    f = F.lr_lambda(cfg, 10)          # 10 micro-batches per epoch / accumulate 2 -> milestones at optimizer steps 70 and 90
    assert [round(f(k), 6) for k in range(6)] == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0] and f(69) == 1.0 and abs(f(70) - 0.1) < 1e-12 \
        and abs(f(90) - 0.01) < 1e-12
    cfg = F.load_config("vqa", os.path.join(fx, "vqa_small.yaml"))
    g = F.lr_lambda(cfg, 8)           # t_total = 5 x 8 / 2 = 20, warm-up 3
    assert [round(g(k), 6) for k in range(5)] == [0.0, round(1 / 3, 6), round(2 / 3, 6), 1.0, round(16 / 17, 6)] and g(20) == 0.0
    ref = os.environ.get("VLBERT_REFERENCE_ROOT", "/root/reference")
    if os.path.isdir(ref):            # the reference's own classes (not available on the GPU box; this is a CPU test)
        sys.path.insert(0, ref)
        try:
            from common.lr_scheduler import WarmupMultiStepLR
            from common.nlp.bert.optimization import WarmupLinearSchedule
        finally:
            sys.path.remove(ref)
        for sched, fn in ((lambda o: WarmupMultiStepLR(o, milestones=[70, 90], gamma=0.1, warmup_factor=0.0, warmup_iters=4,
                                                       warmup_method="linear"), f),
                          (lambda o: WarmupLinearSchedule(o, 3, t_total=20), g)):
            p = torch.nn.Parameter(torch.zeros(1))
            opt = torch.optim.SGD([p], lr=1.0)
            s = sched(opt)
            for k in range(95):
                assert abs(opt.param_groups[0]["lr"] - fn(k)) < 1e-9, (k, opt.param_groups[0]["lr"], fn(k))
                opt.step()
                s.step()


def test_partial_pretrain_renaming_and_partial_load():
    """NETWORK.PARTIAL_PRETRAIN of the fine-tuning entry points: key renaming by the first matching prefix rule, the VCR extras (relationship
    head -> 1-logit classifier, segment-embedding init), then the tolerant partial load -- against the reference's own
    smart_partial_load_model_state_dict where the reference tree is present, and against hand-checked expectations everywhere."""
    import importlib
    import torch
    C = importlib.import_module("vl-bert_amd.common.checkpoint")
    g = torch.Generator().manual_seed(0)
    pre = {
        "module.vlbert.mlm_head.predictions.transform.dense.weight": torch.randn(4, 4, generator=g),
        "module.vlbert.word_embeddings.weight": torch.randn(6, 4, generator=g),
        "module.vlbert.token_type_embeddings.weight": torch.randn(3, 4, generator=g).half(),
        "module.vlbert.relationsip_head.caption_image_relationship.weight": torch.randn(2, 4, generator=g),
        "module.vlbert.relationsip_head.caption_image_relationship.bias": torch.randn(2, generator=g),
        "module.unrelated.weight": torch.randn(2, generator=g),
    }
    rules = ["module.vlbert.mlm_head.predictions.transform->module.final_mlp.0", "module.vlbert->module.vlbert._module", "vlbert->vlbert._module"]
    out = C.partial_pretrain_state_dict(pre, rules, load_rel_head=True, segmb_init=True)
    assert set(out) == {"module.final_mlp.0.dense.weight", "module.vlbert._module.word_embeddings.weight", "module.vlbert._module.token_type_embeddings.weight",
                        "module.vlbert._module.relationsip_head.caption_image_relationship.weight",
                        "module.vlbert._module.relationsip_head.caption_image_relationship.bias", "module.unrelated.weight",
                        "module.final_mlp.1.weight", "module.final_mlp.1.bias"}
    w = pre["module.vlbert.relationsip_head.caption_image_relationship.weight"]
    assert torch.equal(out["module.final_mlp.1.weight"], w[1:2] - w[0:1]) and out["module.final_mlp.1.weight"].shape == (1, 4)
    tt = out["module.vlbert._module.token_type_embeddings.weight"]
    assert tt.dtype == torch.float32 and torch.equal(tt[1], tt[0]) and torch.equal(tt[2], pre["module.vlbert.token_type_embeddings.weight"][2].float())
    assert pre["module.vlbert.token_type_embeddings.weight"].dtype == torch.float16          # the caller's dict is not modified
    assert C.partial_pretrain_state_dict(pre, []) == pre

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.final_mlp = torch.nn.Sequential(torch.nn.Sequential(), torch.nn.Linear(4, 1))
            self.final_mlp[0].add_module("dense", torch.nn.Linear(4, 4, bias=False))
            self.other = torch.nn.Linear(2, 2)

    def fresh():
        torch.manual_seed(3)
        return Toy()
    mine = fresh()
    before = {k: v.clone() for k, v in mine.state_dict().items()}
    lines = []
    taken, unmatched = C.smart_partial_load(mine, out, log=lines.append)
    assert sorted(taken) == ["final_mlp.0.dense.weight", "final_mlp.1.bias", "final_mlp.1.weight"] and len(unmatched) == 5 and len(lines) == 3
    after = mine.state_dict()
    assert torch.equal(after["final_mlp.0.dense.weight"], pre["module.vlbert.mlm_head.predictions.transform.dense.weight"])
    assert torch.equal(after["final_mlp.1.weight"], w[1:2] - w[0:1]) and torch.equal(after["other.weight"], before["other.weight"])
    import os
    ref_root = os.environ.get("VLBERT_REFERENCE_ROOT", "/root/reference")
    if os.path.isdir(ref_root):
        from oracle import ref_import
        ref_import.import_reference()
        from common.utils.load import smart_partial_load_model_state_dict
        theirs = fresh()
        smart_partial_load_model_state_dict(theirs, out)
        for k, v in theirs.state_dict().items():
            assert torch.equal(v, after[k]), k


def test_partial_pretrain_reports_and_bounds_shape_mismatches():
    """Round-4 ADVICE: tensors dropped for a shape mismatch in front of smart_partial_load are logged with both shapes, and a
    checkpoint of a different architecture (most name-matching tensors with another shape) is refused instead of silently leaving the
    network at its random initialisation (the reference raises in load_state_dict, common/utils/load.py:57-81)."""
    C = importlib.import_module("vl-bert_amd.common.checkpoint")
    own = {"vlbert.a.weight": torch.zeros(3, 4), "vlbert.b.weight": torch.zeros(5), "final_mlp.weight": torch.zeros(2, 4)}
    logs = []
    kept, dropped = C.drop_shape_mismatches({"module.vlbert.a.weight": torch.ones(3, 4), "vlbert.b.weight": torch.ones(5),
                                             "final_mlp.weight": torch.ones(7, 4), "unknown": torch.ones(1)}, own, log=logs.append)
    assert sorted(kept) == ["module.vlbert.a.weight", "unknown", "vlbert.b.weight"]
    assert dropped == [("final_mlp.weight", (7, 4), (2, 4))] and "final_mlp.weight: file (7, 4) vs model (2, 4)" in logs[0]
    with pytest.raises(ValueError, match="different architecture"):
        C.drop_shape_mismatches({"vlbert.a.weight": torch.ones(6, 8), "vlbert.b.weight": torch.ones(10), "final_mlp.weight": torch.ones(2, 4)},
                                own, log=logs.append)


def test_traffic_stamp_hash_is_the_same_function_in_bench_and_profile_report():
    """`roofline.traffic` is quoted from profiles/<tag>_gemm_traffic.json only when that file's `gemm_sources_sha` equals the hash of
    the GEMM sources the bench runs (round-4 review: builder-side PMC data must not pass through a driver record unlabelled):
    tools/profile_report.py stamps the file, bench.py checks it -- both must compute the same hash, and a stale file is named, not echoed."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = importlib.import_module("bench")
    src = open(os.path.join(root, "tools", "profile_report.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "gemm_sources_sha")
    ns = {"os": os, "__file__": os.path.join(root, "tools", "profile_report.py")}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "profile_report.py", "exec"), ns)
    assert ns["gemm_sources_sha"]() == bench.gemm_sources_sha() and len(bench.gemm_sources_sha()) == 12
