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
