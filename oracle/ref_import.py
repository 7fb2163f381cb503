"""Import the *real* reference (jackroos/VL-BERT at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/make_golden.py` (run in the build
container, where /root/reference exists) to generate the golden fixtures under
`tests/golden/`.  Nothing in the product path, `bench.py`, `smoke()` or the
`-m gpu` tests imports this file: /root/reference does not exist on the GPU box.

The reference cannot be imported as-is in this image:
  * external/pytorch_pretrained_bert/file_utils.py:18-20 imports boto3/botocore
    (absent)  -> empty stub modules;
  * common/fast_rcnn.py:10-11 imports the compiled extension
    common.lib.roi_pooling.C_ROIPooling                     -> stub module (the
    precomputed-feature path never calls it);
  * pretrain/function/config.py:1 imports easydict (absent) -> 10-line stand-in;
  * BertTokenizer.from_pretrained(dir) needs dir/vocab.txt  -> synthetic file.
No reference source is copied or modified.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VLBERT_REFERENCE_ROOT", "/root/reference")


class _EasyDict(dict):
    """Minimal attribute-dict standing in for the `easydict` package."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(_EasyDict(x) if isinstance(x, dict) else x for x in v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__


def install_stubs():
    for name in ("boto3", "botocore", "botocore.exceptions"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "botocore.exceptions":
                m.ClientError = type("ClientError", (Exception,), {})
            sys.modules[name] = m
    sys.modules["botocore"].exceptions = sys.modules["botocore.exceptions"]
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _EasyDict
        sys.modules["easydict"] = m
    # the compiled ROI pooling extension (only used by the e2e image path)
    name = "common.lib.roi_pooling.C_ROIPooling"
    if name not in sys.modules:
        m = types.ModuleType(name)

        def _missing(*a, **k):
            raise RuntimeError("C_ROIPooling stub: reference extension not built")

        m.roi_align_forward = m.roi_align_backward = _missing
        m.roi_pool_forward = m.roi_pool_backward = _missing
        sys.modules[name] = m


def install_roi_align_oracle():
    """Give the reference's `_ROIAlign` autograd function (common/lib/roi_pooling/roi_align.py:11-44) a C_ROIPooling to call when its
    own FastRCNN e2e module is driven for the vision / VCR golden fixtures.  FORWARD: the reference's OWN compiled CPU kernel
    (oracle/_ref/libroi_align_ref.so = cpu/ROIAlign_cpu.cpp built unmodified by oracle/build_ref.sh) -- the fixture's forward
    arithmetic is the reference binary's, not a restatement.  BACKWARD: the reference has no CPU backward (ROIAlign.h:44 raises), so
    the restatement of its CUDA backward (oracle/roi_align_oracle.py, pinned to the forward as its exact adjoint) stays."""
    import numpy as np
    import torch
    from . import ref_roi_align as REF
    from . import roi_align_oracle as RA
    install_stubs()
    m = sys.modules["common.lib.roi_pooling.C_ROIPooling"]
    if not REF.available():
        raise RuntimeError("oracle/_ref/libroi_align_ref.so is missing: run oracle/build_ref.sh (the fixtures' ROIAlign forward is the "
                           "reference's compiled kernel)")

    def fwd(inp, rois, scale, ph, pw, sr):
        if os.environ.get("VLB_FIXTURE_ROI") == "numpy":      # (debug: the restatement instead of the reference binary)
            return torch.from_numpy(np.ascontiguousarray(RA.roi_align_forward(inp.detach().numpy().astype(np.float32),
                                                                              rois.numpy().astype(np.float32), scale, ph, pw, sr), dtype=np.float32))
        return torch.from_numpy(REF.roi_align_forward(inp.detach().numpy().astype(np.float32), rois.numpy().astype(np.float32),
                                                      float(scale), int(ph), int(pw), int(sr)))

    def bwd(grad, rois, scale, ph, pw, b, c, h, w, sr):
        return torch.from_numpy(np.ascontiguousarray(RA.roi_align_backward(grad.contiguous().numpy().astype(np.float32),
                                                                           rois.numpy().astype(np.float32), scale, ph, pw, b, c, h, w, sr), dtype=np.float32))

    m.roi_align_forward, m.roi_align_backward = fwd, bwd


def make_vocab_dir(path, vocab_size):
    """Synthetic vocab.txt so BertTokenizer.from_pretrained(<dir>) works offline
    (external/pytorch_pretrained_bert/tokenization.py:119-153)."""
    os.makedirs(path, exist_ok=True)
    special = {0: "[PAD]", 100: "[UNK]", 101: "[CLS]", 102: "[SEP]", 103: "[MASK]"}
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        for i in range(vocab_size):
            f.write(special.get(i, "tok%d" % i) + "\n")
    return path


def import_reference():
    """Returns the reference's ResNetVLBERTForPretraining class and AdamW."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # `common.lib.roi_pooling` must resolve to a package whose C_ROIPooling
    # attribute is the stub installed above.
    import common.lib.roi_pooling as rp  # noqa: E402
    rp.C_ROIPooling = sys.modules["common.lib.roi_pooling.C_ROIPooling"]
    from pretrain.modules.resnet_vlbert_for_pretraining import ResNetVLBERTForPretraining
    from common.nlp.bert.optimization import AdamW
    return ResNetVLBERTForPretraining, AdamW


def import_reference_multitask():
    import_reference()
    from pretrain.modules.resnet_vlbert_for_pretraining_multitask import ResNetVLBERTForPretrainingMultitask
    return ResNetVLBERTForPretrainingMultitask


def make_reference_config(cfg, vocab_dir):
    """Attr-dict in the shape pretrain/function/config.py:52-132 defines, filled
    from an oracle `VLBertConfig` (cfgs/pretrain/base_prec_withouttextonly_4x16G_fp32.yaml)."""
    E = _EasyDict
    return E(dict(
        NETWORK=dict(
            IMAGE_FEAT_PRECOMPUTED=True, IMAGE_SEMANTIC=False, IMAGE_FROZEN_BN=True, MASK_RAW_PIXELS=True,
            IMAGE_FINAL_DIM=cfg.hidden_size, BERT_MODEL_NAME=vocab_dir, BERT_PRETRAINED="",
            BERT_PRETRAINED_EPOCH=0,
            WITH_REL_LOSS=cfg.with_rel_loss, WITH_MLM_LOSS=True, WITH_MVRC_LOSS=True,
            MLM_LOSS_NORM_IN_BATCH_FIRST=False, MVRC_LOSS_NORM_IN_BATCH_FIRST=False,
            VLBERT=dict(
                from_scratch=True, word_embedding_frozen=False, pos_embedding_frozen=False,
                obj_pos_id_relative=True, hidden_size=cfg.hidden_size, visual_size=cfg.hidden_size,
                num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                intermediate_size=cfg.intermediate_size, hidden_act="gelu",
                hidden_dropout_prob=cfg.hidden_dropout_prob,
                attention_probs_dropout_prob=cfg.attention_probs_dropout_prob,
                max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=3,
                vocab_size=cfg.vocab_size, initializer_range=0.02,
                visual_scale_text_init=0.0, visual_scale_object_init=0.0, visual_ln=True,
                with_pooler=cfg.with_pooler, visual_region_classes=cfg.visual_region_classes,
                position_padding_idx=-1, input_transform_type=1, object_word_embed_mode=2,
            ),
        ),
    ))
