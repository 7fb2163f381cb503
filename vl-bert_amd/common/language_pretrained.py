"""Initialisation from a language-only BERT / RoBERTa checkpoint -- `load_language_pretrained_model` of the reference
(common/visual_linguistic_bert.py:243-309 for `VisualLinguisticBert`, :382-469 for `VisualLinguisticBertForPretraining`) as pure host
logic over state-dict keys, so that it can be checked on CPU against the reference's own method (tests/test_host_logic_cpu.py) and
applied to the flat parameter buffer of the HIP mirrors (visual_linguistic_bert.py).

`plan(...)` turns a checkpoint's state dict into a list of assignments (own key, tensor, row selector); nothing is copied here.
Behaviour kept from the reference, including its oddities:
  * keys must start with `bert.` / `roberta.` (base class: anything else is "unexpected"); TF-style `gamma` / `beta` -> `weight` / `bias`;
  * `embeddings.token_type_embeddings.weight` fills the first rows only; a one-row table (RoBERTa) is replicated into rows 1 and 2 by
    the base class (:274-281) and into row 1 only by the pretraining class (:415-419);
  * the embedding LayerNorm and the encoder are loaded STRICTLY (`load_state_dict` of the sub-module: a missing or unexpected key is a
    RuntimeError), the pooler only if the checkpoint has pooler keys, the relationship head only if it has `cls.seq_relationship.*`;
  * the pretraining class maps `cls.predictions.*` / `lm_head.*` onto `mlm_head.predictions.*` (strict, decoder tied to the word
    embeddings) and silently ignores `bert.*` keys outside encoder / embeddings / pooler (no `else` at :435).
The word / position embedding tables REPLACE the parameter's data in the reference (`.data = v`), whatever their shape; the mirrors own
fixed-shape views of a flat buffer, so a shape mismatch raises instead (a vocabulary of another size needs another config).
"""
import os

BERT_WEIGHTS_NAME = "pytorch_model.bin"      # (vqa/modules/resnet_vlbert_for_vqa.py:11)


def _cfg(obj, name, default=None):
    return obj.get(name, default) if isinstance(obj, dict) else getattr(obj, name, default)


def resolve_path(network_config):
    """The wrappers' choice of checkpoint (pretrain/modules/resnet_vlbert_for_pretraining.py:29-41, vqa/.../resnet_vlbert_for_vqa.py:37-48,
    vcr/.../resnet_vlbert_for_vcr.py:49-58): `BERT_PRETRAINED-<epoch:04d>.model`, else `BERT_MODEL_NAME/pytorch_model.bin` when that
    directory holds one, else None (the caller prints the reference's warning and trains from scratch)."""
    pre = _cfg(network_config, "BERT_PRETRAINED", "") or ""
    if pre != "":
        return "{}-{:04d}.model".format(pre, int(_cfg(network_config, "BERT_PRETRAINED_EPOCH", 0)))
    name = _cfg(network_config, "BERT_MODEL_NAME", "") or ""
    if name and os.path.isdir(name):
        weight_path = os.path.join(name, BERT_WEIGHTS_NAME)
        if os.path.isfile(weight_path):
            return weight_path
    return None


def _tf_names(k):
    if "gamma" in k:
        k = k.replace("gamma", "weight")
    if "beta" in k:
        k = k.replace("beta", "bias")
    return k


def _strict(sub, got, own_keys, what):
    """nn.Module.load_state_dict(strict=True) of sub-module `sub` given the keys `got`: every own key under `sub.` must be present."""
    want = {k[len(sub):] for k in own_keys if k.startswith(sub)}
    missing = sorted(want - set(got))
    if missing:
        raise RuntimeError("Error(s) in loading state_dict for %s: Missing key(s) in state_dict: %s" % (what, ", ".join(repr(m) for m in missing)))


def plan(pretrained_state_dict, own_keys, with_pooler, pretraining=False, with_rel_head=False, with_mlm_head=True):
    """-> (assign, unexpected): assign = [(own key, tensor, rows)], rows = None (whole tensor) | slice / int (leading rows of the own
    tensor that receive `tensor`).  own_keys: the module's state-dict keys (`encoder.layer.0...`, `pooler.dense.weight`, ...; for the
    pretraining class also `mlm_head.predictions.*`, `relationsip_head.caption_image_relationship.*`)."""
    own = set(own_keys)
    assign, unexpected = [], []          # (the embedding tables are assigned inside the reference's loop, the sub-modules after it)
    enc, ln, pool, rel, mlm = {}, {}, {}, {}, {}
    late = {"ln": [], "enc": [], "pool": [], "rel": [], "mlm": []}
    for _k, v in pretrained_state_dict.items():
        if _k.startswith("bert.") or _k.startswith("roberta."):
            k = _tf_names(_k[len("bert."):] if _k.startswith("bert.") else _k[len("roberta."):])
            bad = _k if pretraining else k            # (the two classes report different spellings of an unexpected key)
            if k.startswith("encoder."):
                if k in own:
                    enc[k[len("encoder."):]] = v
                    late["enc"].append((k, v, None))
                else:
                    unexpected.append(bad)
            elif k.startswith("embeddings."):
                k_ = k[len("embeddings."):]
                if k_ in ("word_embeddings.weight", "position_embeddings.weight"):
                    assign.append((k_, v, None))
                elif k_ == "token_type_embeddings.weight":
                    n = v.shape[0]
                    assign.append((k_, v, slice(0, n)))
                    if n == 1:
                        assign.append((k_, v[0], 1))
                        if not pretraining:
                            assign.append((k_, v[0], 2))
                elif k_.startswith("LayerNorm."):
                    k__ = k_[len("LayerNorm."):]
                    if "embedding_LayerNorm." + k__ in own:
                        ln[k__] = v
                        late["ln"].append(("embedding_LayerNorm." + k__, v, None))
                    else:
                        unexpected.append(bad)
                else:
                    unexpected.append(bad)
            elif with_pooler and k.startswith("pooler."):
                if k in own:
                    pool[k[len("pooler."):]] = v
                    late["pool"].append((k, v, None))
                else:
                    unexpected.append(bad)
            elif not pretraining:
                unexpected.append(bad)
        elif pretraining and _k.startswith("cls.seq_relationship.") and with_rel_head:
            k_ = _tf_names(_k[len("cls.seq_relationship."):])
            full = "relationsip_head.caption_image_relationship." + k_
            if full in own:
                rel[k_] = v
                late["rel"].append((full, v, None))
            else:
                unexpected.append(_k)
        elif pretraining and (_k.startswith("cls.predictions.") or _k.startswith("lm_head.")) and with_mlm_head:
            k_ = _k[len("cls.predictions."):] if _k.startswith("cls.predictions.") else _k[len("lm_head."):]
            if _k.startswith("lm_head."):
                if "dense" in k_ or "layer_norm" in k_:
                    k_ = "transform." + k_
                if "layer_norm" in k_:
                    k_ = k_.replace("layer_norm", "LayerNorm")
            k_ = _tf_names(k_)
            full = "mlm_head.predictions." + k_
            if full in own:
                mlm[k_] = v
                late["mlm"].append((full if k_ != "decoder.weight" else "word_embeddings.weight", v, None))   # tied (modeling.py:463-466)
            else:
                unexpected.append(_k)
        else:
            unexpected.append(_k)
    _strict("embedding_LayerNorm.", ln, own, "BertLayerNorm")
    _strict("encoder.", enc, own, "BertEncoder")
    if with_pooler and pool:
        _strict("pooler.", pool, own, "BertPooler")
    if pretraining and with_rel_head and rel:
        _strict("relationsip_head.caption_image_relationship.", rel, own, "Linear")
    if pretraining and with_mlm_head:
        _strict("mlm_head.predictions.", mlm, own, "BertLMPredictionHead")
    for part in ("ln", "enc", "pool", "rel", "mlm"):      # the order of the load_state_dict calls at :303-309 / :459-469
        assign += late[part]
    return assign, unexpected


def mlm_transform_state_dict(pretrained_state_dict):
    """The `cls.predictions.transform.*` entries under BertPredictionHeadTransform's own names, and the checkpoint keys they came from:
    how the VQA wrapper's "mlm" classifier starts from the language model's MLM transform (vqa/modules/resnet_vlbert_for_vqa.py:97-110)."""
    out, keys = {}, []
    for k, v in pretrained_state_dict.items():
        if k.startswith("cls.predictions.transform."):
            keys.append(k)
            out[_tf_names(k[len("cls.predictions.transform."):])] = v
    return out, keys


def apply(assign, tensors, strict_keys=False):
    """Carry out a plan on `tensors` (own key -> tensor, e.g. a module's named parameters); shapes must match.  strict_keys: the plan
    must name every tensor exactly once-or-more and nothing else (a sub-module's strict load_state_dict)."""
    import torch
    if strict_keys:
        got, want = {k for k, _, _ in assign}, set(tensors)
        if got != want:
            raise RuntimeError("Error(s) in loading state_dict: Missing key(s): %s; Unexpected key(s): %s" % (sorted(want - got), sorted(got - want)))
    with torch.no_grad():
        for key, v, rows in assign:
            dst = tensors[key]
            if rows is not None:
                dst = dst[rows]
            if tuple(dst.shape) != tuple(v.shape):
                raise RuntimeError("language-pretrained checkpoint: %s has shape %s, this model expects %s (the reference would replace the "
                                   "table; the flat-buffer mirrors cannot)" % (key, tuple(v.shape), tuple(dst.shape)))
            dst.copy_(v.to(dtype=dst.dtype, device=dst.device))
