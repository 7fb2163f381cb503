"""Checkpoint files in the reference's on-disk format, for the fused engine.

What it replaces: `Checkpoint.__call__` (common/callbacks/epoch_end_callbacks/checkpoint.py:12-21) -- after every epoch rank 0 writes
`{prefix}-{epoch:04d}.model` = torch.save({'state_dict': net.state_dict(), 'optimizer': optimizer.state_dict()}) -- and `smart_resume`
(common/utils/load.py:20-54): TRAIN.RESUME loads `{prefix}-{BEGIN_EPOCH-1:04d}.model`, TRAIN.AUTO_RESUME the newest existing epoch file,
model weights through the 'module.' prefix tolerant loader, the optimizer through `optimizer.load_state_dict`.

The engine keeps parameters, gradients and Adam moments in flat buffers; here they are presented in the layout the reference's files
have, so either side reads the other's checkpoints:
  * 'state_dict': the reference's key names (engine.state_dict(), tied decoder key included), CPU fp32 tensors;
  * 'optimizer': torch.optim's state-dict form of the reference's AdamW (common/nlp/bert/optimization.py:107-187): 'state' = {index:
    {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups' = [one group: lr, betas, eps, weight_decay, correct_bias, params = indices] with
    the indices in the order of the reference model's `named_parameters()` (pretrain/function/train.py:140-160 builds the group in that
    order; pinned by tests/golden/checkpoint/param_order.json, produced from the reference's own modules).  An extra top-level key
    'param_names' (ignored by torch's loader) records the name of every index, so a resume never depends on ordering conventions.
With the sharded data-parallel optimizer the master weights and moments are authoritative on the owning rank: saving is a COLLECTIVE
(every rank calls save_checkpoint, rank 0 writes).  Host glue: no arithmetic here.
"""
import os
from collections import OrderedDict

import torch


_RESNET_BLOCKS = {18: None, 34: None, 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def reference_vision_param_names(num_layers):
    """`[n for n, _ in FastRCNN(...).named_parameters()]` of the reference's e2e image branch (common/fast_rcnn.py:54-109 over
    common/backbone/resnet/resnet.py:62-118,121-199: Bottleneck registers conv1, bn1, conv2, bn2, conv3, bn3, downsample in that order;
    the module registers backbone, roi_head_feature_extractor (= layer4's blocks), obj_downsample), WITHOUT the wrapper's
    'image_feature_extractor.' prefix.  Every tensor is listed whether it trains or not: frozen stages and the frozen BatchNorm
    weights / biases are nn.Parameters with requires_grad False, the reference's optimizer indexes them like any other and simply
    never creates state for them.  Pinned against the real module by tests/golden/checkpoint/param_order.json (e2e_fastrcnn_*)."""
    blocks = _RESNET_BLOCKS.get(int(num_layers))
    if blocks is None:
        raise ValueError("the e2e image branch is built for ResNet-50 / 101 / 152 (got %r)" % (num_layers,))
    names = ["backbone.conv1.weight", "backbone.bn1.weight", "backbone.bn1.bias"]

    def bottleneck(prefix, downsample):
        for i in (1, 2, 3):
            names.extend([prefix + "conv%d.weight" % i, prefix + "bn%d.weight" % i, prefix + "bn%d.bias" % i])
        if downsample:
            names.extend([prefix + "downsample.0.weight", prefix + "downsample.1.weight", prefix + "downsample.1.bias"])
    for stage in (1, 2, 3):
        for b in range(blocks[stage - 1]):
            bottleneck("backbone.layer%d.%d." % (stage, b), b == 0)
    for b in range(blocks[3]):
        bottleneck("roi_head_feature_extractor.%d." % b, b == 0)
    names.extend(["obj_downsample.1.weight", "obj_downsample.1.bias"])
    return names


def reference_param_order(names, e2e_num_layers=None):
    """`names`: the engine's parameter names (any order) -> the order of the reference model's named_parameters().
    e2e_num_layers (the e2e engine: cfg.image_num_layers): the list then holds EVERY parameter of the reference's image branch --
    frozen convolutions and BatchNorm weights / biases included, which the engine does not keep as optimizer tensors (`names` lacks
    them) -- because the reference's optimizer indexes them (pretrain/function/train.py:139-142 builds its group from
    model.named_parameters(), frozen or not); callers skip the names the engine has no optimizer tensors for."""
    names = list(names)
    if e2e_num_layers is not None:
        vis = ["image_feature_extractor." + n for n in reference_vision_param_names(e2e_num_layers)]
        seen = set(vis)
        rest = reference_param_order([n for n in names if n not in seen])
        return vis + rest
    have = set(names)
    order = []

    def take(n):
        if n in have and n not in order:
            order.append(n)

    # (e2e) the FastRCNN convolution weights are registered first (self.image_feature_extractor is the first sub-module); among
    # themselves they keep the engine's state-dict order (backbone stem .. layer3, then the RoI head)
    for n in names:
        if n.startswith("image_feature_extractor.") and not n.startswith("image_feature_extractor.obj_downsample"):
            take(n)
    for n in ("image_feature_extractor.obj_downsample.1.weight", "image_feature_extractor.obj_downsample.1.bias",
              "object_linguistic_embeddings.weight", "object_mask_visual_embedding.weight", "object_mask_word_embedding.weight",
              "aux_text_visual_embedding.weight", "vlbert.word_embeddings.weight", "vlbert.end_embedding.weight",
              "vlbert.position_embeddings.weight", "vlbert.token_type_embeddings.weight", "vlbert.embedding_LayerNorm.weight",
              "vlbert.embedding_LayerNorm.bias", "vlbert.visual_ln_text.weight", "vlbert.visual_ln_text.bias",
              "vlbert.visual_ln_object.weight", "vlbert.visual_ln_object.bias"):
        take(n)
    layer = 0
    while "vlbert.encoder.layer.%d.attention.self.query.weight" % layer in have:
        p = "vlbert.encoder.layer.%d." % layer
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "attention.output.LayerNorm",
                  "intermediate.dense", "output.dense", "output.LayerNorm"):
            take(p + n + ".weight")
            take(p + n + ".bias")
        layer += 1
    for n in ("vlbert.pooler.dense.weight", "vlbert.pooler.dense.bias", "vlbert.relationsip_head.caption_image_relationship.weight",
              "vlbert.relationsip_head.caption_image_relationship.bias", "vlbert.mlm_head.predictions.bias",
              "vlbert.mlm_head.predictions.transform.dense.weight", "vlbert.mlm_head.predictions.transform.dense.bias",
              "vlbert.mlm_head.predictions.transform.LayerNorm.weight", "vlbert.mlm_head.predictions.transform.LayerNorm.bias",
              "vlbert.mvrc_head.transform.dense.weight", "vlbert.mvrc_head.transform.dense.bias", "vlbert.mvrc_head.region_cls_pred.weight",
              "vlbert.mvrc_head.region_cls_pred.bias"):
        take(n)
    for n in names:          # anything this table does not know keeps its place at the end (never dropped)
        take(n)
    return order


def checkpoint_path(prefix, epoch):
    return "{}-{:04d}.model".format(prefix, epoch)


def _engine_order(eng):
    """Index -> name table of the optimizer state this engine reads and writes (the reference model's named_parameters() order)."""
    e2e = eng.cfg.image_num_layers if eng.vision is not None else None
    order = reference_param_order(eng.P.shapes, e2e_num_layers=e2e)
    missing = [n for n in eng.P.shapes if n not in set(order)]
    if missing:
        raise RuntimeError("checkpoint: engine parameters missing from the reference order: %s" % missing[:4])
    return order


def optimizer_state_dict(eng):
    """The engine's AdamW state in torch.optim's state-dict form (see the module docstring).  A collective with the sharded optimizer."""
    if eng.buckets is not None and eng.buckets.sharded:       # moments live on the owner: gather them like the master weights
        eng.buckets.gather_master(eng.P.m)
        eng.buckets.gather_master(eng.P.v)
    adam = eng.adam.detach().cpu()
    step = int(round(float(adam[5])))
    shapes = eng.P.shapes
    order = _engine_order(eng)
    m, v = eng.P.named(eng.P.m), eng.P.named(eng.P.v)
    vis = eng.vision
    state = {}
    for i, n in enumerate(order):
        if n not in shapes:       # (e2e) a frozen tensor of the image branch: indexed, no state -- as torch's optimizers leave parameters
            continue              # that never received a gradient
        em, ev = m[n].detach().cpu().clone(), v[n].detach().cpu().clone()
        if vis is not None and n in eng._vision_names():       # the engine keeps conv weights as [O, KH, KW, I]; the reference as [O, I, KH, KW]
            em, ev = em.permute(0, 3, 1, 2).contiguous(), ev.permute(0, 3, 1, 2).contiguous()
        state[i] = {"step": step, "exp_avg": em, "exp_avg_sq": ev}
    lr = float(adam[0])

    def exact(dev_value, given):      # the hyper-parameter as the caller gave it (a Python float) while the device copy still is its fp32 image
        return given if abs(float(dev_value) - given) <= 1e-6 * max(abs(given), 1e-30) else float(dev_value)
    hy = eng.hyper
    group = {"lr": exact(adam[0], float(eng.base_lr)) if eng.lr_kind is None else lr,
             "betas": (exact(adam[1], hy["betas"][0]), exact(adam[2], hy["betas"][1])), "eps": exact(adam[3], hy["eps"]),
             "weight_decay": exact(adam[4], hy["weight_decay"]), "correct_bias": True, "params": list(range(len(order)))}
    if eng.lr_kind is not None:      # what the reference's LambdaLR scheduler leaves in the group (train.py:283-285 sets it before resuming)
        group["initial_lr"] = float(eng.base_lr)
    return {"state": state, "param_groups": [group], "param_names": order}


def load_optimizer_state_dict(eng, osd):
    """Inverse of optimizer_state_dict; accepts files written by the reference (no 'param_names': indices follow its named_parameters()
    order) and by this module."""
    shapes = eng.P.shapes
    own = _engine_order(eng)
    order = osd.get("param_names")
    if order is None:
        # a file written by the reference: indices follow named_parameters() -- but only when there is ONE group.  With a non-empty
        # TRAIN.LR_MULT the reference builds several groups (pretrain/function/train.py:139-142) and the indices run in GROUP order;
        # zipping them with named_parameters() would hand moments to the wrong tensors wherever shapes happen to agree.
        if len(osd["param_groups"]) != 1:
            raise ValueError("optimizer state with %d param_groups and no 'param_names' (a reference run with TRAIN.LR_MULT): the index "
                             "-> parameter mapping cannot be recovered; load the weights only (with_optimizer=False)" % len(osd["param_groups"]))
        order = own
    idx = [i for g in osd["param_groups"] for i in g["params"]]
    if len(idx) != len(order):
        raise ValueError("optimizer state has %d parameters, the engine's table %d%s" %
                         (len(idx), len(order), " (e2e: the reference indexes the frozen image-branch tensors too; is IMAGE_NUM_LAYERS the same?  A run with "
                                                "NETWORK.IMAGE_SEMANTIC or NETWORK.ENABLE_CNN_REG_LOSS has object_embed / regularizing_predictor tensors "
                                                "this engine does not model: every later index shifts)" if eng.vision is not None else ""))
    m, v = eng.P.named(eng.P.m), eng.P.named(eng.P.v)
    steps = set()
    visn = eng._vision_names() if eng.vision is not None else ()
    frozen = set(own) - set(shapes)
    for i, n in zip(idx, order):
        if n in frozen:           # (e2e) frozen image-branch tensor: the engine keeps no moments for it; a reference file has no state either
            st_f = osd["state"].get(i)
            if st_f is not None and any(float(torch.as_tensor(st_f[k]).abs().sum()) != 0.0 for k in ("exp_avg", "exp_avg_sq") if k in st_f):
                raise ValueError("optimizer state holds moments for %s, which this engine keeps frozen (IMAGE_FROZEN_BACKBONE_STAGES / "
                                 "IMAGE_FROZEN_BN differ from the run that wrote the file)" % n)
            continue
        if n not in shapes:
            raise KeyError("optimizer state names a parameter the engine does not have: %s" % n)
        st = osd["state"].get(i)
        if st is None:            # a parameter that never received a gradient has no state in torch's optimizers
            m[n].zero_()
            v[n].zero_()
            continue
        em, ev = st["exp_avg"], st["exp_avg_sq"]
        if n in visn:
            em, ev = em.permute(0, 2, 3, 1), ev.permute(0, 2, 3, 1)
        if tuple(em.shape) != tuple(shapes[n]):
            raise ValueError("optimizer state of %s has shape %s, expected %s (is the file's parameter order the reference's?)"
                             % (n, tuple(em.shape), tuple(shapes[n])))
        m[n].copy_(em.to(torch.float32))
        v[n].copy_(ev.to(torch.float32))
        steps.add(int(st["step"]))
    if len(steps) > 1:
        raise ValueError("per-parameter step counts differ (%s): the fused AdamW keeps one counter" % sorted(steps))
    g = osd["param_groups"][0]
    eng.adam[1:5].copy_(torch.tensor([g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]], dtype=torch.float32))
    eng.hyper = dict(betas=(float(g["betas"][0]), float(g["betas"][1])), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]))
    eng.adam[5:6].fill_(float(steps.pop()) if steps else 0.0)
    if eng.lr_kind is None:
        eng.adam[0:1].fill_(float(g["lr"]))       # (with a device-side schedule the lr is recomputed from the step counter each step)


def save_checkpoint(eng, prefix, epoch, rank=0, extra=None):
    """`Checkpoint(prefix, frequent)(epoch, net, optimizer, ...)`: writes `{prefix}-{epoch:04d}.model` on rank 0.  Call on EVERY rank."""
    sd = eng.state_dict()                 # (gathers the master slices first with the sharded optimizer)
    osd = optimizer_state_dict(eng)
    path = checkpoint_path(prefix, epoch)
    if rank == 0:
        ck = OrderedDict(state_dict=OrderedDict((k, t.detach().cpu()) for k, t in sd.items()), optimizer=osd)
        if extra:
            ck.update(extra)
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        tmp = path + ".tmp"
        torch.save(ck, tmp)
        os.replace(tmp, path)             # never leave a half-written epoch file for AUTO_RESUME to find
    return path


def load_checkpoint(eng, path, with_optimizer=True):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    # smart_load_model_state_dict (common/utils/load.py:84-104): tolerate the 'module.' prefix of a DistributedDataParallel wrapper
    sd = OrderedDict((k[len("module."):] if k.startswith("module.") else k, t) for k, t in sd.items())
    eng.load_state_dict({k: t.to(eng.dev) for k, t in sd.items()})
    if with_optimizer and "optimizer" in ck:
        load_optimizer_state_dict(eng, ck["optimizer"])
    eng.sync_weights()
    return ck


def smart_resume(eng, prefix, begin_epoch, end_epoch, resume=False, auto_resume=True, log=print):
    """common/utils/load.py:20-54 -> the epoch to begin with.  RESUME: `{prefix}-{begin_epoch-1:04d}.model` must exist; AUTO_RESUME: the
    newest `{prefix}-{e-1:04d}.model` for e in (begin_epoch, end_epoch]."""
    if resume:
        path = checkpoint_path(prefix, begin_epoch - 1)
        log("continue training from %d (%s)" % (begin_epoch, path))
        load_checkpoint(eng, path)
        return begin_epoch
    if auto_resume:
        for epoch in range(end_epoch, begin_epoch, -1):
            path = checkpoint_path(prefix, epoch - 1)
            if os.path.exists(path):
                load_checkpoint(eng, path)
                log("Auto continue training from {0}".format(path))
                return epoch
    return begin_epoch


def partial_pretrain_state_dict(pretrain_state_dict, prefix_changes=(), load_rel_head=False, segmb_init=False):
    """The `NETWORK.PARTIAL_PRETRAIN` preparation of the fine-tuning entry points (vqa/function/train.py:198-213, vcr/function/train.py:
    200-230): every key is renamed by the FIRST matching 'old_prefix->new_prefix' rule of PARTIAL_PRETRAIN_PREFIX_CHANGES (keys without a
    matching rule stay); VCR extras: LOAD_REL_HEAD seeds the 1-logit answer classifier with (row 1 - row 0) of the pre-trained
    caption-image relationship head, PARTIAL_PRETRAIN_SEGMB_INIT copies the segment embedding of type 0 over type 1 (VCR uses types 0 / 1
    for question / answer where pre-training only ever saw type 0 for text).  Returns a new dict; tensors are not copied unless changed."""
    rules = [tuple(r.split("->")) if isinstance(r, str) else tuple(r) for r in prefix_changes]
    out = {}
    for key, value in pretrain_state_dict.items():
        for old, new in rules:
            if key.startswith(old):
                key = new + key[len(old):]
                break
        out[key] = value
    rel_w = "module.vlbert.relationsip_head.caption_image_relationship.weight"
    rel_b = "module.vlbert.relationsip_head.caption_image_relationship.bias"
    if load_rel_head and rel_w in pretrain_state_dict:
        out["module.final_mlp.1.weight"] = pretrain_state_dict[rel_w][1:2].float() - pretrain_state_dict[rel_w][0:1].float()
        out["module.final_mlp.1.bias"] = pretrain_state_dict[rel_b][1:2].float() - pretrain_state_dict[rel_b][0:1].float()
    if segmb_init:
        k = "module.vlbert._module.token_type_embeddings.weight"
        if k not in out:
            raise KeyError("PARTIAL_PRETRAIN_SEGMB_INIT: %s is not among the renamed keys (the reference indexes it unconditionally)" % k)
        t = out[k].float().clone()
        t[1] = t[0]
        out[k] = t
    return out


def drop_shape_mismatches(state_dict, own, log=print, max_fraction=0.5):
    """Tensors of `state_dict` whose key (as it is, or with 'module.' added / removed) names a tensor of `own` with ANOTHER shape are
    taken out before smart_partial_load (an answer classifier for another vocabulary; the reference would raise in load_state_dict).
    Every dropped key is logged with both shapes (round-4 ADVICE: they used to vanish from both the 'non matched' and the 'non
    pretrain' report), and a file in which more than `max_fraction` of the matching keys have the wrong shape is refused: that is a
    checkpoint of another model (base weights against a large model), not a head to re-initialise.  -> (kept dict, [(key, shape in
    file, shape in model)])."""
    kept, dropped, matched = {}, [], 0
    for k, v in state_dict.items():
        alt = k[len("module."):] if k.startswith("module.") else "module." + k
        kk = k if k in own else (alt if alt in own else None)
        if kk is not None:
            matched += 1
            if tuple(own[kk].shape) != tuple(v.shape):
                dropped.append((k, tuple(v.shape), tuple(own[kk].shape)))
                continue
        kept[k] = v
    if dropped:
        log("[Partial Load] dropped for shape mismatch (left at the model's initialisation): {}".format(
            ["%s: file %s vs model %s" % d for d in dropped]))
        if matched and len(dropped) > max_fraction * matched:
            raise ValueError("PARTIAL_PRETRAIN: %d of the %d tensors that match this model by name have another shape (first: %s: file %s "
                             "vs model %s) -- this is a checkpoint of a different architecture" % ((len(dropped), matched) + dropped[0]))
    return kept, dropped


def smart_partial_load(model, state_dict, log=print):
    """smart_partial_load_model_state_dict (common/utils/load.py:57-81): take every tensor whose key -- as it is, or with a 'module.' prefix
    added / removed -- names a tensor of the model, leave the rest of the model as it is; report what matched and what did not.
    `model`: anything with state_dict() / load_state_dict() (the module mirrors, PretrainEngine).  Returns (loaded keys, unmatched keys)."""
    own = model.state_dict()
    take, unmatched = {}, []
    for key, value in state_dict.items():
        if key not in own:
            key = key[len("module."):] if key.startswith("module.") else "module." + key
        if key in own:
            take[key] = value
        else:
            unmatched.append(key)
    untouched = [k for k in own if k not in take]
    log("[Partial Load] partial load state dict of keys: {}".format(list(take)))
    log("[Partial Load] non matched keys: {}".format(unmatched))
    log("[Partial Load] non pretrain keys: {}".format(untouched))
    merged = dict(own)
    merged.update(take)
    model.load_state_dict(merged)
    return list(take), unmatched
