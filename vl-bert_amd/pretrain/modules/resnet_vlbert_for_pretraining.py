"""Drop-in `ResNetVLBERTForPretraining` (pretrain/modules/resnet_vlbert_for_pretraining.py:14-216) backed by
the HIP engine: same constructor argument (the `config` tree of pretrain/function/config.py), same
`forward(image, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels) -> (outputs, loss)`,
same parameter names / shapes (state-dict contract), so the reference's training loop
(common/trainer.py:101-189: `outputs, loss = net(*batch); loss.backward(); clip_grad_norm_; optimizer.step()`)
and its checkpoints work unchanged.

How it maps onto the engine:
  * every nn.Parameter is a VIEW into the engine's flat fp32 master buffer (and `.grad` a view into the flat
    gradient buffer), so torch optimizers / clip_grad_norm_ / state_dict operate on the engine's storage;
  * `forward` runs the HIP forward and returns a scalar loss with a custom autograd node; `loss.backward()`
    runs the hand-scheduled HIP backward and leaves the gradients in `param.grad`;
  * engines are built per (batch, text length, regions) shape on first use and share the parameter storage.
Supported configuration = the north-star one (cfgs/pretrain/base_prec_withouttextonly_4x16G_fp32.yaml):
IMAGE_FEAT_PRECOMPUTED, WITH_MLM_LOSS, WITH_MVRC_LOSS, visual_ln; the pooler and the relationship head / loss
(WITH_REL_LOSS, off in every shipped pretrain cfg) are available too.
IMAGE_FEAT_PRECOMPUTED: false (cfgs/pretrain/base_e2e_16x16G_fp16.yaml) runs the ResNet trunk / ROIAlign / layer4 head of
vision.py in front: `forward(image, boxes[B,R,4], ...)`.  The trainable convolution weights are Parameters that view the
flat buffer in the engine's [O,KH,KW,I] layout (state_dict / load_state_dict convert from / to the reference's [O,I,KH,KW]);
BatchNorm tensors and the frozen stages are buffers under the reference's names.  One image size per module (static shapes).
Anything else raises NotImplementedError (nothing silently falls back to PyTorch eager).
"""
from collections import OrderedDict

import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import engine as _engine
from ... import ops
from ...common.visual_linguistic_bert import bucketed, lru_get


def _get(obj, name, default=None):
    return getattr(obj, name, default) if not isinstance(obj, dict) else obj.get(name, default)


class _HipLoss(torch.autograd.Function):
    """Scalar loss whose backward is the engine's explicit backward pass."""

    @staticmethod
    def forward(ctx, anchor, module, eng):
        ctx.module, ctx.eng = module, eng
        ctx.seed_snap = eng.seed_snapshot()       # the backward regenerates this forward's dropout masks (engine.mirror_pre_forward)
        return (eng.losses[0] + eng.losses[1] + eng.losses[2] + eng.losses[3]).clone()

    @staticmethod
    def backward(ctx, grad_out):
        module, eng = ctx.module, ctx.eng
        g = float(grad_out)            # compat path only (the reference's own loop syncs every step, trainer.py:159-171)
        if g != 1.0:                    # re-derive d(logits) with the upstream scale (grad accumulation / loss scaling)
            eng._losses_fwd_bwd(g, keep=False)
        module._prepare_grads()
        with eng.seed_guard(ctx.seed_snap):
            eng.backward(train=module.training)
        return None, None, None


class ResNetVLBERTForPretraining(nn.Module):
    MULTITASK = False

    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        net = _get(config, "NETWORK")
        vl = _get(net, "VLBERT")
        self.e2e = not _get(net, "IMAGE_FEAT_PRECOMPUTED", False)
        if self.e2e:
            if not (_get(net, "IMAGE_FROZEN_BN", True) and _get(net, "IMAGE_STRIDE_IN_1x1", True) and _get(net, "IMAGE_C5_DILATED", True)):
                raise NotImplementedError("e2e path needs IMAGE_FROZEN_BN, IMAGE_STRIDE_IN_1x1 and IMAGE_C5_DILATED (the shipped e2e cfgs)")
            if _get(net, "OUTPUT_CONV5", False):
                raise NotImplementedError("OUTPUT_CONV5 is not supported")
        self.with_rel = bool(_get(net, "WITH_REL_LOSS", False))
        if self.with_rel and not _get(vl, "with_pooler", False):
            raise ValueError("WITH_REL_LOSS needs VLBERT.with_pooler (the relationship head reads the pooled output)")
        if self.with_rel and self.MULTITASK:
            raise NotImplementedError("relationship loss in the multitask wrapper is not supported")
        if not (_get(net, "WITH_MLM_LOSS", True) and _get(net, "WITH_MVRC_LOSS", True) and _get(vl, "visual_ln", True)):
            raise NotImplementedError("accelerated path needs WITH_MLM_LOSS, WITH_MVRC_LOSS and visual_ln")
        if _get(net, "IMAGE_SEMANTIC", False) or _get(vl, "word_embedding_frozen", False) or _get(vl, "pos_embedding_frozen", False):
            raise NotImplementedError("IMAGE_SEMANTIC / frozen embeddings are not supported")
        if _get(vl, "visual_size", _get(vl, "hidden_size")) != _get(vl, "hidden_size"):
            raise NotImplementedError("visual_size != hidden_size (visual_1x1 projections) is not supported")
        self.cfg = _engine.ModelConfig(
            hidden_size=_get(vl, "hidden_size"), num_hidden_layers=_get(vl, "num_hidden_layers"),
            num_attention_heads=_get(vl, "num_attention_heads"), intermediate_size=_get(vl, "intermediate_size"),
            vocab_size=_get(vl, "vocab_size", 30522), max_position_embeddings=_get(vl, "max_position_embeddings", 512),
            type_vocab_size=_get(vl, "type_vocab_size", 3), visual_region_classes=_get(vl, "visual_region_classes", 1601),
            hidden_dropout_prob=_get(vl, "hidden_dropout_prob", 0.1),
            attention_probs_dropout_prob=_get(vl, "attention_probs_dropout_prob", 0.1), multitask=self.MULTITASK,
            with_pooler=bool(_get(vl, "with_pooler", False)), with_rel_loss=self.with_rel, e2e=self.e2e,
            image_num_layers=int(_get(net, "IMAGE_NUM_LAYERS", 101)),
            image_frozen_stages=tuple(_get(net, "IMAGE_FROZEN_BACKBONE_STAGES", (1, 2))))
        self.cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("ResNetVLBERTForPretraining (HIP) needs an MI355X: there is no CPU fallback")
        self.device_ = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self._engines = OrderedDict()
        self.flat = _engine.FlatParams(self.cfg, self.device_)
        self._init_scale = (_get(vl, "visual_scale_text_init", 0.0), _get(vl, "visual_scale_object_init", 0.0),
                            _get(vl, "initializer_range", 0.02))
        # parameters: views of the flat master buffer, registered under the reference's names
        self._pnames = {}
        for name, t in self.flat.named(self.flat.master).items():
            self._register(name, nn.Parameter(t, requires_grad=True))
        self._vision_buffers = {}
        if self.e2e:      # BatchNorm tensors + frozen-stage weights: buffers in the reference's layout, handed to each engine's VisionStack
            from ... import vision as _vision
            for key, O, I, k, bn, tr in _vision.conv_table(self.cfg.image_num_layers, self.cfg.image_frozen_stages):
                if not tr:
                    self._register_buffer(_vision.PREFIX + key + ".weight", torch.zeros((O, I, k, k), device=self.device_))
                for suffix, fill in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
                    self._register_buffer(_vision.PREFIX + bn + "." + suffix, torch.full((O,), fill, device=self.device_))
            self._vision_names = set(_vision.vision_param_layout(self.cfg.image_num_layers, self.cfg.image_frozen_stages))
            self._image_size = None
        self.init_weight()
        # language-only BERT initialisation of the vlbert.* parameters (:28-47): BERT_PRETRAINED-<epoch>.model, else
        # BERT_MODEL_NAME/pytorch_model.bin; `from_scratch` skips it.  (The reference loads inside the VLBERT constructor and then
        # re-draws only the wrapper-level embeddings in init_weight, :55-63 -- same end state.)
        from ...common import language_pretrained as _lp
        path = _lp.resolve_path(net)
        if path is None:
            print("Warning: no pretrained language model found, training from scratch!!!", file=sys.stderr)   # (the reference prints to stdout; bench.py owns stdout)
        elif not _get(vl, "from_scratch", False):
            self.load_language_pretrained_model(path)

    def load_language_pretrained_model(self, language_pretrained_model_path):
        """VisualLinguisticBertForPretraining.load_language_pretrained_model (common/visual_linguistic_bert.py:382-469) on this
        wrapper's `vlbert.*` parameters (key mapping: common/language_pretrained.py)."""
        from ...common import language_pretrained as _lp
        sd = torch.load(language_pretrained_model_path, map_location=lambda storage, loc: storage)
        own = {n[len("vlbert."):]: p for n, p in self._pnames.items() if n.startswith("vlbert.")}
        assign, unexpected = _lp.plan(sd, list(own) + ["mlm_head.predictions.decoder.weight"], self.cfg.with_pooler, pretraining=True,
                                      with_rel_head=self.with_rel, with_mlm_head=True)
        if len(unexpected) > 0:
            print("Warnings: Unexpected keys: {}.".format(unexpected))
        _lp.apply(assign, own)
        torch.autograd.graph.increment_version(self.flat.master)

    # -- parameter plumbing -----------------------------------------------------------------------
    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], param)
        self._pnames[dotted] = param

    def _register_buffer(self, dotted, t):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_buffer(parts[-1], t)
        self._vision_buffers[dotted] = t

    def init_weight(self):
        """BaseModel.init_weights / ResNetVLBERTForPretraining.init_weight statistics
        (common/visual_linguistic_bert.py:14-25,330-332; resnet_vlbert_for_pretraining.py:55-63)."""
        vt, vo, std = self._init_scale
        with torch.no_grad():
            for name, p in self._pnames.items():
                if name.endswith("visual_ln_text.weight"):
                    p.fill_(vt)
                elif name.endswith("visual_ln_object.weight"):
                    p.fill_(vo)
                elif "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith(".bias") or name == "object_mask_visual_embedding.weight":
                    p.zero_()
                elif self.e2e and name in self._vision_names:      # kaiming fan_out (resnet.py:153-155); a checkpoint normally follows
                    p.normal_(0.0, (2.0 / (p.shape[0] * p.shape[1] * p.shape[2])) ** 0.5)
                else:
                    p.normal_(0.0, std)
            for name, t in self._vision_buffers.items():
                if name.endswith(".weight") and t.dim() == 4:
                    t.normal_(0.0, (2.0 / (t.shape[0] * t.shape[2] * t.shape[3])) ** 0.5)
        self._buffers_version = getattr(self, "_buffers_version", 0) + 1

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
        sd[prefix + _engine.TIED_DECODER_KEY] = sd[prefix + "vlbert.word_embeddings.weight"]   # tied (modeling.py:463-466)
        if self.e2e:      # trainable convolutions: engine layout [O,KH,KW,I] -> reference layout [O,I,KH,KW]
            for name in self._vision_names:
                sd[prefix + name] = sd[prefix + name].permute(0, 3, 1, 2).contiguous()
        return sd

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        state_dict.pop(_engine.TIED_DECODER_KEY, None)
        if self.e2e:
            for name in self._vision_names:
                if name in state_dict:
                    state_dict[name] = state_dict[name].permute(0, 2, 3, 1).contiguous()
            state_dict = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
            self._buffers_version += 1
        return super().load_state_dict(state_dict, strict=strict)

    def _prepare_grads(self):
        """torch zero_grad(set_to_none=True) drops `.grad`; re-attach the flat views (zeroed) so accumulation
        semantics match autograd: first backward after zero_grad starts from zero, later ones accumulate."""
        fresh = any(p.grad is None for p in self._pnames.values())
        if fresh:
            self.flat.grad.zero_()
            for name, t in self.flat.named(self.flat.grad).items():
                self._pnames[name].grad = t

    def _engine_for(self, B, T, R, B_aux=0, image_size=None):
        eng = lru_get(self._engines, (B, T, R, B_aux, image_size),
                      lambda: _engine.PretrainEngine(self.cfg, B, T, R, device=str(self.device_), keep_logits=True, flat=self.flat,
                                                     B_aux=B_aux, image_size=image_size, seed=ops.rank_seed(1234) // 2))   # per-rank dropout stream
        if self.e2e and getattr(eng, "_buffers_version", None) != self._buffers_version:
            eng.vision.load_state_dict(self._vision_buffers, strict=False)      # BatchNorm tensors + frozen stages (trainables live in flat)
            eng._buffers_version = self._buffers_version
            eng._synced_version = None
        version = self.flat.master._version                       # bumped by any in-place update of a parameter view
        if getattr(eng, "_synced_version", None) != version:      # optimizer step / load_state_dict: refresh bf16 + W^T copies
            eng.sync_weights()
            eng._synced_version = version
        eng._weights_dirty = False
        return eng

    def _padded_logits(self, eng, Bt):
        """fp32 logits re-padded like the reference (:165-167, :195-200) -- needs max_len (host sync)."""
        V, C = self.cfg.vocab_size, self.cfg.visual_region_classes
        mlm_logits = torch.empty((Bt, eng.T, V), dtype=torch.float32, device=self.device_)
        ops.cast_bf16_f32(eng.mlm_logits_copy[:, :V].contiguous(), mlm_logits)
        mvrc = torch.empty((eng.B, eng.R, C), dtype=torch.float32, device=self.device_)
        ops.cast_bf16_f32(eng.mvrc_logits_copy[:, :C].contiguous(), mvrc)
        mvrc[:, int(eng.lay["nobj"][:eng.B].max()):] = -10000.0
        # the reference's encoder runs on the batch trimmed to max(text_len + n_obj) + 1 packed positions
        # (common/visual_linguistic_bert.py:202,152-153), so its text logits stop there and the wrapper pads the rest with -10000
        seq_max = int((eng.lay["text_len"][:Bt] + eng.lay["nobj"][:Bt]).max()) + 1
        if seq_max < eng.T:
            mlm_logits[:, seq_max:] = -10000.0
        return mlm_logits, mvrc

    @staticmethod
    def _bucket_regions(R, Rp, boxes, mvrc_ops, mvrc_labels):
        """R box slots -> Rp: padded rows carry the collator's markers (boxes -2, mvrc_ops 0, labels 0: pretrain/data/collate_batch.py:39-55).
        (The text side needs no copy: the engine accepts text / labels narrower than its T and pads them itself.)"""
        if Rp == R:
            return boxes, mvrc_ops, mvrc_labels
        return F.pad(boxes, (0, 0, 0, Rp - R), value=-2.0), F.pad(mvrc_ops, (0, Rp - R)), F.pad(mvrc_labels, (0, 0, 0, Rp - R))

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, image, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels):
        if (image is not None) != self.e2e:
            raise NotImplementedError("IMAGE_FEAT_PRECOMPUTED configuration takes image=None, the e2e configuration an image batch")
        B, R = boxes.shape[0], boxes.shape[1]
        T = text.shape[1]
        Tp, Rp = bucketed(T, R)        # (T, R) follow each batch's longest caption / largest box count: engines exist per bucket
        boxes_p, ops_p, labels_p = self._bucket_regions(R, Rp, boxes, mvrc_ops, mvrc_labels)
        if self.e2e:
            eng = self._engine_for(B, Tp, Rp, image_size=(int(image.shape[2]), int(image.shape[3])))
            eng.set_batch(boxes_p, im_info, text, relationship_label, mlm_labels, ops_p, labels_p, image=image.float())
        else:
            eng = self._engine_for(B, Tp, Rp)
            eng.set_batch(boxes_p, im_info, text, relationship_label, mlm_labels, ops_p, labels_p)
        eng.mirror_pre_forward(self.training)      # fresh dropout masks for this forward AND its backward
        eng.forward(train=self.training)
        loss = _HipLoss.apply(self._pnames["vlbert.word_embeddings.weight"], self, eng)
        mlm_logits, mvrc = self._padded_logits(eng, B)
        mlm_logits, mvrc = mlm_logits[:, :T], mvrc[:, :R]
        outputs = {
            "relationship_logits": eng.rel_logits_copy[:, :2].float() if self.with_rel else None,
            "relationship_label": relationship_label if self.with_rel else None,
            "mlm_logits": mlm_logits, "mlm_label": mlm_labels,
            "mvrc_logits": mvrc, "mvrc_label": mvrc_labels,
            "relationship_loss": eng.losses[3].clone() if self.with_rel else im_info.new_zeros(()),
            "mlm_loss": eng.losses[0].clone(), "mvrc_loss": eng.losses[1].clone(),
        }
        return outputs, loss


class ResNetVLBERTForPretrainingMultitask(ResNetVLBERTForPretraining):
    """Drop-in for pretrain/modules/resnet_vlbert_for_pretraining_multitask.py:14-290: the image-caption batch plus
    text-only auxiliary batches (`*aux` = text_0, mlm_labels_0, text_1, ...) share ONE encoder pass; the aux samples have no
    regions and see the learned `aux_text_visual_embedding` as their visual half.  Loss = mlm_wvc + mlm_aux + mvrc."""
    MULTITASK = True

    def forward(self, image, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels, *aux):
        if (image is not None) != self.e2e:
            raise NotImplementedError("IMAGE_FEAT_PRECOMPUTED configuration takes image=None, the e2e configuration an image batch")
        if len(aux) == 0 or len(aux) % 2:
            raise ValueError("aux must be (text, mlm_labels) pairs")
        texts, labels = aux[0::2], aux[1::2]
        Ba, Ta = sum(t.shape[0] for t in texts), max(t.shape[1] for t in texts)
        aux_text = texts[0].new_zeros((Ba, Ta))
        aux_labels = labels[0].new_full((Ba, Ta), -1)
        cur = 0
        for t, l in zip(texts, labels):                          # concat the corpora, right-padded (:106-120)
            aux_text[cur:cur + t.shape[0], :t.shape[1]] = t
            aux_labels[cur:cur + t.shape[0], :t.shape[1]] = l
            cur += t.shape[0]
        B, R = boxes.shape[0], boxes.shape[1]
        T = max(text.shape[1], Ta)
        Tp, Rp = bucketed(T, R)
        boxes_p, ops_p, labels_p = self._bucket_regions(R, Rp, boxes, mvrc_ops, mvrc_labels)
        if self.e2e:      # only the caption samples carry an image (cfgs/pretrain/base_e2e_16x16G_fp16.yaml)
            eng = self._engine_for(B, Tp, Rp, Ba, image_size=(int(image.shape[2]), int(image.shape[3])))
            eng.set_batch(boxes_p, im_info, text, relationship_label, mlm_labels, ops_p, labels_p, aux_text, aux_labels,
                          image=image.float())
        else:
            eng = self._engine_for(B, Tp, Rp, Ba)
            eng.set_batch(boxes_p, im_info, text, relationship_label, mlm_labels, ops_p, labels_p, aux_text, aux_labels)
        eng.mirror_pre_forward(self.training)
        eng.forward(train=self.training)
        loss = _HipLoss.apply(self._pnames["vlbert.word_embeddings.weight"], self, eng)
        mlm_logits, mvrc = self._padded_logits(eng, B + Ba)
        mlm_logits, mvrc = mlm_logits[:, :T], mvrc[:, :R]
        lab = eng.in_mlm_labels[:, :T]
        outputs = {
            "relationship_logits": None, "relationship_label": None,
            "mlm_logits_wvc": mlm_logits[:B], "mlm_label_wvc": lab[:B].clone(),
            "mlm_logits_aux": mlm_logits[B:], "mlm_label_aux": lab[B:].clone(),
            "mvrc_logits": mvrc, "mvrc_label": mvrc_labels,
            "relationship_loss": im_info.new_zeros(()), "mlm_loss_wvc": eng.losses[0].clone(),
            "mlm_loss_aux": eng.losses[2].clone(), "mvrc_loss": eng.losses[1].clone(),
        }
        return outputs, loss
