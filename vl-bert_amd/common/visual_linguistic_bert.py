"""Drop-in `VisualLinguisticBert` / `VisualLinguisticBertForPretraining` (common/visual_linguistic_bert.py:31-241,
:335-380) backed by the HIP engine in module-API mode: the reference's constructor argument (the `NETWORK.VLBERT`
config node), forward signatures and state-dict keys, so the callers that build their own text-visual / object
embeddings (pretrain/modules/resnet_vlbert_for_pretraining.py:42-48, vqa/modules/resnet_vlbert_for_vqa.py:49-50,
vcr/modules/resnet_vlbert_for_vcr.py:60) can construct it unchanged.

    VisualLinguisticBert.forward(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask,
                                 object_vl_embeddings, object_mask, output_all_encoded_layers=False,
                                 output_text_and_object_separately=True) -> (text_out, object_out, pooled)
                         ... output_text_and_object_separately=False   -> (sequence_output [B, max_len, H], pooled)
    VisualLinguisticBertForPretraining.forward(same six tensors) -> (relationship_logits=None, mlm_logits, mvrc_logits)

Both are differentiable w.r.t. the parameters and the two embedding inputs (custom autograd node -> the engine's explicit
backward).  Supported: visual_size == hidden_size, visual_ln, optional pooler / relationship head, last layer only,
text and objects returned separately; anything else raises NotImplementedError (no silent eager fallback).
Parameters are views of the engine's flat fp32 master buffer, `.grad` views of its flat gradient buffer.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine as _engine
from .. import ops

_PREFIX = "vlbert."
_MAX_S = 256      # longest packed sequence the attention kernels cover (attention.hip: ATT_SP_MAX)


def shape_buckets():
    """(text multiple, region multiple, max cached engines).  The reference's collators pad every batch to ITS longest question and
    ITS largest box count (vqa/data/collate_batch.py:20-21, pretrain/data/collate_batch.py), so (T, R) changes from batch to batch; an
    engine owns static buffers for one (B, T, R), hence the mirrors round T and R up to a multiple (extra positions are masked out,
    outputs are sliced back) and keep only the most recently used engines.  VLB_MIRROR_BUCKETS="8,4,8" (default); "1,1,N" = exact."""
    v = os.environ.get("VLB_MIRROR_BUCKETS", "8,4,8").split(",")
    return max(1, int(v[0])), max(1, int(v[1])), max(1, int(v[2]))


def bucketed(T, R):
    bt, br, _ = shape_buckets()
    Tp, Rp = (T + bt - 1) // bt * bt, (R + br - 1) // br * br
    if Tp + Rp + 1 > _MAX_S:          # rounding must not push a sequence that fits over the kernels' limit
        Tp, Rp = T, R
    return Tp, Rp


def lru_get(cache, key, make):
    """cache: OrderedDict; newest last; at most shape_buckets()[2] entries (an evicted engine is freed once no autograd graph holds it)"""
    if key in cache:
        cache.move_to_end(key)
        return cache[key]
    cache[key] = make()
    while len(cache) > shape_buckets()[2]:
        cache.popitem(last=False)
    return cache[key]


def _get(obj, name, default=None):
    return getattr(obj, name, default) if not isinstance(obj, dict) else obj.get(name, default)


class _CoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_vis, obj_vl, anchor, module, eng):
        ctx.module, ctx.eng = module, eng
        eng.mirror_pre_forward(module.training)   # fresh dropout masks for this forward and ITS backward (seed advanced before, not after)
        mlm, mvrc, text_out, obj_out, pooled, rel = eng.forward_core(train=module.training)
        ctx.seed_snap = eng.seed_snapshot()
        ctx.in_dtypes = (text_vis.dtype, obj_vl.dtype)
        third = (rel if eng.with_heads else pooled)
        third = third.float() if third is not None else text_vis.new_zeros(())     # placeholder keeps the arity fixed
        ctx.has_third = (rel if eng.with_heads else pooled) is not None
        if eng.with_heads:
            return mlm.float(), mvrc.float(), third
        if eng.seq_out:      # trimmed to the batch's longest packed sequence like the reference (:202; one host sync)
            n = int((eng.lay["text_len"] + eng.lay["nobj"]).max()) + 1
            return eng.sequence_output()[:, :n].float(), text_vis.new_zeros(()), third
        return text_out.float(), obj_out.float(), third

    @staticmethod
    def backward(ctx, g0, g1, g2):
        module, eng = ctx.module, ctx.eng
        module._prepare_grads()
        g2 = g2 if ctx.has_third else None
        with eng.seed_guard(ctx.seed_snap):
            if eng.with_heads:
                d_tv, d_ovl = eng.backward_core(g0, g1, g2, train=module.training)
            elif eng.seq_out:
                d_tv, d_ovl = eng.backward_core_sequence(g0, g2, train=module.training)
            else:
                d_tv, d_ovl = eng.backward_core_hidden(g0, g1, g2, train=module.training)
        return d_tv.to(ctx.in_dtypes[0]), d_ovl.to(ctx.in_dtypes[1]), None, None, None


class VisualLinguisticBert(nn.Module):
    WITH_HEADS = False

    def __init__(self, config, language_pretrained_model_path=None, device=None):
        super().__init__()
        if _get(config, "visual_size", _get(config, "hidden_size")) != _get(config, "hidden_size"):
            raise NotImplementedError("visual_size != hidden_size (visual_1x1 projections) is not supported")
        if not _get(config, "visual_ln", True):
            raise NotImplementedError("accelerated path needs visual_ln")
        self.with_pooler = bool(_get(config, "with_pooler", False))
        if _get(config, "word_embedding_frozen", False) or _get(config, "pos_embedding_frozen", False):
            raise NotImplementedError("frozen embeddings are not supported")
        self.config = config
        # fp32 compute mode of the encoder (the reference's TRAIN.FP16: false configurations; ../encoder_f32.py): `fp32_encoder` in the
        # VLBERT config node, or VLB_ENCODER_FP32=1 -- run it on the fp16 build (VLB_PRECISION=f16) for the 1e-3 class
        import os as _os
        self.fp32_encoder = bool(_get(config, "fp32_encoder", False)) or _os.environ.get("VLB_ENCODER_FP32", "0") == "1"
        self.cfg = _engine.ModelConfig(
            hidden_size=_get(config, "hidden_size"), num_hidden_layers=_get(config, "num_hidden_layers"),
            num_attention_heads=_get(config, "num_attention_heads"), intermediate_size=_get(config, "intermediate_size"),
            vocab_size=_get(config, "vocab_size", 30522), max_position_embeddings=_get(config, "max_position_embeddings", 512),
            type_vocab_size=_get(config, "type_vocab_size", 3), visual_region_classes=_get(config, "visual_region_classes", 1601),
            hidden_dropout_prob=_get(config, "hidden_dropout_prob", 0.1),
            attention_probs_dropout_prob=_get(config, "attention_probs_dropout_prob", 0.1), with_pooler=self.with_pooler,
            with_rel_loss=bool(getattr(self, "with_rel_head", False)))
        self.cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("VisualLinguisticBert (HIP) needs an MI355X: there is no CPU fallback")
        self.device_ = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.flat = _engine.FlatParams(self.cfg, self.device_)
        self._engines = OrderedDict()
        self._pnames = {}
        heads = ("mlm_head.", "mvrc_head.", "relationsip_head.")
        for name, t in self.flat.named(self.flat.master).items():
            if not name.startswith(_PREFIX):
                continue                                       # wrapper-level parameters are not part of this module
            short = name[len(_PREFIX):]
            if not self.WITH_HEADS and short.startswith(heads):
                continue
            self._register(short, nn.Parameter(t, requires_grad=True))
        vt, vo, std = _get(config, "visual_scale_text_init", 0.0), _get(config, "visual_scale_object_init", 0.0), \
            _get(config, "initializer_range", 0.02)
        with torch.no_grad():                                  # BaseModel.init_weights + :330-332
            for name, p in self._pnames.items():
                if name.endswith("visual_ln_text.weight"):
                    p.fill_(vt)
                elif name.endswith("visual_ln_object.weight"):
                    p.fill_(vo)
                elif "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, std)
        if language_pretrained_model_path is not None:         # (:76-78; the pretraining class loads after its heads exist, :335-336)
            self.load_language_pretrained_model(language_pretrained_model_path)

    def load_language_pretrained_model(self, language_pretrained_model_path):
        """Initialise embeddings / encoder / pooler (and, for the pretraining class, the MLM and relationship heads) from a language-only
        BERT or RoBERTa checkpoint (:243-309, :382-469): the key mapping lives in language_pretrained.py (checked against the
        reference's method on CPU); here the assignments land in the flat fp32 master buffer."""
        from . import language_pretrained as lp
        pretrained_state_dict = torch.load(language_pretrained_model_path, map_location=lambda storage, loc: storage)
        own = list(self._pnames)
        if self.WITH_HEADS:
            own.append("mlm_head.predictions.decoder.weight")                    # tied to word_embeddings.weight
        assign, unexpected = lp.plan(pretrained_state_dict, own, self.with_pooler, pretraining=self.WITH_HEADS,
                                     with_rel_head=bool(getattr(self, "with_rel_head", False)), with_mlm_head=True)
        if len(unexpected) > 0:
            print("Warnings: Unexpected keys: {}.".format(unexpected))
        lp.apply(assign, self._pnames)
        torch.autograd.graph.increment_version(self.flat.master)                # cached engines refresh their 16-bit working copies

    def _register(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for q in parts[:-1]:
            if not hasattr(mod, q):
                mod.add_module(q, nn.Module())
            mod = getattr(mod, q)
        mod.register_parameter(parts[-1], param)
        self._pnames[dotted] = param

    def _prepare_grads(self):
        if any(p.grad is None for p in self._pnames.values()):
            self.flat.grad.zero_()
            named = self.flat.named(self.flat.grad)
            for name, p in self._pnames.items():
                p.grad = named[_PREFIX + name]

    def _engine_for(self, B, T, R, sequence=False):
        eng = lru_get(self._engines, (B, T, R, sequence),
                      lambda: _engine.PretrainEngine(self.cfg, B, T, R, device=str(self.device_), flat=self.flat, core=True,
                                                     core_heads=self.WITH_HEADS, core_sequence=sequence,
                                                     seed=ops.rank_seed(1234) // 2,         # per-rank dropout stream
                                                     encoder_fp32=self.fp32_encoder))
        eng._dp_hook = getattr(self, "_dp_hook", None)          # parallel.DistributedDataParallel: gradient buckets leave from backward
        version = self.flat.master._version
        if getattr(eng, "_synced_version", None) != version:
            eng.sync_weights()
            eng._synced_version = version
        eng._weights_dirty = False
        return eng

    def _run(self, text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask,
             sequence=False):
        B, T = text_input_ids.shape
        R = object_vl_embeddings.shape[1]
        Tp, Rp = bucketed(T, R)
        if Tp != T:       # masked-out extra text positions (autograd slices the gradient of the padded embeddings back)
            text_input_ids, text_token_type_ids = F.pad(text_input_ids, (0, Tp - T)), F.pad(text_token_type_ids, (0, Tp - T))
            text_visual_embeddings = F.pad(text_visual_embeddings, (0, 0, 0, Tp - T))
            text_mask = F.pad(text_mask, (0, Tp - T))
        if Rp != R:
            object_vl_embeddings, object_mask = F.pad(object_vl_embeddings, (0, 0, 0, Rp - R)), F.pad(object_mask, (0, Rp - R))
        eng = self._engine_for(B, Tp, Rp, sequence)
        eng.set_core_inputs(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask)
        anchor = next(iter(self._pnames.values()))
        a, b, third = _CoreFn.apply(text_visual_embeddings, object_vl_embeddings, anchor, self, eng)
        if not sequence:  # [B, Tp, .] / [B, Rp, .] -> the caller's T and R (the packed sequence output is trimmed to its own length)
            a, b = a[:, :T], b[:, :R]
        return a, b, third

    def forward(self, text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask,
                output_all_encoded_layers=True, output_text_and_object_separately=False, output_attention_probs=False):
        if output_all_encoded_layers or output_attention_probs:
            return self._inspect(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask,
                                 output_all_encoded_layers, output_text_and_object_separately, output_attention_probs)
        if not output_text_and_object_separately:     # VQA / VCR callers: (sequence_output, pooled_output)  (:139-171)
            seq, _, pooled = self._run(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask,
                                       object_vl_embeddings, object_mask, sequence=True)
            return seq, (pooled if self.with_pooler else None)
        text_out, obj_out, pooled = self._run(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask,
                                              object_vl_embeddings, object_mask)
        return text_out, obj_out, (pooled if self.with_pooler else None)

    @torch.no_grad()
    def _inspect(self, text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask,
                 all_layers, separately, probs):
        """The inspection call forms of the reference (:131-171; its viz/ scripts): every encoder layer's output and / or the attention
        probabilities [B, heads, S, S] of every layer.  The fused 16-bit attention kernel never materialises the probabilities; the fp32
        encoder path (../encoder_f32.py) keeps them per layer for its backward, so these forms run through it -- forward only, fp32,
        returned WITHOUT autograd history (train with the default call forms)."""
        B, T = text_input_ids.shape
        R = object_vl_embeddings.shape[1]
        key = ("inspect", B, T, R)
        eng = lru_get(self._engines, key, lambda: _engine.PretrainEngine(self.cfg, B, T, R, device=str(self.device_), flat=self.flat, core=True,
                                                                         core_heads=False, core_sequence=True,
                                                                         seed=ops.rank_seed(1234) // 2, encoder_fp32=True))
        version = self.flat.master._version
        if getattr(eng, "_synced_version", None) != version:
            eng.sync_weights()
            eng._synced_version = version
        eng._weights_dirty = False
        eng.set_core_inputs(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask)
        eng.mirror_pre_forward(self.training)
        eng.forward_core(train=self.training)
        enc, H, S, L, nh = eng.enc32, self.cfg.hidden_size, eng.S, self.cfg.num_hidden_layers, self.cfg.num_attention_heads
        n = int((eng.lay["text_len"] + eng.lay["nobj"]).max()) + 1          # the batch's longest packed sequence (:202)
        layers = [enc.X[l + 1].view(B, S, H)[:, :n].clone() for l in (range(L) if all_layers else [L - 1])]
        pooled = eng.pooled.float().clone() if self.with_pooler else None
        out_probs = None
        if probs:
            p_a = self.cfg.attention_probs_dropout_prob if self.training else 0.0
            src = enc.Pd if (p_a > 0 and enc.Pd is not None) else enc.P      # the reference returns them after their dropout
            out_probs = [src[l].view(B, nh, S, enc.Sp)[:, :, :n, :n].clone() for l in range(L)]
        if separately:
            rows = eng.lay["obj_rows"][:B].long()                            # packed row of object r of sample b, -1 = no object
            texts, objs = [], []
            for l in (range(L) if all_layers else [L - 1]):
                full = enc.X[l + 1]
                texts.append(full.view(B, S, H)[:, :T].clone())
                objs.append(full[rows.clamp(min=0).view(-1)].view(B, R, H) * (rows >= 0).unsqueeze(-1))
            if not all_layers:
                texts, objs = texts[0], objs[0]
            return (texts, objs, pooled, out_probs) if probs else (texts, objs, pooled)
        enc_out = layers if all_layers else layers[0]
        return (enc_out, pooled, out_probs) if probs else (enc_out, pooled)


class VisualLinguisticBertForPretraining(VisualLinguisticBert):
    WITH_HEADS = True

    def __init__(self, config, language_pretrained_model_path=None, with_rel_head=True, with_mlm_head=True, with_mvrc_head=True,
                 device=None):
        if with_rel_head and not _get(config, "with_pooler", False):
            raise ValueError("with_rel_head needs config.with_pooler (the relationship head reads the pooled output)")
        if not (with_mlm_head and with_mvrc_head):
            raise NotImplementedError("accelerated path computes both the MLM and the MVRC head")
        self.__dict__["with_rel_head"] = bool(with_rel_head)     # read by the base constructor (before nn.Module.__init__)
        super().__init__(config, language_pretrained_model_path, device=device)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
        sd[prefix + "mlm_head.predictions.decoder.weight"] = sd[prefix + "word_embeddings.weight"]   # tied (modeling.py:463-466)
        return sd

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        state_dict.pop("mlm_head.predictions.decoder.weight", None)
        return super().load_state_dict(state_dict, strict=strict)

    def forward(self, text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings, object_mask,
                output_all_encoded_layers=True, output_text_and_object_separately=False):
        mlm_logits, mvrc_logits, rel_logits = self._run(text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask,
                                                        object_vl_embeddings, object_mask)
        return (rel_logits if self.with_rel_head else None), mlm_logits, mvrc_logits
