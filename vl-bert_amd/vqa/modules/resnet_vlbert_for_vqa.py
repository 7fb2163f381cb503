"""Drop-in `ResNetVLBERT` for VQA fine-tuning (vqa/modules/resnet_vlbert_for_vqa.py:14-300) on the HIP library: same constructor
argument (the config tree of vqa/function/config.py), same `train_forward(image, boxes, im_info, question, label) -> (outputs, loss)`
and `inference_forward(image, boxes, im_info, question) -> outputs`, same parameter names (`image_feature_extractor.*`,
`object_linguistic_embeddings.weight`, `vlbert.*`, `final_mlp.*`), so the reference's trainer and checkpoints work unchanged.

Composition, as in the reference:  FastRCNN mirror (precomputed features or images) -> text = [CLS] question [SEP] [MASK] [SEP]
(index plumbing in torch, :142-167) -> VisualLinguisticBert mirror (packed sequence output) -> hidden state at the [MASK] position ->
`final_mlp` -> BCE-with-logits x answers (:226).  The classifier and the loss are ONE autograd node running on the library: bf16 GEMMs
with fused bias / ReLU / GELU epilogues, LayerNorm, counter-RNG dropout (vlb_dropout_bf16), vlb_bce_logits_fwd_bwd (loss and its
gradient in one pass), TN weight-gradient GEMMs.  CLASSIFIER_TYPE "2fc" (config default) and "mlm" (the shipped cfgs/vqa/*.yaml)
are built; "1fc", BLIND, NO_GROUNDING, CLASSIFIER_SIGMOID, cnn_reg_loss raise NotImplementedError.
"""
import sys

import torch
import torch.nn as nn

from ... import ops
from ...common.fast_rcnn import FastRCNN
from ...common.visual_linguistic_bert import VisualLinguisticBert

F32 = torch.float32
CLS, SEP, MASK = 101, 102, 103          # ids of '[CLS]', '[SEP]', '[MASK]' in the BERT vocabularies (tokenizer lookups in the reference)
_TAG0, _TAG1 = 2001, 2002


def _get(obj, name, default=None):
    return getattr(obj, name, default) if not isinstance(obj, dict) else obj.get(name, default)


def _ru(x, m):
    return (x + m - 1) // m * m


class _HeadFn(torch.autograd.Function):
    """hm [B,H] fp32 -> (logits [B,A] fp32, loss): final_mlp + BCE on the device, hand-scheduled backward."""

    @staticmethod
    def forward(ctx, hm, label, module, train, *params):
        B, H = hm.shape
        st = module._head_state(B, hm.device)
        module._sync_head()
        A, Ap = module.answers, module.Ap
        p = module.cls_drop if train else 0.0
        ops.cast_f32_bf16(hm.detach().contiguous(), st["x_in"])
        if module.classifier == "1fc":
            x0 = ops.dropout_bf16(st["x_in"], st["x0"], p, module._seed, _TAG0) if p > 0 else st["x_in"]
            x1 = x0
            ops.gemm_nt(x0, module._w2, st["logits"][:, :A], bias=params[1].detach())
        elif module.classifier == "2fc":
            x0 = ops.dropout_bf16(st["x_in"], st["x0"], p, module._seed, _TAG0) if p > 0 else st["x_in"]
            ops.gemm_nt(x0, module._w1, st["u"][:, :module.hc], bias=params[1].detach(), act=ops.ACT_RELU)
            x1 = ops.dropout_bf16(st["u"], st["x1"], p, module._seed, _TAG1) if p > 0 else st["u"]
            ops.gemm_nt(x1, module._w2, st["logits"][:, :A], bias=params[3].detach())
        else:   # "mlm": BertPredictionHeadTransform (dense + gelu + LayerNorm) -> Dropout -> Linear
            x0 = st["x_in"]
            ops.gemm_nt(x0, module._w1, st["u"], bias=params[1].detach(), act=ops.ACT_GELU_D, pre=st["du_act"])
            ops.layernorm_fwd(st["u"], params[2].detach(), params[3].detach(), st["h"], st["stats"])
            x1 = ops.dropout_bf16(st["h"], st["x1"], p, module._seed, _TAG1) if p > 0 else st["h"]
            ops.gemm_nt(x1, module._w2, st["logits"][:, :A], bias=params[5].detach())
        st["loss"].zero_()
        has_label = label is not None
        if has_label:
            ops.bce_logits_fwd_bwd(st["logits"], A, label.detach().float().contiguous(), st["loss"], logits_copy=st["logits_copy"])
            logits = st["logits_copy"][:, :A].float()
        else:
            logits = st["logits"][:, :A].float()
        ctx.module, ctx.st, ctx.p, ctx.x0, ctx.x1, ctx.label = module, st, p, x0, x1, label
        ctx.mark_non_differentiable(logits)
        return logits, st["loss"][0].clone()

    @staticmethod
    def backward(ctx, _g_logits, g_loss):
        module, st, p = ctx.module, ctx.st, ctx.p
        A = module.answers
        params = module._head_params()
        g = float(g_loss)
        if g != 1.0:      # upstream scale (gradient accumulation / loss scaling): re-derive d(logits) from the kept logits
            st["logits"].copy_(st["logits_copy"])
            st["loss"].zero_()
            ops.bce_logits_fwd_bwd(st["logits"], A, ctx.label.detach().float().contiguous(), st["loss"], gscale=g)
        dlog = st["logits"][:, :A]
        grads = [torch.zeros_like(q, dtype=F32) for q in params]
        if module.classifier == "1fc":
            gw2, gb2 = grads
            ops.wgrad_tn(dlog, ctx.x1, gw2, colsum=gb2, workspace=None)
            ops.gemm_nt(st["logits"], module._w2T, st["dx0"])                                # K = padded answers (zero columns)
            dx = ops.dropout_bf16(st["dx0"], st["dxin"], p, module._seed, _TAG0) if p > 0 else st["dx0"]
        elif module.classifier == "2fc":
            gw1, gb1, gw2, gb2 = grads
            ops.wgrad_tn(dlog, ctx.x1[:, :module.hc], gw2, colsum=gb2, workspace=None)
            ops.gemm_nt(st["logits"], module._w2T, st["dx1"][:, :module.hc], act=ops.ACT_RELU_MASK, aux=st["u"][:, :module.hc])   # K = padded answers (zero columns)
            du = ops.dropout_bf16(st["dx1"], st["du"], p, module._seed, _TAG1) if p > 0 else st["dx1"]
            ops.wgrad_tn(du[:, :module.hc], ctx.x0, gw1, colsum=gb1, workspace=None)
            ops.gemm_nt(du, module._w1T, st["dx0"])
            dx = ops.dropout_bf16(st["dx0"], st["dxin"], p, module._seed, _TAG0) if p > 0 else st["dx0"]
        else:
            gw1, gb1, gg, gbeta, gw2, gb2 = grads
            ops.wgrad_tn(dlog, ctx.x1, gw2, colsum=gb2, workspace=None)
            ops.gemm_nt(st["logits"], module._w2T, st["dx1"])
            dh = ops.dropout_bf16(st["dx1"], st["du"], p, module._seed, _TAG1) if p > 0 else st["dx1"]
            ops.layernorm_bwd(dh, st["u"], st["stats"], params[2].detach(), dx=st["dln"], dgamma=gg, dbeta=gbeta)
            ops.mul_bf16(st["dln"], st["du_act"], st["dpre"])
            ops.wgrad_tn(st["dpre"], ctx.x0, gw1, colsum=gb1, workspace=None)
            dx = ops.gemm_nt(st["dpre"], module._w1T, st["dx0"])
        d_hm = torch.empty(dx.shape, dtype=F32, device=dx.device)
        ops.cast_bf16_f32(dx.contiguous(), d_hm)
        if p > 0:
            ops.rng_advance(module._seed)
        return (d_hm, None, None, None) + tuple(grads)


class ResNetVLBERT(nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        net = _get(config, "NETWORK")
        vl = _get(net, "VLBERT")
        if _get(net, "BLIND", False) or _get(net, "NO_GROUNDING", False) or _get(net, "ENABLE_CNN_REG_LOSS", False):
            raise NotImplementedError("BLIND / NO_GROUNDING / ENABLE_CNN_REG_LOSS are not supported")
        # (CLASSIFIER_SIGMOID is a key of the shared config schema that the reference's VQA module never reads: the answer loss is
        #  always the sigmoid BCE of :226)
        if _get(vl, "object_word_embed_mode", 2) != 2:
            raise NotImplementedError("object_word_embed_mode must be 2 (one shared object word embedding)")
        self.classifier = _get(net, "CLASSIFIER_TYPE", "2fc")
        if self.classifier not in ("2fc", "1fc", "mlm"):
            raise ValueError("Not support classifier type: %s!" % self.classifier)       # (the reference's message, :75-76)
        if not torch.cuda.is_available():
            raise RuntimeError("ResNetVLBERT (HIP) needs an MI355X: there is no CPU fallback")
        dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.device_ = dev
        H = _get(vl, "hidden_size")
        self.H = H
        self.answers = int(_get(_get(config, "DATASET"), "ANSWER_VOCAB_SIZE", 3129))
        self.Ap = _ru(self.answers, 64)
        self.cls_drop = float(_get(net, "CLASSIFIER_DROPOUT", 0.1))
        self.image_feature_extractor = FastRCNN(config, average_pool=True, final_dim=_get(net, "IMAGE_FINAL_DIM", 768), device=dev)
        self.object_linguistic_embeddings = nn.Embedding(1, H).to(dev)
        from ...common import language_pretrained as _lp
        self.language_pretrained_model_path = _lp.resolve_path(net)                    # (:37-47)
        if self.language_pretrained_model_path is None:
            print("Warning: no pretrained language model found, training from scratch!!!", file=sys.stderr)   # (the reference prints to stdout; bench.py owns stdout)
        self.vlbert = VisualLinguisticBert(vl, language_pretrained_model_path=self.language_pretrained_model_path, device=dev)
        mlp = nn.Module()

        def lin(o, i):
            m = nn.Module()
            m.register_parameter("weight", nn.Parameter(torch.empty((o, i), device=dev)))
            m.register_parameter("bias", nn.Parameter(torch.zeros((o,), device=dev)))
            return m
        if self.classifier == "2fc":
            self.hc = int(_get(net, "CLASSIFIER_HIDDEN_SIZE", 1024))
            mlp.add_module("1", lin(self.hc, H))
            mlp.add_module("4", lin(self.answers, self.hc))
        elif self.classifier == "1fc":            # Dropout -> Linear(H, answers)  (:64-68)
            self.hc = H
            mlp.add_module("1", lin(self.answers, H))
        else:
            self.hc = H
            tr = nn.Module()
            tr.add_module("dense", lin(H, H))
            ln = nn.Module()
            ln.register_parameter("weight", nn.Parameter(torch.ones((H,), device=dev)))
            ln.register_parameter("bias", nn.Parameter(torch.zeros((H,), device=dev)))
            tr.add_module("LayerNorm", ln)
            mlp.add_module("0", tr)
            mlp.add_module("2", lin(self.answers, H))
        self.final_mlp = mlp
        self.hcp = _ru(self.hc, 64)
        zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
        self._w1, self._w1T = zb(self.hc, H), zb(H, self.hcp)          # first Linear and its transpose (K padded to 64; unused by "1fc")
        self._w2, self._w2T = zb(self.answers, self.hcp), zb(self.hc, self.Ap)
        self._seed = torch.tensor([ops.rank_seed(30011)], dtype=torch.int32, device=dev)
        self._head_version, self._states = None, {}
        self.init_weight()

    # -- parameters ---------------------------------------------------------------------------------
    def _head_params(self):
        m = self.final_mlp
        if self.classifier == "2fc":
            a, b = getattr(m, "1"), getattr(m, "4")
            return [a.weight, a.bias, b.weight, b.bias]
        if self.classifier == "1fc":
            a = getattr(m, "1")
            return [a.weight, a.bias]
        t, b = getattr(m, "0"), getattr(m, "2")
        return [t.dense.weight, t.dense.bias, t.LayerNorm.weight, t.LayerNorm.bias, b.weight, b.bias]

    def init_weight(self):
        """resnet_vlbert_for_vqa.py:84-110: xavier-uniform classifier Linears, zero biases, N(0, 0.02) object word embedding."""
        with torch.no_grad():
            self.image_feature_extractor.init_weight()
            self.object_linguistic_embeddings.weight.normal_(0.0, 0.02)
            for q in self._head_params():
                if q.dim() == 2:
                    nn.init.xavier_uniform_(q)
            if self.classifier == "mlm":
                t = getattr(self.final_mlp, "0")
                t.dense.bias.zero_()
                if self.language_pretrained_model_path is not None:
                    # the classifier's transform starts from the language model's MLM transform (:97-110).  (Without a checkpoint the
                    # reference dies in torch.load(None); the mirror keeps the random init so that synthetic benches can run.)
                    from ...common import language_pretrained as _lp
                    sd = torch.load(self.language_pretrained_model_path, map_location="cpu")
                    tsd, keys = _lp.mlm_transform_state_dict(sd)
                    print("loading pretrained classifier transform keys: {}.".format(keys))
                    _lp.apply([(k, v, None) for k, v in tsd.items()],
                              {"dense.weight": t.dense.weight, "dense.bias": t.dense.bias, "LayerNorm.weight": t.LayerNorm.weight,
                               "LayerNorm.bias": t.LayerNorm.bias}, strict_keys=True)

    def fix_params(self):
        pass

    def _sync_head(self):
        params = self._head_params()
        ver = tuple(q._version for q in params)
        if ver == self._head_version:
            return
        w1, w2 = (params[0], params[2]) if self.classifier == "2fc" else ((None, params[0]) if self.classifier == "1fc" else (params[0], params[4]))
        if w1 is not None:
            ops.cast_f32_bf16(w1.detach().contiguous(), self._w1)
        tmp = torch.zeros((self.answers, self.hc), dtype=ops.BF16, device=self.device_)
        ops.cast_f32_bf16(w2.detach().contiguous(), tmp)
        self._w2.zero_()
        self._w2[:, :self.hc].copy_(tmp)
        if w1 is not None:
            ops.transpose(self._w1, self._w1T)
        ops.transpose(tmp, self._w2T)
        self._head_version = ver

    def _head_state(self, B, dev):
        if B not in self._states:
            zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
            H, hcp, Ap = self.H, self.hcp, self.Ap
            self._states[B] = dict(x_in=zb(B, H), x0=zb(B, H), u=zb(B, hcp), x1=zb(B, hcp), h=zb(B, hcp), du_act=zb(B, hcp),
                                   logits=zb(B, Ap), logits_copy=zb(B, Ap), dx1=zb(B, hcp), du=zb(B, hcp), dln=zb(B, hcp), dpre=zb(B, hcp),
                                   dx0=zb(B, H), dxin=zb(B, H), stats=torch.zeros((B, 2), dtype=F32, device=dev),
                                   loss=torch.zeros((1,), dtype=F32, device=dev))
        return self._states[B]

    # -- text preparation: index plumbing (prepare_text_from_qa, :142-167, with the single [MASK] answer token of :192-196) ----------
    @staticmethod
    def _prepare_text(question):
        B = question.shape[0]
        qmask = question > 0
        qlen = qmask.sum(1)
        L = int(qlen.max()) + 4
        q_end = (1 + qlen)[:, None]
        a_end = q_end + 2
        j = torch.arange(L, device=question.device)[None, :]
        ids = torch.zeros((B, L), dtype=question.dtype, device=question.device)
        types = ((j > q_end) & (j <= a_end)).to(question.dtype)
        mask = j <= a_end
        ids[:, 0] = CLS
        ids[(j > 0) & (j < q_end)] = question[qmask]
        ids[j == q_end] = SEP
        ids[j == q_end + 1] = MASK
        ids[j == a_end] = SEP
        return ids, types, mask, (a_end - 1).squeeze(1)

    def _features(self, image, boxes, im_info, question):
        box_mask = boxes[:, :, 0] > -1.5
        max_len = int(box_mask.sum(1).max())
        box_mask, boxes = box_mask[:, :max_len], boxes[:, :max_len]
        obj = self.image_feature_extractor(images=image, boxes=boxes, box_mask=box_mask, im_info=im_info, classes=None, segms=None)
        ids, types, text_mask, ans_pos = self._prepare_text(question)
        reps = obj["obj_reps"]
        text_visual = reps[:, 0:1].expand(-1, ids.shape[1], -1)                 # text tags are all 0 (:198-209)
        B, R = box_mask.shape
        ling = self.object_linguistic_embeddings.weight[0].expand(B, R, -1)
        obj_vl = torch.cat((reps, ling), -1)
        seq, _ = self.vlbert(ids, types, text_visual, text_mask, obj_vl, box_mask, output_all_encoded_layers=False)
        return seq[torch.arange(B, device=seq.device), ans_pos]

    def train_forward(self, image, boxes, im_info, question, label):
        hm = self._features(image, boxes, im_info, question)
        logits, loss = _HeadFn.apply(hm, label, self, self.training, *self._head_params())
        return {"label_logits": logits, "label": label, "ans_loss": loss}, loss

    def inference_forward(self, image, boxes, im_info, question):
        hm = self._features(image, boxes, im_info, question)
        logits, _ = _HeadFn.apply(hm, None, self, False, *self._head_params())
        return {"label_logits": logits}

    def forward(self, *inputs, **kwargs):
        """common/module.py:19-24"""
        return self.train_forward(*inputs, **kwargs) if self.training else self.inference_forward(*inputs, **kwargs)
