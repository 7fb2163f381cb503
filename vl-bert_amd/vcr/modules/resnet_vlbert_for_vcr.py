"""Drop-in `ResNetVLBERT` for VCR fine-tuning (vcr/modules/resnet_vlbert_for_vcr.py:15-560) on the HIP library: same constructor
argument (the config tree of vcr/function/config.py), same `train_forward(image, boxes, masks, question, question_align_matrix,
answer_choices, answer_align_matrix, answer_label, im_info) -> (outputs, loss)` / `inference_forward(...) -> outputs`, same
parameter names (`image_feature_extractor.*`, `object_linguistic_embeddings.weight`, `vlbert._module.*`, `final_mlp.*`,
`cnn_loss_reg.*`), so the reference's trainer (vcr/function/train.py, SGD + gradient accumulation) and checkpoints work unchanged.

Composition, as in the reference (:226-399):
  FastRCNN mirror, image branch with the per-object masks (`segms`)                       -> obj_reps [B,R,H]
  per answer choice `[CLS] q [SEP] a [SEP]`, each token's visual embedding = the object its tag points at (index plumbing in
  torch, :116-167), object linguistic embedding = row clamp(class) of the 1-row / 81-row table (:303-308)
  `TimeDistributed` (common/nlp/time_distributed.py): the C answer choices fold into the batch -- the SAME engine rows, B*C
  sequences of up to 256 positions -- around the VisualLinguisticBert mirror with the pooler
  `final_mlp` on the pooled [CLS] -> logits [B,C]; sigmoid BCE with the positive-class weight (:344-356) or softmax CE (:358)
  ENABLE_CNN_REG_LOSS + CNN_LOSS_TOP (the shipped cfgs/vcr/*.yaml): every valid object's final hidden state -> transform
  (Linear + GELU) -> Dropout -> Linear(H, 81) -> CE against its detector class, one autograd node on the library (bf16 GEMMs with
  fused bias / GELU epilogues, vlb_ce_fwd_bwd, TN weight gradients).
The answer classifier (Dropout -> Linear(H,1) | 2fc) and the answer loss (weighted sigmoid BCE with the (w+1)/(2w) rescale, or softmax CE
over the choices) are one autograd node on the library as well (`_AnswerFn`).
Built as input plumbing: BLIND, NO_GROUNDING, NO_OBJ_ATTENTION, ANSWER_FIRST, QA_ONE_SENT.  Not built: object_word_embed_mode 3, IMAGE_SEMANTIC, the
bottom-of-the-CNN form of the regulariser (CNN_LOSS_TOP false), mask_position / mask_label (asserted off in the reference too).
"""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...common.fast_rcnn import FastRCNN
from ...common.visual_linguistic_bert import VisualLinguisticBert

F32 = torch.float32
CLS, SEP = 101, 102          # ids of '[CLS]', '[SEP]' in the BERT vocabularies (tokenizer lookups in the reference)
_TAG_REG, _TAG_A0, _TAG_A1 = 3001, 3002, 3003
NUM_OBJ_CLASSES = 81         # COCO detector classes of the VCR annotations (:26,38)


def _get(obj, name, default=None):
    return getattr(obj, name, default) if not isinstance(obj, dict) else obj.get(name, default)


def _ru(x, m):
    return (x + m - 1) // m * m


class TimeDistributed(nn.Module):
    """common/nlp/time_distributed.py:10-50: fold dimension 1 into dimension 0, apply, unfold every returned tensor."""

    def __init__(self, module):
        super().__init__()
        self._module = module

    def forward(self, *inputs, **kwargs):
        folded = []
        for t in inputs:
            if t.dim() <= 2:
                raise RuntimeError("No dimension to distribute: " + str(tuple(t.shape)))
            folded.append(t.contiguous().view(-1, *t.shape[2:]))
        lead = inputs[-1].shape[:2]
        out = self._module(*folded, **kwargs)
        unfold = lambda o: o.contiguous().view(*lead, *o.shape[1:])
        if isinstance(out, torch.Tensor):
            return unfold(out)
        if isinstance(out, tuple):
            return tuple(unfold(o) for o in out)
        raise ValueError("Not support!")


class _ObjClsFn(torch.autograd.Function):
    """x [n,H] fp32 (final hidden states of the valid objects), labels [n] -> CE loss of Linear(GELU(Linear(x))) (`cnn_loss_reg`,
    :33-37,391-394) on the device: two bf16 GEMMs with fused epilogues, vlb_ce_fwd_bwd (loss and d(logits) in one pass),
    hand-scheduled backward."""

    @staticmethod
    def forward(ctx, x, labels, module, train, w1, b1, w2, b2):
        n, H = x.shape
        st = module._reg_state(n, x.device)
        module._sync_reg()
        p = module.reg_drop if train else 0.0
        ops.cast_f32_bf16(x.detach().contiguous(), st["x0"])
        ops.gemm_nt(st["x0"], module._rw1, st["u"], bias=b1.detach(), act=ops.ACT_GELU_D, pre=st["du_act"])
        x1 = ops.dropout_bf16(st["u"], st["x1"], p, module._seed, _TAG_REG) if p > 0 else st["u"]
        ops.gemm_nt(x1, module._rw2, st["logits"][:, :NUM_OBJ_CLASSES], bias=b2.detach())
        st["loss"].zero_()
        ops.ce_fwd_bwd(st["logits"], NUM_OBJ_CLASSES, labels.contiguous(), st["count"], st["loss"], logits_copy=st["logits_copy"])
        ctx.module, ctx.st, ctx.p, ctx.x1, ctx.labels = module, st, p, x1, labels
        return st["loss"][0].clone()

    @staticmethod
    def backward(ctx, g_loss):
        module, st, p = ctx.module, ctx.st, ctx.p
        g = float(g_loss)
        if g != 1.0:      # upstream scale (CNN_LOSS_WEIGHT, gradient accumulation): re-derive d(logits) from the kept logits
            st["logits"].copy_(st["logits_copy"])
            st["loss"].zero_()
            ops.ce_fwd_bwd(st["logits"], NUM_OBJ_CLASSES, ctx.labels.contiguous(), st["count"], st["loss"], gscale=g)
        w1, b1, w2, b2 = module._reg_params()
        gw1, gb1, gw2, gb2 = (torch.zeros_like(q, dtype=F32) for q in (w1, b1, w2, b2))
        ops.wgrad_tn(st["logits"][:, :NUM_OBJ_CLASSES], ctx.x1, gw2, colsum=gb2, workspace=None)
        ops.gemm_nt(st["logits"], module._rw2T, st["dx1"])                     # K = the padded class dimension (zero columns)
        dh = ops.dropout_bf16(st["dx1"], st["dh"], p, module._seed, _TAG_REG) if p > 0 else st["dx1"]
        ops.mul_bf16(dh, st["du_act"], st["dpre"])
        ops.wgrad_tn(st["dpre"], st["x0"], gw1, colsum=gb1, workspace=None)
        ops.gemm_nt(st["dpre"], module._rw1T, st["dx0"])
        dx = torch.empty(st["dx0"].shape, dtype=F32, device=st["dx0"].device)
        ops.cast_bf16_f32(st["dx0"], dx)
        if p > 0:
            ops.rng_advance(module._seed)
        return dx, None, None, None, gw1, gb1, gw2, gb2


class _AnswerFn(torch.autograd.Function):
    """pooled [B,C,H] fp32 (BertPooler output per answer choice), answer_label [B] -> (label_logits [B,C], ans_loss): `final_mlp`
    (Dropout -> Linear(H,1) | Dropout -> Linear(H,hc) -> ReLU -> Dropout -> Linear(hc,1), :62-82) and the answer loss -- the weighted
    sigmoid BCE with the (w+1)/(2w) rescale, or the softmax CE over the choices (:333-345) -- on the device: bf16 GEMMs with fused
    bias / ReLU epilogues, counter-RNG dropout, vlb_bce_logits_fwd_bwd / vlb_ce_fwd_bwd (loss and d(logits) in one pass),
    hand-scheduled backward."""

    @staticmethod
    def forward(ctx, pooled, answer_label, module, train, *params):
        B, C, H = pooled.shape
        n = B * C
        st = module._cls_state(n, B, pooled.device)
        module._sync_cls()
        p = module.cls_drop if train else 0.0
        ops.cast_f32_bf16(pooled.detach().contiguous().view(n, H), st["x_in"])
        x0 = ops.dropout_bf16(st["x_in"], st["x0"], p, module._seed, _TAG_A0) if p > 0 else st["x_in"]
        if module.classifier == "1fc":
            x1 = x0
            ops.gemm_nt(x0, module._cw2, st["z"][:, :1], bias=params[1].detach())
        else:
            ops.gemm_nt(x0, module._cw1, st["u"], bias=params[1].detach(), act=ops.ACT_RELU)
            x1 = ops.dropout_bf16(st["u"], st["x1"], p, module._seed, _TAG_A1) if p > 0 else st["u"]
            ops.gemm_nt(x1, module._cw2, st["z"][:, :1], bias=params[3].detach())
        logits = st["z"][:, 0].float().view(B, C)                 # (glue: the [B,C] fp32 tensor the outputs dict carries)
        st["loss"].zero_()
        ctx.has_label = answer_label is not None
        if ctx.has_label:
            _AnswerFn._loss(module, st, answer_label, B, C, 1.0)
        ctx.module, ctx.st, ctx.p, ctx.x0, ctx.x1, ctx.label, ctx.shape = module, st, p, x0, x1, answer_label, (B, C, H)
        ctx.mark_non_differentiable(logits)
        return logits, st["loss"][0].clone()

    @staticmethod
    def _loss(module, st, answer_label, B, C, g):
        """loss value into st["loss"], g * d(loss)/d(logit) into column 0 of st["z"] (the kept logits live in st["z_copy"])."""
        n = B * C
        if module.sigmoid:      # mean over the B*C logits of w * BCE, times (w+1)/(2w)
            rescale = (module.pos_weight + 1.0) / (2.0 * module.pos_weight)
            lab = st["lab"]
            lab.zero_()
            lab.view(B, C).scatter_(1, answer_label.long().view(B, 1), 1.0)          # one-hot of the right answer (index plumbing)
            ops.bce_logits_fwd_bwd(st["z"], 1, lab.view(n, 1), st["loss"], gscale=g * rescale, logits_copy=st["z_copy"],
                                   pos_weight=module.pos_weight)
            st["loss"].mul_(rescale)
        else:                   # softmax cross entropy over the C choices of a sample
            st["z_copy"].copy_(st["z"])
            zc = st["zc"]
            zc.zero_()
            zc[:, :C].copy_(st["z"][:, 0].view(B, C))
            ops.ce_fwd_bwd(zc, C, answer_label.long().contiguous().view(-1), st["count"], st["loss"], gscale=g)
            st["z"].zero_()
            st["z"][:, 0].copy_(zc[:, :C].reshape(-1))

    @staticmethod
    def backward(ctx, _g_logits, g_loss):
        module, st, p = ctx.module, ctx.st, ctx.p
        B, C, H = ctx.shape
        n = B * C
        params = module._cls_params()
        grads = [torch.zeros_like(q, dtype=F32) for q in params]
        if not ctx.has_label:
            return (torch.zeros((B, C, H), dtype=F32, device=st["z"].device), None, None, None) + tuple(grads)
        g = float(g_loss)
        if g != 1.0:      # upstream scale (ANS_LOSS_WEIGHT, gradient accumulation): re-derive d(logits) from the kept logits
            st["z"].copy_(st["z_copy"])
            st["loss"].zero_()
            _AnswerFn._loss(module, st, ctx.label, B, C, g)
        dz = st["z"]                                                        # [n, 64]: column 0 live, the rest exactly zero
        gw2p = st["gw2p"]
        gw2p.zero_()
        st["gb2p"].zero_()
        if module.classifier == "1fc":
            gw2, gb2 = grads
            ops.wgrad_tn(dz, ctx.x1, gw2p[:, :H], colsum=st["gb2p"], workspace=None)
            gw2.copy_(gw2p[:1, :H]); gb2.copy_(st["gb2p"][:1])
            ops.gemm_nt(dz, module._cw2T, st["dx0"])                         # K = 64 (one live column)
        else:
            gw1, gb1, gw2, gb2 = grads
            hc = module.hc
            ops.wgrad_tn(dz, ctx.x1, gw2p[:, :hc], colsum=st["gb2p"], workspace=None)
            gw2.copy_(gw2p[:1, :hc]); gb2.copy_(st["gb2p"][:1])
            ops.gemm_nt(dz, module._cw2T, st["dx1"], act=ops.ACT_RELU_MASK, aux=st["u"])
            du = ops.dropout_bf16(st["dx1"], st["du"], p, module._seed, _TAG_A1) if p > 0 else st["dx1"]
            ops.wgrad_tn(du, ctx.x0, gw1, colsum=gb1, workspace=None)
            ops.gemm_nt(du, module._cw1T, st["dx0"])
        dx = ops.dropout_bf16(st["dx0"], st["dxin"], p, module._seed, _TAG_A0) if p > 0 else st["dx0"]
        d_pooled = torch.empty((n, H), dtype=F32, device=dx.device)
        ops.cast_bf16_f32(dx.contiguous(), d_pooled)
        if p > 0:
            ops.rng_advance(module._seed)
        return (d_pooled.view(B, C, H), None, None, None) + tuple(grads)


class ResNetVLBERT(nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        net = _get(config, "NETWORK")
        vl = _get(net, "VLBERT")
        for flag in ("FOR_MASK_VL_MODELING_PRETRAIN", "IMAGE_SEMANTIC"):
            if _get(net, flag, False):
                raise NotImplementedError("NETWORK.%s is not supported" % flag)
        # ablation switches of the reference's forward (:253-330): all of them are input plumbing in front of the same encoder
        self.blind = bool(_get(net, "BLIND", False))                      # no visual input at all: zero features, no object positions
        self.no_grounding = bool(_get(net, "NO_GROUNDING", False))        # every token tagged with box 0 (the whole image)
        self.no_obj_attention = bool(_get(net, "NO_OBJ_ATTENTION", False))      # objects feed the token embeddings but are not attended
        self.answer_first = bool(_get(net, "ANSWER_FIRST", False))        # [CLS] answer [SEP] question [SEP]
        self.qa_one_sent = bool(_get(net, "QA_ONE_SENT", False))          # [CLS] question answer [SEP], one segment
        if self.answer_first and self.qa_one_sent:
            raise NotImplementedError("ANSWER_FIRST with QA_ONE_SENT (the reference raises as well, :276-277)")
        if self.blind and _get(net, "ENABLE_CNN_REG_LOSS", False):
            raise NotImplementedError("BLIND with ENABLE_CNN_REG_LOSS: there are no object positions to classify")
        self.embed_mode = int(_get(vl, "object_word_embed_mode", 2))
        if self.embed_mode not in (1, 2):
            raise NotImplementedError("object_word_embed_mode must be 1 (81 class embeddings) or 2 (one shared embedding)")
        self.enable_cnn_reg_loss = bool(_get(net, "ENABLE_CNN_REG_LOSS", False))
        self.cnn_loss_top = bool(_get(net, "CNN_LOSS_TOP", False))
        if self.enable_cnn_reg_loss and not self.cnn_loss_top:
            raise NotImplementedError("ENABLE_CNN_REG_LOSS needs CNN_LOSS_TOP (the form of the shipped cfgs/vcr/*.yaml)")
        self.classifier = _get(net, "CLASSIFIER_TYPE", "2fc")
        if self.classifier not in ("1fc", "2fc"):
            raise ValueError("Not support classifier type: {}!".format(self.classifier))
        if not torch.cuda.is_available():
            raise RuntimeError("ResNetVLBERT (HIP) needs an MI355X: there is no CPU fallback")
        dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.device_ = dev
        H = int(_get(vl, "hidden_size"))
        self.H = H
        self.sigmoid = bool(_get(net, "CLASSIFIER_SIGMOID", False))
        self.pos_weight = float(_get(net, "CLASSIFIER_SIGMOID_LOSS_POSITIVE_WEIGHT", 1.0))
        self.cls_drop = float(_get(net, "CLASSIFIER_DROPOUT", 0.1))
        self.reg_drop = float(_get(net, "CNN_REG_DROPOUT", 0.0))
        self.ans_loss_weight = float(_get(net, "ANS_LOSS_WEIGHT", 1.0))
        self.cnn_loss_weight = float(_get(net, "CNN_LOSS_WEIGHT", 1.0))
        self.image_feature_extractor = FastRCNN(config, average_pool=True, final_dim=_get(net, "IMAGE_FINAL_DIM", 768),
                                                enable_cnn_reg_loss=False, device=dev)
        self.object_linguistic_embeddings = nn.Embedding(NUM_OBJ_CLASSES if self.embed_mode == 1 else 1, H).to(dev)
        from ...common import language_pretrained as _lp
        self.language_pretrained_model_path = _lp.resolve_path(net)                    # (vcr/modules/resnet_vlbert_for_vcr.py:49-58)
        if self.language_pretrained_model_path is None:
            print("Warning: no pretrained language model found, training from scratch!!!", file=sys.stderr)   # (the reference prints to stdout; bench.py owns stdout)
        self.vlbert = TimeDistributed(VisualLinguisticBert(vl, language_pretrained_model_path=self.language_pretrained_model_path, device=dev))

        def lin(o, i):
            m = nn.Module()
            m.register_parameter("weight", nn.Parameter(torch.empty((o, i), device=dev)))
            m.register_parameter("bias", nn.Parameter(torch.zeros((o,), device=dev)))
            return m
        mlp = nn.Module()
        if self.classifier == "1fc":
            mlp.add_module("1", lin(1, H))
        else:
            hc = int(_get(net, "CLASSIFIER_HIDDEN_SIZE", 1024))
            mlp.add_module("1", lin(hc, H))
            mlp.add_module("4", lin(1, hc))
        self.final_mlp = mlp
        if self.enable_cnn_reg_loss:
            reg = nn.Module()
            tr = nn.Module()
            tr.add_module("dense", lin(H, H))
            reg.add_module("0", tr)
            reg.add_module("2", lin(NUM_OBJ_CLASSES, H))
            self.cnn_loss_reg = reg
            zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
            self.Cp = _ru(NUM_OBJ_CLASSES, 64)
            self._rw1, self._rw1T = zb(H, H), zb(H, H)
            self._rw2, self._rw2T = zb(NUM_OBJ_CLASSES, H), zb(H, self.Cp)
        # bf16 working copies of the classifier: first Linear [hc,H] (2fc) and the 1-output Linear padded to 64 rows (+ transposes)
        zc = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
        self.hc = int(_get(net, "CLASSIFIER_HIDDEN_SIZE", 1024)) if self.classifier != "1fc" else H
        if self.hc % 64:
            raise NotImplementedError("CLASSIFIER_HIDDEN_SIZE must be a multiple of 64")
        self._cw1, self._cw1T = (zc(self.hc, H), zc(H, self.hc)) if self.classifier != "1fc" else (None, None)
        self._cw2, self._cw2T = zc(1, self.hc), zc(self.hc, 64)
        self._cls_version, self._cls_states = None, {}
        self._seed = torch.tensor([ops.rank_seed(40011)], dtype=torch.int32, device=dev)
        self._reg_version, self._states = None, {}
        self.init_weight()

    # -- parameters ---------------------------------------------------------------------------------
    def _cls_params(self):
        m = self.final_mlp
        if self.classifier == "1fc":
            l = getattr(m, "1")
            return [l.weight, l.bias]
        a, b = getattr(m, "1"), getattr(m, "4")
        return [a.weight, a.bias, b.weight, b.bias]

    def _sync_cls(self):
        params = self._cls_params()
        ver = tuple(q._version for q in params)
        if ver == self._cls_version:
            return
        w2 = params[0] if self.classifier == "1fc" else params[2]
        ops.cast_f32_bf16(w2.detach().contiguous(), self._cw2)
        self._cw2T.zero_()
        self._cw2T[:, 0].copy_(self._cw2[0])                           # [hc, 64] with one live column (index plumbing)
        if self.classifier != "1fc":
            ops.cast_f32_bf16(params[0].detach().contiguous(), self._cw1)
            ops.transpose(self._cw1, self._cw1T)
        self._cls_version = ver

    def _cls_state(self, n, B, dev):
        if (n, B) not in self._cls_states:
            zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
            zf = lambda *s: torch.zeros(s, dtype=F32, device=dev)
            H, hc = self.H, self.hc
            self._cls_states[(n, B)] = dict(x_in=zb(n, H), x0=zb(n, H), u=zb(n, hc), x1=zb(n, hc), z=zb(n, 64), z_copy=zb(n, 64), zc=zb(B, 64),
                                            dx1=zb(n, hc), du=zb(n, hc), dx0=zb(n, H), dxin=zb(n, H), lab=zf(n), gw2p=zf(64, max(H, hc)),
                                            gb2p=zf(64), loss=zf(1), count=zf(1))
        return self._cls_states[(n, B)]

    def _reg_params(self):
        r = self.cnn_loss_reg
        t, c = getattr(r, "0"), getattr(r, "2")
        return [t.dense.weight, t.dense.bias, c.weight, c.bias]

    def init_weight(self):
        """:84-97: N(0, 0.02) object word embeddings and regulariser head, xavier-uniform classifier, zero biases."""
        with torch.no_grad():
            self.image_feature_extractor.init_weight()
            self.object_linguistic_embeddings.weight.normal_(0.0, 0.02)
            if self.enable_cnn_reg_loss:
                for q in self._reg_params():
                    if q.dim() == 2:
                        q.normal_(0.0, 0.02)
                    else:
                        q.zero_()
            for m in self.final_mlp.children():
                nn.init.xavier_uniform_(m.weight)
                m.bias.zero_()

    def fix_params(self):
        pass

    def train(self, mode=True):
        super().train(mode)
        self.image_feature_extractor.bn_eval()            # frozen BatchNorm (:99-104); the HIP vision stack folds it anyway
        return self

    def load_state_dict(self, state_dict, strict=True):
        """the FastRCNN mirror converts its convolution layout ([O,KH,KW,I] <-> the reference's [O,I,KH,KW]) and drops the reference's
        `head.0.*` alias keys in its own load_state_dict, which nn.Module's recursion does not call for sub-modules (state_dict's
        recursion does): hand it its slice directly, load the rest here"""
        pre = "image_feature_extractor."
        self.image_feature_extractor.load_state_dict({k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}, strict=strict)
        rest = {k: v for k, v in state_dict.items() if not k.startswith(pre)}
        res = super().load_state_dict(rest, strict=False)
        missing = [k for k in res.missing_keys if not k.startswith(pre)]
        if strict and (missing or res.unexpected_keys):
            raise RuntimeError("Error(s) in loading state_dict for ResNetVLBERT: missing %s, unexpected %s" % (missing, res.unexpected_keys))
        return res

    def _sync_reg(self):
        params = self._reg_params()
        ver = tuple(q._version for q in params)
        if ver == self._reg_version:
            return
        ops.cast_f32_bf16(params[0].detach().contiguous(), self._rw1)
        ops.cast_f32_bf16(params[2].detach().contiguous(), self._rw2)
        ops.transpose(self._rw1, self._rw1T)
        ops.transpose(self._rw2, self._rw2T)          # [H, 81] into the zero-padded [H, Cp] image
        self._reg_version = ver

    def _reg_state(self, n, dev):
        cap = _ru(n, 64)
        if cap not in self._states:
            zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
            H, Cp = self.H, self.Cp
            self._states[cap] = dict(x0=zb(cap, H), u=zb(cap, H), du_act=zb(cap, H), x1=zb(cap, H), logits=zb(cap, Cp),
                                     logits_copy=zb(cap, Cp), dx1=zb(cap, H), dh=zb(cap, H), dpre=zb(cap, H), dx0=zb(cap, H),
                                     loss=torch.zeros((1,), dtype=F32, device=dev), count=torch.zeros((1,), dtype=F32, device=dev))
        return {k: (v[:n] if v.dim() == 2 else v) for k, v in self._states[cap].items()}

    # -- text preparation: index plumbing (prepare_text_from_qa / _qa_onesent / _aq, :136-224) ------------------------
    @staticmethod
    def _prepare_text(question, question_tags, question_mask, answers, answers_tags, answers_mask, order="qa"):
        """order: "qa"  [CLS] q [SEP] a [SEP] (segment 1 = answer + its [SEP]) | "qa_onesent"  [CLS] q a [SEP] (one segment) |
        "aq"  [CLS] a [SEP] q [SEP] (segment 1 = question + its [SEP])"""
        B, Lq = question.shape
        _, C, La = answers.shape
        n_sep = 1 if order == "qa_onesent" else 2
        L = int((question_mask.sum(1) + answers_mask.sum(2).max(1)[0]).max()) + 1 + n_sep
        d = question.device
        question = question[:, None, :].expand(B, C, Lq)
        qmask = question_mask[:, None, :].expand(B, C, Lq)
        nq, na = qmask.sum(2, keepdim=True), answers_mask.sum(2, keepdim=True)
        k = torch.arange(L, device=d)[None, None, :].expand(B, C, L)
        ids = torch.zeros((B, C, L), dtype=question.dtype, device=d)
        tags = torch.zeros((B, C, L), dtype=question.dtype, device=d)
        if order == "qa":
            first_end, last_end = 1 + nq, 2 + nq + na                 # positions of the two [SEP]
            q_in, a_in = (k > 0) & (k < first_end), (k > first_end) & (k < last_end)
        elif order == "aq":
            first_end, last_end = 1 + na, 2 + na + nq
            a_in, q_in = (k > 0) & (k < first_end), (k > first_end) & (k < last_end)
        else:
            first_end, last_end = 1 + nq, 1 + nq + na                 # no separator between question and answer
            q_in, a_in = (k > 0) & (k < first_end), (k >= first_end) & (k < last_end)
        mask = k <= last_end
        types = ((k > first_end) & (k <= last_end)).to(question.dtype) if order != "qa_onesent" else torch.zeros_like(ids)
        ids[:, :, 0] = CLS
        if order != "qa_onesent":
            ids[k == first_end] = SEP
        ids[k == last_end] = SEP
        ids[q_in] = question[qmask]
        ids[a_in] = answers[answers_mask]
        tags[q_in] = question_tags[qmask]
        tags[a_in] = answers_tags[answers_mask]
        return ids, types, tags, mask

    def _encode(self, image, boxes, masks, question, answer_choices, im_info):
        objects = boxes[:, :, -1]
        boxes4 = boxes[:, :, :4]
        box_mask = boxes4[:, :, -1] > -0.5
        max_len = int(box_mask.sum(1).max())
        objects, box_mask, boxes4, segms = objects[:, :max_len], box_mask[:, :max_len], boxes4[:, :max_len], masks[:, :max_len]
        B, R = box_mask.shape
        if self.blind:
            reps = torch.zeros((B, R, self.H), dtype=F32, device=boxes4.device)
        else:
            reps = self.image_feature_extractor(images=image, boxes=boxes4, box_mask=box_mask, im_info=im_info, classes=objects,
                                                segms=segms)["obj_reps"]
        C = answer_choices.shape[1]
        q_ids, q_tags = question[:, :, 0], question[:, :, 1]
        q_tags = q_tags[:, None, :].expand(-1, C, -1)
        q_mask = question[:, :, 0] > 0.5
        a_ids, a_tags = answer_choices[:, :, :, 0], answer_choices[:, :, :, 1]
        a_mask = answer_choices[:, :, :, 0] > 0.5
        order = "aq" if self.answer_first else ("qa_onesent" if self.qa_one_sent else "qa")
        ids, types, tags, text_mask = self._prepare_text(q_ids, q_tags, q_mask, a_ids, a_tags, a_mask, order=order)
        if self.no_grounding:
            tags = torch.zeros_like(tags)
        L = ids.shape[2]
        rows = torch.arange(B, device=ids.device)[:, None, None].expand(B, C, L)
        text_visual = reps[rows.reshape(-1), tags.clamp(min=0).reshape(-1)].view(B, C, L, -1)          # _collect_obj_reps (:116-134)
        if self.blind:
            ling = torch.zeros_like(reps)
        else:
            table = self.object_linguistic_embeddings.weight
            ling = table[objects.long().clamp(min=0, max=table.shape[0] - 1)]
        obj_vl = torch.cat((reps, ling), -1)[:, None].expand(B, C, R, -1)
        attend = torch.zeros_like(box_mask) if (self.no_obj_attention or self.blind) else box_mask
        text_out, obj_out, pooled = self.vlbert(ids, types, text_visual, text_mask, obj_vl, attend[:, None].expand(B, C, R),
                                                output_all_encoded_layers=False, output_text_and_object_separately=True)
        return pooled, obj_out, objects, box_mask

    def _classify(self, pooled, answer_label=None):
        """-> (label_logits [B,C], ans_loss | None) through the HIP head (_AnswerFn)."""
        logits, loss = _AnswerFn.apply(pooled.float(), answer_label, self, self.training, *self._cls_params())
        return logits, (loss if answer_label is not None else None)

    def train_forward(self, image, boxes, masks, question, question_align_matrix, answer_choices, answer_align_matrix, answer_label,
                      im_info, mask_position=None, mask_type=None, mask_label=None):
        if mask_position is not None:
            raise NotImplementedError("mask_position (asserted off in the reference, :365)")
        pooled, obj_out, objects, box_mask = self._encode(image, boxes, masks, question, answer_choices, im_info)
        logits, ans_loss = self._classify(pooled, answer_label)
        B, C = logits.shape
        outputs = {}
        if self.sigmoid:
            outputs["positive_fraction"] = torch.full((), 1.0 / C, dtype=logits.dtype, device=logits.device)     # one right answer per sample
        outputs.update({"label_logits": logits, "label": answer_label.long().view(-1), "ans_loss": ans_loss})
        loss = ans_loss.mean() * self.ans_loss_weight
        if self.enable_cnn_reg_loss:
            R = box_mask.shape[1]
            sel = box_mask[:, None].expand(B, C, R)
            labels = objects[:, None].expand(B, C, R)[sel].long()
            reg_loss = _ObjClsFn.apply(obj_out[sel].float(), labels, self, self.training, *self._reg_params())
            loss = loss + reg_loss * self.cnn_loss_weight
            outputs["cnn_regularization_loss"] = reg_loss
        return outputs, loss

    def inference_forward(self, image, boxes, masks, question, question_align_matrix, answer_choices, answer_align_matrix, *args):
        im_info = args[-1]
        pooled, _, _, _ = self._encode(image, boxes, masks, question, answer_choices, im_info)
        return {"label_logits": self._classify(pooled)[0]}

    def forward(self, *inputs, **kwargs):
        """common/module.py:19-24"""
        return self.train_forward(*inputs, **kwargs) if self.training else self.inference_forward(*inputs, **kwargs)
