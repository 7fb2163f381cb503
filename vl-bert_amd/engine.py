"""MI355X-native VL-BERT pre-training engine: the whole training step of
`ResNetVLBERTForPretraining` (precomputed-feature configuration) as a fixed sequence of
hand-written HIP kernels called through the C ABI.

Reference path being replaced: pretrain/modules/resnet_vlbert_for_pretraining.py:93-216
-> common/fast_rcnn.py:128-203 -> common/visual_linguistic_bert.py:95-241,346-380 ->
external/pytorch_pretrained_bert/modeling.py:268-482, its autograd backward, clip_grad_norm_ +
AdamW.step (common/trainer.py:123-153, common/nlp/bert/optimization.py:129-187) and the DDP gradient
all-reduce (pretrain/function/train.py:89-90).

Design: static shapes [B, S=T+R+1] with in-kernel masking (no .item()/.nonzero() host
syncs), bf16 activations / fp32 accumulation / fp32 master weights, every parameter in ONE flat fp32
buffer (nn.Parameter views alias it; gradients, Adam moments and the bf16 working copy are flat
twins), explicit hand-scheduled backward (no autograd) so gradient buckets can be all-reduced over
RCCL while earlier layers are still running, whole step capturable in a hipGraph.

PyTorch supplies device memory, streams and torch.distributed; no torch arithmetic on the hot path.
"""
import math
import os
from collections import OrderedDict
from dataclasses import dataclass

import torch

from . import ops

F32 = torch.float32
VIS_DIM = 2048            # hard-coded in the reference (common/fast_rcnn.py:107, resnet_vlbert_for_pretraining.py:25)
TAG_EMBED, TAG_DOWNSAMPLE = 1000, 1001


def _ru(x, m):
    return (x + m - 1) // m * m


@dataclass
class ModelConfig:
    """NETWORK.VLBERT keys the hot path reads (pretrain/function/config.py:86-122)."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    vocab_size: int = 30522
    max_position_embeddings: int = 512
    type_vocab_size: int = 3
    visual_region_classes: int = 1601
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    obj_downsample_dropout: float = 0.1
    multitask: bool = False      # ResNetVLBERTForPretrainingMultitask: text-only auxiliary samples (+1 parameter)
    with_pooler: bool = False    # BertPooler on the first token (modeling.py:424-436)
    with_rel_loss: bool = False  # relationship head on the pooled output + its CE loss (needs with_pooler)
    e2e: bool = False            # IMAGE_FEAT_PRECOMPUTED false: ResNet trunk -> ROIAlign -> layer4 head on the device (vision.py)
    image_num_layers: int = 101  # NETWORK.IMAGE_NUM_LAYERS (50 / 101 / 152)
    image_frozen_stages: tuple = (1, 2)   # NETWORK.IMAGE_FROZEN_BACKBONE_STAGES (BatchNorm is always frozen: IMAGE_FROZEN_BN)

    def validate(self):
        H, nh = self.hidden_size, self.num_attention_heads
        if H % 64 or H // nh != 64:
            raise ValueError("engine supports head dim 64 and hidden_size % 64 == 0 (got H=%d, heads=%d)" % (H, nh))
        if self.intermediate_size % 64:
            raise ValueError("intermediate_size must be a multiple of 64")
        if self.with_rel_loss and not self.with_pooler:
            raise ValueError("with_rel_loss needs with_pooler (the relationship head reads the pooled output)")


def param_layout(cfg):
    """Ordered {state_dict name: shape} -- the reference's key names (SURVEY.md §8b state-dict contract).
    Order matters: q/k/v weights (and biases) are adjacent so the fused [3H,H] QKV operand is a plain
    view; the two object linguistic vectors are adjacent so they form a [2,H] table."""
    H, I, V, C = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.visual_region_classes
    s = OrderedDict()
    s["vlbert.word_embeddings.weight"] = (V, H)
    s["vlbert.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s["vlbert.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s["vlbert.end_embedding.weight"] = (1, H)
    s["object_linguistic_embeddings.weight"] = (1, H)
    s["object_mask_word_embedding.weight"] = (1, H)
    if cfg.multitask:
        s["aux_text_visual_embedding.weight"] = (1, H)     # resnet_vlbert_for_pretraining_multitask.py:28
    s["object_mask_visual_embedding.weight"] = (1, VIS_DIM)
    s["image_feature_extractor.obj_downsample.1.weight"] = (H, 2 * VIS_DIM)
    s["image_feature_extractor.obj_downsample.1.bias"] = (H,)
    for ln in ("embedding_LayerNorm", "visual_ln_text", "visual_ln_object"):
        s["vlbert.%s.weight" % ln] = (H,)
        s["vlbert.%s.bias" % ln] = (H,)
    for l in range(cfg.num_hidden_layers):
        p = "vlbert.encoder.layer.%d." % l
        for n in ("query", "key", "value"):
            s[p + "attention.self.%s.weight" % n] = (H, H)
        for n in ("query", "key", "value"):
            s[p + "attention.self.%s.bias" % n] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    if cfg.with_pooler:
        s["vlbert.pooler.dense.weight"] = (H, H)
        s["vlbert.pooler.dense.bias"] = (H,)
    if cfg.with_rel_loss:       # [sic] the typo is part of the reference's key (common/visual_linguistic_bert.py:322)
        s["vlbert.relationsip_head.caption_image_relationship.weight"] = (2, H)
        s["vlbert.relationsip_head.caption_image_relationship.bias"] = (2,)
    p = "vlbert.mlm_head.predictions."
    s[p + "transform.dense.weight"] = (H, H)
    s[p + "transform.dense.bias"] = (H,)
    s[p + "transform.LayerNorm.weight"] = (H,)
    s[p + "transform.LayerNorm.bias"] = (H,)
    s[p + "bias"] = (V,)
    s["vlbert.mvrc_head.transform.dense.weight"] = (H, H)
    s["vlbert.mvrc_head.transform.dense.bias"] = (H,)
    s["vlbert.mvrc_head.region_cls_pred.weight"] = (C, H)
    s["vlbert.mvrc_head.region_cls_pred.bias"] = (C,)
    if cfg.e2e:      # trainable convolution weights of the vision path, LAST (their gradients complete last; parallel.GradBuckets)
        from .vision import vision_param_layout
        s.update(vision_param_layout(cfg.image_num_layers, cfg.image_frozen_stages))
    return s


TIED_DECODER_KEY = "vlbert.mlm_head.predictions.decoder.weight"   # alias of word_embeddings.weight (modeling.py:463-466)


class FlatParams:
    """One contiguous buffer per role (fp32 master / fp32 grad / Adam m / Adam v / bf16 copy); every
    tensor starts on a 64-element boundary (256 B fp32, 128 B bf16)."""

    def __init__(self, cfg, device, align=64):
        """align: the buffers' length is rounded up to it (parallel.shard_alignment(world): every data-parallel bucket then splits
        evenly over the ranks; the zero tail belongs to no parameter)."""
        self.shapes = param_layout(cfg)
        self.offsets = OrderedDict()
        off = 0
        for name, shape in self.shapes.items():
            self.offsets[name] = off
            off = _ru(off + math.prod(shape), 64)
        off = _ru(off, max(64, int(align)))
        self.numel = off
        self.master = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        self.m = torch.zeros(off, dtype=F32, device=device)
        self.v = torch.zeros(off, dtype=F32, device=device)
        self.w16 = torch.zeros(off, dtype=ops.BF16, device=device)

    def view(self, buf, name, shape=None, span=1):
        """View of `name` in `buf`; span>1 extends over the following adjacent tensors."""
        names = list(self.shapes)
        i = names.index(name)
        n = sum(math.prod(self.shapes[k]) for k in names[i:i + span])
        if span > 1:   # adjacency requires no padding in between
            assert self.offsets[names[i + span - 1]] + math.prod(self.shapes[names[i + span - 1]]) - self.offsets[name] == n
        t = buf[self.offsets[name]:self.offsets[name] + n]
        return t.view(*(shape if shape is not None else self.shapes[name]))

    def named(self, buf):
        return OrderedDict((k, self.view(buf, k)) for k in self.shapes)


class PretrainEngine:
    def __init__(self, cfg, B, T, R, device="cuda:0", train=True, lr=1e-4, weight_decay=1e-4, max_grad_norm=10.0,
                 betas=(0.9, 0.999), eps=1e-6, seed=1234, grad_accum=1, process_group=None, keep_logits=False, flat=None,
                 B_aux=0, core=False, core_heads=True, core_sequence=False, lr_schedule=None, warmup_steps=0, t_total=0,
                 image_size=None, dp_mode="default", dp_wire="default", loss_scale=None, encoder_fp32=False):
        cfg.validate()
        if cfg.e2e and (core or image_size is None):
            raise ValueError("e2e needs image_size=(H, W) and a pretraining wrapper (plain or multitask; not the core module mode)")
        # lr_schedule: None (host sets lr) | "constant" | "warmup_constant" | "triangle" (WarmupLinearSchedule,
        # pretrain/function/train.py:316-320) -- evaluated on the device from the step counter each optimizer step.
        kinds = {None: None, "constant": ops.LR_CONSTANT, "warmup_constant": ops.LR_WARMUP_CONSTANT,
                 "triangle": ops.LR_WARMUP_LINEAR, "warmup_linear": ops.LR_WARMUP_LINEAR}
        if lr_schedule not in kinds:
            raise ValueError("lr_schedule must be one of %s" % sorted(k for k in kinds if k))
        if kinds[lr_schedule] == ops.LR_WARMUP_LINEAR and t_total <= warmup_steps:
            raise ValueError("triangle schedule needs t_total > warmup_steps")
        self.lr_kind, self.base_lr, self.warmup_steps, self.t_total = kinds[lr_schedule], lr, warmup_steps, t_total
        self.cfg, self.B, self.T, self.R = cfg, B, T, R
        # Static loss scale (the reference's TRAIN.FP16_LOSS_SCALE, pretrain/function/train.py:345-352): d(loss)/d(logits) leaves the
        # loss kernels multiplied by it, every 16-bit gradient tensor and the flat fp32 gradient carry it, and the optimizer's
        # grad_scale divides it out (clip norm included).  1 in the bf16 build (fp32 exponent range); 4096 by default in the fp16
        # build (VLB_PRECISION=f16): d(logits) ~ 1/n_valid ~ 4e-4 at batch 256 would sit in fp16's subnormals otherwise.  Engines behind
        # the autograd mirrors (flat=... / core=True) hand their parameter gradients to torch unscaled, so they default to 1.
        if loss_scale is None:
            loss_scale = 4096.0 if (ops.BF16 == torch.float16 and flat is None and not core) else 1.0
        self.loss_scale = float(loss_scale)
        # multitask: B_aux text-only samples are appended after the B image-caption samples; they have no
        # objects, their text-visual embedding is the learned aux_text_visual_embedding and their MLM loss is
        # accounted separately (resnet_vlbert_for_pretraining_multitask.py:96-290).  T = max(caption, aux) length.
        # core=True: the engine is driven through the module API of common/visual_linguistic_bert.py:95-171,346-380 --
        # per-token text-visual embeddings and [visual || linguistic] object embeddings come in as tensors, logits go
        # out, d(logits) comes back in (set_core_inputs / forward_core / backward_core); the FastRCNN front end, the
        # losses and the wrapper-level parameters are not used.
        self.core = core
        self.with_heads = core_heads or not core      # core_heads=False: hidden states out, d(hidden states) in
        self.seq_out = bool(core and core_sequence and not core_heads)   # the packed [B,S,H] sequence instead of text / object splits
        self.Ba = B_aux
        if B_aux and not cfg.multitask:
            raise ValueError("B_aux > 0 needs ModelConfig(multitask=True)")
        self.Bt = B + B_aux
        self.S = T + R + 1
        if self.S > 256:
            raise ValueError("sequence %d+%d+1 > 256: the fused attention kernels handle S <= 256" % (T, R))
        self.dev = torch.device(device)
        self.train = train
        self.grad_accum = grad_accum
        self.pg = process_group
        self.keep_logits = keep_logits
        H, I, V, C, L = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.visual_region_classes, cfg.num_hidden_layers
        self.M, self.BT, self.BR = self.Bt * self.S, self.Bt * T, B * R
        self.Mp, self.BTp, self.BRp = _ru(self.M, 64), _ru(self.BT, 64), _ru(self.BR, 64)
        self.Vp, self.Cp = _ru(V, 64), _ru(C, 64)
        d = self.dev
        import torch.distributed as dist
        world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self._own_flat = flat is None
        if flat is None:
            from .parallel import shard_alignment
            flat = FlatParams(cfg, d, align=shard_alignment(world))
        self.P = flat                                                # (`flat` given: storage shared with an nn.Module mirror)
        P = self.P
        self.w16 = P.named(P.w16)
        self.w32 = P.named(P.master)
        self.g32 = P.named(P.grad)
        self.vision = None
        if cfg.e2e:
            from .vision import VisionStack
            self.vision = VisionStack(B, image_size[0], image_size[1], R, device=d, num_layers=cfg.image_num_layers,
                                      frozen_stages=cfg.image_frozen_stages, storage=lambda n, shape: (self.w32[n], self.g32[n]))
            self.in_image = torch.zeros((B, 3, image_size[0], image_size[1]), dtype=F32, device=d)

        def zb(*s):
            return torch.zeros(s, dtype=ops.BF16, device=d)

        def zf(*s):
            return torch.zeros(s, dtype=F32, device=d)

        # bf16 transposed weights for dgrad GEMMs (dX = dY . (W^T)^T); refreshed after every optimizer step
        self.wT = {}
        self._tbatch = None
        self._zero_small, self._fresh_grads = None, False
        for l in range(L):
            p = "vlbert.encoder.layer.%d." % l
            self.wT[p + "qkv"] = zb(H, 3 * H)
            self.wT[p + "attention.output.dense.weight"] = zb(H, H)
            self.wT[p + "intermediate.dense.weight"] = zb(H, I)
            self.wT[p + "output.dense.weight"] = zb(I, H)
        self.wT["vlbert.mlm_head.predictions.transform.dense.weight"] = zb(H, H)
        self.wT["vlbert.word_embeddings.weight"] = zb(H, self.Vp)
        self.wT["vlbert.mvrc_head.transform.dense.weight"] = zb(H, H)
        self.wT["vlbert.mvrc_head.region_cls_pred.weight"] = zb(H, self.Cp)
        self.wT["image_feature_extractor.obj_downsample.1.weight"] = zb(2 * VIS_DIM, H)
        if cfg.with_pooler:
            self.wT["vlbert.pooler.dense.weight"] = zb(H, H)
        if cfg.with_rel_loss:
            self.wT["vlbert.relationsip_head.caption_image_relationship.weight"] = zb(H, 64)   # K padded to one 64-wide tile

        # device-resident step state
        # odd (the advance kernel keeps it odd) and injective in `seed`: `seed | 1` collapsed ranks 2k / 2k+1 of a
        # seed = RNG_SEED + rank launch onto one dropout stream
        self.seed = torch.tensor([((seed * 2 + 1) & 0x7FFFFFFF)], dtype=torch.int32, device=d)
        self.adam = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, 0.0, max_grad_norm, 0.0], dtype=F32, device=d)
        self.hyper = dict(betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))   # as given (checkpoints)
        self.sumsq_ws = zf(2048)  # per-block partial sums of the gradient norm
        self.losses = zf(4)       # mlm (with visual content), mvrc, mlm (aux text), relationship
        self.counts = zf(4)       # n_valid mlm, n_valid mvrc, n_valid mlm aux

        # static batch buffers (graph-capturable: the host copies new batches into them)
        self.in_boxes = zf(B, R, 4 + VIS_DIM)
        self.in_im_info = zf(B, 5)
        Bt = self.Bt
        self.in_text = torch.zeros((Bt, T), dtype=torch.int64, device=d)
        self.in_mlm_labels = torch.full((Bt, T), -1, dtype=torch.int64, device=d)
        self.in_mvrc_ops = torch.zeros((B, R), dtype=torch.int64, device=d)
        self.in_mvrc_labels = zf(B, R, C)
        self.in_rel_label = torch.zeros((B,), dtype=torch.int64, device=d)
        self.text_mask = torch.zeros((Bt, T), dtype=torch.uint8, device=d)
        self.box_mask = torch.zeros((Bt, R), dtype=torch.uint8, device=d)     # aux rows stay 0: no objects
        i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=d)
        self.lay = dict(code=i32(Bt, self.S), text_len=i32(Bt), nobj=i32(Bt), text_rows=i32(Bt, T), obj_rows=i32(Bt, R),
                        attn_mask=zf(Bt, self.S))

        # activations
        M, BT, BR, S, nh = self.M, self.BT, self.BR, self.S, cfg.num_attention_heads
        # Operands of the per-layer weight gradients are allocated with their row count rounded up to 128 (the pad rows stay zero:
        # every kernel writes M rows) and used through [:M] views; the grouped large-tile wgrad (gemm_tn8.hip, K tiles of 128 rows)
        # gets the padded tensors, so it also serves the per-GPU batches of a strong-scaling run (M = 3232 at 32 samples).
        self.Mp128 = _ru(M, 128)
        self._row_padded = {}

        def zbm(cols):
            full = zb(self.Mp128, cols)
            view = full[:M]
            self._row_padded[view.data_ptr()] = full
            return view
        self.a_ds = zb(BR, 2 * VIS_DIM)
        self.obj_reps = zb(BR, H)
        self.objvis, self.st_objvis = zb(BR, H), zf(BR, 2)
        self.textvis, self.st_textvis = zb(Bt, H), zf(Bt, 2)
        self.emb_pre, self.st_emb = zb(M, H), zf(M, 2)
        self.X = [zbm(H) for _ in range(L + 1)]
        self.QKV = [zb(M, 3 * H) for _ in range(L)]
        self.CTX = [zbm(H) for _ in range(L)]
        self.LSE = [zf(Bt, nh, S) for _ in range(L)]
        # Residual stream precision (DESIGN.md "precision"): the pre-LayerNorm sums Z1 / Z2 are kept in fp16 (3 more mantissa bits
        # than bf16, same bytes; |Z| = O(1..10)) and the residual a sublayer adds is the previous LayerNorm's output re-materialised
        # in fp32 from (Z, mean, rstd, gamma, beta) inside the GEMM epilogue, so nothing on the residual path is ever rounded to
        # bf16 -- only the GEMM operands are.  At 12 layers this takes the logits' error against the fp32 reference from 1.3e-2 to
        # ~6e-3 (relative Frobenius; 5e-3 of it is the bf16 rounding of the weights).  VLB_RESIDUAL_STREAM=bf16: the old behaviour.
        import os as _os0
        self.hp_res = _os0.environ.get("VLB_RESIDUAL_STREAM", "f16ln") != "bf16"
        zdt = torch.float16 if self.hp_res else ops.BF16
        zz = lambda *s: torch.zeros(s, dtype=zdt, device=d)
        self.Z1, self.ST1, self.Y1 = [zz(M, H) for _ in range(L)], [zf(M, 2) for _ in range(L)], [zbm(H) for _ in range(L)]
        self.U, self.G = [zb(M, I) for _ in range(L)], [zbm(I) for _ in range(L)]
        self.Z2, self.ST2 = [zz(M, H) for _ in range(L)], [zf(M, 2) for _ in range(L)]
        self.text_out, self.obj_out = zb(BT, H), zb(BR, H)
        self.mlm_u, self.mlm_g, self.mlm_h, self.st_mlm = zb(BT, H), zb(BT, H), zb(BT, H), zf(BT, 2)
        self.mlm_logits = zb(BT, self.Vp)
        self.mvrc_u, self.mvrc_g = zb(BR, H), zb(BR, H)
        self.mvrc_logits = zb(BR, self.Cp)
        self.mvrc_tsum = zf(BR)
        if cfg.with_pooler:
            self.pooled, self.d_pooled, self.d_pool_pre = zb(B, H), zb(B, H), zb(B, H)
        if cfg.with_rel_loss:
            self.rel_logits = zb(B, 64)                   # 2 logits, row padded to 64 (pad columns stay zero)
            self.rel_logits_copy = zb(B, 64) if keep_logits else None
        self.mlm_logits_copy = zb(BT, self.Vp) if keep_logits else None
        # MLM head compaction (DESIGN.md "MLM head"): ~85 % of the text positions carry no label -- their logits are never read by the
        # loss and their d(logits) rows are exactly zero -- so transform -> LayerNorm -> decoder and the head's whole backward run on
        # the labelled rows only, gathered into the first `mlm_cap` rows of the same buffers.  Capacity is a contract with the data
        # pipeline (BERT masking labels 15 % of the valid tokens; at 16384 positions 20 % is 17 standard deviations away): 20 % of the B*T
        # positions rounded up to 256; a batch that exceeds it raises (the
        # device flag is checked at loss_values(); labels handed over as CPU tensors are counted exactly and such a batch simply
        # takes the full path).  Off with keep_logits (the module mirrors return every logit) and in module-API mode.
        import os as _os1
        cap = min(self.BTp, max(256, _ru(int(math.ceil(0.20 * BT)), 256)))
        want = _os1.environ.get("VLB_MLM_COMPACT", "1") != "0" and not keep_logits and not core and cap < BT
        self.mlm_cap = cap if want else None
        self._mlm_compact_now = want
        if want:
            self.sel_pos = torch.full((cap,), -1, dtype=torch.int32, device=d)
            self.sel_src = torch.full((cap,), -1, dtype=torch.int32, device=d)
            self.labels_c = torch.full((cap,), -1, dtype=torch.int64, device=d)
            self.mlm_overflow = torch.zeros((1,), dtype=torch.int32, device=d)
            self.d_text_out_c = zb(cap, H)
        self.mvrc_logits_copy = zb(BR, self.Cp) if keep_logits else None

        # backward scratch
        self.dXa, self.dXb = zb(M, H), zb(M, H)
        # The four weight gradients of a layer go out as ONE grouped launch at the end of the layer's backward (side stream), so their
        # gradient operands (LN2 / LN1 outputs through dropout, dU, dQKV) stay alive until then and are double-buffered by layer
        # parity: the next layer writes the other set while the group still reads this one.
        self.dZ = zb(M, H)
        # Round 6: the weight gradients of TWO layers go out as one table launch of full-K work items (216 items for 256 CUs like the
        # grouped launch of one layer, but no K slices: no fp32 slabs, no reduce launch) -- the upper layer of a pair keeps its operands
        # until the lower one is done, so there are four sets (layer & 3) instead of two.  VLB_WGRAD_PAIRS=0: one grouped launch per layer.
        self._pairs = os.environ.get("VLB_WGRAD_PAIRS", "1") != "0" and os.environ.get("VLB_WGRAD_TN", "1") != "0"
        self._pair_tables = {}
        nset = 4 if self._pairs else 2
        self.dD2, self.dD1 = [zbm(H) for _ in range(nset)], [zbm(H) for _ in range(nset)]
        self.dU2 = [zbm(I) for _ in range(nset)]
        self.dQKV2 = [zbm(3 * H) for _ in range(nset)]
        self.dCTX = zb(M, H)
        self.tG = zb(max(3 * H, I), self.Mp)       # transposed gradients (zero padded columns persist)
        self.tA = zb(max(H, I), self.Mp)           # transposed activations
        self.tG_bt, self.tA_bt = zb(max(self.Vp, H), self.BTp), zb(H, self.BTp)
        self.tG_br, self.tA_br = zb(max(self.Cp, H), self.BRp), zb(max(H, 2 * VIS_DIM), self.BRp)
        self.d_mlm_h, self.d_mlm_g, self.d_mlm_u = zb(BT, H), zb(BT, H), zb(BT, H)
        self.d_text_out, self.d_obj_out = zb(BT, H), zb(BR, H)
        self.d_mvrc_u = zb(BR, H)
        self.d_textvis, self.d_objvis = zf(Bt, H), zf(BR, H)
        self.d_obj_reps = zf(BR, H)
        self.d_yds = zb(BR, H)
        self.d_afeat = zb(BR, VIS_DIM)
        if core:
            self.in_text_type = torch.zeros((Bt, T), dtype=torch.int64, device=d)
            self.tv_in, self.ovl_in = zb(BT, H), zb(BR, 2 * H)          # inputs as bf16: text_visual [B,T,H], object_vl [B,R,2H]
            self.tv_n, self.st_tv = zb(BT, H), zf(BT, 2)                # visual_ln_text per token
            self.d_tv_n, self.d_ol = zf(BT, H), zf(BR, H)               # d(LN(text_visual)), d(object linguistic half)
            self.d_tv_in, self.d_ovl_in = zb(BT, H), zb(BR, 2 * H)      # gradients handed back to autograd
        # fp32 slab workspace for split-K weight gradients (largest request over this engine's wgrad shapes)
        need = [ops.wgrad_workspace_floats(n, k, rp) for n, k, rp in
                ((3 * H, H, self.Mp), (H, H, self.Mp), (I, H, self.Mp), (H, I, self.Mp), (V, H, self.BTp), (H, H, self.BTp),
                 (C, H, self.BRp), (H, H, self.BRp), (H, 2 * VIS_DIM, self.BRp))]
        need_dec = ops.wgrad_workspace_floats(self.BT, H, self.Vp)       # tied-decoder dgrad (K = vocabulary) at small batch
        need.append(need_dec)
        need.append(3 * (3 * H * H + H * H + 2 * I * H) + 64)             # grouped launch of a layer's four gradients, <= 3 K slices
        self.wg_ws = zf(max(max(need), 4))
        # Weight gradients run on a second stream: they only feed the optimizer, while the dgrad chain is the critical
        # path, and at small per-GPU batch neither fills the chip (312 + 432 workgroups for 512 slots at B = 32).
        # Hazards: a wgrad reads the gradient buffer the main stream produced (event after the producer) and the main
        # stream must not overwrite that buffer before the wgrad is done (event recorded after it, waited by the
        # next writer -- a full layer later thanks to the dD / dDb double buffer).  VLB_WGRAD_STREAM=0 serialises.
        import os as _os
        self.side = torch.cuda.Stream(device=d) if (d.type == "cuda" and _os.environ.get("VLB_WGRAD_STREAM", "1") != "0") else None
        self.wg_ws_main = zf(max(need_dec, 4)) if self.side is not None else self.wg_ws    # decoder dgrad split-K (main stream)
        self._pending = {}
        self.ln_ws = zf(ops.ln_bwd_workspace_floats(H))     # per-workgroup partial dgamma/dbeta sums of the LayerNorm backward
        # The encoder's LayerNorm backwards (2 per layer + the MLM head's) leave their partial sums in a workspace slice of their
        # own; one batched launch (ops.ln_param_finalize_batch) adds them into the gradients when those are next needed -- a
        # bucket's all-reduce, or the end of backward -- instead of a 7-10 us finalize launch behind each of the 25 calls.
        n_slices = min(2 * L + 1, 32) if _os0.environ.get("VLB_LN_DEFER", "1") != "0" else 0
        self._ln_slices = list(zf(n_slices, ops.ln_bwd_workspace_floats(H)).unbind(0)) if n_slices else []
        self._ln_pending = []
        self.graph = None
        self._weights_dirty = True
        self.use_tn_wgrad = os.environ.get("VLB_WGRAD_TN", "1") != "0"
        self.buckets = None
        # sharded optimizer: the weight gather of the last update may still be in flight (forward waits per bucket); the transposed /
        # folded weight copies are refreshed once it has landed (backward / the next forward of the vision path)
        self._wT_stale = self._gather_pending = self._vision_stale = False
        self._dp_hook = None         # set by parallel.DistributedDataParallel on the engines of a module mirror (shared flat buffers)
        from .parallel import force_exchange
        if (world > 1 or force_exchange()) and self._own_flat:      # (VLB_DP_FORCE_EXCHANGE: the exchange as identities in a world of one)
            from .parallel import GradBuckets
            vstart = min((o for n, o in self.P.offsets.items() if n.startswith("image_feature_extractor.") and
                          not n.startswith("image_feature_extractor.obj_downsample")), default=None)
            self.buckets = GradBuckets(self.P.grad, self.P.offsets, self.P.numel, L, group=process_group, vision_start=vstart,
                                       wire_dtype=dp_wire, mode=dp_mode)
            if self.buckets.sharded:
                # this rank's slice of every bucket: clip-norm partial + AdamW run on those only, one launch each (ops.ShardRanges);
                # wshard = the compact bf16 image of the updated slices the weight all-gather distributes (parallel.py)
                self.shard_tbl = ops.ShardRanges(self.buckets.owned_rows(), d)
                self.wshard = torch.zeros(self.P.numel // world, dtype=ops.BF16, device=d)
                # the tensors forward / backward read from the fp32 MASTER (not from the gathered 16-bit copy) are replicated after
                # every owner-only update (parallel.GradBuckets.set_replicated_fp32)
                self.buckets.set_replicated_fp32(self._fp32_read_ranges())
        # fp32 compute mode of the encoder (the reference's TRAIN.FP16: false configurations; encoder_f32.py): the layers run on fp32
        # tensors and the fp32 MASTER weights, the embedding side and the heads stay on the 16-bit kernels (use the fp16 build)
        self.enc32 = None
        if encoder_fp32:
            if self.buckets is not None and self.buckets.sharded:
                raise ValueError("encoder_fp32 reads the fp32 master weights on every rank: use dp_mode='allreduce' (the sharded optimizer "
                                 "keeps them authoritative on the owner only)")
            from .encoder_f32 import EncoderF32
            self.enc32 = EncoderF32(self)

    # ------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: {reference state_dict name: tensor}.  The tied decoder key is accepted and ignored."""
        vis = self._vision_names()
        for name in self.P.shapes:
            if name in vis:
                continue
            if name not in sd:
                raise KeyError("missing parameter %s" % name)
            self.w32[name].copy_(sd[name].to(F32))
        if self.vision is not None:     # conv weights ([O,I,KH,KW] in the reference) + BatchNorm tensors + frozen stages
            self.vision.load_state_dict(sd)
        self._weights_dirty = True

    def _fp32_read_ranges(self):
        """[(lo, hi)] of the flat buffer for every parameter the kernels read from the fp32 master: all 1-D tensors (Linear biases,
        LayerNorm gamma / beta, the decoder bias) and the mask visual embedding (obj_prep reads it as fp32).  Matrices and embedding
        tables are read through the 16-bit working copy; the e2e convolution weights travel as master slices with their buckets."""
        vis = self._vision_names()
        return [(self.P.offsets[n], self.P.offsets[n] + math.prod(sh)) for n, sh in self.P.shapes.items()
                if n not in vis and (len(sh) == 1 or n == "object_mask_visual_embedding.weight")]

    def _vision_names(self):
        if self.vision is None:
            return ()
        return {"image_feature_extractor." + k + ".weight" for k, c in self.vision.convs.items() if c.trainable}

    def state_dict(self):
        """With the sharded data-parallel optimizer the fp32 master is authoritative on the owning rank only: the slices are gathered
        first, so this is a COLLECTIVE there (call it on every rank, save on one)."""
        if self.buckets is not None and self.buckets.sharded:
            self.buckets.gather_master(self.P.master)
        vis = self._vision_names()
        sd = OrderedDict((k, v.detach().clone()) for k, v in self.w32.items() if k not in vis)
        sd[TIED_DECODER_KEY] = sd["vlbert.word_embeddings.weight"]
        if self.vision is not None:
            sd.update(self.vision.state_dict())
        return sd

    def sync_weights(self):
        """fp32 master -> bf16 working copy + transposed copies (after load / external modification)."""
        if self.buckets is not None:
            self.buckets.wait_params("all")
        self._wT_stale = self._gather_pending = self._vision_stale = False
        ops.cast_f32_bf16(self.P.master, self.P.w16)
        self._refresh_transposes()
        if self.vision is not None:
            self.vision.refresh_weights()
        self._weights_dirty = False

    def _refresh_transposes(self):
        """bf16 W^T copies for the dgrad GEMMs (dX = dY W as an NT product): all ~50 of them in one launch."""
        if self._tbatch is None:
            H, L = self.cfg.hidden_size, self.cfg.num_hidden_layers
            pairs = []
            for l in range(L):
                p = "vlbert.encoder.layer.%d." % l
                wqkv = self.P.view(self.P.w16, p + "attention.self.query.weight", (3 * H, H), span=3)
                pairs.append((wqkv, self.wT[p + "qkv"]))
                for n in ("attention.output.dense.weight", "intermediate.dense.weight", "output.dense.weight"):
                    pairs.append((self.w16[p + n], self.wT[p + n]))
            extra = [n for n in ("vlbert.pooler.dense.weight", "vlbert.relationsip_head.caption_image_relationship.weight")
                     if n in self.wT]
            for n in ["vlbert.mlm_head.predictions.transform.dense.weight", "vlbert.word_embeddings.weight",
                      "vlbert.mvrc_head.transform.dense.weight", "vlbert.mvrc_head.region_cls_pred.weight",
                      "image_feature_extractor.obj_downsample.1.weight"] + extra:
                pairs.append((self.w16[n], self.wT[n]))
            self._tbatch = ops.TransposeBatch(pairs, self.dev)
        self._tbatch.run()
        if self.enc32 is not None:
            self.enc32.refresh()

    # ------------------------------------------------------------------------------------------
    # batch
    # ------------------------------------------------------------------------------------------
    def set_batch(self, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels, aux_text=None,
                  aux_mlm_labels=None, image=None, mask_raw_pixels=False):
        """Copies a collated batch (pretrain/data/collate_batch.py layout) into the static device buffers
        and derives the masks exactly as resnet_vlbert_for_pretraining.py:106,134 does
        (box_mask = boxes[:,:,0] > -1.5 ; text_mask = text > 0).  `relationship_label` is only read
        with ModelConfig(with_rel_loss=True) (WITH_REL_LOSS is false in the north-star configuration).
        mask_raw_pixels (e2e): the image arrives UNMASKED and the pixels of the regions with mvrc_op == 1 are zeroed here, on the
        device copy -- the step the reference's dataset does per sample (conceptual_captions.py:201-206, MASK_RAW_PIXELS)."""
        B = self.B
        if self.mlm_cap is not None:    # labels still on the host: count them exactly (no sync) and take the full path if they do not fit
            self._mlm_compact_now = True
            if not mlm_labels.is_cuda and (aux_mlm_labels is None or not aux_mlm_labels.is_cuda):
                n_lab = int((mlm_labels >= 0).sum()) + (int((aux_mlm_labels >= 0).sum()) if aux_mlm_labels is not None else 0)
                self._mlm_compact_now = n_lab <= self.mlm_cap
        if self.vision is not None:     # e2e: `image` [B,3,H,W] fp32 (mean-subtracted, collate_batch.py), boxes [B,R,4]; features come from the CNN
            if image is None:
                raise ValueError("engine built with e2e=True needs image=")
            self.in_image.copy_(image, non_blocking=True)
        # the collator pads to the batch's own largest box count (pretrain/data/collate_batch.py:21,39): fewer slots than this engine's R
        # are completed with its padding markers (boxes -2, mvrc_ops 0, soft labels 0)
        Rb = boxes.shape[1]
        if Rb > self.R:
            raise ValueError("batch has %d box slots, engine was built for %d" % (Rb, self.R))
        if Rb < self.R:
            self.in_boxes[:, Rb:].fill_(-2.0)
            self.in_mvrc_ops[:, Rb:].zero_()
            self.in_mvrc_labels[:, Rb:].zero_()
        if self.vision is not None:
            self.in_boxes[:, :Rb, :4].copy_(boxes[:, :, :4], non_blocking=True)
        else:
            self.in_boxes[:, :Rb].copy_(boxes, non_blocking=True)
        self.in_im_info[:, :2].copy_(im_info[:, :2], non_blocking=True)      # (width, height); the datasets append 2-3 more columns
        if self.Ba or text.shape[1] != self.T:
            self.in_text.zero_()
            self.in_mlm_labels.fill_(-1)
        self.in_text[:B, :text.shape[1]].copy_(text, non_blocking=True)
        self.in_mlm_labels[:B, :mlm_labels.shape[1]].copy_(mlm_labels, non_blocking=True)
        if self.Ba:
            if aux_text is None or aux_text.shape[0] != self.Ba:
                raise ValueError("engine built with B_aux=%d needs aux_text of that many rows" % self.Ba)
            self.in_text[B:, :aux_text.shape[1]].copy_(aux_text, non_blocking=True)
            self.in_mlm_labels[B:, :aux_mlm_labels.shape[1]].copy_(aux_mlm_labels, non_blocking=True)
        self.in_mvrc_ops[:, :Rb].copy_(mvrc_ops, non_blocking=True)
        self.in_mvrc_labels[:, :Rb].copy_(mvrc_labels, non_blocking=True)
        if self.cfg.with_rel_loss:
            self.in_rel_label.copy_(relationship_label, non_blocking=True)
        torch.gt(self.in_text, 0, out=self.text_mask.view(torch.bool))
        torch.gt(self.in_boxes[:, :, 0], -1.5, out=self.box_mask[:B].view(torch.bool))
        if mask_raw_pixels:
            if self.vision is None:
                raise ValueError("mask_raw_pixels applies to the e2e configuration (precomputed features are masked by the "
                                 "object_mask_visual_embedding overwrite, resnet_vlbert_for_pretraining.py:114-117)")
            ops.mask_image_boxes(self.in_image, self.in_boxes, self.in_mvrc_ops)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _p(self, train):
        c = self.cfg
        if not train:
            return 0.0, 0.0, 0.0
        return c.hidden_dropout_prob, c.attention_probs_dropout_prob, c.obj_downsample_dropout

    def forward(self, train=None, gscale=1.0):
        train = self.train if train is None else train
        if self._weights_dirty:
            self.sync_weights()
        cfg, B, T, R, S, Bt, Ba = self.cfg, self.B, self.T, self.R, self.S, self.Bt, self.Ba
        H, I, V, C, L, nh = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.visual_region_classes, \
            cfg.num_hidden_layers, cfg.num_attention_heads
        p_h, p_a, p_ds = self._p(train)
        w16, w32, seed = self.w16, self.w32, self.seed
        self.losses.zero_()
        ops.seq_layout_into(self.text_mask, self.box_mask, S, self.lay)
        if self._gather_pending:    # sharded optimizer: the bf16 weights of the last update arrive bucket by bucket (parallel.gather_params)
            if self.vision is not None:
                self.buckets.wait_params("vision")
                if self._vision_stale:
                    self.vision.refresh_weights(trainable_only=True)
                    self._vision_stale = False
            self.buckets.wait_params("front")
        if self.core:
            self._front_core_fwd(p_h)
        else:
            self._front_pretrain_fwd(p_h, p_ds)
        self._encoder_heads_fwd(p_h, p_a)
        if not self.core:
            self._losses_fwd_bwd(gscale * self.loss_scale, True)

    def _front_core_fwd(self, p_h):
        """VisualLinguisticBert.embedding (common/visual_linguistic_bert.py:173-241) on caller-provided embeddings."""
        cfg, T, R, S, Bt = self.cfg, self.T, self.R, self.S, self.Bt
        H = cfg.hidden_size
        w16, w32, seed = self.w16, self.w32, self.seed
        ops.layernorm_fwd(self.tv_in, w32["vlbert.visual_ln_text.weight"], w32["vlbert.visual_ln_text.bias"], self.tv_n, self.st_tv)
        ops.layernorm_fwd(self.ovl_in[:, :H], w32["vlbert.visual_ln_object.weight"], w32["vlbert.visual_ln_object.bias"], self.objvis,
                          self.st_objvis)
        ops.embed_fwd(self.lay, self.in_text, self.in_text_type, w16["vlbert.word_embeddings.weight"],
                      w16["vlbert.position_embeddings.weight"], w16["vlbert.token_type_embeddings.weight"],
                      w16["vlbert.end_embedding.weight"], self.tv_n, (T * H, H), self.objvis, (R * H, H),
                      self.ovl_in[:, H:], (R * 2 * H, 2 * H), None, w32["vlbert.embedding_LayerNorm.weight"],
                      w32["vlbert.embedding_LayerNorm.bias"], self.emb_pre, self.st_emb, self.X[0], Bt, T, R, S, H, drop_p=p_h,
                      seed=seed, tag=TAG_EMBED)

    def _front_pretrain_fwd(self, p_h, p_ds):
        cfg, B, T, R, S, Bt, Ba = self.cfg, self.B, self.T, self.R, self.S, self.Bt, self.Ba
        H = cfg.hidden_size
        w16, w32, seed = self.w16, self.w32, self.seed
        # --- e2e: ResNet trunk -> ROIAlign -> layer4 head -> avg-pool, written into the feature slots of in_boxes; the raw
        #     pixels are masked by the dataset, so no mask embedding is substituted (mask_visual_embed=None, :120-127) ----
        e2e = self.vision is not None
        if e2e:
            self.vision.forward(self.in_image, self.in_boxes)
        # --- FastRCNN: (coord || feature) -> Dropout -> Linear(4096->H) -> ReLU (common/fast_rcnn.py:165-175) -------
        ops.obj_prep_fwd(self.in_boxes, self.in_im_info, None if e2e else self.in_mvrc_ops.view(-1),
                         w32["object_mask_visual_embedding.weight"], self.a_ds, drop_p=p_ds, seed=seed, tag=TAG_DOWNSAMPLE)
        ops.gemm_nt(self.a_ds, w16["image_feature_extractor.obj_downsample.1.weight"], self.obj_reps,
                    bias=w32["image_feature_extractor.obj_downsample.1.bias"], act=ops.ACT_RELU)
        ops.zero_padded_rows(self.obj_reps, self.in_boxes)      # pad_sequence zeros (a padded box 0 feeds the text tokens' visual embedding)
        # --- visual LayerNorms + fused embedding ---------------------------------------------------------
        ops.layernorm_fwd(self.obj_reps, w32["vlbert.visual_ln_object.weight"], w32["vlbert.visual_ln_object.bias"], self.objvis,
                          self.st_objvis)
        reps0 = self.obj_reps.view(B, R * H)[:, :H]          # obj_reps[:, 0]  (text tags are all 0, :132-135)
        ops.layernorm_fwd(reps0, w32["vlbert.visual_ln_text.weight"], w32["vlbert.visual_ln_text.bias"], self.textvis[:B],
                          self.st_textvis[:B])
        if Ba:   # aux text-only samples: every token sees LN(aux_text_visual_embedding) (multitask.py:176)
            ops.layernorm_fwd(w16["aux_text_visual_embedding.weight"], w32["vlbert.visual_ln_text.weight"],
                              w32["vlbert.visual_ln_text.bias"], self.textvis[B:], self.st_textvis[B:], rows=Ba, ldx=0)
        ops.embed_fwd(self.lay, self.in_text, None, w16["vlbert.word_embeddings.weight"], w16["vlbert.position_embeddings.weight"],
                      w16["vlbert.token_type_embeddings.weight"], w16["vlbert.end_embedding.weight"], self.textvis, (H, 0),
                      self.objvis, (R * H, H), w16["object_linguistic_embeddings.weight"], (0, 0), self.in_mvrc_ops,
                      w32["vlbert.embedding_LayerNorm.weight"], w32["vlbert.embedding_LayerNorm.bias"], self.emb_pre, self.st_emb,
                      self.X[0], Bt, T, R, S, H, drop_p=p_h, seed=seed, tag=TAG_EMBED)

    def _encoder_heads_fwd(self, p_h, p_a):
        cfg, S, Bt = self.cfg, self.S, self.Bt
        H, V, C, L, nh = cfg.hidden_size, cfg.vocab_size, cfg.visual_region_classes, cfg.num_hidden_layers, cfg.num_attention_heads
        w16, w32, seed = self.w16, self.w32, self.seed
        # --- encoder -------------------------------------------------------------------------------------
        mask = self.lay["attn_mask"]
        stale = self._gather_pending
        if self.enc32 is not None:
            self.enc32.forward(p_h, p_a)
        for l in range(L if self.enc32 is None else 0):
            p = "vlbert.encoder.layer.%d." % l
            x = self.X[l]
            if stale:
                self.buckets.wait_params(l)
            wqkv = self.P.view(self.P.w16, p + "attention.self.query.weight", (3 * H, H), span=3)
            bqkv = self.P.view(self.P.master, p + "attention.self.query.bias", (3 * H,), span=3)
            ops.gemm_nt(x, wqkv, self.QKV[l], bias=bqkv)
            ops.attention_fwd(self.QKV[l], mask, self.CTX[l], self.LSE[l], Bt, S, H, nh, drop_p=p_a, seed=seed, tag=l * 8 + 0)
            if self.hp_res and l > 0:    # residual = LayerNorm(Z2[l-1]) in fp32 (layer 0: the bf16 embedding output, one rounding)
                pp = "vlbert.encoder.layer.%d." % (l - 1)
                res_kw = dict(res=self.Z2[l - 1], res_ln=(self.ST2[l - 1], w32[pp + "output.LayerNorm.weight"], w32[pp + "output.LayerNorm.bias"]))
            else:
                res_kw = dict(res=x)
            ops.gemm_nt(self.CTX[l], w16[p + "attention.output.dense.weight"], self.Z1[l], bias=w32[p + "attention.output.dense.bias"],
                        drop_p=p_h, seed=seed, tag=l * 8 + 1, **res_kw)
            ops.layernorm_fwd(self.Z1[l], w32[p + "attention.output.LayerNorm.weight"], w32[p + "attention.output.LayerNorm.bias"],
                              self.Y1[l], self.ST1[l])
            ops.gemm_nt(self.Y1[l], w16[p + "intermediate.dense.weight"], self.G[l], bias=w32[p + "intermediate.dense.bias"],
                        act=ops.ACT_GELU_D, pre=self.U[l])
            if self.hp_res:
                res_kw = dict(res=self.Z1[l], res_ln=(self.ST1[l], w32[p + "attention.output.LayerNorm.weight"],
                                                      w32[p + "attention.output.LayerNorm.bias"]))
            else:
                res_kw = dict(res=self.Y1[l])
            ops.gemm_nt(self.G[l], w16[p + "output.dense.weight"], self.Z2[l], bias=w32[p + "output.dense.bias"],
                        drop_p=p_h, seed=seed, tag=l * 8 + 2, **res_kw)
            ops.layernorm_fwd(self.Z2[l], w32[p + "output.LayerNorm.weight"], w32[p + "output.LayerNorm.bias"], self.X[l + 1],
                              self.ST2[l])
        # --- heads ---------------------------------------------------------------------------------------
        if stale:
            self.buckets.wait_params("heads")
            self._gather_pending = False
        xl = self.X[L]
        compact = self._mlm_compact_now and self.with_heads
        nr = self.mlm_cap if compact else self.BT          # rows the MLM head runs on
        if compact:
            ops.mlm_compact(self.in_mlm_labels.view(-1), self.lay["text_rows"].view(-1), self.B * self.T, V, self.sel_pos, self.sel_src,
                            self.labels_c, self.counts[0:1], self.counts[2:3], self.mlm_overflow)
            ops.gather_rows(xl, self.sel_src, self.text_out[:nr])
        else:
            ops.gather_rows(xl, self.lay["text_rows"].view(-1), self.text_out)
        ops.gather_rows(xl, self.lay["obj_rows"].view(-1)[:self.BR], self.obj_out)
        if cfg.with_pooler:       # BertPooler: tanh(dense(first token)) for the image-caption samples; A operand = strided view of X[L]
            x0 = xl.view(Bt, S * H)[:self.B, :H]
            ops.gemm_nt(x0, w16["vlbert.pooler.dense.weight"], self.pooled, bias=w32["vlbert.pooler.dense.bias"], act=ops.ACT_TANH)
        if not self.with_heads:
            return
        if cfg.with_rel_loss:     # relationsip_head: Linear(H -> 2) on the pooled output (common/visual_linguistic_bert.py:505-516)
            pr = "vlbert.relationsip_head.caption_image_relationship."
            ops.gemm_nt(self.pooled, w16[pr + "weight"], self.rel_logits[:, :2], bias=w32[pr + "bias"])
        pm = "vlbert.mlm_head.predictions."
        ops.gemm_nt(self.text_out[:nr], w16[pm + "transform.dense.weight"], self.mlm_g[:nr], bias=w32[pm + "transform.dense.bias"],
                    act=ops.ACT_GELU_D, pre=self.mlm_u[:nr])
        ops.layernorm_fwd(self.mlm_g[:nr], w32[pm + "transform.LayerNorm.weight"], w32[pm + "transform.LayerNorm.bias"], self.mlm_h[:nr],
                          self.st_mlm[:nr])
        ops.gemm_nt(self.mlm_h[:nr], w16["vlbert.word_embeddings.weight"], self.mlm_logits[:nr, :V], bias=w32[pm + "bias"])
        ops.gemm_nt(self.obj_out, w16["vlbert.mvrc_head.transform.dense.weight"], self.mvrc_g,
                    bias=w32["vlbert.mvrc_head.transform.dense.bias"], act=ops.ACT_GELU_D, pre=self.mvrc_u)
        ops.gemm_nt(self.mvrc_g, w16["vlbert.mvrc_head.region_cls_pred.weight"], self.mvrc_logits[:, :C],
                    bias=w32["vlbert.mvrc_head.region_cls_pred.bias"])

    def _losses_fwd_bwd(self, gscale, keep):
        """Loss values + d(logits) written in place over the logits.  Called again by the nn.Module mirror (keep=False:
        logits restored from the kept copies, loss slots untouched) when autograd hands down an upstream scale != 1."""
        B, T, Ba, V, C = self.B, self.T, self.Ba, self.cfg.vocab_size, self.cfg.visual_region_classes
        if keep:
            losses, mcopy, vcopy = self.losses, self.mlm_logits_copy, self.mvrc_logits_copy
        else:
            self.mlm_logits.copy_(self.mlm_logits_copy)
            self.mvrc_logits.copy_(self.mvrc_logits_copy)
            losses, mcopy, vcopy = torch.zeros_like(self.losses), None, None
        nw = B * T     # rows of the image-caption samples; the aux rows follow and get their own mean (multitask.py:224-246)
        if self._mlm_compact_now:      # labelled rows only: caption rows first, then the aux rows, each group with its own mean
            ops.ce_fwd_bwd_compact(self.mlm_logits[:self.mlm_cap], V, self.labels_c, self.counts[0:1], self.counts[2:3], losses[0:1],
                                   losses[2:3], gscale=gscale)
        else:
            ops.ce_fwd_bwd(self.mlm_logits[:nw], V, self.in_mlm_labels.view(-1)[:nw], self.counts[0:1], losses[0:1], gscale=gscale,
                           logits_copy=mcopy[:nw] if mcopy is not None else None)
            if Ba:
                ops.ce_fwd_bwd(self.mlm_logits[nw:], V, self.in_mlm_labels.view(-1)[nw:], self.counts[2:3], losses[2:3],
                               gscale=gscale, logits_copy=mcopy[nw:] if mcopy is not None else None)
        ops.soft_ce_fwd_bwd(self.mvrc_logits, C, self.in_mvrc_labels.view(self.BR, C), self.mvrc_tsum, self.counts[1:2],
                            losses[1:2], gscale=gscale, logits_copy=vcopy)
        if self.cfg.with_rel_loss:     # F.cross_entropy(relationship_logits, relationship_label) (resnet_vlbert_for_pretraining.py:160-161)
            if not keep:
                self.rel_logits.copy_(self.rel_logits_copy)
            ops.ce_fwd_bwd(self.rel_logits, 2, self.in_rel_label, self.counts[3:4], losses[3:4], gscale=gscale,
                           logits_copy=self.rel_logits_copy if keep else None)

    # ------------------------------------------------------------------------------------------
    # backward (explicit; weight gradients are ACCUMULATED into the flat fp32 grad buffer)
    # ------------------------------------------------------------------------------------------
    def _wgrad(self, dy, x, gw, gb, tG, tA, rows_p):
        """gw[N,K] += dy^T x ; gb[N] += colsum(dy) through zero-padded transposes."""
        if self.use_tn_wgrad:   # straight from the row-major operands (LDS transpose reads), bias gradient fused
            if self.side is None:
                ops.wgrad_tn(dy, x, gw, colsum=gb, workspace=self.wg_ws, accumulate=not self._fresh_grads)
                return
            ready = torch.cuda.Event()
            ready.record()                                   # everything the operands depend on is enqueued on the main stream
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                ops.wgrad_tn(dy, x, gw, colsum=gb, workspace=self.wg_ws, accumulate=not self._fresh_grads)
                done = torch.cuda.Event()
                done.record()
            self._pending[dy.data_ptr()] = done              # whoever overwrites `dy` next must wait for this
            return
        N, K = dy.shape[1], x.shape[1]
        tg, ta = tG[:N, :rows_p], tA[:K, :rows_p]
        ops.transpose(dy, tg, colsum=gb)
        ops.transpose(x, ta)
        ops.wgrad_nt(tg, ta, gw, workspace=self.wg_ws)

    def _wgrad_group(self, items):
        """items: [(dy, x, gw, gb)] over the same rows -- the four Linear layers of an encoder layer -- as one grouped launch on the
        side stream (ops.wgrad_tn_group); falls back to single calls without the TN path."""
        if not self.use_tn_wgrad:
            for dy, x, gw, gb in items:
                self._wgrad(dy, x, gw, gb, self.tG, self.tA, self.Mp)
            return
        acc = not self._fresh_grads
        pad = self._row_padded       # row-padded operands (zero pad rows): K a multiple of 128 for the large-tile kernel
        items = [(pad.get(dy.data_ptr(), dy), pad.get(x.data_ptr(), x), gw, gb) for dy, x, gw, gb in items]
        if len({t.shape[0] for it in items for t in it[:2]}) != 1:
            items = [(dy[:self.M], x[:self.M], gw, gb) for dy, x, gw, gb in items]
        if self.side is None:
            ops.wgrad_tn_group(items, workspace=self.wg_ws, accumulate=acc)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            ops.wgrad_tn_group(items, workspace=self.wg_ws, accumulate=acc)
            done = torch.cuda.Event()
            done.record()
        for dy, _, _, _ in items:
            self._pending[dy.data_ptr()] = done              # whoever overwrites an operand next must wait for the group

    def _wgrad_pair(self, items):
        """items: [(dy, x, gw, gb)] of TWO encoder layers (8 products over the same rows) as ONE table launch of full-K 256 x 256 work
        items on the side stream (ops.WgradTable: no K slices, no slabs, no reduce).  The descriptor table of a pair is built once per
        (buffers, accumulate) and replayed.  False: not taken (a product outside what the table kernel covers, or a table that would
        have to be built during stream capture) -- the caller falls back to one grouped launch per layer."""
        acc = not self._fresh_grads
        pad = self._row_padded
        items = [(pad.get(dy.data_ptr(), dy), pad.get(x.data_ptr(), x), gw, gb) for dy, x, gw, gb in items]
        key = (acc,) + tuple(t.data_ptr() for it in items for t in it)
        tab = self._pair_tables.get(key)
        if tab is None:
            if self.dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return False
            if len({t.shape[0] for it in items for t in it[:2]}) != 1:
                return False
            tab = ops.WgradTable([(dy, x, gw, gb, None) for dy, x, gw, gb in items], self.dev, accumulate=acc)
            self._pair_tables[key] = tab
        if not tab.ok:
            return False
        if self.side is None:
            tab.run()
            return True
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            tab.run()
            done = torch.cuda.Event()
            done.record()
        for dy, _, _, _ in items:
            self._pending[dy.data_ptr()] = done              # whoever overwrites an operand next must wait for the launch
        return True

    def _before_write(self, *bufs):
        """Main stream is about to overwrite these buffers: wait for side-stream weight gradients still reading them."""
        for b in bufs:
            ev = self._pending.pop(b.data_ptr(), None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _join_side(self):
        """All side-stream weight gradients issued so far complete before later main-stream work."""
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(self.side)
            torch.cuda.current_stream().wait_event(ev)
            self._pending.clear()

    def _ln_bwd(self, dy, x, stats, gamma, dgamma, dbeta, **kw):
        if not self._ln_slices:             # VLB_LN_DEFER=0: finalize behind every call
            ops.layernorm_bwd(dy, x, stats, gamma, dgamma=dgamma, dbeta=dbeta, workspace=self.ln_ws, **kw)
            return
        if len(self._ln_pending) == len(self._ln_slices):
            self._flush_ln()                 # (more deferred calls than slices: the 24-layer model)
        ws = self._ln_slices[len(self._ln_pending)]
        slabs = ops.layernorm_bwd(dy, x, stats, gamma, dgamma=dgamma, dbeta=dbeta, workspace=ws, defer=True, **kw)
        if slabs:
            self._ln_pending.append((ws, slabs, dgamma, dbeta))

    def _flush_ln(self):
        if self._ln_pending:
            ops.ln_param_finalize_batch(self._ln_pending, self.cfg.hidden_size)
            self._ln_pending = []

    def backward(self, train=None, on_layer_done=None):
        will_launch = None
        if on_layer_done is None and self._dp_hook is not None:
            on_layer_done = self._dp_hook       # module mirror under parallel.DistributedDataParallel: exchange the flat gradient's buckets
        if on_layer_done is not None:       # a bucket is about to be reduced: the deferred LayerNorm parameter gradients first
            user_hook = on_layer_done
            will_launch = getattr(getattr(user_hook, "__self__", None), "will_launch", None)     # GradBuckets.on_done -> its predicate

            def on_layer_done(what):
                self._flush_ln()
                user_hook(what)
        train = self.train if train is None else train
        cfg, B, T, R, S, Bt, Ba = self.cfg, self.B, self.T, self.R, self.S, self.Bt, self.Ba
        H, I, V, C, L, nh = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.visual_region_classes, \
            cfg.num_hidden_layers, cfg.num_attention_heads
        p_h, p_a, p_ds = self._p(train)
        w16, w32, g32, wT, seed = self.w16, self.w32, self.g32, self.wT, self.seed
        Mp, BTp, BRp = self.Mp, self.BTp, self.BRp
        if self._wT_stale:           # sharded optimizer: every gathered weight has landed by now -> the dgrad operands W^T
            self.buckets.wait_params("all")
            self._refresh_transposes()
            self._wT_stale = False
        if self.buckets is not None and on_layer_done is None:
            self.buckets.invalidate()      # (a micro-step without the exchange: the reduced images are stale until a hooked backward)
        if self.with_heads:
            # --- MLM head ------------------------------------------------------------------------------------
            pm = "vlbert.mlm_head.predictions."
            compact = self._mlm_compact_now and not self.core
            nr = self.mlm_cap if compact else self.BT
            nrp = _ru(nr, 64)
            dlog = self.mlm_logits[:nr]                  # [rows, Vp], pad columns zero
            self._wgrad(dlog[:, :V], self.mlm_h[:nr], g32["vlbert.word_embeddings.weight"], g32[pm + "bias"], self.tG_bt, self.tA_bt, nrp)
            ops.gemm_nt_splitk(dlog, wT["vlbert.word_embeddings.weight"], self.d_mlm_h[:nr], workspace=self.wg_ws_main)
            self._ln_bwd(self.d_mlm_h[:nr], self.mlm_g[:nr], self.st_mlm[:nr], w32[pm + "transform.LayerNorm.weight"],
                         g32[pm + "transform.LayerNorm.weight"], g32[pm + "transform.LayerNorm.bias"], dx=self.d_mlm_g[:nr])
            ops.mul_bf16(self.d_mlm_g[:nr], self.mlm_u[:nr], self.d_mlm_u[:nr])
            self._wgrad(self.d_mlm_u[:nr], self.text_out[:nr], g32[pm + "transform.dense.weight"], g32[pm + "transform.dense.bias"], self.tG_bt,
                        self.tA_bt, nrp)
            if compact:      # gradient of the labelled rows back to their text positions; every other position gets exactly zero
                ops.gemm_nt(self.d_mlm_u[:nr], wT[pm + "transform.dense.weight"], self.d_text_out_c)
                self.d_text_out.zero_()
                ops.scatter_rows(self.d_text_out_c, self.sel_pos, self.d_text_out)
            else:
                ops.gemm_nt(self.d_mlm_u, wT[pm + "transform.dense.weight"], self.d_text_out)
            # --- MVRC head -----------------------------------------------------------------------------------
            dlog2 = self.mvrc_logits                     # [BR, Cp]
            self._wgrad(dlog2[:, :C], self.mvrc_g, g32["vlbert.mvrc_head.region_cls_pred.weight"],
                        g32["vlbert.mvrc_head.region_cls_pred.bias"], self.tG_br, self.tA_br, BRp)
            ops.gemm_nt(dlog2, wT["vlbert.mvrc_head.region_cls_pred.weight"], self.d_mvrc_u, act=ops.ACT_MULAUX, aux=self.mvrc_u)
            self._wgrad(self.d_mvrc_u, self.obj_out, g32["vlbert.mvrc_head.transform.dense.weight"],
                        g32["vlbert.mvrc_head.transform.dense.bias"], self.tG_br, self.tA_br, BRp)
            ops.gemm_nt(self.d_mvrc_u, wT["vlbert.mvrc_head.transform.dense.weight"], self.d_obj_out)
            if cfg.with_rel_loss:
                pr = "vlbert.relationsip_head.caption_image_relationship."
                self._wgrad(self.rel_logits[:, :2], self.pooled, g32[pr + "weight"], g32[pr + "bias"], None, None, 0)
                ops.gemm_nt(self.rel_logits, wT[pr + "weight"], self.d_pooled)      # K = 64: two logits + zero padding
        dx = self.dXa
        if not self.seq_out:      # (sequence mode: the caller's d(sequence_output) is already in dXa)
            ops.head_grad_combine(self.d_text_out, self.d_obj_out, self.lay["code"], dx, Bt, T, R, S, H)
        if cfg.with_pooler and (cfg.with_rel_loss or not self.with_heads):
            # d(pooled) came from the relationship head (or from the caller in hidden-state mode): through tanh and the
            # dense layer, then ADDED to the first-token rows of dX (gemm epilogue residual = its own output rows)
            ops.tanh_bwd(self.d_pooled, self.pooled, self.d_pool_pre)
            x0 = self.X[L].view(Bt, S * H)[:self.B, :H]
            self._wgrad(self.d_pool_pre, x0, g32["vlbert.pooler.dense.weight"], g32["vlbert.pooler.dense.bias"], None, None, 0)
            dx0 = dx.view(Bt, S * H)[:self.B, :H]
            ops.gemm_nt(self.d_pool_pre, wT["vlbert.pooler.dense.weight"], dx0, res=dx0)
        if on_layer_done:
            self._join_side()
            on_layer_done("heads")
        # --- encoder, last layer first -------------------------------------------------------------------
        mask = self.lay["attn_mask"]
        if self.enc32 is not None:
            dx = self.enc32.backward(dx, p_h, p_a, on_layer_done, will_launch)
        held = None      # (layer, weight-gradient items) of an odd layer waiting for the layer below it (self._pairs)
        for l in reversed(range(L if self.enc32 is None else 0)):
            p = "vlbert.encoder.layer.%d." % l
            dx_next = self.dXb if dx is self.dXa else self.dXa
            par = l & (len(self.dD2) - 1)
            dD2, dD1, dU, dQKV = self.dD2[par], self.dD1[par], self.dU2[par], self.dQKV2[par]
            # LN2: dZ2 (residual branch) and dD2 (into output.dense, through its dropout; a plain copy when dropout is off --
            # dZ is reused inside the layer, the grouped weight gradient at its end needs its own operand)
            self._before_write(self.dZ, dD2)
            self._ln_bwd(dx, self.Z2[l], self.ST2[l], w32[p + "output.LayerNorm.weight"], g32[p + "output.LayerNorm.weight"],
                         g32[p + "output.LayerNorm.bias"], dx=self.dZ, dx_drop=dD2, drop_p=p_h, seed=seed, tag=l * 8 + 2)
            self._before_write(dU)
            ops.gemm_nt(dD2, wT[p + "output.dense.weight"], dU, act=ops.ACT_MULAUX, aux=self.U[l])
            ops.gemm_nt(dU, wT[p + "intermediate.dense.weight"], dx_next, res=self.dZ)            # dY1
            # LN1
            self._before_write(self.dZ, dD1)
            self._ln_bwd(dx_next, self.Z1[l], self.ST1[l], w32[p + "attention.output.LayerNorm.weight"],
                         g32[p + "attention.output.LayerNorm.weight"], g32[p + "attention.output.LayerNorm.bias"], dx=self.dZ,
                         dx_drop=dD1, drop_p=p_h, seed=seed, tag=l * 8 + 1)
            ops.gemm_nt(dD1, wT[p + "attention.output.dense.weight"], self.dCTX)
            self._before_write(dQKV)
            ops.attention_bwd(self.QKV[l], mask, self.CTX[l], self.LSE[l], self.dCTX, dQKV, Bt, S, H, nh, drop_p=p_a,
                              seed=seed, tag=l * 8 + 0)
            gwqkv = self.P.view(self.P.grad, p + "attention.self.query.weight", (3 * H, H), span=3)
            gbqkv = self.P.view(self.P.grad, p + "attention.self.query.bias", (3 * H,), span=3)
            ops.gemm_nt(dQKV, wT[p + "qkv"], dx_next, res=self.dZ)                                  # dX_l (overwrites dY1)
            # the layer's four weight gradients on the side stream, off the critical dgrad chain: with the layer below as ONE table launch
            # (an odd layer waits for its partner), or as one grouped launch per layer
            wg = [(dD2, self.G[l], g32[p + "output.dense.weight"], g32[p + "output.dense.bias"]),
                  (dU, self.Y1[l], g32[p + "intermediate.dense.weight"], g32[p + "intermediate.dense.bias"]),
                  (dD1, self.CTX[l], g32[p + "attention.output.dense.weight"], g32[p + "attention.output.dense.bias"]),
                  (dQKV, self.X[l], gwqkv, gbqkv)]
            dx = dx_next
            if self._pairs and (l & 1):
                held = (l, wg)              # (its hook fires behind the pair's launch)
                continue
            finished = [l]
            if held is not None:
                if not self._wgrad_pair(held[1] + wg):
                    self._wgrad_group(held[1])
                    self._wgrad_group(wg)
                finished = [held[0], l]
                held = None
            else:
                self._wgrad_group(wg)
            for lf in finished:
                if on_layer_done and (will_launch is None or will_launch(lf)):
                    self._join_side()       # the bucket's weight gradients must be complete before its all-reduce reads them
                    on_layer_done(lf)
        self._join_side()               # embed_bwd adds into the word-embedding gradient the decoder wgrad wrote
        if self.core:
            self._front_core_bwd(dx, p_h)
        else:
            self._front_pretrain_bwd(dx, p_h, p_ds, on_layer_done)
        self._join_side()
        self._flush_ln()
        self._fresh_grads = False       # a further backward before the next zero_grad() accumulates
        if on_layer_done:
            on_layer_done("embed")
            on_layer_done("vision")

    def _front_core_bwd(self, dx, p_h):
        cfg, T, R, S, Bt = self.cfg, self.T, self.R, self.S, self.Bt
        H = cfg.hidden_size
        w32, g32, seed = self.w32, self.g32, self.seed
        self.d_tv_n.zero_()
        self.d_objvis.zero_()
        self.d_ol.zero_()
        pe = "vlbert.embedding_LayerNorm."
        ops.embed_bwd(dx, self.emb_pre, self.st_emb, w32[pe + "weight"], self.lay, self.in_text, self.in_text_type, None,
                      g32["vlbert.word_embeddings.weight"], g32["vlbert.position_embeddings.weight"],
                      g32["vlbert.token_type_embeddings.weight"], g32["vlbert.end_embedding.weight"], g32[pe + "weight"],
                      g32[pe + "bias"], self.d_tv_n, (T * H, H), self.d_objvis, (R * H, H), self.d_ol, (R * H, H), Bt, T, R, S, H,
                      drop_p=p_h, seed=seed, tag=TAG_EMBED)
        # gradients of the two inputs (padded positions: zero rows, as the reference's masked scatter gives)
        ops.layernorm_bwd(self.d_tv_n, self.tv_in, self.st_tv, w32["vlbert.visual_ln_text.weight"], dx=self.d_tv_in,
                          dgamma=g32["vlbert.visual_ln_text.weight"], dbeta=g32["vlbert.visual_ln_text.bias"], workspace=self.ln_ws)
        ops.layernorm_bwd(self.d_objvis, self.ovl_in[:, :H], self.st_objvis, w32["vlbert.visual_ln_object.weight"],
                          dx=self.d_ovl_in[:, :H], dgamma=g32["vlbert.visual_ln_object.weight"],
                          dbeta=g32["vlbert.visual_ln_object.bias"], workspace=self.ln_ws)
        self.d_ovl_in[:, H:].copy_(self.d_ol)      # fp32 -> bf16 strided copy (glue: hands the gradient to autograd)

    def _front_pretrain_bwd(self, dx, p_h, p_ds, on_layer_done=None):
        cfg, B, T, R, S, Bt, Ba = self.cfg, self.B, self.T, self.R, self.S, self.Bt, self.Ba
        H = cfg.hidden_size
        w16, w32, g32, wT, seed = self.w16, self.w32, self.g32, self.wT, self.seed
        BRp = self.BRp
        # --- embedding + visual LayerNorms + obj_downsample ---------------------------------------------
        self.d_objvis.zero_()
        self.d_obj_reps.zero_()
        self.d_textvis.zero_()
        pe = "vlbert.embedding_LayerNorm."
        ops.embed_bwd(dx, self.emb_pre, self.st_emb, w32[pe + "weight"], self.lay, self.in_text, None, self.in_mvrc_ops,
                      g32["vlbert.word_embeddings.weight"], g32["vlbert.position_embeddings.weight"],
                      g32["vlbert.token_type_embeddings.weight"], g32["vlbert.end_embedding.weight"], g32[pe + "weight"],
                      g32[pe + "bias"], self.d_textvis, (H, 0), self.d_objvis, (R * H, H),
                      self.P.view(self.P.grad, "object_linguistic_embeddings.weight", (2, H), span=2), (0, 0), Bt, T, R, S, H,
                      drop_p=p_h, seed=seed, tag=TAG_EMBED, text_vis_zeroed=True)
        if on_layer_done:      # the tied word-embedding gradient (decoder wgrad + this scatter-add) is complete: its 94 MB go out first
            on_layer_done("word_emb")
        ops.layernorm_bwd(self.d_objvis, self.obj_reps, self.st_objvis, w32["vlbert.visual_ln_object.weight"],
                          dx_acc=self.d_obj_reps, dgamma=g32["vlbert.visual_ln_object.weight"],
                          dbeta=g32["vlbert.visual_ln_object.bias"], workspace=self.ln_ws)
        reps0 = self.obj_reps.view(B, R * H)[:, :H]
        ops.layernorm_bwd(self.d_textvis[:B], reps0, self.st_textvis[:B], w32["vlbert.visual_ln_text.weight"],
                          dx_acc=self.d_obj_reps.view(B, R * H)[:, :H], dgamma=g32["vlbert.visual_ln_text.weight"],
                          dbeta=g32["vlbert.visual_ln_text.bias"])
        if Ba:   # all aux samples share one input row: stride-0 input, gradient accumulated into the single parameter row
            ops.layernorm_bwd(self.d_textvis[B:], w16["aux_text_visual_embedding.weight"], self.st_textvis[B:],
                              w32["vlbert.visual_ln_text.weight"], dx_acc=g32["aux_text_visual_embedding.weight"],
                              dgamma=g32["vlbert.visual_ln_text.weight"], dbeta=g32["vlbert.visual_ln_text.bias"],
                              rows=Ba, ldx=0, ldacc=0)
        ops.relu_bwd_cast(self.d_obj_reps, self.obj_reps, self.d_yds)
        pd = "image_feature_extractor.obj_downsample.1."
        self._wgrad(self.d_yds, self.a_ds, g32[pd + "weight"], g32[pd + "bias"], self.tG_br, self.tA_br, BRp)
        # gradient of the mask embedding: feature half of dA = dY W, masked regions only
        ops.gemm_nt(self.d_yds, wT[pd + "weight"][VIS_DIM:], self.d_afeat)
        if self.vision is not None:      # the features are activations of the CNN: RoI head, ROIAlign and trunk backward
            self._join_side()
            hook = None
            if on_layer_done:     # everything but the convolutions is complete: its bucket goes out under the CNN backward
                on_layer_done("embed")
                hook = lambda layer: on_layer_done("vision%d" % layer)
            self.vision.backward(self.d_afeat, self.in_boxes, drop_p=p_ds, seed=seed, tag=TAG_DOWNSAMPLE, on_stage_done=hook)
            return
        ops.masked_colsum(self.d_afeat, self.in_mvrc_ops.view(-1), g32["object_mask_visual_embedding.weight"].view(-1),
                          drop_p=p_ds, seed=seed, tag=TAG_DOWNSAMPLE, row_elems=2 * VIS_DIM, col_off=VIS_DIM)

    # ------------------------------------------------------------------------------------------
    # module-API mode (core=True)
    # ------------------------------------------------------------------------------------------
    def set_core_inputs(self, text_input_ids, text_token_type_ids, text_visual_embeddings, text_mask, object_vl_embeddings,
                        object_mask):
        """Arguments of VisualLinguisticBert.forward (common/visual_linguistic_bert.py:95-104), device tensors."""
        if not self.core:
            raise RuntimeError("engine was not built with core=True")
        Bt, T, R, H = self.Bt, self.T, self.R, self.cfg.hidden_size
        if tuple(text_input_ids.shape) != (Bt, T) or tuple(object_vl_embeddings.shape) != (Bt, R, 2 * H) or \
                tuple(text_visual_embeddings.shape) != (Bt, T, H):
            raise ValueError("core inputs must be text [%d,%d], text_visual [%d,%d,%d], object_vl [%d,%d,%d]"
                             % (Bt, T, Bt, T, H, Bt, R, 2 * H))
        self.in_text.copy_(text_input_ids)
        self.in_text_type.copy_(text_token_type_ids)
        self.tv_in.view(Bt, T, H).copy_(text_visual_embeddings)            # dtype conversion to bf16 (glue)
        self.ovl_in.view(Bt, R, 2 * H).copy_(object_vl_embeddings)
        self.text_mask.view(torch.bool).copy_(text_mask)
        self.box_mask.view(torch.bool).copy_(object_mask)

    def forward_core(self, train=None):
        """-> (mlm_logits [B,T,V], mvrc_logits [B,R,C], text_out [B,T,H], obj_out [B,R,H], pooled [B,H] | None,
        relationship_logits [B,2] | None) as bf16 views of the engine buffers."""
        self.forward(train)
        V, C, H = self.cfg.vocab_size, self.cfg.visual_region_classes, self.cfg.hidden_size
        return (self.mlm_logits[:, :V].view(self.Bt, self.T, V), self.mvrc_logits[:, :C].view(self.B, self.R, C),
                self.text_out.view(self.Bt, self.T, H), self.obj_out.view(self.B, self.R, H),
                self.pooled if self.cfg.with_pooler else None, self.rel_logits[:, :2] if self.cfg.with_rel_loss else None)

    def backward_core_hidden(self, d_text_out, d_obj_out, d_pooled=None, train=None):
        """core_heads=False: d(text_out) [B,T,H] / d(obj_out) [B,R,H] (/ d(pooled) [B,H]) -> input gradients (see backward_core)."""
        H = self.cfg.hidden_size
        if self.cfg.with_pooler:
            if d_pooled is None:
                self.d_pooled.zero_()
            else:
                self.d_pooled.copy_(d_pooled)
        if d_text_out is None:
            self.d_text_out.zero_()
        else:
            self.d_text_out.copy_(d_text_out.reshape(self.BT, H))
        if d_obj_out is None:
            self.d_obj_out.zero_()
        else:
            self.d_obj_out.copy_(d_obj_out.reshape(self.BR, H))
        self.backward(train)
        return self.d_tv_in.view(self.Bt, self.T, H), self.d_ovl_in.view(self.Bt, self.R, 2 * H)

    def sequence_output(self):
        """Last encoder layer as the packed [B, S, H] sequence (text || objects || END || padding), bf16 view."""
        return self.X[self.cfg.num_hidden_layers].view(self.Bt, self.S, self.cfg.hidden_size)

    def backward_core_sequence(self, d_seq, d_pooled=None, train=None):
        """core_sequence=True: d(sequence_output) [B, <=S, H] (rows beyond the given length are zero) -> input gradients."""
        H = self.cfg.hidden_size
        dx = self.dXa.view(self.Bt, self.S, H)
        dx.zero_()
        if d_seq is not None:
            dx[:, :d_seq.shape[1]].copy_(d_seq)
        if self.cfg.with_pooler:
            if d_pooled is None:
                self.d_pooled.zero_()
            else:
                self.d_pooled.copy_(d_pooled)
        self.backward(train)
        return self.d_tv_in.view(self.Bt, self.T, H), self.d_ovl_in.view(self.Bt, self.R, 2 * H)

    def backward_core(self, d_mlm_logits, d_mvrc_logits, d_rel_logits=None, train=None):
        """d(logits) [B,T,V] / [B,R,C] (any float dtype; None = zero) -> parameter gradients accumulated into the flat
        gradient, returns (d text_visual_embeddings [B,T,H], d object_vl_embeddings [B,R,2H]) as bf16 views."""
        V, C, H = self.cfg.vocab_size, self.cfg.visual_region_classes, self.cfg.hidden_size
        if d_mlm_logits is None:
            self.mlm_logits.zero_()
        else:
            self.mlm_logits[:, :V].copy_(d_mlm_logits.reshape(self.BT, V))
        if d_mvrc_logits is None:
            self.mvrc_logits.zero_()
        else:
            self.mvrc_logits[:, :C].copy_(d_mvrc_logits.reshape(self.BR, C))
        if self.cfg.with_rel_loss:
            if d_rel_logits is None:
                self.rel_logits.zero_()
            else:
                self.rel_logits[:, :2].copy_(d_rel_logits)
        self.backward(train)
        return self.d_tv_in.view(self.Bt, self.T, H), self.d_ovl_in.view(self.Bt, self.R, 2 * H)

    # ------------------------------------------------------------------------------------------
    # optimizer
    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Start of an optimizer step.  With the TN weight-gradient path the Linear weight gradients (97 % of the buffer)
        are OVERWRITTEN by the first backward, so only the ranges that are accumulated with atomics are cleared."""
        if not self.use_tn_wgrad or self.core or self.enc32 is not None:      # (module-API mode may run without the heads: nothing overwrites
            # their gradients; the fp32 encoder ACCUMULATES its weight gradients with atomics)
            self.P.grad.zero_()
            self._fresh_grads = False
            return
        if self._zero_small is None:
            covered = sorted((self.P.offsets[n], self.P.offsets[n] + math.prod(self.P.shapes[n])) for n in self._gemm_weight_names())
            ranges, cur = [], 0
            for lo, hi in covered:
                ranges.append((cur, lo))
                cur = hi
            ranges.append((cur, self.P.numel))
            self._zero_small = ops.ZeroRanges(self.P.grad, ranges)
        self._zero_small.run()
        self._fresh_grads = True

    def _gemm_weight_names(self):
        """Parameters whose gradient is produced (whole tensor) by exactly one _wgrad call per backward."""
        L = self.cfg.num_hidden_layers
        names = ["image_feature_extractor.obj_downsample.1.weight", "vlbert.word_embeddings.weight",
                 "vlbert.mlm_head.predictions.transform.dense.weight", "vlbert.mvrc_head.transform.dense.weight",
                 "vlbert.mvrc_head.region_cls_pred.weight"]
        if self.cfg.with_rel_loss:
            names += ["vlbert.pooler.dense.weight", "vlbert.relationsip_head.caption_image_relationship.weight"]
        for l in range(L):
            p = "vlbert.encoder.layer.%d." % l
            names += [p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight",
                      p + "attention.output.dense.weight", p + "intermediate.dense.weight", p + "output.dense.weight"]
        return names

    def optimizer_step(self, lr=None):
        """global-norm clip + AdamW + bf16 / transposed weight refresh + dropout seed advance.  With data
        parallelism the flat gradient holds the SUM over ranks; the 1/world average (DDP semantics,
        pretrain/function/train.py:89-90) is folded into the AdamW kernel's grad_scale."""
        if lr is not None:
            if self.lr_kind is not None:
                raise ValueError("optimizer_step(lr=...) conflicts with the device-side lr_schedule of this engine")
            self.adam[0:1].fill_(lr)
        if self.lr_kind is not None:
            ops.lr_schedule_step(self.adam, self.lr_kind, self.base_lr, self.warmup_steps, self.t_total)
        scale = (self.buckets.grad_scale if self.buckets is not None else 1.0) / self.loss_scale
        self._check_mlm_overflow_now_and_then()
        if self.buckets is not None and self.buckets.pending:
            self.buckets.wait()
        if self.buckets is not None and self.buckets.sharded:
            # sharded optimizer (parallel.py): this rank holds the reduced slices it owns; partial clip norm -> sum over ranks ->
            # AdamW on the owned slices -> the updated bf16 slices are all-gathered bucket by bucket under the next forward
            g = self.buckets.grad_shard
            self.shard_tbl.sumsq(g, self.sumsq_ws, self.adam[7:8])
            self.buckets.all_reduce_scalar(self.adam[7:8])
            self.shard_tbl.adamw(self.P.master, g, self.P.m, self.P.v, self.wshard, self.adam, grad_scale=scale)
            self.buckets.gather_params(self.P.w16, self.wshard, master=self.P.master, vision_master=self.vision is not None)
            self._wT_stale = self._gather_pending = True
            self._vision_stale = self.vision is not None
            ops.rng_advance(self.seed)
            return
        # data parallel: the reduced gradient is read where the exchange left it (the bf16 wire image by default, parallel.py)
        grad = self.buckets.reduced if self.buckets is not None else self.P.grad
        ops.sumsq_det(grad, self.sumsq_ws, self.adam[7:8])     # fixed summation order: replicas stay bit-identical
        ops.adamw_step(self.P.master, grad, self.P.m, self.P.v, self.P.w16, self.adam, grad_scale=scale)
        self._refresh_transposes()
        if self.vision is not None:
            self.vision.refresh_weights(trainable_only=True)
        ops.rng_advance(self.seed)

    def _check_mlm_overflow_now_and_then(self, every=64):
        """MLM head compaction with device-resident labels: the overflow flag (a batch with more labelled positions than mlm_cap ran
        truncated) is otherwise only read by loss_values(); loops that never call it get the check every `every` optimizer steps
        (one 4-byte readback)."""
        if self.mlm_cap is None:
            return
        self._opt_steps = getattr(self, "_opt_steps", 0) + 1
        if self._opt_steps % every == 0 and not torch.cuda.is_current_stream_capturing():
            self._raise_on_mlm_overflow()

    def _raise_on_mlm_overflow(self):
        if int(self.mlm_overflow.cpu()) != 0:
            self.mlm_overflow.zero_()
            raise RuntimeError("MLM head compaction: a batch carried more than mlm_cap = %d labelled text positions (the excess was "
                               "not trained on).  Build the engine with VLB_MLM_COMPACT=0 or hand the labels over as CPU tensors "
                               "(they are then counted exactly and oversize batches take the full path)." % self.mlm_cap)

    # -- dropout seed discipline of the nn.Module mirrors (reference-style loop: net.train(); loss.backward(); optimizer.step()) --
    # Masks are never stored: forward and backward regenerate them from the device-resident seed, so the seed a backward sees must
    # be the one its forward used.  The mirrors therefore advance the seed BEFORE each training forward (not after it) and the
    # autograd nodes run their backward under the seed value snapshotted right after the forward (robust to interleaved
    # forward / backward orders and to several engines sharing autograd).
    def mirror_pre_forward(self, training):
        if training:
            if getattr(self, "_seed_used", False):
                ops.rng_advance(self.seed)
            self._seed_used = True

    def seed_snapshot(self):
        return self.seed.clone()

    class _SeedGuard:
        def __init__(self, eng, snap):
            self.eng, self.snap = eng, snap

        def __enter__(self):
            self.cur = self.eng.seed.clone()
            self.eng.seed.copy_(self.snap)

        def __exit__(self, *exc):
            self.eng.seed.copy_(self.cur)
            return False

    def seed_guard(self, snap):
        return PretrainEngine._SeedGuard(self, snap)

    def broadcast_parameters(self, src=0):
        """Rank `src`'s parameters, optimizer state and (e2e) frozen vision tensors to every rank -- the start-up broadcast of
        pretrain/function/train.py:331-334, so that a checkpoint loaded on rank 0 only cannot leave the replicas diverged."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.pg) == 1:
            return
        for t in (self.P.master, self.P.m, self.P.v, self.adam):
            dist.broadcast(t, src=src, group=self.pg)
        if self.vision is not None:
            for t in self.vision.broadcast_tensors():
                dist.broadcast(t, src=src, group=self.pg)
        self._weights_dirty = True

    def train_step(self, lr=None):
        """zero_grad -> forward -> backward (gradient buckets all-reduced over RCCL as they complete,
        overlapped with the rest of backward) -> clip + AdamW: one optimizer step on the batch currently
        held in the static input buffers."""
        self.zero_grad()
        self.forward(True)
        self.backward(True, on_layer_done=self.buckets.on_done if self.buckets is not None else None)
        if self.buckets is not None:
            self.buckets.wait()
        self.optimizer_step(lr)

    def make_step_graph(self, lr=None):
        """train_step() captured as hipGraph SEGMENTS -> a callable that replays one optimizer step.  A captured graph cannot contain
        the RCCL calls of the data-parallel exchange (work handles, the communicator's own stream), so with more than one rank the
        step is cut at every collective: [graph] bucket launch [graph] ... wait [graph] norm all-reduce [graph] weight gather, and in
        the forward at every per-bucket wait of the sharded optimizer's weight gather -- ~2 x (buckets + 2) graph launches + as many
        collective calls per step instead of ~350 kernel launches from Python (host side ~5 ms -> ~1 ms; at 32 samples per GPU the
        device step is 5-6 ms, so the eager launch loop is the bound there).  One rank: a single graph.
        Call it in steady state (after at least one eager train_step on this engine); the batch is read from the static input
        buffers (set_batch between replays is fine), lr / step count / dropout seed live on the device."""
        return _SegmentedStep(self, lr)

    # ------------------------------------------------------------------------------------------
    # results (host side, sync) -- used by tests / API parity, not by the timed loop
    # ------------------------------------------------------------------------------------------
    def loss_values(self):
        if self.mlm_cap is not None:
            self._raise_on_mlm_overflow()
        l = self.losses.cpu()
        from . import _lib
        bad = _lib.nonfinite_status(reset=True)
        if bad:
            raise FloatingPointError("non-finite LayerNorm input (%s): %s" % (
                "fp16 residual stream" if bad & 1 else "bf16 rows",
                "a pre-LayerNorm sum exceeded the fp16 range (65504); run with VLB_RESIDUAL_STREAM=bf16" if bad & 1
                else "the activations already held inf / NaN"))
        out = dict(mlm_loss=float(l[0]), mvrc_loss=float(l[1]), relationship_loss=float(l[3]), loss=float(l[0] + l[1] + l[2] + l[3]))
        if self.Ba:
            out.update(mlm_loss_wvc=float(l[0]), mlm_loss_aux=float(l[2]))
        return out

    def grads(self):
        """Parameter gradients of the local batch (the loss scale divided out)."""
        inv = 1.0 / self.loss_scale
        return OrderedDict((k, v.detach().clone() * inv if inv != 1.0 else v.detach().clone()) for k, v in self.g32.items())

    def grad_norm(self):
        return float(self.P.grad.double().norm()) / self.loss_scale

    def init_random(self, seed=0, visual_ln_init=0.0):
        """Reference initialisation statistics on the device (BaseModel.init_weights,
        common/visual_linguistic_bert.py:14-25,330-332; resnet_vlbert_for_pretraining.py:55-63): N(0, 0.02)
        weights / embeddings, zero biases, unit LayerNorm gammas, visual_ln gammas = visual_scale_*_init,
        zero mask-visual embedding.  Used by bench.py / smoke (no checkpoints exist offline)."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        vis = self._vision_names()
        if self.vision is not None:
            self.vision.init_random(seed + 1)
        for name, t in self.w32.items():
            if name in vis:
                continue
            if "LayerNorm.weight" in name:
                t.fill_(1.0)
            elif name.endswith("visual_ln_text.weight") or name.endswith("visual_ln_object.weight"):
                t.fill_(visual_ln_init)
            elif name.endswith(".bias") or name == "object_mask_visual_embedding.weight":
                t.zero_()
            else:
                t.normal_(0.0, 0.02, generator=g)
        self._weights_dirty = True


class _SegmentedStep:
    """Capture of PretrainEngine.train_step as alternating [hipGraph segment] / [host call] items (PretrainEngine.make_step_graph)."""

    class _Recorder:
        """Stands in for engine.buckets during capture: the calls that start or wait for a collective END the current segment, are
        recorded (not executed: nothing runs during capture) and a new segment begins; everything else is delegated."""
        BREAKS = ("on_done", "wait", "all_reduce_scalar", "gather_params", "wait_params", "gather_master")

        def __init__(self, real, owner):
            self._real, self._owner = real, owner

        def __getattr__(self, name):
            attr = getattr(self._real, name)
            if name not in self.BREAKS:
                return attr
            owner = self._owner

            def recorded(*a, **kw):
                if name == "wait_params":       # only where a wait can block: gathers pending, and an encoder layer that opens a bucket
                    if not owner.eng._gather_pending or (isinstance(a[0], int) and self._real.layer_key[a[0]] != a[0]):
                        return None
                if name == "on_done" and not self._real.will_launch(a[0]):
                    return None
                owner.cut(lambda: attr(*a, **kw))
            recorded.__self__ = self             # (engine.backward looks up the hook owner's will_launch predicate)
            return recorded

    CHECK_EVERY = 64

    def __init__(self, eng, lr):
        if not eng.dev.type == "cuda":
            raise RuntimeError("make_step_graph needs a GPU engine")
        self.eng, self.items = eng, []
        self.replays = 0
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device=eng.dev)
        self._g = None
        torch.cuda.synchronize()
        real = eng.buckets
        if real is not None:
            if real.pending or real.launched:
                raise RuntimeError("make_step_graph: a data-parallel exchange is in flight")
            real.wait_params("all")          # weight gathers of the last eager step: let them land before anything is captured
            torch.cuda.synchronize()
            eng.buckets = self._Recorder(real, self)
        try:
            self._begin()
            eng.train_step(lr)
            self._end()
        finally:
            eng.buckets = real
        torch.cuda.synchronize()
        self.n_graphs = sum(1 for k, _ in self.items if k == "graph")
        self.n_calls = len(self.items) - self.n_graphs

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        # thread_local: the collective backend's helper threads (gloo's host copies, RCCL's proxy) keep making runtime calls while this
        # thread captures; under the default "global" mode any of them invalidates the capture
        self._ctx = torch.cuda.graph(self._g, pool=self.pool, stream=self.stream, capture_error_mode="thread_local")
        self._ctx.__enter__()

    def _end(self):
        self._ctx.__exit__(None, None, None)
        self.items.append(("graph", self._g))
        self._g = None

    def cut(self, fn):
        self._end()
        self.items.append(("call", fn))
        self._begin()

    def __call__(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()
        # optimizer_step's periodic host-side checks ran once, at capture: the replay loop carries them itself (MLM head compaction
        # overflow flag, one 4-byte readback every CHECK_EVERY replays -- never a silent truncation of the labelled positions)
        self.replays += 1
        if self.eng.mlm_cap is not None and self.replays % self.CHECK_EVERY == 0:
            self.eng._raise_on_mlm_overflow()
