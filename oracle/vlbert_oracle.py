"""CPU oracle for the VL-BERT pre-training hot path.

TEST INFRASTRUCTURE ONLY.  A plain-PyTorch fp32 (or fp64) restatement of what the
reference computes, written functionally over a `{state_dict name: tensor}` dict
so the same parameter names/shapes as the reference's checkpoint are used.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it -- as the checker / reported CPU baseline, never as the thing shipped or
measured.  The product path (`vl-bert_amd/`) never imports `oracle`.

Pinning: `oracle/make_golden.py` runs the REAL reference modules (imported from
/root/reference in the build container) on seeded ragged batches and commits the
inputs/weights/outputs/gradients to `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this restatement against those fixtures
(the reference itself ships no golden vectors -- SURVEY.md §4).

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
import math
from dataclasses import dataclass, asdict

import torch
import torch.nn.functional as F


@dataclass
class VLBertConfig:
    """The subset of cfgs/pretrain/*.yaml NETWORK.VLBERT (+ NETWORK) the hot path reads
    (pretrain/function/config.py:86-122)."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    vocab_size: int = 30522
    max_position_embeddings: int = 512
    type_vocab_size: int = 3
    visual_region_classes: int = 1601
    visual_feat_dim: int = 2048          # hard-coded in the reference (common/fast_rcnn.py:107)
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    obj_downsample_dropout: float = 0.1  # hard-coded p=0.1 (common/fast_rcnn.py:106)
    with_pooler: bool = False
    with_rel_loss: bool = False
    multitask: bool = False              # ResNetVLBERTForPretrainingMultitask: adds aux_text_visual_embedding

    def to_dict(self):
        return asdict(self)


# --------------------------------------------------------------------------- #
# leaf ops
# --------------------------------------------------------------------------- #
def gelu(x):
    """external/pytorch_pretrained_bert/modeling.py:114-120 (erf form, not tanh)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bert_layer_norm(x, weight, bias, eps=1e-12):
    """external/pytorch_pretrained_bert/modeling.py:230-235: TF-style LN, biased
    variance, epsilon inside the sqrt."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight * x + bias


def linear(x, p, prefix):
    return F.linear(x, p[prefix + ".weight"], p[prefix + ".bias"])


def coordinate_embeddings(boxes6, dim=256):
    """common/utils/bbox.py:33-65.  boxes6 [K,6] = (x1,y1,x2,y2,W,H) -> [K,4,2*dim]."""
    w, h = boxes6[:, 4], boxes6[:, 5]
    xc = (boxes6[:, 0] + boxes6[:, 2]) / 2
    yc = (boxes6[:, 1] + boxes6[:, 3]) / 2
    bw = boxes6[:, 2] - boxes6[:, 0]
    bh = boxes6[:, 3] - boxes6[:, 1]
    pos = torch.stack((xc / w * 100, yc / h * 100, bw / w * 100, bh / h * 100), dim=1)
    dim_mat = 1000 ** (torch.arange(dim, dtype=boxes6.dtype, device=boxes6.device) / dim)
    arg = pos.view(-1, 4, 1) / dim_mat.view(1, 1, -1)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def soft_cross_entropy(logits, target):
    """common/utils/misc.py:124-151, reduction='mean': rows are valid iff
    |sum(target)-1| < 0.1; mean over valid rows of -sum(log_softmax * target)."""
    valid = (target.sum(1) - 1).abs() < 1.0e-1
    if int(valid.sum()) == 0:
        return logits.new_zeros(())
    return (-F.log_softmax(logits[valid], 1) * target[valid]).sum(1).mean(0)


# --------------------------------------------------------------------------- #
# a2: FastRCNN precomputed-feature branch
# --------------------------------------------------------------------------- #
def fast_rcnn_precomputed(p, cfg, boxes, box_mask, im_info, train, gen=None):
    """common/fast_rcnn.py:136-142,165-187.  boxes [B,R,4+2048] (features already
    mask-overwritten by the caller).  Returns obj_reps [B,R,H] zero-padded."""
    B, R = box_mask.shape
    inds = box_mask.nonzero()
    feats = boxes[inds[:, 0], inds[:, 1]][:, 4:]
    coord = coordinate_embeddings(
        torch.cat((boxes[inds[:, 0], inds[:, 1]][:, :4], im_info[inds[:, 0], :2]), 1), 256)
    x = torch.cat((coord.reshape(coord.shape[0], -1), feats), -1)
    if train and cfg.obj_downsample_dropout > 0:
        x = F.dropout(x, cfg.obj_downsample_dropout, True)
    y = F.relu(linear(x, p, "image_feature_extractor.obj_downsample.1"))
    out = y.new_zeros((B, R, y.shape[-1]))
    out[inds[:, 0], inds[:, 1]] = y   # == pad_sequence for prefix masks (common/utils/pad_sequence.py:4-17)
    return out


# --------------------------------------------------------------------------- #
# a3: embedding with seamless [text || objects || END] concatenation
# --------------------------------------------------------------------------- #
def vl_embedding(p, cfg, text_ids, text_type_ids, text_visual, text_mask, obj_vl, obj_mask,
                 train):
    """common/visual_linguistic_bert.py:173-241 (visual_ln=True, obj_pos_id_relative=True,
    position_padding_idx=-1, visual_size == hidden_size)."""
    H = cfg.hidden_size
    pre = "vlbert."
    text_ling = F.embedding(text_ids, p[pre + "word_embeddings.weight"])
    text_vis = bert_layer_norm(text_visual, p[pre + "visual_ln_text.weight"], p[pre + "visual_ln_text.bias"])
    text_vl = text_ling + text_vis
    obj_vis = bert_layer_norm(obj_vl[:, :, :H], p[pre + "visual_ln_object.weight"],
                              p[pre + "visual_ln_object.bias"])
    obj_e = obj_vl[:, :, H:] + obj_vis

    bs = text_vl.shape[0]
    max_length = int((text_mask.sum(1) + obj_mask.sum(1)).max()) + 1
    grid_pos = torch.arange(max_length, device=text_ids.device).unsqueeze(0).expand(bs, max_length)
    text_end = text_mask.sum(1, keepdim=True)
    obj_end = text_end + obj_mask.sum(1, keepdim=True)
    is_text = grid_pos < text_end
    is_obj = (grid_pos >= text_end) & (grid_pos < obj_end)
    is_end = grid_pos == obj_end

    vl = text_vl.new_zeros((bs, max_length, H))
    vl[is_text] = text_vl[text_mask]
    vl[is_obj] = obj_e[obj_mask]
    vl[is_end] = p[pre + "end_embedding.weight"][0]

    type_ids = text_type_ids.new_zeros((bs, max_length))
    type_ids[is_text] = text_type_ids[text_mask]
    type_ids[is_obj | is_end] = 2
    pos_ids = grid_pos.clone()                                   # + padding_idx(-1) + 1
    pos_ids[is_obj] = text_end.expand(bs, max_length)[is_obj]
    pos_ids[is_end] = (text_end + 1).squeeze(1)
    emb = vl + F.embedding(pos_ids, p[pre + "position_embeddings.weight"]) \
        + F.embedding(type_ids, p[pre + "token_type_embeddings.weight"])
    emb = bert_layer_norm(emb, p[pre + "embedding_LayerNorm.weight"], p[pre + "embedding_LayerNorm.bias"])
    if train and cfg.hidden_dropout_prob > 0:
        emb = F.dropout(emb, cfg.hidden_dropout_prob, True)
    mask = (grid_pos <= obj_end)
    return emb, mask, is_text, is_obj


# --------------------------------------------------------------------------- #
# a5-a10: encoder
# --------------------------------------------------------------------------- #
def bert_layer(p, cfg, pre, x, ext_mask, train):
    """external/pytorch_pretrained_bert/modeling.py:290-319 (self-attention),
    :329-333 (self-output), :361-364 (intermediate), :374-378 (output)."""
    B, S, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh

    def heads(t):
        return t.view(B, S, nh, d).permute(0, 2, 1, 3)

    q = heads(linear(x, p, pre + "attention.self.query"))
    k = heads(linear(x, p, pre + "attention.self.key"))
    v = heads(linear(x, p, pre + "attention.self.value"))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + ext_mask
    probs = torch.softmax(scores, dim=-1)
    if train and cfg.attention_probs_dropout_prob > 0:
        probs = F.dropout(probs, cfg.attention_probs_dropout_prob, True)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, S, H)

    a = linear(ctx, p, pre + "attention.output.dense")
    if train and cfg.hidden_dropout_prob > 0:
        a = F.dropout(a, cfg.hidden_dropout_prob, True)
    a = bert_layer_norm(a + x, p[pre + "attention.output.LayerNorm.weight"],
                        p[pre + "attention.output.LayerNorm.bias"])
    i = gelu(linear(a, p, pre + "intermediate.dense"))
    o = linear(i, p, pre + "output.dense")
    if train and cfg.hidden_dropout_prob > 0:
        o = F.dropout(o, cfg.hidden_dropout_prob, True)
    return bert_layer_norm(o + a, p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"])


def vlbert_forward(p, cfg, text_ids, text_type_ids, text_visual, text_mask, obj_vl, obj_mask, train):
    """common/visual_linguistic_bert.py:95-171 with output_all_encoded_layers=False,
    output_text_and_object_separately=True (as called from :356-365)."""
    emb, mask, is_text, is_obj = vl_embedding(p, cfg, text_ids, text_type_ids, text_visual, text_mask,
                                              obj_vl, obj_mask, train)
    ext = (1.0 - mask.to(emb.dtype)).unsqueeze(1).unsqueeze(2) * -10000.0
    x = emb
    for l in range(cfg.num_hidden_layers):
        x = bert_layer(p, cfg, "vlbert.encoder.layer.%d." % l, x, ext, train)
    pooled = None
    if cfg.with_pooler:
        pooled = torch.tanh(linear(x[:, 0], p, "vlbert.pooler.dense"))   # modeling.py:430-436
    T, R = text_ids.shape[1], obj_vl.shape[1]
    text_out = x[:, :T]
    obj_out = x.new_zeros((x.shape[0], R, x.shape[2]))
    obj_out[obj_mask] = x[is_obj]
    return text_out, obj_out, pooled, x


def mlm_head(p, x):
    """external/pytorch_pretrained_bert/modeling.py:439-472: dense -> gelu -> LN ->
    decoder tied to word_embeddings.weight (+ output bias)."""
    pre = "vlbert.mlm_head.predictions."
    h = gelu(linear(x, p, pre + "transform.dense"))
    h = bert_layer_norm(h, p[pre + "transform.LayerNorm.weight"], p[pre + "transform.LayerNorm.bias"])
    return F.linear(h, p["vlbert.word_embeddings.weight"]) + p[pre + "bias"]


def mvrc_head(p, x):
    """common/visual_linguistic_bert.py:473-502: dense -> gelu -> Linear(H -> classes)."""
    h = gelu(linear(x, p, "vlbert.mvrc_head.transform.dense"))
    return linear(h, p, "vlbert.mvrc_head.region_cls_pred")


# --------------------------------------------------------------------------- #
# a1: the pre-training module
# --------------------------------------------------------------------------- #
def pretrain_forward(p, cfg, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels,
                     train=False, image=None, vision_params=None, image_num_layers=101):
    """pretrain/modules/resnet_vlbert_for_pretraining.py:93-216.  image=None: precomputed-feature
    configuration.  image [B,3,H,W] + vision_params (oracle/vision_oracle.py, torchvision-style names): the e2e
    configuration -- box features come from the ResNet trunk / ROIAlign / layer4 head (common/fast_rcnn.py:144-156)
    and, as the raw pixels were masked by the dataset, no mask embedding is substituted (:114-127 passes
    mask_visual_embed=None).  Returns (outputs dict, loss) with the reference's shapes (logits re-padded to
    the original lengths with -10000)."""
    boxes = boxes.clone()                                   # the reference mutates its input (:115-117)
    box_mask = boxes[:, :, 0] > -1.5
    origin_len = boxes.shape[1]
    max_len = int(box_mask.sum(1).max())
    box_mask, boxes = box_mask[:, :max_len], boxes[:, :max_len]
    mvrc_ops, mvrc_labels = mvrc_ops[:, :max_len], mvrc_labels[:, :max_len]

    if image is not None:
        from . import vision_oracle as VO
        valid, _ = VO.e2e_features(image, boxes[:, :, :4], vision_params, image_num_layers)
        feats = valid.new_zeros((*box_mask.shape, valid.shape[1])).masked_scatter(box_mask[:, :, None], valid)
    else:
        feats = boxes[:, :, 4:].clone()
        feats[mvrc_ops == 1] = p["object_mask_visual_embedding.weight"][0]
    boxes = torch.cat((boxes[:, :, :4], feats), -1)
    obj_reps = fast_rcnn_precomputed(p, cfg, boxes, box_mask, im_info, train)

    text_mask = text > 0
    text_visual = obj_reps[:, 0:1].expand(-1, text.shape[1], -1)     # tags all 0 (:132-135, :74-91)
    B, R = box_mask.shape
    ling = p["object_linguistic_embeddings.weight"][0].expand(B, R, -1).clone()
    ling[mvrc_ops == 1] = p["object_mask_word_embedding.weight"][0]
    obj_vl = torch.cat((obj_reps, ling), -1)

    text_out, obj_out, pooled, seq = vlbert_forward(p, cfg, text, torch.zeros_like(text), text_visual, text_mask,
                                                    obj_vl, box_mask, train)
    mlm_logits = mlm_head(p, text_out)
    mvrc_logits = mvrc_head(p, obj_out)

    zero = im_info.new_zeros(())
    rel_loss, rel_logits = zero, None
    if cfg.with_rel_loss:
        rel_logits = linear(pooled, p, "vlbert.relationsip_head.caption_image_relationship")
        rel_loss = F.cross_entropy(rel_logits, relationship_label)
    # MLM: logits re-padded with -10000 to mlm_labels' length (:165-167), CE ignore_index=-1 (:176-178)
    pad = mlm_logits.new_full((*mlm_labels.shape, mlm_logits.shape[-1]), -10000.0)
    pad[:, :mlm_logits.shape[1]] = mlm_logits
    mlm_logits = pad
    mlm_loss = F.cross_entropy(mlm_logits.view(-1, mlm_logits.shape[-1]), mlm_labels.view(-1), ignore_index=-1)
    mvrc_loss = soft_cross_entropy(mvrc_logits.reshape(-1, mvrc_logits.shape[-1]),
                                   mvrc_labels.reshape(-1, mvrc_logits.shape[-1]))
    pad = mvrc_logits.new_full((B, origin_len, mvrc_logits.shape[2]), -10000.0)
    pad[:, :mvrc_logits.shape[1]] = mvrc_logits
    mvrc_logits = pad
    lab = mvrc_labels.new_zeros((B, origin_len, mvrc_labels.shape[2]))
    lab[:, :mvrc_labels.shape[1]] = mvrc_labels

    outputs = {
        "relationship_logits": rel_logits, "mlm_logits": mlm_logits, "mlm_label": mlm_labels,
        "mvrc_logits": mvrc_logits, "mvrc_label": lab,
        "relationship_loss": rel_loss, "mlm_loss": mlm_loss, "mvrc_loss": mvrc_loss,
        "sequence_output": seq,
    }
    return outputs, rel_loss + mlm_loss + mvrc_loss


def pretrain_multitask_forward(p, cfg, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels,
                               aux_text, aux_mlm_labels, train=False, image=None, vision_params=None, image_num_layers=101):
    """pretrain/modules/resnet_vlbert_for_pretraining_multitask.py:96-290 (one auxiliary text-only dataset): the aux captions
    are appended as extra samples without objects whose text-visual embedding is the learned `aux_text_visual_embedding`; MLM
    loss is split into the with-visual-content and aux parts.  image=None: precomputed features (:132-135); image + vision_params:
    the e2e configuration of cfgs/pretrain/base_e2e_16x16G_fp16.yaml (MASK_RAW_PIXELS default true => mask_visual_embed=None,
    :137-146) -- only the caption samples carry an image, the text-only samples never touch the CNN."""
    boxes = boxes.clone()
    box_mask = boxes[:, :, 0] > -1.5
    origin_len = boxes.shape[1]
    max_len = int(box_mask.sum(1).max())
    box_mask, boxes = box_mask[:, :max_len], boxes[:, :max_len]
    mvrc_ops, mvrc_labels = mvrc_ops[:, :max_len], mvrc_labels[:, :max_len]
    if image is not None:
        from . import vision_oracle as VO
        valid, _ = VO.e2e_features(image, boxes[:, :, :4], vision_params, image_num_layers)
        feats = valid.new_zeros((*box_mask.shape, valid.shape[1])).masked_scatter(box_mask[:, :, None], valid)
    else:
        feats = boxes[:, :, 4:].clone()
        feats[mvrc_ops == 1] = p["object_mask_visual_embedding.weight"][0]
    boxes = torch.cat((boxes[:, :, :4], feats), -1)
    obj_reps = fast_rcnn_precomputed(p, cfg, boxes, box_mask, im_info, train)
    B, R = box_mask.shape
    Ba = aux_text.shape[0]
    H = cfg.hidden_size
    ling = p["object_linguistic_embeddings.weight"][0].expand(B, R, -1).clone()
    ling[mvrc_ops == 1] = p["object_mask_word_embedding.weight"][0]
    obj_vl = torch.cat((obj_reps, ling), -1)

    T = max(text.shape[1], aux_text.shape[1])
    text_multi = text.new_zeros((B + Ba, T))
    text_multi[:B, :text.shape[1]] = text
    text_multi[B:, :aux_text.shape[1]] = aux_text
    tv = obj_reps.new_zeros((B + Ba, T, H))
    tv[:B, :text.shape[1]] = obj_reps[:, 0:1].expand(-1, text.shape[1], -1)
    tv[B:] = p["aux_text_visual_embedding.weight"][0]
    obj_vl_multi = obj_vl.new_zeros((B + Ba, R, 2 * H))
    obj_vl_multi[:B] = obj_vl
    box_mask_multi = box_mask.new_zeros((B + Ba, R))
    box_mask_multi[:B] = box_mask

    text_out, obj_out, pooled, seq = vlbert_forward(p, cfg, text_multi, torch.zeros_like(text_multi), tv, text_multi > 0,
                                                    obj_vl_multi, box_mask_multi, train)
    mlm_logits = mlm_head(p, text_out)
    mvrc_logits = mvrc_head(p, obj_out)[:B]
    labels_multi = mlm_labels.new_full((B + Ba, T), -1)
    labels_multi[:B, :mlm_labels.shape[1]] = mlm_labels
    labels_multi[B:, :aux_mlm_labels.shape[1]] = aux_mlm_labels
    V = mlm_logits.shape[-1]
    mlm_loss_wvc = F.cross_entropy(mlm_logits[:B].reshape(-1, V), labels_multi[:B].reshape(-1), ignore_index=-1)
    mlm_loss_aux = F.cross_entropy(mlm_logits[B:].reshape(-1, V), labels_multi[B:].reshape(-1), ignore_index=-1)
    mvrc_loss = soft_cross_entropy(mvrc_logits.reshape(-1, mvrc_logits.shape[-1]),
                                   mvrc_labels.reshape(-1, mvrc_logits.shape[-1]))
    pad = mvrc_logits.new_full((B, origin_len, mvrc_logits.shape[2]), -10000.0)
    pad[:, :mvrc_logits.shape[1]] = mvrc_logits
    outputs = {"mlm_logits_wvc": mlm_logits[:B], "mlm_logits_aux": mlm_logits[B:], "mlm_label_wvc": labels_multi[:B],
               "mlm_label_aux": labels_multi[B:], "mvrc_logits": pad, "mlm_loss_wvc": mlm_loss_wvc,
               "mlm_loss_aux": mlm_loss_aux, "mvrc_loss": mvrc_loss, "sequence_output": seq}
    return outputs, mlm_loss_wvc + mlm_loss_aux + mvrc_loss


# --------------------------------------------------------------------------- #
# parameters
# --------------------------------------------------------------------------- #
def param_shapes(cfg):
    """Names/shapes of ResNetVLBERTForPretraining.state_dict() in the precomputed-feature
    configuration (SURVEY.md §8b 'State-dict contract'), tied decoder.weight omitted."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s = {
        "image_feature_extractor.obj_downsample.1.weight": (H, 2 * cfg.visual_feat_dim),
        "image_feature_extractor.obj_downsample.1.bias": (H,),
        "object_linguistic_embeddings.weight": (1, H),
        "object_mask_visual_embedding.weight": (1, cfg.visual_feat_dim),
        "object_mask_word_embedding.weight": (1, H),
        **({"aux_text_visual_embedding.weight": (1, H)} if cfg.multitask else {}),
        "vlbert.word_embeddings.weight": (V, H),
        "vlbert.end_embedding.weight": (1, H),
        "vlbert.position_embeddings.weight": (cfg.max_position_embeddings, H),
        "vlbert.token_type_embeddings.weight": (cfg.type_vocab_size, H),
    }
    for ln in ("embedding_LayerNorm", "visual_ln_text", "visual_ln_object"):
        s["vlbert.%s.weight" % ln] = (H,)
        s["vlbert.%s.bias" % ln] = (H,)
    for l in range(cfg.num_hidden_layers):
        pre = "vlbert.encoder.layer.%d." % l
        for n in ("query", "key", "value"):
            s[pre + "attention.self.%s.weight" % n] = (H, H)
            s[pre + "attention.self.%s.bias" % n] = (H,)
        s[pre + "attention.output.dense.weight"] = (H, H)
        s[pre + "attention.output.dense.bias"] = (H,)
        s[pre + "attention.output.LayerNorm.weight"] = (H,)
        s[pre + "attention.output.LayerNorm.bias"] = (H,)
        s[pre + "intermediate.dense.weight"] = (I, H)
        s[pre + "intermediate.dense.bias"] = (I,)
        s[pre + "output.dense.weight"] = (H, I)
        s[pre + "output.dense.bias"] = (H,)
        s[pre + "output.LayerNorm.weight"] = (H,)
        s[pre + "output.LayerNorm.bias"] = (H,)
    if cfg.with_pooler:
        s["vlbert.pooler.dense.weight"] = (H, H)
        s["vlbert.pooler.dense.bias"] = (H,)
    if cfg.with_rel_loss:
        s["vlbert.relationsip_head.caption_image_relationship.weight"] = (2, H)
        s["vlbert.relationsip_head.caption_image_relationship.bias"] = (2,)
    pre = "vlbert.mlm_head.predictions."
    s[pre + "bias"] = (V,)
    s[pre + "transform.dense.weight"] = (H, H)
    s[pre + "transform.dense.bias"] = (H,)
    s[pre + "transform.LayerNorm.weight"] = (H,)
    s[pre + "transform.LayerNorm.bias"] = (H,)
    s["vlbert.mvrc_head.transform.dense.weight"] = (H, H)
    s["vlbert.mvrc_head.transform.dense.bias"] = (H,)
    s["vlbert.mvrc_head.region_cls_pred.weight"] = (cfg.visual_region_classes, H)
    s["vlbert.mvrc_head.region_cls_pred.bias"] = (cfg.visual_region_classes,)
    return s


def init_params(cfg, seed=0, dtype=torch.float32, randomize_all=True):
    """Random parameters.  `randomize_all=True` is the parity initialisation: every
    tensor (LayerNorm gammas/betas, biases, the visual_ln gammas the reference
    initialises to 0 -- common/visual_linguistic_bert.py:330-332 -- and the mask
    embeddings it zero-fills) gets non-degenerate values so no sub-path is hidden
    (SURVEY.md §8c pitfall i).  `False` reproduces the reference's init statistics
    (normal(0, 0.02) weights, zero biases, unit gammas; BaseModel.init_weights :14-25)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(cfg).items():
        is_ln_w = "LayerNorm.weight" in name or name.endswith("visual_ln_text.weight") \
            or name.endswith("visual_ln_object.weight")
        if randomize_all:
            if is_ln_w:
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
                if "visual_ln" in name:
                    t = 0.5 + 0.1 * torch.randn(shape, generator=g)
            elif name.endswith(".bias") or name.endswith("predictions.bias"):
                t = 0.02 * torch.randn(shape, generator=g)
            elif name == "object_mask_visual_embedding.weight":
                t = 0.5 * torch.rand(shape, generator=g)
            else:
                t = 0.02 * torch.randn(shape, generator=g)
        else:
            if is_ln_w:
                t = torch.zeros(shape) if "visual_ln" in name else torch.ones(shape)
            elif name.endswith(".bias") or name == "object_mask_visual_embedding.weight":
                t = torch.zeros(shape)
            else:
                t = 0.02 * torch.randn(shape, generator=g)
        p[name] = t.to(dtype)
    return p


def loss_and_grads(p, cfg, batch, train=False):
    """Forward + autograd backward.  Returns (outputs, loss, {name: grad}, global L2 grad norm)
    -- the tied MLM decoder weight is counted once (SURVEY.md §8c pitfall v)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    fwd = pretrain_multitask_forward if len(batch) == 9 else pretrain_forward
    outputs, loss = fwd(leaves, cfg, *batch, train=train)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    return outputs, loss.detach(), grads, norm


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0):
    """common/nlp/bert/optimization.py:155-185 (correct_bias=True): Adam update with the
    bias-corrected step size, then decoupled decay `p -= lr*wd*p` using the *updated* p."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_ as called at common/trainer.py:139-145."""
    c = max_norm / (total_norm + 1e-6)
    return min(c, 1.0)


def warmup_linear_lr(step, warmup_steps, t_total):
    """common/nlp/bert/optimization.py:58-62 (WarmupLinearSchedule.lr_lambda)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
