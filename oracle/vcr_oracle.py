"""CPU oracle for the VCR fine-tuning wrapper (BASELINE config 5): `ResNetVLBERT.train_forward`
(vcr/modules/resnet_vlbert_for_vcr.py:226-399) in the configuration of the shipped cfgs/vcr/*.yaml.

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header for the import rules).

  image [B,3,H,W] + boxes [B,R,4 (+ class)] + object masks [B,R,14,14] -> FastRCNN image branch with `segms` (ResNet trunk ->
  ROIAlign -> layer4 -> x mask -> avg-pool, common/fast_rcnn.py:144-156; oracle/vision_oracle.py) -> obj_downsample;
  question [B,Lq,2] / answer_choices [B,C,La,2] = (token id, object tag) pairs -> per choice `[CLS] q [SEP] a [SEP]` with token
  types 0 / 1 (`prepare_text_from_qa`, :136-167); a token's visual embedding is the object its tag points at (tag 0 = the whole
  image box, `_collect_obj_reps`, :116-134); object linguistic embedding = row clamp(class) of a 1-row (mode 2) or 81-row (mode 1)
  table (:303-308); `TimeDistributed` folds the C answer choices into the batch (common/nlp/time_distributed.py:10-50) around
  VisualLinguisticBert with the pooler; `final_mlp` ("1fc": Dropout-Linear(H,1); "2fc") on the pooled [CLS] -> logits [B,C];
  loss = sigmoid BCE with the positive weight / rescale of :344-356 or softmax CE over the choices (:358); with
  ENABLE_CNN_REG_LOSS + CNN_LOSS_TOP a second head (MVRC-style transform Linear+GELU -> Dropout -> Linear(H,81)) classifies every
  valid object's FINAL hidden state into its detector class, CE added with CNN_LOSS_WEIGHT (:389-396).
Everything below the wrapper is vlbert_oracle.py / vision_oracle.py (pinned by their own fixtures); this file is pinned by
tests/golden/vcr/vcr_small.npz, produced by oracle/make_golden.py from the reference's own VCR module.
"""
import torch
import torch.nn.functional as F

from . import vlbert_oracle as O

CLS, SEP = 101, 102


def prepare_text_from_qa(question, question_tags, question_mask, answers, answers_tags, answers_mask):
    """:136-167.  question [B,Lq], question_tags [B,C,Lq], question_mask [B,Lq]; answers / tags / mask [B,C,La].
    -> input_ids, token_type_ids, text_tags, text_mask, each [B,C,L], L = max(|q| + max_c |a_c|) + 3."""
    B, Lq = question.shape
    _, C, La = answers.shape
    L = int((question_mask.sum(1) + answers_mask.sum(2).max(1)[0]).max()) + 3
    question = question[:, None, :].expand(B, C, Lq)
    qmask = question_mask[:, None, :].expand(B, C, Lq)
    q_end = 1 + qmask.sum(2, keepdim=True)
    a_end = q_end + 1 + answers_mask.sum(2, keepdim=True)
    k = torch.arange(L)[None, None, :].expand(B, C, L)
    ids = torch.zeros((B, C, L), dtype=question.dtype)
    types = torch.zeros((B, C, L), dtype=question.dtype)
    tags = torch.zeros((B, C, L), dtype=question.dtype)
    mask = ~(k > a_end)
    types[(k > q_end) & (k <= a_end)] = 1
    q_in = (k > 0) & (k < q_end)
    a_in = (k > q_end) & (k < a_end)
    ids[:, :, 0] = CLS
    ids[k == q_end] = SEP
    ids[k == a_end] = SEP
    ids[q_in] = question[qmask]
    ids[a_in] = answers[answers_mask]
    tags[q_in] = question_tags[qmask]
    tags[a_in] = answers_tags[answers_mask]
    return ids, types, tags, mask


def final_mlp(p, pooled, classifier, train, drop_p):
    def drop(x):
        return F.dropout(x, drop_p, True) if (train and drop_p > 0) else x
    if classifier == "1fc":
        return O.linear(drop(pooled), p, "final_mlp.1")
    h = F.relu(O.linear(drop(pooled), p, "final_mlp.1"))
    return O.linear(drop(h), p, "final_mlp.4")


def vcr_forward(p, cfg, image, boxes, masks, question, answer_choices, answer_label, im_info, vision_params, image_num_layers=101,
                classifier="1fc", classifier_dropout=0.1, sigmoid=True, positive_weight=1.0, cnn_reg_top=True, cnn_reg_dropout=0.0,
                cnn_loss_weight=1.0, ans_loss_weight=1.0, embed_mode=2, train=False):
    """-> (outputs dict, loss) like train_forward.  `p`: parameters under the reference's names, the VL-BERT ones WITHOUT the
    `vlbert._module.` prefix replaced (i.e. `vlbert.<name>`, as vlbert_oracle expects); boxes [B,R,5] = (x1,y1,x2,y2,class), padded
    rows all -1 (vcr/data/collate_batch.py); answer_label [B] or None (inference: logits only)."""
    from . import vision_oracle as VO
    objects = boxes[:, :, -1]
    boxes4 = boxes[:, :, :4]
    box_mask = boxes4[:, :, -1] > -0.5
    max_len = int(box_mask.sum(1).max())
    objects, box_mask, boxes4, segms = objects[:, :max_len], box_mask[:, :max_len], boxes4[:, :max_len], masks[:, :max_len]
    B, R = box_mask.shape
    # common/fast_rcnn.py:136-187 (image branch): the padded boxes carry the marker the feature oracle tests
    marked = boxes4.clone()
    marked[~box_mask] = -2.0
    valid, _ = VO.e2e_features(image, marked, vision_params, image_num_layers, segms=segms)
    feats = valid.new_zeros((B, R, valid.shape[1])).masked_scatter(box_mask[:, :, None], valid)
    obj_reps = O.fast_rcnn_precomputed(p, cfg, torch.cat((marked, feats), -1), box_mask, im_info, train)

    C = answer_choices.shape[1]
    q_ids, q_tags = question[:, :, 0], question[:, :, 1]
    q_tags = q_tags[:, None, :].expand(-1, C, -1)
    q_mask = question[:, :, 0] > 0.5
    a_ids, a_tags = answer_choices[:, :, :, 0], answer_choices[:, :, :, 1]
    a_mask = answer_choices[:, :, :, 0] > 0.5
    ids, types, tags, text_mask = prepare_text_from_qa(q_ids, q_tags, q_mask, a_ids, a_tags, a_mask)
    L = ids.shape[2]
    rows = torch.arange(B)[:, None, None].expand(B, C, L)
    text_visual = obj_reps[rows.reshape(-1), tags.clamp(min=0).reshape(-1)].view(B, C, L, -1)        # _collect_obj_reps
    table = p["object_linguistic_embeddings.weight"]
    ling = table[objects.long().clamp(min=0, max=table.shape[0] - 1)]                                   # [B,R,H]
    obj_vl = torch.cat((obj_reps, ling), -1)[:, None].expand(B, C, R, -1)
    fold = lambda t: t.reshape(B * C, *t.shape[2:])                                                     # TimeDistributed
    text_out, obj_out, pooled, _ = O.vlbert_forward(p, cfg, fold(ids), fold(types), fold(text_visual), fold(text_mask), fold(obj_vl),
                                                    fold(box_mask[:, None].expand(B, C, R)), train)
    logits = final_mlp(p, pooled.view(B, C, -1), classifier, train, classifier_dropout).squeeze(2)
    out = {"label_logits": logits}
    if answer_label is None:
        return out, None
    if sigmoid:
        label_binary = torch.arange(C)[None, :] == answer_label[:, None]
        weight = torch.ones_like(logits)
        weight[label_binary] = positive_weight
        rescale = (positive_weight + 1.0) / (2.0 * positive_weight)
        ans_loss = rescale * F.binary_cross_entropy_with_logits(logits, label_binary.to(logits.dtype), weight=weight)
        out["positive_fraction"] = label_binary.to(logits.dtype).sum() / label_binary.numel()
    else:
        ans_loss = F.cross_entropy(logits, answer_label.long().view(-1))
    out.update(label=answer_label.long().view(-1), ans_loss=ans_loss)
    loss = ans_loss * ans_loss_weight
    if cnn_reg_top:
        sel = box_mask[:, None].expand(B, C, R).reshape(B * C, R)
        h = obj_out[sel]                                                          # final hidden states of the valid objects
        h = O.gelu(O.linear(h, p, "cnn_loss_reg.0.dense"))
        if train and cnn_reg_dropout > 0:
            h = F.dropout(h, cnn_reg_dropout, True)
        reg_logits = O.linear(h, p, "cnn_loss_reg.2")
        reg_loss = F.cross_entropy(reg_logits, objects[:, None].expand(B, C, R).reshape(B * C, R)[sel].long())
        out["cnn_regularization_loss"] = reg_loss
        loss = loss + reg_loss * cnn_loss_weight
    return out, loss


def init_vcr_params(cfg, seed, classifier="1fc", hidden=1024, embed_mode=2, cnn_reg_top=True):
    """vlbert_oracle.init_params (with the pooler) without the pre-training heads + the VCR wrapper's own tensors."""
    base = O.init_params(cfg, seed=seed)
    p = {k: v for k, v in base.items() if "mlm_head" not in k and "mvrc_head" not in k and "object_mask_" not in k
         and "relationsip_head" not in k and "aux_text_visual" not in k and not k.startswith("object_linguistic_embeddings")}
    g = torch.Generator().manual_seed(seed + 211)
    H = cfg.hidden_size

    def lin(name, o, i):
        p[name + ".weight"] = torch.randn(o, i, generator=g) * (2.0 / (o + i)) ** 0.5
        p[name + ".bias"] = 0.02 * torch.randn(o, generator=g)
    p["object_linguistic_embeddings.weight"] = 0.02 * torch.randn(81 if embed_mode == 1 else 1, H, generator=g)
    if classifier == "1fc":
        lin("final_mlp.1", 1, H)
    else:
        lin("final_mlp.1", hidden, H)
        lin("final_mlp.4", 1, hidden)
    if cnn_reg_top:
        lin("cnn_loss_reg.0.dense", H, H)
        lin("cnn_loss_reg.2", 81, H)
    return p
