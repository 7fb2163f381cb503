"""CPU oracle for the VQA fine-tuning wrapper: `ResNetVLBERT.train_forward`
(vqa/modules/resnet_vlbert_for_vqa.py:169-236) on precomputed region features.

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header for the import rules).

  question ids -> [CLS] q [SEP] [MASK] [SEP] with token types 0 / 1 (`prepare_text_from_qa`, :142-167; the answer is the single
  [MASK] token, :192-196), text tags all 0 -> every token sees obj_reps[:, 0] (:203-209), object linguistic embedding = row 0 of
  a 1-row table (:211-215), VisualLinguisticBert with the packed sequence output (:221-227), hm = hidden state at the [MASK]
  position (:230), final_mlp (:55-77: "2fc" Dropout-Linear-ReLU-Dropout-Linear, "1fc", or "mlm" = BertPredictionHeadTransform-
  Dropout-Linear), loss = binary_cross_entropy_with_logits(logits, label) * label.size(1) (:244).
Everything below the head is vlbert_oracle.py (pinned by the pre-training fixtures); this file is pinned by
tests/golden/vqa/vqa_small.npz, produced by oracle/make_golden.py from the reference's own VQA module ("2fc" classifier).
"""
import torch
import torch.nn.functional as F

from . import vlbert_oracle as O

CLS, SEP, MASK = 101, 102, 103


def prepare_text_from_qa(question):
    """question [B, Lq] (0 = padding) -> input_ids, token_type_ids, text_mask [B, L], ans_pos [B]; L = max question length + 4."""
    B = question.shape[0]
    qmask = question > 0.5
    qlen = qmask.sum(1)
    L = int(qlen.max()) + 1 + 3                      # (question + 1 answer token) + [CLS] + 2 x [SEP]
    q_end = 1 + qlen                                  # position of the first [SEP]
    a_end = q_end + 2                                 # position of the second [SEP] (one answer token in between)
    j = torch.arange(L)[None, :]
    ids = torch.zeros((B, L), dtype=question.dtype)
    types = torch.zeros((B, L), dtype=question.dtype)
    mask = j <= a_end[:, None]
    types[(j > q_end[:, None]) & (j <= a_end[:, None])] = 1
    ids[:, 0] = CLS
    ids[j == q_end[:, None]] = SEP
    ids[j == a_end[:, None]] = SEP
    ids[(j > 0) & (j < q_end[:, None])] = question[qmask]
    ids[j == (q_end + 1)[:, None]] = MASK
    return ids, types, mask, a_end - 1


def final_mlp(p, hm, classifier, train, drop_p):
    def drop(x):
        return F.dropout(x, drop_p, True) if (train and drop_p > 0) else x
    if classifier == "2fc":
        h = F.relu(O.linear(drop(hm), p, "final_mlp.1"))
        return O.linear(drop(h), p, "final_mlp.4")
    if classifier == "1fc":
        return O.linear(drop(hm), p, "final_mlp.1")
    if classifier == "mlm":                          # BertPredictionHeadTransform (modeling.py:439-453) -> Dropout -> Linear
        h = O.gelu(O.linear(hm, p, "final_mlp.0.dense"))
        h = O.bert_layer_norm(h, p["final_mlp.0.LayerNorm.weight"], p["final_mlp.0.LayerNorm.bias"])
        return O.linear(drop(h), p, "final_mlp.2")
    raise ValueError(classifier)


def vqa_forward(p, cfg, boxes, im_info, question, label, classifier="2fc", classifier_dropout=0.1, train=False):
    """-> (outputs dict, loss) like train_forward; `label` may be None (inference_forward, :238-300)."""
    box_mask = boxes[:, :, 0] > -1.5
    max_len = int(box_mask.sum(1).max())
    box_mask, boxes = box_mask[:, :max_len], boxes[:, :max_len]
    obj_reps = O.fast_rcnn_precomputed(p, cfg, boxes, box_mask, im_info, train)
    ids, types, text_mask, ans_pos = prepare_text_from_qa(question)
    text_visual = obj_reps[:, 0:1].expand(-1, ids.shape[1], -1)
    B, R = box_mask.shape
    ling = p["object_linguistic_embeddings.weight"][0].expand(B, R, -1)
    obj_vl = torch.cat((obj_reps, ling), -1)
    _, _, _, seq = O.vlbert_forward(p, cfg, ids, types, text_visual, text_mask, obj_vl, box_mask, train)
    hm = seq[torch.arange(B), ans_pos]
    logits = final_mlp(p, hm, classifier, train, classifier_dropout)
    out = {"label_logits": logits}
    if label is None:
        return out, None
    loss = F.binary_cross_entropy_with_logits(logits, label) * label.shape[1]
    out.update(label=label, ans_loss=loss)
    return out, loss


def init_vqa_params(cfg, seed, answer_vocab, classifier="2fc", hidden=1024):
    """vlbert_oracle.init_params without the pre-training heads / mask embeddings + the classifier of `final_mlp`."""
    base = O.init_params(cfg, seed=seed)
    p = {k: v for k, v in base.items() if "mlm_head" not in k and "mvrc_head" not in k and "object_mask_" not in k
         and "relationsip_head" not in k and "aux_text_visual" not in k}
    g = torch.Generator().manual_seed(seed + 101)
    H = cfg.hidden_size

    def lin(name, o, i):
        p[name + ".weight"] = torch.randn(o, i, generator=g) * (2.0 / (o + i)) ** 0.5
        p[name + ".bias"] = 0.02 * torch.randn(o, generator=g)
    if classifier == "2fc":
        lin("final_mlp.1", hidden, H)
        lin("final_mlp.4", answer_vocab, hidden)
    elif classifier == "1fc":
        lin("final_mlp.1", answer_vocab, H)
    else:
        lin("final_mlp.0.dense", H, H)
        p["final_mlp.0.LayerNorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        p["final_mlp.0.LayerNorm.bias"] = 0.02 * torch.randn(H, generator=g)
        lin("final_mlp.2", answer_vocab, H)
    return p
