"""Seeded synthetic pre-training batches in the layout the reference's collator
emits (pretrain/data/collate_batch.py:12-73, conceptual_captions.py:95-97;
SURVEY.md §8a row a0 / §8d).  Data only -- no model arithmetic lives here.

Tuple order == the reference's `data_names` minus `image` (precomputed features):
    boxes [B,R,4+2048] f32 (pad rows = -2), im_info [B,5] f32 (w,h,sx,sy,idx),
    text [B,T] i64 (pad 0), relationship_label [B] i64, mlm_labels [B,T] i64 (pad -1),
    mvrc_ops [B,R] i64 (pad 0), mvrc_labels [B,R,C] f32 soft labels (pad rows 0).
"""
import torch

CLS, SEP, MASK = 101, 102, 103


def make_batch(B, T, R, vocab_size=30522, region_classes=1601, feat_dim=2048, seed=0,
               ragged=False, mlm_prob=0.15, mvrc_prob=0.135):
    g = torch.Generator().manual_seed(seed)
    lo = min(1000, vocab_size // 2)
    text = torch.randint(lo, vocab_size, (B, T), generator=g)
    text[:, 0] = CLS % vocab_size
    if ragged:
        tlen = torch.randint(max(3, T // 2), T + 1, (B,), generator=g)
        tlen[0] = T
        nobj = torch.randint(max(1, R // 2), R + 1, (B,), generator=g)
        nobj[-1] = R
    else:
        tlen = torch.full((B,), T)
        nobj = torch.full((B,), R)
    ar_t = torch.arange(T).unsqueeze(0)
    ar_r = torch.arange(R).unsqueeze(0)
    text[ar_t == (tlen.unsqueeze(1) - 1)] = SEP % vocab_size
    tpad = ar_t >= tlen.unsqueeze(1)
    text[tpad] = 0

    mlm_labels = torch.full((B, T), -1, dtype=torch.long)
    pick = (torch.rand((B, T), generator=g) < mlm_prob) & ~tpad & (ar_t > 0) & (ar_t < tlen.unsqueeze(1) - 1)
    pick[:, 1] = True  # at least one label per sample
    mlm_labels[pick] = text[pick]
    text[pick] = MASK % vocab_size

    x1 = torch.rand((B, R), generator=g) * 400
    y1 = torch.rand((B, R), generator=g) * 400
    w = 10 + torch.rand((B, R), generator=g) * 150
    h = 10 + torch.rand((B, R), generator=g) * 150
    feats = torch.rand((B, R, feat_dim), generator=g)
    boxes = torch.cat((torch.stack((x1, y1, x1 + w, y1 + h), -1), feats), -1)
    boxes[:, 0, :4] = torch.tensor([0.0, 0.0, 599.0, 599.0])   # ADD_IMAGE_AS_A_BOX: whole-image box first
    rpad = ar_r >= nobj.unsqueeze(1)
    boxes[rpad] = -2.0
    im_info = torch.tensor([600.0, 600.0, 1.0, 1.0, 0.0]).repeat(B, 1)
    im_info[:, 4] = torch.arange(B)

    mvrc_ops = (torch.rand((B, R), generator=g) < mvrc_prob).long()
    mvrc_ops[:, 0] = 0
    mvrc_ops[:, min(1, R - 1)] = 1 if R > 1 else 0
    mvrc_ops[rpad] = 0
    lab = torch.softmax(torch.randn((B, R, region_classes), generator=g), -1)
    lab = lab * (mvrc_ops == 1).unsqueeze(-1).float()
    relationship_label = torch.ones((B,), dtype=torch.long)
    return (boxes.float(), im_info, text, relationship_label, mlm_labels, mvrc_ops, lab.float())


def make_aux_text(Ba, Ta, vocab_size=30522, seed=0, ragged=True, mlm_prob=0.15):
    """Text-only auxiliary batch of the multitask pre-training (GeneralCorpus, pretrain/data/datasets/general_corpus.py):
    (aux_text [Ba,Ta] i64 pad 0, aux_mlm_labels [Ba,Ta] i64 pad -1)."""
    g = torch.Generator().manual_seed(seed + 7919)
    lo = min(1000, vocab_size // 2)
    text = torch.randint(lo, vocab_size, (Ba, Ta), generator=g)
    text[:, 0] = CLS % vocab_size
    tlen = torch.randint(max(3, Ta // 2), Ta + 1, (Ba,), generator=g) if ragged else torch.full((Ba,), Ta)
    tlen[0] = Ta
    ar = torch.arange(Ta).unsqueeze(0)
    text[ar == (tlen.unsqueeze(1) - 1)] = SEP % vocab_size
    pad = ar >= tlen.unsqueeze(1)
    text[pad] = 0
    labels = torch.full((Ba, Ta), -1, dtype=torch.long)
    pick = (torch.rand((Ba, Ta), generator=g) < mlm_prob) & ~pad & (ar > 0) & (ar < tlen.unsqueeze(1) - 1)
    pick[:, 1] = True
    labels[pick] = text[pick]
    text[pick] = MASK % vocab_size
    return text, labels


def make_vqa_batch(B, R, Lq, seed, device, answers=3129):
    """One collated VQA micro-batch with precomputed region features (vqa/data/collate_batch.py layout): boxes [B,R,4+2048] (box 0 = the
    whole image), im_info, question ids [B,Lq], soft answer scores [B,answers]."""
    g = torch.Generator().manual_seed(seed)
    Wi, Hi = 1000.0, 600.0
    x1 = torch.rand(B, R, generator=g) * (Wi - 200)
    y1 = torch.rand(B, R, generator=g) * (Hi - 200)
    w = 30 + torch.rand(B, R, generator=g) * 160
    h = 30 + torch.rand(B, R, generator=g) * 160
    boxes = torch.cat((torch.stack((x1, y1, x1 + w, y1 + h), -1), torch.randn(B, R, 2048, generator=g).abs()), -1)
    boxes[:, 0, :4] = torch.tensor([0.0, 0.0, Wi - 1.0, Hi - 1.0])
    im_info = torch.tensor([[Wi, Hi, 1.0, 1.0]] * B)
    question = torch.randint(1000, 30522, (B, Lq), generator=g)
    label = torch.zeros(B, answers)
    idx = torch.randint(0, answers, (B, 3), generator=g)
    label.scatter_(1, idx, torch.tensor([[1.0, 0.6, 0.3]] * B))
    return [t.to(device) for t in (boxes, im_info, question, label)]


def make_vcr_batch(B, C, R, Lq, La, Hi, Wi, seed, device):
    """One collated VCR micro-batch (vcr/data/collate_batch.py layout): image, boxes [B,R,5] (x1,y1,x2,y2,class; box 0 = the whole image),
    object masks [B,R,14,14], question [B,Lq,2] / answer_choices [B,C,La,2] = (token id, object tag), answer_label [B], im_info."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, Hi, Wi, generator=g) * 50.0
    x1 = torch.rand(B, R, generator=g) * (Wi - 200)
    y1 = torch.rand(B, R, generator=g) * (Hi - 200)
    w = 30 + torch.rand(B, R, generator=g) * 160
    h = 30 + torch.rand(B, R, generator=g) * 160
    boxes = torch.stack((x1, y1, x1 + w, y1 + h, torch.randint(1, 81, (B, R), generator=g).float()), -1)
    boxes[:, 0] = torch.tensor([0.0, 0.0, Wi - 1.0, Hi - 1.0, 0.0])
    masks = (torch.rand(B, R, 14, 14, generator=g) < 0.7).float()
    question = torch.zeros((B, Lq, 2), dtype=torch.int64)
    question[:, :, 0] = torch.randint(1000, 30522, (B, Lq), generator=g)
    question[:, :, 1] = torch.randint(-1, R, (B, Lq), generator=g)
    answers = torch.zeros((B, C, La, 2), dtype=torch.int64)
    answers[..., 0] = torch.randint(1000, 30522, (B, C, La), generator=g)
    answers[..., 1] = torch.randint(-1, R, (B, C, La), generator=g)
    label = torch.randint(0, C, (B,), generator=g)
    im_info = torch.tensor([[Wi, Hi, 1.0, 1.0, float(i)] for i in range(B)])
    return [t.to(device) for t in (image, boxes, masks, question, answers, label, im_info)]
