"""The numeric parts of the reference's pre-training samples, as free functions (host side, no model arithmetic).

On-disk formats read here ("on-disk formats"):
  * the annotation file `<DATASET_PATH>/{train,val}_frcnn.json`: JSON LINES, one record per image with 'caption' (a list of words),
    'image' and 'frcnn' (paths below DATASET_PATH, optionally in the `archive.zip@/member` spelling) --
    pretrain/data/datasets/conceptual_captions.py:49-50,80-86;
  * the per-image detector record (one JSON file, or a member of a zip archive): 'num_boxes', 'image_w', 'image_h' and three base64 strings
    'boxes' [n, 4], 'classes' [n, C] (class scores = the soft labels of the masked-region task) and 'features' [n, D], each the raw bytes
    of a row-major float32 array -- conceptual_captions.py:103-120.
Everything that draws random numbers takes the generator as an argument and consumes it in the reference's order (one draw per whole
word / per region, conceptual_captions.py:281-347), so a run seeded like the reference sees the reference's samples
(tests/test_data_cpu.py compares with samples produced by the reference's own dataset class).
"""
import base64
import json
import random

import numpy as np
import torch

MASK_TOKEN, UNK_TOKEN = "[MASK]", "[UNK]"
P_SELECT, P_MASK_WORD, P_RANDOM_WORD, P_MASK_REGION = 0.15, 0.8, 0.9, 0.9


def _f32_rows(text, rows):
    return np.frombuffer(base64.decodebytes(text.encode()), dtype=np.float32).reshape((rows, -1))


def decode_detector_record(rec, with_features=True):
    """One detector record -> boxes [n,4], class scores [n,C], features [n,D] (or None), image width / height; rows ordered by falling
    top class score exactly as `np.argsort(max score)[::-1]` orders them (conceptual_captions.py:104-111,118)."""
    if isinstance(rec, (str, bytes)):
        rec = json.loads(rec)
    n = int(rec["num_boxes"])
    boxes, scores = _f32_rows(rec["boxes"], n), _f32_rows(rec["classes"], n)
    order = np.argsort(scores.max(axis=1))[::-1]
    feats = _f32_rows(rec["features"], n)[order] if with_features else None
    return dict(boxes=boxes[order], scores=scores[order], features=feats, width=rec["image_w"], height=rec["image_h"])


def encode_detector_record(boxes, scores, features, width, height):
    """Inverse of decode_detector_record (rows stored as given): what a feature-extraction job would write."""
    def enc(a):
        return base64.encodebytes(np.ascontiguousarray(a, dtype=np.float32).tobytes()).decode()
    rec = dict(num_boxes=int(len(boxes)), image_w=int(width), image_h=int(height), boxes=enc(boxes), classes=enc(scores))
    if features is not None:
        rec["features"] = enc(features)
    return rec


def prepend_whole_image(boxes, features, width, height):
    """DATASET.ADD_IMAGE_AS_A_BOX: slot 0 = the whole image, its feature the mean over the detected boxes (conceptual_captions.py:131-136)."""
    whole = torch.tensor([[0.0, 0.0, width - 1.0, height - 1.0]])
    boxes = torch.cat((whole, boxes), dim=0)
    if features is not None:
        features = torch.cat((features.mean(dim=0, keepdim=True), features), dim=0)
    return boxes, features


def unmix_masked_regions(features, region_ops):
    """With a whole-image slot and precomputed features the masked regions must not leak through the mean: slot 0 becomes the mean over
    the UNMASKED boxes, computed as the reference does -- scale up, subtract in slot order, divide by (kept + 1e-5) in fp32
    (conceptual_captions.py:179-187)."""
    n_real = features.shape[0] - 1
    gone = 0
    features[0] *= n_real
    for op, row in zip(region_ops, features):
        if op == 1:
            gone += 1
            features[0] -= row
    features[0] /= (n_real - gone + 1e-5)
    return features


def mask_whole_words(words, tokenizer, rng=random):
    """Whole-word masking (conceptual_captions.py:281-322): ONE draw per word; selected with probability 0.15, then 80 % -> [MASK] for every
    word piece, 10 % -> a random vocabulary entry per piece (one more draw each), 10 % kept; labels = the pieces' ids (-1 elsewhere)."""
    vocab = tokenizer.vocab
    pieces_out, labels = [], []
    for word in words:
        pieces = tokenizer.wordpiece_tokenizer.tokenize(word)
        u = rng.random()
        if u >= P_SELECT:
            pieces_out.extend(pieces)
            labels.extend([-1] * len(pieces))
            continue
        u /= P_SELECT
        if u < P_MASK_WORD:
            pieces_out.extend([MASK_TOKEN] * len(pieces))
        elif u < P_RANDOM_WORD:
            pieces_out.extend(rng.choice(list(vocab.keys())) for _ in pieces)
        else:
            pieces_out.extend(pieces)
        labels.extend(vocab[p] if p in vocab else vocab[UNK_TOKEN] for p in pieces)
    return pieces_out, labels


def mask_regions(scores, rng=random):
    """Masked-region selection (conceptual_captions.py:324-347): one draw per region; 15 % selected, of those 90 % get op 1 (feature replaced by
    the mask embedding / pixels zeroed), every selected region keeps its class scores as soft label, the others an all-zero row."""
    ops, labels = [], []
    for row in scores:
        u = rng.random()
        chosen = u < P_SELECT
        ops.append(1 if (chosen and u / P_SELECT < P_MASK_REGION) else 0)
        labels.append(row if chosen else np.zeros_like(row))
    return ops, labels


def sequence_budget(n_text, n_boxes, seq_len):
    """How many text tokens / boxes survive DATASET.SEQ_LEN (conceptual_captions.py:216-227): the longer side gives up one element at a
    time (ties: the text) until the sum fits; at least [CLS] + [SEP] and one box stay."""
    while n_text + n_boxes > seq_len and n_text > 0 and n_boxes > 0:
        if n_boxes > n_text:
            n_boxes -= 1
        else:
            n_text -= 1
    return max(n_text, 2), max(n_boxes, 1)


def zero_masked_pixels(image, boxes, region_ops):
    """NETWORK.MASK_RAW_PIXELS on the host (conceptual_captions.py:201-206): image[:, int(y1):int(y2)+1, int(x1):int(x2)+1] = 0 for op == 1.
    (The engine can do the same on the device copy: set_batch(mask_raw_pixels=True).)"""
    for op, box in zip(region_ops, boxes):
        if op == 1:
            x1, y1, x2, y2 = (int(v) for v in box[:4])
            image[:, y1:y2 + 1, x1:x2 + 1] = 0
    return image
