"""Generate tests/fixtures/cc_tiny/ (a miniature of the reference's on-disk pre-training data) and tests/golden/data/cc_tiny.npz (the
samples and collated batches the REAL reference's dataset classes produce from it).  Build container only:

    python oracle/make_data_golden.py

TEST INFRASTRUCTURE ONLY.  The reference classes run unmodified from /root/reference (pretrain/data/datasets/conceptual_captions.py,
general_corpus.py, collate_batch.py, transforms/transforms.py, external/pytorch_pretrained_bert BertTokenizer).  Two third-party modules
they import are absent from this image and are stood in for, for the import only / with the library's documented behaviour:
  * jsonlines (jsonlines.open(path) -> an iterable of json.loads(line)); pycocotools (an empty stand-in: the package __init__ imports the
    COCO dataset module, which the fixture does not use);
  * torchvision.transforms.functional: resize (PIL bilinear), hflip, to_tensor (HWC uint8 -> CHW float / 255), normalize ((x - mean) / std).
    The image arithmetic of the fixture's image mode therefore pins the reference's OWN steps (size rule, box scaling, flip of the boxes,
    BGR x 255, pixel masking, zero padding) and is "parity unpinned" for those four torchvision functions.
Random draws: Python's `random`, seeded per pass; the fixture records the seeds.
"""
import base64
import io
import json
import os
import random
import sys
import types
import zipfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

FIX = os.path.join(ROOT, "tests", "fixtures", "cc_tiny")
OUT = os.path.join(ROOT, "tests", "golden", "data")
WORDS = ("a the of on in with and two small large red blue green dog cat bird man woman child street park table playing running "
         "sitting standing ball tree river bridge umbrella bicycle skateboard looking holding near beside under over").split()
PIECES = ["##s", "##ing", "##ed", "skate", "##board", "bi", "##cycle", "umb", "##rella", "un", "##der"]
CAPTIONS = [
    "a small dog playing with a red ball in the park", "two birds sitting on a bridge over the river",
    "a man holding an umbrella beside a woman on the street", "child running near a large green tree with a cat",
    "a woman standing under a blue umbrella", "the man and the child looking at skateboards and bicycles near a table",
]
CORPUS = ["The river runs under the bridge.", "Two dogs.", "", "A child is playing with a ball near the table in the park",
          "bicycles and skateboards", "The woman was looking over the street and holding a small umbrella under a large tree"]
SIZES = [(50, 40), (33, 60), (64, 64), (80, 30), (45, 45), (30, 70)]
C, D = 6, 8


def write_fixture():
    rng = np.random.RandomState(7)
    os.makedirs(os.path.join(FIX, "vocab"), exist_ok=True)
    vocab = ["[PAD]"] + ["[unused%d]" % i for i in range(3)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(dict.fromkeys(WORDS)) + PIECES + [".", "an", "at", "is", "was", "runs"]
    with open(os.path.join(FIX, "vocab", "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    with open(os.path.join(FIX, "corpus.doc"), "w") as f:
        f.write("\n".join(CORPUS) + "\n")
    from PIL import Image
    ann = []
    os.makedirs(os.path.join(FIX, "train_image"), exist_ok=True)
    os.makedirs(os.path.join(FIX, "train_frcnn"), exist_ok=True)

    def enc(a):
        return base64.encodebytes(np.ascontiguousarray(a, dtype=np.float32).tobytes()).decode()
    for i, (cap, (w, h)) in enumerate(zip(CAPTIONS, SIZES)):
        n = 3 + (i * 2) % 5
        x1, y1 = rng.uniform(0, w * 0.6, n), rng.uniform(0, h * 0.6, n)
        boxes = np.stack((x1, y1, np.minimum(x1 + rng.uniform(3, w * 0.5, n), w + 2.0), np.minimum(y1 + rng.uniform(3, h * 0.5, n), h + 2.0)), 1)
        scores = rng.dirichlet(np.ones(C) * 0.7, n)
        feats = rng.rand(n, D)
        rec = dict(num_boxes=n, image_w=w, image_h=h, boxes=enc(boxes), classes=enc(scores), features=enc(feats))
        with open(os.path.join(FIX, "train_frcnn", "%04d.json" % i), "w") as f:
            json.dump(rec, f)
        if i != 4:      # record 4 has no image file: the zero-image path
            Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8), "RGB").save(os.path.join(FIX, "train_image", "%04d.png" % i))
        shard = ".%d" % (i % 4)
        ann.append(dict(caption=cap.split(), image="train_image.zip@/%04d.png" % i, frcnn="train_frcnn%s.zip@/%04d.json" % (shard, i)))
    with open(os.path.join(FIX, "train_frcnn.json"), "w") as f:
        for a in ann:
            f.write(json.dumps(a) + "\n")
    # the same records as one zip archive (zip_mode): only record 0, to keep the fixture small
    with zipfile.ZipFile(os.path.join(FIX, "train_frcnn.0.zip"), "w") as z:
        z.write(os.path.join(FIX, "train_frcnn", "0000.json"), "0000.json")


def install_data_stubs():
    m = types.ModuleType("jsonlines")

    def _open(path):
        with open(path) as f:
            return [json.loads(line) for line in f if line.strip()]
    m.open = _open
    sys.modules["jsonlines"] = m
    tv, tvt, F = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    from PIL import Image
    F.resize = lambda img, size: img.resize((size[1], size[0]), Image.BILINEAR)
    F.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
    F.to_tensor = lambda img: torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    F.normalize = lambda t, mean, std: (t - torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)) / torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
    tvt.functional = F
    tv.transforms = tvt
    coco, coco_m = types.ModuleType("pycocotools"), types.ModuleType("pycocotools.coco")      # (imported by the COCO dataset module only)
    coco_m.COCO = type("COCO", (), {})
    coco.coco = coco_m
    sys.modules.update({"pycocotools": coco, "pycocotools.coco": coco_m})
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": F})


def flat(prefix, sample, names, out):
    for name, v in zip(names, sample):
        if v is None:
            continue
        out["%s/%s" % (prefix, name)] = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)


def main():
    write_fixture()
    ref_import.import_reference()
    install_data_stubs()
    from external.pytorch_pretrained_bert import BertTokenizer
    from pretrain.data.datasets.conceptual_captions import ConceptualCaptionsDataset
    from pretrain.data.datasets.general_corpus import GeneralCorpus
    from pretrain.data.collate_batch import BatchCollator
    from pretrain.data.transforms import transforms as T
    tok = BertTokenizer(os.path.join(FIX, "vocab", "vocab.txt"), do_lower_case=True)
    MEANS, STDS = (102.9801, 115.9465, 122.7717), (1.0, 1.0, 1.0)
    out, meta = {}, dict(passes={}, pixel_means=MEANS, pixel_stds=STDS)

    def run(tag, ds, seeds, batch=None):
        names = ds.data_names
        coll = BatchCollator(dataset=ds, append_ind=False)
        for seed in seeds:
            random.seed(seed)
            samples = [ds[i] for i in range(len(ds))]
            for i, s in enumerate(samples):
                flat("%s/s%d/%d" % (tag, seed, i), s, names, out)
            for b0 in range(0, len(samples), batch or len(samples)):
                flat("%s/s%d/batch%d" % (tag, seed, b0), coll(samples[b0:b0 + (batch or len(samples))]), names, out)
        meta["passes"][tag] = dict(seeds=list(seeds), n=len(ds), batch=batch or len(ds))

    # A: precomputed features, whole-image box, three tasks, Resize(60, 100) + flip 0.5, SEQ_LEN 20 (some samples truncated)
    tfA = T.Compose([T.Resize(60, 100), T.RandomHorizontalFlip(0.5), T.ToTensor(), T.Normalize(MEANS, STDS, to_bgr255=True)])
    dsA = ConceptualCaptionsDataset("", "train", FIX, FIX, seq_len=20, with_precomputed_visual_feat=True, tokenizer=tok,
                                    add_image_as_a_box=True, transform=tfA)
    run("prec", dsA, seeds=(1, 2, 3, 4, 5, 6), batch=4)
    # B: images, raw-pixel masking, same transform chain, no truncation
    dsB = ConceptualCaptionsDataset("", "train", FIX, FIX, seq_len=64, with_precomputed_visual_feat=False, mask_raw_pixels=True,
                                    tokenizer=tok, add_image_as_a_box=True, transform=tfA)
    run("image", dsB, seeds=(11, 12, 13), batch=3)
    # C: no whole-image box, no relationship / MLM task, no transform
    dsC = ConceptualCaptionsDataset("", "train", FIX, FIX, seq_len=64, with_precomputed_visual_feat=True, with_rel_task=False,
                                    with_mlm_task=False, tokenizer=tok, add_image_as_a_box=False, transform=None)
    run("plain", dsC, seeds=(21,))
    # D: text-only corpus
    dsD = GeneralCorpus(os.path.join(FIX, "corpus.doc"), None, tokenizer=tok, seq_len=16, min_seq_len=12)
    run("corpus", dsD, seeds=(31, 32, 33))

    # every branch of the random masks must be hit somewhere, or the fixture proves less than it claims
    ids = np.concatenate([v.ravel() for k, v in out.items() if k.startswith("prec/") and k.endswith("/text") and "batch" not in k])
    labs = np.concatenate([v.ravel() for k, v in out.items() if k.startswith("prec/") and k.endswith("/mlm_labels") and "batch" not in k])
    mask_id = tok.vocab["[MASK]"]
    assert ((labs >= 0) & (ids == mask_id)).any() and ((labs >= 0) & (ids == labs)).any() and ((labs >= 0) & (ids != labs) & (ids != mask_id)).any()
    ops = np.concatenate([v.ravel() for k, v in out.items() if k.startswith("prec/") and k.endswith("/mvrc_ops") and "batch" not in k])
    soft = [v for k, v in out.items() if k.startswith("prec/") and k.endswith("/mvrc_labels") and "batch" not in k]
    kept = sum(int(((s.sum(1) > 0) & (o == 0)).sum()) for s, o in zip(soft, [v for k, v in out.items() if k.startswith("prec/") and k.endswith("/mvrc_ops") and "batch" not in k]))
    assert ops.sum() > 0 and kept > 0, (ops.sum(), kept)
    rel = np.array([int(v) for k, v in out.items() if k.startswith("prec/") and k.endswith("/relationship_label") and "batch" not in k])
    assert 0 < rel.sum() < len(rel)
    trunc = [v.shape[0] for k, v in out.items() if k.startswith("prec/") and k.endswith("/boxes") and "batch" not in k]
    assert min(trunc) < max(trunc)
    cases = ["Skateboards, bicycles & umbrellas -- under the Bridge!", "A caf\u00e9 na\u00efve r\u00e9sum\u00e9 \u00c5ngstr\u00f6m", "dogs\tand\ncats\r\n  birds\u00a0playing",
             "\u4e2d\u6587 dog\u72d7cat", "[MASK] the [CLS]dog[SEP] [UNK]. [PAD]", "it's a dog's-ball; (red) {blue} <green> `x` ^ ~ $5 #1", "unfolded " + "x" * 101 + " running",
             "null\x00char \ufffd ctrl\x07bell \u200bzero\u200dwidth", "\u00bfQu\u00e9? \u2014 \u201cquoted\u201d \u2026 \u3001", "RUNNING Runs runs. skateboarding bicycled"]
    meta["tokenizer_cases"] = [[c, tok.basic_tokenizer.tokenize(c), tok.tokenize(c)] for c in cases]
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "cc_tiny.npz"), **out)
    with open(os.path.join(OUT, "cc_tiny_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
