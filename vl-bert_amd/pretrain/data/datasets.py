"""The pre-training datasets of the reference over its on-disk formats, for feeding engine.set_batch() / the module mirrors with real data.

  ConceptualCaptionsDataset  pretrain/data/datasets/conceptual_captions.py:19-230 (image-caption pairs + detector records; data_names
                             image, boxes, im_info, text, relationship_label, mlm_labels, mvrc_ops, mvrc_labels)
  GeneralCorpus              pretrain/data/datasets/general_corpus.py:7-73 (text-only lines for the multitask wrapper; text, mlm_labels)
Same constructor arguments and sample tuples as the reference classes.  The tokenizer is any object with the BertTokenizer surface
(`vocab`, `basic_tokenizer`, `wordpiece_tokenizer`, `tokenize`, `convert_tokens_to_ids`); by default this package's WordPiece tokenizer (wordpiece.py) over
`<pretrained_model_name>/vocab.txt` (there is no network on the MI355X boxes: a model NAME that is not a local directory is an error).
Random draws come from Python's global `random` in the reference's order (flip, relationship, words, regions), so `random.seed(s)`
reproduces the reference's samples.  Host glue: nothing here touches the GPU.
"""
import io
import json
import os
import random
import zipfile

import numpy as np
import torch
from torch.utils.data import Dataset

from . import records as R

ZIP_AT = ".zip@"


def default_tokenizer(name_or_dir):
    vocab = os.path.join(name_or_dir, "vocab.txt") if name_or_dir and os.path.isdir(name_or_dir) else name_or_dir
    if not vocab or not os.path.isfile(vocab):
        raise FileNotFoundError("tokenizer vocabulary %r not found: pass tokenizer= or point NETWORK.BERT_MODEL_NAME at a directory "
                                "with vocab.txt (no download path on this machine)" % (name_or_dir,))
    from .wordpiece import BertTokenizer
    return BertTokenizer(vocab, do_lower_case=True)


class _Archives:
    """`dir/archive.zip@/member` paths (common/utils/zipreader.py): one open ZipFile per archive and process."""

    def __init__(self):
        self.open_files = {}

    def read(self, path):
        cut = path.index(ZIP_AT) + len(ZIP_AT) - 1
        archive, member = path[:cut], path[cut + 1:].strip("/")
        if archive not in self.open_files:
            self.open_files[archive] = zipfile.ZipFile(archive, "r")
        return self.open_files[archive].read(member)


class ConceptualCaptionsDataset(Dataset):
    ANNOTATIONS = {"train": "train_frcnn.json", "val": "val_frcnn.json"}

    def __init__(self, ann_file, image_set, root_path, data_path, seq_len=64, with_precomputed_visual_feat=False, mask_raw_pixels=True,
                 with_rel_task=True, with_mlm_task=True, with_mvrc_task=True, transform=None, test_mode=False, zip_mode=False,
                 cache_mode=False, cache_db=False, ignore_db_cache=True, tokenizer=None, pretrained_model_name=None,
                 add_image_as_a_box=False, aspect_grouping=False, **kwargs):
        if cache_mode or test_mode or aspect_grouping:
            raise NotImplementedError("cache_mode / test_mode / aspect_grouping: refused by the reference too (conceptual_captions.py:45-46,89)")
        self.seq_len, self.transform = seq_len, transform
        self.with_rel_task, self.with_mlm_task, self.with_mvrc_task = with_rel_task, with_mlm_task, with_mvrc_task
        self.with_precomputed_visual_feat, self.mask_raw_pixels = with_precomputed_visual_feat, mask_raw_pixels
        self.add_image_as_a_box, self.zip_mode, self.test_mode = add_image_as_a_box, zip_mode, test_mode
        self.data_path, self.root_path, self.image_set = data_path, root_path, image_set
        self.ann_file = os.path.join(data_path, self.ANNOTATIONS[image_set])       # (the `ann_file` argument is not read: conceptual_captions.py:57)
        self.tokenizer = tokenizer if tokenizer is not None else default_tokenizer(pretrained_model_name)
        self.archives = _Archives()
        with open(self.ann_file, "r") as f:
            self.database = [json.loads(line) for line in f if line.strip()]
        if not zip_mode:      # the annotation names archive members; unpacked trees drop the archive suffix (and the shard digit of the feature archives)
            for rec in self.database:
                flat = rec["frcnn"].replace(ZIP_AT, "")
                for shard in (".0", ".1", ".2", ".3"):
                    flat = flat.replace(shard, "")
                rec["frcnn"] = flat
                rec["image"] = rec["image"].replace(ZIP_AT, "")

    data_names = ["image", "boxes", "im_info", "text", "relationship_label", "mlm_labels", "mvrc_ops", "mvrc_labels"]

    def __len__(self):
        return len(self.database)

    def _bytes(self, path):
        if ZIP_AT in path:
            return self.archives.read(path)
        with open(path, "rb") as f:
            return f.read()

    def _image(self, path):
        from PIL import Image
        return Image.open(io.BytesIO(self._bytes(path))).convert("RGB")

    def __getitem__(self, index):
        rec = self.database[index]
        det = R.decode_detector_record(self._bytes(os.path.join(self.data_path, rec["frcnn"])).decode(),
                                       with_features=self.with_precomputed_visual_feat)
        boxes, scores = torch.as_tensor(det["boxes"].copy()), det["scores"]
        feats = torch.as_tensor(det["features"].copy()) if self.with_precomputed_visual_feat else None
        image, (w0, h0) = None, (det["width"], det["height"])
        if not self.with_precomputed_visual_feat:
            try:
                image = self._image(os.path.join(self.data_path, rec["image"]))
                w0, h0 = image.size
            except Exception:      # the reference trains on with a zero image of the recorded size (conceptual_captions.py:121-128)
                print("Failed to load image {}, use zero image!".format(rec["image"]))
        if self.add_image_as_a_box:
            boxes, feats = R.prepend_whole_image(boxes, feats, w0, h0)
        im_info = torch.tensor([w0, h0, 1.0, 1.0, index])
        if self.transform is not None:
            image, boxes, _, im_info = self.transform(image, boxes, None, im_info)
        if image is None and not self.with_precomputed_visual_feat:
            image = im_info.new_zeros((3, int(im_info[1].item()), int(im_info[0].item())), dtype=torch.float)
        w, h = im_info[0].item(), im_info[1].item()
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clamp(min=0, max=w - 1)
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clamp(min=0, max=h - 1)

        # caption-image relationship: the draw is made whether or not the task is on
        u = random.random()
        related = u < 0.5 or not self.with_rel_task
        caption = rec["caption"]
        if not related:
            other = random.randrange(0, len(self.database))
            while other == index:
                other = random.randrange(0, len(self.database))
            caption = self.database[other]["caption"]
        sentence = " ".join(caption)
        if self.with_mlm_task:
            pieces, labels = R.mask_whole_words(self.tokenizer.basic_tokenizer.tokenize(sentence), self.tokenizer)
        else:
            pieces = self.tokenizer.tokenize(sentence)
            labels = [-1] * len(pieces)
        tokens, mlm_labels = ["[CLS]"] + pieces + ["[SEP]"], [-1] + labels + [-1]

        n = boxes.shape[0]
        if self.with_mvrc_task:
            ops, soft = R.mask_regions(scores)
            if self.add_image_as_a_box:      # the whole-image slot is never a target
                ops, soft = [0] + ops, [np.zeros_like(scores[0])] + soft
                if self.with_precomputed_visual_feat:
                    R.unmix_masked_regions(feats, ops)
            if len(ops) != n or len(soft) != n:
                raise AssertionError("mvrc_ops / mvrc_labels have length %d / %d, expected %d" % (len(ops), len(soft), n))
        else:
            ops, soft = [0] * n, [np.zeros_like(scores[0])] * n
        if image is not None and self.mask_raw_pixels and not self.with_precomputed_visual_feat:
            R.zero_masked_pixels(image, boxes, ops)
        soft = np.stack(soft, axis=0)
        text = self.tokenizer.convert_tokens_to_ids(tokens)
        if self.with_precomputed_visual_feat:
            boxes = torch.cat((boxes, feats), dim=1)
        if len(text) + n > self.seq_len:
            keep_t, keep_b = R.sequence_budget(len(text), n, self.seq_len)
            boxes, ops, soft = boxes[:keep_b], ops[:keep_b], soft[:keep_b]
            text = text[:keep_t - 1] + text[-1:]
            mlm_labels = mlm_labels[:keep_t - 1] + mlm_labels[-1:]
        return image, boxes, im_info, text, int(related), mlm_labels, ops, soft


class GeneralCorpus(Dataset):
    def __init__(self, ann_file, pretrained_model_name=None, tokenizer=None, seq_len=64, min_seq_len=64, encoding="utf-8", on_memory=True,
                 **kwargs):
        if not on_memory:
            raise NotImplementedError("only the in-memory corpus exists (general_corpus.py:11)")
        self.tokenizer = tokenizer if tokenizer is not None else default_tokenizer(pretrained_model_name)
        self.vocab = self.tokenizer.vocab
        self.seq_len, self.min_seq_len, self.ann_file, self.encoding, self.test_mode = seq_len, min_seq_len, ann_file, encoding, False
        lines = []
        for path in ann_file.split("+"):      # 'a.doc+b.doc': several corpora, one line per sample
            with open(path, "r", encoding=encoding) as f:
                lines.extend(line.strip("\n").strip("\r").strip("\n") for line in f.readlines())
        self.corpus = [line.strip() for line in lines if line.strip() != ""]

    data_names = ["text", "mlm_labels"]

    def __len__(self):
        return len(self.corpus)

    def __getitem__(self, item):
        words = self.tokenizer.basic_tokenizer.tokenize(self.corpus[item])
        nxt = (item + 1) % len(self.corpus)
        while len(words) < self.min_seq_len:      # short lines borrow the following ones
            words.extend(self.tokenizer.basic_tokenizer.tokenize(self.corpus[nxt]))
            nxt = (nxt + 1) % len(self.corpus)
        pieces, labels = R.mask_whole_words(words, self.tokenizer)
        ids = self.tokenizer.convert_tokens_to_ids(pieces)
        return ids[:self.seq_len], labels[:self.seq_len]


DATASET_CATALOGS = {"conceptual_captions": ConceptualCaptionsDataset, "general_corpus": GeneralCorpus}
