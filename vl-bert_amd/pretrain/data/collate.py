"""Batch collation in the layout engine.set_batch() and the module mirrors consume (pretrain/data/collate_batch.py:5-74).

Per field: images zero-padded to the batch's largest height / width; boxes padded with -2 rows to the batch's largest box count (box_mask =
boxes[..., 0] > -1.5 downstream); text with 0; mlm_labels with -1; mvrc_ops with 0; mvrc_labels with zero rows; every other field stacked
as it is.  Output: one tensor per entry of `dataset.data_names`, in that order (None for an absent image).
"""
import torch

_ROW_PAD = {"boxes": -2, "mvrc_labels": 0}
_SEQ_PAD = {"text": 0, "mlm_labels": -1, "mvrc_ops": 0}
_BOX_SIDED = ("boxes", "mvrc_ops", "mvrc_labels")


def _padded(rows, length, value, trailing=()):
    rows = [torch.as_tensor(r) for r in rows]
    out = torch.full((len(rows), length) + tuple(trailing), value, dtype=rows[0].dtype)
    for i, r in enumerate(rows):
        k = min(length, r.shape[0])
        out[i, :k] = r[:k]
    return out


class BatchCollator:
    def __init__(self, dataset, append_ind=False):
        self.data_names = list(dataset.data_names)
        self.test_mode = getattr(dataset, "test_mode", False)
        self.append_ind = append_ind

    def __call__(self, samples):
        samples = list(samples)
        col = {name: [s[i] for s in samples] for i, name in enumerate(self.data_names)}
        n_boxes = max(torch.as_tensor(b).shape[0] for b in col["boxes"]) if "boxes" in col else 0
        n_text = max(len(t) for t in col["text"]) if "text" in col else 0
        out = []
        for name in self.data_names:
            items = col[name]
            if name == "image":
                if items[0] is None:
                    out.append(None)
                    continue
                c = items[0].shape[0]
                h, w = max(im.shape[1] for im in items), max(im.shape[2] for im in items)
                batch = torch.zeros((len(items), c, h, w), dtype=items[0].dtype)
                for i, im in enumerate(items):
                    batch[i, :, :im.shape[1], :im.shape[2]] = im
                out.append(batch)
            elif name in _ROW_PAD:
                first = torch.as_tensor(items[0])
                out.append(_padded(items, n_boxes, _ROW_PAD[name], trailing=first.shape[1:]))
            elif name in _SEQ_PAD:
                out.append(_padded(items, n_boxes if name in _BOX_SIDED else n_text, _SEQ_PAD[name]))
            else:
                out.append(torch.stack([torch.as_tensor(v) for v in items], dim=0))
        if self.append_ind:
            out.append(torch.arange(len(samples), dtype=torch.int64))
        return tuple(out)
