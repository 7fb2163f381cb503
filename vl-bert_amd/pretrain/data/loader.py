"""Data loaders from a reference-style config (pretrain/data/build.py:41-140) + the multitask zip (common/utils/multi_task_dataloader.py).

`make_dataloader(cfg, mode, distributed, num_replicas, rank)` reads the DATASET.* / NETWORK.* / TRAIN.* keys the reference reads and returns a
torch DataLoader whose batches are tuples in `dataset.data_names` order -- `engine.set_batch(*batch[1:], image=batch[0])` for the
image-caption datasets.  With a LIST under DATASET (the multitask yamls) `make_dataloaders` returns one loader per entry and
`MultiTaskDataLoader` concatenates their batches, the first loader defining the epoch.
One process per GPU here: the batch size is TRAIN.BATCH_IMAGES per process (the reference multiplies by len(GPUS) for its single-process
DataParallel mode, build.py:49-51 -- pass gpus_per_process to reproduce that).
"""
import copy
import math

import torch
from torch.utils.data import BatchSampler, DataLoader, RandomSampler, Sampler, SequentialSampler

from .collate import BatchCollator
from .datasets import DATASET_CATALOGS
from .transforms import build_transforms


class DistributedSampler(Sampler):
    """Rank r sees the r-th CONTIGUOUS slice of the (epoch-seeded) permutation, the list padded by wrap-around to a multiple of the world
    size (pretrain/data/samplers/distributed.py:41-60 -- note: torch's own sampler strides instead)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank, self.shuffle, self.epoch = dataset, num_replicas, rank, shuffle, 0
        self.num_samples = int(math.ceil(len(dataset) / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(self.epoch)).tolist() if self.shuffle else list(range(n))
        order += order[:self.total_size - n]
        return iter(order[self.rank * self.num_samples:(self.rank + 1) * self.num_samples])


def _get(node, key, default=None):
    return node[key] if key in node else default


def make_dataloader(cfg, dataset=None, mode="train", distributed=False, num_replicas=None, rank=None, expose_sampler=False,
                    gpus_per_process=1, tokenizer=None, drop_last=False):
    part = {"train": cfg.TRAIN, "val": _get(cfg, "VAL", {}), "test": _get(cfg, "TEST", {})}[mode]
    ds = cfg.DATASET
    if dataset is None:
        name = ds.DATASET
        if name not in DATASET_CATALOGS:
            raise KeyError("DATASET.DATASET %r: this package reads %s" % (name, sorted(DATASET_CATALOGS)))
        key = mode.upper()
        dataset = DATASET_CATALOGS[name](
            ann_file=_get(ds, key + "_ANNOTATION_FILE", ""), image_set=_get(ds, key + "_IMAGE_SET", ""),
            seq_len=_get(ds, "SEQ_LEN", 64), min_seq_len=_get(ds, "MIN_SEQ_LEN", 0),
            with_precomputed_visual_feat=cfg.NETWORK.IMAGE_FEAT_PRECOMPUTED, mask_raw_pixels=_get(cfg.NETWORK, "MASK_RAW_PIXELS", True),
            with_rel_task=cfg.NETWORK.WITH_REL_LOSS, with_mlm_task=cfg.NETWORK.WITH_MLM_LOSS, with_mvrc_task=cfg.NETWORK.WITH_MVRC_LOSS,
            root_path=_get(ds, "ROOT_PATH", ""), data_path=_get(ds, "DATASET_PATH", ""), test_mode=(mode == "test"),
            transform=build_transforms(cfg, mode), zip_mode=_get(ds, "ZIP_MODE", False), cache_mode=_get(ds, "CACHE_MODE", False),
            cache_db=(rank is None or rank == 0), ignore_db_cache=_get(ds, "IGNORE_DB_CACHE", True),
            add_image_as_a_box=_get(ds, "ADD_IMAGE_AS_A_BOX", True), aspect_grouping=bool(_get(part, "ASPECT_GROUPING", False)),
            pretrained_model_name=_get(cfg.NETWORK, "BERT_MODEL_NAME", None), tokenizer=tokenizer)
    shuffle = bool(_get(part, "SHUFFLE", mode == "train"))
    if distributed:
        sampler = DistributedSampler(dataset, num_replicas=num_replicas, rank=rank, shuffle=shuffle)
    else:
        sampler = RandomSampler(dataset) if shuffle else SequentialSampler(dataset)
    batch = int(part.BATCH_IMAGES) * gpus_per_process
    loader = DataLoader(dataset, batch_sampler=BatchSampler(sampler, batch, drop_last=drop_last),
                        num_workers=int(_get(cfg, "NUM_WORKERS_PER_GPU", 0)) * gpus_per_process, pin_memory=False,
                        collate_fn=BatchCollator(dataset, append_ind=bool(_get(ds, "APPEND_INDEX", False))))
    return (loader, sampler) if expose_sampler else loader


def make_dataloaders(cfg, mode="train", **kw):
    """cfg.DATASET is a list and the BATCH_IMAGES of every mode lists: one loader per entry (build.py:121-140)."""
    out = []
    for i, entry in enumerate(cfg.DATASET):
        one = copy.deepcopy(cfg)
        one["DATASET"] = entry
        for sect in ("TRAIN", "VAL", "TEST"):
            if sect in one and isinstance(_get(one[sect], "BATCH_IMAGES"), (list, tuple)):
                one[sect]["BATCH_IMAGES"] = one[sect]["BATCH_IMAGES"][i]
        out.append(make_dataloader(one, mode=mode, **kw))
    return out


class MultiTaskDataLoader:
    """Batches of several loaders side by side; the FIRST loader is the master (its length is the epoch), the others restart when they run
    out (and move their sampler's epoch along)."""

    def __init__(self, loaders):
        if len(loaders) < 2:
            raise ValueError("MultiTaskDataLoader needs at least two loaders")
        self.loaders = list(loaders)
        self.streams = [iter(l) for l in self.loaders]
        self.lens = [len(l) for l in self.loaders]
        self.batches_served = 0

    def __len__(self):
        return self.lens[0]

    def __iter__(self):
        if self.batches_served > 0:
            self.streams[0] = iter(self.loaders[0])
        return self

    def __next__(self):
        out = tuple(next(self.streams[0]))      # StopIteration of the master ends the epoch
        for k in range(1, len(self.loaders)):
            sampler = getattr(getattr(self.loaders[k], "batch_sampler", None), "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(int(self.batches_served / self.lens[k]))
            try:
                extra = next(self.streams[k])
            except StopIteration:
                self.streams[k] = iter(self.loaders[k])
                extra = next(self.streams[k])
            out += tuple(extra)
        self.batches_served += 1
        return out
