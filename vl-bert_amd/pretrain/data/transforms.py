"""Image / box transforms of the pre-training loaders without torchvision (absent from the MI355X image): PIL + torch.

Contract = pretrain/data/transforms/{build,transforms}.py: every transform maps (image, boxes, masks, im_info) to the same tuple;
`build_transforms(cfg, mode)` = Resize(SCALES) -> RandomHorizontalFlip(<MODE>.FLIP_PROB) -> ToTensor -> Normalize(PIXEL_MEANS, PIXEL_STDS,
BGR x 255).  `image` may be None (precomputed features): the box / im_info arithmetic still runs and the flip still consumes its draw.
"""
import random

import numpy as np
import torch


def scaled_size(width, height, short, longest):
    """Target (width, height): short side -> `short` unless the long side would pass `longest` (transforms.py:33-55; integer truncation
    as there)."""
    lo, hi = float(min(width, height)), float(max(width, height))
    if longest is not None and hi / lo * short > longest:
        short = int(longest * lo / hi)
    if (width <= height and width == short) or (height <= width and height == short):
        return width, height
    if width < height:
        return short, int(short * height / width)
    return int(short * width / height), short


class Resize:
    def __init__(self, min_size, max_size):
        self.min_size, self.max_size = min_size, max_size

    def __call__(self, image, boxes, masks, im_info):
        w0, h0 = im_info[0], im_info[1]
        w, h = scaled_size(w0, h0, self.min_size, self.max_size)
        if image is not None:
            from PIL import Image
            image = image.resize((int(w), int(h)), Image.BILINEAR)      # (torchvision's functional.resize on a PIL image)
        rx, ry = w * 1.0 / w0, h * 1.0 / h0
        if boxes is not None:
            boxes[:, [0, 2]] *= rx
            boxes[:, [1, 3]] *= ry
        im_info[0], im_info[1] = w, h
        im_info[2], im_info[3] = rx, ry
        return image, boxes, masks, im_info


class RandomHorizontalFlip:
    def __init__(self, prob=0.5, rng=random):
        self.prob, self.rng = prob, rng

    def __call__(self, image, boxes, masks, im_info):
        if self.rng.random() < self.prob:
            w = im_info[0]
            if image is not None:
                from PIL import Image
                image = image.transpose(Image.FLIP_LEFT_RIGHT)
            if boxes is not None:
                boxes[:, [0, 2]] = w - 1 - boxes[:, [2, 0]]
            if masks is not None:
                masks = torch.flip(masks, dims=(2,))
        return image, boxes, masks, im_info


class ToTensor:
    def __call__(self, image, boxes, masks, im_info):
        if image is not None:      # HWC uint8 -> CHW float in [0, 1]
            image = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
        return image, boxes, masks, im_info


class Normalize:
    def __init__(self, mean, std, to_bgr255=True):
        self.mean, self.std, self.to_bgr255 = mean, std, to_bgr255

    def __call__(self, image, boxes, masks, im_info):
        if image is not None:
            if self.to_bgr255:
                image = image[[2, 1, 0]] * 255
            mean = torch.as_tensor(self.mean, dtype=image.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=image.dtype).view(-1, 1, 1)
            image = (image - mean) / std
        return image, boxes, masks, im_info


class Compose:
    def __init__(self, steps):
        self.steps = list(steps)

    def __call__(self, image, boxes, masks, im_info):
        for step in self.steps:
            image, boxes, masks, im_info = step(image, boxes, masks, im_info)
        return image, boxes, masks, im_info


def build_transforms(cfg, mode="train"):
    flip = (cfg.get(mode.upper()) or {}).get("FLIP_PROB", 0.5 if mode == "train" else 0)      # defaults of pretrain/function/config.py:137,166,174
    short, longest = cfg.SCALES[0], cfg.SCALES[1]
    if short > longest:
        raise ValueError("SCALES must be (short side, long side)")
    return Compose([Resize(short, longest), RandomHorizontalFlip(flip), ToTensor(),
                    Normalize(cfg.NETWORK.PIXEL_MEANS, cfg.NETWORK.PIXEL_STDS, to_bgr255=True)])
