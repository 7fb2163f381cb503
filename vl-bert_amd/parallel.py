"""Data-parallel gradient exchange for the flat gradient buffer -- RCCL over xGMI.

Replaces torch.nn.parallel.DistributedDataParallel / apex DDP at pretrain/function/train.py:89-90,353-354.
One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Because the backward is hand-scheduled (engine.backward), gradient readiness is known statically:
the flat buffer is laid out [embeddings | obj_downsample | LayerNorms | layer 0 .. L-1 | heads] and
backward finishes heads first, then layers L-1 .. 0, then the embedding side.  Buckets are therefore
CONTIGUOUS slices of the flat buffer; each is all-reduced (SUM) with async_op=True as soon as the
kernels producing it have been enqueued: the collective waits (on the communicator's stream) for
exactly those kernels and then overlaps with the rest of backward.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU) so ring all-reduce is per-link bound: few, large buckets (default
~64 MB fp32, i.e. 2-3 encoder layers) keep every link busy without serialising on launch latency.
The 1/world_size average is folded into the AdamW kernel's grad_scale (no extra pass).

Wire format: bf16 by default (VLB_DP_WIRE=fp32 keeps fp32).  Each bucket is converted into its slice of ONE persistent flat bf16
image right when it becomes ready (a 16 B/lane HIP cast, ~20 us per 64 MB bucket, under the rest of backward), RCCL reduces that
slice in place, and the optimizer (vlb_sumsq_bf16_det / vlb_adamw_step_gbf16) consumes the reduced image as it arrived: half
the bytes on every xGMI link and no conversion pass back to fp32.  Every rank receives the same reduced bits, so replicas stay
bit-identical.  The tied word-embedding gradient (94 MB fp32: decoder wgrad + embedding scatter-add) completes last and cannot
overlap; it is its own bucket ("word_emb"), launched right after the embedding backward, in front of the remaining front-end
gradients ("embed": position / type tables, obj_downsample, the three input LayerNorms).

Gradient accumulation: call `reduce_*` only on the boundary micro-step (the reference all-reduces on
every micro-batch, common/trainer.py:117-118,132-153).
"""
import os

import torch
import torch.distributed as dist

_WIRE = {"bf16": torch.bfloat16, "fp32": None, "f32": None}


def default_wire_dtype():
    v = os.environ.get("VLB_DP_WIRE", "bf16").lower()
    if v not in _WIRE:
        raise ValueError("VLB_DP_WIRE must be bf16 or fp32 (got %r)" % v)
    return _WIRE[v]


class GradBuckets:
    def __init__(self, flat_grad, offsets, numel, num_layers, group=None, bucket_bytes=64 << 20, wire_dtype="default", vision_start=None):
        """flat_grad: the flat fp32 gradient tensor; offsets: {param name: start offset} in layout order.
        vision_start: offset of the e2e convolution weights appended after the heads (their gradients are produced LAST, after the
        embedding side, by vision.VisionStack.backward) -- they form their own bucket, launched on on_done("vision")."""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.wire_dtype = default_wire_dtype() if wire_dtype == "default" else wire_dtype
        if self.wire_dtype == flat_grad.dtype:
            self.wire_dtype = None
        # persistent wire image (same offsets as the flat gradient); the optimizer reads `reduced`
        self.wire = torch.zeros(numel, dtype=self.wire_dtype, device=flat_grad.device) if self.wire_dtype is not None else None
        names = list(offsets)
        layer_start = [offsets["vlbert.encoder.layer.%d.attention.self.query.weight" % l] for l in range(num_layers)]
        head_start = offsets["vlbert.mlm_head.predictions.transform.dense.weight"]
        bounds = layer_start + [head_start]
        tail = numel if vision_start is None else vision_start
        front_end = layer_start[0] if num_layers else head_start
        we_end = offsets["vlbert.position_embeddings.weight"]      # the word-embedding table is first in the layout
        self.ranges = {"heads": (head_start, tail), "word_emb": (0, we_end), "embed": (we_end, front_end)}
        self.vision_keys = []
        if vision_start is not None:
            # one bucket per stage of the vision path, in layout order (layer2 | layer3 | RoI head): VisionStack.backward finishes
            # the RoI head first and the trunk stage by stage, so each stage's 40-100 MB all-reduce overlaps the rest of it
            def stage(n):
                if "roi_head_feature_extractor" in n:
                    return "vision4"
                for L in (1, 2, 3):
                    if ".backbone.layer%d." % L in n:
                        return "vision%d" % L
                return "vision0"
            vnames = [n for n in names if offsets[n] >= vision_start]
            starts = []
            for n in vnames:
                k = stage(n)
                if not starts or starts[-1][0] != k:
                    assert k not in [q for q, _ in starts], "vision parameters of one stage must be contiguous in the flat buffer"
                    starts.append((k, offsets[n]))
            for i, (k, lo) in enumerate(starts):
                self.ranges[k] = (lo, starts[i + 1][1] if i + 1 < len(starts) else numel)
                self.vision_keys.append(k)
        self.launched = set()
        # group consecutive layers (in backward order) into buckets of ~bucket_bytes
        self.layer_bucket = {}
        cur, cur_bytes = [], 0
        for l in reversed(range(num_layers)):
            cur.append(l)
            cur_bytes += (bounds[l + 1] - bounds[l]) * 4
            if cur_bytes >= bucket_bytes or l == 0:
                lo, hi = bounds[min(cur)], bounds[max(cur) + 1]
                self.layer_bucket[min(cur)] = (lo, hi)     # ready once its LOWEST layer finished backward
                cur, cur_bytes = [], 0
        self.pending = []
        assert names[0] in offsets

    def coverage(self):
        """All ranges, for tests: they must tile [0, numel) exactly."""
        r = [self.ranges["word_emb"], self.ranges["embed"]] + sorted(self.layer_bucket.values()) + [self.ranges["heads"]]
        r += [self.ranges[k] for k in self.vision_keys]
        return r

    def _launch(self, lo, hi):
        if self.world == 1:
            return
        t = self.flat[lo:hi]
        if self.wire is not None:
            w = self.wire[lo:hi]
            if t.is_cuda:
                from . import ops
                ops.cast_f32_bf16(t, w)           # HIP kernel on the compute stream; the collective below waits for it
            else:
                w.copy_(t)                        # (gloo CPU tests)
            work = dist.all_reduce(w, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(t, group=self.group, async_op=True)
        self.pending.append(work)

    def on_done(self, what):
        """engine.backward hook: `what` is "heads", a layer index, "embed", "vision<stage>" or "vision" (= every vision stage not
        reduced yet).  A range is launched once per step (wait() re-arms)."""
        if what == "vision":
            for k in self.vision_keys:
                self.on_done(k)
            return
        if what in self.launched:
            return
        self.launched.add(what)
        if what in self.vision_keys:
            self._launch(*self.ranges[what])
        elif what == "heads":
            # the tied word-embedding gradient is only complete after the embedding backward -> it lives in
            # the "embed" range; the head range holds transform / decoder bias / MVRC head gradients
            self._launch(*self.ranges["heads"])
        elif what == "word_emb":
            self._launch(*self.ranges["word_emb"])
        elif what == "embed":
            self.on_done("word_emb")          # (callers without the finer hook)
            self._launch(*self.ranges["embed"])
        elif what in self.layer_bucket:
            self._launch(*self.layer_bucket[what])

    def will_launch(self, what):
        """True when on_done(what) would start a collective now (engine.backward joins its side stream only then)."""
        if self.world == 1 or what in self.launched:
            return False
        return what in self.layer_bucket or what in self.ranges or what in ("vision", "embed")

    def wait(self):
        for work in self.pending:
            work.wait()
        self.pending = []
        self.launched = set()

    @property
    def reduced(self):
        """The tensor holding the reduced (SUM over ranks) gradient after wait(): the bf16 wire image, or the flat fp32 buffer."""
        return self.wire if self.wire is not None else self.flat

    @property
    def grad_scale(self):
        return 1.0 / self.world
