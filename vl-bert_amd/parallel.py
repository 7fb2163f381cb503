"""Data-parallel gradient exchange for the flat gradient buffer -- RCCL over xGMI.

Replaces torch.nn.parallel.DistributedDataParallel / apex DDP at pretrain/function/train.py:89-90,353-354.
One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

Because the backward is hand-scheduled (engine.backward), gradient readiness is known statically:
the flat buffer is laid out [embeddings | obj_downsample | LayerNorms | layer 0 .. L-1 | heads] and
backward finishes heads first, then layers L-1 .. 0, then the embedding side.  Buckets are therefore
CONTIGUOUS slices of the flat buffer; each is exchanged with async_op=True as soon as the kernels
producing it have been enqueued: the collective waits (on the communicator's stream) for exactly those
kernels and then overlaps with the rest of backward.  xGMI is point-to-point (7 links x ~153 GB/s per
GPU) so ring collectives are per-link bound: few, large buckets (default ~64 MB fp32, i.e. 2-3 encoder
layers) keep every link busy without serialising on launch latency.  The 1/world_size average is folded
into the AdamW kernel's grad_scale (no extra pass).

Two exchange modes (VLB_DP_MODE, default "sharded" when the buffer can be cut evenly):

* "allreduce": every bucket is all-reduced (SUM); every rank then runs the whole clip + AdamW (what DDP implies).
* "sharded": the optimizer is sharded over the ranks (ZeRO-1 style, MI355X-first: the replicated AdamW streams 3.4 GB per step on
  EVERY rank -- 0.6 ms, 10 % of a 32-sample step -- for an update all ranks compute identically).  A bucket is REDUCE-SCATTERED: rank r
  receives the reduced r-th slice of it into a compact image; the clip norm is the all-reduced sum of the ranks' partial sums of
  squares; AdamW runs on the owned slices only (1/world of the traffic) and emits the bf16 working copy of those slices, which an
  ALL-GATHER per bucket distributes in forward order, overlapped with the next step's forward (engine.forward waits per bucket
  right before the first use).  Wire bytes equal the all-reduce's (reduce-scatter + all-gather are its two halves); fp32 master
  weights and Adam moments are authoritative on the owner only (engine.state_dict() gathers the master: it is a collective).
  Bucket boundaries are moved onto a 64 x world grid so that every bucket splits evenly -- always INTO the bucket that fires
  later, so a slice never leaves before its last producer ran.

Wire format: bf16 by default up to 8 ranks (VLB_DP_WIRE=fp32 keeps fp32; above 8 ranks the default is fp32 -- a bf16 running sum
over many ranks is not validated, tests/test_parallel_cpu.py bounds the 8-rank case against the fp32 sum).  Each bucket is converted
into its slice of ONE persistent flat bf16 image right when it becomes ready (a 16 B/lane HIP cast, ~20 us per 64 MB bucket, under
the rest of backward), RCCL reduces that slice, and the optimizer (vlb_sumsq_*_det / vlb_adamw_step_*) consumes the reduced image as
it arrived: half the bytes on every xGMI link and no conversion pass back to fp32.  Every rank receives the same reduced bits, so
replicas stay bit-identical.  The tied word-embedding gradient (94 MB fp32: decoder wgrad + embedding scatter-add) completes last
and cannot overlap; it is its own bucket ("word_emb"), launched right after the embedding backward, in front of the remaining
front-end gradients ("embed": position / type tables, obj_downsample, the three input LayerNorms).

Gradient accumulation: call `reduce_*` only on the boundary micro-step (the reference all-reduces on
every micro-batch, common/trainer.py:117-118,132-153).
"""
import os

import torch
import torch.distributed as dist

def _wire16():
    """The 16-bit wire type = the library's 16-bit type (bfloat16; IEEE fp16 in the VLB_PRECISION=f16 build, whose cast kernel emits fp16)."""
    from . import _lib
    return _lib.act_torch_dtype()


_WIRE = {"bf16": "16", "f16": "16", "fp16": "16", "16": "16", "fp32": None, "f32": None}
BF16_WIRE_MAX_WORLD = 8      # validated bound of the bf16 running sum (tests/test_parallel_cpu.py::test_bf16_wire_error_world8)


def default_wire_dtype(world=1):
    v = os.environ.get("VLB_DP_WIRE", "").lower()
    if not v:
        return _wire16() if world <= BF16_WIRE_MAX_WORLD else None
    if v not in _WIRE:
        raise ValueError("VLB_DP_WIRE must be bf16 (the library's 16-bit type) or fp32 (got %r)" % v)
    return _wire16() if _WIRE[v] else None


def default_mode():
    v = os.environ.get("VLB_DP_MODE", "sharded").lower()
    if v not in ("sharded", "allreduce"):
        raise ValueError("VLB_DP_MODE must be sharded or allreduce (got %r)" % v)
    return v


def force_exchange():
    """VLB_DP_FORCE_EXCHANGE=1 and an initialised process group: run the data-parallel exchange even in a world of one."""
    return os.environ.get("VLB_DP_FORCE_EXCHANGE", "0") == "1" and dist.is_available() and dist.is_initialized()


def shard_alignment(world):
    """Element grid every bucket boundary (and the flat buffer's length) lies on in sharded mode: a bucket then splits into `world`
    slices of whole 64-element (256-B fp32 / 128-B bf16) units."""
    return 64 * max(1, int(world))


class GradBuckets:
    def __init__(self, flat_grad, offsets, numel, num_layers, group=None, bucket_bytes=64 << 20, wire_dtype="default", vision_start=None,
                 mode="default", emulate_collectives=None):
        """flat_grad: the flat fp32 gradient tensor; offsets: {param name: start offset} in layout order.
        vision_start: offset of the e2e convolution weights appended after the heads (their gradients are produced LAST, after the
        embedding side, by vision.VisionStack.backward) -- they form their own buckets, launched on on_done("vision<stage>").
        mode: "allreduce" | "sharded" | "default" (VLB_DP_MODE; sharded needs numel % shard_alignment(world) == 0, else allreduce).
        emulate_collectives: carry reduce-scatter / all-gather as all-reduces (backends without them for device tensors: gloo over
        CUDA tensors, which is how two ranks share one GPU in tools/dp2_check.py); default: decided from the backend."""
        self.flat = flat_grad
        self.group = group
        self.numel = numel
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # single: nothing to exchange (a world of one).  VLB_DP_FORCE_EXCHANGE=1 keeps the whole exchange ON in a world of one -- every
        # collective is then the identity, but the communicator's reduce-scatter / all-gather / all-reduce calls, their stream
        # ordering and the graph segments cut around them all execute: the way a 1-GPU box runs the RCCL code path (tests/test_dp_gpu.py)
        self._force = force_exchange()
        self.wire_dtype = default_wire_dtype(self.world) if wire_dtype == "default" else wire_dtype
        if self.wire_dtype == flat_grad.dtype:
            self.wire_dtype = None
        mode = default_mode() if mode == "default" else mode
        if mode not in ("sharded", "allreduce"):
            raise ValueError("mode must be sharded or allreduce")
        G = shard_alignment(self.world)
        if mode == "sharded" and numel % G:
            mode = "allreduce"       # (a flat buffer that was not padded for this world size: engine.FlatParams(align=...))
        self.sharded = mode == "sharded"
        if emulate_collectives is None:
            backend = dist.get_backend(group) if dist.is_initialized() else "none"
            emulate_collectives = backend == "gloo" and flat_grad.is_cuda
        self.emulate = bool(emulate_collectives)
        # persistent wire image (same offsets as the flat gradient); the optimizer reads `reduced` (allreduce) / `grad_shard` (sharded)
        self.wire = torch.zeros(numel, dtype=self.wire_dtype, device=flat_grad.device) if self.wire_dtype is not None else None
        names = list(offsets)
        layer_start = [offsets["vlbert.encoder.layer.%d.attention.self.query.weight" % l] for l in range(num_layers)]
        head_start = offsets["vlbert.mlm_head.predictions.transform.dense.weight"]
        tail = numel if vision_start is None else vision_start
        front_end = layer_start[0] if num_layers else head_start
        we_end = offsets["vlbert.position_embeddings.weight"]      # the word-embedding table is first in the layout
        # ---- buckets in ADDRESS order: (key, lo, hi, fire) ; fire = position in the engine's completion order -------------------
        # group consecutive layers (in backward order) into buckets of ~bucket_bytes; a bucket is ready once its LOWEST layer finished
        bounds = layer_start + [head_start]
        groups, cur, cur_bytes = [], [], 0
        for l in reversed(range(num_layers)):
            cur.append(l)
            cur_bytes += (bounds[l + 1] - bounds[l]) * 4
            if cur_bytes >= bucket_bytes or l == 0:
                groups.append((min(cur), bounds[min(cur)], bounds[max(cur) + 1], max(cur)))
                cur, cur_bytes = [], 0
        vis = []
        if vision_start is not None:
            # one bucket per stage of the vision path, in layout order (layer2 | layer3 | RoI head): VisionStack.backward finishes
            # the RoI head first and the trunk stage by stage, so each stage's 40-100 MB exchange overlaps the rest of it
            def stage(n):
                if "roi_head_feature_extractor" in n:
                    return "vision4"
                for L in (1, 2, 3):
                    if ".backbone.layer%d." % L in n:
                        return "vision%d" % L
                return "vision0"
            starts = []
            for n in [n for n in names if offsets[n] >= vision_start]:
                k = stage(n)
                if not starts or starts[-1][0] != k:
                    assert k not in [q for q, _ in starts], "vision parameters of one stage must be contiguous in the flat buffer"
                    starts.append((k, offsets[n]))
            vis = [(k, lo, starts[i + 1][1] if i + 1 < len(starts) else numel) for i, (k, lo) in enumerate(starts)]
        # completion order: heads, layer buckets from the top, word_emb, embed, then the vision stages from the RoI head down
        order = ["heads"] + [g[0] for g in groups] + ["word_emb", "embed"] + [k for k, _, _ in reversed(vis)]
        fire = {k: i for i, k in enumerate(order)}
        addr = [("word_emb", 0, we_end), ("embed", we_end, front_end)] + [(g[0], g[1], g[2]) for g in sorted(groups, key=lambda g: g[1])] \
            + [("heads", head_start, tail)] + vis
        if self.sharded:
            # boundaries onto the G grid, each moved INTO the later-firing neighbour (its elements wait for that bucket's launch)
            cuts = [0]
            for (ka, _, hi), (kb, _, _) in zip(addr[:-1], addr[1:]):
                up = fire[ka] > fire[kb]          # the lower bucket fires later: it takes the first elements of the upper one
                cuts.append((hi + G - 1) // G * G if up else hi // G * G)
            cuts.append(numel)
            addr = [(k, cuts[i], cuts[i + 1]) for i, (k, _, _) in enumerate(addr)]
            for k, lo, hi in addr:
                if hi <= lo:
                    raise ValueError("sharded exchange: bucket %r is smaller than the %d-element shard grid" % (k, G))
        self.buckets = addr
        self.ranges = {k: (lo, hi) for k, lo, hi in addr if not isinstance(k, int)}
        self.layer_bucket = {k: (lo, hi) for k, lo, hi in addr if isinstance(k, int)}
        self.layer_key = {}                       # encoder layer -> key of the bucket that holds (the bulk of) its parameters
        for key, _, _, top in groups:
            for l in range(key, top + 1):
                self.layer_key[l] = key
        self.vision_keys = [k for k, _, _ in vis]
        self.launched = set()
        self.pending = []
        self._complete = self.single
        self._null = False                        # null_collectives(): timing mode, every collective call skipped
        # ---- sharded mode: compact images + the forward-order weight gather --------------------------------------------------
        if self.sharded:
            wdt = self.wire_dtype if self.wire_dtype is not None else flat_grad.dtype
            self.gshard = torch.zeros(numel // self.world, dtype=wdt, device=flat_grad.device)
            # forward order: the vision stages (the CNN runs first), the front end, the encoder from layer 0 up, the heads
            self.gather_order = list(self.vision_keys) + ["word_emb", "embed"] + sorted(self.layer_bucket) + ["heads"]
            self._gather_idx = {k: i for i, k in enumerate(self.gather_order)}
            self._gathers = []                    # [(index in gather_order, work, post)] still in flight
            self._stage32 = None
        self._repl = None                         # set_replicated_fp32()
        assert names[0] in offsets

    @property
    def single(self):
        """Nothing to exchange: a world of one without VLB_DP_FORCE_EXCHANGE (derived, so that `world` stays the one source of truth)."""
        return self.world == 1 and not self._force

    # ------------------------------------------------------------------------------------------------------------------
    def coverage(self):
        """All ranges in address order, for tests: they must tile [0, numel) exactly."""
        return [(lo, hi) for _, lo, hi in self.buckets]

    def _range(self, key):
        return self.layer_bucket[key] if isinstance(key, int) else self.ranges[key]

    def piece(self, key):
        """(offset into the flat buffers, offset into the compact images, length) of the slice of bucket `key` this rank owns."""
        lo, hi = self._range(key)
        n = (hi - lo) // self.world
        return lo + self.rank * n, lo // self.world, n

    def owned_rows(self):
        """Rows for ops.ShardRanges: this rank's slice of every bucket, in address order."""
        return [self.piece(k) for k, _, _ in self.buckets]

    def _cast_to_wire(self, lo, hi):
        t, w = self.flat[lo:hi], self.wire[lo:hi]
        if t.is_cuda:
            from . import ops
            ops.cast_f32_bf16(t, w)           # HIP kernel on the compute stream; the collective below waits for it
        else:
            w.copy_(t)                        # (gloo CPU tests)
        return w

    def null_collectives(self, on=True):
        """Measurement mode (bench.py's exposed-communication figure): every collective call of the exchange -- bucket reduce(-scatter),
        norm all-reduce, weight gather, fp32 replication -- is SKIPPED while the local work around it (wire casts, sharded clip + AdamW,
        graph segments, host calls) runs unchanged: the step time that remains is this rank's compute-only time at the same per-GPU
        batch and launch structure.  The numbers computed in this mode are meaningless (stale slices); switch it off again and
        re-broadcast before training on."""
        self._null = bool(on)

    def _launch(self, lo, hi):
        if self.single:
            return
        src = self._cast_to_wire(lo, hi) if self.wire is not None else self.flat[lo:hi]
        if self._null:
            return
        if not self.sharded:
            self.pending.append((dist.all_reduce(src, group=self.group, async_op=True), None))
            return
        n = (hi - lo) // self.world
        out = self.gshard[lo // self.world:lo // self.world + n]
        if self.emulate:        # all-reduce the bucket, keep the owned slice
            mine = src[self.rank * n:(self.rank + 1) * n]
            self.pending.append((dist.all_reduce(src, group=self.group, async_op=True), lambda: out.copy_(mine)))
        else:
            self.pending.append((dist.reduce_scatter_tensor(out, src, group=self.group, async_op=True), None))

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
        if self.single or what in self.launched:
            return False
        return what in self.layer_bucket or what in self.ranges or what in ("vision", "embed")

    def wait(self):
        for work, post in self.pending:
            work.wait()
            if post is not None:
                post()
        self.pending = []
        # every range must have gone out this step: the optimizer reads the wire / compact image, and a slice that was not launched
        # would silently hold the previous step's gradient (a backward() without the on_done hook on a data-parallel engine)
        keys = {k for k, _, _ in self.buckets}
        self._complete = self.single or keys <= self.launched
        self._missing = sorted(map(str, keys - self.launched))
        self.launched = set()

    def invalidate(self):
        """A backward ran without the exchange hooks (a gradient-accumulation micro-step): the images are stale until the next
        hooked backward + wait()."""
        if not self.single:
            self._complete = False
            self._missing = ["all (the last backward ran without the exchange hook)"]

    def _check_complete(self):
        if not self._complete:
            raise RuntimeError("data-parallel exchange incomplete: gradient ranges %s were not reduced in the last step (backward() must "
                               "run with on_layer_done=buckets.on_done, then buckets.wait(), before optimizer_step)" % getattr(self, "_missing", "?"))

    @property
    def reduced(self):
        """allreduce mode: the tensor holding the reduced (SUM over ranks) gradient after wait() -- the bf16 wire image, or the flat
        fp32 buffer (always the flat buffer at world size 1, where nothing is exchanged)."""
        if self.sharded:
            raise RuntimeError("sharded exchange: the reduced gradient exists as this rank's slices only (grad_shard / owned_rows)")
        if self.single:
            return self.flat
        self._check_complete()
        return self.wire if self.wire is not None else self.flat

    @property
    def grad_shard(self):
        """sharded mode: the compact image of this rank's reduced slices (offsets: owned_rows()[i][1])."""
        self._check_complete()
        return self.gshard

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # ------------------------------------------------------------------------------------------------------------------
    # sharded mode: norm exchange, weight gather
    # ------------------------------------------------------------------------------------------------------------------
    def all_reduce_scalar(self, t):
        """Sum of the ranks' partial squared gradient norms (every rank receives the same bits)."""
        if not self.single and not self._null:
            dist.all_reduce(t, group=self.group)

    def _all_gather(self, out, inp, async_op=True):
        """out[world * n] <- the ranks' inp[n] in rank order -> (work, post)."""
        n = inp.numel()
        if self.emulate:        # zero everything but the own slice, sum over ranks
            out.zero_()
            out[self.rank * n:(self.rank + 1) * n].copy_(inp)
            w = dist.all_reduce(out, group=self.group, async_op=async_op)
        else:
            w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)
        return w

    def set_replicated_fp32(self, ranges):
        """ranges: [(lo, hi)] element ranges of the flat fp32 master that the compute path reads AS FP32 on every rank (Linear biases,
        LayerNorm gamma / beta, the mask embedding -- engine.PretrainEngine._fp32_read_ranges; ~0.15 % of the buffer).  The sharded
        AdamW updates the master on the owner only and the weight gather distributes the 16-bit copy, so these would stay frozen at
        their start-up values on the non-owners (replicas computing with different biases).  gather_params therefore replicates them:
        each rank packs the parts it owns into a zeroed compact image, one fp32 all-reduce (SUM; exactly one rank contributes a
        non-zero term per element, so the sum is that rank's value bit for bit) distributes them, and the first wait_params of the
        next forward unpacks the image into the master -- in front of the first kernel that reads a bias."""
        self._repl = None
        if not self.sharded or self.single:
            return
        own, full, off = [], [], 0
        for lo, hi in sorted((int(a), int(b)) for a, b in ranges if b > a):
            full.append((off, lo, hi - lo))                       # unpack: compact -> master
            for key, _, _ in self.buckets:
                p0, _, n = self.piece(key)
                a, b = max(lo, p0), min(hi, p0 + n)
                if b > a:
                    own.append((a, off + (a - lo), b - a))        # pack: master -> compact
            off += (hi - lo + 3) // 4 * 4
        self._repl = dict(numel=off, own=own, full=full, buf=None, pack=None, unpack=None)

    def _copy_rows(self, which, src, dst):
        r = self._repl
        if src.is_cuda:
            from . import ops
            if r[which] is None:
                r[which] = ops.CopyRanges(r["own" if which == "pack" else "full"], src.device)
            r[which].run(src, dst)
        else:                                                     # (gloo CPU tests)
            for a, b, n in r["own" if which == "pack" else "full"]:
                dst[b:b + n].copy_(src[a:a + n])

    def _exchange_replicated(self, master):
        r = self._repl
        if r["buf"] is None:
            r["buf"] = torch.zeros(max(r["numel"], 4), dtype=master.dtype, device=master.device)
        buf = r["buf"]
        buf.zero_()
        self._copy_rows("pack", master, buf)
        work = dist.all_reduce(buf, group=self.group, async_op=True)
        self._gathers.append((-1, work, lambda: self._copy_rows("unpack", buf, master)))

    def gather_params(self, w16, wshard, master=None, vision_master=None):
        """After the sharded AdamW: distribute the updated bf16 working copy (compact image `wshard` -> flat `w16`), one async
        all-gather per bucket in FORWARD order; wait_params(key) blocks the compute stream right before the first use.  The vision
        stages' convolution weights are folded from the fp32 master (vision.py), so for those buckets the fp32 master slices travel
        instead (`vision_master`, default: when `master` is given and there are vision buckets -- the older calling form).  `master` +
        set_replicated_fp32(): the fp32-read tensors are replicated first (see there)."""
        if self.single or self._null:
            return
        if vision_master is None:
            vision_master = master is not None
        if getattr(self, "_repl", None) is not None:
            if master is None:
                raise RuntimeError("gather_params: the replicated fp32 tensors (set_replicated_fp32) need master=")
            self._exchange_replicated(master)
        if not vision_master:
            master = None
        for i, key in enumerate(self.gather_order):
            lo, hi = self._range(key)
            p0, c0, n = self.piece(key)
            if key in self.vision_keys and master is not None:
                if self._stage32 is None:      # compact fp32 staging of the owned vision slices (e2e configuration only)
                    self._stage32 = {k: torch.empty(self.piece(k)[2], dtype=master.dtype, device=master.device) for k in self.vision_keys}
                st = self._stage32[key]
                st.copy_(master[p0:p0 + n])
                work = self._all_gather(master[lo:hi], st)
            else:
                work = self._all_gather(w16[lo:hi], wshard[c0:c0 + n])
            self._gathers.append((i, work, None))

    def gather_master(self, master):
        """Blocking: every rank's authoritative fp32 master slices into the full flat `master` (checkpoints; a collective)."""
        if self.single or not self.sharded:
            return
        self.wait_params("all")
        stage = torch.empty(max(self.piece(k)[2] for k, _, _ in self.buckets), dtype=master.dtype, device=master.device)
        for key, lo, hi in self.buckets:
            p0, _, n = self.piece(key)
            stage[:n].copy_(master[p0:p0 + n])
            self._all_gather(master[lo:hi], stage[:n], async_op=False)

    def wait_params(self, key):
        """The compute stream waits for the weight gathers up to and including bucket `key` in forward order: "vision" (every vision
        stage), "front" (+ the embedding side), an encoder layer index, "heads" / "all"."""
        if not self.sharded or not self._gathers:
            return
        if key == "all" or key == "heads":
            upto = len(self.gather_order)
        elif key == "vision":
            upto = len(self.vision_keys) - 1
        elif key == "front":
            upto = self._gather_idx["embed"]
        else:
            upto = self._gather_idx[self.layer_key[key]]
        while self._gathers and self._gathers[0][0] <= upto:
            _, work, post = self._gathers.pop(0)
            work.wait()
            if post is not None:
                post()


class DistributedDataParallel(torch.nn.Module):
    """torch.nn.parallel.DistributedDataParallel for the module mirrors (vqa / vcr `ResNetVLBERT`, `VisualLinguisticBert*`): what the
    reference's fine-tuning trainers wrap their model in (vqa/function/train.py:327, vcr/function/train.py:330).  After every backward
    the gradients of all parameters are averaged over the ranks.

    How, MI355X-first: the VL-BERT core's parameters live in ONE flat buffer whose backward is hand-scheduled (engine.backward), so its
    gradient goes out in the same contiguous buckets as the pre-training engine's -- `GradBuckets` launched from the engine's layer
    hooks, overlapped with the rest of the backward (fp32 all-reduce in place on the .grad storage).  The remaining parameters (region
    feature projection, classifier, the ResNet path of the VCR model) are coalesced into one staging buffer when the backward pass
    ends (autograd engine callback), all-reduced once and copied back.  The 1/world average is one HIP pass (vlb_scale_f32).
    `no_sync()` skips the exchange for gradient-accumulation micro-steps, as torch's does."""

    def __init__(self, module, device_ids=None, output_device=None, process_group=None, bucket_bytes=64 << 20):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.require_backward_grad_sync = True
        self._cores = [m for m in module.modules() if hasattr(m, "flat") and hasattr(m, "_engine_for")]
        self._buckets = []
        self._flat_ids = set()
        for core in self._cores:
            core._prepare_grads()
            fl = core.flat
            self._buckets.append(GradBuckets(fl.grad, fl.offsets, fl.numel, core.cfg.num_hidden_layers, group=process_group,
                                             bucket_bytes=bucket_bytes, wire_dtype=None, mode="allreduce"))
            for p in core._pnames.values():
                self._flat_ids.add(id(p))
        self._rest = [p for p in module.parameters() if id(p) not in self._flat_ids and p.requires_grad]
        self._stage = None
        self._armed = False
        # rank 0's parameters and buffers everywhere (DDP's start-up broadcast)
        if self.world > 1:
            with torch.no_grad():
                for core in self._cores:
                    dist.broadcast(core.flat.master, src=0, group=process_group)
                    torch.autograd.graph.increment_version(core.flat.master)
                for t in list(self._rest) + [b for b in module.buffers() if b.is_floating_point()]:
                    if t.is_contiguous():
                        dist.broadcast(t.data, src=0, group=process_group)
                    else:                 # a broadcast into a temporary copy would be discarded: copy the received values back
                        tmp = t.data.contiguous()
                        dist.broadcast(tmp, src=0, group=process_group)
                        t.data.copy_(tmp)

    def no_sync(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self.require_backward_grad_sync = self.require_backward_grad_sync, False
            try:
                yield
            finally:
                self.require_backward_grad_sync = old
        return ctx()

    def _hook(self, k):
        b = self._buckets[k]

        def on_done(what):
            if not self._armed:          # first hook of this backward: the finalizer runs when the autograd pass ends
                self._armed = True
                torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            b.on_done(what)
        on_done.__self__ = b             # engine.backward asks the hook's owner for its will_launch predicate
        return on_done

    def forward(self, *inputs, **kwargs):
        sync = self.world > 1 and self.require_backward_grad_sync and torch.is_grad_enabled()
        for k, core in enumerate(self._cores):
            hook = self._hook(k) if sync else None
            core._dp_hook = hook
            for eng in core._engines.values():
                eng._dp_hook = hook
        out = self.module(*inputs, **kwargs)
        if sync:
            # the finalizer must run after EVERY synchronised backward, also one in which no core hook fires (a loss that does not
            # reach a VL-BERT core): any output that carries a graph arms it when its gradient arrives
            for t in self._tensors(out):
                if t.requires_grad:
                    t.register_hook(self._arm)
        return out

    @staticmethod
    def _tensors(x):
        if torch.is_tensor(x):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from DistributedDataParallel._tensors(v)
        elif isinstance(x, (list, tuple)):
            for v in x:
                yield from DistributedDataParallel._tensors(v)

    def _arm(self, grad):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        return None

    def _finalize(self):
        from . import ops
        self._armed = False
        for b in self._buckets:
            for key, _, _ in b.buckets:          # ranges the backward did not reach (a head that was not used) still take part
                b.on_done(key)
            b.wait()
            ops.scale_f32(b.flat, 1.0 / self.world)
        self._reduce_rest()

    def _reduce_rest(self):
        """The parameters outside the flat cores, one coalesced all-reduce.  The staging layout covers EVERY such parameter, whether or
        not this rank's backward reached it (a parameter used on one rank only -- a conditional branch, an unused head -- would
        otherwise give the ranks staging buffers of different sizes: a hang or a silent mix-up); missing gradients take part as zeros
        and a parameter receives a gradient as soon as ANY rank produced one (its flag is reduced with the data)."""
        from . import ops
        rest = self._rest
        if not rest or self.world == 1:
            return
        n = sum(p.numel() for p in rest)
        if self._stage is None or self._stage.numel() != n + len(rest):
            self._stage = torch.zeros(n + len(rest), dtype=torch.float32, device=rest[0].device)
        st = self._stage
        st.zero_()
        views, off = [], 0
        for p in rest:
            views.append(st[off:off + p.numel()].view(p.shape))
            off += p.numel()
        have = [i for i, p in enumerate(rest) if p.grad is not None]
        if have:
            torch._foreach_copy_([views[i] for i in have], [rest[i].grad for i in have])
            st[n:][torch.tensor(have, device=st.device)] = 1.0
        dist.all_reduce(st, group=self.group)
        if st.is_cuda:
            ops.scale_f32(st[:n], 1.0 / self.world)
        else:
            st[:n].mul_(1.0 / self.world)
        # (the flags are read back -- a host sync -- only when this rank is missing a gradient; otherwise every parameter is copied)
        used = (st[n:] > 0).tolist() if len(have) < len(rest) else [True] * len(rest)
        dst, src = [], []
        for i, p in enumerate(rest):
            if not used[i]:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dst.append(p.grad)
            src.append(views[i])
        if dst:
            torch._foreach_copy_(dst, src)

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.module.load_state_dict(*a, **k)
