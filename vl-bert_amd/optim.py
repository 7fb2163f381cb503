"""Optimisers of the reference's fine-tuning entry points over the HIP library.

`FusedSGD` = torch.optim.SGD(lr, momentum, weight_decay) as the VCR trainer builds it (vcr/function/train.py:124-128; dampening 0, no
Nesterov): same constructor arguments, same `param_groups` / `state_dict` layout (`momentum_buffer` per parameter), so the reference's
LR schedulers and checkpoints work unchanged; `step()` runs vlb_sgd_momentum_step (optim.hip: weight decay, momentum and the update
in ONE pass over each parameter, no temporaries).  Parameters that are consecutive slices of one flat allocation (the `FlatParams`
storage behind the VisualLinguisticBert mirrors) are updated by a single launch over the whole range.

`FusedAdamW` = the reference's own `AdamW` (common/nlp/bert/optimization.py:107-187, the optimiser of the pre-training and VQA entry
points: Adam with bias correction, weight decay applied to the weights AFTER the Adam update with the un-corrected lr), same constructor
arguments and per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`); `step()` runs vlb_adamw_step per flat run.
"""
import torch

from . import ops


_MAX_PAD = 63      # FlatParams starts every tensor on a 64-element boundary: a gap of fewer elements can only be alignment padding


def _gap_is_zero(t, gap):
    """The `gap` elements that follow the contiguous tensor t in its allocation are all zero (checked once, when the runs are built)."""
    pad = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset() + t.numel(), (gap,))
    return int(torch.count_nonzero(pad)) == 0


def _flat_runs(params):
    """[(first index, last index + 1, n elements)] of maximal runs whose parameter AND gradient tensors lie back to back in one
    allocation -- up to the alignment padding of the flat layout (engine.FlatParams: zero master, zero gradient, so an update of the
    padding is a no-op: p = g = 0 keeps m = v = p = 0 under AdamW and SGD alike; a gap that holds anything but zeros -- somebody
    else's data -- is never merged across).  The VL-BERT-large mirror is 11 runs instead of 131."""
    runs, start = [], 0
    for i in range(1, len(params) + 1):
        if i < len(params):
            a, b = params[i - 1], params[i]
            gap = (b.data_ptr() - a.data_ptr()) // 4 - a.numel()
            same = (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                    and a.grad.untyped_storage().data_ptr() == b.grad.untyped_storage().data_ptr()
                    and (b.data_ptr() - a.data_ptr()) % 4 == 0 and 0 <= gap <= _MAX_PAD
                    and b.grad.data_ptr() - a.grad.data_ptr() == b.data_ptr() - a.data_ptr())
            if same and (gap == 0 or (a.is_contiguous() and a.grad.is_contiguous()
                                      and _gap_is_zero(a.detach(), gap) and _gap_is_zero(a.grad, gap))):
                continue
        first, last = params[start], params[i - 1]
        runs.append((start, i, (last.data_ptr() - first.data_ptr()) // 4 + last.numel()))
        start = i
    return runs


_FLAT_KEYS = ("flat_m", "flat_v", "flat_buffer", "dev", "hyper", "hyper_clip")


def clip_grad_norm_(parameters, max_norm, optimizer, grad_scale=None):
    """torch.nn.utils.clip_grad_norm_ as the reference's trainer calls it (common/trainer.py:139-145), fused into the next
    `optimizer.step()`: the squared gradient norm is summed on the device by the HIP kernels (vlb_sumsq_f32 over the flat runs of the
    optimizer's parameters -- `parameters` is accepted for signature parity), kept on the device, and the step's kernel multiplies
    every gradient by min(1, max_norm / (norm + 1e-6)) as it reads it -- no pass over the gradients to rescale them, no host sync.
    grad_scale: an fp16 loss scale to divide out first (norm and clip are computed on the unscaled gradients).  Returns the total norm
    (0-dim device tensor, unscaled).  The gradients themselves are left as they are."""
    if not isinstance(optimizer, (FusedAdamW, FusedSGD)):
        raise TypeError("clip_grad_norm_ fuses into FusedAdamW / FusedSGD")
    if grad_scale is not None:
        optimizer.grad_scale = float(grad_scale)
    total = optimizer._sumsq_all()
    optimizer._clip = (float(max_norm), total)
    return total.sqrt() * optimizer.grad_scale


class _FlatStateMixin:
    """The per-run flat buffers (`flat_*`, `dev`) are an implementation detail: state_dict() exports the reference's per-parameter
    entries only (they are views of the flat buffers), and load_state_dict() drops the flat buffers so that the next step() re-creates
    them FROM the loaded per-parameter tensors and re-binds the views -- otherwise a reloaded checkpoint would leave the kernels
    updating buffers the exported state no longer aliases."""

    grad_scale = 1.0      # multiplies every gradient inside the step kernel (1 / loss scale of an fp16 run; 1 / world of a DP sum)
    _clip = None          # (max_norm, device sum of squares) left by clip_grad_norm_ for the next step()

    def _group_runs(self, gi, group, who):
        ps = [p for p in group["params"] if p.grad is not None]
        _check_fp32_gpu(ps, who)
        # memory order (a module tree yields query.weight, query.bias, key.weight, ... while the flat layout keeps the three weights,
        # then the three biases, adjacent): runs are found among address-sorted parameters
        ps.sort(key=lambda p: (p.untyped_storage().data_ptr(), p.data_ptr()))
        key = (gi, tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps))
        if self._runs.get("key%d" % gi) != key:          # layout changed (first step / re-allocated gradients): re-derive the runs
            self._runs["key%d" % gi] = key
            # The runs are FOUND in address order but VISITED in the order of the group's parameter list: which of two separate
            # allocations lies lower is the allocator's business and differs between the processes of a data-parallel job, and the
            # squared norm is accumulated run by run (fp32, not associative) -- in address order two replicas holding bit-identical
            # gradients derived clip coefficients one ulp apart and drifted (round-4 GPUTEST: dp2_mirror_check `assert same`).
            where = {id(p): i for i, p in enumerate(group["params"])}
            self._runs[gi] = sorted(_flat_runs(ps), key=lambda r: min(where[id(p)] for p in ps[r[0]:r[1]]))
        return ps, self._runs[gi]

    def _sumsq_all(self):
        total = None
        for gi, group in enumerate(self.param_groups):
            ps, runs = self._group_runs(gi, group, type(self).__name__)
            for a, b, n in runs:
                first = ps[a]
                if total is None:
                    total = torch.zeros(1, dtype=torch.float32, device=first.device)
                    if getattr(self, "_sumsq_ws", None) is None or self._sumsq_ws.device != first.device:
                        self._sumsq_ws = torch.zeros(2048, dtype=torch.float32, device=first.device)
                # fixed summation order: data-parallel replicas hold identical gradients and must derive the identical clip coefficient
                ops.sumsq_det(first.grad.as_strided((n,), (1,), first.grad.storage_offset()), self._sumsq_ws, total)
        return total if total is not None else torch.zeros(1)

    def zero_grad(self, set_to_none=False):
        """set_to_none=False (what the flat-storage mirrors need: their .grad tensors are views of one buffer): one fill per flat run
        instead of one per parameter (VL-BERT-large: 3 instead of ~390)."""
        if set_to_none:
            return super().zero_grad(set_to_none=True)
        for gi, group in enumerate(self.param_groups):
            if any(p.grad is not None and not (p.is_cuda and p.grad.dtype == torch.float32) for p in group["params"]):
                return super().zero_grad(set_to_none=False)
            ps, runs = self._group_runs(gi, group, type(self).__name__)
            for a, b, n in runs:
                first = ps[a]
                first.grad.as_strided((n,), (1,), first.grad.storage_offset()).zero_()

    def state_dict(self):
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if n not in _FLAT_KEYS} for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            for n in _FLAT_KEYS:
                st.pop(n, None)
        self._runs = {}


def _check_fp32_gpu(ps, who):
    for p in ps:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and p.grad.dtype == torch.float32):
            raise RuntimeError("%s needs contiguous fp32 GPU parameters and gradients; there is no CPU path" % who)


class FusedAdamW(_FlatStateMixin, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid hyper-parameter")          # the reference's checks (optimization.py:117-124)
        if not correct_bias:
            raise NotImplementedError("FusedAdamW implements correct_bias=True (every shipped configuration)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._runs = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip, self._clip = self._clip, None
        for gi, group in enumerate(self.param_groups):
            ps, runs = self._group_runs(gi, group, "FusedAdamW")
            hyper = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), float(group["weight_decay"]))
            for a, b, n in runs:
                first = ps[a]
                st = self.state[first]
                if "flat_m" not in st or st["flat_m"].numel() != n:
                    st["flat_m"] = torch.zeros(n, dtype=torch.float32, device=first.device)
                    st["flat_v"] = torch.zeros(n, dtype=torch.float32, device=first.device)
                    steps = [self.state[p].get("step", 0) for p in ps[a:b]]
                    if len(set(steps)) != 1:
                        raise RuntimeError("FusedAdamW: parameters of one flat run carry different step counts")
                    # device-resident {lr, beta1, beta2, eps, weight_decay, step, max_norm (0: no clip), sumsq}: one per run
                    st["dev"] = torch.tensor(list(hyper) + [float(steps[0]), 0.0, 0.0], dtype=torch.float32, device=first.device)
                    st["hyper"] = hyper
                    for p in ps[a:b]:                        # the reference's per-parameter state: views of the run's buffers
                        off = (p.data_ptr() - first.data_ptr()) // 4
                        for name, flat in (("exp_avg", st["flat_m"]), ("exp_avg_sq", st["flat_v"])):
                            old = self.state[p].get(name)
                            view = flat[off:off + p.numel()].view_as(p)
                            if old is not None:
                                view.copy_(old.to(view.device, view.dtype).view_as(view))
                            self.state[p][name] = view
                        self.state[p].setdefault("step", 0)
                if st["hyper"] != hyper:                     # an LR scheduler (or the user) changed the group's values
                    st["dev"][0:5].copy_(torch.tensor(hyper, dtype=torch.float32), non_blocking=True)
                    st["hyper"] = hyper
                pf = first.data.as_strided((n,), (1,), first.storage_offset())
                gf = first.grad.as_strided((n,), (1,), first.grad.storage_offset())
                if clip is not None:                         # {max_norm, sumsq} of the device state: the kernel derives the coefficient
                    st["dev"][6:7].fill_(clip[0])
                    st["dev"][7:8].copy_(clip[1])
                elif float(st["hyper_clip"] if "hyper_clip" in st else 0.0) != 0.0:
                    st["dev"][6:7].zero_()
                st["hyper_clip"] = clip[0] if clip is not None else 0.0
                ops.adamw_step(pf, gf, st["flat_m"], st["flat_v"], None, st["dev"], grad_scale=self.grad_scale)      # (advances the run's step count)
                for p in ps[a:b]:
                    self.state[p]["step"] += 1
                    torch.autograd.graph.increment_version(p)
        return loss


class FusedSGD(_FlatStateMixin, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("FusedSGD implements dampening = 0, nesterov = False (the reference's configuration)")
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("negative hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self._runs = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip, self._clip = self._clip, None
        for gi, group in enumerate(self.param_groups):
            ps, runs = self._group_runs(gi, group, "FusedSGD")
            for a, b, n in runs:
                first = ps[a]
                st = self.state[first]
                if "flat_buffer" not in st or st["flat_buffer"].numel() != n:
                    st["flat_buffer"] = torch.zeros(n, dtype=torch.float32, device=first.device)
                    for p in ps[a:b]:                        # torch.optim.SGD's per-parameter state: views of the run's buffer
                        off = (p.data_ptr() - first.data_ptr()) // 4
                        old = self.state[p].get("momentum_buffer")
                        view = st["flat_buffer"][off:off + p.numel()].view_as(p)
                        if old is not None:
                            view.copy_(old.to(view.device, view.dtype).view_as(view))
                        self.state[p]["momentum_buffer"] = view
                pf = first.data.as_strided((n,), (1,), first.storage_offset())
                gf = first.grad.as_strided((n,), (1,), first.grad.storage_offset())
                ops.sgd_momentum_step(pf, gf, st["flat_buffer"], group["lr"], group["momentum"], group["weight_decay"],
                                      sumsq=clip[1] if clip is not None else None, max_norm=clip[0] if clip is not None else 0.0,
                                      grad_scale=self.grad_scale)
                for p in ps[a:b]:      # the kernel wrote through raw pointers: tell autograd (the module mirrors key their bf16 /
                    torch.autograd.graph.increment_version(p)      # transposed weight copies on the parameters' version counters)
        return loss
