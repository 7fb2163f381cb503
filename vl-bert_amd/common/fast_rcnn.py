"""Drop-in `FastRCNN` for the precomputed-feature configuration (common/fast_rcnn.py:15-203 with
`NETWORK.IMAGE_FEAT_PRECOMPUTED: true`, the branch at :136-142,165-187): same constructor arguments, same
`forward(images, boxes, box_mask, im_info, classes=None, segms=None, mvrc_ops=None, mask_visual_embed=None)`
-> {'obj_reps_raw', 'obj_reps'}, same parameter names (`obj_downsample.1.{weight,bias}`).

    boxes [B,R,4+2048] (pad rows -2) -> coordinate embeddings (common/utils/bbox.py:33-65) || 2048-d feature (masked
    regions replaced by `mask_visual_embed`) -> Dropout(0.1) -> Linear(4096 -> final_dim) -> ReLU -> [B,R,final_dim],
    rows of invalid boxes zero (pad_sequence, common/utils/pad_sequence.py:4-17).

Everything arithmetic runs in the HIP library (vlb_obj_prep_fwd, the bf16 GEMM with fused bias+ReLU, the TN weight
gradient, vlb_masked_colsum for the mask-embedding gradient); autograd sees one node.

IMAGE_FEAT_PRECOMPUTED false (the image branch, :144-156): `forward(images [B,3,H,W], boxes [B,R,4], ...)` runs
`vision.VisionStack` (ResNet trunk -> ROIAlign -> dilated layer4 head -> avg-pool) in front of the same node; its backward
continues through the RoI head, ROIAlign and the trainable trunk stages.  Parameters / buffers carry the reference's names
(`backbone.*`, `roi_head_feature_extractor.*`); trainable convolution weights are stored as [O,KH,KW,I] (state_dict /
load_state_dict convert from / to the reference's [O,I,KH,KW]).  `segms` [B,R,14,14] (VCR's object masks) are multiplied into the
RoI-head output before the pool (:152-156).  Not supported on this branch: IMAGE_SEMANTIC (`classes` is accepted and unused, as
in the reference with that option off), `mask_visual_embed`, the bottom-of-CNN cnn_reg_loss, OUTPUT_CONV5.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops

VIS_DIM = 2048
_TAG = 1001     # dropout site tag of the downsample input (engine.TAG_DOWNSAMPLE)


def _get(obj, name, default=None):
    return getattr(obj, name, default) if not isinstance(obj, dict) else obj.get(name, default)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, mask_embed, module, boxes, im_info, sel, idx, train):
        B, R = boxes.shape[0], boxes.shape[1]
        H = weight.shape[0]
        d = boxes.device
        st = module._state(B, R, d)
        module._sync_weights()
        memb = mask_embed.detach().float().reshape(-1) if mask_embed is not None else module._zero_embed
        p = module.drop_p if train else 0.0
        ops.obj_prep_fwd(boxes, im_info, sel, memb, st["a"], drop_p=p, seed=module._seed, tag=_TAG)
        ops.gemm_nt(st["a"], module._w16, st["y"], bias=bias.detach(), act=ops.ACT_RELU)
        ops.gather_rows(st["y"], idx, st["out"])                 # rows of padded boxes -> 0
        ctx.module, ctx.st, ctx.sel, ctx.idx, ctx.p = module, st, sel, idx, p
        ctx.has_embed = mask_embed is not None
        ctx.embed_shape = tuple(mask_embed.shape) if mask_embed is not None else None
        return st["out"].view(B, R, H).float()

    @staticmethod
    def backward(ctx, g):
        module, st = ctx.module, ctx.st
        H = module._w16.shape[0]
        g = g.contiguous().float().view(-1, H)
        ops.relu_bwd_cast(g, st["y"], st["dy_all"])
        ops.gather_rows(st["dy_all"], ctx.idx, st["dy"])         # no gradient from padded boxes
        gw = torch.zeros_like(module._master_w)
        gb = torch.zeros((H,), dtype=torch.float32, device=g.device)
        ops.wgrad_tn(st["dy"], st["a"], gw, colsum=gb, workspace=None)
        g_embed = None
        if ctx.has_embed:     # d(feature half of the GEMM input), summed over the masked regions (through their dropout)
            ops.gemm_nt(st["dy"], module._wT[VIS_DIM:], st["dfeat"])
            ge = torch.zeros((VIS_DIM,), dtype=torch.float32, device=g.device)
            ops.masked_colsum(st["dfeat"], ctx.sel, ge, drop_p=ctx.p, seed=module._seed, tag=_TAG, row_elems=2 * VIS_DIM,
                              col_off=VIS_DIM)
            g_embed = ge.view(ctx.embed_shape)
        if ctx.p > 0:
            ops.rng_advance(module._seed)
        return gw, gb, g_embed, None, None, None, None, None, None


class _FnE2E(torch.autograd.Function):
    """images -> VisionStack -> (coord || feature) -> Dropout -> Linear -> ReLU, one autograd node; conv weights are inputs so that
    autograd delivers their gradients (accumulated by VisionStack.backward into the module's flat-layout buffers)."""

    @staticmethod
    def forward(ctx, weight, bias, module, vs, images, boxes_full, im_info, idx, train, segms, *conv_weights):
        B, R = boxes_full.shape[0], boxes_full.shape[1]
        H = weight.shape[0]
        st = module._state(B, R, boxes_full.device)
        module._sync_weights()
        module._sync_vision(vs)
        vs.forward(images, boxes_full, segms)                    # fills boxes_full[:, :, 4:] with post_roialign
        p = module.drop_p if train else 0.0
        ops.obj_prep_fwd(boxes_full, im_info, None, module._zero_embed, st["a"], drop_p=p, seed=module._seed, tag=_TAG)
        ops.gemm_nt(st["a"], module._w16, st["y"], bias=bias.detach(), act=ops.ACT_RELU)
        ops.gather_rows(st["y"], idx, st["out"])
        ctx.module, ctx.vs, ctx.st, ctx.idx, ctx.p, ctx.boxes = module, vs, st, idx, p, boxes_full
        return st["out"].view(B, R, H).float()

    @staticmethod
    def backward(ctx, g):
        module, vs, st = ctx.module, ctx.vs, ctx.st
        H = module._w16.shape[0]
        g = g.contiguous().float().view(-1, H)
        ops.relu_bwd_cast(g, st["y"], st["dy_all"])
        ops.gather_rows(st["dy_all"], ctx.idx, st["dy"])
        gw = torch.zeros_like(module._master_w)
        gb = torch.zeros((H,), dtype=torch.float32, device=g.device)
        ops.wgrad_tn(st["dy"], st["a"], gw, colsum=gb, workspace=None)
        ops.gemm_nt(st["dy"], module._wT[VIS_DIM:], st["dfeat"])      # d(feature half of the GEMM input)
        for t in module._conv_grads.values():
            t.zero_()
        vs.backward(st["dfeat"], ctx.boxes, drop_p=ctx.p, seed=module._seed, tag=_TAG)
        if ctx.p > 0:
            ops.rng_advance(module._seed)
        return (gw, gb, None, None, None, None, None, None, None, None) + tuple(t.clone() for t in module._conv_grads.values())


class FastRCNN(nn.Module):
    def __init__(self, config, average_pool=True, final_dim=768, enable_cnn_reg_loss=False, device=None):
        super().__init__()
        net = _get(config, "NETWORK")
        self.e2e = not _get(net, "IMAGE_FEAT_PRECOMPUTED", False)
        if self.e2e and not (average_pool and _get(net, "IMAGE_FROZEN_BN", True) and _get(net, "IMAGE_STRIDE_IN_1x1", True)
                             and _get(net, "IMAGE_C5_DILATED", True) and not _get(net, "OUTPUT_CONV5", False)):
            raise NotImplementedError("image branch: needs average_pool, IMAGE_FROZEN_BN, IMAGE_STRIDE_IN_1x1, IMAGE_C5_DILATED, no OUTPUT_CONV5")
        if enable_cnn_reg_loss or _get(net, "IMAGE_SEMANTIC", False):
            raise NotImplementedError("cnn_reg_loss / IMAGE_SEMANTIC object-class embeddings are not supported")
        if not torch.cuda.is_available():
            raise RuntimeError("FastRCNN (HIP) needs an MI355X: there is no CPU fallback")
        self.final_dim = final_dim
        self.drop_p = 0.1                                      # hard-coded in the reference (common/fast_rcnn.py:106)
        dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        down = nn.Module()
        down.register_parameter("weight", nn.Parameter(torch.empty((final_dim, 2 * VIS_DIM), device=dev)))
        down.register_parameter("bias", nn.Parameter(torch.zeros((final_dim,), device=dev)))
        self.obj_downsample = nn.Module()
        self.obj_downsample.add_module("1", down)              # Sequential(Dropout, Linear, ReLU): the Linear is entry "1"
        self._w16 = torch.zeros((final_dim, 2 * VIS_DIM), dtype=ops.BF16, device=dev)
        self._wT = torch.zeros((2 * VIS_DIM, final_dim), dtype=ops.BF16, device=dev)
        self._zero_embed = torch.zeros((VIS_DIM,), dtype=torch.float32, device=dev)
        self._seed = torch.tensor([ops.rank_seed(20011)], dtype=torch.int32, device=dev)
        self._version, self._states = None, OrderedDict()
        self._stacks, self._conv_params, self._conv_grads, self._vbuffers, self._vversion = OrderedDict(), {}, {}, {}, 0
        if self.e2e:
            from .. import vision as _vision
            self._vision = _vision
            self._nl = int(_get(net, "IMAGE_NUM_LAYERS", 101))
            self._frozen_stages = tuple(_get(net, "IMAGE_FROZEN_BACKBONE_STAGES", (1, 2)))
            P = len(_vision.PREFIX)
            for key, O, I, k, bn, tr in _vision.conv_table(self._nl, self._frozen_stages):
                if tr:
                    w = nn.Parameter(torch.empty((O, k, k, I), device=dev).normal_(0.0, (2.0 / (O * k * k)) ** 0.5))
                    self._set(key + ".weight", w, param=True)
                    self._conv_params[_vision.PREFIX + key + ".weight"] = w
                    self._conv_grads[_vision.PREFIX + key + ".weight"] = torch.zeros((O, k, k, I), device=dev)
                else:
                    t = torch.empty((O, I, k, k), device=dev).normal_(0.0, (2.0 / (O * k * k)) ** 0.5)
                    self._set(key + ".weight", t, param=False)
                    self._vbuffers[_vision.PREFIX + key + ".weight"] = t
                for suffix, fill in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
                    t = torch.full((O,), fill, device=dev)
                    self._set(bn + "." + suffix, t, param=False)
                    self._vbuffers[_vision.PREFIX + bn + "." + suffix] = t
            assert P
        self.init_weight()

    def _set(self, dotted, t, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        if param:
            mod.register_parameter(parts[-1], t)
        else:
            mod.register_buffer(parts[-1], t)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
        for name in self._conv_params:            # engine layout [O,KH,KW,I] -> reference layout [O,I,KH,KW]
            k = prefix + name[len("image_feature_extractor."):]
            sd[k] = sd[k].permute(0, 3, 1, 2).contiguous()
        if self.e2e:      # the reference registers the RoI head twice: `head` = Sequential(roi_head_feature_extractor, pool, flatten)
            alias = prefix + "roi_head_feature_extractor."      # (common/fast_rcnn.py:80-84), so its checkpoints carry `head.0.*` too
            for k in [k for k in sd if k.startswith(alias)]:
                sd[prefix + "head.0." + k[len(alias):]] = sd[k]
        return sd

    def load_state_dict(self, state_dict, strict=True):
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked") and not k.startswith("head.0.")}
        for name in self._conv_params:
            k = name[len("image_feature_extractor."):]
            if k in state_dict:
                state_dict[k] = state_dict[k].permute(0, 2, 3, 1).contiguous()
        self._vversion += 1
        return super().load_state_dict(state_dict, strict=strict)

    def _stack(self, B, R, Hi, Wi, dev):
        # the collators pad the images of a batch to that batch's largest one (clip_pad_images), so (Hi, Wi) varies: a stack owns every
        # activation of the trunk for one geometry (6.4 GB at 8 x 600 x 1000) -- only the most recently used ones are kept
        from .visual_linguistic_bert import lru_get
        storage = lambda name, shape: (self._conv_params[name].data, self._conv_grads[name])
        return lru_get(self._stacks, (B, R, Hi, Wi),
                       lambda: self._vision.VisionStack(B, Hi, Wi, R, device=dev, num_layers=self._nl, frozen_stages=self._frozen_stages,
                                                        storage=storage))

    def _sync_vision(self, vs):
        """BatchNorm tensors / frozen weights after construction or load_state_dict; folded bf16 operands after any weight update."""
        if getattr(vs, "_mirror_buffers", None) != self._vversion:
            vs.load_state_dict(self._vbuffers, strict=False)
            vs._mirror_buffers = self._vversion
        ver = tuple(p._version for p in self._conv_params.values())
        if vs._dirty or getattr(vs, "_mirror_weights", None) != ver:
            vs.refresh_weights()
            vs._mirror_weights = ver

    @property
    def _master_w(self):
        return getattr(self.obj_downsample, "1").weight

    def init_weight(self):
        """common/fast_rcnn.py:111-118: normal(0, 0.01) weight, zero bias."""
        lin = getattr(self.obj_downsample, "1")
        with torch.no_grad():
            lin.weight.normal_(0.0, 0.01)
            lin.bias.zero_()

    def bn_eval(self):
        pass                                                   # no BatchNorm on the precomputed branch

    def _sync_weights(self):
        w = self._master_w
        if self._version != w._version:
            ops.cast_f32_bf16(w.detach().contiguous(), self._w16)
            ops.transpose(self._w16, self._wT)
            self._version = w._version

    def _state(self, B, R, dev):
        def make():
            n, H = B * R, self.final_dim
            zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=dev)
            return dict(a=zb(n, 2 * VIS_DIM), y=zb(n, H), out=zb(n, H), dy_all=zb(n, H), dy=zb(n, H), dfeat=zb(n, VIS_DIM))
        from .visual_linguistic_bert import lru_get       # (R follows each batch's largest box count: keep the recent shapes only)
        return lru_get(self._states, (B, R), make)

    def _forward_e2e(self, images, boxes, box_mask, im_info, classes, segms, mvrc_ops, mask_visual_embed):
        # `classes` only feeds the IMAGE_SEMANTIC object-class embedding and the bottom-of-CNN regulariser (common/fast_rcnn.py:139,
        # 158-163), both rejected by the constructor: accepted and unused here, as in the reference with those options off
        if mask_visual_embed is not None:
            raise NotImplementedError("image branch: mask_visual_embed is not supported")
        B, R = boxes.shape[0], boxes.shape[1]
        dev = boxes.device
        full = torch.zeros((B, R, 4 + VIS_DIM), dtype=torch.float32, device=dev)
        full[:, :, :4] = boxes[:, :, :4].float()
        full[:, :, 0][~box_mask.bool()] = -2.0                  # the padding marker the kernels test (collate_batch.py:39)
        vs = self._stack(B, R, int(images.shape[2]), int(images.shape[3]), dev)
        ar = torch.arange(B * R, dtype=torch.int32, device=dev)
        idx = torch.where(box_mask.reshape(-1).bool(), ar, torch.full_like(ar, -1))
        lin = getattr(self.obj_downsample, "1")
        obj_reps = _FnE2E.apply(lin.weight, lin.bias, self, vs, images.float().contiguous(), full, im_info.float().contiguous(), idx,
                                self.training, segms, *self._conv_params.values())
        return {"obj_reps_raw": full[:, :, 4:].detach().clone(), "obj_reps": obj_reps}

    def forward(self, images, boxes, box_mask, im_info, classes=None, segms=None, mvrc_ops=None, mask_visual_embed=None):
        if (images is not None) != self.e2e:
            raise NotImplementedError("IMAGE_FEAT_PRECOMPUTED configuration takes images=None, the image branch an image batch")
        if self.e2e:
            return self._forward_e2e(images, boxes, box_mask, im_info, classes, segms, mvrc_ops, mask_visual_embed)
        B, R = boxes.shape[0], boxes.shape[1]
        boxes = boxes.contiguous().float()
        if boxes.shape[2] != 4 + VIS_DIM:
            raise ValueError("precomputed boxes must be [B, R, 4 + 2048]")
        use_mask = mvrc_ops is not None and mask_visual_embed is not None
        sel = (mvrc_ops.reshape(-1).to(torch.int64) if use_mask else torch.zeros((B * R,), dtype=torch.int64, device=boxes.device))
        ar = torch.arange(B * R, dtype=torch.int32, device=boxes.device)
        idx = torch.where(box_mask.reshape(-1).bool(), ar, torch.full_like(ar, -1))     # index plumbing (the reference: nonzero())
        lin = getattr(self.obj_downsample, "1")
        obj_reps = _Fn.apply(lin.weight, lin.bias, mask_visual_embed if use_mask else None, self, boxes, im_info.float().contiguous(),
                             sel, idx, self.training)
        raw = boxes[:, :, 4:].clone()
        if use_mask:                                           # data movement only (common/fast_rcnn.py:170-172)
            raw[mvrc_ops.bool() & box_mask.bool()] = mask_visual_embed.detach().reshape(-1).to(raw.dtype)
        raw[~box_mask.bool()] = 0
        return {"obj_reps_raw": raw, "obj_reps": obj_reps}
