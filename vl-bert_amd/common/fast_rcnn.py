"""Drop-in `FastRCNN` for the precomputed-feature configuration (common/fast_rcnn.py:15-203 with
`NETWORK.IMAGE_FEAT_PRECOMPUTED: true`, the branch at :136-142,165-187): same constructor arguments, same
`forward(images, boxes, box_mask, im_info, classes=None, segms=None, mvrc_ops=None, mask_visual_embed=None)`
-> {'obj_reps_raw', 'obj_reps'}, same parameter names (`obj_downsample.1.{weight,bias}`).

    boxes [B,R,4+2048] (pad rows -2) -> coordinate embeddings (common/utils/bbox.py:33-65) || 2048-d feature (masked
    regions replaced by `mask_visual_embed`) -> Dropout(0.1) -> Linear(4096 -> final_dim) -> ReLU -> [B,R,final_dim],
    rows of invalid boxes zero (pad_sequence, common/utils/pad_sequence.py:4-17).

Everything arithmetic runs in the HIP library (vlb_obj_prep_fwd, the bf16 GEMM with fused bias+ReLU, the TN weight
gradient, vlb_masked_colsum for the mask-embedding gradient); autograd sees one node.  The ResNet-101 / RoIAlign image
branch (IMAGE_FEAT_PRECOMPUTED false) is not built: constructing it raises NotImplementedError.
"""
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


class FastRCNN(nn.Module):
    def __init__(self, config, average_pool=True, final_dim=768, enable_cnn_reg_loss=False, device=None):
        super().__init__()
        net = _get(config, "NETWORK")
        if not _get(net, "IMAGE_FEAT_PRECOMPUTED", False):
            raise NotImplementedError("ResNet-101 / RoIAlign image branch is not built (SURVEY.md §8f rank 3); "
                                      "set NETWORK.IMAGE_FEAT_PRECOMPUTED")
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
        self._w16 = torch.zeros((final_dim, 2 * VIS_DIM), dtype=torch.bfloat16, device=dev)
        self._wT = torch.zeros((2 * VIS_DIM, final_dim), dtype=torch.bfloat16, device=dev)
        self._zero_embed = torch.zeros((VIS_DIM,), dtype=torch.float32, device=dev)
        self._seed = torch.tensor([20011], dtype=torch.int32, device=dev)
        self._version, self._states = None, {}
        self.init_weight()

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
        key = (B, R)
        if key not in self._states:
            n, H = B * R, self.final_dim
            zb = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=dev)
            self._states[key] = dict(a=zb(n, 2 * VIS_DIM), y=zb(n, H), out=zb(n, H), dy_all=zb(n, H), dy=zb(n, H),
                                     dfeat=zb(n, VIS_DIM))
        return self._states[key]

    def forward(self, images, boxes, box_mask, im_info, classes=None, segms=None, mvrc_ops=None, mask_visual_embed=None):
        if images is not None:
            raise NotImplementedError("precomputed-feature configuration: pass images=None")
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
