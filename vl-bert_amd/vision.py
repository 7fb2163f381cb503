"""End-to-end vision path of the hot path: ResNet trunk -> ROIAlign -> dilated layer4 RoI head
-> average pool, forward and hand-scheduled backward, on the HIP kernels of csrc/gemm.hip + csrc/vision.hip.

Mirrors what `FastRCNN.forward` does when IMAGE_FEAT_PRECOMPUTED is false (common/fast_rcnn.py:144-156):
    img_feats = backbone(images)['body4']            common/backbone/resnet/resnet.py:175-199
    roi_align_res = ROIAlign(14x14, 1/16)(...)       common/lib/roi_pooling/roi_align.py:49-66
    post_roialign = head(roi_align_res)              layer4 (stride 1 / dilation 2) -> AvgPool2d(14) -> Flattener (:74-84)
and writes post_roialign into the feature slots of the padded box rows ([B, R, 4 + 2048] fp32), i.e. the tensor the
precomputed-feature path reads -- coordinate embedding, obj_downsample and everything after are shared with it.

MI355X design (not the reference's NCHW + cuDNN):
  * activations NHWC bf16 = row-major [N*H*W, C] matrices: 1x1 convolutions are plain MFMA NT GEMMs, 3x3 convolutions are
    the same GEMM over an im2col image (tap-major [rows, 9C]); the stride of the caffe-style stride-in-1x1 blocks is a row
    subsample in front of the GEMMs;
  * frozen BatchNorm (IMAGE_FROZEN_BN, eval mode: common/fast_rcnn.py:88-100,122-126) is folded into the bf16 working weights
    (scale) and the GEMM bias (shift); the weight gradient of the fp32 master is scale[o] * dW_folded;
  * ReLU, residual add and ReLU backward ride in GEMM epilogues (act 2 / 7 / 8): a Bottleneck forward is 3-4 GEMMs + 1 im2col,
    its backward 3-4 dgrad GEMMs + 1 im2col + 3-4 TN weight-gradient GEMMs, no elementwise passes;
  * the 3x3 data gradient is the forward gather on dY with mirrored taps (padding = dilation), so there is no col2im scatter;
  * frozen stages (IMAGE_FROZEN_BACKBONE_STAGES = [1, 2]: stem + layer1, resnet.py:201-222) run forward only;
  * everything is statically shaped (all B*R box slots are pooled; padded boxes produce zeros / get no gradient), so the step
    is graph-capturable; 288 GB HBM holds every activation and im2col image (~10 GB at 8 x 600x1000), nothing is recomputed.
Convention for backward: the gradient handed to a block is dL/d(pre-ReLU block output), i.e. already masked by (y > 0) by
whoever produced it (the next block's last dgrad epilogue, the ROIAlign-backward cast, the avg-pool backward).
"""
import os
from collections import OrderedDict

import torch

from . import ops

F32 = torch.float32
MODEL_LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}    # resnet.py model_layers
PREFIX = "image_feature_extractor."
VIS_DIM = 2048
BN_EPS = 1e-5
STEM_K = 192         # 7*7*3 = 147 taps x channels, padded to a multiple of the GEMM's 64-wide K tile


def _ru(x, m):
    return (x + m - 1) // m * m


def block_table(num_layers=101, stride_in_1x1=True, c5_dilated=True):
    """One dict per Bottleneck, in forward order (resnet.py:160-173; RoI head: common/fast_rcnn.py:74-78)."""
    out, inplanes = [], 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), MODEL_LAYERS[num_layers])):
        stride, dil = (1 if li == 0 else 2), 1
        if li == 3 and c5_dilated:
            stride, dil = 1, 2
        for b in range(nb):
            first = b == 0
            key = ("roi_head_feature_extractor.%d." % b) if li == 3 else ("backbone.layer%d.%d." % (li + 1, b))
            out.append(dict(key=key, layer=li + 1, index=b, inplanes=inplanes, planes=planes, stride=stride if first else 1, dil=dil,
                            downsample=first and (stride != 1 or inplanes != planes * 4), stage=li + 2,
                            stride_in_1x1=stride_in_1x1))
            inplanes = planes * 4
    return out


def conv_table(num_layers=101, frozen_stages=(1, 2)):
    """[(state-dict key without '.weight', O, I, k, bn key, trainable)] for every convolution, forward order."""
    convs = [("backbone.conv1", 64, 3, 7, "backbone.bn1", 1 not in frozen_stages)]
    for b in block_table(num_layers):
        tr = b["stage"] not in frozen_stages
        k, P, C = b["key"], b["planes"], b["inplanes"]
        convs.append((k + "conv1", P, C, 1, k + "bn1", tr))
        convs.append((k + "conv2", P, P, 3, k + "bn2", tr))
        convs.append((k + "conv3", 4 * P, P, 1, k + "bn3", tr))
        if b["downsample"]:
            convs.append((k + "downsample.0", 4 * P, C, 1, k + "downsample.1", tr))
    return convs


def vision_param_layout(num_layers=101, frozen_stages=(1, 2)):
    """Trainable tensors {PREFIX + reference key: engine-internal shape [O, KH, KW, I]} (the reference stores [O, I, KH, KW];
    `VisionStack.load_state_dict` / `state_dict` permute).  BatchNorm tensors and frozen stages are not optimizer state."""
    return OrderedDict((PREFIX + key + ".weight", (O, k, k, I)) for key, O, I, k, bn, tr in conv_table(num_layers, frozen_stages) if tr)


class _Conv:
    __slots__ = ("key", "O", "I", "k", "taps", "kf", "trainable", "w32", "g32", "bn", "wf", "wb", "scale", "shift")


class VisionStack:
    def __init__(self, N, Himg, Wimg, R, device="cuda:0", num_layers=101, frozen_stages=(1, 2), c5_dilated=True, storage=None,
                 pooled=14, spatial_scale=1.0 / 16, sampling_ratio=1):
        """N images of Himg x Wimg, R box slots per image.  storage(name, shape) -> (fp32 master view, fp32 grad view) places the
        trainable weights inside the caller's flat optimizer buffers (PretrainEngine); None allocates private ones."""
        if 1 not in frozen_stages:
            raise NotImplementedError("the stem (stage 1) is frozen in every shipped configuration; its backward is not built")
        self.N, self.Himg, self.Wimg, self.R, self.K = N, Himg, Wimg, R, N * R
        self.dev = torch.device(device)
        self.num_layers, self.frozen_stages = num_layers, tuple(frozen_stages)
        self.pooled, self.scale, self.sr = pooled, spatial_scale, sampling_ratio
        d = self.dev
        zb = lambda *s: torch.zeros(s, dtype=ops.BF16, device=d)
        zf = lambda *s: torch.zeros(s, dtype=F32, device=d)
        self.blocks = block_table(num_layers, True, c5_dilated)
        # 3x3 convolutions: implicit GEMM (gather in the GEMM's LDS-DMA address generator) for forward and data gradient; the
        # im2col image is only materialised in backward, for the TN weight gradient.  VLB_CONV_IMPLICIT=0: explicit im2col + GEMM.
        self.implicit = os.environ.get("VLB_CONV_IMPLICIT", "1") != "0"
        self.zero16 = torch.zeros(64, dtype=ops.BF16, device=d)
        # Weight gradients run on a second stream (they only feed the optimizer; the dgrad chain is the critical path and these
        # GEMMs are too small to fill 256 CUs one at a time).  Hazards are tracked per buffer: a wgrad starts after the event
        # recorded behind its producers, and whoever overwrites one of its inputs first waits for the event recorded behind it
        # (da / db are double-buffered by block parity so that wait is two blocks old).  VLB_VISION_WGRAD_STREAM=0 serialises.
        self.side = torch.cuda.Stream(device=d) if (d.type == "cuda" and os.environ.get("VLB_VISION_WGRAD_STREAM", "1") != "0") else None
        # Round 4: the three / four weight gradients of a Bottleneck are independent of each other, and each of them is a 100-300-tile
        # launch (+ its slab reduce) that leaves CUs idle: they rotate over VLB_VISION_WGRAD_STREAMS side streams (default 3), each with
        # its own split-K slab workspace, so that they overlap each other as well as the data-gradient chain.
        n_side = max(1, int(os.environ.get("VLB_VISION_WGRAD_STREAMS", "3"))) if self.side is not None else 0
        self.sides = [self.side] + [torch.cuda.Stream(device=d) for _ in range(n_side - 1)] if self.side is not None else []
        self._side_rr = 0
        self._pending = {}
        min_train = min(b["stage"] for b in self.blocks if b["stage"] not in self.frozen_stages)
        if any(b["stage"] in self.frozen_stages and b["stage"] > min_train for b in self.blocks):
            raise NotImplementedError("frozen stages must be a prefix of the network")
        # ---- parameters -------------------------------------------------------------------------------------------
        self.convs = OrderedDict()
        self.frozen = OrderedDict()              # fp32: BatchNorm tensors of every conv + weights of the frozen stages ([O, taps, I])
        self._own = OrderedDict()
        need_dgrad = self._dgrad_set()
        for key, O, I, k, bn, tr in conv_table(num_layers, frozen_stages):
            c = _Conv()
            c.key, c.O, c.I, c.k, c.taps, c.trainable = key, O, I, k, k * k, tr
            c.kf = STEM_K if key == "backbone.conv1" else k * k * I
            name = PREFIX + key + ".weight"
            if tr:
                if storage is not None:
                    w, g = storage(name, (O, k, k, I))
                else:
                    w, g = zf(O, k, k, I), zf(O, k, k, I)
                    self._own[name] = (w, g)
                c.w32, c.g32 = w.view(O, k * k, I), g.view(O, k * k * I)
            else:
                c.w32, c.g32 = zf(O, k * k, I), None
                self.frozen[name] = c.w32
            c.bn = tuple(zf(O) for _ in range(4))
            c.bn[0].fill_(1.0)
            c.bn[3].fill_(1.0)
            for t, suffix in zip(c.bn, ("weight", "bias", "running_mean", "running_var")):
                self.frozen[PREFIX + bn + "." + suffix] = t
            c.wf = zb(O, c.kf)
            c.wb = zb(I, k * k * O) if key in need_dgrad else None
            c.scale, c.shift = zf(O), zf(O)
            self.convs[key] = c
        self._dirty = True
        self._prep = {}
        # ---- geometry + activations -------------------------------------------------------------------------------
        cs = ops.conv_out_size
        self.H1, self.W1 = cs(Himg, 7, 2, 3, 1), cs(Wimg, 7, 2, 3, 1)
        self.Hp, self.Wp = cs(self.H1, 3, 2, 1, 1), cs(self.W1, 3, 2, 1, 1)
        self.col1 = zb(N * self.H1 * self.W1, STEM_K)
        self.c1 = zb(N * self.H1 * self.W1, 64)
        self.p1 = zb(N * self.Hp * self.Wp, 64)
        h, w, n = self.Hp, self.Wp, N
        self.groups = OrderedDict()              # per layer: geometry + backward scratch
        shared_col = {}
        # Round 4: the 1x1-convolution weight gradients of layer3 (22 identical Bottlenecks + the first: 46-47 products of 256 x 1024 /
        # 1024 x 256 outputs over 19 152 rows, each a 100-300-tile launch of the 128 x 128 TN kernel + a slab reduce) are DEFERRED: every
        # block keeps its dz / da gradient tensors, and when the stage's data-gradient chain is through, ONE table-driven launch of the
        # large-tile core computes all of them as full-K 256 x 256 items with the BatchNorm scale in the epilogue (ops.WgradTable: no K
        # slices, no slabs, no reduce launches).  Its operands are allocated with their row count rounded up to 128 (zero pad rows:
        # every kernel writes M rows).  +1.1 GB of gradient tensors at 8 images of 600 x 1000.  VLB_VISION_WGRAD_DEFER=0 switches it off.
        want_defer = os.environ.get("VLB_VISION_WGRAD_DEFER", "1") != "0" and self.implicit and d.type == "cuda"
        self.defer_layers = {3} if want_defer else set()
        self._row_parent = {}

        def zbp(M, C):                           # [M, C] view of a zero [round128(M), C] allocation
            full = zb(_ru(M, 128), C)
            view = full[:M]
            self._row_parent[view.data_ptr()] = full
            return view
        max_wg, max_dwf = 0, 0
        for b in self.blocks:
            L, P, C = b["layer"], b["planes"], b["inplanes"]
            if L == 4 and b["index"] == 0:       # the RoI head runs on the pooled RoI maps
                self.H3, self.W3, self.C3 = h, w, C          # body4: stride-16 trunk output (1024 channels)
                n, h, w = self.K, pooled, pooled
                self.roi = zb(self.K * pooled * pooled, C)
            b["n"], b["hin"], b["win"] = n, h, w
            if b["stride"] == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            b["h"], b["w"] = h, w
            M = n * h * w
            b["M"] = M
            tr = b["stage"] not in self.frozen_stages
            b["trainable"] = tr
            # (a first block without stride reads the previous stage's output; the table kernel needs >= 256 reduction rows)
            dfr = tr and L in self.defer_layers and (b["index"] > 0 or b["stride"] == 2) and _ru(M, 128) >= 256
            if tr and L in self.defer_layers and not dfr:
                self.defer_layers.discard(L)
            za = zbp if dfr else zb
            b["xs"] = za(M, C) if b["stride"] == 2 else None
            b["a"], b["b"], b["y"] = zb(M, P), za(M, P), za(M, 4 * P)
            b["r"] = zb(M, 4 * P) if b["downsample"] else None
            if tr and not self.implicit:
                b["col"] = zb(M, 9 * P)                      # kept for the weight gradient
            elif (tr and P % 128 != 0) or not self.implicit:  # implicit: only when the TN gather cannot be used (C % 128), built in backward
                if (M, P) not in shared_col:
                    shared_col[(M, P)] = zb(M, 9 * P)
                b["col"] = shared_col[(M, P)]
            else:
                b["col"] = None
            if L not in self.groups:
                g = dict(M=M, P=P, C=C, n=n, h=h, w=w)
                if tr:
                    g.update(dzA=zb(M, 4 * P), dzB=zb(M, 4 * P), da=[zb(M, P), zb(M, P)], db=[zb(M, P), zb(M, P)],
                             dcol=None if self.implicit else zb(M, 9 * P), tmp=zb(M, C), dxs=zb(M, C))
                self.groups[L] = g
            if tr:
                for (mo, no) in ((4 * P, P), (P, 9 * P), (P, C), (4 * P, C)):
                    max_wg = max(max_wg, ops.wgrad_workspace_floats(mo, no, M))
                    max_dwf = max(max_dwf, mo * no)
        self.M3 = N * self.H3 * self.W3
        self.Cout = 4 * self.blocks[-1]["planes"]        # 2048
        if self.Cout != VIS_DIM:
            raise ValueError("RoI head width %d != %d" % (self.Cout, VIS_DIM))
        self.P_roi = self.blocks[-1]["h"] * self.blocks[-1]["w"]
        # ROIAlign backward: gather form (VLB_ROI_BWD_GATHER=0: the atomic scatter into an fp32 map + separate mask / cast pass)
        self.roi_gather = os.environ.get("VLB_ROI_BWD_GATHER", "1") != "0" and self.C3 % 4 == 0 and pooled < 256
        self.dfeat32 = None if self.roi_gather else zf(self.M3, self.C3)
        self.roi_ws = ops.roi_align_gather_workspace(self.K, self.H3, self.W3, pooled, d) if self.roi_gather else None
        self.wg_ws = zf(max(max_wg, max_dwf, 4))         # split-K slabs (at least one slab of the largest weight)
        # ---- deferred 1x1 weight gradients: per-block gradient tensors + one descriptor table per stage ----------------
        self._tables = {}
        for L in sorted(self.defer_layers):
            g = self.groups[L]
            blks = [b for b in self.blocks if b["layer"] == L]
            M, P = g["M"], g["P"]
            g["dz"] = [zbp(M, 4 * P) for _ in blks]          # dz[i]: gradient of block i's pre-ReLU output
            g["da_list"] = [zbp(M, P) for _ in blks]
            g["dzA"], g["dzB"] = g["dz"][-1], None           # the stage's entry buffer (written by the next stage's backward)
            par = lambda t: self._row_parent[t.data_ptr()]
            items = []
            for i, b in enumerate(blks):
                k = b["key"]
                c1, c3 = self.convs[k + "conv1"], self.convs[k + "conv3"]
                xin = b["xs"] if b["stride"] == 2 else blks[i - 1]["y"]
                items.append((par(g["dz"][i]), par(b["b"]), c3.g32, None, c3.scale))
                items.append((par(g["da_list"][i]), par(xin), c1.g32, None, c1.scale))
                if b["downsample"]:
                    cd = self.convs[k + "downsample.0"]
                    items.append((par(g["dz"][i]), par(xin), cd.g32, None, cd.scale))
            tab = ops.WgradTable(items, d, accumulate=True)
            if not tab.ok:       # (eligibility was decided above from the row count; the leading dimensions are multiples of 8)
                raise RuntimeError("deferred weight gradients of layer%d: the table kernel refused the shapes" % L)
            self._tables[L] = tab
        self.wg_wss = [self.wg_ws] + [zf(self.wg_ws.numel()) for _ in range(len(self.sides) - 1)]       # one per side stream

    # ------------------------------------------------------------------------------------------------------------------
    def _dgrad_set(self):
        """convolutions whose input gradient is needed (everything trainable except the inputs of the first trainable block)."""
        need = set()
        first = True
        for b in self.blocks:
            if b["stage"] in self.frozen_stages:
                continue
            k = b["key"]
            need.update((k + "conv2", k + "conv3"))
            if not first:
                need.add(k + "conv1")
                if b["downsample"]:
                    need.add(k + "downsample.0")
            first = False
        return need

    def load_state_dict(self, sd, strict=True):
        """sd: reference names (PREFIX + backbone.* / roi_head_feature_extractor.*), conv weights in the reference's [O,I,KH,KW]."""
        for key, c in self.convs.items():
            name = PREFIX + key + ".weight"
            if name in sd:
                c.w32.copy_(sd[name].to(F32).permute(0, 2, 3, 1).reshape(c.O, c.taps, c.I))
            elif strict:
                raise KeyError("missing parameter %s" % name)
        for name, t in self.frozen.items():
            if name.endswith(".weight") and t.dim() == 3:
                continue
            if name in sd:
                t.copy_(sd[name].to(F32))
            elif strict:
                raise KeyError("missing buffer %s" % name)
        self._dirty = True

    def broadcast_tensors(self):
        """fp32 tensors that are NOT in the engine's flat parameter buffer (BatchNorm tensors, weights of the frozen stages):
        what a start-up parameter broadcast has to cover besides the flat buffer (engine.broadcast_parameters)."""
        self._dirty = True
        bad = [n for n, t in self.frozen.items() if not t.is_contiguous()]
        if bad:      # a view would be broadcast into a temporary and the replica left unsynchronised
            raise RuntimeError("vision.broadcast_tensors: non-contiguous frozen tensors %s" % bad[:4])
        return list(self.frozen.values())

    def state_dict(self):
        sd = OrderedDict()
        for key, c in self.convs.items():
            sd[PREFIX + key + ".weight"] = c.w32.detach().view(c.O, c.k, c.k, c.I).permute(0, 3, 1, 2).contiguous().clone()
        for name, t in self.frozen.items():
            if not (name.endswith(".weight") and t.dim() == 3):
                sd[name] = t.detach().clone()
        return sd

    def init_random(self, seed=0):
        """Convolutions ~ N(0, sqrt(2 / fan_out)) (resnet.py:153-155); BatchNorm identity except a 0.2 gain on the BN that closes
        each residual branch and the stem, which keeps random-init activations O(1) through 33 blocks (bench / smoke only)."""
        g = torch.Generator(device=self.dev).manual_seed(seed)
        for c in self.convs.values():
            c.w32.normal_(0.0, (2.0 / (c.O * c.taps)) ** 0.5, generator=g)
            c.bn[0].fill_(0.2 if (c.key.endswith("conv3") or c.key == "backbone.conv1") else 1.0)
            c.bn[1].zero_()
            c.bn[2].zero_()
            c.bn[3].fill_(1.0)
        self._dirty = True

    def grads(self):
        """{reference name: gradient in the reference's [O,I,KH,KW] layout} of the trainable convolutions."""
        return OrderedDict((PREFIX + key + ".weight", c.g32.detach().view(c.O, c.k, c.k, c.I).permute(0, 3, 1, 2).contiguous().clone())
                           for key, c in self.convs.items() if c.trainable)

    def zero_grad(self):
        for w, g in self._own.values():
            g.zero_()

    def refresh_weights(self, trainable_only=False):
        """fp32 master (+ frozen BN) -> folded bf16 operands; after load_state_dict and after every optimizer step."""
        key = bool(trainable_only)
        if key not in self._prep:
            self._prep[key] = ops.ConvPrepareBatch([(c.w32, c.bn, c.wf, c.wb, c.scale, c.shift) for c in self.convs.values()
                                                    if c.trainable or not trainable_only], self.dev, eps=BN_EPS)
        self._prep[key].run()
        self._dirty = False

    # ------------------------------------------------------------------------------------------------------------------
    def _block_fwd(self, b, x):
        cv, k, P = self.convs, b["key"], b["planes"]
        n, hin, win, h, w = b["n"], b["hin"], b["win"], b["h"], b["w"]
        xs = x
        if b["stride"] == 2:
            xs = ops.subsample2_nhwc(x, b["xs"], n, hin, win, b["inplanes"])
        c1, c2, c3 = cv[k + "conv1"], cv[k + "conv2"], cv[k + "conv3"]
        ops.gemm_nt(xs, c1.wf, b["a"], bias=c1.shift, act=ops.ACT_RELU)
        if self.implicit:
            ops.conv3x3_nhwc(b["a"], c2.wf, b["b"], n, h, w, P, b["dil"], self.zero16, bias=c2.shift, act=ops.ACT_RELU)
        else:
            ops.im2col_nhwc(b["a"], b["col"], n, h, w, P, 3, 1, b["dil"], b["dil"])
            ops.gemm_nt(b["col"], c2.wf, b["b"], bias=c2.shift, act=ops.ACT_RELU)
        res = x
        if b["downsample"]:
            cd = cv[k + "downsample.0"]
            res = ops.gemm_nt(xs, cd.wf, b["r"], bias=cd.shift)
        ops.gemm_nt(b["b"], c3.wf, b["y"], bias=c3.shift, res=res, act=ops.ACT_RES_RELU)
        b["x"] = x
        return b["y"]

    def forward(self, images, boxes, segms=None):
        """images fp32 [N,3,Himg,Wimg]; boxes fp32 [N,R,4+2048] (padded rows x1 = -2): fills boxes[:, :, 4:] with post_roialign.
        segms (optional, fp32 [N,R,pooled,pooled]): VCR's object masks, multiplied into the RoI-head output before the average pool
        (common/fast_rcnn.py:152-156); remembered for backward."""
        if self._dirty:
            self.refresh_weights()
        N, K = self.N, self.K
        assert tuple(images.shape) == (N, 3, self.Himg, self.Wimg) and images.dtype == F32
        assert boxes.shape[0] == N and boxes.shape[1] == self.R and boxes.shape[2] >= 4 + VIS_DIM and boxes.is_contiguous()
        stem = self.convs["backbone.conv1"]
        ops.im2col_image(images, self.col1)
        ops.gemm_nt(self.col1, stem.wf, self.c1, bias=stem.shift, act=ops.ACT_RELU)
        x = ops.maxpool3x3s2_nhwc(self.c1, self.p1, N, self.H1, self.W1, 64)
        box_rows = boxes.view(K, boxes.shape[2])
        for b in self.blocks:
            if b["layer"] == 4 and b["index"] == 0:
                self.body4 = x
                x = ops.roi_align_nhwc_fwd(x, box_rows, self.R, self.roi, N, self.H3, self.W3, self.C3, self.pooled, self.scale, self.sr)
            x = self._block_fwd(b, x)
        self._segm = None
        if segms is not None:
            assert tuple(segms.shape) == (N, self.R, self.pooled, self.pooled) and self.P_roi == self.pooled * self.pooled
            self._segm = segms.to(F32).contiguous().view(K, self.P_roi)
        ops.avgpool_rows_fwd(x, box_rows, 4, K, self.P_roi, self.Cout, pad_col=0, segm=self._segm)
        return x

    # ------------------------------------------------------------------------------------------------------------------
    def _side_run(self, fn, *reads):
        """Run fn(workspace) on the next weight-gradient stream after everything enqueued so far; `reads` are the transient buffers it
        reads."""
        if self.side is None:
            fn(self.wg_ws)
            return
        i = self._side_rr
        self._side_rr = (i + 1) % len(self.sides)
        side = self.sides[i]
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            fn(self.wg_wss[i])
            done = torch.cuda.Event()
            done.record()
        for t in reads:          # a buffer read by several pending weight gradients: the next writer waits for all of them
            self._pending.setdefault(t.data_ptr(), []).append(done)

    def _before_write(self, *bufs):
        for t in bufs:
            evs = self._pending.pop(t.data_ptr(), None) if t is not None else None
            for ev in evs or ():
                torch.cuda.current_stream().wait_event(ev)

    def _join_side(self):
        for side in self.sides:
            ev = torch.cuda.Event()
            ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
        self._pending.clear()

    def _wgrad(self, c, dy, x, conv=None):
        """g32 += scale[o] * (dy^T x) for the folded operand; conv = (n, h, w, C, dil): x is the NHWC activation and the im2col
        gather happens inside the TN GEMM."""
        def run(ws):   # the BatchNorm scale is applied by the split-K slab reduce (no separate finalize pass)
            if conv is None:
                ops.wgrad_tn_rowscale(dy, x, c.g32, c.scale, ws, accumulate=True)
            else:
                ops.conv3x3_wgrad_tn(dy, x, c.g32, *conv, workspace=ws, accumulate=True, rowscale=c.scale)
        self._side_run(run, dy, x)

    def _block_bwd(self, b, dz, dx_out, need_dx, mask_input):
        """dz: masked gradient of this block's pre-ReLU output [M, 4P].  Writes the (masked) gradient of the block input into
        dx_out ([M_in, C]) when need_dx."""
        cv, k, P, C = self.convs, b["key"], b["planes"], b["inplanes"]
        g = self.groups[b["layer"]]
        n, h, w = b["n"], b["h"], b["w"]
        c1, c2, c3 = cv[k + "conv1"], cv[k + "conv2"], cv[k + "conv3"]
        x = b["x"]
        xs = b["xs"] if b["stride"] == 2 else x
        defer = b["layer"] in self._tables       # this stage's 1x1 weight gradients go out as ONE table launch when the stage is through
        da, db = (g["da_list"][b["index"]] if defer else g["da"][b["index"] & 1]), g["db"][b["index"] & 1]
        if not defer:
            self._wgrad(c3, dz, b["b"])
        self._before_write(db)
        ops.gemm_nt(dz, c3.wb, db, act=ops.ACT_RELU_MASK, aux=b["b"])
        self._before_write(da)
        if self.implicit:
            if b["col"] is None:      # weight gradient with the gather inside the TN GEMM
                self._wgrad(c2, db, b["a"], conv=(n, h, w, P, b["dil"]))
            else:
                self._join_side()     # (shared im2col buffer)
                ops.im2col_nhwc(b["a"], b["col"], n, h, w, P, 3, 1, b["dil"], b["dil"])
                self._wgrad(c2, db, b["col"])
            ops.conv3x3_nhwc(db, c2.wb, da, n, h, w, P, b["dil"], self.zero16, act=ops.ACT_RELU_MASK, aux=b["a"])
        else:
            self._wgrad(c2, db, b["col"])
            ops.im2col_nhwc(db, g["dcol"], n, h, w, P, 3, 1, b["dil"], b["dil"])
            ops.gemm_nt(g["dcol"], c2.wb, da, act=ops.ACT_RELU_MASK, aux=b["a"])
        cd = cv[k + "downsample.0"] if b["downsample"] else None
        if not defer:
            self._wgrad(c1, da, xs)
            if cd is not None:
                self._wgrad(cd, dz, xs)
        if not need_dx:
            return
        res = dz
        if cd is not None:
            res = ops.gemm_nt(dz, cd.wb, g["tmp"])
        target = g["dxs"] if b["stride"] == 2 else dx_out
        self._before_write(dx_out)
        if mask_input:
            ops.gemm_nt(da, c1.wb, target, res=res, act=ops.ACT_RELU_MASK, aux=xs)
        else:
            ops.gemm_nt(da, c1.wb, target, res=res)
        if b["stride"] == 2:
            ops.upsample2_zero_nhwc(target, dx_out, n, b["hin"], b["win"], C)

    def backward(self, d_feat, boxes, drop_p=0.0, seed=None, tag=0, drop_row_elems=2 * VIS_DIM, drop_col0=VIS_DIM, on_stage_done=None):
        """d_feat bf16 [K, 2048]: gradient w.r.t. the feature half of obj_downsample's (dropped-out) input; accumulates the weight
        gradients of the trainable convolutions.  on_stage_done(layer) is called when every weight gradient of layer 4 (RoI head),
        3, 2 is complete on the current stream (data-parallel bucket hook)."""
        self._backward(d_feat, boxes, drop_p, seed, tag, drop_row_elems, drop_col0, on_stage_done)
        self._join_side()        # every weight gradient is complete for whatever the caller enqueues next

    def _deferred_wgrads(self, L):
        """The stage's deferred 1x1 weight gradients: one table launch on a side stream, behind everything enqueued so far."""
        tab = self._tables.get(L)
        if tab is not None:
            self._side_run(lambda ws: tab.run())

    def _stage_done(self, hook, layer):
        if hook is not None:
            self._join_side()
            hook(layer)

    def _backward(self, d_feat, boxes, drop_p, seed, tag, drop_row_elems, drop_col0, hook):
        K = self.K
        box_rows = boxes.view(K, boxes.shape[2])
        blocks = [b for b in self.blocks if b["trainable"]]
        first = blocks[0]
        last = self.blocks[-1]
        g4 = self.groups[4]
        self._before_write(g4["dzA"])
        cur = ops.avgpool_rows_bwd(d_feat, last["y"], box_rows, g4["dzA"], K, self.P_roi, self.Cout, drop_p=drop_p, seed=seed, tag=tag,
                                   drop_row_elems=drop_row_elems, drop_col0=drop_col0, segm=getattr(self, "_segm", None))
        for b in reversed(blocks):
            L, g = b["layer"], self.groups[b["layer"]]
            if b["index"] > 0:
                if L in self._tables:
                    other = g["dz"][b["index"] - 1]
                else:
                    other = g["dzB"] if cur is g["dzA"] else g["dzA"]
                self._block_bwd(b, cur, other, True, True)
                cur = other
                continue
            # first block of a layer: its input is the previous layer's output (or the pooled RoI maps for the head)
            if L == 4:
                if 4 in self.frozen_stages or all(bb["layer"] == 4 for bb in blocks):
                    self._block_bwd(b, cur, None, False, False)
                    self._stage_done(hook, 4)
                    return
                self._block_bwd(b, cur, g["dxs"], True, False)          # ROIAlign output: no ReLU in front, no mask
                self._stage_done(hook, 4)
                g3 = self.groups[3]
                self._before_write(g3["dzA"])
                if self.roi_gather:      # gather per feature pixel: no atomics / memset, ReLU mask + bf16 cast in the store
                    cur = ops.roi_align_nhwc_bwd_gather(g["dxs"], box_rows, self.R, self.roi_ws, self.N, self.H3, self.W3, self.C3,
                                                        act=self.body4, dx_bf16=g3["dzA"], pooled=self.pooled,
                                                        spatial_scale=self.scale, sampling_ratio=self.sr)
                else:
                    ops.roi_align_nhwc_bwd(g["dxs"], box_rows, self.R, self.dfeat32, self.N, self.H3, self.W3, self.C3, self.pooled,
                                           self.scale, self.sr)
                    cur = ops.relu_mask_cast(self.dfeat32, self.body4, g3["dzA"])
            elif b is first:
                self._block_bwd(b, cur, None, False, False)
                self._deferred_wgrads(L)
                self._stage_done(hook, L)
            else:
                prev = self.groups[L - 1]
                self._block_bwd(b, cur, prev["dzA"], True, True)
                cur = prev["dzA"]
                self._deferred_wgrads(L)
                self._stage_done(hook, L)
