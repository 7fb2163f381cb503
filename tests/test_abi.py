"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vlbert_hip.h declares, and the ctypes signatures match the header prototypes.
No compute calls (no GPU here)."""
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vlbert_hip.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|long|const char\*)\s+(vlb_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2)
        sig = ""
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "vlb_stream_t" in a:
                sig += "s"
            elif "*" in a:
                sig += "p"
            elif re.match(r"(const\s+)?long\b", a):
                sig += "l"
            elif re.match(r"(const\s+)?uint32_t\b", a):
                sig += "u"
            elif re.match(r"(const\s+)?float\b", a):
                sig += "f"
            elif re.match(r"(const\s+)?int\b", a):
                sig += "i"
            else:
                raise AssertionError("unparsed arg %r in %s" % (a, name))
        protos[name] = sig
    return protos


def test_header_declares_expected_entry_points():
    protos = header_prototypes()
    assert len(protos) >= 25
    for must in ("vlb_gemm_nt_bf16", "vlb_attention_fwd", "vlb_attention_bwd", "vlb_layernorm_fwd", "vlb_layernorm_bwd",
                 "vlb_embed_fwd", "vlb_ce_fwd_bwd", "vlb_adamw_step", "vlb_roi_align_fwd", "vlb_roi_align_bwd"):
        assert must in protos


def test_ctypes_signatures_match_header():
    lib = importlib.import_module("vl-bert_amd._lib")
    protos = header_prototypes()
    for name, sig in lib._SIGS.items():
        assert name in protos, "binding %s has no prototype in include/vlbert_hip.h" % name
        assert protos[name] == sig, "%s: header %s vs binding %s" % (name, protos[name], sig)
    for name in protos:
        assert name in lib._SIGS or name in ("vlb_last_error", "vlb_version", "vlb_act_dtype", "vlb_device_info", "vlb_wgrad_workspace_floats",
                                            "vlb_layernorm_bwd_workspace_floats", "vlb_layernorm_bwd_slabs", "vlb_gemm_set_option",
                                                "vlb_roi_align_gather_workspace_bytes", "vlb_nonfinite_status", "vlb_wgrad_tn_table_desc_bytes"), name


def test_library_loads_and_exports_every_symbol():
    lib = importlib.import_module("vl-bert_amd._lib")
    if not os.path.isfile(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    h = lib.load()
    for name in header_prototypes():
        assert hasattr(h, name), "libvlbert_hip.so does not export %s" % name
    assert h.vlb_version() >= 100
    assert isinstance(h.vlb_last_error(), bytes)
    assert h.vlb_act_dtype() == 0                      # the default build computes in bfloat16


def test_fp16_build_of_the_library_exports_the_same_abi():
    """libvlbert_hip_f16.so = the same sources with IEEE fp16 as the 16-bit type (the reference's Apex fp16 mode): same symbols."""
    import ctypes
    lib = importlib.import_module("vl-bert_amd._lib")
    path = lib._default_path("f16")
    if not os.path.isfile(path):
        import __graft_entry__
        __graft_entry__.build()
    h = ctypes.CDLL(path)
    for name in header_prototypes():
        assert hasattr(h, name), "libvlbert_hip_f16.so does not export %s" % name
    assert h.vlb_act_dtype() == 1


def test_missing_library_fails_loudly(monkeypatch):
    lib = importlib.import_module("vl-bert_amd._lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libvlbert_hip.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        lib.load()


def test_train_end2end_entry_point_dry_run(tmp_path):
    """The reference-style command line (pretrain/train_end2end.py) resolves a cfgs/pretrain/*.yaml the reference's way: lr scaled by
    world x batch x accumulate (pretrain/function/train.py:133-138), triangle schedule totals (:316-320); without a GPU it refuses
    to run (no CPU path) unless --dry-run."""
    import torch
    tr = importlib.import_module("vl-bert_amd.pretrain.train_end2end")
    y = tmp_path / "cfg.yaml"
    y.write_text("MODULE: ResNetVLBERTForPretraining\nRNG_SEED: 7\nSCALES: [600, 1000]\n"
                 "NETWORK:\n  IMAGE_FEAT_PRECOMPUTED: false\n  IMAGE_NUM_LAYERS: 101\n  VLBERT:\n    num_hidden_layers: 12\n    hidden_size: 768\n"
                 "TRAIN:\n  BATCH_IMAGES: 8\n  LR: 1.0e-7\n  WD: 0.0001\n  CLIP_GRAD_NORM: 10\n  LR_SCHEDULE: triangle\n  WARMUP: true\n"
                 "  WARMUP_STEPS: 8000\n  END_EPOCH: 10\n  GRAD_ACCUMULATE_STEPS: 2\n  FP16: true\n")
    r = tr.main(["--cfg", str(y), "--dry-run", "--steps-per-epoch", "1000"])
    assert r["e2e"] and r["image_size"] == (600, 1000) and r["per_gpu_batch"] == 8 and r["accumulate"] == 2
    assert abs(r["lr"] - 1e-7 * 8 * 2) < 1e-15 and r["warmup_steps"] == 8000 and r["t_total"] == 5000 and r["lr_schedule"] == "triangle"
    ref = "/root/reference/cfgs/pretrain/base_prec_withouttextonly_4x16G_fp32.yaml"
    if os.path.isfile(ref):            # the reference's own file, as it is
        r2 = tr.main(["--cfg", ref, "--dry-run"])
        assert not r2["e2e"] and r2["per_gpu_batch"] == 64 and abs(r2["lr"] - 64e-7) < 1e-15 and not r2["multitask"]
    ref3 = "/root/reference/cfgs/pretrain/base_e2e_16x16G_fp16.yaml"
    if os.path.isfile(ref3):           # multitask cfg: BATCH_IMAGES [8, 8] = 8 caption + 8 text-only samples per GPU, lr scaled by their sum
        r3 = tr.main(["--cfg", ref3, "--dry-run"])
        assert r3["multitask"] and r3["e2e"] and r3["per_gpu_batch"] == 8 and r3["per_gpu_aux_batch"] == 8
        assert r3["global_batch"] == 16 and abs(r3["lr"] - 1.0e-7 * 16 * r3["accumulate"]) < 1e-15
    # the repository's own cfgs/pretrain/base_e2e_16x16G_fp16.yaml (BASELINE config 3): resolves the same way and carries the
    # reference file's value for every key it names
    mine = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs", "pretrain", "base_e2e_16x16G_fp16.yaml")
    r4 = tr.main(["--cfg", mine, "--dry-run"])
    assert r4["multitask"] and r4["e2e"] and (r4["per_gpu_batch"], r4["per_gpu_aux_batch"]) == (8, 8) and r4["image_size"] == (600, 1000)
    assert abs(r4["lr"] - 1.6e-6) < 1e-15 and r4["warmup_steps"] == 16000 and r4["fp16_requested"] and r4["seed"] == 12345
    if os.path.isfile(ref3):
        import yaml
        assert r4 == r3

        def same(a, b, path=""):
            for k, v in b.items():
                assert k in a, path + k
                if isinstance(v, dict):
                    same(a[k], v, path + k + ".")
                else:
                    assert a[k] == v, (path + k, a[k], v)
        with open(ref3) as f, open(mine) as g:
            same(yaml.safe_load(f), yaml.safe_load(g))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU execution path"):
            tr.main(["--cfg", str(y), "--steps", "1"])


def test_vision_host_tables_match_the_reference_state_dict_layout():
    """Host logic of the e2e path (no GPU): the convolution / parameter tables of vl-bert_amd/vision.py name exactly the tensors of the
    reference's FastRCNN state dict (as restated by oracle/vision_oracle.py, itself pinned to the reference module), with the
    reference's shapes up to the documented [O,I,KH,KW] -> [O,KH,KW,I] permutation, and the frozen / trainable split of
    IMAGE_FROZEN_BACKBONE_STAGES = [1, 2] + IMAGE_FROZEN_BN."""
    from oracle import vision_oracle as VO
    V = importlib.import_module("vl-bert_amd.vision")
    for nl in (50, 101):
        P = VO.init_vision_params(0, nl, randomize_bn=False)
        ref = {"image_feature_extractor." + k: v for k, v in VO.split_state_dict(P).items()}
        frozen = {"image_feature_extractor." + k for k in VO.split_state_dict({n: P[n] for n in VO.frozen_names(P)})}
        convs = V.conv_table(nl, (1, 2))
        names = set()
        for key, O, I, k, bn, tr in convs:
            w = "image_feature_extractor." + key + ".weight"
            assert tuple(ref[w].shape) == (O, I, k, k), w
            assert (w in frozen) == (not tr), w
            names.add(w)
            for suffix in ("weight", "bias", "running_mean", "running_var"):
                b = "image_feature_extractor." + bn + "." + suffix
                assert tuple(ref[b].shape) == (O,) and b in frozen, b
                names.add(b)
        assert names == set(ref), names ^ set(ref)
        lay = V.vision_param_layout(nl, (1, 2))
        assert set(lay) == {n for n in ref if n not in frozen}
        for n, shp in lay.items():
            O, I, kh, kw = ref[n].shape
            assert shp == (O, kh, kw, I), n
        # geometry of the e2e configuration: 600x1000 -> stride-16 body4 of 38x63, 14x14 RoI maps
        blocks = V.block_table(nl)
        assert [b["stride"] for b in blocks if b["index"] == 0] == [1, 2, 2, 1] and blocks[-1]["dil"] == 2
        assert sum(b["downsample"] for b in blocks) == 4


def test_vision_stack_geometry_on_host():
    """VisionStack bookkeeping without a GPU (buffers on the CPU device, no kernel is launched): the e2e configuration's geometry --
    600x1000 images -> 300x500 stem -> 150x250 -> layer2 75x125 -> layer3 (body4) 38x63, 14x14 RoI maps for every box slot --
    and the hazard-free buffer plan (ping-pong gradients, parity-double-buffered da / db, no im2col image with the implicit GEMMs)."""
    V = importlib.import_module("vl-bert_amd.vision")
    vs = V.VisionStack(2, 600, 1000, 36, device="cpu", num_layers=101)
    assert (vs.H1, vs.W1, vs.Hp, vs.Wp, vs.H3, vs.W3, vs.C3, vs.P_roi, vs.Cout) == (300, 500, 150, 250, 38, 63, 1024, 196, 2048)
    by = {b["key"]: b for b in vs.blocks}
    assert by["backbone.layer2.0."]["M"] == 2 * 75 * 125 and by["backbone.layer2.0."]["stride"] == 2
    assert by["backbone.layer3.22."]["M"] == 2 * 38 * 63 and by["roi_head_feature_extractor.0."]["M"] == 72 * 196
    assert by["roi_head_feature_extractor.0."]["dil"] == 2 and by["roi_head_feature_extractor.0."]["downsample"]
    assert not by["backbone.layer1.0."]["trainable"] and by["backbone.layer2.0."]["trainable"]
    assert all(b["col"] is None for b in vs.blocks if b["trainable"])                 # implicit GEMM: no im2col image kept
    g3 = vs.groups[3]
    assert g3["dzA"].shape == (2 * 38 * 63, 1024) and g3["da"][0].data_ptr() != g3["da"][1].data_ptr() and g3["dxs"].shape == (2 * 38 * 63, 512)
    assert len(vs.convs) == 104 and sum(c.trainable for c in vs.convs.values()) == 93
    need = vs._dgrad_set()
    assert "backbone.layer2.0.conv1" not in need and "backbone.layer2.0.conv2" in need and "roi_head_feature_extractor.0.downsample.0" in need
    sd = vs.state_dict()
    assert tuple(sd["image_feature_extractor.backbone.layer2.0.conv2.weight"].shape) == (128, 128, 3, 3)      # reference layout out
    with pytest.raises(NotImplementedError):
        V.VisionStack(1, 64, 64, 2, device="cpu", num_layers=50, frozen_stages=(2,))
