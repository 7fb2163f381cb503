"""One tiny training step of the hot path on cuda:0, checked against the CPU oracle
(__graft_entry__.smoke).  The oracle is the checker only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import vlbert_oracle as O          # test infrastructure: the checker
    from . import _lib, engine, synthetic

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU: the product path has no CPU fallback")
    name, cus = _lib.device_info(0)
    if not name.startswith("gfx950"):
        raise RuntimeError("libvlbert_hip.so is built for gfx950 only, found %s" % name)
    cfg = O.VLBertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         vocab_size=512, max_position_embeddings=64, visual_region_classes=50)
    B, T, R = 3, 12, 5
    params = O.init_params(cfg, seed=3)
    batch = synthetic.make_batch(B, T, R, vocab_size=cfg.vocab_size, region_classes=cfg.visual_region_classes, seed=11,
                                 ragged=True)
    mc = engine.ModelConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                            vocab_size=512, max_position_embeddings=64, visual_region_classes=50)
    eng = engine.PretrainEngine(mc, B, T, R, device="cuda:0", train=False)
    eng.load_state_dict({k: v.cuda() for k, v in params.items()})
    eng.set_batch(*[t.cuda() for t in batch])
    eng.zero_grad()
    eng.forward(train=False)
    eng.backward(train=False)
    eng.optimizer_step(lr=1e-4)
    torch.cuda.synchronize()
    _, loss, _, norm = O.loss_and_grads(params, cfg, batch, train=False)
    lv = eng.loss_values()
    gn = eng.grad_norm()
    print("smoke: %s (%d CUs) loss hip %.5f oracle %.5f | grad-norm hip %.5f oracle %.5f" % (name, cus, lv["loss"], float(loss), gn, norm))
    assert abs(lv["loss"] - float(loss)) <= 1e-2 * float(loss), "loss mismatch"
    assert abs(gn - norm) <= 1e-2 * norm, "grad-norm mismatch"
    run_e2e()


def run_e2e():
    """The e2e vision path on the reference-generated fixture (tests/golden/vision/vision_small.npz: ResNet-50 trunk, ROIAlign,
    dilated layer4 head on 2 images of 96x128): forward features + one backward, against the fixture / the vision oracle."""
    import numpy as np
    from oracle import vision_oracle as VO         # test infrastructure: the checker
    from . import ops, vision
    z = np.load(os.path.join(ROOT, "tests", "golden", "vision", "vision_small.npz"), allow_pickle=False)
    nl = int(z["num_layers"])
    P = VO.init_vision_params(int(z["seed"]), nl)
    img, boxes4 = torch.from_numpy(z["img"]), torch.from_numpy(z["boxes"])
    N, R = boxes4.shape[:2]
    vs = vision.VisionStack(N, img.shape[2], img.shape[3], R, device="cuda:0", num_layers=nl)
    vs.load_state_dict({"image_feature_extractor." + k: v.cuda() for k, v in VO.split_state_dict(P).items()})
    boxes = torch.zeros((N, R, 4 + 2048), device="cuda:0")
    boxes[:, :, :4] = boxes4.cuda()
    vs.forward(img.cuda(), boxes)
    vs.zero_grad()
    vs.backward(torch.from_numpy(z["Wr"]).view(N * R, -1).to(ops.BF16).cuda(), boxes)
    torch.cuda.synchronize()
    raw = torch.from_numpy(z["obj_reps_raw"])
    err = float((boxes[:, :, 4:].cpu() - raw).abs().max()) / float(raw.abs().max())
    name = "image_feature_extractor.roi_head_feature_extractor.2.conv3.weight"
    want = dict(zip([str(k) for k in z["grad_names"]], z["grad_norms"]))[name[len("image_feature_extractor."):]]
    got = float(vs.grads()[name].double().norm())
    print("smoke e2e: post_roialign max err / scale %.2e | |d %s| hip %.5f reference %.5f" % (err, name.split(".", 1)[1], got, want))
    assert err <= 2e-2, "e2e feature mismatch"
    assert abs(got - want) <= 3e-2 * want, "e2e gradient-norm mismatch"
