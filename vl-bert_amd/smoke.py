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
