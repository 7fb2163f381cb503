"""Generate tests/golden/checkpoint/ by running the REAL reference (imported from /root/reference) on CPU.  Build container only:

    python oracle/make_checkpoint_golden.py

TEST INFRASTRUCTURE ONLY (see oracle/vlbert_oracle.py header).  What is produced pins the on-disk checkpoint format the drop-in must
read and write (common/callbacks/epoch_end_callbacks/checkpoint.py:12-21, common/utils/load.py:20-54):

  * param_order.json -- `[n for n, _ in model.named_parameters()]` of the reference's ResNetVLBERTForPretraining (plain, pooler +
    relationship head) and ResNetVLBERTForPretrainingMultitask: the index order of `optimizer.state_dict()["state"]` /
    `["param_groups"][..]["params"]` (pretrain/function/train.py:140-160 builds one group over named_parameters(); every shipped
    pretrain cfg has LR_MULT: []).
  * ref_small-0000.model -- a checkpoint written by the reference's own `Checkpoint` callback after 2 steps of the reference's AdamW on a
    seeded batch (tiny model: 2 layers x 64), plus ref_small_batch.npz = the batch and the reference's loss on it after reloading:
    the drop-in must load this file, reproduce the loss, continue the optimizer trajectory, and write a file the reference loads.
"""
import importlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.vlbert_oracle import VLBertConfig, init_params  # noqa: E402

synthetic = importlib.import_module("vl-bert_amd.synthetic")
OUT = os.path.join(ROOT, "tests", "golden", "checkpoint")

SMALL = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128, vocab_size=300,
             max_position_embeddings=64, visual_region_classes=50)


def build(cfg, tag):
    RefModel, RefAdamW = ref_import.import_reference()
    if cfg.multitask:
        RefModel = ref_import.import_reference_multitask()
    vocab_dir = ref_import.make_vocab_dir(os.path.join(tempfile.gettempdir(), "vlb_vocab_ckpt_%s" % tag), cfg.vocab_size)
    torch.manual_seed(0)
    return RefModel(ref_import.make_reference_config(cfg, vocab_dir)), RefAdamW


def e2e_fastrcnn_names(num_layers):
    """named_parameters() of the reference's FastRCNN with IMAGE_FEAT_PRECOMPUTED false (common/fast_rcnn.py:54-109): the head of the
    e2e models' optimizer index order (image_feature_extractor is the wrappers' first sub-module)."""
    from oracle import vision_oracle as VO
    ref_import.import_reference()
    ref_import.install_roi_align_oracle()
    import common.lib.roi_pooling as rp
    rp.C_ROIPooling = sys.modules["common.lib.roi_pooling.C_ROIPooling"]
    import common.lib.roi_pooling.roi_align as ra_mod
    ra_mod.C_ROIPooling = rp.C_ROIPooling
    from common.fast_rcnn import FastRCNN as RefFastRCNN
    P = VO.init_vision_params(7, num_layers)
    E = ref_import._EasyDict
    cfg = E(dict(NETWORK=dict(IMAGE_FEAT_PRECOMPUTED=False, IMAGE_SEMANTIC=False, IMAGE_STRIDE_IN_1x1=True, IMAGE_C5_DILATED=True,
                              IMAGE_NUM_LAYERS=num_layers, IMAGE_PRETRAINED="oracle_init", IMAGE_PRETRAINED_EPOCH=0,
                              OUTPUT_CONV5=False, IMAGE_FROZEN_BN=True, IMAGE_FROZEN_BACKBONE_STAGES=[1, 2])))
    real_load = torch.load
    torch.load = lambda path, *a, **k: dict(P) if str(path).startswith("oracle_init") else real_load(path, *a, **k)
    try:
        model = RefFastRCNN(cfg, average_pool=True, final_dim=768, enable_cnn_reg_loss=False)
    finally:
        torch.load = real_load
    return [n for n, _ in model.named_parameters()], [n for n, p in model.named_parameters() if p.requires_grad]


def add_e2e_order():
    """param_order.json += e2e_fastrcnn_{50,101} (all parameters) and e2e_fastrcnn_{50,101}_trainable (requires_grad) -- ADVICE r4."""
    path = os.path.join(OUT, "param_order.json")
    order = json.load(open(path))
    for nl in (50, 101):
        order["e2e_fastrcnn_%d" % nl], order["e2e_fastrcnn_%d_trainable" % nl] = e2e_fastrcnn_names(nl)
    with open(path, "w") as f:
        json.dump(order, f, indent=0)


def main():
    if "--e2e-order-only" in sys.argv:
        add_e2e_order()
        return
    os.makedirs(OUT, exist_ok=True)
    order = {}
    for tag, kw in (("plain", {}), ("pooler_rel", dict(with_pooler=True, with_rel_loss=True)), ("multitask", dict(multitask=True))):
        model, _ = build(VLBertConfig(**SMALL, **kw), tag)
        order[tag] = [n for n, _ in model.named_parameters()]
    with open(os.path.join(OUT, "param_order.json"), "w") as f:
        json.dump(order, f, indent=0)

    cfg = VLBertConfig(**SMALL)
    model, RefAdamW = build(cfg, "ckpt")
    params = init_params(cfg, seed=21)
    sd = dict(params)
    sd["vlbert.mlm_head.predictions.decoder.weight"] = sd["vlbert.word_embeddings.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()                                   # dropout off: the trajectory must be reproducible by the drop-in
    batch = synthetic.make_batch(3, 12, 5, vocab_size=cfg.vocab_size, region_classes=cfg.visual_region_classes, seed=41, ragged=True)
    # pretrain/function/train.py:140-160: one param group over named_parameters(), lr = LR x batch size
    opt = RefAdamW([{"params": [p for _, p in model.named_parameters()]}], lr=2e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4,
                   correct_bias=True)
    losses = []
    for _ in range(2):
        outputs, loss = model(None, *[t.clone() for t in batch])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        losses.append(float(loss))
    Checkpoint = importlib.import_module("common.callbacks.epoch_end_callbacks.checkpoint").Checkpoint
    prefix = os.path.join(OUT, "ref_small")
    Checkpoint(prefix, 1)(0, model, opt, None)     # -> ref_small-0000.model
    # what the reference computes after resuming from that file: the loss on the same batch, then one more step, the loss again
    ck = torch.load(prefix + "-0000.model", map_location="cpu")
    model2, _ = build(cfg, "ckpt2")
    model2.load_state_dict(ck["state_dict"])
    model2.eval()
    opt2 = RefAdamW([{"params": [p for _, p in model2.named_parameters()]}], lr=2e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=1e-4,
                    correct_bias=True)
    opt2.load_state_dict(ck["optimizer"])
    after = []
    for _ in range(2):
        outputs, loss = model2(None, *[t.clone() for t in batch])
        opt2.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model2.parameters(), 10.0)
        opt2.step()
        after.append(float(loss))
    probe = "vlbert.encoder.layer.1.output.dense.weight"
    np.savez_compressed(os.path.join(OUT, "ref_small_batch.npz"), cfg_keys=np.array(list(SMALL.keys())),
                        cfg_vals=np.array([float(v) for v in SMALL.values()]), losses_before=np.array(losses), losses_after=np.array(after),
                        probe_name=np.array(probe), probe_after=dict(model2.named_parameters())[probe].detach().numpy(),
                        **{"in_%d" % i: t.numpy() for i, t in enumerate(batch)})
    print("wrote", OUT, "losses", losses, after, "optimizer keys", list(ck["optimizer"].keys()),
          "group keys", list(ck["optimizer"]["param_groups"][0].keys()), "state[0] keys", list(ck["optimizer"]["state"][0].keys()))
    add_e2e_order()


if __name__ == "__main__":
    main()
