"""The reference's fine-tuning entry points over the module mirrors:

    python -m vl-bert_amd.vqa.train_end2end --cfg cfgs/vqa/large_4x16G_fp32.yaml [--dist] [--steps N]      (vqa/train_end2end.py:12-60)
    python -m vl-bert_amd.vcr.train_end2end --cfg cfgs/vcr/large_q2a_4x16G_fp16.yaml [--dist] [--steps N]  (vcr/train_end2end.py)

Same command line as the reference's scripts (--cfg / --model-dir / --log-dir / --dist / --slurm / --do-test / --cudnn-off /
--partial-pretrain).  The YAML is read as it is; the keys the training loop consumes follow vqa/function/train.py:96-330 and
vcr/function/train.py:96-335:
  * model = MODULE (`ResNetVLBERT`) built from NETWORK.*, wrapped in DistributedDataParallel with --dist (:327 / :330);
  * lr = TRAIN.LR x world x BATCH_IMAGES x GRAD_ACCUMULATE_STEPS (:116-121 / :118-123); AdamW(betas 0.9/0.999, eps 1e-6, WD,
    correct_bias) or SGD(momentum TRAIN.MOMENTUM, WD) (:124-143) -> FusedAdamW / FusedSGD;
  * 'triangle' = WarmupLinearSchedule, 'step' = WarmupMultiStepLR(LR_STEP epochs, LR_FACTOR, linear warm-up from WARMUP_FACTOR)
    (common/lr_scheduler.py:7-49, common/nlp/bert/optimization.py:49-62), evaluated per optimizer step;
  * per step: GRAD_ACCUMULATE_STEPS micro-batches of loss / accumulate, clip_grad_norm_(CLIP_GRAD_NORM), optimizer step
    (common/trainer.py:101-189); TRAIN.FP16 -> the fp16 build + static loss scale FP16_LOSS_SCALE ("--compute cfg").
Not reproduced: the data side (datasets, tokeniser, image decoding: batches are synthetic in the collated layouts of
vqa/data/collate_batch.py / vcr/data/collate_batch.py), validation / test-set csv writing, tensorboard.  No CPU path: without a GPU the
program stops with an error unless --dry-run is given.
"""
import argparse
import importlib
import json
import os
import sys
import time


class AttrDict(dict):
    def __getattr__(self, key):          # (AttributeError, not KeyError: copy / pickle probe for dunder attributes)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    @staticmethod
    def wrap(x):
        if isinstance(x, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in x.items()})
        return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


# defaults of vqa/function/config.py / vcr/function/config.py for the keys the mirrors and this loop read
_NET = {"IMAGE_FEAT_PRECOMPUTED": False, "IMAGE_SEMANTIC": False, "IMAGE_STRIDE_IN_1x1": True, "IMAGE_C5_DILATED": True, "IMAGE_NUM_LAYERS": 101,
        "OUTPUT_CONV5": False, "IMAGE_FROZEN_BN": True, "IMAGE_FROZEN_BACKBONE_STAGES": [1, 2], "IMAGE_FINAL_DIM": 768, "BLIND": False,
        "NO_GROUNDING": False, "NO_OBJ_ATTENTION": False, "ANSWER_FIRST": False, "QA_ONE_SENT": False, "FOR_MASK_VL_MODELING_PRETRAIN": False,
        "ENABLE_CNN_REG_LOSS": False, "CNN_LOSS_TOP": False, "CNN_REG_DROPOUT": 0.0, "CNN_LOSS_WEIGHT": 1.0, "ANS_LOSS_WEIGHT": 1.0,
        "CLASSIFIER_TYPE": "2fc", "CLASSIFIER_HIDDEN_SIZE": 1024, "CLASSIFIER_DROPOUT": 0.1, "CLASSIFIER_SIGMOID": False,
        "CLASSIFIER_SIGMOID_LOSS_POSITIVE_WEIGHT": 1.0,
        "VLBERT": {"hidden_size": 768, "visual_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
                   "vocab_size": 30522, "max_position_embeddings": 512, "type_vocab_size": 3, "visual_ln": True, "with_pooler": False,
                   "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1, "initializer_range": 0.02,
                   "visual_scale_text_init": 0.0, "visual_scale_object_init": 0.0, "object_word_embed_mode": 2}}
_TRAIN = {"BATCH_IMAGES": 1, "LR": 1e-5, "WD": 1e-4, "MOMENTUM": 0.9, "CLIP_GRAD_NORM": -1, "LR_SCHEDULE": "step", "LR_FACTOR": 0.1, "LR_STEP": (),
          "WARMUP": False, "WARMUP_METHOD": "linear", "WARMUP_FACTOR": 1.0 / 3, "WARMUP_STEPS": 1000, "BEGIN_EPOCH": 0, "END_EPOCH": 1,
          "GRAD_ACCUMULATE_STEPS": 1, "OPTIMIZER": "SGD", "FP16": False, "FP16_LOSS_SCALE": 128.0}
DEFAULTS = {"vqa": {"RNG_SEED": 12345, "MODULE": "ResNetVLBERT", "LOG_FREQUENT": 100, "SCALES": (600, 1000), "MODEL_PREFIX": "",
                    "DATASET": {"ANSWER_VOCAB_SIZE": 3129}, "NETWORK": dict(_NET, IMAGE_FEAT_PRECOMPUTED=True), "TRAIN": _TRAIN},
            # vcr/function/config.py:113: the VCR model reads the pooled output
            "vcr": {"RNG_SEED": 12345, "MODULE": "ResNetVLBERT", "LOG_FREQUENT": 100, "SCALES": (600, 1000), "MODEL_PREFIX": "",
                    "DATASET": {"TASK": "Q2A"}, "NETWORK": dict(_NET, VLBERT=dict(_NET["VLBERT"], with_pooler=True)), "TRAIN": _TRAIN}}


def load_config(task, path):
    import copy
    import yaml
    cfg = copy.deepcopy(DEFAULTS[task])
    if path:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    return AttrDict.wrap(cfg)


def lr_lambda(config, steps_per_epoch):
    """Multiplier of the base lr at optimizer step k (0-based), as the reference's schedulers compute it."""
    tr = config.TRAIN
    accum = int(tr.GRAD_ACCUMULATE_STEPS)
    warm = int(tr.WARMUP_STEPS) if tr.WARMUP else 0
    if tr.LR_SCHEDULE == "triangle":      # WarmupLinearSchedule (common/nlp/bert/optimization.py:49-62)
        t_total = int(int(tr.END_EPOCH) * steps_per_epoch / accum)

        def f(k):
            if k < warm:
                return float(k) / float(max(1, warm))
            return max(0.0, float(t_total - k) / float(max(1.0, t_total - warm)))
        return f
    if tr.LR_SCHEDULE == "step":          # WarmupMultiStepLR (common/lr_scheduler.py:7-49)
        steps = tr.LR_STEP
        if isinstance(steps, str):
            steps = [float(x) for x in steps.split(",") if x.strip()]
        miles = sorted(int(e * steps_per_epoch / accum) for e in steps)
        gamma, wf = float(tr.LR_FACTOR), float(tr.WARMUP_FACTOR)

        def f(k):
            w = 1.0
            if k < warm:
                if tr.WARMUP_METHOD == "constant":
                    w = wf
                else:
                    a = float(k) / max(1, warm)
                    w = wf * (1 - a) + a
            return w * gamma ** sum(1 for m in miles if m <= k)
        return f
    raise NotImplementedError("TRAIN.LR_SCHEDULE %s (supported: triangle, step)" % tr.LR_SCHEDULE)


def parse_args(task, argv=None):
    ap = argparse.ArgumentParser("Train Cognition Network (%s) on the MI355X module mirror" % task)
    ap.add_argument("--cfg", type=str, help="path to a reference config file (cfgs/%s/*.yaml)" % task)
    ap.add_argument("--model-dir", type=str, help="accepted (epoch checkpoints of the mirrors: torch.save(state_dict) at the end of the run)")
    ap.add_argument("--log-dir", type=str, help="accepted for command-line compatibility")
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--slurm", action="store_true")
    ap.add_argument("--do-test", action="store_true", help="accepted; test-set csv writing is data-side and not built")
    ap.add_argument("--cudnn-off", action="store_true")
    ap.add_argument("--partial-pretrain", type=str, help="checkpoint whose matching keys are loaded (smart_partial_load_model_state_dict)")
    ap.add_argument("--steps", type=int, default=5, help="optimizer steps to run on synthetic batches")
    ap.add_argument("--steps-per-epoch", type=int, default=10000, help="stands in for len(train_loader) in the LR schedule")
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp16", "fp32", "cfg"], help="as pretrain/train_end2end: cfg = what the "
                    "YAML names (TRAIN.FP16 true -> the fp16 build + FP16_LOSS_SCALE, false -> fp32 encoder)")
    ap.add_argument("--dry-run", action="store_true", help="resolve and print the configuration, touch no GPU")
    return ap.parse_args(argv)


def resolve(task, config, world, args):
    tr = config.TRAIN
    B, accum = int(tr.BATCH_IMAGES), int(tr.GRAD_ACCUMULATE_STEPS)
    compute = (("fp16" if tr.FP16 else "fp32") if args.compute == "cfg" else args.compute)
    if tr.OPTIMIZER not in ("AdamW", "SGD"):
        raise NotImplementedError("TRAIN.OPTIMIZER %s (supported: AdamW, SGD)" % tr.OPTIMIZER)
    return dict(task=task, module=config.MODULE, per_gpu_batch=B, accumulate=accum, world=world, lr=float(tr.LR) * world * B * accum,
                optimizer=tr.OPTIMIZER, momentum=float(tr.MOMENTUM), weight_decay=float(tr.WD), clip_grad_norm=float(tr.CLIP_GRAD_NORM),
                lr_schedule=tr.LR_SCHEDULE, warmup_steps=int(tr.WARMUP_STEPS) if tr.WARMUP else 0, compute=compute,
                loss_scale=float(tr.FP16_LOSS_SCALE) if isinstance(tr.FP16_LOSS_SCALE, (int, float)) else 128.0,
                precomputed=bool(config.NETWORK.IMAGE_FEAT_PRECOMPUTED), seed=int(config.RNG_SEED))


def main(task, argv=None):
    args = parse_args(task, argv)
    config = load_config(task, args.cfg)
    if args.slurm:
        os.environ.setdefault("RANK", os.environ.get("SLURM_PROCID", "0"))
        os.environ.setdefault("WORLD_SIZE", os.environ.get("SLURM_NTASKS", "1"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.dist else 1
    rank = int(os.environ.get("RANK", "0")) if args.dist else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.dist else 0
    r = resolve(task, config, world, args)
    if args.dry_run:
        if rank == 0:
            print(json.dumps({"resolved": r, "NETWORK.VLBERT": dict(config.NETWORK.VLBERT)}, indent=1, default=str))
        return r
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (one process per GPU: the host driver only supports dmabuf IPC)
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("%s/train_end2end: no GPU visible.  The MI355X mirrors have no CPU execution path (--dry-run checks a "
                           "configuration without a GPU)" % task)
    pkg = __package__.rsplit(".", 1)[0]
    if r["compute"] != "bf16":
        importlib.import_module(pkg + "._lib").set_precision("f16")
    os.environ["VLB_ENCODER_FP32"] = "1" if r["compute"] == "fp32" else "0"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    syn = importlib.import_module(pkg + ".synthetic")
    OPT = importlib.import_module(pkg + ".optim")
    M = importlib.import_module(pkg + ".%s.modules.resnet_vlbert_for_%s" % (task, task))
    torch.manual_seed(r["seed"])
    net = getattr(M, config.MODULE)(config, device=dev)
    # NETWORK.PARTIAL_PRETRAIN (+ _PREFIX_CHANGES, LOAD_REL_HEAD, PARTIAL_PRETRAIN_SEGMB_INIT) -- vqa/function/train.py:198-214,
    # vcr/function/train.py:200-232; --partial-pretrain overrides the path.  A configured file that does not exist is reported and
    # skipped (the shipped YAMLs name ./model/pretrained_model/...: the reference would stop there; the bench-style runs start from random weights)
    ckpt_path = args.partial_pretrain or str(config.NETWORK.get("PARTIAL_PRETRAIN", "") or "")
    if ckpt_path and not os.path.isfile(ckpt_path):
        if args.partial_pretrain:
            raise FileNotFoundError(ckpt_path)
        if rank == 0:
            print("[Partial Load] NETWORK.PARTIAL_PRETRAIN %s not found: starting from the module's own initialisation" % ckpt_path, flush=True)
        ckpt_path = ""
    if ckpt_path:
        C = importlib.import_module(pkg + ".common.checkpoint")
        sd = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        sd = sd.get("state_dict", sd)
        sd = C.partial_pretrain_state_dict(sd, config.NETWORK.get("PARTIAL_PRETRAIN_PREFIX_CHANGES", []) or [],
                                           load_rel_head=bool(config.NETWORK.get("LOAD_REL_HEAD", False)),
                                           segmb_init=bool(config.NETWORK.get("PARTIAL_PRETRAIN_SEGMB_INIT", False)))
        own = net.state_dict()      # (shape mismatches -- another answer vocabulary -- are left to the module's initialisation, and SAID)
        sd, _ = C.drop_shape_mismatches(sd, own, log=(print if rank == 0 else (lambda *a: None)))
        C.smart_partial_load(net, sd, log=(print if rank == 0 else (lambda *a: None)))      # (load_state_dict copies across devices)
    net.train()
    if world > 1:
        net = importlib.import_module(pkg + ".parallel").DistributedDataParallel(net)
    if r["optimizer"] == "AdamW":
        opt = OPT.FusedAdamW(net.parameters(), lr=r["lr"], betas=(0.9, 0.999), eps=1e-6, weight_decay=r["weight_decay"])
    else:
        opt = OPT.FusedSGD(net.parameters(), lr=r["lr"], momentum=r["momentum"], weight_decay=r["weight_decay"])
    f = lr_lambda(config, args.steps_per_epoch)
    B, accum = r["per_gpu_batch"], r["accumulate"]
    scale = r["loss_scale"] if r["compute"] == "fp16" else 1.0
    Hi, Wi = int(config.SCALES[0]), int(config.SCALES[1])

    def batch(i):
        seed = 1000 * rank + i
        if task == "vqa":
            return syn.make_vqa_batch(B, 100, 124, seed, dev, answers=int(config.DATASET.ANSWER_VOCAB_SIZE))
        return syn.make_vcr_batch(B, 4, 55, 80, 117, Hi, Wi, seed, dev)
    if rank == 0:
        print("%s/train_end2end: %s | %d GPU(s) x batch %d x accumulate %d | %s lr %.3e wd %.1e clip %.1f | schedule %s warmup %d | compute %s" %
              (task, config.MODULE, world, B, accum, r["optimizer"], r["lr"], r["weight_decay"], r["clip_grad_norm"], r["lr_schedule"],
               r["warmup_steps"], r["compute"]), flush=True)
    t0, last = time.time(), None
    for step in range(args.steps):
        for g in opt.param_groups:
            g["lr"] = r["lr"] * f(step)
        opt.zero_grad(set_to_none=False)
        for micro in range(accum):
            b = batch(step * accum + micro)
            if task == "vqa":
                boxes, im_info, question, label = b
                outputs, loss = net(None, boxes, im_info, question, label)
            else:
                image, boxes, masks, question, answers, label, im_info = b
                outputs, loss = net(image, boxes, masks, question, None, answers, None, label, im_info)
            (loss * (scale / accum)).backward()
            last = loss.detach()
        if r["clip_grad_norm"] > 0:
            OPT.clip_grad_norm_(net.parameters(), r["clip_grad_norm"], opt, grad_scale=(1.0 / scale) if scale != 1.0 else None)
        elif scale != 1.0:
            opt.grad_scale = 1.0 / scale
        opt.step()
        if rank == 0 and ((step + 1) % max(1, min(int(config.LOG_FREQUENT), args.steps)) == 0 or step + 1 == args.steps):
            print("step %d  lr %.3e  loss %.4f  %.1f samples/s" % (step + 1, opt.param_groups[0]["lr"], float(last),
                                                                   (step + 1) * B * accum * world / (time.time() - t0)), flush=True)
    if args.model_dir and rank == 0:
        os.makedirs(args.model_dir, exist_ok=True)
        path = os.path.join(args.model_dir, "%s-%04d.model" % (config.get("MODEL_PREFIX", "") or ("vl-bert_" + task), 0))
        core = net.module if hasattr(net, "module") else net
        torch.save({"state_dict": {k: v.detach().cpu() for k, v in core.state_dict().items()}, "optimizer": opt.state_dict()}, path)
        print("checkpoint %s" % path, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return net, opt, float(last)
