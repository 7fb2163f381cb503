"""Entry point with the reference's command line (pretrain/train_end2end.py:11-48) over the HIP engine:

    python -m vl-bert_amd.pretrain.train_end2end --cfg cfgs/pretrain/base_prec_withouttextonly_4x16G_fp32.yaml [--dist]
           [--steps N] [--steps-per-epoch M] [--batch-images B] [--data] [--dry-run]

It reads the reference's YAML files as they are (the keys of pretrain/function/config.py that the hot path consumes: NETWORK.*,
TRAIN.{BATCH_IMAGES, LR, WD, CLIP_GRAD_NORM, LR_SCHEDULE, WARMUP, WARMUP_STEPS, END_EPOCH, BEGIN_EPOCH, GRAD_ACCUMULATE_STEPS},
SCALES, RNG_SEED, LOG_FREQUENT, MODULE) and runs the fused training step of vl-bert_amd/engine.py with the reference's
hyper-parameter conventions:
  * lr = TRAIN.LR x world_size x BATCH_IMAGES x GRAD_ACCUMULATE_STEPS               (pretrain/function/train.py:133-138)
  * 'triangle' = WarmupLinearSchedule(WARMUP_STEPS, t_total = END_EPOCH x steps/epoch / accumulate)     (:316-320)
  * clip_grad_norm_(CLIP_GRAD_NORM), AdamW(betas 0.9/0.999, eps 1e-6, WD, bias-corrected)               (:146-160)
  * one process per GPU with --dist (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher; backend nccl = RCCL).
Checkpoints: with --model-dir the reference's epoch files `{prefix}-{epoch:04d}.model` are written and TRAIN.RESUME / AUTO_RESUME
honoured (vl-bert_amd/common/checkpoint.py).  Batches are synthetic by default (the collated layout of pretrain/data/collate_batch.py);
with --data they come from the data sets the YAML names, read by vl-bert_amd/pretrain/data (the reference's annotation / detector-record /
corpus formats, its sampling and collation).  There is no CPU execution path: without a GPU
the program stops with an error unless --dry-run is given, which only resolves and prints the configuration (the "plumbing"
check of the reference's scripts/nondist_run.sh case).
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
        if isinstance(x, (list, tuple)):
            return type(x)(AttrDict.wrap(v) for v in x)
        return x


DEFAULTS = {
    "RNG_SEED": 12345, "MODULE": "ResNetVLBERTForPretraining", "LOG_FREQUENT": 100, "SCALES": (600, 1000),
    "OUTPUT_PATH": "", "MODEL_PREFIX": "", "CHECKPOINT_FREQUENT": 1,
    "NETWORK": {"IMAGE_FEAT_PRECOMPUTED": True, "IMAGE_NUM_LAYERS": 101, "IMAGE_C5_DILATED": True, "IMAGE_STRIDE_IN_1x1": True,
                "IMAGE_FROZEN_BACKBONE_STAGES": [1, 2], "IMAGE_FROZEN_BN": True, "IMAGE_SEMANTIC": False, "OUTPUT_CONV5": False,
                "WITH_REL_LOSS": False, "WITH_MLM_LOSS": True, "WITH_MVRC_LOSS": True, "VLBERT": {}},
    "TRAIN": {"BATCH_IMAGES": 64, "LR": 1.0e-7, "WD": 1.0e-4, "CLIP_GRAD_NORM": 10, "LR_SCHEDULE": "triangle", "WARMUP": True,
              "WARMUP_STEPS": 8000, "BEGIN_EPOCH": 0, "END_EPOCH": 10, "GRAD_ACCUMULATE_STEPS": 1, "OPTIMIZER": "AdamW", "FP16": False,
              "RESUME": False, "AUTO_RESUME": True},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(path):
    """update_config (pretrain/function/config.py:134-163): YAML over defaults, nested keys merged."""
    import copy
    import yaml
    cfg = copy.deepcopy(DEFAULTS)
    if path:
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    return AttrDict.wrap(cfg)


def resolve(config, world, args):
    """The numbers the step needs, with the reference's conventions."""
    tr = config.TRAIN
    bi = tr.BATCH_IMAGES
    # multitask cfgs: BATCH_IMAGES = [caption batch, text-only batch, ...] per GPU (MultiTaskDataLoader,
    # common/utils/multi_task_dataloader.py:17-57); the LR scales with their SUM (pretrain/function/train.py:133-138)
    if isinstance(bi, (list, tuple)):
        per_gpu, per_gpu_aux = int(bi[0]), int(sum(bi[1:]))
    else:
        per_gpu, per_gpu_aux = int(bi), 0
    if args.batch_images:
        per_gpu_aux = args.batch_images if per_gpu_aux else 0
        per_gpu = args.batch_images
    accum = int(tr.GRAD_ACCUMULATE_STEPS)
    steps_per_epoch = args.steps_per_epoch or 10000
    if tr.OPTIMIZER != "AdamW":
        raise NotImplementedError("TRAIN.OPTIMIZER %s: the fused step implements AdamW (every shipped pretrain cfg)" % tr.OPTIMIZER)
    sched = {"triangle": "triangle", "constant": "constant"}.get(tr.LR_SCHEDULE)
    if sched is None:
        raise NotImplementedError("TRAIN.LR_SCHEDULE %s (supported: triangle, constant)" % tr.LR_SCHEDULE)
    multitask = config.MODULE == "ResNetVLBERTForPretrainingMultitask"
    if config.MODULE not in ("ResNetVLBERTForPretraining", "ResNetVLBERTForPretrainingMultitask"):
        raise NotImplementedError("MODULE %s" % config.MODULE)
    if not multitask:
        per_gpu, per_gpu_aux = per_gpu + per_gpu_aux, 0
    total = per_gpu + per_gpu_aux
    return dict(per_gpu_batch=per_gpu, per_gpu_aux_batch=per_gpu_aux, world=world, global_batch=total * world, accumulate=accum,
                lr=float(tr.LR) * world * total * accum, weight_decay=float(tr.WD), clip_grad_norm=float(tr.CLIP_GRAD_NORM),
                lr_schedule=sched, warmup_steps=int(tr.WARMUP_STEPS) if tr.WARMUP else 0,
                t_total=int(int(tr.END_EPOCH) * steps_per_epoch / accum), steps_per_epoch=steps_per_epoch,
                e2e=not config.NETWORK.IMAGE_FEAT_PRECOMPUTED, image_size=tuple(config.SCALES), multitask=multitask,
                fp16_requested=bool(tr.FP16), seed=int(config.RNG_SEED),
                compute=(("fp16" if tr.FP16 else "fp32") if args.compute == "cfg" else args.compute),
                loss_scale=(float(tr.FP16_LOSS_SCALE) if isinstance(tr.get("FP16_LOSS_SCALE", None), (int, float)) else None))


def parse_args(argv=None):
    ap = argparse.ArgumentParser("Train VL-BERT on the MI355X engine")
    ap.add_argument("--cfg", type=str, help="path to a reference-style config file (cfgs/pretrain/*.yaml)")
    ap.add_argument("--model-dir", type=str, help="root of the checkpoint directory (pretrain/train_end2end.py:21,39-40): epoch files "
                    "`<model-dir>/<OUTPUT_PATH>/<cfg name>/<image set>_train/<MODEL_PREFIX>-<epoch:04d>.model` in the reference's format are written every "
                    "CHECKPOINT_FREQUENT epochs (an epoch = --steps-per-epoch optimizer steps) and TRAIN.RESUME / TRAIN.AUTO_RESUME are "
                    "honoured (vl-bert_amd/common/checkpoint.py).  Without it nothing is written")
    ap.add_argument("--log-dir", type=str, help="accepted for command-line compatibility")
    ap.add_argument("--dist", action="store_true", help="one process per GPU (torch.distributed.run / SLURM environment)")
    ap.add_argument("--slurm", action="store_true")
    ap.add_argument("--do-test", action="store_true")
    ap.add_argument("--cudnn-off", action="store_true", help="accepted and ignored (no cuDNN / MIOpen on this path)")
    ap.add_argument("--steps", type=int, default=20, help="optimizer steps to run on synthetic batches")
    ap.add_argument("--steps-per-epoch", type=int, default=0, help="stands in for len(train_loader) in the LR schedule (default: 10000 "
                    "with synthetic batches, len(loader) with --data)")
    ap.add_argument("--batch-images", type=int, default=0, help="override TRAIN.BATCH_IMAGES (per GPU)")
    ap.add_argument("--text-len", type=int, default=0, help="text capacity of the step's static buffers (default 64; DATASET.SEQ_LEN with --data)")
    ap.add_argument("--regions", type=int, default=0, help="box capacity (default 36; DATASET.SEQ_LEN with --data)")
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp16", "fp32", "cfg"],
                    help="arithmetic of the step.  bf16 (default): bf16 operands, fp32 accumulation / master weights, no loss scaling.  "
                         "fp16: the fp16 build of the library + static loss scale TRAIN.FP16_LOSS_SCALE (the reference's Apex mode, "
                         "pretrain/function/train.py:345-352; 'dynamic' -> 4096).  fp32: every encoder tensor in fp32 (encoder_f32.py), fp16 "
                         "embedding / head kernels around it.  cfg: what the YAML names -- TRAIN.FP16 true -> fp16, false -> fp32")
    ap.add_argument("--data", action="store_true",
                    help="read the data sets the YAML names (DATASET.*: annotation jsonl, detector records with base64 boxes / class scores / "
                         "features, images, text corpora -- vl-bert_amd/pretrain/data) instead of synthetic batches; steps per epoch = "
                         "len(loader) unless --steps-per-epoch is given; the engine's text / box capacities default to DATASET.SEQ_LEN")
    ap.add_argument("--dry-run", action="store_true", help="resolve and print the configuration, touch no GPU")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    config = load_config(args.cfg)
    if args.slurm:
        os.environ.setdefault("RANK", os.environ.get("SLURM_PROCID", "0"))
        os.environ.setdefault("WORLD_SIZE", os.environ.get("SLURM_NTASKS", "1"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.dist else 1
    rank = int(os.environ.get("RANK", "0")) if args.dist else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.dist else 0
    loader = None
    if args.data:        # the reference's make_dataloader(s) + MultiTaskDataLoader (pretrain/function/train.py:60-66,113-118), one process per GPU
        data = importlib.import_module(__package__ + ".data")
        if "DATASET" not in config:
            raise ValueError("--data: the configuration has no DATASET section")
        kw = dict(mode="train", distributed=world > 1, num_replicas=world, rank=rank, drop_last=True)      # (static batch: the tail is dropped)
        if isinstance(config.DATASET, (list, tuple)):
            if args.batch_images:
                raise ValueError("--batch-images with a multitask DATASET list: edit TRAIN.BATCH_IMAGES instead")
            loaders = data.make_dataloaders(config, **kw)
            loader = data.MultiTaskDataLoader(loaders) if len(loaders) > 1 else loaders[0]
            seq_len = max(int(d.get("SEQ_LEN", 64)) for d in config.DATASET)
        else:
            if args.batch_images:
                config.TRAIN["BATCH_IMAGES"] = args.batch_images
            loader = data.make_dataloader(config, **kw)
            seq_len = int(config.DATASET.get("SEQ_LEN", 64))
        if len(loader) == 0:
            raise ValueError("--data: fewer samples than one batch per rank")
        args.steps_per_epoch = args.steps_per_epoch or len(loader)
        args.text_len, args.regions = args.text_len or seq_len, args.regions or seq_len
    args.text_len, args.regions = args.text_len or 64, args.regions or 36
    r = resolve(config, world, args)
    if args.dry_run:
        if rank == 0:
            print(json.dumps({"resolved": r, "NETWORK.VLBERT": dict(config.NETWORK.VLBERT)}, indent=1, default=str))
        return r
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (one process per GPU: the host driver only supports dmabuf IPC)
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("train_end2end: no GPU visible.  The MI355X engine has no CPU execution path (use --dry-run to check a "
                           "configuration without a GPU)")
    pkg = __package__.rsplit(".", 1)[0]
    if r["compute"] != "bf16":      # fp16 and fp32 modes run on the fp16 build of the library (vl-bert_amd/_lib.py: one build per process)
        importlib.import_module(pkg + "._lib").set_precision("f16")
    engine = importlib.import_module(pkg + ".engine")
    syn = importlib.import_module(pkg + ".synthetic")
    ops = importlib.import_module(pkg + ".ops")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    vl = config.NETWORK.VLBERT
    g = lambda k, d: vl[k] if k in vl else d
    mc = engine.ModelConfig(hidden_size=g("hidden_size", 768), num_hidden_layers=g("num_hidden_layers", 12),
                            num_attention_heads=g("num_attention_heads", 12), intermediate_size=g("intermediate_size", 3072),
                            vocab_size=g("vocab_size", 30522), max_position_embeddings=g("max_position_embeddings", 512),
                            type_vocab_size=g("type_vocab_size", 3), visual_region_classes=g("visual_region_classes", 1601),
                            hidden_dropout_prob=g("hidden_dropout_prob", 0.1),
                            attention_probs_dropout_prob=g("attention_probs_dropout_prob", 0.1), multitask=r["multitask"],
                            with_pooler=bool(g("with_pooler", False)), with_rel_loss=bool(config.NETWORK.WITH_REL_LOSS), e2e=r["e2e"],
                            image_num_layers=int(config.NETWORK.IMAGE_NUM_LAYERS),
                            image_frozen_stages=tuple(config.NETWORK.IMAGE_FROZEN_BACKBONE_STAGES))
    B, T, R = r["per_gpu_batch"], args.text_len, args.regions
    B_aux = r["per_gpu_aux_batch"]
    eng = engine.PretrainEngine(mc, B, T, R, device="cuda:%d" % local_rank, train=True, lr=r["lr"], weight_decay=r["weight_decay"],
                                max_grad_norm=r["clip_grad_norm"], seed=r["seed"] + rank, B_aux=B_aux,
                                lr_schedule=r["lr_schedule"], warmup_steps=r["warmup_steps"], t_total=max(r["t_total"], r["warmup_steps"] + 1),
                                image_size=r["image_size"] if r["e2e"] else None, grad_accum=r["accumulate"],
                                encoder_fp32=r["compute"] == "fp32", dp_mode="allreduce" if r["compute"] == "fp32" else "default",
                                loss_scale=r["loss_scale"] if r["compute"] == "fp16" else None)
    eng.init_random(seed=r["seed"], visual_ln_init=float(g("visual_scale_object_init", 0.0)))
    # NETWORK.PARTIAL_PRETRAIN (+ _PREFIX_CHANGES): start from a pre-trained checkpoint's matching tensors (pretrain/function/train.py:297-313,
    # common/utils/load.py:57-81).  Empty in every shipped pre-training YAML; a configured file that does not exist is reported and skipped
    pp = str(config.NETWORK.get("PARTIAL_PRETRAIN", "") or "")
    if pp and not os.path.isfile(pp):
        if rank == 0:
            print("[Partial Load] NETWORK.PARTIAL_PRETRAIN %s not found: starting from the random initialisation" % pp, flush=True)
        pp = ""
    if pp:
        ckpt_mod = importlib.import_module(pkg + ".common.checkpoint")
        sd = torch.load(pp, map_location="cpu", weights_only=False)
        sd = ckpt_mod.partial_pretrain_state_dict(sd.get("state_dict", sd), config.NETWORK.get("PARTIAL_PRETRAIN_PREFIX_CHANGES", []) or [])
        own = eng.state_dict()
        sd, dropped = ckpt_mod.drop_shape_mismatches(sd, own, log=(print if rank == 0 else (lambda *a: None)))
        sd = {k: v.to(eng.dev) for k, v in sd.items()}
        ckpt_mod.smart_partial_load(eng, sd, log=(print if rank == 0 else (lambda *a: None)))
        eng.sync_weights()
    eng.broadcast_parameters(src=0)       # rank 0's parameters / optimizer state everywhere (pretrain/function/train.py:331-334)
    if rank == 0:
        print("train_end2end: %s | %d GPU(s) x batch %d | lr %.3e wd %.1e clip %.1f | schedule %s warmup %d t_total %d%s" %
              (config.MODULE, world, B + B_aux, r["lr"], r["weight_decay"], r["clip_grad_norm"], r["lr_schedule"], r["warmup_steps"], r["t_total"],
               " | compute %s%s" % (r["compute"], " (loss scale %g)" % eng.loss_scale if eng.loss_scale != 1.0 else "")), flush=True)
    # checkpoints in the reference's format + resume (common/callbacks/epoch_end_callbacks/checkpoint.py, common/utils/load.py:20-54)
    ckpt = importlib.import_module(pkg + ".common.checkpoint")
    spe = max(1, int(r["steps_per_epoch"] // r["accumulate"]))           # optimizer steps per epoch
    begin_epoch, prefix = int(config.TRAIN.BEGIN_EPOCH), None
    if args.model_dir:
        cfg_name = os.path.splitext(os.path.basename(args.cfg))[0] if args.cfg else "default"
        # the reference's layout (pretrain/train_end2end.py:26-27 + common/utils/create_logger.py:24-48 + pretrain/function/train.py:41-46):
        # <model_dir>/<OUTPUT_PATH>/<cfg name>/<TRAIN_IMAGE_SET of the first data set>_train/<MODEL_PREFIX>-NNNN.model, so that RESUME /
        # AUTO_RESUME pick up a run directory the reference wrote and the reverse (round-4 ADVICE: the <image set>_train level was
        # missing, and lstrip("./") strips CHARACTERS -- '../out' became 'out', '/abs/out' became 'abs/out')
        ds = config.get("DATASET", None)
        ds = ds[0] if isinstance(ds, (list, tuple)) and len(ds) else ds
        image_set = str(ds.get("TRAIN_IMAGE_SET", "train")) if ds is not None and hasattr(ds, "get") else "train"
        prefix = os.path.normpath(os.path.join(args.model_dir, str(config.OUTPUT_PATH), cfg_name, image_set + "_train",
                                               str(config.MODEL_PREFIX) or "vl-bert"))
        begin_epoch = ckpt.smart_resume(eng, prefix, begin_epoch, int(config.TRAIN.END_EPOCH), resume=bool(config.TRAIN.RESUME),
                                        auto_resume=bool(config.TRAIN.AUTO_RESUME), log=(print if rank == 0 else (lambda *a: None)))
        if begin_epoch > int(config.TRAIN.BEGIN_EPOCH) or config.TRAIN.RESUME:
            eng.broadcast_parameters(src=0)
    first_step = begin_epoch * spe
    t0, seen = time.time(), 0
    stream = {"it": None, "epoch": begin_epoch}

    def next_real_batch():
        while True:
            if stream["it"] is None:
                for ld in (getattr(loader, "loaders", None) or [loader])[:1]:      # the master loader's sampler follows the epoch (train.py:335)
                    smp = getattr(getattr(ld, "batch_sampler", None), "sampler", None)
                    if hasattr(smp, "set_epoch"):
                        smp.set_epoch(stream["epoch"])
                stream["it"] = iter(loader)
            try:
                return next(stream["it"])
            except StopIteration:
                stream["it"], stream["epoch"] = None, stream["epoch"] + 1

    def load_real_batch():
        batch = next_real_batch()
        image, boxes, im_info, text, rel, mlm, ops_, soft = batch[:8]
        if text.shape[1] > T or boxes.shape[1] > R:
            raise ValueError("batch of %d tokens / %d boxes exceeds the step's capacities (--text-len %d, --regions %d)" % (text.shape[1], boxes.shape[1], T, R))
        kw = {}
        if r["multitask"]:
            kw.update(aux_text=batch[8].cuda(non_blocking=True), aux_mlm_labels=batch[9].cuda(non_blocking=True))
        if r["e2e"]:
            Hi, Wi = r["image_size"]
            if image.shape[2] > Hi or image.shape[3] > Wi:
                raise ValueError("image batch %dx%d does not fit the step's static %dx%d image buffer (SCALES; portrait images need their own engine)"
                                 % (image.shape[2], image.shape[3], Hi, Wi))
            full = torch.zeros((image.shape[0], 3, Hi, Wi), dtype=torch.float32)
            full[:, :, :image.shape[2], :image.shape[3]] = image
            kw["image"] = full.cuda(non_blocking=True)      # (raw-pixel masking was done by the dataset, as in the reference)
        eng.set_batch(*[t.cuda(non_blocking=True) for t in (boxes, im_info, text, rel, mlm, ops_, soft)], **kw)

    def load_batch(seed_off):
        if loader is not None:
            return load_real_batch()
        batch = list(syn.make_batch(B, T, R, seed=1000 * rank + seed_off))
        kw = {}
        if r["multitask"]:
            aux_text, aux_lab = syn.make_aux_text(B_aux, T, seed=5000 * (rank + 1) + seed_off)
            kw.update(aux_text=aux_text.cuda(non_blocking=True), aux_mlm_labels=aux_lab.cuda(non_blocking=True))
        if r["e2e"]:
            Hi, Wi = r["image_size"]
            gi = torch.Generator().manual_seed(7000 * (rank + 1) + seed_off)
            kw["image"] = (torch.randn(B, 3, Hi, Wi, generator=gi) * 50.0).cuda(non_blocking=True)
            batch[0][:, :, 0].clamp_(0, Wi - 170)
            batch[0][:, :, 1].clamp_(0, Hi - 170)
            batch[0][:, :, 2] = torch.minimum(batch[0][:, :, 2], torch.full_like(batch[0][:, :, 2], Wi - 1.0))
            batch[0][:, :, 3] = torch.minimum(batch[0][:, :, 3], torch.full_like(batch[0][:, :, 3], Hi - 1.0))
            batch[1][:, 0], batch[1][:, 1] = Wi, Hi
            kw["mask_raw_pixels"] = bool(config.NETWORK.get("MASK_RAW_PIXELS", True))      # conceptual_captions.py:201-206
        eng.set_batch(*[t.cuda(non_blocking=True) for t in batch], **kw)

    accum = r["accumulate"]
    for step in range(first_step, first_step + args.steps):
        if accum == 1:
            load_batch(step)
            eng.train_step()
        else:      # common/trainer.py:117-153: loss / accumulate per micro-batch, exchange + optimizer on the boundary micro-step
            eng.zero_grad()
            for micro in range(accum):
                load_batch(step * accum + micro)
                eng.forward(True, gscale=1.0 / accum)
                last = micro == accum - 1
                eng.backward(True, on_layer_done=eng.buckets.on_done if (last and eng.buckets is not None) else None)
                if not last:
                    ops.rng_advance(eng.seed)      # fresh dropout masks per micro-batch (optimizer_step advances after the last one)
            if eng.buckets is not None:
                eng.buckets.wait()
            eng.optimizer_step()
        seen += (B + B_aux) * world * accum
        if prefix is not None and (step + 1) % spe == 0:          # epoch end: Checkpoint(model_prefix, CHECKPOINT_FREQUENT) on rank 0
            epoch = (step + 1) // spe - 1
            if (epoch + 1) % max(1, int(config.CHECKPOINT_FREQUENT)) == 0:
                path = ckpt.save_checkpoint(eng, prefix, epoch, rank=rank)       # (a collective with the sharded optimizer)
                if rank == 0:
                    print("epoch %d done: checkpoint %s" % (epoch, path), flush=True)
        if (step + 1 - first_step) % max(1, min(int(config.LOG_FREQUENT), args.steps)) == 0 or step + 1 == first_step + args.steps:
            lv = eng.loss_values()          # host sync, like the reference's Speedometer + metric readout
            if rank == 0:
                print("step %d  lr %.3e  loss %.4f (mlm %.4f mvrc %.4f)  %.1f samples/s" %
                      (step + 1, float(eng.adam[0]), lv["loss"], lv["mlm_loss"], lv["mvrc_loss"], seen / (time.time() - t0)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return eng


if __name__ == "__main__":
    main(sys.argv[1:])
