"""CPU (gloo, world_size 2) tests of the data-parallel gradient exchange: bucket tiling, SUM semantics,
and 1/world folded into the optimizer scale -- the multi-GPU path is RCCL with the same torch.distributed
calls (backend "nccl")."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout():
    E = importlib.import_module("vl-bert_amd.engine")
    cfg = E.ModelConfig(hidden_size=64, num_hidden_layers=5, num_attention_heads=1, intermediate_size=128, vocab_size=300,
                        max_position_embeddings=32, visual_region_classes=20)
    shapes = E.param_layout(cfg)
    offsets, off = {}, 0
    import math
    for n, s in shapes.items():
        offsets[n] = off
        off = E._ru(off + math.prod(s), 64)
    return cfg, offsets, off


def test_buckets_tile_the_flat_buffer():
    P = importlib.import_module("vl-bert_amd.parallel")
    cfg, offsets, numel = _layout()
    for bucket_bytes in (1, 200_000, 1 << 30):
        b = P.GradBuckets(torch.zeros(numel), offsets, numel, cfg.num_hidden_layers, bucket_bytes=bucket_bytes)
        cov = b.coverage()
        assert cov[0][0] == 0 and cov[-1][1] == numel
        for (lo, hi), (lo2, hi2) in zip(cov[:-1], cov[1:]):
            assert hi == lo2 and lo < hi
        # every layer belongs to exactly one bucket that fires at-or-after its own completion
        fired = sorted(b.layer_bucket)
        assert fired[0] == 0


def test_will_launch_predicate_matches_on_done():
    """engine.backward joins its side stream (and calls the hook) only for the events that start a collective: the predicate must be
    true exactly where on_done() launches -- bucket boundaries, heads, word_emb / embed -- once per step, and never at world size 1."""
    P = importlib.import_module("vl-bert_amd.parallel")
    cfg, offsets, numel = _layout()
    b = P.GradBuckets(torch.zeros(numel), offsets, numel, cfg.num_hidden_layers, bucket_bytes=200_000)
    assert not any(b.will_launch(w) for w in list(range(cfg.num_hidden_layers)) + ["heads", "embed", "word_emb"])      # world 1
    b.world = 2
    launched = []
    b._launch = lambda lo, hi: launched.append((lo, hi))
    for what in ["heads"] + list(reversed(range(cfg.num_hidden_layers))) + ["word_emb", "embed"]:
        before = len(launched)
        expect = b.will_launch(what)
        b.on_done(what)
        assert (len(launched) > before) == expect, what
        assert not b.will_launch(what)                    # once per step
    cov = sorted(launched)
    assert cov[0][0] == 0 and cov[-1][1] == numel and all(a[1] == c[0] for a, c in zip(cov[:-1], cov[1:]))
    b.wait()
    assert b.will_launch("heads")                         # re-armed


def test_buckets_with_vision_tail_tile_the_flat_buffer():
    """e2e layout: the trainable convolution weights sit behind the heads and form their own (last) bucket."""
    E = importlib.import_module("vl-bert_amd.engine")
    P = importlib.import_module("vl-bert_amd.parallel")
    import math
    cfg = E.ModelConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128, vocab_size=300,
                        max_position_embeddings=32, visual_region_classes=20, e2e=True, image_num_layers=50)
    shapes = E.param_layout(cfg)
    offsets, off = {}, 0
    for n, s in shapes.items():
        offsets[n] = off
        off = E._ru(off + math.prod(s), 64)
    vis = [n for n in shapes if n.startswith("image_feature_extractor.") and "obj_downsample" not in n]
    assert len(vis) == 42 and list(shapes)[-len(vis):] == vis            # appended LAST, contiguous
    vstart = min(offsets[n] for n in vis)
    b = P.GradBuckets(torch.zeros(off), offsets, off, cfg.num_hidden_layers, bucket_bytes=1 << 20, vision_start=vstart)
    cov = b.coverage()
    assert cov[0][0] == 0 and cov[-1][1] == off and b.vision_keys == ["vision2", "vision3", "vision4"]
    assert b.ranges["vision2"][0] == vstart and b.ranges["vision4"][1] == off
    for (lo, hi), (lo2, hi2) in zip(cov[:-1], cov[1:]):
        assert hi == lo2 and lo < hi
    launched = []
    b._launch = lambda lo, hi: launched.append((lo, hi))
    # the engine's order in e2e mode: heads, layers, embed (before the CNN backward), RoI head, layer3, layer2, then the catch-alls
    for what in ("heads", 1, 0, "word_emb", "embed", "vision4", "vision3", "vision2", "embed", "vision"):
        b.on_done(what)
    assert sorted(launched) == sorted(cov) and len(launched) == len(cov)          # every range exactly once
    assert launched[-3:] == [b.ranges["vision4"], b.ranges["vision3"], b.ranges["vision2"]]
    b.wait()
    b.on_done("vision")
    assert len(launched) == len(cov) + 3                                            # re-armed for the next step


def _worker(rank, world, port, numel_holder):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module("vl-bert_amd.parallel")
    cfg, offsets, numel = _layout()
    g = torch.Generator().manual_seed(rank)
    grad = torch.randn(numel, generator=g)
    mine = grad.clone()
    b = P.GradBuckets(grad, offsets, numel, cfg.num_hidden_layers, bucket_bytes=300_000, wire_dtype=None)
    assert b.reduced is grad
    # replay the engine's completion order
    b.on_done("heads")
    for l in reversed(range(cfg.num_hidden_layers)):
        b.on_done(l)
    b.on_done("embed")
    b.wait()
    others = [torch.randn(numel, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    expect = sum(others)
    assert torch.allclose(grad, expect, atol=1e-5), (grad - expect).abs().max()
    assert abs(b.grad_scale - 1.0 / world) < 1e-12
    # bf16 wire format (the default): the reduced gradient stays in the persistent bf16 image the optimizer reads; the word-embedding
    # table goes as its own bucket in front of the rest of the front end
    grad2 = mine.clone()
    b2 = P.GradBuckets(grad2, offsets, numel, cfg.num_hidden_layers)
    assert b2.wire_dtype == torch.bfloat16 and b2.reduced.dtype == torch.bfloat16 and b2.reduced.numel() == numel
    launched = []
    orig = b2._launch
    b2._launch = lambda lo, hi: (launched.append((lo, hi)), orig(lo, hi))[1]
    for what in ["heads"] + list(reversed(range(cfg.num_hidden_layers))) + ["word_emb", "embed"]:
        b2.on_done(what)
    b2.wait()
    assert launched[-2] == b2.ranges["word_emb"] and launched[-1] == b2.ranges["embed"] and sorted(launched) == sorted(b2.coverage())
    assert torch.equal(grad2, mine)                                    # the local fp32 gradient is left untouched
    assert torch.allclose(b2.reduced.float(), expect, atol=0.05, rtol=0.02)
    same = b2.reduced.clone()
    dist.broadcast(same, src=0)
    assert torch.equal(same, b2.reduced)                               # every rank holds the same reduced bits
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, None), nprocs=2, join=True)
