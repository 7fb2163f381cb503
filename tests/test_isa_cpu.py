"""Compile-outcome contracts of the hot non-GEMM kernels, checked on the ISA hipcc emits (no GPU: hipcc cross-compiles gfx950 here).

The kernels' speed depends on things the source does not guarantee: that all loads of a prologue / loop iteration are issued as ONE
burst with counted waits (a load under a lane condition becomes `branch + load + s_waitcnt vmcnt(0)`: ten dependent HBM round trips
in the attention backward before round 3's fix -- DESIGN.md §3, "Serialised prologue loads"), that the register count stays under the
occupancy step the launch geometry assumes, and that nothing spills.  tools/isa_lint.py extracts those facts; this test pins them."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)

pytestmark = pytest.mark.skipif(not L.available(), reason="hipcc not installed")
CSRC = os.path.join(ROOT, "vl-bert_amd", "csrc")


@pytest.fixture(scope="module")
def attention():
    return L.kernels(L.compile_isa(os.path.join(CSRC, "attention.hip")))


@pytest.fixture(scope="module")
def layernorm():
    return L.kernels(L.compile_isa(os.path.join(CSRC, "layernorm.hip")))


def test_attention_prologues_issue_their_loads_in_one_burst(attention):
    fwd = L.find(attention, r"attn_fwd_kernelILi4ELi1E")
    bwd2 = L.find(attention, r"attn_bwd2_kernel")
    bwd = L.find(attention, r"attn_bwd_kernelILi4ELi1E")
    # forward: K and V tiles (4 x 16 B per thread), the first Q fragment (2) and the mask row before the first wait
    assert L.loads_before_first_wait(fwd["body"]) >= 7
    # backward: Q, K, V, dO tiles (8), O for the row dots (2), mask and lse rows (2)
    assert L.loads_before_first_wait(bwd2["body"]) >= 12
    assert L.loads_before_first_wait(bwd["body"]) >= 12
    for k in (fwd, bwd2, bwd):
        assert L.serialized_loads(k["body"]) == 0
        assert k["spill"] == 0 and k["scratch"] == 0
    # the first wait of the backward is a COUNTED one (the oldest load of the burst, not the whole queue)
    body = bwd2["body"]
    first = next(l for l in body if "s_waitcnt" in l and "vmcnt(" in l)
    assert "vmcnt(0)" not in first, first


def test_attention_register_budgets(attention):
    # launch geometry: forward 4 workgroups of 8 waves per CU (<= 64 VGPRs), backward 2 (<= 128)
    assert L.find(attention, r"attn_fwd_kernelILi4ELi1E")["vgpr"] <= 64
    assert L.find(attention, r"attn_bwd2_kernel")["vgpr"] <= 128
    assert L.find(attention, r"attn_bwd_kernelILi4ELi1E")["vgpr"] <= 128
    assert L.find(attention, r"attn_fwd_kernelILi8ELi2E")["vgpr"] <= 128
    assert L.find(attention, r"attn_bwd_kernelILi8ELi2E")["vgpr"] <= 128


def test_layernorm_rows_are_loaded_in_one_burst(layernorm):
    fwd = L.find(layernorm, r"layernorm_fwd_kernelILi3ELi2E")        # H = 768, two rows per wave: 2 x three 8-B chunks per lane + gamma / beta
    assert L.loads_before_first_wait(fwd["body"]) >= 6 and L.serialized_loads(fwd["body"]) == 0
    assert fwd["vgpr"] <= 64 and fwd["spill"] == 0
    fwd4 = L.find(layernorm, r"layernorm_fwd_kernelILi4ELi2E")       # H = 1024
    assert L.serialized_loads(fwd4["body"]) == 0 and fwd4["vgpr"] <= 80
    bwd = L.find(layernorm, r"layernorm_bwd4_kernelILi3ELi1E")        # the encoder's backward at H = 768
    assert bwd["vgpr"] <= 128 and bwd["spill"] == 0 and bwd["scratch"] == 0       # 4 waves per SIMD: LN_MAX_BLOCKS = 4 workgroups per CU
    assert L.serialized_loads(bwd["body"]) == 0
    # inside the row loop: x (3), dy (3) and the row statistics are in flight together -- find the loop's first row load
    body = bwd["body"]
    loop = next(i for i, l in enumerate(body) if "global_load_dwordx2" in l)
    assert L.loads_before_first_wait(body, loop) >= 7
    bwd1024 = L.find(layernorm, r"layernorm_bwd4_kernelILi4ELi1E")    # H = 1024 (VL-BERT-large): 3 waves per SIMD, no spill
    assert bwd1024["vgpr"] <= 168 and bwd1024["spill"] == 0 and L.serialized_loads(bwd1024["body"]) == 0


def test_grouped_weight_gradient_main_loop_is_clean():
    ks = L.kernels(L.compile_isa(os.path.join(CSRC, "gemm_tn8.hip")))
    # descriptors in the kernel argument | in a device table (round 4)  x  16x16x32 | 32x32x16 matrix instructions (round 6)
    for inst, mfmas in ((r"gemm_tn8_kernelILb0ELb0E", 128), (r"gemm_tn8_kernelILb1ELb0E", 128), (r"gemm_tn8_kernelILb0ELb1E", 64),
                        (r"gemm_tn8_kernelILb1ELb1E", 64)):
        k = L.find(ks, inst)
        lo, hi = L.mfma_region(k["body"])
        inner = L.region_counts(k["body"], lo, hi)
        assert inner["mfma"] == mfmas and inner["scratch"] == 0     # two K tiles x four phases x 16 (8) MFMAs, no spill traffic in the loop
        assert inner["vmcnt0"] == 0                                  # the producer never stops (round 6): one form of the counted wait, no drain
        assert k["spill"] == 0 and k["vgpr"] <= 256 and k["scratch"] == 0, inst      # (round 6: a four-way switch over csum[] once put it in scratch)
        # round 6: no `live` flag in front of the load segments -- what is left are the item switch of the producer (taken once per
        # item) and the four column-sum turns; 37 with the flag
        assert inner["branches"] <= 28, (inst, inner["branches"])


def test_large_tile_forward_core_has_no_producer_branches():
    """gemm_nt_p8: the K loop of every instantiation stages its half-images unconditionally (round 6) and retires them with ONE counted
    wait form: no vmcnt(0) between the first and the last MFMA of the two-K-tile loop body except in the epilogues that follow it."""
    ks = L.kernels(L.compile_isa(os.path.join(CSRC, "gemm_p8.hip")))
    seen = 0
    for name, k in ks.items():
        if "gemm_nt_p8_kernel" not in name:
            continue
        body = k["body"]
        first = next(i for i, l in enumerate(body) if "v_mfma_" in l)
        # the loop body = up to the 2 x 4 x 16 (12, 20) MFMAs after the first one
        per_tile = 8 * {"ILi3E": 12, "ILi4E": 16, "ILi5E": 20}[re.search(r"kernel(ILi\dE)", name).group(1)]
        idx = [i for i, l in enumerate(body) if "v_mfma_" in l][:per_tile]
        inner = L.region_counts(body, first, idx[-1])
        assert inner["mfma"] == per_tile and inner["lds_dma"] >= 12, (name, inner)
        assert inner["vmcnt0"] == 0, (name, inner)
        seen += 1
    assert seen >= 30


def test_no_instruction_touches_an_lds_read_still_in_flight():
    """The GEMM / attention cores issue their fragment reads from inline asm and retire them with explicit lgkmcnt waits: hipcc does not know
    the destination registers are still in flight.  tools/isa_lint.landing_register_violations replays every kernel's instruction stream
    (LDS reads complete in issue order; a counted wait retires all but the N youngest) and reports any instruction that reads or writes a
    register whose read has not been waited for -- the class of defect found (on global loads) in round 4's LayerNorm experiment."""
    synthetic = ["\t;;#ASMSTART", "\tds_read_b128 v[10:13], v1", "\t;;#ASMEND", "\t;;#ASMSTART", "\tds_read_b128 v[14:17], v1 offset:64", "\t;;#ASMEND",
                 "\tv_add_f32_e32 v20, v21, v22", "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(1)", "\t;;#ASMEND", "\tv_mov_b32_e32 v30, v11",
                 "\tv_mov_b32_e32 v31, v15", "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(0)", "\t;;#ASMEND", "\tv_mov_b32_e32 v32, v15"]
    bad = L.landing_register_violations(synthetic)
    assert [(i, r) for i, _, r in bad] == [(11, 15)]          # v11 was released by the counted wait, v15 was not
    total = 0
    for src in ("gemm_p8.hip", "gemm_tn8.hip", "gemm.hip", "attention.hip"):
        ks = L.kernels(L.compile_isa(os.path.join(CSRC, src)))
        assert ks, src
        for name, k in ks.items():
            v = L.landing_register_violations(k["body"])
            assert not v, (src, name, v[:3])
            total += 1
    assert total >= 100
