"""ctypes binding of the C-ABI library `libvlbert_hip.so` (include/vlbert_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError
is raised (the reference's ops raise RuntimeError through AT_ASSERTM / AT_ERROR,
common/lib/roi_pooling/cuda/ROIAlign_cuda.cu:263-264,297).  The product path never routes
through PyTorch eager kernels or the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The 16-bit type is a build-time choice of the library (csrc/vlb_common.h): "bf16" -> libvlbert_hip.so (default), "f16" ->
# libvlbert_hip_f16.so (IEEE fp16 activations / working weights / gradients + a static loss scale: the reference's Apex fp16 mode).
# One precision per process: VLB_PRECISION at start-up, or set_precision() before the first engine / module is built.
PRECISION = os.environ.get("VLB_PRECISION", "bf16").lower()
if PRECISION in ("fp16", "half", "float16"):
    PRECISION = "f16"
if PRECISION not in ("bf16", "f16"):
    raise ValueError("VLB_PRECISION must be bf16 or f16 (got %r)" % PRECISION)


def _default_path(precision):
    return os.path.join(_HERE, "libvlbert_hip.so" if precision == "bf16" else "libvlbert_hip_f16.so")


# VLB_LIB_PATH: an alternative build of the same library (A/B measurements of compile-time variants on one GPU box)
LIB_PATH = os.environ.get("VLB_LIB_PATH") or _default_path(PRECISION)

_P, _L, _I, _F, _U = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_uint32

# signature strings: p pointer | l long | i int | f float | u uint32 | s stream
_SIGS = {
    "vlb_gemm_nt_bf16": "plplpliiipiplplplfpuiis",
    "vlb_transpose_batched_bf16": "ppiis",
    "vlb_gemm_nt_bf16_splitk": "plplpliiipls",
    "vlb_transpose_bf16": "plpliips",
    "vlb_wgrad_nt_bf16": "plplpliiipls",
    "vlb_wgrad_tn_bf16": "plplpliiipplis",
    "vlb_wgrad_tn_group_bf16": "ippppppipppplis",
    "vlb_wgrad_tn_table_launch": "piis",
    "vlb_tn8_set_stamps": "p",
    "vlb_wgrad_tn_table_pack": "i" + "p" * 11 + "ipl",      # (returns the item count: called directly, not through call())
    "vlb_zero_ranges_f32": "pppiis",
    "vlb_copy_ranges_f32": "ppppiis",
    "vlb_layernorm_fwd": "plppplpiifis",
    "vlb_layernorm_bwd": "pliplppplplfpuplpppiiis",
    "vlb_layernorm_bwd_deferred": "pliplppplplfpuplpppiiis",
    "vlb_ln_param_finalize_batch": "ippppis",
    "vlb_gemm_nt_bf16_ex": "plplpliiipiplplplpppifpuiis",
    "vlb_attention_fwd": "ppppiiiifpus",
    "vlb_attention_bwd": "ppppppiiiifpus",
    "vlb_seq_layout": "ppiiiipppppps",
    "vlb_obj_prep_fwd": "plplpppiifpus",
    "vlb_masked_colsum": "plpiipfpuuus",
    "vlb_embed_fwd": "pppp" "pppp" "pll" "pll" "pll" "p" "pp" "ppp" "iiiiiii" "f" "fpu" "s",
    "vlb_embed_bwd": "pppp" "ppppp" "pppppp" "pll" "pll" "pll" "iiiiiii" "fpu" "i" "s",
    "vlb_gather_rows": "pppiis",
    "vlb_scatter_rows": "pppiis",
    "vlb_mlm_compact": "ppiiiipppppps",
    "vlb_ce_fwd_bwd_compact": "pliipppfpps",
    "vlb_head_grad_combine": "ppppiiiiis",
    "vlb_relu_bwd_cast": "pppls",
    "vlb_dgelu_mul": "pppls",
    "vlb_mul_bf16": "pppls",
    "vlb_tanh_bwd": "pppls",
    "vlb_ce_fwd_bwd": "pliippfppls",
    "vlb_soft_ce_fwd_bwd": "pliiplppfppls",
    "vlb_sumsq_f32": "plps",
    "vlb_zero_padded_rows_bf16": "plpliis",
    "vlb_bce_logits_fwd_bwd": "pliiplffppls",
    "vlb_dropout_bf16": "pplfpus",
    "vlb_sumsq_f32_det": "plpips",
    "vlb_adamw_step": "ppppplpfs",
    "vlb_adamw_step_gbf16": "ppppplpfs",
    "vlb_sgd_momentum_step": "pppplfffpffs",
    "vlb_sumsq_bf16_det": "plpips",
    "vlb_sumsq_ranges_det": "pippiiipips",
    "vlb_adamw_step_ranges": "ppipppppiiipfs",
    "vlb_lr_schedule_step": "pifffs",
    "vlb_conv_weight_prepare": "pppppfppppiiiis",
    "vlb_conv_weight_prepare_batched": "ppiifs",
    "vlb_conv_wgrad_finalize": "pppiiiis",
    "vlb_conv3x3_nhwc_bf16": "piiiiiplplipiplps",
    "vlb_conv3x3_wgrad_tn_bf16": "plpiiiiiplipplis",
    "vlb_wgrad_tn_rowscale_bf16": "plplpliiipplis",
    "vlb_im2col_nhwc_bf16": "ppliiiiiiiiis",
    "vlb_im2col_image_f32": "ppiiiiiiiiis",
    "vlb_mask_image_boxes_f32": "piiiiplips",
    "vlb_maxpool3x3s2_nhwc": "ppiiiis",
    "vlb_subsample2_nhwc": "ppiiiis",
    "vlb_upsample2_zero_nhwc": "ppiiiis",
    "vlb_roi_align_nhwc_fwd": "pplipiiiiiifis",
    "vlb_roi_align_nhwc_bwd": "pplipiiiiiiifis",
    "vlb_roi_align_nhwc_bwd_gather": "pplippppliiiiiifis",
    "vlb_relu_mask_cast": "pppls",
    "vlb_avgpool_rows_fwd": "ppliiiiips",
    "vlb_avgpool_rows_bwd": "plpplpiiifpuuups",
    "vlb_gemm_nt_f32": "plplpliiiiillllllplfiplplplfpuiis",
    "vlb_gemm_tn_f32": "plplpliiiiillllllfiips",
    "vlb_transpose_f32": "plpliiiiillllps",
    "vlb_layernorm_f32_fwd": "plppplpiifs",
    "vlb_layernorm_f32_bwd": "plplppplplfpuppiis",
    "vlb_softmax_f32_fwd": "ppippiiifpus",
    "vlb_softmax_f32_bwd": "ppiiifpus",
    "vlb_scale_f32": "plfs",
    "vlb_cast_f32_bf16": "ppls",
    "vlb_cast_bf16_f32": "ppls",
    "vlb_rng_advance": "ps",
    "vlb_roi_align_fwd": "pppiiiiiifis",
    "vlb_roi_align_bwd": "pppiiiiiiifis",
    # RCCL exchange for a C / C++ host (csrc/comm.hip; this Python host issues its collectives through torch.distributed)
    "vlb_comm_unique_id": "p",
    "vlb_comm_init": "iipp",
    "vlb_comm_allreduce_bucket": "pplis",
    "vlb_comm_reduce_scatter_bucket": "ppplis",
    "vlb_comm_allgather_bucket": "ppplis",
    "vlb_comm_finalize": "p",
}
_CT = {"p": _P, "l": _L, "i": _I, "f": _F, "u": _U, "s": _P}

_lib = None


def load():
    """Load the shared library (idempotent).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Import torch FIRST so the
    # process has exactly one HIP runtime -- the one that owns torch's streams and allocations; loading this
    # library first would bind it to /opt/rocm's copy and leave two runtimes fighting over the device.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "libvlbert_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or vl-bert_amd/csrc/build.sh).  There is no CPU / eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.vlb_act_dtype.restype = _I
    lib.vlb_act_dtype.argtypes = []
    if lib.vlb_act_dtype() != (1 if PRECISION == "f16" else 0):
        raise RuntimeError("%s was built for %s but VLB_PRECISION is %s" % (LIB_PATH, "f16" if lib.vlb_act_dtype() else "bf16", PRECISION))
    lib.vlb_last_error.restype = ctypes.c_char_p
    lib.vlb_last_error.argtypes = []
    lib.vlb_version.restype = _I
    lib.vlb_device_info.restype = _I
    lib.vlb_device_info.argtypes = [_I, ctypes.c_char_p, _I]
    lib.vlb_wgrad_workspace_floats.restype = _L
    lib.vlb_wgrad_workspace_floats.argtypes = [_I, _I, _I]
    lib.vlb_layernorm_bwd_workspace_floats.restype = _L
    lib.vlb_layernorm_bwd_workspace_floats.argtypes = [_I]
    lib.vlb_layernorm_bwd_slabs.restype = _I
    lib.vlb_layernorm_bwd_slabs.argtypes = [_I]
    lib.vlb_wgrad_tn_table_desc_bytes.restype = _L
    lib.vlb_wgrad_tn_table_desc_bytes.argtypes = []
    lib.vlb_gemm_set_option.restype = _I
    lib.vlb_gemm_set_option.argtypes = [ctypes.c_char_p, _I]
    lib.vlb_nonfinite_status.restype = _I
    lib.vlb_nonfinite_status.argtypes = [_I]
    lib.vlb_roi_align_gather_workspace_bytes.restype = _L
    lib.vlb_roi_align_gather_workspace_bytes.argtypes = [_I, _I, _I, _I, _I]
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = [_CT[c] for c in sig]
    _lib = lib
    return lib


def set_precision(precision):
    """Switch the process to the other build of the library (tests; A/B runs).  Tensors of engines built before the switch keep
    their dtype: do not interleave engines of two precisions."""
    global PRECISION, LIB_PATH, _lib
    precision = {"fp16": "f16", "half": "f16", "float16": "f16"}.get(precision, precision)
    if precision not in ("bf16", "f16"):
        raise ValueError("precision must be bf16 or f16")
    if precision != PRECISION:
        PRECISION, LIB_PATH, _lib = precision, _default_path(precision), None
        from . import ops
        ops.BF16 = act_torch_dtype()
    return load()


def act_torch_dtype():
    import torch
    return torch.float16 if PRECISION == "f16" else torch.bfloat16


def exported_names():
    return ["vlb_last_error", "vlb_version", "vlb_act_dtype", "vlb_device_info", "vlb_wgrad_workspace_floats",
            "vlb_layernorm_bwd_workspace_floats", "vlb_layernorm_bwd_slabs", "vlb_gemm_set_option",
            "vlb_roi_align_gather_workspace_bytes", "vlb_nonfinite_status", "vlb_wgrad_tn_table_desc_bytes"] + sorted(_SIGS)


def nonfinite_status(reset=True):
    """Sticky flag of the LayerNorm forwards on the current device (vlb_nonfinite_status): 0 clean | bit 0 an fp16 pre-LayerNorm row
    overflowed | bit 1 a bf16 row held inf / NaN."""
    lib = load()
    v = lib.vlb_nonfinite_status(1 if reset else 0)
    if v < 0:
        raise RuntimeError("vlb_nonfinite_status: %s" % lib.vlb_last_error().decode())
    return v


def gemm_set_option(name, value):
    lib = load()
    if lib.vlb_gemm_set_option(name.encode(), int(value)) != 0:
        raise RuntimeError("vlb_gemm_set_option: %s" % lib.vlb_last_error().decode())


def call(name, *args):
    """Invoke an entry point; non-zero return -> RuntimeError(vlb_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.vlb_last_error().decode()))


def device_info(device=0):
    lib = load()
    buf = ctypes.create_string_buffer(64)
    cus = lib.vlb_device_info(device, buf, 64)
    if cus < 0:
        raise RuntimeError("vlb_device_info failed: %s" % lib.vlb_last_error().decode())
    return buf.value.decode(), cus
