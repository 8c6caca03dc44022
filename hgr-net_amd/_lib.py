"""ctypes binding of libhgr.so (include/hgr.h).  There is no fallback: if the library is missing or
an entry point fails, the caller gets an exception - the product path never runs on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# torch must be imported BEFORE libhgr.so is dlopen'ed: the wheel bundles its own libamdhip64.so.7, and a
# process that ends up with two HIP runtimes (torch's and /opt/rocm's, pulled in by libhgr's NEEDED entry)
# loses the device in whichever initialises second ("no ROCm-capable device is detected").  With torch's copy
# already loaded, libhgr's NEEDED libamdhip64.so.7 resolves to it by soname and both share one runtime.
import torch  # noqa: F401

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("HGR_LIB", _HERE / "lib" / "libhgr.so"))

HGR_BF16, HGR_F16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL, EPI_BIAS_RELU, EPI_BIAS_ADD16_RELU, EPI_ACCUM, EPI_QGELU_GRAD16 = 0, 1, 2, 3, 4, 5, 6, 7

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes, exactly the prototypes of include/hgr.h
SIGNATURES = {
    "hgr_gemm_nt": [_p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_set_tile": [_i],
    "hgr_gemm_set_tail": [_i, _i],
    "hgr_gemm_set_persist": [_i],
    "hgr_gemm_set_ws": [_i],
    "hgr_gemm_set_p8": [_i],
    "hgr_im2col_patches": [_p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_im2col_patches_ex": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "hgr_vit_assemble": [_p, _p, _p, _i, _i, _i, _p],
    "hgr_preprocess_bicubic": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _i, _p],
    "hgr_im2col_patches_u8": [_p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _p],
    "hgr_vit_embed_ln": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "hgr_layernorm": [_p, _p, _p, _p, _i, _i, _l, _p, _f, _i, _i, _p],
    "hgr_mha": [_p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_mha_rows": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "hgr_mha_stats": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_text_embed": [_p, _l, _p, _p, _p, _i, _i, _i, _i, _p],
    "hgr_eot_index": [_p, _l, _p, _i, _i, _p],
    "hgr_l2norm_rows": [_p, _p, _p, _i, _i, _i, _p],
    "hgr_topk_rows": [_p, _l, _p, _i, _i, _p, _p, _i, _p],
    "hgr_level_argmax": [_p, _l, _p, _i, _p, _i, _p, _p, _i, _p],
    "hgr_eval_rows": [_p, _l, _i, _p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _i, _p],
    "hgr_conv3x3_nhwc": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hgr_conv3x3_pool2_nhwc": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "hgr_stem_conv1": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_stem_im2col": [_p, _p, _i, _i, _i, _p],
    "hgr_avgpool2_nhwc": [_p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_attnpool_tokens": [_p, _p, _p, _i, _i, _i, _i, _p],
    "hgr_attnpool_attend": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "hgr_transpose16": [_p, _l, _p, _l, _i, _i, _p],
    "hgr_transpose16_colsum": [_p, _l, _p, _l, _i, _i, _i, _p, _i, _f, _p, _p],
    "hgr_colsum": [_p, _l, _i, _i, _i, _i, _p, _i, _f, _p, _p],
    "hgr_cast16": [_p, _p, _l, _i, _p],
    "hgr_cast16_transpose": [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p],
    "hgr_quickgelu16": [_p, _p, _p, _l, _i, _i, _p],
    "hgr_layernorm_bwd": [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _l, _p, _f, _i, _p],
    "hgr_layernorm_bwd_cast": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _p, _f, _i, _p],
    "hgr_layernorm_bwd_cast_colsum": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _p, _f, _i, _p],
    "hgr_layernorm_bwd_scratch_floats": [_i, _i],
    "hgr_mha_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_mha_bwd_stats": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_mha_bwd_colsum": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_ce_rows": [_p, _l, _p, _i, _i, _f, _p, _p, _l, _p],
    "hgr_l2norm_bwd": [_p, _p, _p, _i, _i, _i, _p],
    "hgr_matmul_f32": [_p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _f, _i, _p],
    "hgr_embed_scatter_add": [_p, _l, _p, _p, _i, _i, _i, _i, _p],
    "hgr_rows_axpy": [_p, _l, _p, _p, _i, _i, _f, _p],
    "hgr_rows_gather": [_p, _p, _p, _i, _i, _p],
    "hgr_ctx_splice": [_p, _p, _p, _i, _i, _i, _i, _p],
    "hgr_ctx_splice_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "hgr_sumsq": [_p, _l, _p, _p],
    "hgr_eval_counters": [_p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _p],
    "hgr_dot_f32": [_p, _p, _l, _p, _f, _i, _p],
    "hgr_conv3x3_nhwc_plain": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_splitk": [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_tn_splitk": [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_tn_tile": [_i, _i],
    "hgr_conv3x3_wgrad_splitk": [_p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "hgr_relu_bwd16": [_p, _p, _p, _l, _i, _p],
    "hgr_add16": [_p, _p, _p, _l, _i, _p],
    "hgr_avgpool2_bwd_nhwc": [_p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_attnpool_tokens_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "hgr_im2col3x3_t": [_p, _p, _i, _i, _i, _i, _l, _i, _p],
    "hgr_bn_fold": [_p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _p],
    "hgr_bn_unfold_grad": [_p, _l, _p, _p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _p],
    "hgr_csr_group_aggregate": [_p, _l, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _l, _i, _f, _i, _p],
    "hgr_gemm_nt_res_stats": [_p, _l, _p, _l, _p, _p, _l, _p, _p, _i, _i, _i, _i, _p],
    "hgr_vit_head": [_p, _p, _l, _l, _p, _p, _f, _p, _p, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_res_stats_guard": [_p, _l, _p, _l, _p, _p, _l, _p, _p, _f, _p, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_ln": [_p, _l, _p, _l, _p, _l, _p, _p, _p, _f, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_ln_mha": [_p, _l, _p, _l, _p, _p, _p, _f, _p, _l, _i, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_bias_gelu_dual": [_p, _l, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _p],
    "hgr_gemm_nt_qgelu_grad_colsum": [_p, _l, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _p],
    "hgr_vit_embed_ln_stats": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p],
    "hgr_row_stats16": [_p, _p, _p, _p, _i, _i, _i, _p],
    "hgr_pair_rows_f32": [_p, _p, _p, _i, _i, _l, _p, _i, _p],
    "hgr_logits_eval_workspace_bytes": [_i, _i],
    "hgr_logits_eval": [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _p],
    "hgr_logits_eval_tile_stage": [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _p],
    "hgr_logits_eval_row_stage": [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _p],
    "hgr_comm_unique_id": [_p],
    "hgr_comm_init": [_i, _i, _p],
    "hgr_comm_destroy": [],
    "hgr_comm_rank": [],
    "hgr_comm_world": [],
    "hgr_allreduce": [_p, _p, _l, _i, _i, _p],
    "hgr_allgather": [_p, _p, _l, _i, _p],
    "hgr_broadcast": [_p, _l, _i, _i, _p],
    "hgr_adamw": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p, _f, _f, _p],
}


class HgrError(RuntimeError):
    pass


_lib = None


ABI_VERSION = 4          # HGR_ABI_VERSION of include/hgr.h this wrapper was written against


def load() -> C.CDLL:
    """Load libhgr.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.is_file():
        raise HgrError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {_HERE / 'csrc'}`; there is no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    lib.hgr_abi_version.restype = _i
    lib.hgr_last_error.restype = C.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _l if name.endswith(("_scratch_floats", "_workspace_bytes")) else _i
    if lib.hgr_abi_version() != ABI_VERSION:
        raise HgrError(f"libhgr.so ABI {lib.hgr_abi_version()} != {ABI_VERSION} (include/hgr.h): stale build, run `make -C {_HERE / 'csrc'}`")
    _lib = lib
    return lib


# True only while tree_model captures the TAIL graph of a pipelined evaluation step: the host code of the step's head is walked again
# (same views, same workspace buffers) with its launches skipped, up to ops.split_point().  Never set on any other path.
# Launch mute of the CALLING THREAD (tree_model._eager_phase walks the head's host code with its launches muted while it captures
# the tail graph).  Thread-local: another thread that drives ops meanwhile keeps launching.
import threading

_tls = threading.local()


def set_muted(on: bool) -> bool:
    prev = getattr(_tls, "muted", False)
    _tls.muted = bool(on)
    return prev


def muted() -> bool:
    return getattr(_tls, "muted", False)


def call(name: str, *args) -> None:
    if getattr(_tls, "muted", False):
        return
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HgrError(f"{name} failed ({rc}): {lib.hgr_last_error().decode()}")
