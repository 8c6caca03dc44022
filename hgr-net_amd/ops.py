"""Tensor-level wrappers over the C ABI: take torch CUDA(HIP) tensors, pass raw device pointers,
sizes and the current HIP stream to libhgr.so.  torch is plumbing here (device memory, streams);
all arithmetic happens in the hand-written kernels."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_ACCUM, EPI_BIAS, EPI_BIAS_ADD16_RELU, EPI_BIAS_QUICKGELU, EPI_BIAS_RELU, EPI_BIAS_RESIDUAL, EPI_NONE, EPI_QGELU_GRAD16,
                   HGR_BF16, HGR_F16)

# when a list, every gemm_nt call appends (name, start_event, end_event, algorithmic flops, algorithmic bytes, tag);
# events are recorded on the launch stream (torch's current stream).  Used by bench.py's roofline pass only.
PROFILE = None

# bumped by every raw kernel that writes parameter storage behind torch's back (hgr_adamw): tensors written through a
# ctypes pointer keep their data_ptr and _version, so caches keyed on those (CLIP._prepared, the HIP graphs of
# tree_model.forward) also key on this counter.
WEIGHTS_GEN = [0]

# set by tree_model while it captures the head / tail graphs of a pipelined evaluation step (model/clip_tree.py): called at the point
# of the step behind which only the class-token rows are worked on (clip/model.py:_run_blocks)
SPLIT_HOOK = None


def split_point(name: str) -> None:
    h = SPLIT_HOOK
    if h is not None:
        h(name)


TORCH16 = {HGR_BF16: torch.bfloat16, HGR_F16: torch.float16}
# the residual stream between the folded-LayerNorm GEMMs is a PAIR (csrc/hgr_common.h: pair_split / pair_dec): hi = x rounded to the
# MFMA type and one byte q = the next 8 mantissa bits of x (19 significant bits with f16, 3 bytes).  PAIR_LO is the byte plane's dtype.
PAIR_LO = torch.uint8


def pair_value(xh: torch.Tensor, xl: torch.Tensor) -> torch.Tensor:
    """fp32 value of every row of the pair (decoded by hgr_pair_rows_f32 - the kernels' own decoder)."""
    out = torch.empty(xh.shape, dtype=torch.float32, device=xh.device)
    return pair_rows_f32(xh.contiguous(), xl.contiguous(), out)
DT_OF = {torch.bfloat16: HGR_BF16, torch.float16: HGR_F16}


def dtype_code(name) -> int:
    if name in (HGR_BF16, HGR_F16):
        return name
    return {"bf16": HGR_BF16, "bfloat16": HGR_BF16, "f16": HGR_F16, "fp16": HGR_F16, "float16": HGR_F16,
            torch.bfloat16: HGR_BF16, torch.float16: HGR_F16}[name]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise _lib.HgrError("libhgr ops need device tensors (no CPU path)")
    return t.data_ptr()


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE, n: Optional[int] = None, tag: str = "") -> torch.Tensor:
    """out[M, :N] = epilogue(a[M,K] @ w[N,K]^T); `out` fp32 or the operands' 16-bit type (may be wider than N).
    `tag` only labels the PROFILE record (bench.py's roofline pass)."""
    assert a.dim() == 2 and w.dim() == 2 and out.dim() == 2 and a.dtype == w.dtype and a.stride(1) == 1 and w.stride(1) == 1
    m, k = a.shape
    nn = w.shape[0] if n is None else n
    assert w.shape[1] == k and out.shape[0] == m and out.shape[1] >= nn and out.stride(1) == 1
    out32 = out.dtype == torch.float32
    assert out32 or out.dtype == a.dtype
    if residual is not None:
        assert residual.stride(1) == 1 and residual.dtype == (a.dtype if epilogue in (EPI_BIAS_ADD16_RELU, EPI_QGELU_GRAD16) else torch.float32)
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.call("hgr_gemm_nt", _dev(a), a.stride(0), _dev(w), w.stride(0), _dev(out), out.stride(0),
              _dev(bias), _dev(residual), residual.stride(0) if residual is not None else 0,
              m, nn, k, DT_OF[a.dtype], epilogue, 1 if out32 else 0, _stream())
    if prof is not None:
        ev1.record()
        byts = 2 * m * k + 2 * nn * k + (4 if out32 else 2) * m * nn + (4 * m * nn if residual is not None else 0)
        prof.append(("gemm_nt", ev0, ev1, 2.0 * m * nn * k, float(byts), tag))
    return out


def _prof_begin():
    if PROFILE is None:
        return None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    return ev0, ev1


def _prof_end(ev, flops, byts, tag):
    if ev is not None:
        ev[1].record()
        PROFILE.append(("gemm_nt", ev[0], ev[1], float(flops), float(byts), tag))


# Range guard of the LayerNorm-folded residual stream (hgr_gemm_nt_res_stats_guard): a 64-column slot whose sum of squares exceeds
# LN_GUARD_SUMSQ = 16384^2 could hold an element beyond a quarter of the f16 range (65504); below it every |x| <= 16384.
LN_GUARD_SUMSQ = 16384.0 ** 2


def gemm_nt_res_stats(a: torch.Tensor, w: torch.Tensor, xh: torch.Tensor, xl: torch.Tensor, bias: torch.Tensor, stats: torch.Tensor,
                      tag: str = "", flag: Optional[torch.Tensor] = None) -> None:
    """(xh, xl) += a @ w^T + bias on the residual stream kept as a pair (xh = x in the MFMA type = the next GEMM's A operand, xl one
    byte per element: the rounding error of xh on an 8-bit grid, see PAIR_LO) and, for the LayerNorm that follows, stats[m, slot] = (sum, sum of squares) of every 64-column slot of the
    new row (hgr_gemm_nt_res_stats).  `flag` (uint32 / int32 [1], zeroed by the caller): range guard, non-zero afterwards when a
    slot's sum of squares exceeded LN_GUARD_SUMSQ or was inf / NaN (hgr_gemm_nt_res_stats_guard)."""
    m, k = a.shape
    n = w.shape[0]
    assert a.dtype == w.dtype == xh.dtype and xl.dtype == PAIR_LO and xh.shape == xl.shape == (m, n) and xh.stride() == xl.stride()
    assert stats.dtype == torch.float32 and stats.numel() >= m * (n // 64) * 2 and a.stride(1) == w.stride(1) == xh.stride(1) == 1
    ev = _prof_begin()
    if flag is None:
        _lib.call("hgr_gemm_nt_res_stats", _dev(a), a.stride(0), _dev(w), w.stride(0), _dev(xh), _dev(xl), xh.stride(0), _dev(bias),
                  _dev(stats), m, n, k, DT_OF[a.dtype], _stream())
    else:
        assert flag.numel() == 1 and flag.element_size() == 4
        _lib.call("hgr_gemm_nt_res_stats_guard", _dev(a), a.stride(0), _dev(w), w.stride(0), _dev(xh), _dev(xl), xh.stride(0), _dev(bias),
                  _dev(stats), LN_GUARD_SUMSQ, _dev(flag), m, n, k, DT_OF[a.dtype], _stream())
    _prof_end(ev, 2.0 * m * n * k, 2 * m * k + 2 * n * k + 6 * m * n + 8 * m * (n // 64), tag)     # pair: 3 bytes read + 3 written per element; slot statistics


def gemm_nt_bias_gelu_dual(a: torch.Tensor, w: torch.Tensor, pre: torch.Tensor, post: torch.Tensor, bias: torch.Tensor) -> None:
    """pre = a @ w^T + bias, post = QuickGELU(pre as rounded), both 16-bit, one launch (training forward of the MLP)."""
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype == pre.dtype == post.dtype and a.stride(1) == w.stride(1) == pre.stride(1) == post.stride(1) == 1
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and pre.shape == (m, n) and post.shape == (m, n) and bias.dtype == torch.float32 and bias.numel() == n
    _lib.call("hgr_gemm_nt_bias_gelu_dual", _dev(a), a.stride(0), _dev(w), w.stride(0), _dev(pre), pre.stride(0), _dev(post), post.stride(0),
              _dev(bias), m, n, k, DT_OF[a.dtype], _stream())


def gemm_nt_qgelu_grad_colsum(dy: torch.Tensor, wt: torch.Tensor, dx: torch.Tensor, pre: torch.Tensor, part: torch.Tensor) -> None:
    """dx (16-bit) = (dy @ wt^T) * g'(pre) - gemm_nt(..., epilogue=EPI_QGELU_GRAD16) bit for bit - and part[u, :] = column sums of dx
    as rounded over rows [64 u, 64 u + 64): the bias gradient of the layer below is colsum(part) (hgr_gemm_nt_qgelu_grad_colsum)."""
    assert dy.dim() == 2 and wt.dim() == 2 and dy.dtype == wt.dtype == dx.dtype == pre.dtype and dy.stride(1) == wt.stride(1) == dx.stride(1) == pre.stride(1) == 1
    m, k = dy.shape
    n = wt.shape[0]
    assert wt.shape[1] == k and dx.shape == (m, n) and pre.shape == (m, n)
    assert part.dtype == torch.float32 and part.is_contiguous() and part.shape == ((m + 63) // 64, n)
    _lib.call("hgr_gemm_nt_qgelu_grad_colsum", _dev(dy), dy.stride(0), _dev(wt), wt.stride(0), _dev(dx), dx.stride(0), _dev(pre), pre.stride(0),
              _dev(part), m, n, k, DT_OF[dy.dtype], _stream())


def gelu_dual_ok(m: int, n: int, k: int, lda: int, ldw: int) -> bool:
    """Shape contract of hgr_gemm_nt_bias_gelu_dual (the 256 x 128 tile kernel: whole column tiles, 32-bit operand offsets)."""
    return n % 128 == 0 and k % 64 == 0 and k >= 128 and m * lda * 2 < (1 << 32) and n * ldw * 2 < (1 << 32)


def gemm_nt_ln(x16: torch.Tensor, wfold: torch.Tensor, out: torch.Tensor, ln_s: torch.Tensor, ln_c: torch.Tensor, stats: torch.Tensor,
               eps: float = 1e-5, quickgelu: bool = False, tag: str = "") -> torch.Tensor:
    """out (16-bit) = act(LayerNorm(x) @ W^T + b) from the un-normalised 16-bit rows x16, the gamma-folded weight and the
    row statistics of the producer (hgr_gemm_nt_ln; see include/hgr.h)."""
    m, k = x16.shape
    n = wfold.shape[0]
    assert x16.dtype == wfold.dtype == out.dtype and out.shape[0] == m and out.shape[1] >= n and ln_s.numel() == n and ln_c.numel() == n
    assert x16.stride(1) == wfold.stride(1) == out.stride(1) == 1 and stats.dtype == torch.float32 and stats.numel() >= m * (k // 64) * 2
    ev = _prof_begin()
    _lib.call("hgr_gemm_nt_ln", _dev(x16), x16.stride(0), _dev(wfold), wfold.stride(0), _dev(out), out.stride(0), _dev(ln_s), _dev(ln_c),
              _dev(stats), float(eps), m, n, k, DT_OF[x16.dtype], 1 if quickgelu else 0, _stream())
    _prof_end(ev, 2.0 * m * n * k, 2 * m * k + 2 * n * k + 2 * m * n, tag)
    return out


def gemm_nt_ln_mha(x16: torch.Tensor, wfold: torch.Tensor, att: torch.Tensor, ln_s: torch.Tensor, ln_c: torch.Tensor, stats: torch.Tensor,
                   b: int, l: int, heads: int, causal: bool, eps: float = 1e-5, tag: str = "qkv_mha") -> torch.Tensor:
    """att [b*l, heads*64] = softmax(q k^T / 8 [+ causal]) v with [q | k | v] = LayerNorm(x) @ W_in^T + b_in, one launch, no qkv tensor
    (hgr_gemm_nt_ln_mha; see include/hgr.h).  Same bits as gemm_nt_ln into a qkv buffer + mha."""
    m, k = x16.shape
    w = heads * 64
    assert m == b * l and k == w and wfold.shape == (3 * w, k) and x16.dtype == wfold.dtype == att.dtype and att.shape == (m, w)
    assert x16.stride(1) == wfold.stride(1) == att.stride(1) == 1 and ln_s.numel() == ln_c.numel() == 3 * w
    assert stats.dtype == torch.float32 and stats.numel() >= m * (k // 64) * 2
    ev = _prof_begin()
    _lib.call("hgr_gemm_nt_ln_mha", _dev(x16), x16.stride(0), _dev(wfold), wfold.stride(0), _dev(ln_s), _dev(ln_c), _dev(stats), float(eps),
              _dev(att), att.stride(0), b, l, heads, 1 if causal else 0, DT_OF[x16.dtype], _stream())
    _prof_end(ev, 2.0 * m * 3 * w * k + 4.0 * b * heads * l * l * 64, 2 * m * k + 2 * 3 * w * k + 2 * m * w, tag)
    return att


def ln_mha_ok(w: int, l: int, rows: Optional[int] = None, ldx: Optional[int] = None) -> bool:
    """Shape contract of hgr_gemm_nt_ln_mha: whole sequences inside a 256-row tile, head width 64, row width a multiple of 128, and
    (the kernel's LDS-DMA addresses its operands with 32-bit byte offsets) operands below 4 GB.  Larger inputs are REJECTED further
    down as well: hgr_gemm_nt_ln has the same 32-bit offsets (ln_common_checks raises HgrError) - slice the batch (the towers' callers
    never get there: 4 GB of 16-bit rows is 2.8 M tokens of width 768)."""
    if not (1 <= l <= 64 and w % 128 == 0):
        return False
    if rows is not None and rows * (ldx if ldx is not None else w) * 2 >= (1 << 32):
        return False
    return 3 * w * w * 2 < (1 << 32)


def vit_embed_ln_stats(patches, cls, pos, gamma, beta, xh, xl, stats, b, g, eps=1e-5):
    _lib.call("hgr_vit_embed_ln_stats", _dev(patches), _dev(cls), _dev(pos), _dev(gamma), _dev(beta), _dev(xh), _dev(xl), _dev(stats),
              b, g, xh.shape[1], eps, DT_OF[xh.dtype], _stream())


def row_stats16(x: torch.Tensor, xh: torch.Tensor, xl: torch.Tensor, stats: torch.Tensor) -> None:
    assert x.dtype == torch.float32 and x.is_contiguous() and xh.is_contiguous() and xl.is_contiguous() and xh.shape == xl.shape == x.shape
    assert xl.dtype == PAIR_LO and stats.numel() >= x.shape[0] * (x.shape[1] // 64) * 2
    _lib.call("hgr_row_stats16", _dev(x), _dev(xh), _dev(xl), _dev(stats), x.shape[0], x.shape[1], DT_OF[xh.dtype], _stream())


def pair_rows_f32(xh: torch.Tensor, xl: torch.Tensor, out: torch.Tensor, row_mul: int = 1, row_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = xh[src] + xl[src] (fp32), src = i * row_mul + row_idx[i]: selected rows of the 16-bit pair back in fp32."""
    assert xh.is_contiguous() and xl.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32 and out.shape[1] == xh.shape[1] and xl.dtype == PAIR_LO
    _lib.call("hgr_pair_rows_f32", _dev(xh), _dev(xl), _dev(out), out.shape[0], out.shape[1], row_mul, _dev(row_idx), DT_OF[xh.dtype], _stream())
    return out


def vit_head(xh: torch.Tensor, xl: torch.Tensor, row_mul: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, proj_t: torch.Tensor,
             out: torch.Tensor) -> torch.Tensor:
    """out[b] = LayerNorm(xh[b * row_mul] + xl[b * row_mul]) (16-bit) @ proj_t^T, fp32 [B, D] (hgr_vit_head: ln_post + visual.proj)."""
    assert xh.dtype == proj_t.dtype and xl.dtype == PAIR_LO and xh.stride() == xl.stride() and xh.stride(1) == 1
    assert out.dtype == torch.float32 and out.is_contiguous() and proj_t.is_contiguous() and proj_t.shape[1] == xh.shape[1] and out.shape[1] == proj_t.shape[0]
    ev = _prof_begin()
    _lib.call("hgr_vit_head", _dev(xh), _dev(xl), xh.stride(0), int(row_mul), _dev(gamma), _dev(beta), float(eps), _dev(proj_t), _dev(out),
              out.shape[0], xh.shape[1], out.shape[1], DT_OF[xh.dtype], _stream())
    _prof_end(ev, 2.0 * out.shape[0] * out.shape[1] * xh.shape[1], 0, "head")
    return out


def gemm_set_tile(tile: int) -> int:
    """Pin hgr_gemm_nt's tile plan (0 = cost model, 128, 256); returns the previous setting."""
    lib = _lib.load()
    prev = lib.hgr_gemm_set_tile(int(tile))
    if prev < 0:
        raise _lib.HgrError(f"hgr_gemm_set_tile failed ({prev}): {lib.hgr_last_error().decode()}")
    return prev


def gemm_set_tail(enabled: bool, full_panels: int = -1) -> int:
    """Tail plan of the 256 x 128 tile kernel on / off (hgr_gemm_set_tail); returns the previous setting."""
    lib = _lib.load()
    prev = lib.hgr_gemm_set_tail(1 if enabled else 0, int(full_panels))
    if prev < 0:
        raise _lib.HgrError(f"hgr_gemm_set_tail failed ({prev}): {lib.hgr_last_error().decode()}")
    return prev


def gemm_set_persist(enabled: bool) -> int:
    """Persistent form of the residual-producer GEMMs on / off (hgr_gemm_set_persist); returns the previous setting."""
    lib = _lib.load()
    prev = lib.hgr_gemm_set_persist(1 if enabled else 0)
    if prev < 0:
        raise _lib.HgrError(f"hgr_gemm_set_persist failed ({prev}): {lib.hgr_last_error().decode()}")
    return prev


def im2col_patches(image: torch.Tensor, out: torch.Tensor, patch: int) -> torch.Tensor:
    b, c, r, r2 = image.shape
    assert c == 3 and r == r2 and image.dtype == torch.float32 and image.is_contiguous() and out.is_contiguous()
    _lib.call("hgr_im2col_patches", _dev(image), _dev(out), b, r, patch, out.shape[1], DT_OF[out.dtype], _stream())
    return out


def im2col_patches_tokens(image: torch.Tensor, out: torch.Tensor, patch: int, l: int) -> torch.Tensor:
    """Patches written straight into the [B, L] token layout (row b*L + 1 + p); class rows stay as they are."""
    b, c, r, _ = image.shape
    assert image.dtype == torch.float32 and image.is_contiguous() and out.is_contiguous() and out.shape[0] == b * l
    _lib.call("hgr_im2col_patches_ex", _dev(image), _dev(out), b, r, patch, out.shape[1], l, 1, DT_OF[out.dtype], _stream())
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # clip/clip.py:77
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def im2col_patches_u8(image: torch.Tensor, out: torch.Tensor, patch: int, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """uint8 NHWC [B, R, R, 3] -> normalised 16-bit patch rows in (py, px, c) order."""
    import ctypes
    b, r, r2, c = image.shape
    assert image.dtype == torch.uint8 and c == 3 and r == r2 and image.is_contiguous() and out.is_contiguous()
    g = r // patch
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.call("hgr_im2col_patches_u8", _dev(image), _dev(out), b, r, patch, out.shape[1], g * g, 0, m3, s3, DT_OF[out.dtype], _stream())
    return out


def preprocess_bicubic(src: torch.Tensor, off: torch.Tensor, hw: torch.Tensor, tab: Optional[torch.Tensor], xb: torch.Tensor, xk: torch.Tensor, kx: int,
                       yb: torch.Tensor, yk: torch.Tensor, ky: int, out_u8: Optional[torch.Tensor], out_f32: Optional[torch.Tensor],
                       mean=CLIP_MEAN, std=CLIP_STD, b: int = 0, r: int = 0) -> None:
    """Resize + center crop (+ normalise) of packed uint8 RGB images; tables from hgr_net_amd.preprocess."""
    import ctypes
    assert src.dtype == torch.uint8 and off.dtype == torch.int64 and all(t.dtype == torch.int32 for t in (hw, xb, xk, yb, yk))
    t = xb.numel() // (2 * r)                                  # number of table sets
    assert off.numel() == b and hw.numel() == 2 * b and xb.numel() == 2 * t * r and xk.numel() == t * r * kx and yk.numel() == t * r * ky
    assert (tab is None and t == b) or (tab is not None and tab.dtype == torch.int32 and tab.numel() == b)
    assert out_u8 is None or (out_u8.dtype == torch.uint8 and out_u8.is_contiguous() and out_u8.numel() == b * r * r * 3)
    assert out_f32 is None or (out_f32.dtype == torch.float32 and out_f32.is_contiguous() and out_f32.numel() == b * r * r * 3)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.call("hgr_preprocess_bicubic", _dev(src), _dev(off), _dev(hw), _dev(tab), _dev(xb), _dev(xk), kx, _dev(yb), _dev(yk), ky,
              _dev(out_u8), _dev(out_f32), m3, s3, b, r, _stream())


def vit_assemble(t: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, b: int, l: int) -> torch.Tensor:
    _lib.call("hgr_vit_assemble", _dev(t), _dev(cls), _dev(pos), b, l, t.shape[1], _stream())
    return t


def vit_embed_ln(patches, cls, pos, gamma, beta, x, b, g, eps=1e-5):
    _lib.call("hgr_vit_embed_ln", _dev(patches), _dev(cls), _dev(pos), _dev(gamma), _dev(beta), _dev(x), b, g, x.shape[1], eps, _stream())
    return x


def layernorm(x: torch.Tensor, gamma, beta, out: torch.Tensor, rows: Optional[int] = None, row_mul: int = 1,
              row_idx: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    w = x.shape[-1]
    rows = out.shape[0] if rows is None else rows
    out32 = out.dtype == torch.float32
    _lib.call("hgr_layernorm", _dev(x), _dev(gamma), _dev(beta), _dev(out), rows, w, row_mul, _dev(row_idx), eps,
              HGR_BF16 if out32 else DT_OF[out.dtype], 1 if out32 else 0, _stream())
    return out


def mha(qkv: torch.Tensor, out: torch.Tensor, b: int, l: int, heads: int, causal: bool, stats: Optional[torch.Tensor] = None,
        q_rows: int = 0) -> torch.Tensor:
    """``stats`` fp32 [b, heads, l, 2]: also keep every row's softmax statistics (max, 1 / sum) for mha_bwd (training forward).
    ``q_rows`` > 0: only the first q_rows query rows of every sequence are computed and written (hgr_mha_rows)."""
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape[1] == 3 * heads * 64 and out.shape[1] == heads * 64
    if q_rows:
        assert stats is None
        _lib.call("hgr_mha_rows", _dev(qkv), _dev(out), b, l, heads, 1 if causal else 0, int(q_rows), DT_OF[qkv.dtype], _stream())
        return out
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() == b * heads * l * 2
        _lib.call("hgr_mha_stats", _dev(qkv), _dev(out), _dev(stats), b, l, heads, 1 if causal else 0, DT_OF[qkv.dtype], _stream())
        return out
    _lib.call("hgr_mha", _dev(qkv), _dev(out), b, l, heads, 1 if causal else 0, DT_OF[qkv.dtype], _stream())
    return out


def text_embed(tokens: torch.Tensor, emb: torch.Tensor, pos: torch.Tensor, x: torch.Tensor, l: int) -> torch.Tensor:
    assert tokens.dtype == torch.int64 and tokens.stride(1) == 1
    n = tokens.shape[0]
    _lib.call("hgr_text_embed", _dev(tokens), tokens.stride(0), _dev(emb), _dev(pos), _dev(x), n, l, emb.shape[1], emb.shape[0], _stream())
    return x


def eot_index(tokens: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    assert tokens.dtype == torch.int64 and out.dtype == torch.int32
    _lib.call("hgr_eot_index", _dev(tokens), tokens.stride(0), _dev(out), tokens.shape[0], tokens.shape[1], _stream())
    return out


def l2norm_rows(x: torch.Tensor, y16: Optional[torch.Tensor] = None, y32: Optional[torch.Tensor] = None) -> None:
    assert x.dtype == torch.float32 and x.is_contiguous()
    dt = DT_OF[y16.dtype] if y16 is not None else HGR_BF16
    _lib.call("hgr_l2norm_rows", _dev(x), _dev(y16), _dev(y32), x.shape[0], x.shape[1], dt, _stream())


def topk_rows(logits: torch.Tensor, k: int, cols: Optional[torch.Tensor] = None, n_cols: Optional[int] = None,
              want_values: bool = False):
    """int32 [rows, k] node ids of the k largest logits over `cols` (int32) or the first n_cols columns."""
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    rows = logits.shape[0]
    nc = cols.numel() if cols is not None else (logits.shape[1] if n_cols is None else n_cols)
    idx = torch.empty((rows, k), dtype=torch.int32, device=logits.device)
    val = torch.empty((rows, k), dtype=torch.float32, device=logits.device) if want_values else None
    _lib.call("hgr_topk_rows", _dev(logits), logits.stride(0), _dev(cols), nc, k, _dev(idx), _dev(val), rows, _stream())
    return (idx, val) if want_values else idx


def level_argmax(logits: torch.Tensor, depth: torch.Tensor, n_levels: int, cols: Optional[torch.Tensor] = None,
                 n_cols: Optional[int] = None, want_top1: bool = False):
    """int32 [rows, n_levels]: per depth level the arg-max node id over `cols` (main.py:162-176);
    with want_top1 also int32 [rows, 1]: the unmasked top-1 over `cols` (main.py:157)."""
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and depth.dtype == torch.int32
    rows = logits.shape[0]
    nc = cols.numel() if cols is not None else (logits.shape[1] if n_cols is None else n_cols)
    out = torch.empty((rows, n_levels), dtype=torch.int32, device=logits.device)
    top1 = torch.empty((rows, 1), dtype=torch.int32, device=logits.device) if want_top1 else None
    _lib.call("hgr_level_argmax", _dev(logits), logits.stride(0), _dev(cols), nc, _dev(depth), n_levels, _dev(out), _dev(top1), rows, _stream())
    return (out, top1) if want_top1 else out


class EvalIndex:
    """Dense per-column maps for the fused evaluation kernel (built once per model on the host): depth of every node,
    column -> position in the train / test subsets (-1 = absent), position -> node id, and the reference's -1-filler
    position per level."""

    def __init__(self, depth: torch.Tensor, train_cols: torch.Tensor, test_cols: Optional[torch.Tensor], n_levels: int):
        import numpy as np
        d = depth.cpu().numpy()
        n = len(d)
        tr = train_cols.cpu().numpy().astype(np.int64)
        assert d.min() >= 0 and n_levels <= 32 and d[tr].max() < n_levels, "a subset node lies deeper than n_levels"
        assert len(set(tr.tolist())) == len(tr), "train columns must be distinct"
        lv = d[tr]
        filler = np.full(n_levels, -1, dtype=np.int32)
        for l in range(n_levels):
            other = np.nonzero(lv != l)[0]
            if other.size:
                filler[l] = int(other[0])
        tpos = np.full(n, -1, dtype=np.int32)
        tpos[tr] = np.arange(len(tr), dtype=np.int32)
        dev = depth.device
        self.n_nodes, self.n_levels, self.n_train = n, n_levels, len(tr)
        self.lvl8 = torch.from_numpy(np.minimum(d, 255).astype(np.uint8)).to(dev)
        self.train_pos = torch.from_numpy(tpos).to(dev)
        self.train_cols = train_cols.to(torch.int32).contiguous()
        self.filler = torch.from_numpy(filler).to(dev)
        self.test_pos = self.test_cols = None
        self.n_test = 0
        if test_cols is not None:
            te = test_cols.cpu().numpy().astype(np.int64)
            assert len(set(te.tolist())) == len(te), "test columns must be distinct"
            epos = np.full(n, -1, dtype=np.int32)
            epos[te] = np.arange(len(te), dtype=np.int32)
            self.test_pos = torch.from_numpy(epos).to(dev)
            self.test_cols = test_cols.to(torch.int32).contiguous()
            self.n_test = len(te)


def eval_rows(logits: torch.Tensor, index: EvalIndex, k: int):
    """(level arg-max [rows, n_levels], top-1 [rows, 1], top-k [rows, k]) int32 node ids from one fused kernel."""
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.shape[1] >= index.n_nodes
    rows = logits.shape[0]
    dev = logits.device
    lvl = torch.empty((rows, index.n_levels), dtype=torch.int32, device=dev)
    top1 = torch.empty((rows, 1), dtype=torch.int32, device=dev)
    topk = torch.empty((rows, max(k, 1)), dtype=torch.int32, device=dev)
    _lib.call("hgr_eval_rows", _dev(logits), logits.stride(0), index.n_nodes, _dev(index.lvl8), _dev(index.train_pos), _dev(index.train_cols),
              index.n_train, index.n_levels, _dev(index.filler), _dev(index.test_pos), _dev(index.test_cols), index.n_test, k,
              _dev(lvl), _dev(top1), _dev(topk), rows, _stream())
    return lvl, top1, topk


class LogitsEvalPlan:
    """Level-sorted view of the class matrix for hgr_logits_eval (include/hgr.h): the column permutation (levels contiguous, each
    padded to a multiple of 32 columns, total to a multiple of 96 = one CU's slab), the per-column train / test positions in permuted
    order and the first 32-column slice of every level.  Built once per model from an EvalIndex; `bind(zsl16)` gathers the permuted
    16-bit class matrix (once per update_classifier)."""
    SLICE, SLAB = 32, 96

    def __init__(self, index: EvalIndex):
        import numpy as np
        lvl = index.lvl8.cpu().numpy().astype(np.int64)
        tpos = index.train_pos.cpu().numpy()
        epos = index.test_pos.cpu().numpy() if index.test_pos is not None else np.full(len(lvl), -1, dtype=np.int32)
        cols, level_first = [], [0]
        for l in range(index.n_levels):
            c = np.nonzero(lvl == l)[0]
            pad = (-c.size) % self.SLICE
            cols.append(np.concatenate([c, np.full(pad, -1, dtype=np.int64)]))
            level_first.append(level_first[-1] + (c.size + pad) // self.SLICE)
        assert int((lvl >= index.n_levels).sum()) == 0, "a node lies deeper than n_levels"
        perm = np.concatenate(cols)
        tail = (-perm.size) % self.SLAB                              # whole 96-column slabs; the slices behind the last level belong to no level
        if tail or perm.size == 0:
            perm = np.concatenate([perm, np.full(tail or self.SLAB, -1, dtype=np.int64)])
        dev = index.lvl8.device
        valid = perm >= 0
        safe = np.where(valid, perm, 0)
        self.index = index
        self.n_perm = int(perm.size)
        self.perm = torch.from_numpy(safe).to(dev)
        self.valid = torch.from_numpy(valid).to(dev)
        self.tpos = torch.from_numpy(np.where(valid, tpos[safe], -1).astype(np.int32)).to(dev)
        self.epos = torch.from_numpy(np.where(valid, epos[safe], -1).astype(np.int32)).to(dev)
        self.level_first = torch.from_numpy(np.asarray(level_first, dtype=np.int32)).to(dev)
        self.zsl = None
        self._src = None
        self._ws = None

    MAX_SLICES = 1024            # hgr_logits_eval: 32-column slices per row (csrc/hgr_gemm.hip) = 32 768 permuted columns

    @property
    def supported(self) -> bool:
        """Whether hgr_logits_eval covers this hierarchy (padding every level to 32 columns brings hierarchies somewhat below
        32 768 nodes over its capacity): callers fall back to forward() + hgr_eval_rows otherwise."""
        return self.n_perm // self.SLICE <= self.MAX_SLICES and self.index.n_levels <= 32

    def bind(self, zsl16: torch.Tensor, algo_d: Optional[int] = None) -> "LogitsEvalPlan":
        """``algo_d``: the embedding width the product stands for when zsl16 is a K-concatenated operand (clip_tree.LOGITS_SPLIT) -
        profiling reports the ALGORITHMIC flops 2 B N D, not the doubled K."""
        self.algo_d = int(algo_d or zsl16.shape[1])
        key = (zsl16.data_ptr(), zsl16._version, tuple(zsl16.shape))
        if self._src != key:
            z = zsl16[self.perm]
            z[~self.valid] = 0
            self.zsl, self._src = z.contiguous(), key
        return self

    def workspace(self, rows: int, dev) -> torch.Tensor:
        need = int(_lib.load().hgr_logits_eval_workspace_bytes(rows, self.n_perm))
        if self._ws is None or self._ws.numel() < need or self._ws.device != torch.device(dev):
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws


def logits_eval(feat16: torch.Tensor, plan: LogitsEvalPlan, k: int, stage: str = ""):
    """(level arg-max [rows, n_levels], top-1 [rows, 1], top-k [rows, k]) int32 node ids straight from the L2-normalised 16-bit
    image features: the logits GEMM with the evaluation in its epilogue (hgr_logits_eval) - no [rows, N] logits in memory.
    ``stage`` = "tile" / "row": that stage alone (measurement; the row stage reads the workspace the last tile stage wrote)."""
    ix = plan.index
    assert plan.zsl is not None, "LogitsEvalPlan.bind(zsl16) first"
    assert feat16.is_contiguous() and feat16.dtype == plan.zsl.dtype and feat16.shape[1] == plan.zsl.shape[1]
    rows, d = feat16.shape
    dev = feat16.device
    lvl = torch.empty((rows, ix.n_levels), dtype=torch.int32, device=dev)
    top1 = torch.empty((rows, 1), dtype=torch.int32, device=dev)
    topk = torch.empty((rows, max(k, 1)), dtype=torch.int32, device=dev)
    prof = PROFILE
    ev = _prof_begin()
    _lib.call({"": "hgr_logits_eval", "tile": "hgr_logits_eval_tile_stage", "row": "hgr_logits_eval_row_stage"}[stage],
              _dev(feat16), _dev(plan.zsl), rows, d, plan.n_perm, _dev(plan.tpos), _dev(plan.epos), _dev(plan.level_first),
              ix.n_levels, _dev(ix.filler), _dev(ix.train_cols), ix.n_train, _dev(ix.test_cols), ix.n_test, k,
              _dev(lvl), _dev(top1), _dev(topk), _dev(plan.workspace(rows, dev)), DT_OF[feat16.dtype], _stream())
    da = getattr(plan, "algo_d", d)
    _prof_end(ev, 2.0 * rows * ix.n_nodes * da, 2 * rows * da + 2 * ix.n_nodes * da, "logits_eval" + ("_" + stage if stage else ""))
    return lvl, top1, topk


def eval_counters(pred: torch.Tensor, targets: Optional[torch.Tensor], target: int, top1: torch.Tensor, lv: torch.Tensor,
                  parents: torch.Tensor, levels: torch.Tensor, acc: torch.Tensor) -> None:
    """Advance the nine float64 evaluation counters with one batch (main.py:139-191) from eval_rows' outputs."""
    assert pred.dtype == top1.dtype == lv.dtype == parents.dtype == levels.dtype == torch.int32 and acc.dtype == torch.float64
    assert pred.is_contiguous() and lv.is_contiguous() and top1.is_contiguous() and acc.numel() == 9
    assert targets is None or (targets.dtype == torch.int64 and targets.is_contiguous() and targets.numel() == pred.shape[0])
    _lib.call("hgr_eval_counters", _dev(pred), pred.shape[1], _dev(targets), int(target), _dev(top1), _dev(lv), lv.shape[1], _dev(parents),
              _dev(levels), parents.numel(), _dev(acc), pred.shape[0], _stream())


# ---- ModifiedResNet (RN) tower -------------------------------------------------------------------
def conv3x3_nhwc(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, b: int, h: int, wd: int, c: int,
                 stride: int = 1) -> torch.Tensor:
    """out NHWC = relu(conv3x3(x NHWC, pad 1) + bias); w [Cout, Kp] in (ky, kx, c) order (BN folded)."""
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and x.dtype == w.dtype == out.dtype
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.call("hgr_conv3x3_nhwc", _dev(x), _dev(w), _dev(bias), _dev(out), b, h, wd, c, w.shape[0], stride, w.shape[1],
              DT_OF[x.dtype], _stream())
    if prof is not None:
        ev1.record()
        mo = out.shape[0]
        prof.append(("gemm_nt", ev0, ev1, 2.0 * mo * 9 * c * w.shape[0], float(2 * x.numel() + 2 * w.numel() + 2 * out.numel()), "conv3x3"))
    return out


def conv3x3_pool2_nhwc(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, b: int, h: int, wd: int, c: int) -> torch.Tensor:
    """out NHWC [b, h/2, wd/2, Cout] = avgpool2(relu(conv3x3(x NHWC, pad 1) + bias)); C = 32 only (the stem)."""
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and x.dtype == w.dtype == out.dtype
    _lib.call("hgr_conv3x3_pool2_nhwc", _dev(x), _dev(w), _dev(bias), _dev(out), b, h, wd, c, w.shape[0], w.shape[1], DT_OF[x.dtype], _stream())
    return out


def stem_conv1(image: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out NHWC [B*Ho*Ho, Cout] = relu(conv3x3 stride 2 pad 1 (image fp32 NCHW) + bias); w [Cout, Kp] folded, (ky, kx, c) order."""
    assert image.dtype == torch.float32 and image.is_contiguous() and image.shape[1] == 3 and image.shape[2] == image.shape[3]
    assert w.is_contiguous() and out.is_contiguous() and w.dtype == out.dtype and out.shape[1] == w.shape[0]
    _lib.call("hgr_stem_conv1", _dev(image), _dev(w), _dev(bias), _dev(out), image.shape[0], image.shape[2], w.shape[0], w.shape[1],
              DT_OF[out.dtype], _stream())
    return out


def stem_im2col(image: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    assert image.dtype == torch.float32 and image.is_contiguous() and out.is_contiguous() and out.shape[1] == 64
    _lib.call("hgr_stem_im2col", _dev(image), _dev(out), image.shape[0], image.shape[2], DT_OF[out.dtype], _stream())
    return out


def avgpool2_nhwc(x: torch.Tensor, out: torch.Tensor, b: int, h: int, w: int, c: int) -> torch.Tensor:
    _lib.call("hgr_avgpool2_nhwc", _dev(x), _dev(out), b, h, w, c, DT_OF[x.dtype], _stream())
    return out


def attnpool_tokens(x: torch.Tensor, pos: torch.Tensor, out: torch.Tensor, b: int, s: int, c: int) -> torch.Tensor:
    _lib.call("hgr_attnpool_tokens", _dev(x), _dev(pos), _dev(out), b, s, c, DT_OF[x.dtype], _stream())
    return out


def attnpool_attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, b: int, l: int, heads: int) -> torch.Tensor:
    assert q.dtype == torch.float32 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    _lib.call("hgr_attnpool_attend", _dev(q), _dev(k), _dev(v), _dev(out), b, l, heads, DT_OF[k.dtype], _stream())
    return out


# ---- training: backward / optimizer ------------------------------------------------------------------
def transpose16(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[c, r] = x[r, c]; out is [cols, ld >= rows] (its padding columns must already be zero)."""
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1 and x.element_size() == 2 and out.dtype == x.dtype
    _lib.call("hgr_transpose16", _dev(x), x.stride(0), _dev(out), out.stride(0), x.shape[0], x.shape[1], _stream())
    return out


def transpose16_colsum(x: torch.Tensor, out_t: torch.Tensor, colsum_out: torch.Tensor, scratch: torch.Tensor, accumulate: bool = True,
                       alpha: float = 1.0) -> torch.Tensor:
    """out_t[c][r] = x[r][c] and colsum_out[c] (+)= alpha * sum_r x[r][c] in one pass over x (16-bit)."""
    assert x.dim() == 2 and x.stride(1) == 1 and out_t.stride(1) == 1 and x.dtype == out_t.dtype and x.dtype in DT_OF
    assert colsum_out.dtype == torch.float32 and scratch.dtype == torch.float32 and colsum_out.numel() >= x.shape[1]
    rows, cols = x.shape
    assert out_t.shape[0] >= cols and out_t.stride(0) >= rows and scratch.numel() >= (rows + 63) // 64 * cols
    _lib.call("hgr_transpose16_colsum", _dev(x), x.stride(0), _dev(out_t), out_t.stride(0), rows, cols, DT_OF[x.dtype], _dev(colsum_out),
              1 if accumulate else 0, alpha, _dev(scratch), _stream())
    return out_t


def colsum(x: torch.Tensor, out: torch.Tensor, scratch: torch.Tensor, accumulate: bool = True, alpha: float = 1.0) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32 and scratch.dtype == torch.float32
    rows, cols = x.shape
    assert scratch.numel() >= (rows + 511) // 512 * cols
    f32 = x.dtype == torch.float32
    _lib.call("hgr_colsum", _dev(x), x.stride(0), rows, cols, 1 if f32 else 0, HGR_BF16 if f32 else DT_OF[x.dtype], _dev(out),
              1 if accumulate else 0, alpha, _dev(scratch), _stream())
    return out


def cast16(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel()
    _lib.call("hgr_cast16", _dev(x), _dev(out), x.numel(), DT_OF[out.dtype], _stream())
    return out


def cast16_transpose(x: torch.Tensor, y: torch.Tensor, yt: torch.Tensor) -> None:
    """y[r, c] = (16-bit) x[r, c] and yt[c, r] = the same value, one launch (yt's columns beyond x.shape[0] are left as they are)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and y.stride(1) == 1 and yt.stride(1) == 1 and y.dtype == yt.dtype and y.dtype in DT_OF
    rows, cols = x.shape
    assert y.shape[0] == rows and y.shape[1] >= cols and yt.shape[0] >= cols and yt.stride(0) >= rows
    _lib.call("hgr_cast16_transpose", _dev(x), x.stride(0), _dev(y), y.stride(0), _dev(yt), yt.stride(0), rows, cols, DT_OF[y.dtype], _stream())


def quickgelu16(a: torch.Tensor, out: torch.Tensor, du: Optional[torch.Tensor] = None) -> torch.Tensor:
    """forward: out = a*sigmoid(1.702a); with du: out = du * g'(a)."""
    assert a.is_contiguous() and out.is_contiguous() and (du is None or du.is_contiguous())
    _lib.call("hgr_quickgelu16", _dev(a), _dev(du), _dev(out), a.numel(), 0 if du is None else 1, DT_OF[a.dtype], _stream())
    return out


def layernorm_bwd_scratch(rows: int, w: int) -> int:
    return int(_lib.load().hgr_layernorm_bwd_scratch_floats(rows, w))


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dx: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                  scratch: torch.Tensor, rows: Optional[int] = None, row_mul: int = 1, row_idx: Optional[torch.Tensor] = None, eps: float = 1e-5,
                  dx16: Optional[torch.Tensor] = None, dx16_colsum: Optional[torch.Tensor] = None) -> None:
    """dx[src rows] += dLN; dgamma += ...; dbeta += ...  (all fp32, accumulate).  ``dx16``: also the updated dx rows as a 16-bit copy;
    ``dx16_colsum`` (fp32 [w], with dx16): += the column sums of those 16-bit rows (the bias gradient of the Linear they are dY of)."""
    w = x.shape[-1]
    rows = dy.shape[0] if rows is None else rows
    f32 = dy.dtype == torch.float32
    assert scratch.numel() >= layernorm_bwd_scratch(rows, w) and dy.is_contiguous()
    if dx16 is not None and dx16_colsum is not None:
        assert dx16.is_contiguous() and dx16.shape == dx.shape and (f32 or dx16.dtype == dy.dtype)
        assert dx16_colsum.dtype == torch.float32 and dx16_colsum.is_contiguous() and dx16_colsum.numel() == w
        _lib.call("hgr_layernorm_bwd_cast_colsum", _dev(dy), 1 if f32 else 0, _dev(x), _dev(gamma), _dev(dx), _dev(dx16), _dev(dgamma), _dev(dbeta),
                  _dev(dx16_colsum), _dev(scratch), rows, w, row_mul, _dev(row_idx), eps, DT_OF[dx16.dtype], _stream())
        return
    assert dx16_colsum is None
    if dx16 is not None:
        assert dx16.is_contiguous() and dx16.shape == dx.shape and (f32 or dx16.dtype == dy.dtype)
        _lib.call("hgr_layernorm_bwd_cast", _dev(dy), 1 if f32 else 0, _dev(x), _dev(gamma), _dev(dx), _dev(dx16), _dev(dgamma), _dev(dbeta), _dev(scratch),
                  rows, w, row_mul, _dev(row_idx), eps, DT_OF[dx16.dtype], _stream())
        return
    _lib.call("hgr_layernorm_bwd", _dev(dy), 1 if f32 else 0, _dev(x), _dev(gamma), _dev(dx), _dev(dgamma), _dev(dbeta), _dev(scratch),
              rows, w, row_mul, _dev(row_idx), eps, HGR_BF16 if f32 else DT_OF[dy.dtype], _stream())


def mha_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, dqkv: torch.Tensor, b: int, l: int, heads: int, causal: bool,
            stats: Optional[torch.Tensor] = None, colsum_part: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``colsum_part`` (fp32 [b, 3 * heads * 64], every entry written): per-sequence column sums of dqkv as rounded - summed over b
    they are the in_proj bias gradient (hgr_mha_bwd_colsum)."""
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() == b * heads * l * 2
    if colsum_part is not None:
        assert colsum_part.dtype == torch.float32 and colsum_part.is_contiguous() and colsum_part.shape == (b, 3 * heads * 64)
        _lib.call("hgr_mha_bwd_colsum", _dev(qkv), _dev(out), _dev(dout), _dev(dqkv), _dev(stats), _dev(colsum_part), b, l, heads, 1 if causal else 0,
                  DT_OF[qkv.dtype], _stream())
        return dqkv
    if stats is not None:
        _lib.call("hgr_mha_bwd_stats", _dev(qkv), _dev(out), _dev(dout), _dev(dqkv), _dev(stats), b, l, heads, 1 if causal else 0,
                  DT_OF[qkv.dtype], _stream())
        return dqkv
    _lib.call("hgr_mha_bwd", _dev(qkv), _dev(out), _dev(dout), _dev(dqkv), b, l, heads, 1 if causal else 0, DT_OF[qkv.dtype], _stream())
    return dqkv


def ce_rows(logits: torch.Tensor, labels: torch.Tensor, loss_rows: torch.Tensor, dlogits: Optional[torch.Tensor] = None, gscale: float = 1.0) -> None:
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and labels.dtype == torch.int32
    _lib.call("hgr_ce_rows", _dev(logits), logits.stride(0), _dev(labels), logits.shape[0], logits.shape[1], gscale, _dev(loss_rows),
              _dev(dlogits), dlogits.stride(0) if dlogits is not None else 0, _stream())


def l2norm_bwd(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous() and x.dtype == dy.dtype == dx.dtype == torch.float32
    _lib.call("hgr_l2norm_bwd", _dev(x), _dev(dy), _dev(dx), x.shape[0], x.shape[1], 1 if accumulate else 0, _stream())
    return dx


def matmul_f32(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    """out[M,N] (+)= alpha * a[M,K] @ b[K,N] for fp32 2-D tensors with ARBITRARY strides (pass .t() views freely)."""
    assert a.dtype == b.dtype == out.dtype == torch.float32 and out.stride(1) == 1 and a.shape[1] == b.shape[0]
    m, k = a.shape
    n = b.shape[1]
    _lib.call("hgr_matmul_f32", _dev(a), a.stride(0), a.stride(1), _dev(b), b.stride(0), b.stride(1), _dev(out), out.stride(0), m, n, k,
              alpha, 1 if accumulate else 0, _stream())
    return out


# ---- ModifiedResNet tower in training ---------------------------------------------------------------------------
def conv3x3_plain(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, b: int, h: int, wd: int, c: int) -> torch.Tensor:
    """out NHWC = conv3x3(x NHWC, pad 1, stride 1) with w [Cout, Kp] in (ky, kx, c) order; no bias, no activation."""
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and x.dtype == w.dtype == out.dtype
    _lib.call("hgr_conv3x3_nhwc_plain", _dev(x), _dev(w), _dev(out), b, h, wd, c, w.shape[0], w.shape[1], DT_OF[x.dtype], _stream())
    return out


def gemm_nt_splitk(a: torch.Tensor, w: torch.Tensor, partial: torch.Tensor, kc: int) -> torch.Tensor:
    """partial[s] [M, N] fp32 = a[:, s*kc:(s+1)*kc] @ w[:, same]^T for every K slice."""
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.dtype == w.dtype and a.shape[1] == w.shape[1]
    m, k = a.shape
    n = w.shape[0]
    s = (k + kc - 1) // kc
    assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.numel() >= s * m * n
    _lib.call("hgr_gemm_nt_splitk", _dev(a), a.stride(0), _dev(w), w.stride(0), _dev(partial), n, m, n, k, kc, DT_OF[a.dtype], _stream())
    return partial


def splitk_slices(tiles: int, m: int, slots: int = 512) -> int:
    """Number of reduction slices for a split-K weight gradient with `tiles` output tiles over `m` rows, each slice at least 512
    rows deep: from the smallest count s0 that fills >= 85 % of whole rounds of `slots` workgroups up to 1.25 s0, the one that fills
    them best (16 tiles on 256 slots: 16 slices = 100 % instead of 14 = 87.5 %; measured on the 1024 x 1024 out_proj gradient of
    ViT-L/14 at batch 256: the idle CUs cost more than the two extra partial tiles)."""
    smax = max(1, min(m // 512, 4096))
    best = (0.0, 1)
    first = None
    for s in range(1, smax + 1):
        wgs = tiles * s
        eff = wgs / (-(-wgs // slots) * slots)
        if first is None and eff >= 0.85:
            first, best = s, (eff, s)
        elif first is not None:
            if s > max(first + 1, int(first * 1.25)):
                break
            if eff > best[0] + 1e-9:
                best = (eff, s)
        elif eff > best[0]:
            best = (eff, s)
    return best[1]


def tn_slices(na: int, nb: int, m: int) -> int:
    """Reduction slices of hgr_gemm_tn_splitk for a [na, nb] product over m rows, for the tile plan the library will run
    (hgr_gemm_tn_tile: 256 x 256 tiles, one workgroup per CU, or 128 x 128, two per CU)."""
    t = int(_lib.load().hgr_gemm_tn_tile(na, nb))
    return splitk_slices(-(-na // t) * -(-nb // t), m, slots=256 if t == 256 else 512)


def gemm_tn_splitk(p: torch.Tensor, q: torch.Tensor, partial: torch.Tensor, kc: int) -> torch.Tensor:
    """partial[s] [Na, Nb] = p[rows of slice s]^T . q[rows of slice s]; p [M, Na], q [M, Nb] 16-bit row-major."""
    assert p.dim() == 2 and q.dim() == 2 and p.stride(1) == 1 and q.stride(1) == 1 and p.dtype == q.dtype and p.shape[0] == q.shape[0]
    m, na = p.shape
    nb = q.shape[1]
    s = (m + kc - 1) // kc
    assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.numel() >= s * na * nb
    _lib.call("hgr_gemm_tn_splitk", _dev(p), p.stride(0), _dev(q), q.stride(0), _dev(partial), nb, m, na, nb, kc, DT_OF[p.dtype], _stream())
    return partial


def conv3x3_wgrad_splitk(dy: torch.Tensor, x: torch.Tensor, partial: torch.Tensor, b: int, h: int, w: int, c: int, kc: int) -> torch.Tensor:
    """partial[s] [Cout, 9 C] in (ky, kx, c) order = dy[slice]^T . im2col(x)[slice] for a 3x3 / pad 1 / stride 1 convolution."""
    assert dy.dim() == 2 and dy.stride(1) == 1 and x.is_contiguous() and dy.dtype == x.dtype and dy.shape[0] == b * h * w
    cout = dy.shape[1]
    s = (b * h * w + kc - 1) // kc
    assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.numel() >= s * cout * 9 * c
    _lib.call("hgr_conv3x3_wgrad_splitk", _dev(dy), dy.stride(0), _dev(x), _dev(partial), 9 * c, b, h, w, c, cout, kc, DT_OF[dy.dtype], _stream())
    return partial


def relu_bwd16(dy: torch.Tensor, y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = dy if out is None else out
    assert dy.is_contiguous() and y.is_contiguous() and out.is_contiguous() and dy.numel() == y.numel() == out.numel()
    _lib.call("hgr_relu_bwd16", _dev(dy), _dev(y), _dev(out), dy.numel(), DT_OF[dy.dtype], _stream())
    return out


def add16(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = a if out is None else out
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and a.numel() == b.numel() == out.numel()
    _lib.call("hgr_add16", _dev(a), _dev(b), _dev(out), a.numel(), DT_OF[a.dtype], _stream())
    return out


def avgpool2_bwd_nhwc(dy: torch.Tensor, dx: torch.Tensor, b: int, h: int, w: int, c: int) -> torch.Tensor:
    assert dy.is_contiguous() and dx.is_contiguous() and dx.numel() == 4 * dy.numel() == b * h * w * c
    _lib.call("hgr_avgpool2_bwd_nhwc", _dev(dy), _dev(dx), b, h, w, c, DT_OF[dy.dtype], _stream())
    return dx


def attnpool_tokens_bwd(dtok: torch.Tensor, dx: torch.Tensor, b: int, s: int, c: int) -> torch.Tensor:
    assert dtok.is_contiguous() and dx.is_contiguous() and dtok.numel() == b * (s + 1) * c and dx.numel() == b * s * c
    _lib.call("hgr_attnpool_tokens_bwd", _dev(dtok), _dev(dx), b, s, c, DT_OF[dtok.dtype], _stream())
    return dx


def im2col3x3_t(xt: torch.Tensor, out: torch.Tensor, b: int, h: int, w: int) -> torch.Tensor:
    """xt [C, ld] (transposed NHWC activation) -> out [9C, ld]: the transposed im2col of a 3x3 / pad 1 convolution."""
    c, ld = xt.shape
    assert xt.is_contiguous() and out.is_contiguous() and tuple(out.shape) == (9 * c, ld) and ld >= b * h * w
    _lib.call("hgr_im2col3x3_t", _dev(xt), _dev(out), b, h, w, c, ld, DT_OF[xt.dtype], _stream())
    return out


def bn_fold(w: torch.Tensor, bn, w16: torch.Tensor, bias: torch.Tensor) -> None:
    """conv weight fp32 [Cout, Cin, kh, kw] + BatchNorm2d (running stats) -> folded w16 [Cout, Kp] ((ky, kx, ci) order), bias."""
    cout, cin, kh, kw = w.shape
    assert w.is_contiguous() and w.dtype == torch.float32 and w16.is_contiguous() and bias.dtype == torch.float32
    _lib.call("hgr_bn_fold", _dev(w), _dev(bn.weight.data), _dev(bn.bias.data), _dev(bn.running_mean), _dev(bn.running_var), float(bn.eps),
              _dev(w16), _dev(bias), cout, cin, kh * kw, w16.shape[1], DT_OF[w16.dtype], _stream())


def bn_unfold_grad(gwf: torch.Tensor, gbf: torch.Tensor, w: torch.Tensor, bn, g_w: torch.Tensor, g_gamma: torch.Tensor, g_beta: torch.Tensor) -> None:
    cout, cin, kh, kw = w.shape
    assert gwf.dtype == gbf.dtype == g_w.dtype == torch.float32 and gwf.stride(1) == 1 and g_w.is_contiguous() and w.is_contiguous()
    _lib.call("hgr_bn_unfold_grad", _dev(gwf), gwf.stride(0), _dev(gbf), _dev(w), _dev(bn.weight.data), _dev(bn.running_mean),
              _dev(bn.running_var), float(bn.eps), _dev(g_w), _dev(g_gamma), _dev(g_beta), cout, cin, kh * kw, _stream())


def csr_group_aggregate(support: torch.Tensor, op, att: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
                        slope: float = 1.0, normalize: bool = False) -> torch.Tensor:
    """DGP graph propagation step: out[i] = act(sum_e att[grp_e] / deg_e * (support[col_e] + bias)); `op` carries the
    merged CSR and the work-item tables (hgr_net_amd.baseline.dgp.GraphOperator)."""
    assert support.dtype == out.dtype == att.dtype == torch.float32 and support.stride(1) == 1 and out.stride(1) == 1
    assert support.shape[0] == out.shape[0] == op.n and support.shape[1] == out.shape[1] and att.numel() == op.D
    c = support.shape[1]
    _lib.call("hgr_csr_group_aggregate", _dev(support), support.stride(0), _dev(op.item_row), _dev(op.item_e0), _dev(op.item_e1),
              _dev(op.item_slot), op.item_row.numel(), _dev(op.col), _dev(op.inv_deg), _dev(op.grp), _dev(att), op.D, _dev(bias),
              _dev(op.split_row), _dev(op.split_slot0), _dev(op.split_n), op.split_row.numel(), _dev(op.partial(c)), _dev(out),
              out.stride(0), c, slope, 1 if normalize else 0, _stream())
    return out


def dot_f32(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    """out[0] (+)= alpha * sum(a * b) for contiguous fp32 tensors of equal size."""
    assert a.dtype == b.dtype == out.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    _lib.call("hgr_dot_f32", _dev(a), _dev(b), a.numel(), _dev(out), alpha, 1 if accumulate else 0, _stream())
    return out


def embed_scatter_add(tokens: torch.Tensor, dx: torch.Tensor, dtable: torch.Tensor, l: int) -> None:
    assert tokens.dtype == torch.int64 and tokens.stride(1) == 1 and dx.is_contiguous() and dtable.is_contiguous()
    _lib.call("hgr_embed_scatter_add", _dev(tokens), tokens.stride(0), _dev(dx), _dev(dtable), tokens.shape[0], l, dtable.shape[1], dtable.shape[0], _stream())


def rows_axpy(dst: torch.Tensor, src: torch.Tensor, dst_mul: int = 1, dst_idx: Optional[torch.Tensor] = None, alpha: float = 1.0) -> None:
    assert dst.dtype == src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
    _lib.call("hgr_rows_axpy", _dev(dst), dst_mul, _dev(dst_idx), _dev(src), src.shape[0], src.shape[1], alpha, _stream())


def ctx_splice(x: torch.Tensor, ctx: torch.Tensor, pos: torch.Tensor, n: int, l: int) -> torch.Tensor:
    assert x.dtype == ctx.dtype == torch.float32 and x.is_contiguous() and ctx.is_contiguous()
    _lib.call("hgr_ctx_splice", _dev(x), _dev(ctx), _dev(pos), n, l, x.shape[1], ctx.shape[0], _stream())
    return x


def ctx_splice_bwd(dx: torch.Tensor, dctx: torch.Tensor, n: int, l: int) -> None:
    assert dx.dtype == dctx.dtype == torch.float32 and dx.is_contiguous() and dctx.is_contiguous()
    _lib.call("hgr_ctx_splice_bwd", _dev(dx), _dev(dctx), n, l, dx.shape[1], dctx.shape[0], _stream())


def rows_gather(src: torch.Tensor, idx: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    assert src.dtype == out.dtype == torch.float32 and src.is_contiguous() and out.is_contiguous() and idx.dtype == torch.int32
    _lib.call("hgr_rows_gather", _dev(out), _dev(src), _dev(idx), out.shape[0], out.shape[1], _stream())
    return out


def sumsq(x: torch.Tensor, out: torch.Tensor) -> None:
    assert x.dtype == torch.float32 and x.is_contiguous()
    _lib.call("hgr_sumsq", _dev(x), x.numel(), _dev(out), _stream())


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8,
          wd: float = 0.0, sumsq_total: Optional[torch.Tensor] = None, max_norm: float = 1.0, grad_scale: float = 1.0) -> None:
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == g.dtype == m.dtype == v.dtype == torch.float32
    WEIGHTS_GEN[0] += 1
    _lib.call("hgr_adamw", _dev(p), _dev(g), _dev(m), _dev(v), p.numel(), lr, betas[0], betas[1], eps, wd, step, _dev(sumsq_total), max_norm, grad_scale, _stream())
