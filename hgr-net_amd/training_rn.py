"""ModifiedResNet image tower (clip/model.py:94-150) in the OM training step: forward with saved activations and the
hand-written backward chain.  The reference's README trains `--arch RN50`; autograd does this for it.

Layout and arithmetic follow the inference engine (clip/model.py `_rn_forward`): 16-bit NHWC activations, every
conv + BatchNorm pair folded into one convolution (BatchNorm uses its running statistics even in training,
model/clip_tree.py:46), 1x1 convolutions = `hgr_gemm_nt`, 3x3 = the implicit-GEMM kernel.  Backward, per folded conv:

  * dX: 1x1 -> NT product against the transposed weight; 3x3 -> `hgr_conv3x3_nhwc_plain` with the flipped, in/out
    swapped weight (all 3x3 convolutions that need a data gradient have stride 1);
  * dW' = dY^T . Xcol is a product with a tiny output and a reduction over every pixel of the batch: both operands are
    transposed (3x3: `hgr_im2col3x3_t` builds Xcol^T from X^T, nine shifted masked copies), `hgr_gemm_nt_splitk`
    spreads the reduction over the chip and `hgr_colsum` adds the fp32 partials; db' = column sum of dY;
  * `hgr_bn_unfold_grad` maps (dW', db') to conv.weight / bn.weight / bn.bias gradients once per step.

The attention pool runs as a packed-QKV self-attention over all 1 + HW tokens (`hgr_mha` / `hgr_mha_bwd`; only token
0's output is used, so its gradient is the only non-zero row of d_out).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import EPI_BIAS, EPI_BIAS_ADD16_RELU, EPI_BIAS_RELU


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _grad(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p.data, dtype=torch.float32)
    return p.grad


class _ConvBN:
    """One conv + BatchNorm pair in training form: folded 16-bit weight, fp32 folded bias, folded-gradient accumulators."""

    def __init__(self, conv, bn, dt):
        w = conv.weight
        self.conv, self.bn, self.dt = conv, bn, dt
        self.cout, self.cin, kh, kw = w.shape
        self.khw = kh * kw
        self.k = self.cin * self.khw
        self.kp = _pad64(self.k)
        dev = w.device
        self.w16 = torch.empty(self.cout, self.kp, dtype=dt, device=dev)
        self.b32 = torch.empty(self.cout, dtype=torch.float32, device=dev)
        ops.bn_fold(w.data.contiguous(), bn, self.w16, self.b32)
        self.gw = torch.zeros(self.cout, self.kp, dtype=torch.float32, device=dev)
        self.gb = torch.zeros(self.cout, dtype=torch.float32, device=dev)
        self._wt = None
        self._wflip = None

    def wt16(self) -> torch.Tensor:
        """[Cin, pad64(Cout)]: the 1x1 weight transposed, operand of dX = dY . W'."""
        if self._wt is None:
            self._wt = torch.zeros(self.cin, _pad64(self.cout), dtype=self.dt, device=self.w16.device)
            ops.transpose16(self.w16[:, : self.cin], self._wt)
        return self._wt

    def wflip16(self) -> torch.Tensor:
        """[Cin, pad64(9 Cout)] in (ky, kx, co) order: the 3x3 weight rotated by 180 degrees with in/out swapped."""
        if self._wflip is None:
            w = self.w16[:, : self.k].view(self.cout, 3, 3, self.cin).flip(1, 2).permute(3, 1, 2, 0).reshape(self.cin, 9 * self.cout)
            self._wflip = torch.zeros(self.cin, _pad64(9 * self.cout), dtype=self.dt, device=w.device)
            self._wflip[:, : 9 * self.cout] = w
        return self._wflip

    def finish(self) -> None:
        ops.bn_unfold_grad(self.gw, self.gb, self.conv.weight.data.contiguous(), self.bn, _grad(self.conv.weight), _grad(self.bn.weight),
                           _grad(self.bn.bias))


class _Block:
    def __init__(self, blk, dt):
        self.c1, self.c2, self.c3 = _ConvBN(blk.conv1, blk.bn1, dt), _ConvBN(blk.conv2, blk.bn2, dt), _ConvBN(blk.conv3, blk.bn3, dt)
        self.down = _ConvBN(blk.downsample[1], blk.downsample[2], dt) if blk.downsample is not None else None
        self.stride = blk.stride
        self.planes = blk.conv1.weight.shape[0]


class RNTower:
    """Training-mode ModifiedResNet: `fwd(image) -> (feat fp32 [B, D], saves)`, `bwd(dfeat, saves)` adds every parameter gradient."""

    def __init__(self, visual, dt: torch.dtype, scratch):
        self.v, self.dt, self.scratch = visual, dt, scratch
        self.dev = visual.conv1.weight.device
        width = visual.conv3.weight.shape[0]
        if width % 64 or (width & (width - 1)):
            raise NotImplementedError(f"RN width {width}: the training tower handles widths that are powers of two >= 64 (RN50 / RN101); "
                                      "RN50x4 / RN50x16 run the zero-shot path only (clip/model.py pads their channels)")
        self.width = width
        self.stem = [_ConvBN(visual.conv1, visual.bn1, dt), _ConvBN(visual.conv2, visual.bn2, dt), _ConvBN(visual.conv3, visual.bn3, dt)]
        self.blocks = [_Block(b, dt) for li in (1, 2, 3, 4) for b in getattr(visual, f"layer{li}")]
        a = visual.attnpool
        e = a.q_proj.weight.shape[0]
        self.e, self.heads = e, a.num_heads
        if e != 64 * a.num_heads:
            raise NotImplementedError("attention pool: head width must be 64")
        w_in = torch.cat([a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data], 0).contiguous()      # [3E, E]
        self.w_in16 = torch.empty(3 * e, e, dtype=dt, device=self.dev)
        ops.cast16(w_in, self.w_in16)
        self.b_in = torch.cat([a.q_proj.bias.data, a.k_proj.bias.data, a.v_proj.bias.data], 0).contiguous()
        self.w_in_t = torch.zeros(e, 3 * e, dtype=dt, device=self.dev)                                              # [E, 3E]
        ops.transpose16(self.w_in16, self.w_in_t)
        self.w_c16 = torch.empty(a.c_proj.weight.shape, dtype=dt, device=self.dev)
        ops.cast16(a.c_proj.weight.data.contiguous(), self.w_c16)
        self.w_c_t = torch.zeros(e, _pad64(self.w_c16.shape[0]), dtype=dt, device=self.dev)                         # [E, D]
        ops.transpose16(self.w_c16, self.w_c_t)
        self._buf: Dict[str, torch.Tensor] = {}

    # -- buffers / generic pieces ----------------------------------------------------------------------
    def _tmp(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        t = self._buf.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.dev)
            self._buf[name] = t
        v = t[:n].view(*shape)
        if zero:
            v.zero_()
        return v

    def _wgrad(self, gw: torch.Tensor, dy16: torch.Tensor, xcol_t: torch.Tensor, m: int, gb: Optional[torch.Tensor] = None) -> None:
        """gw [Cout, >= K] += dy16[:m]^T . xcol_t^T   (xcol_t [K, Mp], zero beyond column m);  gb [Cout] += column sums of dy16
        from the same pass that transposes it."""
        cout, (k, mp) = dy16.shape[1], xcol_t.shape
        dyt = self._tmp("dyt", (cout, mp), self.dt, zero=mp != m)
        if gb is not None and dy16.stride(0) % 8 == 0 and dy16.data_ptr() % 16 == 0:
            ops.transpose16_colsum(dy16, dyt, gb, self.scratch(((dy16.shape[0] + 63) // 64) * cout), accumulate=True)
        else:
            ops.transpose16(dy16, dyt)
            if gb is not None:
                self._bias_grad(gb, dy16)
        if cout >= 256 and k >= 256:                          # 256^2 tiles, one workgroup per CU
            tiles = -(-cout // 256) * -(-k // 256)
            s = max(1, min(mp // 128, -(-256 // tiles)))
        else:
            tiles = -(-cout // 128) * -(-k // 128)
            s = max(1, min(mp // 64, -(-768 // tiles)))
        kc = _pad64(-(-mp // s))
        s = -(-mp // kc)
        part = self._tmp("part", (s, cout * k), torch.float32)
        ops.gemm_nt_splitk(dyt, xcol_t, part, kc)
        if gw.shape[1] == k:
            ops.colsum(part, gw.view(-1), self.scratch(cout * k), accumulate=True)
        else:                                                # accumulator is K-padded: reduce, then add the live columns
            red = self._tmp("red", (cout, k), torch.float32)
            ops.colsum(part, red.view(-1), self.scratch(cout * k), accumulate=False)
            gw[:, :k].add_(red)

    def _bias_grad(self, gb: torch.Tensor, dy16: torch.Tensor) -> None:
        m, n = dy16.shape
        ops.colsum(dy16, gb, self.scratch(((m + 511) // 512) * n), accumulate=True)

    def _reduce(self, part: torch.Tensor, s: int, gw: torch.Tensor, cout: int, k: int) -> None:
        """gw [Cout, >= K] += sum of the s fp32 slices part [s, Cout * K]."""
        if gw.shape[1] == k and s > 1:
            ops.colsum(part, gw.view(-1), self.scratch(cout * k), accumulate=True)
            return
        if s > 1:
            red = self._tmp("red", (cout, k), torch.float32)
            ops.colsum(part, red.view(-1), self.scratch(cout * k), accumulate=False)
        else:
            red = part[0].view(cout, k)
        gw[:, :k].add_(red)

    def _wgrad_tn(self, gw: torch.Tensor, dy16: torch.Tensor, x16: torch.Tensor, gb: Optional[torch.Tensor], conv: Optional[tuple] = None) -> bool:
        """Weight (and bias) gradient from the operands as stored: dW = dY^T X (1x1) or dY^T im2col(X) (conv = (b, h, c))."""
        from .training import WGRAD_TN
        m, cout = dy16.shape
        k = 9 * conv[2] if conv else x16.shape[1]
        if not WGRAD_TN or cout % 8 or k % 8 or dy16.stride(0) % 8 or dy16.data_ptr() % 16 or x16.data_ptr() % 16 or not x16.is_contiguous():
            return False
        s = ops.splitk_slices(-(-cout // 128) * -(-k // 128), m) if conv else ops.tn_slices(cout, k, m)
        kc = _pad64(-(-m // s))
        s = -(-m // kc)
        part = self._tmp("part", (s, cout * k), torch.float32)
        if conv:
            ops.conv3x3_wgrad_splitk(dy16, x16, part, conv[0], conv[1], conv[1], conv[2], kc)
        else:
            ops.gemm_tn_splitk(dy16, x16, part, kc)
        self._reduce(part, s, gw, cout, k)
        if gb is not None:
            self._bias_grad(gb, dy16)
        return True

    def _conv1x1_bwd(self, c: _ConvBN, dy16: torch.Tensor, x16: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        m = dy16.shape[0]
        if not self._wgrad_tn(c.gw, dy16, x16, c.gb):
            mp = _pad64(m)
            xt = self._tmp("xt", (c.cin, mp), self.dt, zero=mp != m)
            ops.transpose16(x16, xt)
            self._wgrad(c.gw, dy16, xt, m, c.gb)
        if not need_dx:
            return None
        dx = torch.empty(m, c.cin, dtype=self.dt, device=self.dev)
        wt = c.wt16()
        ops.gemm_nt(dy16, wt[:, : c.cout] if wt.shape[1] != c.cout else wt, dx, n=c.cin)
        return dx

    def _conv3x3_bwd(self, c: _ConvBN, dy16: torch.Tensor, x16: torch.Tensor, b: int, h: int, need_dx: bool = True) -> Optional[torch.Tensor]:
        m = dy16.shape[0]
        if not self._wgrad_tn(c.gw, dy16, x16, c.gb, conv=(b, h, c.cin)):
            mp = _pad64(m)
            xt = self._tmp("xt", (c.cin, mp), self.dt, zero=mp != m)
            ops.transpose16(x16, xt)
            col_t = self._tmp("colt", (9 * c.cin, mp), self.dt)
            ops.im2col3x3_t(xt, col_t, b, h, h)
            self._wgrad(c.gw, dy16, col_t, m, c.gb)
        if not need_dx:
            return None
        dx = torch.empty(m, c.cin, dtype=self.dt, device=self.dev)
        ops.conv3x3_plain(dy16, c.wflip16(), dx, b, h, h, c.cout)
        return dx

    # -- forward -----------------------------------------------------------------------------------------
    def fwd(self, image: torch.Tensor):
        v, dt, dev = self.v, self.dt, self.dev
        image = image.float().contiguous()
        b, _, r, _ = image.shape
        h = (r - 1) // 2 + 1
        m = b * h * h
        s1, s2, s3 = self.stem
        col = torch.empty(m, 64, dtype=dt, device=dev)
        ops.stem_im2col(image, col)
        a1 = torch.empty(m, s1.cout, dtype=dt, device=dev)
        ops.gemm_nt(col, s1.w16, a1, bias=s1.b32, epilogue=EPI_BIAS_RELU)
        a2 = torch.empty(m, s2.cout, dtype=dt, device=dev)
        ops.conv3x3_nhwc(a1, s2.w16, s2.b32, a2, b, h, h, s1.cout)
        a3 = torch.empty(m, s3.cout, dtype=dt, device=dev)
        ops.conv3x3_nhwc(a2, s3.w16, s3.b32, a3, b, h, h, s2.cout)
        hs = h
        h //= 2
        x = torch.empty(b * h * h, self.width, dtype=dt, device=dev)
        ops.avgpool2_nhwc(a3, x, b, 2 * h, 2 * h, self.width)
        saves = dict(b=b, hs=hs, col=col, a1=a1, a2=a2, a3=a3, blocks=[])
        cin = self.width
        for k in self.blocks:
            m = b * h * h
            pl = k.planes
            t1 = torch.empty(m, pl, dtype=dt, device=dev)
            ops.gemm_nt(x, k.c1.w16, t1, bias=k.c1.b32, epilogue=EPI_BIAS_RELU)
            t2 = torch.empty(m, pl, dtype=dt, device=dev)
            ops.conv3x3_nhwc(t1, k.c2.w16, k.c2.b32, t2, b, h, h, pl)
            ho = h // k.stride
            mo = b * ho * ho
            t2p, xin = t2, x
            if k.stride > 1:
                t2p = torch.empty(mo, pl, dtype=dt, device=dev)
                ops.avgpool2_nhwc(t2, t2p, b, h, h, pl)
                if k.down is not None:
                    xin = torch.empty(mo, cin, dtype=dt, device=dev)
                    ops.avgpool2_nhwc(x, xin, b, h, h, cin)
            idn = xin
            if k.down is not None:
                idn = torch.empty(mo, 4 * pl, dtype=dt, device=dev)
                ops.gemm_nt(xin, k.down.w16, idn, bias=k.down.b32, epilogue=EPI_BIAS)
            out = torch.empty(mo, 4 * pl, dtype=dt, device=dev)
            ops.gemm_nt(t2p, k.c3.w16, out, bias=k.c3.b32, residual=idn, epilogue=EPI_BIAS_ADD16_RELU)
            saves["blocks"].append(dict(x=x, t1=t1, t2=t2, t2p=t2p, xin=xin, out=out, h=h, cin=cin))
            x, cin, h = out, 4 * pl, ho
        # attention pool as packed-QKV self-attention over the 1 + h*h tokens; token 0 is the output
        e, l = self.e, h * h + 1
        a = v.attnpool
        tok = torch.empty(b * l, e, dtype=dt, device=dev)
        ops.attnpool_tokens(x, a.positional_embedding.data, tok, b, h, e)
        qkv = torch.empty(b * l, 3 * e, dtype=dt, device=dev)
        ops.gemm_nt(tok, self.w_in16, qkv, bias=self.b_in, epilogue=EPI_BIAS)
        att = torch.empty(b * l, e, dtype=dt, device=dev)
        ops.mha(qkv, att, b, l, self.heads, False)
        o16 = att.view(b, l * e)[:, :e]                                     # token-0 rows, lda = l*e
        feat = torch.empty(b, v.output_dim, dtype=torch.float32, device=dev)
        ops.gemm_nt(o16, self.w_c16, feat, bias=a.c_proj.bias.data, epilogue=EPI_BIAS)
        saves.update(xlast=x, hlast=h, tok=tok, qkv=qkv, att=att, l=l)
        return feat, saves

    # -- backward ----------------------------------------------------------------------------------------
    def bwd(self, dfeat: torch.Tensor, s: dict) -> None:
        v, dt, dev = self.v, self.dt, self.dev
        a = v.attnpool
        b, l, e, h = s["b"], s["l"], self.e, s["hlast"]
        d_out = v.output_dim
        # c_proj: feat = att0 . Wc^T + bc
        o32 = self._tmp("o32", (b, e), torch.float32)
        o16 = s["att"].view(b, l * e)[:, :e]
        o32.copy_(o16)                                                         # tiny [B, E] cast (plumbing)
        ops.matmul_f32(dfeat.t(), o32, _grad(a.c_proj.weight), accumulate=True)                 # dWc += dfeat^T att0
        ops.colsum(dfeat, _grad(a.c_proj.bias), self.scratch(((b + 511) // 512) * d_out), accumulate=True)
        d16 = torch.empty(b, _pad64(d_out), dtype=dt, device=dev) if d_out % 64 else torch.empty(b, d_out, dtype=dt, device=dev)
        if d_out % 64:
            d16.zero_()
            d16[:, :d_out] = dfeat
        else:
            ops.cast16(dfeat.contiguous(), d16)
        datt = torch.zeros(b * l, e, dtype=dt, device=dev)                     # only token 0 receives gradient
        ops.gemm_nt(d16, self.w_c_t, datt.view(b, l * e)[:, :e], n=e)          # d att0 = dfeat . Wc  -> rows b*l of datt
        dqkv = torch.empty(b * l, 3 * e, dtype=dt, device=dev)
        ops.mha_bwd(s["qkv"], s["att"], datt, dqkv, b, l, self.heads, False)
        # in-projection: qkv = tok . Win^T + bin   (q, k, v are separate parameters: scatter the packed gradient)
        m = b * l
        mp = _pad64(m)
        gwin = self._tmp("gwin", (3 * e, e), torch.float32, zero=True)
        gbin = self._tmp("gbin", (3 * e,), torch.float32, zero=True)
        if not self._wgrad_tn(gwin, dqkv, s["tok"], gbin):
            tok_t = self._tmp("xt", (e, mp), dt, zero=mp != m)
            ops.transpose16(s["tok"], tok_t)
            self._wgrad(gwin, dqkv, tok_t, m, gbin)
        for i, lin in enumerate((a.q_proj, a.k_proj, a.v_proj)):
            _grad(lin.weight).add_(gwin[i * e:(i + 1) * e])
            _grad(lin.bias).add_(gbin[i * e:(i + 1) * e])
        dtok = torch.empty(m, e, dtype=dt, device=dev)
        ops.gemm_nt(dqkv, self.w_in_t, dtok, n=e)
        ops.colsum(dtok.view(b, l * e), _grad(a.positional_embedding).view(-1), self.scratch(((b + 511) // 512) * l * e), accumulate=True)
        dx = torch.empty(b * h * h, e, dtype=dt, device=dev)
        ops.attnpool_tokens_bwd(dtok, dx, b, h * h, e)
        # bottlenecks, last to first
        for k, sv in zip(reversed(self.blocks), reversed(s["blocks"])):
            hh, cin, pl = sv["h"], sv["cin"], k.planes
            ho = hh // k.stride
            g = ops.relu_bwd16(dx, sv["out"])                                 # out = relu(c3(t2p) + idn)
            d_t2p = self._conv1x1_bwd(k.c3, g, sv["t2p"])
            if k.down is not None:
                d_xin = self._conv1x1_bwd(k.down, g, sv["xin"])
                if k.stride > 1:
                    d_skip = torch.empty(b * hh * hh, cin, dtype=dt, device=dev)
                    ops.avgpool2_bwd_nhwc(d_xin, d_skip, b, hh, hh, cin)
                else:
                    d_skip = d_xin
            else:
                d_skip = g
            if k.stride > 1:
                d_t2 = torch.empty(b * hh * hh, pl, dtype=dt, device=dev)
                ops.avgpool2_bwd_nhwc(d_t2p, d_t2, b, hh, hh, pl)
            else:
                d_t2 = d_t2p
            ops.relu_bwd16(d_t2, sv["t2"])
            d_t1 = self._conv3x3_bwd(k.c2, d_t2, sv["t1"], b, hh)
            ops.relu_bwd16(d_t1, sv["t1"])
            d_main = self._conv1x1_bwd(k.c1, d_t1, sv["x"])
            dx = ops.add16(d_main, d_skip)
        # stem: avgpool, conv3, conv2, conv1 (the image needs no gradient)
        hs = s["hs"]
        d_a3 = torch.empty(b * hs * hs, self.width, dtype=dt, device=dev)
        ops.avgpool2_bwd_nhwc(dx, d_a3, b, hs, hs, self.width)
        s1, s2, s3 = self.stem
        ops.relu_bwd16(d_a3, s["a3"])
        d_a2 = self._conv3x3_bwd(s3, d_a3, s["a2"], b, hs)
        ops.relu_bwd16(d_a2, s["a2"])
        d_a1 = self._conv3x3_bwd(s2, d_a2, s["a1"], b, hs)
        ops.relu_bwd16(d_a1, s["a1"])
        m = b * hs * hs
        mp = _pad64(m)
        if not self._wgrad_tn(s1.gw, d_a1, s["col"], s1.gb):
            col_t = self._tmp("xt", (64, mp), dt, zero=mp != m)
            ops.transpose16(s["col"], col_t)
            self._wgrad(s1.gw, d_a1, col_t, m, s1.gb)
        for c in self.stem:
            c.finish()
        for k in self.blocks:
            for c in (k.c1, k.c2, k.c3, k.down):
                if c is not None:
                    c.finish()
