"""GPU: backward / optimizer kernels of the OM training step against torch autograd on the CPU (fp32
reference of the same op - the reference's training arithmetic is torch autograd)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hgr_net_amd import ops, synth
from oracle import clip_ref

DEV = "cuda"
DTS = [torch.bfloat16, torch.float16]


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((scale * synth.normal(seed, "tr", int(np.prod(shape)))).astype(np.float32).reshape(shape))


@pytest.mark.parametrize("dt", DTS)
def test_transpose_colsum_cast(dt):
    r, c = 300, 200
    x = _rand((r, c), 1).to(dt)
    ld = (r + 63) // 64 * 64
    out = torch.zeros(c, ld, dtype=dt, device=DEV)
    ops.transpose16(x.to(DEV), out)
    assert torch.equal(out[:, :r].cpu(), x.t()) and (out[:, r:] == 0).all()
    scratch = torch.empty(4 * c, dtype=torch.float32, device=DEV)
    acc = torch.ones(c, dtype=torch.float32, device=DEV)
    ops.colsum(x.to(DEV), acc, scratch, accumulate=True, alpha=0.5)
    assert torch.allclose(acc.cpu(), 1 + 0.5 * x.float().sum(0), rtol=1e-5, atol=1e-4)
    xf = _rand((1100, 96), 2)
    o2 = torch.empty(96, dtype=torch.float32, device=DEV)
    ops.colsum(xf.to(DEV), o2, torch.empty(3 * 96, dtype=torch.float32, device=DEV), accumulate=False)
    assert torch.allclose(o2.cpu(), xf.sum(0), rtol=1e-5, atol=1e-4)
    y = torch.empty(1100, 96, dtype=dt, device=DEV)
    ops.cast16(xf.to(DEV), y)
    assert torch.equal(y.cpu(), xf.to(dt))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,cols", [(64, 64), (300, 200), (12800, 768), (5000, 40), (100003, 64), (200000, 256)])
def test_transpose_with_fused_column_sums(dt, rows, cols):
    """dY^T for the weight gradient and colsum(dY) for the bias gradient from one pass: the transposition is exact, the
    sums are exact on integer data (any order) and match an fp64 sum of random data to fp32 rounding."""
    gen = torch.Generator().manual_seed(rows * 7 + cols)
    ldc = (cols + 7) // 8 * 8
    xi = torch.zeros(rows, ldc, dtype=dt)
    xi[:, :cols] = torch.randint(-3, 4, (rows, cols), generator=gen).to(dt)
    ld = (rows + 63) // 64 * 64
    scratch = torch.empty((rows + 63) // 64 * cols, dtype=torch.float32, device=DEV)
    for x, exact in ((xi, True), (_rand((rows, ldc), 9).to(dt), False)):
        yt = torch.zeros(cols, ld, dtype=dt, device=DEV)
        acc = torch.full((cols,), 1.5, dtype=torch.float32, device=DEV)
        ops.transpose16_colsum(x.to(DEV)[:, :cols], yt, acc, scratch, accumulate=True)
        assert torch.equal(yt[:, :rows].cpu(), x[:, :cols].t()) and (yt[:, rows:] == 0).all()
        want = x[:, :cols].double().sum(0)
        if exact:
            assert torch.equal(acc.cpu(), (want + 1.5).float())
        else:
            assert (acc.cpu().double() - 1.5 - want).abs().max() < 1e-5 * rows ** 0.5 * 4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,cols", [(16384, 64), (40000, 40), (70001, 192), (17000, 1024), (1605632, 64)])
def test_colsum_many_bands(dt, rows, cols):
    """Bias gradients over 10^4..10^6 pixels: >= 32 bands of 512 rows take the wave-parallel final stage.  Integer values,
    so any summation order gives the same fp32 result and the comparison is exact; run twice for reproducibility."""
    gen = torch.Generator().manual_seed(rows + cols)
    x = torch.randint(-3, 4, (rows, cols), generator=gen).to(dt)
    want = x.double().sum(0).float()
    scratch = torch.empty((rows + 511) // 512 * cols, dtype=torch.float32, device=DEV)
    out = torch.full((cols,), 2.0, dtype=torch.float32, device=DEV)
    ops.colsum(x.to(DEV), out, scratch, accumulate=True)
    assert torch.equal(out.cpu(), want + 2.0)
    xf = x.float().to(DEV)
    o2 = torch.empty(cols, dtype=torch.float32, device=DEV)
    ops.colsum(xf, o2, scratch, accumulate=False, alpha=0.5)
    assert torch.equal(o2.cpu(), 0.5 * want)


@pytest.mark.parametrize("dt", DTS)
def test_gemm_accumulate_and_backward_forms(dt):
    """dX = dY W and dW += dY^T X through hgr_gemm_nt on transposed operands."""
    m, n, k = 200, 192, 128                         # y[m,n] = x[m,k] w[n,k]^T
    x, w, dy = _rand((m, k), 3).to(dt), _rand((n, k), 4, 0.1).to(dt), _rand((m, n), 5, 0.1).to(dt)
    wt = w.t().contiguous()                          # [k, n]
    dx = torch.empty(m, k, dtype=dt, device=DEV)
    ops.gemm_nt(dy.to(DEV), wt.to(DEV), dx)          # contraction over n (multiple of 64)
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert (dx.float().cpu() - dy.float() @ w.float()).abs().max() < tol
    mp = (m + 63) // 64 * 64
    dyt = torch.zeros(n, mp, dtype=dt, device=DEV)
    xt = torch.zeros(k, mp, dtype=dt, device=DEV)
    ops.transpose16(dy.to(DEV), dyt)
    ops.transpose16(x.to(DEV), xt)
    g0 = _rand((n, k), 6)
    dw = g0.clone().to(DEV)
    ops.gemm_nt(dyt, xt, dw, epilogue=6)             # ACCUM
    assert (dw.cpu() - (g0 + dy.float().t() @ x.float())).abs().max() < 1e-3


@pytest.mark.parametrize("dt", DTS)
def test_quickgelu_fwd_bwd(dt):
    a = _rand((64, 256), 7, 2.0).to(dt)
    du = _rand((64, 256), 8).to(dt)
    u = torch.empty_like(a, device=DEV)
    ops.quickgelu16(a.to(DEV), u)
    af = a.float().requires_grad_(True)
    ref = clip_ref.quick_gelu(af)
    assert (u.float().cpu() - ref.detach()).abs().max() < (3e-2 if dt == torch.bfloat16 else 4e-3)
    ref.backward(du.float())
    da = torch.empty_like(a, device=DEV)
    ops.quickgelu16(a.to(DEV), da, du=du.to(DEV))
    assert (da.float().cpu() - af.grad).abs().max() < (3e-2 if dt == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize("w", [64, 768, 1024])
def test_layernorm_bwd(w):
    rows, L = 333, 3
    x = _rand((rows * L, w), 9, 2.0) + 0.3
    g, b = 1 + 0.1 * _rand((w,), 10), 0.1 * _rand((w,), 11)
    dy = _rand((rows * L, w), 12)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (w,), gr, br, 1e-5).backward(dy)
    dx0 = _rand((rows * L, w), 13)
    dx = dx0.clone().to(DEV)
    dg = torch.zeros(w, device=DEV)
    db = torch.zeros(w, device=DEV)
    scratch = torch.empty(ops.layernorm_bwd_scratch(rows * L, w), dtype=torch.float32, device=DEV)
    ops.layernorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), dx, dg, db, scratch)
    assert torch.allclose(dx.cpu(), dx0 + xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dg.cpu(), gr.grad, rtol=1e-4, atol=2e-3) and torch.allclose(db.cpu(), br.grad, rtol=1e-4, atol=2e-3)
    # gathered rows (EOT / class token): only the source rows receive gradient; 16-bit dy
    idx = torch.from_numpy(synth.randint(1, "i", rows, 0, L).astype(np.int32))
    dy16 = _rand((rows, w), 14).to(torch.bfloat16)
    xr2 = x.clone().requires_grad_(True)
    pick = xr2.view(rows, L, w)[torch.arange(rows), idx.long()]
    torch.nn.functional.layer_norm(pick, (w,), g, b, 1e-5).backward(dy16.float())
    dx2 = torch.zeros(rows * L, w, device=DEV)
    ops.layernorm_bwd(dy16.to(DEV), x.to(DEV), g.to(DEV), dx2, dg, db, scratch, rows=rows, row_mul=L, row_idx=idx.to(DEV))
    assert torch.allclose(dx2.cpu(), xr2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,cols,accumulate", [(1, 4096, True), (5, 3072 * 1024, True), (16, 1000, False), (32, 260, True), (7, 262, True)])
def test_colsum_few_fp32_rows(rows, cols, accumulate):
    """hgr_colsum on a handful of fp32 rows (the split-K slices of a weight gradient): the one-pass kernel (cols % 4 == 0) and the band
    kernels (odd widths) against a double-precision sum, with and without accumulation and alpha."""
    x = _rand((rows, cols), 17 * rows + cols % 1000, 1.0).to(DEV)
    start = _rand((cols,), 19 * rows, 1.0).to(DEV)
    out = start.clone()
    ops.colsum(x, out, torch.empty(max(1 << 16, cols), dtype=torch.float32, device=DEV), accumulate=accumulate, alpha=0.5)
    want = (start.double() if accumulate else 0.0) + 0.5 * x.double().sum(dim=0)
    assert float((out.double() - want).abs().max()) <= 1e-6 * (1.0 + float(x.abs().sum(dim=0).max()))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("n,k,ldy", [(1024, 588, 640), (768, 3072, 3072), (130, 70, 72), (64, 64, 64), (3, 5, 5)])
def test_cast16_transpose_equals_cast16_then_transpose16(dt, n, k, ldy):
    """hgr_cast16_transpose: both outputs carry hgr_cast16's bits (row-major copy with a wider leading dimension, transposed copy with
    padding columns that stay untouched) - vector tiles, ragged edges and the element-wise fallback for odd strides."""
    x = _rand((n, k), 5 * n + k, 0.7).to(DEV)
    y = torch.full((n, ldy), 3.0, dtype=dt, device=DEV)
    ldt = (n + 63) // 64 * 64
    yt = torch.full((k, ldt), 5.0, dtype=dt, device=DEV)
    ops.cast16_transpose(x, y, yt)
    if (n * k) % 4 == 0:
        ref = torch.empty(n, k, dtype=dt, device=DEV)
        ops.cast16(x.contiguous(), ref)
    else:
        ref = x.to(dt)
    assert torch.equal(y[:, :k], ref) and bool((y[:, k:] == 3.0).all())
    assert torch.equal(yt[:, :n], ref.t()) and bool((yt[:, n:] == 5.0).all())


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_bwd_cast_equals_bwd_then_cast16(dt):
    """hgr_layernorm_bwd_cast: dx, dgamma, dbeta as hgr_layernorm_bwd; the 16-bit copy equals hgr_cast16 of the updated dx."""
    rows, w = 300, 768
    x = _rand((rows, w), 71, 1.0).to(DEV)
    dy = _rand((rows, w), 72, 0.5).to(dt).to(DEV)
    g = (_rand((w,), 73, 0.2) + 1.0).to(DEV)
    dx0 = _rand((rows, w), 74, 0.3).to(DEV)
    scratch = torch.empty(ops.layernorm_bwd_scratch(rows, w), device=DEV)
    dx1, dg1, db1 = dx0.clone(), torch.zeros(w, device=DEV), torch.zeros(w, device=DEV)
    ops.layernorm_bwd(dy, x, g, dx1, dg1, db1, scratch)
    c1 = torch.empty(rows, w, dtype=dt, device=DEV)
    ops.cast16(dx1, c1)
    dx2, dg2, db2 = dx0.clone(), torch.zeros(w, device=DEV), torch.zeros(w, device=DEV)
    c2 = torch.full((rows, w), 7.0, dtype=dt, device=DEV)
    ops.layernorm_bwd(dy, x, g, dx2, dg2, db2, scratch, dx16=c2)
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2) and torch.equal(c1, c2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,w", [(300, 768), (5000, 1024), (2, 512)])
def test_layernorm_bwd_cast_colsum_equals_cast_then_colsum(dt, rows, w):
    """hgr_layernorm_bwd_cast_colsum: dx, dgamma, dbeta, dx16 as hgr_layernorm_bwd_cast bit for bit; the accumulated column sums equal
    those of the ROUNDED dx16 (hgr_colsum of it) up to the fp32 order of the adds, on top of what the target already held."""
    x = _rand((rows, w), 81, 1.0).to(DEV)
    dy = _rand((rows, w), 82, 0.5).to(dt).to(DEV)
    g = (_rand((w,), 83, 0.2) + 1.0).to(DEV)
    dx0 = _rand((rows, w), 84, 0.3).to(DEV)
    scratch = torch.empty(ops.layernorm_bwd_scratch(rows, w), device=DEV)
    dx1, dg1, db1 = dx0.clone(), torch.zeros(w, device=DEV), torch.zeros(w, device=DEV)
    c1 = torch.empty(rows, w, dtype=dt, device=DEV)
    ops.layernorm_bwd(dy, x, g, dx1, dg1, db1, scratch, dx16=c1)
    start = _rand((w,), 85, 2.0).to(DEV)
    want = start.clone()
    ops.colsum(c1, want, torch.empty(max(1 << 16, ((rows + 511) // 512) * w), device=DEV), accumulate=True)
    dx2, dg2, db2 = dx0.clone(), torch.zeros(w, device=DEV), torch.zeros(w, device=DEV)
    c2 = torch.full((rows, w), 7.0, dtype=dt, device=DEV)
    got = start.clone()
    ops.layernorm_bwd(dy, x, g, dx2, dg2, db2, scratch, dx16=c2, dx16_colsum=got)
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2) and torch.equal(c1, c2)
    ref64 = start.double() + c1.double().sum(dim=0)
    bound = 1e-6 * float(c1.float().abs().sum(dim=0).max()) + 1e-6
    assert float((got.double() - ref64).abs().max()) <= bound and float((want.double() - ref64).abs().max()) <= bound


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("L,causal", [(5, True), (16, True), (1, True), (12, False), (29, True), (32, True), (32, False), (50, False), (64, False), (37, True),
                                      (77, True), (65, False), (130, True), (257, False), (129, False), (193, False), (65, True), (66, False)])
def test_mha_bwd_vs_autograd(dt, L, causal):
    b, heads = 2, 2
    w = heads * 64
    qkv = _rand((b * L, 3 * w), 20 + L, 0.7).to(dt)
    do = _rand((b * L, w), 21 + L, 0.5).to(dt)
    q = qkv.float().clone().requires_grad_(True)
    qq, kk, vv = q.view(b, L, 3 * w).split(w, dim=-1)
    sh = lambda t: t.reshape(b, L, heads, 64).transpose(1, 2)
    s = (sh(qq) @ sh(kk).transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    o = (torch.softmax(s, -1) @ sh(vv)).transpose(1, 2).reshape(b * L, w)
    o.backward(do.float())
    dqkv = torch.empty(b * L, 3 * w, dtype=dt, device=DEV)
    ops.mha_bwd(qkv.to(DEV), o.detach().to(dt).to(DEV), do.to(DEV), dqkv, b, L, heads, causal)
    tol = 3e-2 if dt == torch.bfloat16 else 4e-3
    assert (dqkv.float().cpu() - q.grad).abs().max() < tol * max(1.0, float(q.grad.abs().max()))
    # with the row statistics the forward kept (hgr_mha_stats / hgr_mha_bwd_stats): the same forward output, and the same gradient up to
    # the last-bit differences of the statistics (the forward sums Q K^T in another MFMA shape than the backward's recomputation)
    st = torch.full((b, heads, L, 2), float("nan"), device=DEV)
    o2 = torch.empty(b * L, w, dtype=dt, device=DEV)
    ops.mha(qkv.to(DEV), o2, b, L, heads, causal, stats=st)
    o1 = torch.empty_like(o2)
    ops.mha(qkv.to(DEV), o1, b, L, heads, causal)
    assert torch.equal(o1, o2) and bool(torch.isfinite(st).all())
    dq2 = torch.empty_like(dqkv)
    ops.mha_bwd(qkv.to(DEV), o.detach().to(dt).to(DEV), do.to(DEV), dq2, b, L, heads, causal, stats=st)
    assert (dq2.float().cpu() - q.grad).abs().max() < tol * max(1.0, float(q.grad.abs().max()))
    assert float((dq2.float() - dqkv.float()).abs().max()) <= (2.0 ** -6 if dt == torch.bfloat16 else 2.0 ** -9) * max(1.0, float(q.grad.abs().max()))
    # hgr_mha_bwd_colsum: the same gradients bit for bit (with and without statistics), and per sequence the column sums of the ROUNDED rows
    for stt, want in ((st, dq2), (None, dqkv)):
        dq3 = torch.full_like(dqkv, 7.0)
        part = torch.full((b, 3 * w), float("nan"), device=DEV)
        ops.mha_bwd(qkv.to(DEV), o.detach().to(dt).to(DEV), do.to(DEV), dq3, b, L, heads, causal, stats=stt, colsum_part=part)
        assert torch.equal(dq3, want)
        ref = want.double().view(b, L, 3 * w).sum(dim=1)
        assert float((part.double() - ref).abs().max()) <= 1e-6 * float(want.float().abs().view(b, L, 3 * w).sum(dim=1).max()) + 1e-7


def test_ce_l2norm_matmul_scatter():
    rows, n, d = 37, 257, 64
    lg = _rand((rows, n), 30, 3.0)
    lab = torch.from_numpy(synth.randint(2, "lab", rows, 0, n).astype(np.int32))
    lr = lg.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lr, lab.long()) * 0.37
    loss.backward()
    loss_rows = torch.empty(rows, device=DEV)
    dl = torch.empty(rows, n, device=DEV)
    ops.ce_rows(lg.to(DEV), lab.to(DEV), loss_rows, dl, gscale=0.37 / rows)
    assert abs(float(loss_rows.mean()) * 0.37 - float(loss)) < 1e-5
    assert torch.allclose(dl.cpu(), lr.grad, rtol=1e-4, atol=1e-6)
    x = _rand((rows, d), 31, 2.0)
    dy = _rand((rows, d), 32)
    xr = x.clone().requires_grad_(True)
    (xr / xr.norm(dim=-1, keepdim=True)).backward(dy)
    dx = torch.empty(rows, d, device=DEV)
    ops.l2norm_bwd(x.to(DEV), dy.to(DEV), dx)
    assert torch.allclose(dx.cpu(), xr.grad, rtol=1e-4, atol=1e-6)
    a, bm = _rand((70, 45), 33), _rand((45, 90), 34)
    out = torch.ones(70, 90, device=DEV)
    ops.matmul_f32(a.to(DEV), bm.to(DEV), out, alpha=2.0, accumulate=True)
    assert torch.allclose(out.cpu(), 1 + 2 * a @ bm, rtol=1e-5, atol=1e-4)
    out2 = torch.empty(45, 45, device=DEV)
    ops.matmul_f32(a.to(DEV).t(), a.to(DEV), out2)                       # transposed view through strides
    assert torch.allclose(out2.cpu(), a.t() @ a, rtol=1e-5, atol=1e-4)
    tok = synth.make_tokens(9, 11, 512)[:, :12].contiguous()
    dxe = _rand((9 * 12, 32), 35)
    tab = torch.zeros(512, 32, device=DEV)
    ops.embed_scatter_add(tok.to(DEV), dxe.to(DEV), tab, 12)
    ref = torch.zeros(512, 32).index_add_(0, tok.reshape(-1), dxe)
    assert torch.allclose(tab.cpu(), ref, rtol=1e-5, atol=1e-5)
    dst = torch.zeros(9 * 12, 32, device=DEV)
    idx = torch.from_numpy(synth.randint(3, "e", 9, 0, 12).astype(np.int32))
    src = _rand((9, 32), 36)
    ops.rows_axpy(dst, src.to(DEV), dst_mul=12, dst_idx=idx.to(DEV), alpha=-1.5)
    want = torch.zeros(9, 12, 32)
    want[torch.arange(9), idx.long()] = -1.5 * src
    assert torch.allclose(dst.cpu().view(9, 12, 32), want)


def test_adamw_and_clip_match_torch():
    p0, g = _rand((1000,), 40), _rand((1000,), 41, 3.0)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-2, weight_decay=0.1)
    p, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    tot = torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        p_ref.grad = g.clone() * step
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gd = (g * step).to(DEV)
        tot.zero_()
        ops.sumsq(gd, tot)
        ops.adamw(p, gd, m, v, lr=1e-2, step=step, wd=0.1, sumsq_total=tot, max_norm=1.0)
    assert torch.allclose(p.cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)


def test_sumsq_is_bit_deterministic_and_accumulates():
    """The global gradient norm must have the same BITS on every data-parallel rank (and run to run): the clip factor multiplies
    every gradient, so a last-bit difference makes the replicas' weights drift apart (found by bench.py's dp_check: gradient buffers
    identical after the all-reduce, weights not).  hgr_sumsq: per-block partials + a fixed-order final sum by the last block; over
    150 M values (the ViT-B/32 CLIP parameter count: 1024 blocks), a ragged size, and accumulation into a non-zero total."""
    g = torch.Generator(device=DEV).manual_seed(9)
    for n in (151_277_313, 1000, 262_144 * 3 + 17):
        x = torch.randn(n, generator=g, device=DEV)
        tots = []
        for rep in range(6):
            tot = torch.full((1,), 0.5 if rep == 5 else 0.0, device=DEV)
            ops.sumsq(x, tot)
            tots.append(float(tot.item()))
        assert len(set(tots[:5])) == 1, tots
        ref = float((x.double() ** 2).sum())
        assert abs(tots[0] - ref) <= 1e-5 * ref and abs(tots[5] - 0.5 - ref) <= 1e-5 * ref, (tots, ref)


# ---- ModifiedResNet tower backward pieces (training_rn.py) ---------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
def test_relu_add_avgpool_attnpool_backward_kernels(dt):
    g = torch.Generator().manual_seed(1)
    b, h, c = 2, 6, 64
    y = torch.randn(b * h * h, c, generator=g).to(dt).to(DEV)
    dy = torch.randn(b * h * h, c, generator=g).to(dt).to(DEV)
    out = ops.relu_bwd16(dy.clone(), y)
    assert torch.equal(out, torch.where(y.float() > 0, dy, torch.zeros_like(dy)))
    s = ops.add16(dy.clone(), y)
    assert torch.equal(s, (dy.float() + y.float()).to(dt))
    # AvgPool2d(2) backward against autograd
    x = torch.randn(b, c, h, h, generator=g, requires_grad=True)
    go = torch.randn(b, c, h // 2, h // 2, generator=g).to(dt).float()
    torch.nn.functional.avg_pool2d(x, 2).backward(go)
    dx = torch.empty(b * h * h, c, dtype=dt, device=DEV)
    ops.avgpool2_bwd_nhwc(go.permute(0, 2, 3, 1).reshape(-1, c).to(dt).to(DEV), dx, b, h, h, c)
    assert torch.equal(dx.float().cpu().view(b, h, h, c), x.grad.permute(0, 2, 3, 1).to(dt).float())
    # attention-pool token assembly backward: tokens = cat(mean, x) + pos
    sp = h * h
    xt = torch.randn(b, sp, c, generator=g, requires_grad=True)
    tok = torch.cat([xt.mean(1, keepdim=True), xt], 1)
    gt = torch.randn(b, sp + 1, c, generator=g).to(dt).float()
    tok.backward(gt)
    dxx = torch.empty(b * sp, c, dtype=dt, device=DEV)
    ops.attnpool_tokens_bwd(gt.to(dt).reshape(-1, c).to(DEV), dxx, b, sp, c)
    assert torch.allclose(dxx.float().cpu().view(b, sp, c), xt.grad, rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("dt", DTS)
def test_conv3x3_data_and_weight_gradients_vs_autograd(dt):
    """dX through hgr_conv3x3_nhwc_plain with the flipped weight, dW through transpose + im2col3x3_t + split-K GEMM
    + column sum, against torch autograd of F.conv2d on the same 16-bit-rounded operands (fp32 math)."""
    g = torch.Generator().manual_seed(2)
    b, h, cin, cout = 3, 10, 64, 128
    x = torch.randn(b, cin, h, h, generator=g).to(dt).float().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dt).float().requires_grad_(True)
    go = torch.randn(b, cout, h, h, generator=g).to(dt).float()
    torch.nn.functional.conv2d(x, w, padding=1).backward(go)
    m = b * h * h
    x16 = x.detach().permute(0, 2, 3, 1).reshape(m, cin).to(dt).to(DEV).contiguous()
    dy16 = go.permute(0, 2, 3, 1).reshape(m, cout).to(dt).to(DEV).contiguous()
    w16 = w.detach().permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(dt).to(DEV).contiguous()          # (ky, kx, ci) order
    # data gradient
    wflip = w16.view(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()
    dx = torch.empty(m, cin, dtype=dt, device=DEV)
    ops.conv3x3_plain(dy16, wflip, dx, b, h, h, cout)
    ref_dx = x.grad.permute(0, 2, 3, 1).reshape(m, cin)
    tol = 2 ** -6 if dt == torch.bfloat16 else 2 ** -9
    assert torch.allclose(dx.float().cpu(), ref_dx, rtol=tol, atol=tol * float(ref_dx.abs().max()))
    # weight gradient
    mp = (m + 63) // 64 * 64
    xt = torch.zeros(cin, mp, dtype=dt, device=DEV)
    ops.transpose16(x16, xt)
    colt = torch.empty(9 * cin, mp, dtype=dt, device=DEV)
    ops.im2col3x3_t(xt, colt, b, h, h)
    unf = torch.nn.functional.unfold(x.detach(), 3, padding=1)                                   # [b, cin*9, h*h], (ci, ky, kx) order
    ref_col = unf.view(b, cin, 9, h * h).permute(2, 1, 0, 3).reshape(9 * cin, m)                  # (tap, ci) rows, pixel columns
    assert torch.equal(colt[:, :m].float().cpu(), ref_col) and not colt[:, m:].any()
    dyt = torch.zeros(cout, mp, dtype=dt, device=DEV)
    ops.transpose16(dy16, dyt)
    for kc in (64, 128, mp):
        s = (mp + kc - 1) // kc
        part = torch.empty(s, cout * 9 * cin, dtype=torch.float32, device=DEV)
        ops.gemm_nt_splitk(dyt, colt, part, kc)
        gw = part.sum(0).view(cout, 9, cin).permute(0, 2, 1).reshape(cout, cin, 3, 3).cpu()
        assert torch.allclose(gw, w.grad, rtol=1e-3, atol=1e-3 * float(w.grad.abs().max())), kc


def test_bn_fold_and_unfold_vs_autograd():
    """hgr_bn_fold equals the inference engine's fold; hgr_bn_unfold_grad equals autograd through
    w' = w * gamma / sigma, b' = beta - mean * gamma / sigma."""
    from hgr_net_amd.clip.model import _fold
    g = torch.Generator().manual_seed(3)
    for cin, k in ((64, 3), (128, 1), (3, 3)):
        cout = 32
        conv = torch.nn.Conv2d(cin, cout, k, bias=False)
        bn = torch.nn.BatchNorm2d(cout)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
            bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
        conv, bn = conv.to(DEV), bn.to(DEV)
        ref_w16, ref_b = _fold(conv, bn, torch.float16)
        kp = ref_w16.shape[1]
        w16 = torch.empty(cout, kp, dtype=torch.float16, device=DEV)
        bias = torch.empty(cout, dtype=torch.float32, device=DEV)
        ops.bn_fold(conv.weight.data.contiguous(), bn, w16, bias)
        assert torch.allclose(w16.float(), ref_w16.float(), rtol=2e-3, atol=1e-6) and torch.allclose(bias, ref_b, rtol=1e-6, atol=1e-7)
        # unfold: autograd reference
        kk = cin * k * k
        gwf = torch.randn(cout, kp, generator=g).to(DEV)
        gbf = torch.randn(cout, generator=g).to(DEV)
        wr = conv.weight.detach().clone().requires_grad_(True)
        ga = bn.weight.detach().clone().requires_grad_(True)
        be = bn.bias.detach().clone().requires_grad_(True)
        sc = ga / torch.sqrt(bn.running_var + bn.eps)
        wf = (wr * sc.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(cout, kk)
        bf = be - bn.running_mean * sc
        ((wf * gwf[:, :kk]).sum() + (bf * gbf).sum()).backward()
        g_w, g_g, g_b = torch.zeros_like(wr), torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
        ops.bn_unfold_grad(gwf, gbf, conv.weight.data.contiguous(), bn, g_w, g_g, g_b)
        assert torch.allclose(g_w, wr.grad, rtol=1e-5, atol=1e-6)
        assert torch.allclose(g_g, ga.grad, rtol=1e-4, atol=1e-5) and torch.allclose(g_b, be.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k,kc", [(512, 384, 1024, 256), (300, 260, 1984, 512), (128, 576, 640, 64), (2048, 512, 896, 128)])
def test_gemm_nt_splitk_partials(dt, m, n, k, kc):
    """Every K slice's partial product (both tile sizes: outputs >= 256 x 256 take the 256^2 kernel), exact on integers."""
    gen = torch.Generator().manual_seed(m + k)
    a = torch.randint(-2, 3, (m, k), generator=gen).float().to(dt).to(DEV)
    w = torch.randint(-1, 2, (n, k), generator=gen).float().to(dt).to(DEV)
    s = (k + kc - 1) // kc
    part = torch.full((s, m * n), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm_nt_splitk(a, w, part, kc)
    for i in range(s):
        ref = a[:, i * kc:(i + 1) * kc].float() @ w[:, i * kc:(i + 1) * kc].float().t()
        assert torch.equal(part[i].view(m, n), ref), i


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,na,nb,kc", [(64, 128, 128, 64), (200, 192, 128, 64), (1000, 40, 264, 256), (12800, 768, 768, 1280),
                                        (777, 136, 72, 128), (4096, 3072, 768, 4096), (130, 8, 8, 64),
                                        (700, 1016, 760, 256), (8192, 1024, 4096, 2048), (333, 256, 256, 64)])     # the last three: 256 x 256 tiles, ragged edges
def test_gemm_tn_splitk_exact(dt, m, na, nb, kc):
    """Weight gradient with untransposed operands: partial[s] = P[slice]^T . Q[slice].  Small-integer operands, so the
    fp32 accumulation is exact in any order: every slice must equal the fp64 product bit for bit (ragged M, Na, Nb;
    strided operands)."""
    gen = torch.Generator().manual_seed(m + na + nb)
    pf = torch.zeros(m, na + 8, dtype=dt)
    pf[:, :na] = torch.randint(-2, 3, (m, na), generator=gen).to(dt)
    qf = torch.randint(-2, 3, (m, nb), generator=gen).to(dt)
    s = (m + kc - 1) // kc
    from hgr_net_amd import _lib
    if na >= 1016 or (na, nb) in ((768, 768), (3072, 768), (256, 256)):
        assert _lib.load().hgr_gemm_tn_tile(na, nb) == 256
    part = torch.full((s, na, nb), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm_tn_splitk(pf.to(DEV)[:, :na], qf.to(DEV), part, kc)
    for i in range(s):
        want = pf[i * kc:(i + 1) * kc, :na].double().t() @ qf[i * kc:(i + 1) * kc].double()
        assert torch.equal(part[i].cpu(), want.float()), i


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,h,w,c,cout,kc", [(2, 8, 8, 64, 64, 64), (3, 7, 9, 32, 40, 128), (1, 14, 14, 128, 128, 64), (5, 12, 12, 8, 16, 256),
                                             (2, 9, 9, 40, 80, 64), (4, 28, 28, 64, 64, 1024)])
def test_conv3x3_wgrad_splitk_vs_autograd(dt, b, h, w, c, cout, kc):
    """dW of a 3x3 / pad 1 / stride 1 convolution from NHWC x and dY, taps gathered by the loader; against autograd's
    conv2d weight gradient on the same 16-bit values (integers: exact)."""
    gen = torch.Generator().manual_seed(b * h + c)
    x = torch.randint(-2, 3, (b, c, h, w), generator=gen).float()
    dy = torch.randint(-2, 3, (b, cout, h, w), generator=gen).float()
    wt = torch.zeros(cout, c, 3, 3, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, padding=1).backward(dy)
    want = wt.grad.permute(0, 2, 3, 1).reshape(cout, 9 * c)                      # (ky, kx, c) order
    xn = x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)
    dyn = dy.permute(0, 2, 3, 1).reshape(b * h * w, cout).contiguous().to(dt).to(DEV)
    s = (b * h * w + kc - 1) // kc
    part = torch.empty(s, cout, 9 * c, dtype=torch.float32, device=DEV)
    ops.conv3x3_wgrad_splitk(dyn, xn, part, b, h, w, c, kc)
    assert torch.equal(part.sum(0).cpu(), want)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [0, 128, 256, 2])
@pytest.mark.parametrize("m,n,k", [(512, 256, 128), (1000, 328, 192), (2560, 3072, 768), (300, 64, 64)])
def test_gemm_qgelu_grad16_epilogue(dt, tile, m, n, k):
    """HGR_EPI_QGELU_GRAD16: C = (A W^T) * g'(pre) in one rounding, on every tile plan (ragged M / N included), against the
    fp32 product times the closed form of d/dx x*sigmoid(1.702x) - and against the two-pass route it replaces
    (GEMM -> 16 bit, hgr_quickgelu16 backward), which rounds twice."""
    a = _rand((m, k), 3 * m + n, 0.5).to(dt)
    w = _rand((n, k), 5 * n + k, 0.2).to(dt)
    pre = _rand((m, n + 8), 7 * m + k, 1.5).to(dt)[:, :n]                      # strided pre-activation
    prev = ops.gemm_set_tile(tile)
    try:
        out = torch.full((m, n), 7.0, dtype=dt, device=DEV)
        ops.gemm_nt(a.to(DEV), w.to(DEV), out, residual=pre.to(DEV), epilogue=ops.EPI_QGELU_GRAD16)
        plain = torch.empty(m, n, dtype=dt, device=DEV)
        ops.gemm_nt(a.to(DEV), w.to(DEV), plain)
    finally:
        ops.gemm_set_tile(prev)
    x = pre.float()
    sg = torch.sigmoid(1.702 * x)
    want = (a.float() @ w.float().t()) * sg * (1 + 1.702 * x * (1 - sg))
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    err = (out.float().cpu() - want).abs()
    assert float((err - eps * want.abs()).max()) < 2e-3, float(err.max())                   # one rounding of the exact value (+ MFMA sum order)
    two = torch.empty(m, n, dtype=dt, device=DEV)
    ops.quickgelu16(pre.contiguous().to(DEV), two, du=plain)
    assert float((out.float() - two.float()).abs().max()) <= 3 * eps * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(512, 256, 128), (1000, 384, 192), (2560, 3072, 768), (77, 128, 128), (33 * 256 + 128, 2048, 256), (4181, 1024, 512)])
def test_gemm_nt_qgelu_grad_colsum_equals_gemm_then_colsum(dt, m, n, k):
    """hgr_gemm_nt_qgelu_grad_colsum: the 16-bit data gradient equals hgr_gemm_nt(HGR_EPI_QGELU_GRAD16) bit for bit, and the sum of the
    per-64-row column sums equals the column sums of that ROUNDED output (full tiles, half tiles of the tail plan, a ragged last panel,
    a launch of fewer rows than one tile).  The sums are fp32 adds of the same values in another order: tolerance, not bits."""
    dy = _rand((m, k), 3 * m + n, 0.5).to(dt).to(DEV)
    wt = _rand((n, k), 5 * n + k, 0.2).to(dt).to(DEV)
    pre = _rand((m, n), 7 * n + m, 1.5).to(dt).to(DEV)
    out = torch.full((m, n), 7.0, dtype=dt, device=DEV)
    units = (m + 63) // 64
    part = torch.full((units, n), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_nt_qgelu_grad_colsum(dy, wt, out, pre, part)
    ref = torch.empty(m, n, dtype=dt, device=DEV)
    ops.gemm_nt(dy, wt, ref, residual=pre, epilogue=ops.EPI_QGELU_GRAD16)
    assert torch.equal(out, ref)
    assert bool(torch.isfinite(part).all()), "every (unit, column) entry must be written"
    # unit by unit against a double-precision sum of the rounded rows
    pad = torch.zeros(units * 64, n, dtype=torch.float64, device=DEV)
    pad[:m] = ref.double()
    want = pad.view(units, 64, n).sum(dim=1)
    scale = float(pad.abs().view(units, 64, n).sum(dim=1).max()) + 1e-30
    assert float((part.double() - want).abs().max()) <= 64 * 2.0 ** -24 * scale
    # and through hgr_colsum into a bias gradient
    gb = torch.zeros(n, dtype=torch.float32, device=DEV)
    ops.colsum(part, gb, torch.empty(max(1 << 16, ((units + 511) // 512) * n), dtype=torch.float32, device=DEV), accumulate=True)
    gb_ref = torch.zeros(n, dtype=torch.float32, device=DEV)
    ops.colsum(ref, gb_ref, torch.empty(max(1 << 16, ((m + 511) // 512) * n), dtype=torch.float32, device=DEV), accumulate=True)
    assert float((gb - gb_ref).abs().max()) <= 1e-5 * float(ref.float().abs().sum(dim=0).max()) + 1e-6


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(512, 256, 128), (1000, 384, 192), (2560, 3072, 768), (77, 128, 128)])
def test_gemm_nt_bias_gelu_dual_equals_two_passes(dt, m, n, k):
    """hgr_gemm_nt_bias_gelu_dual: the pre-activation equals hgr_gemm_nt(+bias) bit for bit and the activation equals
    hgr_quickgelu16 of it bit for bit (ragged last row panel included)."""
    a = _rand((m, k), 3 * m + n, 0.5).to(dt).to(DEV)
    w = _rand((n, k), 5 * n + k, 0.2).to(dt).to(DEV)
    bias = _rand((n,), 11 * n, 0.3).to(DEV)
    pre = torch.full((m, n), 7.0, dtype=dt, device=DEV)
    post = torch.full((m, n), 7.0, dtype=dt, device=DEV)
    ops.gemm_nt_bias_gelu_dual(a, w, pre, post, bias)
    ref = torch.empty(m, n, dtype=dt, device=DEV)
    ops.gemm_nt(a, w, ref, bias=bias, epilogue=ops.EPI_BIAS)
    act = torch.empty_like(ref)
    ops.quickgelu16(ref, act)
    assert torch.equal(pre, ref) and torch.equal(post, act)
