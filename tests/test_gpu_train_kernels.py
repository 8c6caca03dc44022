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


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("L,causal", [(5, True), (16, True), (50, False), (64, False), (37, True), (77, True), (65, False), (130, True), (257, False)])
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
