"""GPU: every libhgr kernel, called through the C ABI, against the CPU oracle / exact references."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hgr_net_amd import ops, synth
from hgr_net_amd._lib import EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL, EPI_NONE
from oracle import clip_ref, tree_ref

DEV = "cuda"
DTS = [torch.bfloat16, torch.float16]


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((scale * synth.normal(seed, "t", int(np.prod(shape)))).astype(np.float32).reshape(shape))


@pytest.mark.parametrize("dt", DTS)
def test_gemm_identity_asymmetric_exact(dt):
    """A = I against an asymmetric integer B: catches a transposed or permuted C write exactly."""
    m = n = 256
    k = 256
    a = torch.eye(m, k, dtype=torch.float32)
    w = (torch.arange(n).view(-1, 1) * 3 + torch.arange(k).view(1, -1) * 7) % 61 - 30.0   # w[n][k], asymmetric
    out = torch.empty(m, n, dtype=torch.float32, device=DEV)
    ops.gemm_nt(a.to(dt).to(DEV), w.to(dt).to(DEV), out)
    assert torch.equal(out.cpu(), a @ w.t())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (1, 1, 64), (200, 300, 128), (130, 129, 192), (64, 2053, 512), (1000, 768, 3072)])
def test_gemm_shapes_edges(dt, m, n, k):
    a = _rand((m, k), 1).to(dt)
    w = _rand((n, k), 2).to(dt)
    ref = a.float() @ w.float().t()
    ld = (n + 3) // 4 * 4
    for ldc in sorted({n, ld, ld + 4}):                     # odd ldc exercises the scalar store path
        out = torch.full((m, ldc), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm_nt(a.to(DEV), w.to(DEV), out, n=n)
        got = out.cpu()
        assert torch.allclose(got[:, :n], ref, rtol=1e-4, atol=2e-4 * k ** 0.5)
        assert (got[:, n:] == 7.0).all()                    # never writes past N


@pytest.mark.parametrize("dt", DTS)
def test_gemm_epilogues(dt):
    m, n, k = 300, 384, 256
    a, w = _rand((m, k), 3).to(dt), _rand((n, k), 4, 0.1).to(dt)
    bias, res = _rand((n,), 5), _rand((m, n), 6)
    base = a.float() @ w.float().t() + bias
    tol = dict(rtol=1e-2, atol=1e-2) if dt == torch.bfloat16 else dict(rtol=2e-3, atol=2e-3)
    out16 = torch.empty(m, n, dtype=dt, device=DEV)
    ops.gemm_nt(a.to(DEV), w.to(DEV), out16, bias=bias.to(DEV), epilogue=EPI_BIAS)
    assert torch.allclose(out16.float().cpu(), base, **tol)
    ops.gemm_nt(a.to(DEV), w.to(DEV), out16, bias=bias.to(DEV), epilogue=EPI_BIAS_QUICKGELU)
    assert torch.allclose(out16.float().cpu(), clip_ref.quick_gelu(base), **tol)
    x = res.clone().to(DEV)                                  # in-place residual accumulate (fp32 stream)
    ops.gemm_nt(a.to(DEV), w.to(DEV), x, bias=bias.to(DEV), residual=x, epilogue=EPI_BIAS_RESIDUAL)
    assert torch.allclose(x.cpu(), base + res, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [0, 128, 256, 2])
@pytest.mark.parametrize("m,n,k", [(4400, 4200, 256), (8192, 4224, 512), (5120, 3072, 320)])
def test_gemm_many_tiles_exact(dt, tile, m, n, k):
    """More 256x256 tiles than CUs (several rounds of workgroups, edge tiles in both dimensions) under every tile
    plan and epilogue.  Small-integer operands make every sum exact in fp32 and in the
    16-bit outputs, so the comparison is equality, and a repeat launch must reproduce it."""
    gen = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=gen).float()
    w = torch.randint(-1, 2, (n, k), generator=gen).float()
    bias = torch.randint(-4, 5, (n,), generator=gen).float()
    res = torch.randint(-8, 9, (m, n), generator=gen).float()
    ad, wd, bd = a.to(dt).to(DEV), w.to(dt).to(DEV), bias.to(DEV)
    base = (ad.float() @ wd.float().t())                    # exact: |sum| <= 2 * 512
    prev = ops.gemm_set_tile(tile)
    try:
        for rep in range(2):
            out32 = torch.empty(m, n, dtype=torch.float32, device=DEV)
            ops.gemm_nt(ad, wd, out32)
            assert torch.equal(out32, base)
            out16 = torch.empty(m, n, dtype=dt, device=DEV)
            ops.gemm_nt(ad, wd, out16, bias=bd, epilogue=EPI_BIAS)
            assert torch.equal(out16, (base + bd).to(dt))
            ops.gemm_nt(ad, wd, out16, bias=bd, epilogue=4)
            assert torch.equal(out16, torch.relu(base + bd).to(dt))
            ops.gemm_nt(ad, wd, out16, bias=bd, epilogue=EPI_BIAS_QUICKGELU)
            ref = clip_ref.quick_gelu(base + bd)
            assert torch.allclose(out16.float(), ref, rtol=2 ** -7 if dt == torch.bfloat16 else 2 ** -10, atol=1e-3)
            x = res.to(DEV)
            ops.gemm_nt(ad, wd, x, bias=bd, residual=x, epilogue=EPI_BIAS_RESIDUAL)
            assert torch.equal(x, base + bd + res.to(DEV))
            idn = res.to(dt).to(DEV)
            ops.gemm_nt(ad, wd, out16, bias=bd, residual=idn, epilogue=5)
            assert torch.equal(out16, torch.relu(base + bd + idn.float()).to(dt))
            acc = res.to(DEV)
            ops.gemm_nt(ad, wd, acc, epilogue=6)
            assert torch.equal(acc, base + res.to(DEV))
    finally:
        ops.gemm_set_tile(prev)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(512, 21841, 512), (200, 5000, 128), (1000, 4099, 192), (1, 4096, 128), (300, 9000, 256), (64, 70000, 64 * 3)])
def test_gemm_few_rows_many_columns_exact(dt, m, n, k):
    """fp32, epilogue-free products with few rows and many columns (the class logits, clip_tree.py:331-333: batch x 21 841):
    small-integer operands make every sum exact, so the comparison with the fp32 product is equality; ragged rows and
    columns, padded and unpadded ldc; never writes past N; on random data the cost-model plan and the pinned 128-tile
    plan agree bit for bit, and so do two launches."""
    gen = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=gen).float()
    w = torch.randint(-1, 2, (n, k), generator=gen).float()
    ad, wd = a.to(dt).to(DEV), w.to(dt).to(DEV)
    base = ad.float() @ wd.float().t()
    for ldc in sorted({(n + 3) // 4 * 4, (n + 63) // 64 * 64 + 64}):
        out = torch.full((m, ldc), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm_nt(ad, wd, out, n=n)
        assert torch.equal(out[:, :n], base)
        assert (out[:, n:] == 7.0).all()
    ar, wr = _rand((m, k), 11).to(dt).to(DEV), _rand((n, k), 12, 0.05).to(dt).to(DEV)
    ld = (n + 63) // 64 * 64
    o1, o2, o3 = (torch.zeros((m, ld), dtype=torch.float32, device=DEV) for _ in range(3))
    ops.gemm_nt(ar, wr, o1, n=n)
    ops.gemm_nt(ar, wr, o2, n=n)
    prev = ops.gemm_set_tile(128)
    try:
        ops.gemm_nt(ar, wr, o3, n=n)
    finally:
        ops.gemm_set_tile(prev)
    assert torch.equal(o1, o2) and torch.equal(o1, o3)
    ref = ar.float().cpu() @ wr.float().cpu().t()
    assert torch.allclose(o1[:, :n].cpu(), ref, rtol=1e-4, atol=2e-4 * k ** 0.5)


def test_gemm_rejects_bad_shapes():
    from hgr_net_amd._lib import HgrError
    a = torch.zeros(8, 96, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(8, 8, dtype=torch.float32, device=DEV)
    with pytest.raises(HgrError):
        ops.gemm_nt(a, a, out)                               # K % 64 != 0 fails loudly, no launch


@pytest.mark.parametrize("w", [64, 512, 768, 1024])
def test_layernorm_and_l2norm(w):
    rows = 37
    x = _rand((rows, w), 7, 3.0) + 0.5
    g, b = 1 + 0.1 * _rand((w,), 8), 0.1 * _rand((w,), 9)
    ref = torch.nn.functional.layer_norm(x, (w,), g, b, 1e-5)
    y32 = torch.empty(rows, w, dtype=torch.float32, device=DEV)
    ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), y32)
    assert torch.allclose(y32.cpu(), ref, rtol=1e-5, atol=1e-5)
    for dt in DTS:
        y = torch.empty(rows, w, dtype=dt, device=DEV)
        ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), y)
        assert torch.equal(y.cpu(), y32.cpu().to(dt)) or torch.allclose(y.float().cpu(), ref, rtol=1e-2, atol=1e-2)
    # strided / indexed rows (ln_post on token 0, ln_final on the EOT row)
    L = 5
    xs = _rand((rows * L, w), 10)
    idx = torch.from_numpy(synth.randint(3, "idx", rows, 0, L).astype(np.int32))
    y = torch.empty(rows, w, dtype=torch.float32, device=DEV)
    ops.layernorm(xs.to(DEV), g.to(DEV), b.to(DEV), y, rows=rows, row_mul=L, row_idx=idx.to(DEV))
    pick = xs.view(rows, L, w)[torch.arange(rows), idx.long()]
    assert torch.allclose(y.cpu(), torch.nn.functional.layer_norm(pick, (w,), g, b, 1e-5), rtol=1e-5, atol=1e-5)
    z32 = torch.empty(rows, w, dtype=torch.float32, device=DEV)
    z16 = torch.empty(rows, w, dtype=torch.bfloat16, device=DEV)
    ops.l2norm_rows(x.to(DEV), y16=z16, y32=z32)
    assert torch.allclose(z32.cpu(), x / x.norm(dim=-1, keepdim=True), rtol=1e-6, atol=1e-7)
    assert torch.equal(z16.cpu(), z32.cpu().to(torch.bfloat16))


def test_vit_embed_ln():
    b, g, w = 3, 49, 768
    pe, cls, pos = _rand((b * g, w), 11), _rand((w,), 12), _rand((g + 1, w), 13)
    ga, be = 1 + 0.1 * _rand((w,), 14), 0.1 * _rand((w,), 15)
    x = torch.empty(b * (g + 1), w, dtype=torch.float32, device=DEV)
    ops.vit_embed_ln(pe.to(DEV), cls.to(DEV), pos.to(DEV), ga.to(DEV), be.to(DEV), x, b, g)
    t = torch.cat([cls.expand(b, 1, w), pe.view(b, g, w)], 1) + pos
    ref = torch.nn.functional.layer_norm(t, (w,), ga, be, 1e-5).view(-1, w)
    assert torch.allclose(x.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("L,causal", [(1, False), (7, True), (16, True), (33, True), (50, False), (77, True), (130, False), (257, False)])
def test_mha_vs_oracle(dt, L, causal):
    b, heads = 3, 2
    w = heads * 64
    qkv = _rand((b * L, 3 * w), 20 + L, 1.0).to(dt)
    out = torch.empty(b * L, w, dtype=dt, device=DEV)
    ops.mha(qkv.to(DEV), out, b, L, heads, causal)
    q, k, v = qkv.float().view(b, L, 3 * w).split(w, dim=-1)
    sh = lambda t: t.reshape(b, L, heads, 64).transpose(1, 2)
    s = (sh(q) @ sh(k).transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    ref = (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(b * L, w)
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert (out.float().cpu() - ref).abs().max() < tol


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("L,q_rows,causal", [(50, 1, False), (50, 20, False), (257, 1, False), (33, 5, True), (16, 16, True)])
def test_mha_leading_query_rows_only(dt, L, q_rows, causal):
    """hgr_mha_rows: the first q_rows query rows of every sequence carry the bits hgr_mha writes, the other rows of `out` are not
    touched (a ViT's last block needs the class token's row alone, clip/model.py:231)."""
    b, heads = 4, 3
    w = heads * 64
    qkv = _rand((b * L, 3 * w), 70 + L, 1.0).to(dt).to(DEV)
    full = torch.empty(b * L, w, dtype=dt, device=DEV)
    ops.mha(qkv, full, b, L, heads, causal)
    part = torch.full((b * L, w), 7.0, dtype=dt, device=DEV)
    ops.mha(qkv, part, b, L, heads, causal, q_rows=q_rows)
    f, p_ = full.view(b, L, w), part.view(b, L, w)
    assert torch.equal(p_[:, :q_rows], f[:, :q_rows])
    assert bool((p_[:, q_rows:] == 7.0).all())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("r,p", [(64, 32), (224, 32), (224, 16), (28, 14)])
def test_im2col_exact(dt, r, p):
    b = 2
    img = _rand((b, 3, r, r), 30)
    k = 3 * p * p
    kp = (k + 63) // 64 * 64
    out = torch.full((b * (r // p) ** 2, kp), 9.0, dtype=dt, device=DEV)
    ops.im2col_patches(img.to(DEV), out, p)
    g = r // p
    ref = img.reshape(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b * g * g, k).to(dt)
    assert torch.equal(out[:, :k].cpu(), ref)
    assert (out[:, k:] == 0).all()


def test_text_embed_and_eot_exact():
    n, ctx, w, vocab = 9, 77, 64, 512
    tok = synth.make_tokens(n, 11, vocab)
    emb, pos = _rand((vocab, w), 31), _rand((ctx, w), 32)
    eot = torch.empty(n, dtype=torch.int32, device=DEV)
    ops.eot_index(tok.to(DEV), eot)
    assert torch.equal(eot.cpu().long(), tok.argmax(-1))
    L = int(eot.max()) + 1
    x = torch.empty(n * L, w, dtype=torch.float32, device=DEV)
    ops.text_embed(tok.to(DEV), emb.to(DEV), pos.to(DEV), x, L)
    assert torch.equal(x.cpu(), (emb[tok[:, :L]] + pos[:L]).view(-1, w))


@pytest.mark.parametrize("n,k", [(40, 20), (999, 20), (21841, 20), (21841, 1)])
def test_topk_rows_exact(n, k):
    rows = 5
    lg = _rand((rows, n), 40 + n, 0.05)
    lg[0, 3] = lg[0, 17]                                     # an exact tie: lowest position first
    ld = (n + 63) // 64 * 64
    buf = torch.zeros(rows, ld, dtype=torch.float32, device=DEV)
    buf[:, :n] = lg.to(DEV)
    cols = torch.from_numpy(np.argsort(synth.uniform(1, "perm", n), kind="stable")[: max(k, n // 2)].astype(np.int32))
    idx, val = ops.topk_rows(buf[:, :n], k, cols=cols.to(DEV), want_values=True)
    for r in range(rows):
        sub = lg[r, cols.long()].numpy()
        want = cols.numpy()[tree_ref.topk_desc(sub, k)]
        assert np.array_equal(idx[r].cpu().numpy(), want)
        assert np.array_equal(val[r].cpu().numpy(), lg[r, torch.from_numpy(want).long()].numpy())
    idx2 = ops.topk_rows(buf[:, :n], k, n_cols=n)            # no subset: all columns
    for r in range(rows):
        assert np.array_equal(idx2[r].cpu().numpy(), tree_ref.topk_desc(lg[r].numpy(), k))


def test_topk_rows_duplicates_and_fallback():
    """heavily duplicated values: exact tie order, and the candidate-overflow fallback path"""
    rows, n, k = 3, 5000, 20
    lg = torch.zeros(rows, n)
    lg[0] = 0.25                                             # all equal -> every element is a candidate (fallback)
    lg[1] = torch.from_numpy((synth.randint(1, "dup", n, 0, 7)).astype(np.float32))   # 7 distinct values
    lg[2, ::2] = 1.0                                         # 2500 ties for the maximum
    idx, val = ops.topk_rows(lg.to(DEV), k, n_cols=n, want_values=True)
    for r in range(rows):
        want = tree_ref.topk_desc(lg[r].numpy(), k)
        assert np.array_equal(idx[r].cpu().numpy(), want)
        assert np.array_equal(val[r].cpu().numpy(), lg[r].numpy()[want])


@pytest.mark.parametrize("n,levels", [(90, 8), (3000, 12), (21841, 12), (500, 20)])
def test_level_argmax_exact(n, levels):
    rows = 4
    lg = _rand((rows, n), 50 + n, 0.05)
    depth = synth.randint(2, "depth", n, 0, levels).astype(np.int32)
    if levels >= 12:
        depth[depth == 5] = 4                                # an empty level: every column carries the -1 filler
    perm = np.argsort(synth.uniform(2, "perm", n), kind="stable").astype(np.int32)
    for cols in (None, perm[: n - n // 3]):
        tr = np.arange(n, dtype=np.int64) if cols is None else cols.astype(np.int64)
        got, top1 = ops.level_argmax(lg.to(DEV), torch.from_numpy(depth).to(DEV), levels,
                                     cols=None if cols is None else torch.from_numpy(cols).to(DEV), n_cols=n, want_top1=True)
        got = got.cpu().numpy()
        for rr in range(rows):
            assert int(top1[rr, 0]) == tr[tree_ref.topk_desc(lg[rr, torch.from_numpy(tr)].numpy(), 1)[0]]
        for l in range(levels):
            same = [int(i) for i in np.nonzero(depth == l)[0]]
            want = tree_ref.level_argmax(lg.numpy(), tr, same, n)
            assert np.array_equal(got[:, l], want), (l, cols is None)


# ---- ModifiedResNet kernels ------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,h,w,c,cout,stride", [(2, 8, 8, 64, 64, 1), (1, 7, 9, 32, 32, 1), (3, 14, 14, 128, 128, 1),
                                                  (2, 16, 16, 8, 16, 1), (2, 12, 12, 64, 128, 2), (1, 5, 5, 512, 512, 1),
                                                  # >= 1024 output pixels and <= 64 channels: the tall 256 x 64 tile variant
                                                  (5, 16, 16, 64, 64, 1), (3, 20, 19, 32, 32, 1), (9, 12, 12, 32, 64, 1),
                                                  (2, 48, 48, 64, 64, 2),
                                                  # channel counts that are not powers of two (RN50x4 / RN50x16 widths):
                                                  # the loader's tap / channel split is a multiply-high division
                                                  (2, 9, 9, 40, 40, 1), (1, 12, 12, 48, 128, 1), (2, 18, 18, 192, 192, 2),
                                                  (1, 9, 9, 320, 320, 1), (1, 6, 6, 640, 640, 1), (2, 24, 24, 24, 48, 1),
                                                  # 32 input channels, stride 1, 32 / 64 outputs: the direct halo-tile kernel (ragged tiles too)
                                                  (2, 16, 16, 32, 32, 1), (1, 33, 17, 32, 64, 1), (3, 40, 48, 32, 32, 1), (2, 5, 70, 32, 64, 1),
                                                  # ... with more tiles than the 512 persistent workgroups (600 / 1300: every workgroup walks 1 - 3 tiles)
                                                  (6, 160, 160, 32, 64, 1), (13, 150, 152, 32, 32, 1),
                                                  # 64 -> 64 channels with >= 4096 pixels: the halo-tile kernel's 8-wave form (54 tiles; 400 tiles on 256 workgroups)
                                                  (6, 40, 42, 64, 64, 1), (20, 70, 60, 64, 64, 1),
                                                  # C % 64 == 0, stride 1, Cout % 128 == 0 and >= 256 tiles of 256 x 128: gemm_nt_duo with the implicit-im2col
                                                  # loader - full tiles only (282), tail plan with half tiles (633 tiles), ragged last row panel + 4 column tiles
                                                  (20, 60, 60, 128, 128, 1), (45, 60, 60, 128, 128, 1), (24, 30, 31, 256, 512, 1),
                                                  # 256^2 tiles with a non-power-of-two C (>= 1024 tiles of 256)
                                                  (40, 80, 80, 192, 192, 1)])
def test_conv3x3_implicit_gemm_vs_conv2d(dt, b, h, w, c, cout, stride):
    x = _rand((b, c, h, w), 60).to(dt)                       # NCHW reference layout
    wt = _rand((cout, c, 3, 3), 61, (2.0 / (9 * c)) ** 0.5).to(dt)
    bias = _rand((cout,), 62, 0.1)
    ref = torch.relu(torch.nn.functional.conv2d(x.float(), wt.float(), bias, stride=stride, padding=1))
    ho, wo = ref.shape[2], ref.shape[3]
    k = 9 * c
    kp = (k + 63) // 64 * 64
    w2 = torch.zeros(cout, kp, dtype=dt)
    w2[:, :k] = wt.permute(0, 2, 3, 1).reshape(cout, k)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(b * ho * wo, cout, dtype=dt, device=DEV)
    ops.conv3x3_nhwc(xn, w2.to(DEV), bias.to(DEV), out, b, h, w, c, stride)
    got = out.float().cpu().view(b, ho, wo, cout).permute(0, 3, 1, 2)
    tol = 3e-2 if dt == torch.bfloat16 else 4e-3
    assert (got - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,h,w,cout", [(2, 16, 16, 64), (1, 34, 18, 64), (3, 40, 48, 32), (2, 6, 70, 64), (5, 176, 160, 64), (6, 160, 162, 32)])
def test_conv3x3_pool2_equals_conv_then_pool(dt, b, h, w, cout):
    """The fused stem tail must give exactly what the two kernels give (the pool averages the 16-bit conv outputs)."""
    x = _rand((b * h * w, 32), 150).to(dt).to(DEV)
    k = 9 * 32
    kp = (k + 63) // 64 * 64
    w2 = torch.zeros(cout, kp, dtype=dt)
    w2[:, :k] = _rand((cout, k), 151, (2.0 / k) ** 0.5).to(dt)
    bias = _rand((cout,), 152, 0.1).to(DEV)
    full = torch.empty(b * h * w, cout, dtype=dt, device=DEV)
    ops.conv3x3_nhwc(x, w2.to(DEV), bias, full, b, h, w, 32)
    want = torch.empty(b * (h // 2) * (w // 2), cout, dtype=dt, device=DEV)
    ops.avgpool2_nhwc(full, want, b, h, w, cout)
    got = torch.full_like(want, 3.0)
    ops.conv3x3_pool2_nhwc(x, w2.to(DEV), bias, got, b, h, w, 32)
    assert torch.equal(got, want)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,r,cout", [(2, 64, 32), (1, 224, 32), (3, 36, 40), (2, 100, 48), (1, 288, 40), (5, 8, 32)])
def test_stem_conv1_vs_conv2d(dt, b, r, cout):
    """3x3 / stride 2 / pad 1 from the fp32 NCHW image, bias + ReLU, NHWC out: against conv2d on the same 16-bit values."""
    img = _rand((b, 3, r, r), 160)
    wt = _rand((cout, 3, 3, 3), 161, 0.3).to(dt)
    bias = _rand((cout,), 162, 0.2)
    ref = torch.relu(torch.nn.functional.conv2d(img.to(dt).float(), wt.float(), bias, stride=2, padding=1))
    ho = ref.shape[2]
    w2 = torch.zeros(cout, 64, dtype=dt)
    w2[:, :27] = wt.permute(0, 2, 3, 1).reshape(cout, 27)
    out = torch.full((b * ho * ho, cout), 9.0, dtype=dt, device=DEV)
    ops.stem_conv1(img.to(DEV), w2.to(DEV), bias.to(DEV), out)
    got = out.float().cpu().view(b, ho, ho, cout).permute(0, 3, 1, 2)
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert (got - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dt", DTS)
def test_gemm_relu_epilogues(dt):
    m, n, k = 260, 192, 128
    a, w = _rand((m, k), 63).to(dt), _rand((n, k), 64, 0.1).to(dt)
    bias, idn = _rand((n,), 65), _rand((m, n), 66).to(dt)
    base = a.float() @ w.float().t() + bias
    tol = dict(rtol=1e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=2e-3, atol=3e-3)
    out = torch.empty(m, n, dtype=dt, device=DEV)
    ops.gemm_nt(a.to(DEV), w.to(DEV), out, bias=bias.to(DEV), epilogue=4)
    assert torch.allclose(out.float().cpu(), torch.relu(base), **tol)
    ops.gemm_nt(a.to(DEV), w.to(DEV), out, bias=bias.to(DEV), residual=idn.to(DEV), epilogue=5)
    assert torch.allclose(out.float().cpu(), torch.relu(base + idn.float()), **tol)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(1500, 64, 256), (1024, 40, 64), (4097, 64, 576)])
def test_gemm_tall_tiles_for_narrow_outputs(dt, m, n, k):
    """N <= 64 with a fused bias + ReLU (1x1 convolutions into the 64-channel ResNet stages) runs the 256 x 64 tile
    arrangement: ragged M and N edges, integer operands so that the comparison is exact."""
    gen = torch.Generator().manual_seed(m + n)
    a = torch.randint(-2, 3, (m, k), generator=gen).float()
    w = torch.randint(-1, 2, (n, k), generator=gen).float()
    bias = torch.randint(-4, 5, (n,), generator=gen).float()
    ld = (n + 7) // 8 * 8
    out = torch.full((m, ld), 3.0, dtype=dt, device=DEV)
    ops.gemm_nt(a.to(dt).to(DEV), w.to(dt).to(DEV), out, bias=bias.to(DEV), epilogue=4, n=n)
    assert torch.equal(out[:, :n].float().cpu(), torch.relu(a @ w.t() + bias).to(dt).float())
    assert (out[:, n:] == 3.0).all()


@pytest.mark.parametrize("dt", DTS)
def test_stem_im2col_avgpool_attnpool(dt):
    b, r = 2, 20
    img = _rand((b, 3, r, r), 67)
    ho = (r - 1) // 2 + 1
    col = torch.full((b * ho * ho, 64), 5.0, dtype=dt, device=DEV)
    ops.stem_im2col(img.to(DEV), col)
    unf = torch.nn.functional.unfold(img, 3, padding=1, stride=2)          # [b, c*9, L] in (c, ky, kx) order
    ref = unf.view(b, 3, 9, ho * ho).permute(0, 3, 2, 1).reshape(b * ho * ho, 27).to(dt)   # -> (ky, kx, c)
    assert torch.equal(col[:, :27].cpu(), ref) and (col[:, 27:] == 0).all()
    # 2x2 average pool, NHWC
    h, w, c = 6, 10, 24
    x = _rand((b, h, w, c), 68).to(dt)
    out = torch.empty(b, h // 2, w // 2, c, dtype=dt, device=DEV)
    ops.avgpool2_nhwc(x.to(DEV), out, b, h, w, c)
    ref = torch.nn.functional.avg_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert (out.float().cpu() - ref).abs().max() < (2e-2 if dt == torch.bfloat16 else 2e-3)
    # attention pool: tokens and the single-query attention vs the oracle's arithmetic
    s, e, heads = 3, 128, 2
    l = s * s + 1
    xx = _rand((b, s, s, e), 69).to(dt)
    pos = _rand((l, e), 70, 0.1)
    tok = torch.empty(b * l, e, dtype=dt, device=DEV)
    ops.attnpool_tokens(xx.to(DEV), pos.to(DEV), tok, b, s, e)
    cells = xx.float().view(b, s * s, e)
    tref = torch.cat([cells.mean(1, keepdim=True), cells], 1) + pos
    assert (tok.float().cpu().view(b, l, e) - tref).abs().max() < (3e-2 if dt == torch.bfloat16 else 3e-3)
    q = _rand((b, e), 71)
    k16, v16 = _rand((b * l, e), 72).to(dt), _rand((b * l, e), 73).to(dt)
    o = torch.empty(b, e, dtype=dt, device=DEV)
    ops.attnpool_attend(q.to(DEV), k16.to(DEV), v16.to(DEV), o, b, l, heads)
    qh = q.view(b, 1, heads, 64).transpose(1, 2)
    kh = k16.float().view(b, l, heads, 64).transpose(1, 2)
    vh = v16.float().view(b, l, heads, 64).transpose(1, 2)
    oref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(b, e)
    assert (o.float().cpu() - oref).abs().max() < (2e-2 if dt == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("l,heads", [(1, 1), (50, 32), (64, 2), (65, 3), (82, 40), (145, 48), (256, 2)])
def test_attnpool_attend_long_sequences(dt, l, heads):
    """Token counts of every CLIP ResNet: 50 (RN50/RN101 at 224), 82 (RN50x4 at 288), 145 (RN50x16 at 384); each lane
    of the wave scores up to four keys."""
    b, e = 3, heads * 64
    q = _rand((b, e), 171)
    k16, v16 = _rand((b * l, e), 172).to(dt), _rand((b * l, e), 173).to(dt)
    o = torch.empty(b, e, dtype=dt, device=DEV)
    ops.attnpool_attend(q.to(DEV), k16.to(DEV), v16.to(DEV), o, b, l, heads)
    qh = q.view(b, 1, heads, 64).transpose(1, 2)
    kh = k16.float().view(b, l, heads, 64).transpose(1, 2)
    vh = v16.float().view(b, l, heads, 64).transpose(1, 2)
    oref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(b, e)
    assert (o.float().cpu() - oref).abs().max() < (2e-2 if dt == torch.bfloat16 else 2e-3)


def test_gemm_race_screen_deep_pipeline():
    """The 256^2 kernel keeps 5 LDS-DMA pieces in flight across counted waits and raw barriers; a mis-placed wait
    shows up as rare wrong tiles.  Screen: many launches at several shapes (incl. the split big+small launch),
    every result bit-identical to the first and correct against an fp32 product of the same 16-bit inputs."""
    for (m, n, k) in [(25600, 2304, 768), (2560, 3072, 768), (4096, 4096, 1024), (1536, 768, 3072), (777, 1000, 256)]:
        a = (torch.rand(m, k, device=DEV) * 2 - 1).half()
        w = ((torch.rand(n, k, device=DEV) * 2 - 1) * 0.05).half()
        ref = (a.float() @ w.float().t())
        first = None
        out = torch.empty(m, n, dtype=torch.float32, device=DEV)
        for it in range(25):
            out.fill_(float("nan"))
            ops.gemm_nt(a, w, out)
            if first is None:
                first = out.clone()
                assert (first - ref).abs().max() < 2e-3 * max(1.0, float(ref.abs().max()))
            else:
                assert torch.equal(out, first), (m, n, k, it)


@pytest.mark.parametrize("n,levels,dup", [(90, 8, False), (48, 5, False), (3000, 12, False), (21841, 12, False), (5000, 6, True)])
def test_eval_rows_fused_exact(n, levels, dup):
    """hgr_eval_rows (LDS-atomic level-segmented arg-max + top-1 + top-k) == the oracle's per-level masking / top-k, bit-exact."""
    rows, k = 4, 20
    lg = _rand((rows, n), 70 + n, 0.05)
    if dup:                                                  # heavy duplication: exercises tie order and the in-kernel fallback
        lg = torch.from_numpy(synth.randint(5, "dupv", rows * n, 0, 3).astype(np.float32).reshape(rows, n)) * 0.1
    lg[0, 3] = lg[0, 17]
    depth = synth.randint(2, "depth", n, 0, levels).astype(np.int32)
    if levels >= 12:
        depth[depth == 5] = 4                                # an empty level
    perm = np.argsort(synth.uniform(2, "perm", n), kind="stable").astype(np.int32)
    train = perm[: n - n // 3]
    test = np.sort(perm[n - n // 2:])
    ld = (n + 63) // 64 * 64
    buf = torch.zeros(rows, ld, dtype=torch.float32, device=DEV)
    buf[:, :n] = lg.to(DEV)
    test = perm[n - n // 2:]                                  # unsorted: tie order must follow the subset positions
    if n == 48:
        test = np.arange(8, 32, dtype=np.int32)[::-1].copy()  # 24 clustered test columns: several share one slice
    index = ops.EvalIndex(torch.from_numpy(depth).to(DEV), torch.from_numpy(train).to(DEV), torch.from_numpy(test).to(DEV), levels)
    lvl, top1, topk = ops.eval_rows(buf[:, :n], index, k)
    tr = train.astype(np.int64)
    for l in range(levels):
        same = [int(i) for i in np.nonzero(depth == l)[0]]
        assert np.array_equal(lvl[:, l].cpu().numpy(), tree_ref.level_argmax(lg.numpy(), tr, same, n)), l
    for r in range(rows):
        assert int(top1[r, 0]) == tr[tree_ref.topk_desc(lg[r, torch.from_numpy(tr)].numpy(), 1)[0]]
        assert np.array_equal(topk[r].cpu().numpy(), test[tree_ref.topk_desc(lg[r, torch.from_numpy(test.astype(np.int64))].numpy(), k)])


@pytest.mark.parametrize("seed", range(6))
def test_eval_counters_vs_host_recount(seed):
    """hgr_eval_counters against a literal recount of main.py:139-148 (top-k hits), :157-160 (hit_ratio) and :177-191
    (path / point overlap) on random predictions: path lengths 1..9, per-row or scalar targets, repeated accumulation."""
    rng = np.random.default_rng(seed)
    rows, k, n_levels, n_nodes = int(rng.integers(1, 300)), 20, int(rng.integers(1, 14)), 500
    L = int(rng.integers(1, min(n_levels, 9) + 1))
    parents = rng.choice(n_nodes, L, replace=False).astype(np.int32)
    levels = np.sort(rng.choice(n_levels, L, replace=False)).astype(np.int32)
    target = int(parents[-1])
    pred = np.stack([rng.choice(n_nodes, k, replace=False) for _ in range(rows)]).astype(np.int32)
    pred[rng.random(rows) < 0.4, int(rng.integers(0, k))] = target                 # plant hits at one rank
    top1 = np.where(rng.random(rows) < 0.5, rng.choice(parents, rows), rng.integers(0, n_nodes, rows)).astype(np.int32)
    lv = rng.integers(0, n_nodes, (rows, n_levels)).astype(np.int32)
    for i in range(L):                                                             # plant level matches
        m = rng.random(rows) < 0.6
        lv[m, levels[i]] = parents[i]
    acc = torch.zeros(9, dtype=torch.float64, device=DEV)
    per_row = seed % 2 == 0
    tg = torch.full((rows,), target, dtype=torch.int64, device=DEV) if per_row else None
    for _ in range(2):                                                             # accumulates
        ops.eval_counters(torch.from_numpy(pred).to(DEV), tg, target, torch.from_numpy(top1).to(DEV), torch.from_numpy(lv).to(DEV),
                          torch.from_numpy(parents).to(DEV), torch.from_numpy(levels).to(DEV), acc)
    want = np.zeros(9)
    for r in range(rows):
        hit = np.nonzero(pred[r] == target)[0]
        j = int(hit[0]) if hit.size else k
        for c, kk in enumerate((1, 2, 5, 10, 20)):
            want[c] += j < kk
        want[5] += int((top1[r] == parents).sum())
        match = lv[r, levels] == parents
        want[7] += match.sum() / L
        want[6] += (match[:-1] & match[1:]).sum() / (L - 1) if L > 1 else float(match[0])
        want[8] += 1
    assert np.allclose(acc.cpu().numpy(), 2 * want, rtol=0, atol=1e-9)


# ---- LayerNorm folded into the GEMMs around it (hgr_gemm_nt_res_stats / hgr_gemm_nt_ln) ------------------------------------
def _slot_stats(x):
    m, n = x.shape
    xs = x.double().view(m, n // 64, 64)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).float()


def _pair(x, dt):
    """Reference encoder of the residual pair (include/hgr.h), on the bits of x: t = bits(x) + half an ulp of the MFMA type; hi = t with
    the S dropped mantissa bits cleared (S = 13 for f16, 16 for bf16: exactly representable), q = the next 8 bits of t."""
    s_ = 13 if dt == torch.float16 else 16
    t = x.float().contiguous().view(torch.int32) + (1 << (s_ - 1))
    q = (t >> (s_ - 8)) & 255
    if dt == torch.float16:                              # below the f16 normal range hi is not a bit copy: no extra bits there
        q = torch.where((t & 0x7FFFFFFF) < 0x38800000, torch.full_like(q, 128), q)
    hi = (t & ~((1 << s_) - 1)).view(torch.float32).to(dt)
    return hi, q.to(torch.uint8)


def _pair_err_bound(dt):
    """|x - decode| < ulp(hi) / 256 <= 2^-10 |x| / 256 (f16; 2^-7 with bf16)."""
    return (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7) / 256


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k", [(512, 256, 128), (1000, 768, 768), (25600, 768, 768), (257, 128, 3072), (3, 1024, 192)])
def test_gemm_nt_res_stats(dt, m, n, k):
    """Producer of the folded LayerNorm on the residual stream kept as a pair: (xh, xl) += a w^T + b.  The fp32 value it
    forms (old pair + the product, same K order as hgr_gemm_nt's residual epilogue) must come back from the new pair to
    ulp(xh) / 256 (2^-18 |x| with f16 hi: 11 + 8 bits; 2^-15 |x| with bf16 hi); xh / xl must be the reference encoder's split of
    that value; the
    slot statistics = (sum, sum of squares) of every 64-column slot of the new rows (fp32 sums + a fixed-order DPP reduction:
    1e-5 relative to fp64); ragged M; bit-deterministic."""
    a, w = _rand((m, k), 11).to(dt).to(DEV), _rand((n, k), 12, 0.1).to(dt).to(DEV)
    bias, x0 = _rand((n,), 13).to(DEV), _rand((m, n), 14, 2.0).to(DEV)
    xh0, xl0 = _pair(x0, dt)
    want = ops.pair_value(xh0, xl0)                                  # what the kernel reads back as the old residual (hgr_pair_rows_f32: its decoder)
    assert float(((want - x0).abs() - _pair_err_bound(dt) * x0.abs()).max()) <= 1e-7
    ops.gemm_nt(a, w, want, bias=bias, residual=want, epilogue=EPI_BIAS_RESIDUAL)
    xh, xl = xh0.clone(), xl0.clone()
    stats = torch.full((m, n // 64, 2), -1.0, dtype=torch.float32, device=DEV)
    ops.gemm_nt_res_stats(a, w, xh, xl, bias, stats)
    # the kernel's fp32 value may differ from `want` in its last bit (the order of its three additions is its own): hi and the byte
    # are compared with the reference encoder of `want` up to that - equal almost everywhere, never more than one step apart
    rh, rl = _pair(want, dt)
    assert float((xh.float() - rh.float()).abs().max()) <= float((want.abs().max())) * (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7)
    assert float((xh == rh).float().mean()) > 0.999 and float((xl == rl).float().mean()) > 0.98
    back = ops.pair_value(xh, xl)
    assert float(((back - want).abs() - _pair_err_bound(dt) * want.abs()).max()) <= 1e-7
    ref = _slot_stats(want.cpu())
    got = stats.cpu()
    assert torch.allclose(got[..., 0], ref[..., 0], rtol=1e-5, atol=1e-4)
    assert torch.allclose(got[..., 1], ref[..., 1], rtol=1e-5, atol=1e-4)
    again = torch.empty_like(stats)
    xh2, xl2 = xh0.clone(), xl0.clone()
    ops.gemm_nt_res_stats(a, w, xh2, xl2, bias, again)
    assert torch.equal(again, stats) and torch.equal(xh2, xh) and torch.equal(xl2, xl)          # bit-deterministic
    # selected rows of the pair back in fp32 (ln_post / ln_final read them this way)
    rows = min(m, 5)
    idx = torch.arange(rows, dtype=torch.int32, device=DEV) % 2
    sel = torch.empty(rows, n, device=DEV)
    mul = max(1, m // rows)
    ops.pair_rows_f32(xh, xl, sel, row_mul=mul, row_idx=idx if mul > 1 else None)
    src = torch.arange(rows, device=DEV) * mul + (idx.long() if mul > 1 else 0)
    assert torch.equal(sel, back[src])


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m", [256, 200])
def test_gemm_nt_res_stats_pair_corner_cases(dt, m):
    """The producer's own split / decode of the residual pair (round 6: pair_split2 / pair_put_q / pair_dec4 - a float compare with |.| and
    SDWA byte shifts where the first form masked, compared and shifted per element) on the values that separate the forms: zeros of
    both signs, fp32 denormals, the f16 subnormal range and its upper edge, mantissa carries into the next binade, the largest f16,
    negative twins.  With A = 0 the product vanishes: the new pair must be the reference encoder's split of (old value + bias),
    interior tiles (m = 256) and the guarded edge-tile path (m = 200) alike."""
    n, k = 128, 128
    sp = torch.tensor([0.0, -0.0, 1e-40, -1e-40, 1e-8, -1e-8, 5.9e-8, 2.98e-8, 6e-5, -6e-5, 6.2e-5, 6.103515625e-05, 6.1035e-05, -6.1035e-05, 3.0517578125e-05,
                       1.9999, -1.9999, 2047.9999, -2047.9999, 65000.0, -65000.0, 0.999755859375, 0.99987793, 1.0, -1.0, 3.14159, 1e-3, -1e-3, 0.33333334, 1024.5,
                       7.62939453125e-06, -7.62939453125e-06], dtype=torch.float32)
    x0 = sp.repeat((m * n + sp.numel() - 1) // sp.numel())[: m * n].view(m, n).contiguous().to(DEV)
    x0 = x0 * (1.0 + 2.0 ** -12 * (torch.arange(m, device=DEV) % 7).float()[:, None])        # walk the low mantissa bits row by row
    xh0, xl0 = _pair(x0, dt)
    old = ops.pair_value(xh0, xl0)
    a = torch.zeros(m, k, dtype=dt, device=DEV)
    w = _rand((n, k), 5, 0.1).to(dt).to(DEV)
    for bias_val in (0.0, 2.0 ** -20, -3.0e-5):
        bias = torch.full((n,), bias_val, dtype=torch.float32, device=DEV)
        xh, xl = xh0.clone(), xl0.clone()
        stats = torch.empty(m, n // 64, 2, dtype=torch.float32, device=DEV)
        ops.gemm_nt_res_stats(a, w, xh, xl, bias, stats)
        want = (0.0 + bias[None, :]) + old                          # the kernel's association: (product + bias) + old value
        rh, rl = _pair(want, dt)
        assert torch.equal(xh.view(torch.int16), rh.view(torch.int16)), (bias_val, int((xh != rh).sum()))
        assert torch.equal(xl, rl), (bias_val, int((xl != rl).sum()))
        back = ops.pair_value(xh, xl)
        assert torch.isfinite(back).all()
        ref = _slot_stats(want.cpu())
        assert torch.allclose(stats.cpu()[..., 0], ref[..., 0], rtol=1e-5, atol=1e-2) and torch.allclose(stats.cpu()[..., 1], ref[..., 1], rtol=1e-5, atol=1e2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("m,n,k,panels", [(25600, 768, 768, -1), (25600, 768, 3072, -1), (25600, 2304, 768, 86), (25523, 768, 128, -1),
                                           (25600, 768, 192, 0), (25600, 768, 256, 1), (25600, 768, 256, 99), (26000, 1536, 256, -1)])
def test_gemm_tail_plan_is_bit_identical(dt, m, n, k, panels):
    """The tail plan of the 256 x 128 tile kernel (half tiles on the last row panels, hgr_gemm_set_tail; DESIGN.md 4.1, "tail plan") against the
    all-full-tile launch of the same call: every output element sums K in the same order on either tile, so every entry point that
    rides the kernel - plain 16-bit / fp32 epilogues, the LayerNorm producer (pair + slot statistics) and consumer, the dual-output
    forward - must give the same BITS on random data.  ViT-B/32 tower shapes at batch 512 (600 / 1800 tiles on 512 slots), a ragged
    M (last half panel partly empty, last full panel absent), 2- and 3-K-tile reductions (the prologue / drain forms of the
    half-tile pipeline), forced panel counts 0 (all half tiles), 1 and tiles_m - 1."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)               # device-side random operands: the hash generator of synth
    rnd = lambda shape, scale=1.0: scale * torch.randn(shape, generator=g, device=DEV)      # would spend minutes on 10^8 values
    a, w = rnd((m, k)).to(dt), rnd((n, k), 0.1).to(dt)
    bias, x0 = rnd((n,)), rnd((m, n), 2.0)
    xh0, xl0 = _pair(x0, dt)
    gamma_s, c = rnd((n,)), rnd((n,))
    xk = rnd((m, k), 1.5) if k % 128 == 0 else None

    def run():
        out = {}
        o16 = torch.empty(m, n, dtype=dt, device=DEV)
        ops.gemm_nt(a, w, o16, bias=bias, epilogue=EPI_BIAS_QUICKGELU)
        out["gelu16"] = o16
        o32 = x0.clone()
        ops.gemm_nt(a, w, o32, bias=bias, residual=o32, epilogue=EPI_BIAS_RESIDUAL)
        out["res32"] = o32
        xh, xl = xh0.clone(), xl0.clone()
        stats = torch.zeros((m, n // 64, 2), dtype=torch.float32, device=DEV)
        ops.gemm_nt_res_stats(a, w, xh, xl, bias, stats)
        out["xh"], out["xl"], out["stats"] = xh, xl, stats
        if k % 128 == 0:
            # consumer on ITS operand shape: rows of width k with statistics of their own
            kh, kl = _pair(xk, dt)
            st = torch.empty((m, k // 64, 2), dtype=torch.float32, device=DEV)
            ops.row_stats16(xk, kh, kl, st)
            y = torch.empty(m, n, dtype=dt, device=DEV)
            ops.gemm_nt_ln(kh, w, y, gamma_s, c, st, 1e-5, quickgelu=True)
            out["ln"] = y
        if ops.gelu_dual_ok(m, n, k, a.stride(0), w.stride(0)):
            pre, post = torch.empty(m, n, dtype=dt, device=DEV), torch.empty(m, n, dtype=dt, device=DEV)
            ops.gemm_nt_bias_gelu_dual(a, w, pre, post, bias)
            out["pre"], out["post"] = pre, post
        return out

    prev = ops.gemm_set_tail(False)
    try:
        full = run()
        ops.gemm_set_tail(True, min(panels, (m + 255) // 256 - 1) if panels >= 0 else -1)
        tail = run()
    finally:
        ops.gemm_set_tail(bool(prev))
    for key in full:
        assert torch.equal(full[key], tail[key]), key


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tail", [False, True])
@pytest.mark.parametrize("m,n,k", [(25600, 768, 768), (25600, 768, 3072), (25523, 768, 256), (40000, 768, 128), (16384, 1024, 256), (12800, 768, 192)])
def test_gemm_res_stats_persistent_is_bit_identical(dt, tail, m, n, k):
    """The persistent form of the residual producers (hgr_gemm_set_persist: workgroup b walks tiles b, b + 512, ... on launches of one to
    two rounds of the chip's 512 slots) against one workgroup per tile: same tiles, same arithmetic, so pair and slot statistics must be
    the same BITS.  600 / 940 / 512 / 300 tiles: inside the window, at its upper edge (1024 would be the last one in), exactly one
    round and less than a round (both not persistent: the switch must then change nothing either); with and without the tail plan
    (half tiles are the last virtual blocks), a ragged last row panel, 2- / 3-K-tile reductions."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    rnd = lambda shape, scale=1.0: scale * torch.randn(shape, generator=g, device=DEV)
    a, w = rnd((m, k)).to(dt), rnd((n, k), 0.1).to(dt)
    bias, x0 = rnd((n,)), rnd((m, n), 2.0)
    xh0, xl0 = _pair(x0, dt)

    def run():
        xh, xl = xh0.clone(), xl0.clone()
        stats = torch.zeros((m, n // 64, 2), dtype=torch.float32, device=DEV)
        ops.gemm_nt_res_stats(a, w, xh, xl, bias, stats)
        return xh, xl, stats

    prev_t = ops.gemm_set_tail(tail)
    prev_p = ops.gemm_set_persist(False)
    try:
        one = run()
        ops.gemm_set_persist(True)
        walk = run()
    finally:
        ops.gemm_set_persist(bool(prev_p))
        ops.gemm_set_tail(bool(prev_t))
    for x, y, name in zip(one, walk, ("xh", "xl", "stats")):
        assert torch.equal(x, y), name


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("m,n,k", [(512, 384, 128), (1000, 2304, 768), (25600, 3072, 768), (300, 128, 1024)])
def test_gemm_nt_ln(dt, gelu, m, n, k):
    """Consumer of the folded LayerNorm against fp32 LayerNorm -> Linear (clip/model.py:153-159 feeding :171 / :177-180),
    rows with a non-zero mean and unequal scales; tolerance = the 16-bit rounding of the operands, as for the unfused path,
    and the two paths are compared with each other too."""
    x = (_rand((m, k), 21, 1.5) + 0.3 * _rand((m, 1), 22) + 0.2)
    x = x * (0.5 + torch.rand(m, 1, generator=torch.Generator().manual_seed(3)))
    w, b = _rand((n, k), 23, 0.05), _rand((n,), 24)
    gamma, beta = 1.0 + 0.2 * _rand((k,), 25), 0.1 * _rand((k,), 26)
    ref = torch.nn.functional.layer_norm(x, (k,), gamma, beta, 1e-5) @ w.t() + b
    if gelu:
        ref = clip_ref.quick_gelu(ref)
    xd = x.to(DEV)
    x16 = torch.empty(m, k, dtype=dt, device=DEV)
    xlo = torch.empty(m, k, dtype=ops.PAIR_LO, device=DEV)
    stats = torch.empty(m, k // 64, 2, dtype=torch.float32, device=DEV)
    ops.row_stats16(xd, x16, xlo, stats)
    rh, rl = _pair(xd, dt)
    assert torch.equal(x16, rh) and torch.equal(xlo, rl)
    # the pair's corner cases: zeros of both signs, fp32 denormals, the f16 subnormal range (no extra bits there: a stale byte on
    # hi = 0 would decode to a NaN pattern - it once tripped the range guard of every tower), mantissa carries, the largest f16
    sp = torch.tensor([0.0, -0.0, 1e-40, -1e-40, 1e-8, -1e-8, 5.9e-8, 6e-5, -6e-5, 6.2e-5, 1.9999, -1.9999, 2047.9999, 65000.0, -65000.0, 3.0517578125e-05],
                      dtype=torch.float32).repeat(k // 16 if k >= 16 else 1)[:k]
    xs = xd.clone()
    xs[0, : sp.numel()] = sp.to(DEV)
    ops.row_stats16(xs, x16, xlo, stats)
    back = ops.pair_value(x16, xlo)
    assert torch.isfinite(back).all()
    assert float(((back - xs).abs() - _pair_err_bound(dt) * xs.abs()).max()) <= 2.0 ** -24
    rh, rl = _pair(xs, dt)
    assert torch.equal(x16, rh) and torch.equal(xlo, rl)
    ops.row_stats16(xd, x16, xlo, stats)                               # back to the test's operands
    assert float((x16 != xd.to(dt)).float().mean()) < 1e-3            # hi = round-to-nearest, ties away from zero: RNE except on exact ties
    st = _slot_stats(x)
    assert torch.allclose(stats.cpu(), st, rtol=1e-5, atol=1e-4)
    wf = (w * gamma[None, :]).to(dt)
    s = wf.float().sum(1).to(DEV)
    c = (w @ beta + b).to(DEV)
    out = torch.full((m, n), 7.0, dtype=dt, device=DEV)
    ops.gemm_nt_ln(x16, wf.to(DEV), out, s, c, stats, 1e-5, quickgelu=gelu)
    tol = dict(rtol=2e-2, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=3e-3, atol=4e-3)
    assert torch.allclose(out.float().cpu(), ref, **tol), float((out.float().cpu() - ref).abs().max())
    # the unfused path on the same data: LayerNorm kernel -> GEMM with bias (+ QuickGELU)
    h16 = torch.empty(m, k, dtype=dt, device=DEV)
    ops.layernorm(xd, gamma.to(DEV), beta.to(DEV), h16)
    out2 = torch.empty(m, n, dtype=dt, device=DEV)
    ops.gemm_nt(h16, w.to(dt).to(DEV), out2, bias=b.to(DEV), epilogue=EPI_BIAS_QUICKGELU if gelu else EPI_BIAS)
    e_fused = float((out.float().cpu() - ref).abs().max())
    e_plain = float((out2.float().cpu() - ref).abs().max())
    assert e_fused < 2.0 * e_plain + 1e-3, (e_fused, e_plain)          # folding does not cost accuracy


@pytest.mark.parametrize("dt", DTS)
def test_vit_embed_ln_stats_equals_unfused(dt):
    b, g, w = 5, 49, 768
    pe = _rand((b * g, w), 31).to(DEV)
    cls, pos = _rand((w,), 32).to(DEV), _rand((g + 1, w), 33).to(DEV)
    gamma, beta = (1.0 + 0.1 * _rand((w,), 34)).to(DEV), (0.1 * _rand((w,), 35)).to(DEV)
    x1 = torch.empty(b * (g + 1), w, device=DEV)
    ops.vit_embed_ln(pe, cls, pos, gamma, beta, x1, b, g)
    xh = torch.empty(b * (g + 1), w, dtype=dt, device=DEV)
    xl = torch.empty(b * (g + 1), w, dtype=ops.PAIR_LO, device=DEV)
    stats = torch.empty(b * (g + 1), w // 64, 2, device=DEV)
    ops.vit_embed_ln_stats(pe, cls, pos, gamma, beta, xh, xl, stats, b, g)
    rh, rl = _pair(x1, dt)
    assert torch.equal(xh, rh) and torch.equal(xl, rl)
    assert torch.allclose(stats.cpu(), _slot_stats(x1.cpu()), rtol=1e-5, atol=1e-4)


# ---- logits GEMM with the evaluation in its epilogue (hgr_logits_eval) --------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", ["n21841", "ragged_rows", "empty_level", "ties", "all_equal", "clustered_test", "d1024"])
def test_logits_eval_bit_exact_vs_gemm_plus_eval_rows(dt, case):
    """hgr_logits_eval (no logits in memory: tile stage -> per-slice keys / maxima, row stage -> level arg-max, top-1, recomputed
    candidates -> top-20) must give EXACTLY the ids of hgr_gemm_nt (fp32 logits) + hgr_eval_rows on the same operands
    (main.py:136-176 on model/clip_tree.py:331): the full-size class matrix, ragged row counts, an empty level, exact ties
    (duplicated class rows), all-equal logits (the k-round fallback), a tiny clustered test set, D = 1024."""
    rows, n, d, levels, k = 64, 3000, 256, 9, 20
    if case == "n21841":
        rows, n, d, levels = 512, 21841, 512, 12
    if case == "ragged_rows":
        rows, n = 37, 1111
    if case == "d1024":
        rows, n, d = 300, 5000, 1024
    f = _rand((rows, d), 81)
    f = f / f.norm(dim=1, keepdim=True)
    z = _rand((n, d), 82)
    z = z / z.norm(dim=1, keepdim=True)
    if case == "ties":
        z[5::7] = z[3]                                        # many identical class rows: exact ties across levels and subsets
        z[100:140] = z[99]
    if case == "all_equal":
        z[:] = z[0]
    depth = synth.randint(3, "depth", n, 0, levels).astype(np.int32)
    depth[:3] = 0
    if case == "empty_level":
        depth[depth == 4] = 3
    perm = np.argsort(synth.uniform(4, "perm", n), kind="stable").astype(np.int32)
    train = perm[: n - n // 3].copy()
    test = perm[n - n // 2:].copy()                           # unsorted: tie order follows the subset positions
    if case == "clustered_test":
        test = np.arange(40, 64, dtype=np.int32)[::-1].copy()
    f16, z16 = f.to(dt).to(DEV), z.to(dt).to(DEV)
    index = ops.EvalIndex(torch.from_numpy(depth).to(DEV), torch.from_numpy(train).to(DEV), torch.from_numpy(test).to(DEV), levels)
    ld = (n + 63) // 64 * 64
    lg = torch.empty(rows, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f16, z16, lg, n=n)
    want = ops.eval_rows(lg[:, :n], index, k)
    plan = ops.LogitsEvalPlan(index).bind(z16)
    assert plan.n_perm % 96 == 0 and int(plan.valid.sum()) == n
    got = ops.logits_eval(f16, plan, k)
    for w, g_, name in zip(want, got, ("level arg-max", "top-1", "top-k")):
        assert torch.equal(w, g_), (case, name, int((w != g_).sum()))
    again = ops.logits_eval(f16, plan, k)
    assert all(torch.equal(a, b) for a, b in zip(got, again))           # run-to-run identical


def _logits_eval_oracle(lg: np.ndarray, depth: np.ndarray, train: np.ndarray, test: np.ndarray, n_levels: int, k: int):
    """main.py:136-176 on a logits matrix with the oracle's own helpers (oracle/tree_ref.py): top-k over the test columns,
    top-1 over the train columns, per level the -1-filled arg-max over the train columns."""
    n = lg.shape[1]
    pred = test[np.stack([tree_ref.topk_desc(r, k) for r in lg[:, test]])]
    p1 = train[np.array([tree_ref.topk_desc(r, 1)[0] for r in lg[:, train]])]
    lv = np.stack([tree_ref.level_argmax(lg, train, np.nonzero(depth == l)[0].tolist(), n) for l in range(n_levels)], axis=1)
    return lv, p1, pred


def test_logits_eval_vs_reference_fixture(golden_dir, tmp_path):
    """hgr_logits_eval DIRECTLY against the reference's arrays (tests/golden/tree_smallvit_n300: the class matrix `zsl_weights` the
    reference's update_classifier produced, `pred_top20` / `dict_path_i` its main.test computed - main.py:136-176): the fused kernel is
    fed the fixture's class matrix and the oracle's image features of the fixture's batches, and every id must equal the reference's
    wherever the reference's own logits decide it by more than twice the measured 16-bit logit error."""
    import json
    from hgr_net_amd.hierarchy import build_hierarchy
    meta = json.load(open(golden_dir / "tree_smallvit_n300.json"))
    z = np.load(golden_dir / "tree_smallvit_n300.npz")
    cfg = meta["config"]
    d = meta["dag"]
    h = build_hierarchy(synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"]))
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    # train / test columns exactly as tree_model builds them (model/clip_tree.py:52-66 of the reference): positions in `nodes`
    pos = {w: i for i, w in enumerate(h.nodes)}
    train = np.array([pos[w] for w in splits["all"] if w in pos], dtype=np.int32)
    test = np.array([pos[w] for w in splits["rest"] if w in pos], dtype=np.int32)
    depth = np.asarray(h.depth, dtype=np.int32)
    n_levels = int(depth.max()) + 1
    sd = synth.clip_state_dict(cfg, 0)
    zsl = torch.from_numpy(z["zsl_weights"])
    index = ops.EvalIndex(torch.from_numpy(depth).to(DEV), torch.from_numpy(train).to(DEV), torch.from_numpy(test).to(DEV), n_levels)
    plan = ops.LogitsEvalPlan(index).bind(zsl.to(torch.float16).to(DEV))
    checked = {"top": 0, "level": 0}
    for i in range(meta["batches"]):
        img = synth.images(meta["bsz"], cfg["image_resolution"], meta["image_seed0"] + i)
        with torch.no_grad():
            f = clip_ref.encode_image(sd, img)
            f = f / f.norm(dim=-1, keepdim=True)
        ref_lg = z["logits"][i]
        assert np.abs((f @ zsl.t()).numpy() - ref_lg).max() < 2e-5            # the oracle's features reproduce the reference's logits
        f16 = f.to(torch.float16).to(DEV)
        lv, p1, pred = ops.logits_eval(f16, plan, 20)
        err = float(np.abs((f16.float().cpu() @ zsl.to(torch.float16).float().t()).numpy() - ref_lg).max())
        assert err < 1e-3
        pred, lv = pred.cpu().numpy(), lv.cpu().numpy()
        # top-20 over the test columns: position j is decided when the reference separates it from both neighbours by > 2 err
        sub = ref_lg[:, test]
        for r in range(meta["bsz"]):
            order = tree_ref.topk_desc(sub[r], 21)
            assert np.array_equal(test[order[:20]], z["pred_top20"][i][r])   # the oracle's rule on the reference's logits = the reference's ids
            v = sub[r][order]
            for j in range(20):
                if v[j] - v[j + 1] > 2 * err and (j == 0 or v[j - 1] - v[j] > 2 * err):
                    assert pred[r, j] == z["pred_top20"][i][r, j], (i, r, j)
                    checked["top"] += 1
        # per-level arg-max along the target's ancestor path (dict_path of main.py:162-176)
        tgt = meta["targets"][i]
        parents = list(h.c2p[tgt]) + [tgt]
        want = z[f"dict_path_{i}"].astype(np.int64)
        for j, p in enumerate(parents):
            lvl = len(h.c2p[p])
            cols = train[depth[train] == lvl]
            for r in range(meta["bsz"]):
                top2 = np.sort(ref_lg[r, cols])[::-1][:2] if cols.size >= 2 else None
                if top2 is None or top2[0] - top2[1] > 2 * err:
                    assert lv[r, lvl] == want[r, j], (i, r, j, lvl)
                    checked["level"] += 1
    assert checked["top"] >= 300 and checked["level"] >= 60, checked


@pytest.mark.parametrize("dt", DTS)
def test_logits_eval_planted_ties_at_group_boundaries(dt):
    """Adversarial ordering case for hgr_logits_eval, exact: small-integer operands make every logit an exactly representable integer
    (so the GEMM is exact in any summation order and the ids are decided by the tie rule alone), and for each probed row the 22 best
    test columns are PLANTED so that ranks 19 / 20 / 21 are one unit apart or exactly tied and sit on both sides of a 16-column
    group boundary, a 32-column slice boundary and a 96-column slab boundary of the level-sorted class matrix.  Oracle: the
    reference's rule (main.py:136-176) through oracle/tree_ref on the integer logits."""
    n, dd, levels, k = 900, 128, 4, 20
    rows = 24
    rng = np.random.RandomState(7)
    f = rng.randint(-2, 3, size=(rows, dd)).astype(np.float32)
    f[f == 0] = 1.0                                                    # every entry +-1 / +-2: sum |f| is the row's attainable maximum
    zc = rng.randint(-1, 2, size=(n, dd)).astype(np.float32)
    depth = np.zeros(n, dtype=np.int32)
    depth[520:] = 1 + (np.arange(n - 520) % (levels - 1))             # level 0 = columns 0..519 in id order: permuted position == id there
    # rows 0..7: the boundary between permuted positions b-1 | b, in three tie flavours
    # (16, 304: 16-column group boundaries inside a 32-column slice; 64, 128, 256: slice boundaries inside a 96-column slab; 96, 192, 384:
    # slab boundaries - another CU's tile; the 22-column windows around them are disjoint)
    cases = [(16, "tie_20_21"), (96, "step"), (64, "tie_20_21"), (192, "tie_19_20"), (128, "tie_20_21"), (256, "step"), (304, "tie_19_20"),
             (384, "tie_19_20")]
    for r, (b, flavour) in enumerate(cases):
        cols = list(range(b - 11, b + 11))                             # 22 planted columns straddling the boundary, 11 on each side
        # values: ranks 1..18 strictly decreasing on alternating sides, then the flavour at 19 / 20 / 21, rank 22 lower still
        top = 2.0 * np.abs(f[r]).sum()
        vals = [top - 2 * j for j in range(18)]
        x = top - 2 * 18
        tail = {"tie_20_21": [x, x - 2, x - 2, x - 4], "tie_19_20": [x, x, x - 2, x - 4], "step": [x, x - 2, x - 4, x - 6]}[flavour]
        order = [cols[11 + (j // 2) * (1 if j % 2 == 0 else -1) - (1 if j % 2 else 0)] for j in range(18)]
        rest = [c for c in cols if c not in order]
        # ranks 19..22 alternate sides of the boundary: b-1, b, b-2, b+1 ... taken from what is left, nearest first
        rest.sort(key=lambda c: (abs(c - b + 0.5), c))
        order += rest
        for c, v in zip(order, vals + tail):
            row = 2.0 * np.sign(f[r])                                  # attains `top`; lower it in steps of 2 by zeroing |f| = 1 entries / halving
            need = int(round((top - v) / 2))
            ones = np.nonzero(np.abs(f[r]) == 1)[0]
            twos = np.nonzero(np.abs(f[r]) == 2)[0]
            # zero `a` entries with |f| = 1 (each -2) and halve `h` entries with |f| = 2 (each -2)
            a = min(need, ones.size)
            row[ones[:a]] = 0.0
            row[twos[: need - a]] = np.sign(f[r][twos[: need - a]])
            assert f[r] @ row == v
            zc[c] = row
    lg = (f.astype(np.int64) @ zc.astype(np.int64).T).astype(np.float32)
    perm = np.argsort(synth.uniform(4, "perm", n), kind="stable").astype(np.int32)
    train = perm[: n - n // 4].copy()
    # test set: every level-0 column, in an order that puts the far side of each boundary FIRST (ties must follow the subset
    # position, not the column id), plus a few deeper ones
    test = np.concatenate([np.arange(519, -1, -1, dtype=np.int32), np.arange(520, 600, dtype=np.int32)])
    f16, z16 = torch.from_numpy(f).to(dt).to(DEV), torch.from_numpy(zc).to(dt).to(DEV)
    index = ops.EvalIndex(torch.from_numpy(depth).to(DEV), torch.from_numpy(train).to(DEV), torch.from_numpy(test).to(DEV), levels)
    plan = ops.LogitsEvalPlan(index).bind(z16)
    assert np.array_equal(plan.perm[:520].cpu().numpy(), np.arange(520))           # the boundaries above are where the test says they are
    lv, p1, pred = ops.logits_eval(f16, plan, k)
    olv, op1, opred = _logits_eval_oracle(lg, depth, train, test, levels, k)
    assert np.array_equal(pred.cpu().numpy(), opred)
    assert np.array_equal(p1.cpu().numpy().ravel(), op1)
    assert np.array_equal(lv.cpu().numpy(), olv)
    for r, (b, flavour) in enumerate(cases):                                        # the planted columns really are the row's top 20
        assert set(opred[r].tolist()) <= set(range(b - 11, b + 11))
    # and the unfused route gives the same ids on the same operands
    ld = (n + 63) // 64 * 64
    out = torch.empty(rows, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f16, z16, out, n=n)
    assert np.array_equal(out[:, :n].cpu().numpy(), lg)
    for w, g_ in zip(ops.eval_rows(out[:, :n], index, k), (lv, p1, pred)):
        assert torch.equal(w, g_)


# ---- in_proj GEMM + attention in one launch (hgr_gemm_nt_ln_mha) -------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,l,heads,causal", [(512, 50, 12, False),      # ViT-B/32 image tower at batch 512: 103 row tiles x 12 heads, last tile ragged
                                              (7, 50, 4, False),         # fewer sequences than one tile holds + 1: a full and a 2-sequence tile
                                              (3, 64, 2, False),         # L = 64: 4 sequences per tile, every key slot live
                                              (33, 17, 2, True),         # prompts: causal, 15 sequences per tile, query tiles straddle two sequences
                                              (5, 1, 2, False),          # one token per sequence: the softmax of a single score
                                              (40, 16, 8, True)])        # 16 sequences per tile, no straddling
def test_gemm_ln_mha_equals_gemm_then_mha(dt, b, l, heads, causal):
    """hgr_gemm_nt_ln_mha (in_proj of the folded ln_1 + scaled dot-product attention in one launch, clip/model.py:171,183-186) must give
    the BITS of hgr_gemm_nt_ln into a qkv buffer followed by hgr_mha, and agree with the fp32 oracle (oracle/clip_ref.mha on the
    LayerNorm-ed rows) to the 16-bit operand tolerance."""
    w = heads * 64
    m = b * l
    x = (_rand((m, w), 41, 1.2) + 0.2 * _rand((m, 1), 42))
    x = x * (0.6 + torch.rand(m, 1, generator=torch.Generator().manual_seed(9)))
    w_in, b_in = _rand((3 * w, w), 43, w ** -0.5), 0.1 * _rand((3 * w,), 44)
    gamma, beta = 1.0 + 0.2 * _rand((w,), 45), 0.1 * _rand((w,), 46)
    xd = x.to(DEV)
    x16 = torch.empty(m, w, dtype=dt, device=DEV)
    xlo = torch.empty(m, w, dtype=ops.PAIR_LO, device=DEV)
    stats = torch.empty(m, w // 64, 2, dtype=torch.float32, device=DEV)
    ops.row_stats16(xd, x16, xlo, stats)
    wf = (w_in * gamma[None, :]).to(dt)
    s = wf.float().sum(1).to(DEV)
    c = (w_in @ beta + b_in).to(DEV)
    wf = wf.to(DEV)
    qkv = torch.empty(m, 3 * w, dtype=dt, device=DEV)
    want = torch.empty(m, w, dtype=dt, device=DEV)
    ops.gemm_nt_ln(x16, wf, qkv, s, c, stats, 1e-5)
    ops.mha(qkv, want, b, l, heads, causal)
    got = torch.full((m, w), 7.0, dtype=dt, device=DEV)
    ops.gemm_nt_ln_mha(x16, wf, got, s, c, stats, b, l, heads, causal, 1e-5)
    assert torch.equal(got, want), (int((got != want).sum()), float((got.float() - want.float()).abs().max()))
    again = torch.empty_like(got)
    ops.gemm_nt_ln_mha(x16, wf, again, s, c, stats, b, l, heads, causal, 1e-5)
    assert torch.equal(again, got)                                            # run-to-run identical
    if m <= 4096:                                                             # oracle: fp32 attention of the LayerNorm-ed rows
        sd = {"p.attn.in_proj_weight": w_in, "p.attn.in_proj_bias": b_in,
              "p.attn.out_proj.weight": torch.eye(w), "p.attn.out_proj.bias": torch.zeros(w)}
        h = torch.nn.functional.layer_norm(x, (w,), gamma, beta, 1e-5).view(b, l, w)
        ref = clip_ref.mha(h, sd, "p.attn", heads, causal, clip_ref.identity).reshape(m, w)
        tol = dict(rtol=2e-2, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=3e-3, atol=4e-3)
        assert torch.allclose(got.float().cpu(), ref, **tol), float((got.float().cpu() - ref).abs().max())


@pytest.mark.parametrize("dt", DTS)
def test_gemm_p8_equals_duo(dt):
    """Round 5: the persistent 256 x 256 form of the LayerNorm-folded consumer GEMM (csrc/hgr_gemm_p8.hip, hgr_gemm_set_p8: one 512-thread
    workgroup per CU walks its XCD's tiles, the next tile's first K-tile requested ahead of the store epilogue) gives the bits of
    gemm_nt_duo through hgr_gemm_nt_ln (clip/model.py:177-187: ln_2 -> c_fc -> QuickGELU), on launches of one to several tiles per
    workgroup, every statistics-slot count the epilogue unrolls, and agrees with an fp32 LayerNorm + linear of the same rows."""
    from hgr_net_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(23)
    # (m, n, k), QuickGELU: 768 / 1024 / 512 / 320 / 320 tiles on 256 CUs; K = 512 ... 1024 (ln_slots 8, 12, 16) and 640 (10), 384 (the generic loop)
    for (m, n, k), act in (((16384, 3072, 768), True), ((16384, 4096, 512), False), ((16384, 2048, 1024), True), ((20480, 1024, 640), False),
                           ((8192, 2560, 384), True)):
        a = (torch.rand(m, k, device=DEV, generator=g) * 2 - 1 + 0.3 * torch.rand(m, 1, device=DEV, generator=g)).to(dt)
        w = ((torch.rand(n, k, device=DEV, generator=g) * 2 - 1) * 0.05).to(dt)
        s_ = w.float().sum(1)                                          # gamma = 1: ln_s = row sums of the folded weight
        c_ = torch.rand(n, device=DEV, generator=g) - 0.5
        x = a.float().view(m, k // 64, 64)
        stats = torch.stack([x.sum(-1), (x * x).sum(-1)], dim=-1).contiguous()
        outs = []
        for mode in (0, 1):
            prev = lib.hgr_gemm_set_p8(mode)
            try:
                out = torch.full((m, n), float("nan"), dtype=dt, device=DEV)
                ops.gemm_nt_ln(a, w, out, s_, c_, stats, quickgelu=act)
                outs.append(out)
            finally:
                lib.hgr_gemm_set_p8(prev)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (m, n, k, act)
        rows = slice(0, 1024)
        ref = torch.nn.functional.layer_norm(a[rows].float(), (k,), eps=1e-5) @ w.float().t() + c_
        if act:
            ref = ref * torch.sigmoid(1.702 * ref)
        tol = dict(rtol=3e-2, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=5e-3, atol=5e-3)
        assert torch.allclose(outs[1][rows].float(), ref, **tol), float((outs[1][rows].float() - ref).abs().max())
    assert lib.hgr_gemm_set_p8(2) in (0, 1, 2)                         # back to "by shape", whatever a failed case left behind


@pytest.mark.parametrize("dt", DTS)
def test_gemm_ws_equals_duo(dt):
    """Round 5: the role-split kernel (csrc/hgr_gemm_ws.hip: four matrix waves + four helper waves per CU, the epilogue of tile i
    under the MFMAs of tile i + 1; hgr_gemm_set_ws, off by default) gives the bits of gemm_nt_duo through every entry point it
    covers - hgr_gemm_nt (NONE / BIAS / BIAS_RELU, 16-bit out), hgr_gemm_nt_ln (+- QuickGELU), hgr_gemm_nt_res_stats_guard (pair,
    slot statistics, guard flag) - on launches of several tiles per workgroup (persistent walk, dump hand-over, counted waits)."""
    from hgr_net_amd import _lib
    from hgr_net_amd._lib import EPI_BIAS_RELU
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(11)

    def both(fn):
        outs = []
        for on in (0, 1):
            prev = lib.hgr_gemm_set_ws(on)
            try:
                outs.append(fn())
            finally:
                lib.hgr_gemm_set_ws(prev)
        torch.cuda.synchronize()
        return outs

    def mk(m, n, k):
        a = (torch.rand(m, k, device=DEV, generator=g) * 2 - 1).to(dt)
        w = ((torch.rand(n, k, device=DEV, generator=g) * 2 - 1) * 0.05).to(dt)
        return a, w

    # hgr_gemm_nt: 768 tiles on 256 CUs = 3 per workgroup; K = 768 (12 K-tiles: no plain-loop iteration) and 1024 (two)
    for (m, n, k), epi in (((8192, 3072, 768), EPI_NONE), ((16384, 1536, 1024), EPI_BIAS), ((8192, 3072, 768), EPI_BIAS_RELU)):
        a, w = mk(m, n, k)
        bias = None if epi == EPI_NONE else torch.rand(n, device=DEV, generator=g) - 0.5

        def run():
            out = torch.full((m, n), float("nan"), dtype=dt, device=DEV)
            ops.gemm_nt(a, w, out, bias=bias, epilogue=epi)
            return out
        o0, o1 = both(run)
        assert torch.equal(o0.view(torch.int16), o1.view(torch.int16)), (m, n, k, epi)
        ref = a.float() @ w.float().t() + (0 if bias is None else bias)
        if epi == EPI_BIAS_RELU:
            ref = ref.clamp_min(0)
        assert torch.allclose(o1.float(), ref, rtol=2e-2, atol=2e-2)
    # hgr_gemm_nt_ln: the folded-LayerNorm consumer, with and without QuickGELU
    for (m, n, k), act in (((8192, 3072, 768), True), ((16384, 1536, 1024), False)):
        a, w = mk(m, n, k)
        s_, c_ = torch.rand(n, device=DEV, generator=g) - 0.5, torch.rand(n, device=DEV, generator=g) - 0.5
        x = a.float().view(m, k // 64, 64)
        stats = torch.stack([x.sum(-1), (x * x).sum(-1)], dim=-1).contiguous()

        def run():
            out = torch.full((m, n), float("nan"), dtype=dt, device=DEV)
            ops.gemm_nt_ln(a, w, out, s_, c_, stats, quickgelu=act)
            return out
        o0, o1 = both(run)
        assert torch.equal(o0.view(torch.int16), o1.view(torch.int16)), (m, n, k, act)
    # hgr_gemm_nt_res_stats_guard: the residual producer (in place on the pair)
    for m, n, k in ((25600, 768, 768), (16384, 1024, 3072)):
        a, w = mk(m, n, k)
        bias = torch.rand(n, device=DEV, generator=g) - 0.5
        xh0 = (torch.rand(m, n, device=DEV, generator=g) * 4 - 2).to(dt)
        xl0 = torch.randint(0, 256, (m, n), device=DEV, generator=g, dtype=torch.int32).to(torch.uint8)

        def run():
            xh, xl = xh0.clone(), xl0.clone()
            st = torch.full((m, n // 64, 2), float("nan"), device=DEV)
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.gemm_nt_res_stats(a, w, xh, xl, bias, st, flag=flag)
            return xh, xl, st, flag
        r0, r1 = both(run)
        assert torch.equal(r0[0].view(torch.int16), r1[0].view(torch.int16)) and torch.equal(r0[1], r1[1]), (m, n, k)
        assert torch.equal(r0[2].view(torch.int32), r1[2].view(torch.int32)) and int(r0[3]) == int(r1[3]) == 0, (m, n, k)
    # the guard trips alike: one huge element in the old stream
    m, n, k = 16384, 768, 768                       # 384 tiles: covered by the role-split kernel
    a, w = mk(m, n, k)
    bias = torch.zeros(n, device=DEV)
    xh0 = torch.zeros(m, n, dtype=dt, device=DEV)
    xh0[4097, 300] = 30000.0

    def run():
        xh, xl = xh0.clone(), torch.full((m, n), 128, dtype=torch.uint8, device=DEV)
        st = torch.empty((m, n // 64, 2), device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.gemm_nt_res_stats(a, w, xh, xl, bias, st, flag=flag)
        return int(flag)
    f0, f1 = both(run)
    assert f0 == f1 and f0 != 0
