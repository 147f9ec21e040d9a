"""Per-kernel parity on the GPU: every C-ABI entry point against plain PyTorch fp32 of the same op.

Tolerances: fp32 kernels are compared at rtol/atol ~1e-5..1e-4 (fp32 summation-order noise only);
bf16 kernels are compared against an fp32 reference computed from the SAME bf16-rounded inputs, so
the only error left is the final bf16 rounding of the output (rel 2^-8 = 3.9e-3) plus fp32
accumulation order.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_OUT = dict(rtol=1e-2, atol=1e-2)


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def test_cast_and_transpose(vited, gpu):
    ops = vited.ops
    for n in (1, 3, 4, 1023, 4096 * 3 + 1):
        x = _rand((n,), gpu, n)
        assert torch.equal(ops.cast(x, torch.bfloat16), x.to(torch.bfloat16))
        assert torch.equal(ops.cast(x.to(torch.bfloat16), torch.float32), x.to(torch.bfloat16).float())
    for r, c in ((384, 1152), (33, 65), (1, 7), (1536, 384)):
        w = _rand((r, c), gpu, r * c)
        assert torch.equal(ops.cast_transpose(w, torch.bfloat16), w.t().contiguous().to(torch.bfloat16))
        assert torch.equal(ops.cast_transpose(w, torch.float32), w.t().contiguous())


def test_cast_weights_multi_tensor_matches_single_casts(vited, gpu):
    """vited_cast_weights: every shadow of every weight in one launch, bit-equal to the per-weight casts."""
    ops = vited.ops
    shapes = [(384, 1152), (4, 384), (384, 192), (100, 36), (65, 130), (1536, 384), (64, 64), (1, 8)]
    ws = [_rand(sh, gpu, 7 + i) for i, sh in enumerate(shapes)]
    entries = []
    for i, w in enumerate(ws):
        n = torch.full(w.shape, 7.0, device=gpu, dtype=torch.bfloat16) if i % 3 != 1 else None
        t = torch.full((w.shape[1], w.shape[0]), 7.0, device=gpu, dtype=torch.bfloat16) if i % 3 != 2 else None
        entries.append((w, n, t))
    plan = ops.WeightShadowPlan(entries)
    for rnd in range(2):   # second round: weights changed in place, same plan
        plan.run()
        for w, n, t in entries:
            if n is not None:
                assert torch.equal(n, w.to(torch.bfloat16))
            if t is not None:
                assert torch.equal(t, w.t().contiguous().to(torch.bfloat16))
        for w in ws:
            w.mul_(1.5).add_(0.25)


@pytest.mark.parametrize('S,p,C', [(64, 8, 3), (64, 32, 3), (128, 16, 3), (32, 8, 1)])
def test_patchify_matches_unfold(vited, gpu, S, p, C):
    ops = vited.ops
    pairs = _rand((5, 2, C, S, S), gpu, S + p)
    for img in (pairs[:, 0], pairs[:, 1], pairs[:, 0].contiguous()):
        ref = F.unfold(img.contiguous(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, C * p * p)
        assert torch.equal(ops.patchify(img, p, torch.float32), ref)
        assert torch.equal(ops.patchify(img, p, torch.bfloat16), ref.to(torch.bfloat16))
    idx = torch.tensor([4, 0, 0, 3, 1, 2, 2], device=gpu)
    img = pairs[:, 1]
    ref = F.unfold(img[idx].contiguous(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, C * p * p)
    assert torch.equal(ops.patchify(img, p, torch.float32, batch_index=idx), ref)


def test_slice_rows_cls_and_sum_rows(vited, gpu):
    ops = vited.ops
    x = _rand((7, 65, 384), gpu, 1)
    assert torch.equal(ops.slice_rows_cast(x, 1, 64, torch.float32), x[:, 1:].reshape(-1, 384))
    assert torch.equal(ops.slice_rows_cast(x, 1, 64, torch.bfloat16), x[:, 1:].reshape(-1, 384).to(torch.bfloat16))
    cls, pos = _rand((384,), gpu, 2), _rand((65, 384), gpu, 3)
    y = x.clone()
    ops.write_cls_row(y, cls, pos)
    assert torch.equal(y[:, 0], (cls + pos[0]).expand(7, -1)) and torch.equal(y[:, 1:], x[:, 1:])
    for b, w in ((1, 5), (17, 384), (1000, 1536), (4096, 384), (33, 65 * 384)):
        t = _rand((b, w), gpu, b + w)
        torch.testing.assert_close(ops.sum_rows(t), t.double().sum(0).float(), rtol=1e-5, atol=1e-4 * math.sqrt(b))
        tb = t.to(torch.bfloat16)
        torch.testing.assert_close(ops.sum_rows(tb), tb.double().sum(0).float(), rtol=1e-5, atol=1e-4 * math.sqrt(b))
    # determinism
    t = _rand((4096, 384), gpu, 9)
    assert torch.equal(ops.sum_rows(t), ops.sum_rows(t))


@pytest.mark.parametrize('rows,dim', [(1, 32), (65, 384), (1000, 384), (130, 768), (7, 48), (33280, 384), (24601, 384)])
def test_layernorm_fwd_bwd(vited, gpu, rows, dim):
    ops = vited.ops
    x = _rand((rows, dim), gpu, rows, 2.0) + 0.5
    g, b = _rand((dim,), gpu, 1) * 0.2 + 1, _rand((dim,), gpu, 2) * 0.1
    dy = _rand((rows, dim), gpu, 3)
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.layer_norm(xr, (dim,), gr, br, 1e-6)
    yr.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, torch.float32)
    torch.testing.assert_close(y, yr.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(mean, x.mean(1), rtol=1e-5, atol=1e-6)
    y16, _, _ = ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16)
    torch.testing.assert_close(y16.float(), yr.detach(), **BF16_OUT)
    dx_in = _rand((rows, dim), gpu, 4)
    dx, lp, dg, db = ops.layernorm_bwd(dy, x, g, mean, rstd, dx_in=dx_in, want_lp=True)
    torch.testing.assert_close(dx, xr.grad + dx_in, rtol=1e-4, atol=1e-5)
    assert torch.equal(lp, dx.to(torch.bfloat16))
    torch.testing.assert_close(dg, gr.grad, rtol=1e-4, atol=1e-4 * math.sqrt(rows))
    torch.testing.assert_close(db, br.grad, rtol=1e-4, atol=1e-4 * math.sqrt(rows))
    dx2, lp2, _, _ = ops.layernorm_bwd(dy.to(torch.bfloat16), x, g, mean, rstd)
    assert lp2 is None
    xr.grad = None
    F.layer_norm(xr, (dim,), g, b, 1e-6).backward(dy.to(torch.bfloat16).float())
    torch.testing.assert_close(dx2, xr.grad, rtol=1e-4, atol=1e-5)


def test_layernorm_strided_rows(vited, gpu):
    """Final norm on the cls rows only: row stride = N2 * D, gradient scattered into a zero tensor."""
    ops = vited.ops
    B, N, D = 9, 65, 384
    x3 = _rand((B, N, D), gpu, 5)
    g, b = _rand((D,), gpu, 6) + 1, _rand((D,), gpu, 7)
    y, mean, rstd = ops.layernorm_fwd(x3[:, 0, :], g, b, 1e-6, torch.float32)
    torch.testing.assert_close(y, F.layer_norm(x3[:, 0, :], (D,), g, b, 1e-6), rtol=1e-5, atol=1e-5)
    dy = _rand((B, D), gpu, 8)
    dx = torch.zeros_like(x3)
    lp = torch.zeros((B, N, D), dtype=torch.bfloat16, device=gpu)
    ops.layernorm_bwd(dy, x3[:, 0, :], g, mean, rstd, dx_out=dx[:, 0, :], dx_lp=lp[:, 0, :])
    xr = x3.clone().requires_grad_()
    F.layer_norm(xr[:, 0, :], (D,), g, b, 1e-6).backward(dy)
    torch.testing.assert_close(dx, xr.grad, rtol=1e-4, atol=1e-5)
    assert torch.equal(lp, dx.to(torch.bfloat16))


def _gemm_ref(a, w_nk, bias):
    acc = a.double() @ w_nk.double().t()
    if bias is not None:
        acc = acc + bias.double()
    return acc


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(256, 384, 384), (130, 1152, 384), (65, 384, 1536), (200, 128, 192), (33, 4, 384),
                                   (70, 32, 32), (128, 384, 64)])
def test_gemm_all_epilogues(vited, gpu, dtype, M, N, K):
    ops, L = vited.ops, vited._lib
    a = _rand((M, K), gpu, 1, dtype=dtype)
    w = _rand((N, K), gpu, 2, 1 / math.sqrt(K), dtype=dtype)
    bias = _rand((N,), gpu, 3)
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else BF16_OUT
    expect_mfma = dtype == torch.bfloat16 and K % 64 == 0 and N % 16 == 0
    ref = _gemm_ref(a, w, bias)
    out = ops.gemm(a, w, bias=bias)
    assert ops.last_paths()[0] == (2 if expect_mfma else 1)
    torch.testing.assert_close(out.double(), ref, **tol)
    torch.testing.assert_close(ops.gemm(a, w).double(), _gemm_ref(a, w, None), **tol)
    # B given as [K, N]
    out_kn = ops.gemm(a, w.t().contiguous(), b_layout=L.B_KN, bias=bias)
    torch.testing.assert_close(out_kn.double(), ref, **tol)
    # GELU: pre-activation + activation
    z, u = ops.gemm(a, w, epilogue=L.EPI_GELU, bias=bias)
    torch.testing.assert_close(z.double(), ref, **tol)
    torch.testing.assert_close(u.double(), F.gelu(ref), **tol)
    # fp32 store
    o32 = ops.gemm(a, w, epilogue=L.EPI_STORE_F32, bias=bias)
    assert o32.dtype == torch.float32
    torch.testing.assert_close(o32.double(), ref, rtol=2e-4, atol=2e-4)
    # residual (fp32 stream)
    res = _rand((M, N), gpu, 4)
    y = ops.gemm(a, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=res)
    assert y.dtype == torch.float32
    torch.testing.assert_close(y.double(), ref + res.double(), rtol=2e-4, atol=2e-4)
    # multiply by gelu'(aux)
    aux = _rand((M, N), gpu, 5, dtype=dtype)
    xg = aux.double().requires_grad_()
    F.gelu(xg).sum().backward()
    dz = ops.gemm(a, w, epilogue=L.EPI_MUL_GELU_GRAD, aux=aux)
    torch.testing.assert_close(dz.double(), _gemm_ref(a, w, None) * xg.grad, **tol)


@pytest.mark.parametrize('M,N,K', [(8192, 1152, 384), (8321, 768, 384), (16640, 1536, 384), (8257, 384, 1536), (8320, 384, 384),
                                   (9000, 384, 1152),
                                   # the 256 x 128 tile under the GELU' and multiply epilogues (fc1 / dz at M >= 8,192, N >= 768,
                                   # K <= 512): a ragged last row tile, the narrowest N and the shortest / a non-384 contraction
                                   (16555, 1536, 384), (16391, 768, 128), (16400, 1024, 512)])
def test_gemm_bench_scale_tiles(vited, gpu, M, N, K):
    """The tile variants only large launches select (256-row / 8-wave tiles for M >= 8192 - plain store, fc1 + GELU' and the
    dz multiply -, BK = 64 at N = 384, the XCD-aware tile order over thousands of tiles) with ragged last row tiles
    (M % 128 != 0): every epilogue, every output element, against fp64 on the same bf16-rounded operands; plus the
    weight-gradient kernel's full split-K geometry."""
    ops, L = vited.ops, vited._lib
    a = _rand((M, K), gpu, 21, dtype=torch.bfloat16)
    w = _rand((N, K), gpu, 22, 1 / math.sqrt(K), dtype=torch.bfloat16)
    bias = _rand((N,), gpu, 23)
    ref = _gemm_ref(a, w, bias)
    out = ops.gemm(a, w, bias=bias)
    assert ops.last_paths()[0] == 2
    torch.testing.assert_close(out.double(), ref, **BF16_OUT)
    z, u = ops.gemm(a, w, epilogue=L.EPI_GELU, bias=bias)
    torch.testing.assert_close(z.double(), ref, **BF16_OUT)
    torch.testing.assert_close(u.double(), F.gelu(ref), **BF16_OUT)
    res = _rand((M, N), gpu, 24)
    y = ops.gemm(a, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=res)
    torch.testing.assert_close(y.double(), ref + res.double(), rtol=2e-4, atol=2e-4)
    aux = _rand((M, N), gpu, 25, dtype=torch.bfloat16)
    xg = aux.double().requires_grad_()
    F.gelu(xg).sum().backward()
    dz = ops.gemm(a, w, epilogue=L.EPI_MUL_GELU_GRAD, aux=aux)
    torch.testing.assert_close(dz.double(), _gemm_ref(a, w, None) * xg.grad, **BF16_OUT)
    # what fc1 saves in training (gelu'(z), gelu(z)) and the GELU backward as one multiply (dz = (dy . W2^T) * gelu'(z))
    gd, u2 = ops.gemm(a, w, epilogue=L.EPI_GELU_GRAD, bias=bias)
    zg = ref.clone().requires_grad_()
    F.gelu(zg).sum().backward()
    torch.testing.assert_close(gd.double(), zg.grad, **BF16_OUT)
    torch.testing.assert_close(u2.double(), F.gelu(ref), **BF16_OUT)
    dzm = ops.gemm(a, w, epilogue=L.EPI_MUL, aux=aux)
    torch.testing.assert_close(dzm.double(), _gemm_ref(a, w, None) * aux.double(), **BF16_OUT)
    o32 = ops.gemm(a, w, epilogue=L.EPI_STORE_F32)
    torch.testing.assert_close(o32.double(), _gemm_ref(a, w, None), rtol=2e-4, atol=2e-4)
    # dW = dY^T X and dbias over all M rows (512 workgroups, ragged last split), also accumulated onto existing values
    dy = _rand((M, N), gpu, 26, dtype=torch.bfloat16)
    dw, db = ops.linear_bwd_weight(dy, a)
    dw_ref = dy.double().t() @ a.double()
    torch.testing.assert_close(dw.double(), dw_ref, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=2e-4, atol=2e-3)
    acc_w, acc_b = torch.ones(N, K, device=gpu), torch.full((N,), 2.0, device=gpu)
    ops.linear_bwd_weight(dy, a, dw_out=acc_w, db_out=acc_b)
    torch.testing.assert_close(acc_w.double(), dw_ref + 1, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(acc_b.double(), dy.double().sum(0) + 2, rtol=2e-4, atol=2e-3)


# ---------------------------------------------------------------------------------------------
# row-complete Linear + LayerNorm kernels (gemm_row.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,K', [(64, 384), (80, 1536), (65 * 8, 384), (1000, 1152), (4096 + 33, 768), (65 * 64, 1536), (16384, 384), (13, 64), (200, 192),
                                  # the 96-row tile (one round of 256 tiles) and the 160-row tile (one round where 128 / 144 rows need two)
                                  (24001, 384), (39993, 1152)])
def test_linear_residual_layernorm_fwd(vited, gpu, M, K):
    """y = residual + a W^T + b and h = LayerNorm(y) in one kernel against fp64 on the same bf16-rounded operands: ragged M
    (partial last tile, both tile heights: 64-row tiles and the 80-row tiles picked for 65-row batches), every K of the step."""
    ops = vited.ops
    N = 384
    a = _rand((M, K), gpu, 1 + M, 1.0, torch.bfloat16)
    w = _rand((N, K), gpu, 2 + K, K ** -0.5, torch.bfloat16)
    bias, res = _rand((N,), gpu, 3, 0.5), _rand((M, N), gpu, 4, 2.0)
    gamma, beta = 1.0 + _rand((N,), gpu, 5, 0.2), _rand((N,), gpu, 6, 0.2)
    assert ops.linear_layernorm_supported(M, N, K, torch.bfloat16)
    y, h, mean, rstd = ops.linear_residual_layernorm_fwd(a, w, bias, res, gamma, beta, 1e-6)
    ref = res.double() + a.double() @ w.double().t() + bias.double()
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=2e-5 * K ** 0.5)
    mu = ref.mean(dim=1)
    var = ref.var(dim=1, unbiased=False)
    torch.testing.assert_close(mean.double(), mu, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd.double(), (var + 1e-6).rsqrt(), rtol=1e-4, atol=1e-6)
    href = (ref - mu[:, None]) * (var[:, None] + 1e-6).rsqrt() * gamma.double() + beta.double()
    torch.testing.assert_close(h.double(), href, **BF16_OUT)
    # the same numbers as the two-kernel form it replaces (same MFMA order, same LayerNorm arithmetic): bit for bit
    if K % 64 == 0:      # (other K: the two-kernel form runs the portable fp32-FMA GEMM, another summation order)
        y2 = ops.gemm(a, w, epilogue=vited._lib.EPI_RESIDUAL, bias=bias, residual=res)
        h2, m2, r2 = ops.layernorm_fwd(y2, gamma, beta, 1e-6, torch.bfloat16)
        assert torch.equal(y, y2) and torch.equal(h, h2) and torch.equal(mean, m2) and torch.equal(rstd, r2)
    # without a LayerNorm (h == null) and in place on the residual
    y3, h3, _, _ = ops.linear_residual_layernorm_fwd(a, w, None, res.clone(), None, None)
    assert h3 is None
    torch.testing.assert_close(y3.double(), ref - bias.double(), rtol=1e-5, atol=2e-5 * K ** 0.5)
    inplace = res.clone()
    ops.linear_residual_layernorm_fwd(a, w, bias, inplace, gamma, beta, 1e-6, out=inplace)
    assert torch.equal(inplace, y)


def test_slot_schedule_kernels_give_the_same_bits_on_every_launch(vited, gpu):
    """Race screen for the kernels whose LDS buffers are recycled behind counted waits and raw barriers (gemm_row.hip, the wide
    weight-gradient kernel): an LDS-DMA that lands late, or a buffer rewritten early, shows as a result that differs between
    launches.  Full-chip shapes (two tiles per CU, both tile heights), every launch preceded by unrelated memory traffic that
    shifts the timing, 25 launches each, every output bit-identical to the first launch and correct against fp64 samples."""
    ops = vited.ops
    N = 384
    noise = torch.empty(64 << 20, device=gpu)
    for M, K in ((65536, 1536), (66560, 384), (65 * 1024, 1152), (65536, 768), (24576, 1536), (72 * 1025, 384)):
        a = _rand((M, K), gpu, 11, 1.0, torch.bfloat16)
        w = _rand((N, K), gpu, 12, K ** -0.5, torch.bfloat16)
        bias, res = _rand((N,), gpu, 13, 0.5), _rand((M, N), gpu, 14, 2.0)
        gamma, beta = 1.0 + _rand((N,), gpu, 15, 0.2), _rand((N,), gpu, 16, 0.2)
        first = None
        for it in range(25):
            noise.fill_(float(it))
            y, h, mean, rstd = ops.linear_residual_layernorm_fwd(a, w, bias, res, gamma, beta, 1e-6)
            dx, dx_lp, dg, db = ops.linear_layernorm_bwd(a, w, res, gamma, mean, rstd, want_lp=True)
            out = (y, h, mean, rstd, dx, dx_lp, dg, db)
            if first is None:
                first = [t.clone() for t in out]
                rows = torch.tensor([0, 1, 127, 128, 143, 144, M // 2 + 5, M - 130, M - 1], device=gpu)
                ref = res[rows].double() + a[rows].double() @ w.double().t() + bias.double()
                torch.testing.assert_close(y[rows].double(), ref, rtol=1e-5, atol=2e-5 * K ** 0.5)
            else:
                for t, f in zip(out, first):
                    assert torch.equal(t, f), (M, K, it)
    # the batched weight-gradient launch of one encoder block at the bench batch
    m = 65536
    shapes = [(m, 1152, 384), (m, 384, 384), (m, 1536, 384), (m, 384, 1536)]
    items = []
    for i, (mm, n, k) in enumerate(shapes):
        items.append((_rand((mm, n), gpu, 40 + i, 1.0, torch.bfloat16), _rand((mm, k), gpu, 50 + i, 1.0, torch.bfloat16),
                      torch.empty(n, k, device=gpu), torch.empty(n, device=gpu)))
    first = None
    for it in range(25):
        noise.fill_(float(it))
        assert ops.linear_bwd_weight_batched(items, False)
        if first is None:
            first = [(dw.clone(), db.clone()) for _, _, dw, db in items]
            dy, x, dw, db = items[1]
            torch.testing.assert_close(dw.double(), dy.double().t() @ x.double(), rtol=2e-4, atol=1e-2)
        else:
            for (_, _, dw, db), (fw, fb) in zip(items, first):
                assert torch.equal(dw, fw) and torch.equal(db, fb), it


@pytest.mark.parametrize('M,K', [(64, 384), (80, 1536), (65 * 8, 1152), (1000, 768), (4096 + 33, 1536), (65 * 64, 384), (16384, 1152), (13, 64), (300, 192),
                                  (24001, 768), (39993, 384)])
def test_linear_layernorm_bwd(vited, gpu, M, K):
    """dx = dx_in + LN'(dy Wt^T) with the column sums, one kernel, against fp64 autograd through LayerNorm on the same
    bf16-rounded operands.  d(LayerNorm output) stays fp32 inside the kernel, so it is MORE accurate than the two-kernel form
    (which rounds it to bf16): both are checked against fp64, the fused one at the tighter tolerance."""
    ops = vited.ops
    N = 384
    dy = _rand((M, K), gpu, 11 + M, 1.0, torch.bfloat16)
    wt = _rand((N, K), gpu, 12 + K, K ** -0.5, torch.bfloat16)
    x = _rand((M, N), gpu, 13, 1.5) + 0.3
    gamma = 1.0 + _rand((N,), gpu, 14, 0.2)
    dx_in = _rand((M, N), gpu, 15, 1.0)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(N, dtype=torch.float64, device=gpu, requires_grad=True)
    yd = F.layer_norm(xd, (N,), gd, bd, 1e-6)
    dh = dy.double() @ wt.double().t()
    yd.backward(dh)
    mu = x.double().mean(dim=1)
    rs = (x.double().var(dim=1, unbiased=False) + 1e-6).rsqrt()
    mean, rstd = mu.float(), rs.float()
    dx, dx_lp, dg, db = ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dx_in=dx_in, want_lp=True)
    want = xd.grad + dx_in.double()
    scale = float(want.abs().max())
    torch.testing.assert_close(dx.double(), want, rtol=1e-4, atol=1e-5 * scale)
    assert torch.equal(dx_lp, dx.to(torch.bfloat16))
    torch.testing.assert_close(dg.double(), gd.grad, rtol=1e-4, atol=1e-5 * float(gd.grad.abs().max()) * M ** 0.5)
    torch.testing.assert_close(db.double(), bd.grad, rtol=1e-4, atol=1e-5 * float(bd.grad.abs().max()) * M ** 0.5)
    # no incoming residual gradient, no bf16 copy, accumulate onto existing column sums, dx written in place of dx_in
    acc_g, acc_b = torch.full((N,), 2.0, device=gpu), torch.full((N,), -1.0, device=gpu)
    dx2, lp2, _, _ = ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dgamma=acc_g, dbeta=acc_b)
    assert lp2 is None
    torch.testing.assert_close(dx2.double(), xd.grad, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(acc_g - 2.0, dg, rtol=1e-4, atol=1e-4 * float(dg.abs().max()))
    torch.testing.assert_close(acc_b + 1.0, db, rtol=1e-4, atol=1e-4 * float(db.abs().max()))
    buf = dx_in.clone()
    ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dx_in=buf, dx_out=buf)
    assert torch.equal(buf, dx)
    # deferred column sums: two LayerNorms finished by ONE launch (overwrite and accumulate), same numbers as the immediate form
    queue = []
    d1, _, g1, b1 = ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dx_in=dx_in, defer=queue)
    acc_g2, acc_b2 = torch.full((N,), 2.0, device=gpu), torch.full((N,), -1.0, device=gpu)
    ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dgamma=acc_g2, dbeta=acc_b2, defer=queue)
    assert len(queue) == 2 and torch.equal(d1, dx)
    ops.layernorm_bwd_finish(queue)
    assert not queue and torch.equal(g1, dg) and torch.equal(b1, db) and torch.equal(acc_g2, acc_g) and torch.equal(acc_b2, acc_b)
    # the two-kernel form, for scale: same result within the bf16 rounding of dh it adds
    dh_lp = ops.gemm(dy, wt)
    dx_u, _, dg_u, _ = ops.layernorm_bwd(dh_lp, x, gamma, mean, rstd, dx_in=dx_in)
    torch.testing.assert_close(dx_u, dx, rtol=2e-2, atol=2e-2 * scale)


def test_linear_layernorm_rejects_other_shapes(vited, gpu):
    ops = vited.ops
    assert not ops.linear_layernorm_supported(128, 768, 384, torch.bfloat16)      # N != 384
    assert not ops.linear_layernorm_supported(128, 384, 48, torch.bfloat16)       # K % 32
    assert not ops.linear_layernorm_supported(128, 384, 384, torch.float32)       # fp32 exact path keeps the separate kernels
    a, w = _rand((64, 384), gpu, 1, 1.0, torch.bfloat16), _rand((768, 384), gpu, 2, 0.05, torch.bfloat16)
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.linear_residual_layernorm_fwd(a, w, None, torch.zeros(64, 768, device=gpu))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_patch_embed_row_remap(vited, gpu, dtype):
    """RESIDUAL epilogue with the cls-row remap and the broadcast pos_embed table."""
    ops, L = vited.ops, vited._lib
    B, N1, D, K = 6, 64, 384, 192
    patches = _rand((B * N1, K), gpu, 1, dtype=dtype)
    w = _rand((D, K), gpu, 2, 0.1, dtype=dtype)
    bias, pos = _rand((D,), gpu, 3), _rand((N1 + 1, D), gpu, 4)
    tok = (_gemm_ref(patches, w, bias)).view(B, N1, D)
    x = ops.gemm(patches, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=pos, rows_per_batch=N1,
                 out_rows_per_batch=N1 + 1, row_offset=1, residual_bcast=True, out_rows=B * (N1 + 1))
    torch.testing.assert_close(x.view(B, N1 + 1, D)[:, 1:].double(), tok + pos[1:].double(), rtol=2e-4, atol=2e-4)
    x1 = ops.gemm(patches, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=pos[1:], rows_per_batch=N1,
                  out_rows_per_batch=N1, row_offset=0, residual_bcast=True, out_rows=B * N1)
    torch.testing.assert_close(x1.view(B, N1, D).double(), tok + pos[1:].double(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(4096, 384, 384), (1000, 1152, 384), (777, 384, 1536), (640, 384, 192), (65, 4, 384),
                                   (130, 32, 32), (50, 768, 384),
                                   # the wide (128 x 384) tile: ragged last m-stage, ragged n-tile, several k-panels
                                   (4099, 1152, 384), (5000, 200, 768), (4128, 384, 1536), (8192, 8, 384)])
def test_linear_bwd_weight(vited, gpu, dtype, M, N, K):
    ops = vited.ops
    dy = _rand((M, N), gpu, 1, dtype=dtype)
    x = _rand((M, K), gpu, 2, dtype=dtype)
    dw, db = ops.linear_bwd_weight(dy, x)
    expect_mfma = dtype == torch.bfloat16 and N % 8 == 0 and K % 8 == 0
    assert ops.last_paths()[0] == (2 if expect_mfma else 1)
    ref = dy.double().t() @ x.double()
    torch.testing.assert_close(dw.double(), ref, rtol=1e-4, atol=1e-4 * math.sqrt(M))
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-4, atol=1e-4 * math.sqrt(M))
    dw2, db2 = ops.linear_bwd_weight(dy, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)  # deterministic (slabs, no atomics)


@pytest.mark.parametrize('accumulate', [False, True])
def test_linear_bwd_weight_batched(vited, gpu, accumulate):
    """The weight gradients of one transformer block in ONE launch (vited_linear_bwd_weight_batched): the seven products of a
    decoder block at 65-token batches - different N, K, row counts (the kv projection sees the 64-token context) and a strided
    operand - against fp64 on the same bf16-rounded operands, overwriting and accumulating; and the refusal of sets the wide
    kernel does not cover."""
    ops = vited.ops
    m2, m1 = 65 * 80, 64 * 80           # 5,200 / 5,120 rows
    shapes = [(m2, 1152, 384, True), (m2, 384, 384, True), (m2, 384, 384, True), (m1, 768, 384, True), (m2, 384, 384, False),
              (m2, 1536, 384, True), (m2, 384, 1536, True)]
    items, refs = [], []
    for i, (m, n, k, bias) in enumerate(shapes):
        dy = _rand((m, n), gpu, 100 + i, 1.0, torch.bfloat16)
        if i == 1:                      # a column slice of a wider tensor: row stride != N
            wide = _rand((m, 2 * n), gpu, 200 + i, 1.0, torch.bfloat16)
            dy = wide[:, n:]
        x = _rand((m, k), gpu, 300 + i, 1.0, torch.bfloat16)
        dw = torch.full((n, k), 0.5 if accumulate else float('nan'), device=gpu)
        db = torch.full((n,), -0.25 if accumulate else float('nan'), device=gpu) if bias else None
        items.append((dy, x, dw, db))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    assert ops.linear_bwd_weight_batched(items, accumulate)
    for (dy, x, dw, db), (rw, rb) in zip(items, refs):
        base_w, base_b = (0.5, -0.25) if accumulate else (0.0, 0.0)
        torch.testing.assert_close(dw.double(), rw + base_w, rtol=2e-4, atol=2e-3 * (x.shape[0] / 4096) ** 0.5)
        if db is not None:
            torch.testing.assert_close(db.double(), rb + base_b, rtol=2e-4, atol=2e-3)
    # not covered: K not a multiple of 384, too few rows, or an fp32 operand -> nothing is launched
    small = (_rand((512, 384), gpu, 1, 1.0, torch.bfloat16), _rand((512, 384), gpu, 2, 1.0, torch.bfloat16), torch.zeros(384, 384, device=gpu), None)
    assert not ops.linear_bwd_weight_batched([items[0], small], accumulate)
    odd = (_rand((m2, 384), gpu, 3, 1.0, torch.bfloat16), _rand((m2, 192), gpu, 4, 1.0, torch.bfloat16), torch.zeros(384, 192, device=gpu), None)
    assert not ops.linear_bwd_weight_batched([items[0], odd], accumulate)
    assert not ops.linear_bwd_weight_batched(items * 6, accumulate)               # more than 40 products


def test_linear_bwd_weight_wide_strided_operands(vited, gpu):
    """dW of the wide tile from column views of wider buffers (the fused qkv gradient / hidden activations are read in place)."""
    ops = vited.ops
    M = 4500
    dyb = _rand((M, 1152 + 64), gpu, 1, dtype=torch.bfloat16)
    xb = _rand((M, 768 + 32), gpu, 2, dtype=torch.bfloat16)
    dy, x = dyb[:, 64:], xb[:, 32:]
    dw, db = ops.linear_bwd_weight(dy, x)
    assert ops.last_paths()[0] == 2
    torch.testing.assert_close(dw.double(), dy.double().t() @ x.double(), rtol=1e-4, atol=1e-4 * math.sqrt(M))
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-4, atol=1e-4 * math.sqrt(M))


def _sdpa_ref(q, k, v, heads, scale):
    B, Nq, D = q.shape
    hd = D // heads
    qh, kh, vh = (t.double().view(B, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * scale
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, D), torch.logsumexp(s, -1)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,Nq,Nk,hd', [(3, 12, 64, 64, 32), (3, 12, 65, 65, 32), (2, 12, 65, 64, 32), (2, 6, 257, 256, 64),
                                          (1, 6, 1025, 1024, 64), (2, 1, 5, 4, 32), (1, 2, 1, 1, 64), (2, 3, 200, 130, 32),
                                          (1, 6, 1024, 1024, 64), (1, 6, 1025, 1025, 64), (2, 2, 65, 65, 64), (1, 4, 129, 81, 32),
                                          (300, 12, 65, 64, 32), (257, 6, 64, 64, 32), (129, 5, 65, 65, 32),
                                          # 65 = 4 x 16 + 1: the one-extra-row path of the backward (query, key, both; cls-only queries)
                                          (7, 12, 64, 65, 32), (5, 12, 1, 65, 32), (9, 7, 16, 65, 32), (1030, 12, 65, 65, 32)])
def test_attention_fwd_bwd(vited, gpu, dtype, B, H, Nq, Nk, hd):
    ops = vited.ops
    D = H * hd
    scale = hd ** -0.5
    self_attn = Nq == Nk
    if self_attn:  # packed qkv projection [B, N, 3, H, hd], consumed in place
        qkv = _rand((B, Nq, 3 * D), gpu, 1, dtype=dtype)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    else:          # q [B, Nq, D] + packed kv [B, Nk, 2, H, hd]
        q = _rand((B, Nq, D), gpu, 1, dtype=dtype)
        kv = _rand((B, Nk, 2 * D), gpu, 2, dtype=dtype)
        k, v = kv[:, :, :D], kv[:, :, D:]
    o, lse = ops.attention_fwd(q, k, v, H, scale)
    mfma = dtype == torch.bfloat16                              # bf16: short-sequence or tiled (flash) MFMA kernels
    assert ops.last_paths()[1] == (2 if mfma else 1)
    qr, kr, vr = (t.double().clone().requires_grad_() for t in (q, k, v))
    o_ref, lse_ref = _sdpa_ref(qr, kr, vr, H, scale)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else BF16_OUT
    torch.testing.assert_close(o.double(), o_ref.detach(), **tol)
    torch.testing.assert_close(lse.double(), lse_ref.detach(), rtol=1e-4, atol=1e-4)
    do = _rand((B, Nq, D), gpu, 3, dtype=dtype)
    o_ref.backward(do.double())
    if self_attn:
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:]
    else:
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[:, :, :D], dkv[:, :, D:]
    # the kernels consume the SAVED (storage-dtype) output o, like the reference's SDPA backward
    ops.attention_bwd(q, k, v, o, do, lse, H, scale, dq, dk, dv)
    assert ops.last_paths()[1] == (2 if mfma else 1)
    btol = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dq.double(), qr.grad, **btol)
    torch.testing.assert_close(dk.double(), kr.grad, **btol)
    torch.testing.assert_close(dv.double(), vr.grad, **btol)


def test_ops_refuse_cpu_tensors(vited, gpu):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        vited.ops.cast(torch.zeros(4), torch.bfloat16)


# ---------------------------------------------------------------------------------------------
# fused MLP branch (vited_mlp_fwd): y = x + fc2(gelu(fc1(LN(x))))  (vision_transformer.py:126,271; timm Mlp)
# ---------------------------------------------------------------------------------------------
def _mlp_case(gpu, rows, seed, ld_pad=0):
    d, hid = 384, 1536
    xs = _rand((rows, d + ld_pad), gpu, seed)[:, :d]                    # optionally row-strided
    gamma, beta = 1.0 + 0.2 * _rand((d,), gpu, seed + 1), 0.1 * _rand((d,), gpu, seed + 2)
    w1, b1 = _rand((hid, d), gpu, seed + 3, 0.06), 0.1 * _rand((hid,), gpu, seed + 4)
    w2, b2 = _rand((d, hid), gpu, seed + 5, 0.03), 0.1 * _rand((d,), gpu, seed + 6)
    return xs, gamma, beta, w1, b1, w2, b2


@pytest.mark.parametrize('rows', [128, 65 * 3, 1, 16, 4160, 640 + 17])
def test_mlp_fused_forward_and_saved_tensors(vited, gpu, rows):
    """The fused kernel against plain PyTorch fp32 of the same op chain evaluated on the SAME bf16-rounded operands
    (LayerNorm output, weights, hidden activation rounded where the kernel rounds them), ragged row counts included:
    y within 2e-3 (fp32 accumulation order only), saved bf16 tensors within one bf16 ulp of the reference values."""
    ops = vited.ops
    x, gamma, beta, w1, b1, w2, b2 = _mlp_case(gpu, rows, 100 + rows, ld_pad=8 if rows == 128 else 0)
    w1b, w2b = w1.to(torch.bfloat16), w2.to(torch.bfloat16)
    y, (mean, rstd, h, gd, u) = ops.mlp_fwd(x, gamma, beta, w1b, b1, w2b, b2, 1e-6, save=True)
    y2, none = ops.mlp_fwd(x, gamma, beta, w1b, b1, w2b, b2, 1e-6, save=False)
    assert none is None and torch.equal(y, y2)
    xr = x.contiguous()
    mu, var = xr.mean(-1), xr.var(-1, unbiased=False)
    h_ref = F.layer_norm(xr, (384,), gamma, beta, 1e-6)
    torch.testing.assert_close(mean, mu, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd, (var + 1e-6).rsqrt(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(h.float(), h_ref, **BF16_OUT)
    z = h.float() @ w1b.float().t() + b1                               # from the kernel's own (bf16) h: isolates fc1
    u_ref = F.gelu(z)
    cdf = 0.5 * (1 + torch.erf(z / math.sqrt(2)))
    gd_ref = cdf + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    torch.testing.assert_close(u.float(), u_ref, **BF16_OUT)
    torch.testing.assert_close(gd.float(), gd_ref, **BF16_OUT)
    y_ref = xr + u.float() @ w2b.float().t() + b2                      # from the kernel's own (bf16) u: isolates fc2 + residual
    torch.testing.assert_close(y, y_ref, rtol=2e-3, atol=2e-3)
    # and end to end against the unfused kernels of the same library
    hh, m2, r2 = ops.layernorm_fwd(xr, gamma, beta, 1e-6, torch.bfloat16)
    gd2, u2 = ops.gemm(hh, w1b, epilogue=vited._lib.EPI_GELU_GRAD, bias=b1)
    y3 = ops.gemm(u2, w2b, epilogue=vited._lib.EPI_RESIDUAL, bias=b2, residual=xr)
    torch.testing.assert_close(y, y3, rtol=5e-3, atol=5e-3)   # u differs by one bf16 ulp in places (fp32 summation order before the rounding)


def test_mlp_fused_rejects_other_shapes(vited, gpu):
    ops = vited.ops
    x = _rand((64, 256), gpu, 1)
    with pytest.raises(RuntimeError, match='unsupported'):
        ops.mlp_fwd(x, torch.ones(256, device=gpu), torch.zeros(256, device=gpu), _rand((1024, 256), gpu, 2).bfloat16(), torch.zeros(1024, device=gpu),
                    _rand((256, 1024), gpu, 3).bfloat16(), torch.zeros(256, device=gpu), 1e-6)


@pytest.mark.parametrize('batch,tokens,heads', [(256, 64, 12), (5, 65, 12), (3, 256, 6)])
def test_block_fwd_matches_the_oracle_block(vited, gpu, batch, tokens, heads):
    """vited_block_fwd (Block.forward, vision_transformer.py:124-127, as one C-ABI call) against the CPU oracle's
    ``encoder_block`` on the same fp32 parameters: bf16 tolerance (3e-2), and identical to the op-by-op HIP sequence."""
    from oracle import vited_oracle as vo
    torch.manual_seed(batch + tokens)
    s = vo.ViTEDShape(num_heads=heads)
    blk = vo._encoder_bag(s)
    for p in blk.parameters():
        if p.dim() > 1:
            torch.nn.init.trunc_normal_(p, std=.04)
        else:
            torch.nn.init.normal_(p, std=.05)
    blk.norm1.weight.data.add_(1.0)
    blk.norm2.weight.data.add_(1.0)
    x = torch.randn(batch, tokens, 384)
    with torch.no_grad():
        want = vo.encoder_block(blk, x, heads)
    g = lambda t: t.detach().to(gpu)
    bf = lambda t: t.detach().to(gpu).to(torch.bfloat16).contiguous()
    y = vited.ops.block_fwd(g(x), heads, g(blk.norm1.weight), g(blk.norm1.bias), bf(blk.attn.qkv.weight), g(blk.attn.qkv.bias),
                            bf(blk.attn.proj.weight), g(blk.attn.proj.bias), g(blk.norm2.weight), g(blk.norm2.bias), bf(blk.mlp.fc1.weight),
                            g(blk.mlp.fc1.bias), bf(blk.mlp.fc2.weight), g(blk.mlp.fc2.bias))
    torch.testing.assert_close(y.cpu(), want, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('batch,tokens,ctx_tokens,heads', [(64, 65, 64, 12), (5, 65, 64, 12), (2, 257, 256, 6)])
def test_cross_block_fwd_matches_the_oracle_block(vited, gpu, batch, tokens, ctx_tokens, heads):
    """vited_cross_block_fwd (CrossBlock.forward, vision_transformer.py:268-272, as one C-ABI call) against the CPU oracle's
    ``decoder_block`` on the same fp32 parameters: bf16 tolerance (3e-2).  65 / 64 tokens at 12 x 32 heads (config A; batch 64 =
    4,160 rows takes the row-complete proj + norm_cross kernel and the fused MLP) and 257 / 256 tokens at 6 x 64 (config H's
    head shape: the flash attention kernels)."""
    from oracle import vited_oracle as vo
    torch.manual_seed(batch + tokens)
    s = vo.ViTEDShape(num_heads=heads)
    blk = vo._decoder_bag(s)
    for p in blk.parameters():
        if p.dim() > 1:
            torch.nn.init.trunc_normal_(p, std=.04)
        else:
            torch.nn.init.normal_(p, std=.05)
    for nm in (blk.norm1, blk.norm_cross, blk.norm_context, blk.norm2):
        nm.weight.data.add_(1.0)
    x, ctx = torch.randn(batch, tokens, 384), torch.randn(batch, ctx_tokens, 384)
    with torch.no_grad():
        want = vo.decoder_block(blk, x, ctx, heads)
    g = lambda t: t.detach().to(gpu)
    bf = lambda t: t.detach().to(gpu).to(torch.bfloat16).contiguous()
    ln = lambda m: (g(m.weight), g(m.bias))
    ca = blk.cross_attn
    y = vited.ops.cross_block_fwd(g(x), g(ctx), heads, ln(blk.norm1), bf(blk.attn.qkv.weight), g(blk.attn.qkv.bias), bf(blk.attn.proj.weight),
                                  g(blk.attn.proj.bias), ln(blk.norm_cross), ln(blk.norm_context), bf(ca.q.weight), g(ca.q.bias),
                                  bf(ca.kv.weight), g(ca.kv.bias), bf(ca.proj.weight), g(ca.proj.bias), ln(blk.norm2),
                                  bf(blk.mlp.fc1.weight), g(blk.mlp.fc1.bias), bf(blk.mlp.fc2.weight), g(blk.mlp.fc2.bias))
    torch.testing.assert_close(y.cpu(), want, rtol=3e-2, atol=3e-2)


def test_context_fold_unfold_and_segmented_backward(vited, gpu):
    """csrc/context_fold.hip + the segmented row-complete backward: norm_context + kv of L decoder blocks through folded
    weights.  kv_l = LN(x; g_l, b_l) W_l^T + c_l must equal xhat (W_l o g_l)^T + (W_l b_l + c_l); the gradients that come back
    through the folded weights (unfold) and the one-kernel d(features) over L separate d(kv) tensors are compared with fp64
    autograd through the unfolded formulation."""
    ops = vited.ops
    L, D, N2, M = 3, 384, 768, 1000
    ws = [_rand((N2, D), gpu, 10 + l, 0.05) for l in range(L)]
    cs = [_rand((N2,), gpu, 20 + l, 0.1) for l in range(L)]
    gs = [1.0 + _rand((D,), gpu, 30 + l, 0.2) for l in range(L)]
    bs = [_rand((D,), gpu, 40 + l, 0.2) for l in range(L)]
    wf, wft, bf = ops.fold_context_weights(ws, cs, gs, bs)
    for l in range(L):
        want = (ws[l] * gs[l][None, :]).to(torch.bfloat16)
        assert torch.equal(wf[l * N2:(l + 1) * N2], want) and torch.equal(wft[:, l * N2:(l + 1) * N2], want.t())
        torch.testing.assert_close(bf[l * N2:(l + 1) * N2], cs[l] + ws[l] @ bs[l], rtol=1e-5, atol=1e-5)
    # forward equivalence in fp64 (the fold is exact algebra)
    x = _rand((M, D), gpu, 50, 1.5) + 0.2
    xd = x.double().requires_grad_(True)
    prm = [[t.double().requires_grad_(True) for t in (ws[l], cs[l], gs[l], bs[l])] for l in range(L)]
    dkv = torch.stack([_rand((M, N2), gpu, 60 + l, 1.0, torch.bfloat16) for l in range(L)])          # [L, M, N2]
    outs = [F.linear(F.layer_norm(xd, (D,), g, b, 1e-6), w, c) for w, c, g, b in prm]
    torch.autograd.backward(outs, [dkv[l].double() for l in range(L)])
    mu, var = x.double().mean(1), x.double().var(1, unbiased=False)
    xhat = ((x.double() - mu[:, None]) * (var[:, None] + 1e-6).rsqrt())
    torch.testing.assert_close(xhat @ (ws[0].double() * gs[0].double()).t() + (cs[0].double() + ws[0].double() @ bs[0].double()),
                               outs[0].detach(), rtol=1e-9, atol=1e-9)
    # d(features): ONE kernel contracting over the L separate d(kv) tensors against the transposed folded weights
    mean, rstd = mu.float(), (var + 1e-6).rsqrt().float()
    ones = torch.ones(D, device=gpu)
    dx, _, _, _ = ops.linear_layernorm_bwd(dkv, wft, x, ones, mean, rstd)
    torch.testing.assert_close(dx.double(), xd.grad, rtol=2e-2, atol=2e-2 * float(xd.grad.abs().max()))       # bf16 folded weights
    dx_cat, _, _, _ = ops.linear_layernorm_bwd(dkv.permute(1, 0, 2).reshape(M, L * N2).contiguous(), wft, x, ones, mean, rstd)
    assert torch.equal(dx, dx_cat), 'segmented operand != the same columns as one tensor'
    # unfold: gradients of the folded weights / bias -> dW, dc, dgamma, dbeta per block
    dwf = torch.cat([dkv[l].double().t() @ xhat for l in range(L)]).float()
    dbf = torch.cat([dkv[l].double().sum(0) for l in range(L)]).float()
    for accumulate in (False, True):
        tgt = [[torch.full_like(t, 0.5 if accumulate else float('nan')) for t in (ws[l], cs[l], gs[l], bs[l])] for l in range(L)]
        ops.unfold_context_grads(dwf, dbf, ws, gs, bs, [t[0] for t in tgt], [t[1] for t in tgt], [t[2] for t in tgt], [t[3] for t in tgt], accumulate)
        base = 0.5 if accumulate else 0.0
        for l in range(L):
            for got, ref in zip(tgt[l], prm[l]):
                torch.testing.assert_close(got.double() - base, ref.grad, rtol=1e-3, atol=1e-3 * float(ref.grad.abs().max()))
