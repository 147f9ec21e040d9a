"""Seeded shape fuzzing of the C-ABI contraction, attention and LayerNorm entry points against fp64 PyTorch on the SAME
(bf16-rounded) operands: random row counts around every tile boundary, every column count the dispatchers accept, strided operand
views, all epilogues.  The fixed cases of test_gpu_ops.py pin the shapes of the shipped configs; this pins the dispatch logic
between them (which kernel variant a shape takes must never change the result)."""
import math
import os
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
MORE = int(os.environ.get('VITED_FUZZ_SEEDS', '0'))      # extra seeds for a one-off hunt: VITED_FUZZ_SEEDS=40 pytest tests/test_gpu_fuzz.py
BF16_OUT = dict(rtol=1e-2, atol=1e-2)


def _rand(shape, dev, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def _strided(rows, cols, dev, seed, dtype, rng, scale=1.0):
    """[rows, cols] view with a dense last dim: either contiguous or a column slice of a wider buffer (16-byte aligned)."""
    if rng.random() < 0.5:
        return _rand((rows, cols), dev, seed, scale, dtype)
    pad_l, pad_r = 8 * rng.randrange(0, 5), 8 * rng.randrange(0, 5)
    return _rand((rows, pad_l + cols + pad_r), dev, seed, scale, dtype)[:, pad_l: pad_l + cols]


@pytest.mark.parametrize('seed', range(6 + MORE))
def test_fuzz_gemm_and_weight_gradient(vited, gpu, seed):
    ops, L = vited.ops, vited._lib
    rng = random.Random(1000 + seed)
    for case in range(7):
        K = rng.choice([64, 128, 192, 384, 384, 768, 1152, 1536])
        N = rng.choice([16, 64, 128, 144, 384, 384, 768, 1152, 1536])
        M = rng.choice([1, 17, 127, 129, 255, 1000, 4095, 4096, 4097, 8191, 8193, 9999, 12345, 16511])
        M = max(1, M + rng.randrange(-3, 4))
        a = _strided(M, K, gpu, 7 * case + 1, torch.bfloat16, rng)
        w = _rand((N, K), gpu, 7 * case + 2, 1 / math.sqrt(K), torch.bfloat16)
        bias = _rand((N,), gpu, 7 * case + 3)
        ref = a.double() @ w.double().t()
        tag = f'seed {seed} case {case}: M={M} N={N} K={K} a.stride={a.stride()}'
        epi = rng.choice(['store', 'gelu', 'gelu_grad', 'residual', 'mul', 'mul_gelu_grad', 'f32'])
        if epi == 'store':
            torch.testing.assert_close(ops.gemm(a, w, bias=bias).double(), ref + bias.double(), msg=lambda m: f'{tag} store: {m}', **BF16_OUT)
        elif epi == 'gelu':
            z, u = ops.gemm(a, w, epilogue=L.EPI_GELU, bias=bias)
            torch.testing.assert_close(u.double(), F.gelu(ref + bias.double()), msg=lambda m: f'{tag} gelu: {m}', **BF16_OUT)
            torch.testing.assert_close(z.double(), ref + bias.double(), msg=lambda m: f'{tag} z: {m}', **BF16_OUT)
        elif epi == 'gelu_grad':
            zz = (ref + bias.double()).requires_grad_()
            F.gelu(zz).sum().backward()
            gd, h = ops.gemm(a, w, epilogue=L.EPI_GELU_GRAD, bias=bias)
            torch.testing.assert_close(h.double(), F.gelu(zz.detach()), msg=lambda m: f'{tag} h: {m}', **BF16_OUT)
            torch.testing.assert_close(gd.double(), zz.grad, msg=lambda m: f'{tag} gelu\': {m}', **BF16_OUT)
        elif epi == 'residual':
            res = _rand((M, N), gpu, 7 * case + 4)
            y = ops.gemm(a, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=res)
            torch.testing.assert_close(y.double(), ref + bias.double() + res.double(), rtol=3e-4, atol=3e-4, msg=lambda m: f'{tag} res: {m}')
        elif epi == 'mul':
            aux = _rand((M, N), gpu, 7 * case + 5, dtype=torch.bfloat16)
            torch.testing.assert_close(ops.gemm(a, w, epilogue=L.EPI_MUL, aux=aux).double(), ref * aux.double(), msg=lambda m: f'{tag} mul: {m}', **BF16_OUT)
        elif epi == 'mul_gelu_grad':
            aux = _rand((M, N), gpu, 7 * case + 5, dtype=torch.bfloat16)
            xg = aux.double().requires_grad_()
            F.gelu(xg).sum().backward()
            torch.testing.assert_close(ops.gemm(a, w, epilogue=L.EPI_MUL_GELU_GRAD, aux=aux).double(), ref * xg.grad, msg=lambda m: f'{tag} dz: {m}', **BF16_OUT)
        else:
            torch.testing.assert_close(ops.gemm(a, w, epilogue=L.EPI_STORE_F32).double(), ref, rtol=3e-4, atol=3e-4, msg=lambda m: f'{tag} f32: {m}')
        # weight gradient of the same geometry: dW[N, K] = dY^T X over the M rows (+ bias gradient)
        Nw = rng.choice([8, 72, 128, 200, 384, 768, 1152])
        dy = _strided(M, Nw, gpu, 7 * case + 6, torch.bfloat16, rng)
        dw, db = ops.linear_bwd_weight(dy, a)
        atol = 2e-4 * math.sqrt(M) + 1e-3
        torch.testing.assert_close(dw.double(), dy.double().t() @ a.double(), rtol=3e-4, atol=atol, msg=lambda m: f'{tag} dW N={Nw}: {m}')
        torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=3e-4, atol=atol, msg=lambda m: f'{tag} db N={Nw}: {m}')


def _sdpa(q, k, v, heads, scale):
    B, Nq, D = q.shape
    hd = D // heads
    qh, kh, vh = (t.double().view(B, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax((qh @ kh.transpose(-1, -2)) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, D)


@pytest.mark.parametrize('seed', range(4 + MORE))
def test_fuzz_attention(vited, gpu, seed):
    """Short (register-resident) and long (flash) sequences, self- and cross-shaped, packed strided q/k/v views, both head sizes."""
    ops = vited.ops
    rng = random.Random(2000 + seed)
    for case in range(5):
        hd = rng.choice([32, 64])
        heads = rng.choice([1, 3, 6, 12]) if hd == 32 else rng.choice([1, 2, 6])
        D = heads * hd
        if rng.random() < 0.6:
            nq, nk = rng.randrange(1, 81), rng.randrange(1, 81)
        else:
            nq, nk = rng.choice([81, 129, 257, 300, 513]), rng.choice([96, 128, 255, 321, 512])
        B = rng.choice([1, 2, 5, 9])
        packed = _rand((B, max(nq, nk), 3 * D), gpu, 11 * case + seed, 1.0, torch.bfloat16)
        q, k, v = packed[:, :nq, :D], packed[:, :nk, D:2 * D], packed[:, :nk, 2 * D:]
        scale = hd ** -0.5
        tag = f'seed {seed} case {case}: B={B} heads={heads} hd={hd} nq={nq} nk={nk}'
        qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
        ref = _sdpa(qd, kd, vd, heads, scale)
        o, lse = ops.attention_fwd(q, k, v, heads, scale)
        torch.testing.assert_close(o.double(), ref.detach(), msg=lambda m: f'{tag} o: {m}', **BF16_OUT)
        do = _rand((B, nq, D), gpu, 11 * case + seed + 5, 1.0, torch.bfloat16)
        ref.backward(do.double())
        dpacked = torch.zeros_like(packed)
        dq, dk, dv = dpacked[:, :nq, :D], dpacked[:, :nk, D:2 * D], dpacked[:, :nk, 2 * D:]
        ops.attention_bwd(q, k, v, o, do, lse, heads, scale, dq, dk, dv)
        for name, got, want in (('dq', dq, qd.grad), ('dk', dk, kd.grad), ('dv', dv, vd.grad)):
            torch.testing.assert_close(got.double(), want, rtol=3e-2, atol=3e-2, msg=lambda m: f'{tag} {name}: {m}')


@pytest.mark.parametrize('seed', range(3 + MORE))
def test_fuzz_layernorm(vited, gpu, seed):
    ops = vited.ops
    rng = random.Random(3000 + seed)
    for case in range(6):
        dim = rng.choice([4, 32, 128, 384, 384, 768, 1000, 1024])
        rows = rng.choice([1, 2, 31, 33, 255, 1000, 4097, 20000]) + rng.randrange(0, 3)
        x = _strided(rows, dim, gpu, 5 * case + 1, torch.float32, rng, scale=2.0) + 0.5
        g, b = _rand((dim,), gpu, 5 * case + 2) + 1.0, _rand((dim,), gpu, 5 * case + 3)
        tag = f'seed {seed} case {case}: rows={rows} dim={dim} x.stride={x.stride()}'
        xd, gd_, bd = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
        ref = F.layer_norm(xd, (dim,), gd_, bd, 1e-6)
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, torch.float32)
        torch.testing.assert_close(y.double(), ref.detach(), rtol=2e-4, atol=2e-4, msg=lambda m: f'{tag} y: {m}')
        dy = _rand((rows, dim), gpu, 5 * case + 4)
        ref.backward(dy.double())
        dx_in = _rand((rows, dim), gpu, 5 * case + 5)
        dx, dx_lp, dg, db = ops.layernorm_bwd(dy, x, g, mean, rstd, dx_in=dx_in, want_lp=True)
        torch.testing.assert_close(dx.double(), xd.grad + dx_in.double(), rtol=5e-4, atol=5e-4, msg=lambda m: f'{tag} dx: {m}')
        torch.testing.assert_close(dx_lp.double(), xd.grad + dx_in.double(), msg=lambda m: f'{tag} dx_lp: {m}', **BF16_OUT)
        atol = 1e-4 * math.sqrt(rows) + 1e-4
        torch.testing.assert_close(dg.double(), gd_.grad, rtol=5e-4, atol=atol, msg=lambda m: f'{tag} dgamma: {m}')
        torch.testing.assert_close(db.double(), bd.grad, rtol=5e-4, atol=atol, msg=lambda m: f'{tag} dbeta: {m}')
