"""torch.ops.vited.*: registration, fake (shape/dtype) kernels and the no-CPU-fallback rule on CPU; numerics,
autograd, autocast, opcheck and torch.compile tracing on the GPU (SURVEY.md section 8(b) "who calls it")."""
import pytest
import torch
import torch.nn.functional as F


def test_operators_are_registered_with_fake_kernels(vited):
    names = vited.custom_ops.OPERATORS
    for n in names:
        assert hasattr(torch.ops.vited, n), n
    m = 'meta'
    x = torch.empty(3, 65, 384, device=m)
    g = torch.empty(384, device=m)
    y, mean, rstd = torch.ops.vited.layernorm(x, g, g, 1e-6, True)
    assert y.shape == x.shape and y.dtype == torch.bfloat16 and mean.shape == rstd.shape == (195,)
    w, b = torch.empty(1152, 384, device=m), torch.empty(1152, device=m)
    assert torch.ops.vited.linear(x, w, b).shape == (3, 65, 1152)
    assert torch.ops.vited.linear(x.bfloat16(), w, None).dtype == torch.bfloat16
    dx, dw, db = torch.ops.vited.linear_backward(torch.empty(3, 65, 1152, device=m), x, w, True)
    assert dx.shape == x.shape and dw.shape == w.shape and dw.dtype == torch.float32 and db.shape == (1152,)
    q, kv = torch.empty(3, 65, 384, device=m, dtype=torch.bfloat16), torch.empty(3, 64, 384, device=m, dtype=torch.bfloat16)
    o, lse = torch.ops.vited.attention(q, kv, kv, 12, 32 ** -0.5)
    assert o.shape == q.shape and lse.shape == (3, 12, 65) and lse.dtype == torch.float32
    dq, dk, dv = torch.ops.vited.attention_backward(q, kv, kv, o, o, lse, 12, 32 ** -0.5)
    assert dq.shape == q.shape and dk.shape == dv.shape == kv.shape
    w1, b1, w2, b2 = (torch.empty(1536, 384, device=m), torch.empty(1536, device=m), torch.empty(384, 1536, device=m),
                      torch.empty(384, device=m))
    y, z, h = torch.ops.vited.mlp(x, w1, b1, w2, b2)
    assert y.shape == x.shape and z.shape == h.shape == (3, 65, 1536)
    outs = torch.ops.vited.mlp_backward(y, x, z, h, w1, w2)
    assert [tuple(t.shape) for t in outs] == [(3, 65, 384), (1536, 384), (1536,), (384, 1536), (384,)]
    assert torch.ops.vited.patchify(torch.empty(2, 3, 64, 64, device=m), 8, True).shape == (2, 64, 192)
    wp = torch.empty(384, 384, device=m)
    y, h, mean, rstd = torch.ops.vited.linear_residual_layernorm(x.bfloat16(), wp, g, x, g, g, 1e-6)
    assert y.shape == h.shape == x.shape and y.dtype == torch.float32 and h.dtype == torch.bfloat16 and mean.shape == (195,)
    dx, dg, db = torch.ops.vited.linear_layernorm_backward(torch.empty(3, 65, 1152, device=m), w, x, g, mean, rstd, None)
    assert dx.shape == x.shape and dg.shape == db.shape == (384,)


def test_operators_have_no_cpu_kernel(vited):
    with pytest.raises(NotImplementedError):
        torch.ops.vited.linear(torch.zeros(2, 8), torch.zeros(4, 8), None)
    with pytest.raises(NotImplementedError):
        torch.ops.vited.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-6, False)


# ------------------------------------------------------------------------------------------------
def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _grads(fn, tensors, seed=5):
    ts = [t.detach().clone().requires_grad_(True) for t in tensors]
    out = fn(*ts)
    w = _rand(out.shape, out.device, seed).to(out.dtype)
    (out.float() * w.float()).sum().backward()
    return out.detach(), [t.grad for t in ts]


def _close(a, b, tol):
    a, b = a.float(), b.float()
    err = (a - b).norm() / b.norm().clamp_min(1e-12)
    assert err < tol, f'relative error {err:.3e} >= {tol}'


@pytest.mark.gpu
@pytest.mark.parametrize('lowp', [False, True])
def test_custom_ops_match_torch_reference(vited, gpu, lowp):
    """fp32: rtol 1e-4 of the PyTorch fp32 op (global relative error); bf16: 2e-2 (bf16 operands, fp32 accumulate)."""
    tol = 2e-2 if lowp else 1e-4
    dt = torch.bfloat16 if lowp else torch.float32
    B, N, D, H = 6, 65, 384, 12
    x = _rand((B, N, D), gpu, 1)
    gamma, beta = 1 + 0.1 * _rand((D,), gpu, 2), 0.1 * _rand((D,), gpu, 3)
    # LayerNorm
    o, g = _grads(lambda x, a, b: torch.ops.vited.layernorm(x, a, b, 1e-6, lowp)[0], [x, gamma, beta])
    o_ref, g_ref = _grads(lambda x, a, b: F.layer_norm(x, (D,), a, b, 1e-6), [x, gamma, beta])
    _close(o, o_ref, tol)
    for a, b in zip(g, g_ref):
        _close(a, b, tol)
    # Linear
    w, bias = _rand((3 * D, D), gpu, 4, 0.05), _rand((3 * D,), gpu, 5, 0.1)
    o, g = _grads(lambda x, w, b: torch.ops.vited.linear(x.to(dt), w, b), [x, w, bias])
    o_ref, g_ref = _grads(lambda x, w, b: F.linear(x, w, b), [x, w, bias])
    _close(o, o_ref, tol)
    for a, b in zip(g, g_ref):
        assert a.dtype == torch.float32
        _close(a, b, tol)
    # Mlp
    w1, b1, w2, b2 = _rand((4 * D, D), gpu, 6, 0.05), _rand((4 * D,), gpu, 7, 0.1), _rand((D, 4 * D), gpu, 8, 0.03), _rand((D,), gpu, 9, 0.1)
    o, g = _grads(lambda x, *p: torch.ops.vited.mlp(x.to(dt), *p)[0], [x, w1, b1, w2, b2])
    o_ref, g_ref = _grads(lambda x, w1, b1, w2, b2: F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2), [x, w1, b1, w2, b2])
    _close(o, o_ref, tol)
    for a, b in zip(g, g_ref):
        _close(a, b, 2 * tol)
    # attention on strided slices of a packed qkv projection (self) and q + packed kv (cross)
    qkv = _rand((B, N, 3 * D), gpu, 10, 0.5)
    scale = (D // H) ** -0.5

    def sdpa_ref(qkv):
        q, k, v = (t.reshape(B, N, H, D // H).transpose(1, 2) for t in qkv.split(D, dim=-1))
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)

    def sdpa_ours(qkv):
        t = qkv.to(dt)
        return torch.ops.vited.attention(t[..., :D], t[..., D:2 * D], t[..., 2 * D:], H, scale)[0]

    o, g = _grads(sdpa_ours, [qkv])
    o_ref, g_ref = _grads(sdpa_ref, [qkv])
    _close(o, o_ref, tol)
    _close(g[0], g_ref[0], 2 * tol)
    # residual Linear + the LayerNorm behind it (one kernel), forward and autograd, against x + F.linear -> F.layer_norm
    if lowp:
        wp, bp = _rand((D, D), gpu, 20, 0.05), _rand((D,), gpu, 21, 0.1)
        res = _rand((B, N, D), gpu, 22, 2.0)
        wy, wh = _rand((B, N, D), gpu, 23), _rand((B, N, D), gpu, 24)

        def fused(a, w, b, r, gm, bt):
            y, h, _, _ = torch.ops.vited.linear_residual_layernorm(a.to(dt), w, b, r, gm, bt, 1e-6)
            return y * wy + h.float() * wh

        def plain(a, w, b, r, gm, bt):
            y = r + F.linear(a, w, b)
            return y * wy + F.layer_norm(y, (D,), gm, bt, 1e-6) * wh

        o, g = _grads(fused, [x, wp, bp, res, gamma, beta])
        o_ref, g_ref = _grads(plain, [x, wp, bp, res, gamma, beta])
        _close(o, o_ref, tol)
        for a, b in zip(g, g_ref):
            _close(a, b, 2 * tol)
        # ... and the backward-only pairing: dX GEMM + LayerNorm backward in one kernel
        xin = _rand((B, N, D), gpu, 25, 1.5)
        mu, var = xin.mean(-1), xin.var(-1, unbiased=False)
        dyq = _rand((B, N, 3 * D), gpu, 26)
        dx, dg_, db_ = torch.ops.vited.linear_layernorm_backward(dyq, w, xin, gamma, mu.reshape(-1), (var + 1e-6).rsqrt().reshape(-1), None)
        xr, gr, br = xin.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        F.linear(F.layer_norm(xr, (D,), gr, br, 1e-6), w).backward(dyq)
        _close(dx, xr.grad, tol)
        _close(dg_, gr.grad, tol)
        _close(db_, br.grad, tol)
    # patchify + linear == Conv2d(k = s = p)
    img = _rand((B, 3, 64, 64), gpu, 11)
    cw, cb = _rand((D, 3, 8, 8), gpu, 12, 0.05), _rand((D,), gpu, 13, 0.1)
    tok = torch.ops.vited.linear(torch.ops.vited.patchify(img, 8, lowp), cw.reshape(D, -1), cb)
    _close(tok, F.conv2d(img, cw, cb, stride=8).flatten(2).transpose(1, 2), tol)


@pytest.mark.gpu
def test_custom_ops_autocast_opcheck_and_compile(vited, gpu):
    D, H = 384, 12
    x = _rand((4, 64, D), gpu, 1).requires_grad_(True)
    w, b = _rand((D, D), gpu, 2, 0.05).requires_grad_(True), _rand((D,), gpu, 3, 0.1).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = torch.ops.vited.linear(x, w, b)          # autocast rule: bf16 activations, like F.linear in the reference
    assert y.dtype == torch.bfloat16
    y.float().sum().backward()
    assert x.grad.dtype == w.grad.dtype == torch.float32 and w.grad.shape == w.shape
    # schema / fake-kernel / autograd-registration consistency checks of torch.library
    torch.library.opcheck(torch.ops.vited.linear.default, (x.detach(), w.detach(), b.detach()),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))
    g = torch.ones(D, device=gpu)
    torch.library.opcheck(torch.ops.vited.layernorm.default, (x.detach(), g, g, 1e-6, True),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))
    q = x.detach().bfloat16()
    torch.library.opcheck(torch.ops.vited.attention.default, (q, q, q, H, 0.17),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))

    # a reference-style pre-LN attention branch written with the operators traces under torch.compile
    def branch(x, g, w_qkv, b_qkv, w, b):
        n = torch.ops.vited.layernorm(x, g, g * 0, 1e-6, True)[0]
        qkv = torch.ops.vited.linear(n, w_qkv, b_qkv)
        o = torch.ops.vited.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], H, (D // H) ** -0.5)[0]
        return x + torch.ops.vited.linear(o, w, b).float()

    w_qkv, b_qkv = _rand((3 * D, D), gpu, 4, 0.05), _rand((3 * D,), gpu, 5, 0.1)
    eager = branch(x.detach(), g, w_qkv, b_qkv, w.detach(), b.detach())
    compiled = torch.compile(branch, backend='aot_eager', fullgraph=True)(x.detach(), g, w_qkv, b_qkv, w.detach(), b.detach())
    assert torch.equal(eager, compiled)
