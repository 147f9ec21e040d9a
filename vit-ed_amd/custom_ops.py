"""``torch.ops.vited.*``: the C-ABI kernels registered as PyTorch custom operators ("who
calls it": ``torch.library.custom_op`` + ``register_autograd`` + ``register_fake`` + an autocast rule -> ctypes
-> ``libvited_hip.so``).

These are the op-level building blocks of the reference's layers, usable from ordinary PyTorch code, under
``torch.autocast`` and under ``torch.compile`` tracing (fake kernels give shapes / dtypes without a GPU):

  ===========================  =====================================================  ==============================
  operator                     reference code it replaces                             C entry points
  ===========================  =====================================================  ==============================
  ``vited::layernorm``         ``nn.LayerNorm(eps=1e-6)`` vision_transformer.py:108   vited_layernorm_fwd / _bwd
  ``vited::linear``            ``nn.Linear`` (:33,36,151,152,154 and timm Mlp.fc2)     vited_gemm, vited_linear_bwd_weight
  ``vited::mlp``               timm ``Mlp`` fc1 -> GELU(erf) -> fc2 (:115,259)        vited_gemm (GELU / GELU' epilogues)
  ``vited::attention``         ``F.scaled_dot_product_attention`` (:60-65,181-186)    vited_attention_fwd / _bwd
  ``vited::patchify``          im2col of timm ``PatchEmbed``'s Conv2d(k=s=p) (:383)   vited_patchify
  ``vited::linear_residual_layernorm``  ``x = x + proj(...)`` followed by the next      vited_linear_residual_layernorm_fwd,
                               sub-block's ``norm(x)`` (:124-127, 268-272)             vited_linear_layernorm_bwd
  ===========================  =====================================================  ==============================

The model itself (``model.VisionTransformerCustom``) drives the same C entry points through two coarser
autograd Functions (``functions.EncoderFn / DecoderFn``) that additionally fuse residual adds into GEMM
epilogues and accumulate weight gradients in place; the operators here are the drop-in granularity for code that
wants to keep the reference's ``nn.Module`` structure.

CUDA(HIP)-only: there is no CPU kernel behind any of them (calling one with CPU tensors raises
``NotImplementedError`` from the dispatcher), by design - the product path never falls back.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op, register_autograd

from . import ops
from ._lib import B_NK, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_STORE  # noqa: F401

_LOWP = torch.bfloat16


def _rows(x: Tensor) -> Tensor:
    return x.reshape(-1, x.shape[-1])


def _act(x: Tensor) -> Tensor:
    """Dense 2-D activation operand in a dtype the kernels take (fp32 or bf16)."""
    x2 = _rows(x)
    if x2.dtype not in (torch.float32, _LOWP):
        x2 = x2.to(_LOWP)
    return x2 if x2.stride(-1) == 1 else x2.contiguous()


def _weight_for(x2: Tensor, w: Tensor) -> Tensor:
    return w.detach().reshape(w.shape[0], -1).to(x2.dtype).contiguous()


def _bias32(b: Optional[Tensor]) -> Optional[Tensor]:
    return None if b is None else b.detach().float().contiguous()


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
@custom_op('vited::layernorm', mutates_args=(), device_types='cuda')
def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float, lowp: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """y = LN(x) over the last dim; x fp32 [..., D].  Returns (y in bf16 if ``lowp`` else fp32, mean, rstd)."""
    x2 = _rows(x.float())
    y, mean, rstd = ops.layernorm_fwd(x2 if x2.stride(-1) == 1 else x2.contiguous(), weight.float().contiguous(),
                                      bias.float().contiguous(), eps, _LOWP if lowp else torch.float32)
    return y.view(x.shape), mean, rstd


@layernorm.register_fake
def _(x, weight, bias, eps, lowp):
    rows = x.numel() // x.shape[-1]
    return (x.new_empty(x.shape, dtype=_LOWP if lowp else torch.float32), x.new_empty(rows, dtype=torch.float32),
            x.new_empty(rows, dtype=torch.float32))


@custom_op('vited::layernorm_backward', mutates_args=(), device_types='cuda')
def layernorm_backward(dy: Tensor, x: Tensor, weight: Tensor, mean: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    x2 = _rows(x.float())
    x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
    dx, _, dgamma, dbeta = ops.layernorm_bwd(_act(dy), x2, weight.float().contiguous(), mean, rstd)
    return dx.view(x.shape), dgamma, dbeta


@layernorm_backward.register_fake
def _(dy, x, weight, mean, rstd):
    return (x.new_empty(x.shape, dtype=torch.float32), weight.new_empty(weight.shape, dtype=torch.float32),
            weight.new_empty(weight.shape, dtype=torch.float32))


def _ln_setup(ctx, inputs, output):
    x, weight, _bias, _eps, _lowp = inputs
    _y, mean, rstd = output
    ctx.save_for_backward(x, weight, mean, rstd)


def _ln_backward(ctx, dy, _dmean, _drstd):
    x, weight, mean, rstd = ctx.saved_tensors
    dx, dgamma, dbeta = torch.ops.vited.layernorm_backward(dy, x, weight, mean, rstd)
    return dx.to(x.dtype), dgamma.to(weight.dtype), dbeta.to(weight.dtype), None, None


register_autograd('vited::layernorm', _ln_backward, setup_context=_ln_setup)


# ---------------------------------------------------------------------------------------------
# Linear (+ GELU)
# ---------------------------------------------------------------------------------------------
@custom_op('vited::linear', mutates_args=(), device_types='cuda')
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """y = x W^T + b; x [..., K] fp32 or bf16 (the output takes x's dtype), W [N, K], b [N] | None."""
    x2 = _act(x)
    y = ops.gemm(x2, _weight_for(x2, weight), b_layout=B_NK, epilogue=EPI_STORE, bias=_bias32(bias))
    return y.view(*x.shape[:-1], weight.shape[0])


@linear.register_fake
def _(x, weight, bias):
    dt = x.dtype if x.dtype in (torch.float32, _LOWP) else _LOWP
    return x.new_empty((*x.shape[:-1], weight.shape[0]), dtype=dt)


@custom_op('vited::linear_backward', mutates_args=(), device_types='cuda')
def linear_backward(dy: Tensor, x: Tensor, weight: Tensor, has_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """(dx, dW fp32, db fp32 (empty when ``has_bias`` is False)) of vited::linear."""
    x2 = _act(x)
    dy2 = _act(dy).to(x2.dtype)
    wt = weight.detach().reshape(weight.shape[0], -1).t().to(x2.dtype).contiguous()   # [K, N]: NT operand of dX = dY W
    dx = ops.gemm(dy2, wt, b_layout=B_NK, epilogue=EPI_STORE)
    dw, db = ops.linear_bwd_weight(dy2, x2, want_bias=has_bias)
    if db is None:
        db = dw.new_empty(0)
    return dx.view(x.shape), dw.view(weight.shape), db


@linear_backward.register_fake
def _(dy, x, weight, has_bias):
    dt = x.dtype if x.dtype in (torch.float32, _LOWP) else _LOWP
    return (x.new_empty(x.shape, dtype=dt), weight.new_empty(weight.shape, dtype=torch.float32),
            weight.new_empty(weight.shape[0] if has_bias else 0, dtype=torch.float32))


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None
    ctx.bias_dtype = None if bias is None else bias.dtype


def _linear_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.vited.linear_backward(dy, x, weight, ctx.has_bias)
    return dx.to(x.dtype), dw.to(weight.dtype), (db.to(ctx.bias_dtype) if ctx.has_bias else None)


register_autograd('vited::linear', _linear_backward, setup_context=_linear_setup)


@custom_op('vited::mlp', mutates_args=(), device_types='cuda')
def mlp(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """timm ``Mlp``: y = fc2(gelu(fc1(x))) with the erf GELU fused into fc1's GEMM epilogue.
    Returns (y, z = fc1 pre-activation, h = gelu(z)); z and h are what the backward needs."""
    x2 = _act(x)
    z, h = ops.gemm(x2, _weight_for(x2, w1), b_layout=B_NK, epilogue=EPI_GELU, bias=_bias32(b1))
    y = ops.gemm(h, _weight_for(x2, w2), b_layout=B_NK, epilogue=EPI_STORE, bias=_bias32(b2))
    hidden = (*x.shape[:-1], w1.shape[0])
    return y.view(*x.shape[:-1], w2.shape[0]), z.view(hidden), h.view(hidden)


@mlp.register_fake
def _(x, w1, b1, w2, b2):
    dt = x.dtype if x.dtype in (torch.float32, _LOWP) else _LOWP
    hidden = (*x.shape[:-1], w1.shape[0])
    return x.new_empty((*x.shape[:-1], w2.shape[0]), dtype=dt), x.new_empty(hidden, dtype=dt), x.new_empty(hidden, dtype=dt)


@custom_op('vited::mlp_backward', mutates_args=(), device_types='cuda')
def mlp_backward(dy: Tensor, x: Tensor, z: Tensor, h: Tensor, w1: Tensor,
                 w2: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(dx, dW1, db1, dW2, db2): dz = (dy W2) * gelu'(z) comes out of ONE GEMM (GELU' in its epilogue)."""
    x2 = _act(x)
    dy2 = _act(dy).to(x2.dtype)
    z2, h2 = _act(z).to(x2.dtype), _act(h).to(x2.dtype)
    w2t = w2.detach().t().to(x2.dtype).contiguous()      # [hidden, D]: NT operand of dh = dy W2
    w1t = w1.detach().t().to(x2.dtype).contiguous()      # [D, hidden]: NT operand of dx = dz W1
    dz = ops.gemm(dy2, w2t, b_layout=B_NK, epilogue=EPI_MUL_GELU_GRAD, aux=z2)
    dw2, db2 = ops.linear_bwd_weight(dy2, h2)
    dx = ops.gemm(dz, w1t, b_layout=B_NK, epilogue=EPI_STORE)
    dw1, db1 = ops.linear_bwd_weight(dz, x2)
    return dx.view(x.shape), dw1, db1, dw2, db2


@mlp_backward.register_fake
def _(dy, x, z, h, w1, w2):
    dt = x.dtype if x.dtype in (torch.float32, _LOWP) else _LOWP
    f = torch.float32
    return (x.new_empty(x.shape, dtype=dt), w1.new_empty(w1.shape, dtype=f), w1.new_empty(w1.shape[0], dtype=f),
            w2.new_empty(w2.shape, dtype=f), w2.new_empty(w2.shape[0], dtype=f))


def _mlp_setup(ctx, inputs, output):
    x, w1, b1, w2, b2 = inputs
    _y, z, h = output
    ctx.save_for_backward(x, z, h, w1, w2)
    ctx.bias_dtypes = (b1.dtype, b2.dtype)


def _mlp_backward(ctx, dy, _dz, _dh):
    x, z, h, w1, w2 = ctx.saved_tensors
    dx, dw1, db1, dw2, db2 = torch.ops.vited.mlp_backward(dy, x, z, h, w1, w2)
    return dx.to(x.dtype), dw1.to(w1.dtype), db1.to(ctx.bias_dtypes[0]), dw2.to(w2.dtype), db2.to(ctx.bias_dtypes[1])


register_autograd('vited::mlp', _mlp_backward, setup_context=_mlp_setup)


# ---------------------------------------------------------------------------------------------
# residual Linear fused with the LayerNorm that follows it on the residual stream (gemm_row.hip)
# ---------------------------------------------------------------------------------------------
@custom_op('vited::linear_residual_layernorm', mutates_args=(), device_types='cuda')
def linear_residual_layernorm(a: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Tensor, gamma: Tensor, beta: Tensor,
                              eps: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """y = residual + a W^T + b (fp32) and h = LayerNorm(y; gamma, beta) (bf16) with its row statistics, in one kernel when the
    row-complete tile covers the shape (384 output columns), as two kernels otherwise.  a [..., K], residual [..., N]."""
    a2 = _act(a).to(_LOWP)
    r2 = _rows(residual.float())
    r2 = r2 if r2.stride(-1) == 1 else r2.contiguous()
    w = _weight_for(a2, weight)
    g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    if ops.linear_layernorm_supported(a2.shape[0], w.shape[0], a2.shape[1], a2.dtype):
        y, h, mean, rstd = ops.linear_residual_layernorm_fwd(a2, w, _bias32(bias), r2, g, b, eps)
    else:
        from ._lib import EPI_RESIDUAL
        y = ops.gemm(a2, w, b_layout=B_NK, epilogue=EPI_RESIDUAL, bias=_bias32(bias), residual=r2)
        h, mean, rstd = ops.layernorm_fwd(y, g, b, eps, _LOWP)
    return y.view(residual.shape), h.view(residual.shape), mean, rstd


@linear_residual_layernorm.register_fake
def _(a, weight, bias, residual, gamma, beta, eps):
    rows = residual.numel() // residual.shape[-1]
    return (residual.new_empty(residual.shape, dtype=torch.float32), residual.new_empty(residual.shape, dtype=_LOWP),
            residual.new_empty(rows, dtype=torch.float32), residual.new_empty(rows, dtype=torch.float32))


@custom_op('vited::linear_layernorm_backward', mutates_args=(), device_types='cuda')
def linear_layernorm_backward(dy: Tensor, weight: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor,
                              dx_in: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """Input gradient of  y = LayerNorm(x; gamma, beta) W^T + b  given dy:  dx = (dx_in or 0) + LN'(dy W), dgamma, dbeta - the
    input-gradient GEMM and the LayerNorm backward in one kernel (d LayerNorm-output never exists in memory)."""
    dy2 = _act(dy).to(_LOWP)
    x2 = _rows(x.float())
    x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
    wt = weight.detach().reshape(weight.shape[0], -1).t().to(_LOWP).contiguous()     # [K_in, N_out]: NT operand of dX = dY W
    g = gamma.detach().float().contiguous()
    din = None if dx_in is None else _rows(dx_in.float()).contiguous()
    if ops.linear_layernorm_supported(dy2.shape[0], wt.shape[0], dy2.shape[1], dy2.dtype):
        dx, _, dg, db = ops.linear_layernorm_bwd(dy2, wt, x2, g, mean, rstd, dx_in=din)
    else:
        dh = ops.gemm(dy2, wt, b_layout=B_NK, epilogue=EPI_STORE)
        dx, _, dg, db = ops.layernorm_bwd(dh, x2, g, mean, rstd, dx_in=din)
    return dx.view(x.shape), dg, db


@linear_layernorm_backward.register_fake
def _(dy, weight, x, gamma, mean, rstd, dx_in):
    return (x.new_empty(x.shape, dtype=torch.float32), gamma.new_empty(gamma.shape, dtype=torch.float32),
            gamma.new_empty(gamma.shape, dtype=torch.float32))


def _lrl_setup(ctx, inputs, output):
    a, weight, bias, residual, gamma, beta, _eps = inputs
    y, _h, mean, rstd = output
    ctx.save_for_backward(a, weight, y, gamma, mean, rstd)
    ctx.has_bias = bias is not None
    ctx.dtypes = (a.dtype, None if bias is None else bias.dtype, residual.dtype, beta.dtype)


def _lrl_backward(ctx, dy, dh, _dmean, _drstd):
    a, weight, y, gamma, mean, rstd = ctx.saved_tensors
    # d(y) = dy + LN'(dh); then the residual passes it through and the Linear splits it into da, dW, db
    dln, dgamma, dbeta = torch.ops.vited.layernorm_backward(dh, y, gamma, mean, rstd)
    dtot = dln if dy is None else dln + dy.float()
    da, dw, db = torch.ops.vited.linear_backward(dtot, a, weight, ctx.has_bias)
    return (da.to(ctx.dtypes[0]), dw.to(weight.dtype), db.to(ctx.dtypes[1]) if ctx.has_bias else None, dtot.to(ctx.dtypes[2]),
            dgamma.to(gamma.dtype), dbeta.to(ctx.dtypes[3]), None)


register_autograd('vited::linear_residual_layernorm', _lrl_backward, setup_context=_lrl_setup)


# ---------------------------------------------------------------------------------------------
# scaled-dot-product attention (no mask, no dropout: the only form the reference uses)
# ---------------------------------------------------------------------------------------------
def _tokens(t: Tensor) -> Tensor:
    t = t if t.dtype in (torch.float32, _LOWP) else t.to(_LOWP)
    return t if (t.dim() == 3 and t.stride(2) == 1) else t.contiguous()


@custom_op('vited::attention', mutates_args=(), device_types='cuda')
def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tuple[Tensor, Tensor]:
    """softmax(q k^T * scale) v per head.  q [B, Nq, H*hd], k / v [B, Nk, H*hd] (token-major, the layout of the
    packed qkv / kv projections - strided last-dim slices are read in place).  Returns (o [B, Nq, H*hd], lse)."""
    q, k, v = _tokens(q), _tokens(k).to(q.dtype), _tokens(v).to(q.dtype)
    return ops.attention_fwd(q, k, v, heads, scale)


@attention.register_fake
def _(q, k, v, heads, scale):
    dt = q.dtype if q.dtype in (torch.float32, _LOWP) else _LOWP
    return q.new_empty(q.shape, dtype=dt), q.new_empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32)


@custom_op('vited::attention_backward', mutates_args=(), device_types='cuda')
def attention_backward(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse: Tensor, heads: int,
                       scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    q, k, v = _tokens(q), _tokens(k).to(q.dtype), _tokens(v).to(q.dtype)
    dq, dk, dv = torch.empty(q.shape, dtype=q.dtype, device=q.device), torch.empty(k.shape, dtype=q.dtype, device=q.device), \
        torch.empty(v.shape, dtype=q.dtype, device=q.device)
    ops.attention_bwd(q, k, v, o.contiguous(), do.to(q.dtype).contiguous(), lse, heads, scale, dq, dk, dv)
    return dq, dk, dv


@attention_backward.register_fake
def _(q, k, v, o, do, lse, heads, scale):
    dt = q.dtype if q.dtype in (torch.float32, _LOWP) else _LOWP
    return q.new_empty(q.shape, dtype=dt), k.new_empty(k.shape, dtype=dt), v.new_empty(v.shape, dtype=dt)


def _attn_setup(ctx, inputs, output):
    q, k, v, heads, scale = inputs
    o, lse = output
    ctx.save_for_backward(q, k, v, o, lse)
    ctx.heads, ctx.scale = heads, scale


def _attn_backward(ctx, do, _dlse):
    q, k, v, o, lse = ctx.saved_tensors
    dq, dk, dv = torch.ops.vited.attention_backward(q, k, v, o, do, lse, ctx.heads, ctx.scale)
    return dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype), None, None


register_autograd('vited::attention', _attn_backward, setup_context=_attn_setup)


# ---------------------------------------------------------------------------------------------
# patch extraction (the image needs no gradient: misc/engine.py never asks for one)
# ---------------------------------------------------------------------------------------------
@custom_op('vited::patchify', mutates_args=(), device_types='cuda')
def patchify(img: Tensor, patch: int, lowp: bool) -> Tensor:
    """img fp32 [B, C, S, S] -> [B, (S/p)^2, C*p*p] rows in the Conv2d weight's (c, i, j) order."""
    b, c, s, _ = img.shape
    out = ops.patchify(img.float(), patch, _LOWP if lowp else torch.float32)
    return out.view(b, (s // patch) ** 2, c * patch * patch)


@patchify.register_fake
def _(img, patch, lowp):
    b, c, s, _ = img.shape
    return img.new_empty((b, (s // patch) ** 2, c * patch * patch), dtype=_LOWP if lowp else torch.float32)


# ---------------------------------------------------------------------------------------------
# autocast: inside torch.autocast('cuda', bfloat16) the GEMM / attention operators take bf16 activations,
# exactly as F.linear / SDPA do in the reference under torch.cuda.amp.autocast (misc/engine.py:208)
# ---------------------------------------------------------------------------------------------
for _name in ('vited::linear', 'vited::mlp', 'vited::attention'):
    torch.library.register_autocast(_name, 'cuda', _LOWP)

OPERATORS = ('layernorm', 'layernorm_backward', 'linear', 'linear_backward', 'mlp', 'mlp_backward', 'attention',
             'attention_backward', 'patchify', 'linear_residual_layernorm', 'linear_layernorm_backward')
