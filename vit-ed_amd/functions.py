"""Encoder / decoder autograd Functions of the ViT-ED hot path, composed from the HIP ops.

Two coarse Functions instead of ~230 per-op autograd nodes: the whole encoder
(models/vision_transformer.py:382-388 ``forward_first_part``) and the whole decoder + head
(:390-405 ``prepare_x2`` / ``cross_part`` / ``forward_second_part`` and timm ``forward_head``).
Inside each, forward and backward are explicit sequences of kernel launches with hand-managed
saved activations, which is what lets the residual adds, the low-precision copies of the residual
gradient and the accumulation of d(context) over the c_depth decoder blocks live in kernel
epilogues instead of separate elementwise passes.

Data types: the residual stream, LayerNorm statistics, parameters and parameter gradients are
fp32; activations that feed contractions are ``rt.act_dtype`` (bf16 on the MFMA path, fp32 on the
exact path).
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._lib import (B_KN, B_NK, EPI_GELU_GRAD, EPI_MUL, EPI_RESIDUAL, EPI_STORE, EPI_STORE_F32)

LN_EPS = 1e-6  # partial(nn.LayerNorm, eps=1e-6), vision_transformer.py:348

ENC_BLOCK_KEYS = ('norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight',
                  'attn.proj.bias', 'norm2.weight', 'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias',
                  'mlp.fc2.weight', 'mlp.fc2.bias')
DEC_BLOCK_KEYS = ('norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight',
                  'attn.proj.bias', 'norm_cross.weight', 'norm_cross.bias', 'norm_context.weight',
                  'norm_context.bias', 'cross_attn.q.weight', 'cross_attn.q.bias', 'cross_attn.kv.weight',
                  'cross_attn.kv.bias', 'cross_attn.proj.weight', 'cross_attn.proj.bias', 'norm2.weight',
                  'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias')
ENC_SHARED_KEYS = ('patch_embed.proj.weight', 'patch_embed.proj.bias', 'pos_embed')
DEC_SHARED_KEYS = ('patch_embed.proj.weight', 'patch_embed.proj.bias', 'pos_embed', 'cls_token', 'norm.weight',
                   'norm.bias', 'head.weight', 'head.bias')


_step_hook_installed = False


def _install_optimizer_step_hook():
    """The bf16 weight shadows are validated by the parameters' version counters.  ``torch.optim.*(fused=True)`` updates the
    parameters WITHOUT bumping them (measured: ``p._version`` is unchanged by a fused AdamW step, while the default / foreach
    implementations bump it), so a model trained by such an optimizer in the reference's own loop would keep running on its
    initial weights.  A global optimizer post-step hook bumps the versions of whatever any optimizer just stepped; optimizers
    that refresh the shadows themselves (optim.FlatAdamW) opt out with ``manages_weight_shadows``."""
    global _step_hook_installed
    if _step_hook_installed:
        return
    from torch.optim.optimizer import register_optimizer_step_post_hook

    def bump(optimizer, args, kwargs):
        if getattr(optimizer, 'manages_weight_shadows', False):
            return
        params = [p for g in optimizer.param_groups for p in g['params'] if torch.is_tensor(p)]
        if params:
            torch.autograd.graph.increment_version(params)

    register_optimizer_step_post_hook(bump)
    _step_hook_installed = True


class Runtime:
    """Per-model launch context: static shape, activation dtype and the low-precision weight shadows."""

    def __init__(self, *, img_size, patch_size, in_chans, num_classes, embed_dim, depth, c_depth, num_heads,
                 act_dtype=torch.bfloat16):
        self.img_size, self.patch_size, self.in_chans = img_size, patch_size, in_chans
        self.num_classes, self.dim, self.depth, self.c_depth, self.heads = num_classes, embed_dim, depth, c_depth, num_heads
        self.n1 = (img_size // patch_size) ** 2
        self.n2 = self.n1 + 1
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.act_dtype = act_dtype
        self.direct_grads = False   # accumulate parameter gradients straight into existing p.grad (engine.FlatGradients)
        self.input_mean, self.input_std = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)   # Normalize() of data/transforms.py:14-18, for uint8 inputs
        self.keep_attn = False      # MODEL.PJS.KEEP_ATTN: also materialise the attention maps (visualisation slow path)
        self.attn_store = {}        # (kind, block index, 'attn' | 'cross_attn') -> {'attn': ..., 'grad': ...}
        self.cls_tail = os.environ.get('VITED_CLS_TAIL', '1') != '0'   # last decoder block on the cls rows only (exact; see _dec_block_fwd)
        self.fused_mlp = os.environ.get('VITED_FUSED_MLP', '1') != '0'   # vited_mlp_fwd on the no-grad paths
        self.fused_ln = os.environ.get('VITED_FUSED_LN', '1') != '0'     # LayerNorm inside the neighbouring Linear's kernel (gemm_row.hip)
        self.batch_dw = os.environ.get('VITED_BATCH_DW', '1') != '0'     # a block's weight gradients in one launch (vited_linear_bwd_weight_batched)
        self.group_dw = os.environ.get('VITED_GROUP_DW', '1') != '0'     # ... and several blocks' together while they fit one round of workgroups
        self.dw_queue = None        # inside a block's backward: [(dy, x, dW target, dbias target | None, accumulate)]
        self.ln_queue = None        # inside a Function's backward: deferred LayerNorm column sums (ops.layernorm_bwd_finish)
        self.fold_context = os.environ.get('VITED_FOLD_CONTEXT', '1') != '0'   # norm_context + kv of all decoder blocks as one GEMM
        self._fold_bufs = None      # (folded W [L 2D, D], its transpose, folded bias): refreshed in place every forward
        self._unit_ln = None        # (ones, zeros) for the affine-free LayerNorm of the features
        self.tap = None             # test/diagnostic: a dict that receives clones of per-block activations and gradients
        self.block_events = None    # measurement (bench.py): a list that receives (kind, block, 'fwd' | 'bwd', start event, end event)
        self.pinned = False         # a captured hipGraph reads the shadow buffers: never free one, only refresh in place
        self._retired = []
        self._shadow = {}
        if not self.exact:
            _install_optimizer_step_hook()

    @property
    def exact(self):
        return self.act_dtype == torch.float32

    def weight(self, w: torch.Tensor) -> torch.Tensor:
        """[N, K] operand of the forward GEMM in the activation dtype (cached per parameter version)."""
        w2 = w.detach().reshape(w.shape[0], -1)
        if self.exact:
            return w2
        return self._cached(w, 'n', lambda out=None: ops.cast(w2, self.act_dtype, out=out))

    def weight_t(self, w: torch.Tensor):
        """Operand + layout of the input-gradient GEMM dX = dY . W: the transposed bf16 shadow (NT
        MFMA kernel) or, on the exact path, W itself read as [K, N]."""
        w2 = w.detach().reshape(w.shape[0], -1)
        if self.exact:
            return w2, B_KN
        return self._cached(w, 't', lambda out=None: ops.cast_transpose(w2.contiguous(), self.act_dtype, out=out)), B_NK

    def _cached(self, w, tag, make):
        key = (id(w), tag)
        ent = self._shadow.get(key)
        if ent is None or ent[0] != w._version or ent[1] != w.data_ptr():
            if ent is not None and ent[1] == w.data_ptr():
                # refresh IN PLACE: a captured graph keeps reading the same shadow buffer
                ent = (w._version, w.data_ptr(), make(ent[2]))
            else:
                if ent is not None and self.pinned:
                    self._retired.append(ent[2])     # the graph still reads the old buffer: keep it alive
                ent = (w._version, w.data_ptr(), make())
            self._shadow[key] = ent
        return ent[2]

    def refresh_shadows(self, params):
        """Recast every cached shadow in place with ONE multi-tensor launch (call after an optimizer step when
        replaying a graph: the captured kernels keep reading the same shadow buffers)."""
        by_id = {id(p): p for p in params}
        per_param = {}
        for (pid, tag), ent in self._shadow.items():
            if pid in by_id:
                per_param.setdefault(pid, {})[tag] = ent[2]
        entries = []
        for pid, tags in per_param.items():
            p = by_id[pid]
            entries.append((p.detach().reshape(p.shape[0], -1), tags.get('n'), tags.get('t')))
        if not entries:
            return
        key = tuple((w.data_ptr(), 0 if n is None else n.data_ptr(), 0 if t is None else t.data_ptr()) for w, n, t in entries)
        plan = getattr(self, '_shadow_plan', None)
        if plan is None or plan.key != key:
            plan = self._shadow_plan = ops.WeightShadowPlan(entries)
        plan.run()
        for pid, tags in per_param.items():
            p = by_id[pid]
            for tag, buf in tags.items():
                self._shadow[(pid, tag)] = (p._version, p.data_ptr(), buf)


class _BlockSpan:
    """Brackets one block's launches with HIP events on the launch stream when ``rt.block_events`` is a list (bench.py's
    ``fused_block`` figure: time of the attention + MLP block's kernels); free otherwise."""

    def __init__(self, rt, kind, index, phase):
        self.sink, self.tag = rt.block_events, (kind, index, phase)

    def __enter__(self):
        if self.sink is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.sink.append(self.tag + (self.e0, e1))
        return False


def _lp(rt: Runtime, t_f32: torch.Tensor) -> torch.Tensor:
    return t_f32 if rt.exact else ops.cast(t_f32, rt.act_dtype)


# ---------------------------------------------------------------------------------------------
# shared pieces
# ---------------------------------------------------------------------------------------------
def _gtarget(rt, p):
    """p.grad when gradients may be accumulated straight into it (saves autograd's AccumulateGrad add and
    a temporary per parameter): only when the caller pre-attached dense fp32 .grad tensors."""
    if p is None or not rt.direct_grads:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or g.device != p.device:
        return None
    return g


def _ln_bwd(rt, dy, x, gamma, beta, mean, rstd, **kw):
    """LayerNorm backward; returns (dx, dx_lp, dgamma | None, dbeta | None) - None when accumulated in place."""
    gg, gb = _gtarget(rt, gamma), _gtarget(rt, beta)
    if gg is not None and gb is not None:
        dx, dx_lp, _, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=gg, dbeta=gb, **kw)
        return dx, dx_lp, None, None
    return ops.layernorm_bwd(dy, x, gamma, mean, rstd, **kw)


def _weight_grads(rt, dy, x_saved, w, b):
    """(dW | None, db | None) of y = x W^T + b; None when accumulated straight into w.grad / b.grad.  Inside a _DwBatch (one
    transformer block's backward) the product is only QUEUED: the block's weight gradients then go out as one launch."""
    gw = _gtarget(rt, w)
    gb = _gtarget(rt, b) if b is not None else None
    direct = gw is not None and (b is None or gb is not None)
    q = rt.dw_queue
    if q is not None and dy.dtype == torch.bfloat16 and dy.shape[0] >= 4096 and x_saved.shape[1] % 384 == 0:
        if direct:
            q.append((dy, x_saved, gw.view(gw.shape[0], -1), gb, True))
            return None, None
        dw = torch.empty((dy.shape[1], x_saved.shape[1]), dtype=torch.float32, device=dy.device)
        db = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device) if b is not None else None
        q.append((dy, x_saved, dw, db, False))
        return dw.view_as(w), db
    if direct:
        ops.linear_bwd_weight(dy, x_saved, want_bias=b is not None, dw_out=gw.view(gw.shape[0], -1), db_out=gb)
        return None, None
    dw, db = ops.linear_bwd_weight(dy, x_saved)
    return dw.view_as(w), (db if b is not None else None)


class _DwBatch:
    """``with _DwBatch(rt):`` around one block's backward: the weight-gradient products issued inside are collected and
    launched together (vited_linear_bwd_weight_batched) - they only read tensors the block's backward already produced, and
    nothing inside a backward pass consumes a weight gradient.

    ``with _DwBatch(rt, blocks=True) as g:`` around a LOOP over blocks, ``with g.block():`` around each: the products of several
    blocks go out together, flushed when another block of the same size would no longer fit one round of 256 workgroups
    (an encoder block of the embed-384 models is 36 output tiles of 128 x 384: seven blocks = 252 tiles = ONE row range per
    product - no split-M slabs to write and sum - where one block alone is cut into 7 row ranges)."""

    ROUND = 256

    def __init__(self, rt, blocks=False, kind=None):
        self.rt, self.blocks, self.kind = rt, blocks and rt.group_dw, kind

    def __enter__(self):
        self.outer = self.rt.dw_queue
        self.rt.dw_queue = [] if (self.rt.batch_dw and not self.rt.exact) else None
        self.mark = 0
        return self

    def block(self):
        return _DwBlock(self)

    @staticmethod
    def _tiles(items):
        return sum(-(-dy.shape[1] // 128) * (x.shape[1] // 384) for dy, x, _dw, _db, _a in items)

    def _end_of_block(self):
        q = self.rt.dw_queue
        if q is None:
            return
        if not self.blocks:
            self.flush()
            return
        this, count = self._tiles(q[self.mark:]), len(q) - self.mark      # the block that just ended: the next one is taken to be alike
        self.mark = len(q)
        if self._tiles(q) + this > self.ROUND or len(q) + count > ops.MAX_BATCHED_WEIGHT_GRADS:
            self.flush()

    def flush(self):
        q = self.rt.dw_queue
        if not q:
            return
        self.rt.dw_queue, self.mark = [], 0
        with _BlockSpan(self.rt, self.kind, len(q), 'dw'):     # (bench.py: these launches belong to the blocks queued since the last flush)
            self._launch(q)

    @staticmethod
    def _launch(q):
        for acc in (True, False):
            group = [(dy, x, dw, db) for dy, x, dw, db, a in q if a == acc]
            for i in range(0, len(group), ops.MAX_BATCHED_WEIGHT_GRADS):
                part = group[i: i + ops.MAX_BATCHED_WEIGHT_GRADS]
                if len(part) > 1 and ops.linear_bwd_weight_batched(part, acc):
                    continue
                for dy, x, dw, db in part:
                    ops.linear_bwd_weight(dy, x, want_bias=db is not None, dw_out=dw, db_out=db) if acc else \
                        _overwrite_weight_grad(dy, x, dw, db)

    def __exit__(self, exc_type, *exc):
        if exc_type is None:
            self.flush()
        self.rt.dw_queue = self.outer
        return False


class _DwBlock:
    def __init__(self, group):
        self.group = group

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        if exc_type is None:
            self.group._end_of_block()
        return False


def _overwrite_weight_grad(dy, x, dw, db):
    got_w, got_b = ops.linear_bwd_weight(dy, x, want_bias=db is not None)
    dw.copy_(got_w)
    if db is not None:
        db.copy_(got_b)


def _linear_bwd(rt, dy, x_saved, w, b=None, want_dx=True, aux=None):
    """(dx | None, dW, db) of y = x W^T + b given dy (activation dtype)."""
    dx = None
    if want_dx:
        wt, layout = rt.weight_t(w)
        if aux is not None:
            dx = ops.gemm(dy, wt, b_layout=layout, epilogue=EPI_MUL, aux=aux)
        else:
            dx = ops.gemm(dy, wt, b_layout=layout)
    dw, db = _weight_grads(rt, dy, x_saved, w, b)
    return dx, dw, db


def _row_kernel_ok(rt, m, n, k, dtype, *rowwise):
    """The row-complete Linear + LayerNorm kernels (gemm_row.hip) take this product: bf16, 384 output columns, dense rows."""
    return (rt.fused_ln and not rt.exact and ops.linear_layernorm_supported(m, n, k, dtype)
            and all(t is None or (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0) for t in rowwise))


def _res_linear(rt, a, w, bias, residual, ln=None):
    """y = residual + a W^T + bias (fp32) and, with ``ln`` = (gamma, beta) of the LayerNorm that FOLLOWS on the residual stream
    (the next sub-block's norm: vision_transformer.py:124-127, 268-272), also (h, mean, rstd) = LayerNorm(y) - in ONE kernel when
    the row-complete kernel covers the shape, otherwise as vited_gemm(RESIDUAL) + vited_layernorm_fwd.
    Returns (y, (h, mean, rstd) | None)."""
    wsh = rt.weight(w)
    if ln is not None and _row_kernel_ok(rt, a.shape[0], wsh.shape[0], a.shape[1], a.dtype, a, residual):
        y, h, mean, rstd = ops.linear_residual_layernorm_fwd(a, wsh, bias, residual, ln[0], ln[1], LN_EPS)
        return y, (h, mean, rstd)
    y = ops.gemm(a, wsh, epilogue=EPI_RESIDUAL, bias=bias, residual=residual)
    return y, (ops.layernorm_fwd(y, ln[0], ln[1], LN_EPS, rt.act_dtype) if ln is not None else None)


def _linear_ln_bwd(rt, dy, h_saved, w, bias, x, gamma, beta, mean, rstd, dx_in=None, dx_out=None, want_lp=True):
    """Backward of  y = LayerNorm(x; gamma, beta) W^T + bias  given dy: the input-gradient GEMM and the LayerNorm backward in
    ONE kernel when the row-complete kernel covers the shape (d(LayerNorm output) then never exists in HBM).
    Returns (dx fp32 = dx_in + ..., dx_lp | None, dgamma, dbeta, dW, dbias) - gradients are None when accumulated in place."""
    want_lp = want_lp and not rt.exact
    wt, layout = rt.weight_t(w)
    if layout == B_NK and _row_kernel_ok(rt, dy.shape[0], wt.shape[0], dy.shape[1], dy.dtype, dy, x, dx_in, dx_out):
        gg, gb = _gtarget(rt, gamma), _gtarget(rt, beta)
        direct = gg is not None and gb is not None
        dx, dx_lp, dg, db = ops.linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dx_in=dx_in, dx_out=dx_out, want_lp=want_lp,
                                                     dgamma=gg if direct else None, dbeta=gb if direct else None, defer=rt.ln_queue)
        if direct:
            dg = db = None
    else:
        dh = ops.gemm(dy, wt, b_layout=layout)
        dx, dx_lp, dg, db = _ln_bwd(rt, dh, x, gamma, beta, mean, rstd, dx_in=dx_in, dx_out=dx_out, want_lp=want_lp)
    dw, dbias = _weight_grads(rt, dy, h_saved, w, bias)
    return dx, dx_lp, dg, db, dw, dbias


def _keep_attention(rt, key, q, k):
    """KEEP_ATTN slow path (vision_transformer.py:67-75,188-195; consumer: scripts/visualise_attentions.py): materialise
    softmax(q k^T * scale) [B, h, Nq, Nk] with plain PyTorch ops, beside the fused kernels that never form it."""
    b, nq, d = q.shape
    qh = q.float().reshape(b, nq, rt.heads, rt.head_dim).transpose(1, 2)
    kh = k.float().reshape(b, k.shape[1], rt.heads, rt.head_dim).transpose(1, 2)
    rt.attn_store.setdefault(key, {})['attn'] = torch.softmax((qh * rt.scale) @ kh.transpose(-2, -1), dim=-1)


def _keep_attention_grad(rt, key, do, v):
    """What the reference's ``attn.register_hook(self.save_attn_gradients)`` records: d loss / d attn = dO V^T."""
    b, nq, d = do.shape
    doh = do.float().reshape(b, nq, rt.heads, rt.head_dim).transpose(1, 2)
    vh = v.float().reshape(b, v.shape[1], rt.heads, rt.head_dim).transpose(1, 2)
    rt.attn_store.setdefault(key, {})['grad'] = doh @ vh.transpose(-2, -1)


def _self_attn_fwd(rt, h, wqkv, bqkv, batch, n, key=None):
    d = rt.dim
    qkv = ops.gemm(h, rt.weight(wqkv), bias=bqkv)                   # [M, 3D], columns [3][h][hd] (:58)
    qkv3 = qkv.view(batch, n, 3 * d)
    o, lse = ops.attention_fwd(qkv3[:, :, 0:d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:3 * d], rt.heads, rt.scale)
    if rt.keep_attn and key is not None:
        _keep_attention(rt, key, qkv3[:, :, 0:d], qkv3[:, :, d:2 * d])
    return qkv, o.view(batch * n, d), lse


def _self_attn_bwd(rt, do, qkv, o, lse, batch, n, key=None):
    d = rt.dim
    qkv3 = qkv.view(batch, n, 3 * d)
    if rt.keep_attn and key is not None:
        _keep_attention_grad(rt, key, do.view(batch, n, d), qkv3[:, :, 2 * d:3 * d])
    dqkv = torch.empty_like(qkv)
    dqkv3 = dqkv.view(batch, n, 3 * d)
    ops.attention_bwd(qkv3[:, :, 0:d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:3 * d], o.view(batch, n, d),
                      do.view(batch, n, d), lse, rt.heads, rt.scale, dqkv3[:, :, 0:d], dqkv3[:, :, d:2 * d],
                      dqkv3[:, :, 2 * d:3 * d])
    return dqkv


FUSED_MLP_TILE = 128        # token rows per workgroup of vited_mlp_fwd (one workgroup per CU)
FUSED_MLP_CUS = 256


def _fused_mlp_rows(rt, x, w1, grad):
    """How many leading rows the fused MLP kernel (vited_mlp_fwd) takes: it runs one 128-row workgroup per CU, so a last
    round that fills less than a quarter of the chip is left to the unfused kernels (66,560 rows = 520 tiles = 2 rounds + 8
    tiles: the 8 tiles would cost a third round).  0 = do not use it.  Measured (profiles/mlp_probe.py, M = 65,536): it beats
    LayerNorm + fc1/GELU + fc2/residual when nothing is saved for backward (269 vs 344 us) and loses when the backward's
    operands must be written (376 us; in the training step: +0.9 ms), so it serves the no-grad paths (evaluation,
    similarity-matrix inference)."""
    if grad or rt.exact or not rt.fused_mlp or x.shape[1] != 384 or tuple(w1.shape) != (1536, 384) or x.stride(0) != 384:
        return 0
    tiles = x.shape[0] // FUSED_MLP_TILE
    rem = tiles % FUSED_MLP_CUS
    if tiles >= FUSED_MLP_CUS and rem < FUSED_MLP_CUS // 4:
        tiles -= rem
    elif x.shape[0] % FUSED_MLP_TILE:
        tiles += 1                      # ragged last tile: the kernel clamps rows
    return min(tiles * FUSED_MLP_TILE, x.shape[0])


def _mlp_fwd(rt, x, g, b, w1, b1, w2, b2, grad=True, ln=None, next_ln=None):
    """x + fc2(gelu(fc1(LayerNorm(x)))).  ``ln`` = (h, mean, rstd) when the LayerNorm was already produced by the kernel that
    wrote x; ``next_ln`` = (gamma, beta) of the LayerNorm that follows on the output.  Returns (y, saved | None, next | None)."""
    rows = _fused_mlp_rows(rt, x, w1, grad) if ln is None else 0
    if rows:
        y = torch.empty_like(x)
        ops.mlp_fwd(x[:rows], g, b, rt.weight(w1), b1, rt.weight(w2), b2, LN_EPS, save=False, out=(y[:rows], None, None, None, None, None))
        if rows < x.shape[0]:
            xt = x[rows:]
            ht, _, _ = ops.layernorm_fwd(xt, g, b, LN_EPS, rt.act_dtype)
            _, ut = ops.gemm(ht, rt.weight(w1), epilogue=EPI_GELU_GRAD, bias=b1)
            ops.gemm(ut, rt.weight(w2), epilogue=EPI_RESIDUAL, bias=b2, residual=xt, out=y[rows:])
        nxt = ops.layernorm_fwd(y, next_ln[0], next_ln[1], LN_EPS, rt.act_dtype) if next_ln is not None else None
        return y, None, nxt
    h, mean, rstd = ln if ln is not None else ops.layernorm_fwd(x, g, b, LN_EPS, rt.act_dtype)
    # fc1 saves gelu'(z) and gelu(z) (one exponential serves both): the backward of the activation is then one multiply
    gd, u = ops.gemm(h, rt.weight(w1), epilogue=EPI_GELU_GRAD, bias=b1)
    y, nxt = _res_linear(rt, u, w2, b2, x, next_ln)
    return y, (mean, rstd, h, gd, u), nxt


def _mlp_bwd(rt, dy, dy_lp, x, g, b, w1, b1, w2, b2, saved):
    mean, rstd, h, gd, u = saved
    dz, dw2, db2 = _linear_bwd(rt, dy_lp, u, w2, b2, aux=gd)
    dx, dx_lp, dg, db, dw1, db1 = _linear_ln_bwd(rt, dz, h, w1, b1, x, g, b, mean, rstd, dx_in=dy)
    return dx, (dx if rt.exact else dx_lp), (dg, db, dw1, db1, dw2, db2)


def _attn_branch_fwd(rt, x, g, b, wqkv, bqkv, wproj, bproj, batch, n, key=None, ln=None, next_ln=None):
    """x + proj(attention(qkv(LayerNorm(x)))); ``ln`` / ``next_ln`` as in _mlp_fwd.  Returns (y, saved, next | None)."""
    h, mean, rstd = ln if ln is not None else ops.layernorm_fwd(x, g, b, LN_EPS, rt.act_dtype)
    qkv, o, lse = _self_attn_fwd(rt, h, wqkv, bqkv, batch, n, key)
    y, nxt = _res_linear(rt, o, wproj, bproj, x, next_ln)
    return y, (mean, rstd, h, qkv, o, lse), nxt


def _attn_branch_bwd(rt, dy, dy_lp, x, g, b, wqkv, bqkv, wproj, bproj, saved, batch, n, key=None):
    mean, rstd, h, qkv, o, lse = saved
    do, dwp, dbp = _linear_bwd(rt, dy_lp, o, wproj, bproj)
    dqkv = _self_attn_bwd(rt, do, qkv, o, lse, batch, n, key)
    dx, dx_lp, dg, db, dwq, dbq = _linear_ln_bwd(rt, dqkv, h, wqkv, bqkv, x, g, b, mean, rstd, dx_in=dy)
    return dx, (dx if rt.exact else dx_lp), (dg, db, dwq, dbq, dwp, dbp)


def _patch_tokens_fwd(rt, img, pw, pb, pos, with_cls, cls=None, batch_index=None):
    """timm PatchEmbed + pos-embed (+ cls row): returns x fp32 [B*rows, D] and the saved patch matrix."""
    patches = ops.patchify(img, rt.patch_size, rt.act_dtype, batch_index, mean=rt.input_mean, std=rt.input_std)
    batch = patches.shape[0] // rt.n1
    pos2 = pos.view(rt.n2, rt.dim)
    rows = rt.n2 if with_cls else rt.n1
    if with_cls:
        x = ops.gemm(patches, rt.weight(pw), epilogue=EPI_RESIDUAL, bias=pb, residual=pos2, rows_per_batch=rt.n1,
                     out_rows_per_batch=rt.n2, row_offset=1, residual_bcast=True, out_rows=batch * rt.n2)
        ops.write_cls_row(x.view(batch, rt.n2, rt.dim), cls.view(-1), pos2)
    else:
        x = ops.gemm(patches, rt.weight(pw), epilogue=EPI_RESIDUAL, bias=pb, residual=pos2[1:], rows_per_batch=rt.n1,
                     out_rows_per_batch=rt.n1, row_offset=0, residual_bcast=True, out_rows=batch * rt.n1)
    return x, patches, batch, rows


def _patch_tokens_bwd(rt, dx, patches, pw, pb, pos, with_cls, batch):
    """dx fp32 [B*rows, D] -> (d patch weight, d patch bias, d pos_embed, d cls | None)."""
    rows = rt.n2 if with_cls else rt.n1
    dx3 = dx.view(batch, rows, rt.dim)
    dpos_rows = ops.sum_rows(dx3.view(batch, rows * rt.dim)).view(rows, rt.dim)
    dpos = torch.zeros_like(pos)
    dcls = None
    if with_cls:
        dpos[0] = dpos_rows
        dcls = dpos_rows[0].clone().view(1, 1, rt.dim)
        dtok = ops.slice_rows_cast(dx3, 1, rt.n1, rt.act_dtype)
    else:
        dpos[0, 1:] = dpos_rows
        dtok = dx if rt.exact else ops.cast(dx, rt.act_dtype)
    dw, db = _weight_grads(rt, dtok, patches, pw, pb)
    return dw, db, dpos, dcls


# ---------------------------------------------------------------------------------------------
# encoder: forward_first_part (vision_transformer.py:382-388)
# ---------------------------------------------------------------------------------------------
class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt: Runtime, img, *params):
        pw, pb, pos = params[:3]
        nb = len(ENC_BLOCK_KEYS)
        blocks = [params[3 + i * nb: 3 + (i + 1) * nb] for i in range(rt.depth)]
        grad = any(ctx.needs_input_grad)  # False under no_grad: nothing is saved for inference
        x, patches, batch, n = _patch_tokens_fwd(rt, img, pw, pb, pos, with_cls=False)
        tape = []
        ln1 = None          # (h, mean, rstd) of this block's norm1 when the previous block's fc2 kernel already produced it
        for i, P in enumerate(blocks):
            g1, b1, wqkv, bqkv, wproj, bproj, g2, b2, w1, bb1, w2, bb2 = P
            # on the no-grad path the one-kernel MLP does its own LayerNorm: nothing to hand over
            chain = grad or not _fused_mlp_rows(rt, x, w1, grad)
            with _BlockSpan(rt, 'enc', i, 'fwd'):
                xa, sa, ln2 = _attn_branch_fwd(rt, x, g1, b1, wqkv, bqkv, wproj, bproj, batch, n, key=('blocks', i, 'attn'), ln=ln1,
                                               next_ln=(g2, b2) if chain else None)
                nxt = (blocks[i + 1][0], blocks[i + 1][1]) if (chain and i + 1 < rt.depth) else None
                xb, sm, ln1 = _mlp_fwd(rt, xa, g2, b2, w1, bb1, w2, bb2, grad, ln=ln2, next_ln=nxt)
            if grad:
                tape.append((x, sa, xa, sm))
            x = xb
            if rt.tap is not None:
                rt.tap[f'enc.x.{len(tape) - 1 if grad else 0}'] = x.clone()
        if grad:
            ctx.rt, ctx.tape, ctx.patches, ctx.batch, ctx.params = rt, tape, patches, batch, params
        return x.view(batch, n, rt.dim)

    @staticmethod
    def backward(ctx, dout):
        rt, batch, n = ctx.rt, ctx.batch, ctx.rt.n1
        params = ctx.params
        pw, pb, pos = params[:3]
        nb = len(ENC_BLOCK_KEYS)
        dx = dout.contiguous().view(batch * n, rt.dim).float()
        dx_lp = _lp(rt, dx)
        grads = [None] * len(params)
        rt.ln_queue = [] if not rt.exact else None
        with _DwBatch(rt, blocks=True, kind='enc') as dwg:
            for i in reversed(range(rt.depth)):
                g1, b1, wqkv, bqkv, wproj, bproj, g2, b2, w1, bb1, w2, bb2 = params[3 + i * nb: 3 + (i + 1) * nb]
                x, sa, xa, sm = ctx.tape[i]
                ctx.tape[i] = None
                with _BlockSpan(rt, 'enc', i, 'bwd'), dwg.block():
                    dx, dx_lp, (dg2, db2, dw1, dbb1, dw2, dbb2) = _mlp_bwd(rt, dx, dx_lp, xa, g2, b2, w1, bb1, w2, bb2, sm)
                    dx, dx_lp, (dg1, db1, dwq, dbq, dwp, dbp) = _attn_branch_bwd(rt, dx, dx_lp, x, g1, b1, wqkv, bqkv, wproj, bproj, sa, batch, n,
                                                                                 key=('blocks', i, 'attn'))
                if rt.tap is not None:
                    rt.tap[f'enc.dx.{i}'] = dx.clone()      # gradient w.r.t. the INPUT of encoder block i
                base = 3 + i * nb
                blk = [dg1, db1, dwq, dbq, dwp, dbp, dg2, db2, dw1, dbb1, dw2, dbb2]
                grads[base: base + nb] = blk
        if rt.ln_queue is not None:
            ops.layernorm_bwd_finish(rt.ln_queue)       # the encoder's 2 x depth LayerNorm column sums: one launch per 16
            rt.ln_queue = None
        dpw, dpb, dpos, _ = _patch_tokens_bwd(rt, dx, ctx.patches, pw, pb, pos, with_cls=False, batch=batch)
        grads[0], grads[1], grads[2] = dpw, dpb, dpos
        ctx.tape = ctx.patches = None
        return (None, None, *grads)


# ---------------------------------------------------------------------------------------------
# pair-cached decoder for similarity-matrix inference (hisfrag.py:213-231)
# ---------------------------------------------------------------------------------------------
# hisfrag.py:226-229 calls model(x1[idx1], x2[idx2]) per pair batch, so the reference (and DecoderFn) re-embeds image 2 and
# re-runs norm_context + the kv projection of every decoder block for EVERY pair.  Neither depends on the pair:
#   * prepare_x2 (vision_transformer.py:390-395) depends on image j only  -> image2_tokens(), once per image;
#   * cross-attention keys / values (:177-179) depend on image i's features only -> context_kv(), once per image-1 row block;
# the pair batch then gathers token rows by index j and the attention kernel reads K / V by index i (vited_attention_fwd_indexed).
@torch.no_grad()
def image2_tokens(rt: Runtime, img, pw, pb, pos, cls, block0=None):
    """Everything of the decoder that depends on image 2 ALONE, once per image:
      * x  = patch embedding + cls row + pos_embed (timm _pos_embed), the decoder's input stream;
      * x' = x + attn(norm1(x)) of the FIRST CrossBlock (its self-attention sees image 2 only: the features enter at the
             cross-attention that follows, vision_transformer.py:269-270) - unless that block is also the last one (cls-only);
      * q  = Linear_q(norm_cross(x')) of that block's cross-attention (:176), activation dtype.
    Returns (x' or x as [n, N2, D] fp32, q [n, N2, D] or None)."""
    x, _, batch, _ = _patch_tokens_fwd(rt, img, pw, pb, pos, with_cls=True, cls=cls)
    if block0 is None or (rt.cls_tail and rt.c_depth == 1):
        return x.view(batch, rt.n2, rt.dim), None
    g1, b1, wqkv, bqkv, wproj, bproj, gc, bc, gx, bx, wq, bq = block0[:12]
    xa, _, lnq = _attn_branch_fwd(rt, x, g1, b1, wqkv, bqkv, wproj, bproj, batch, rt.n2, next_ln=(gc, bc))
    q = ops.gemm(lnq[0], rt.weight(wq), bias=bq)
    return xa.view(batch, rt.n2, rt.dim), q.view(batch, rt.n2, rt.dim)


@torch.no_grad()
def context_kv(rt: Runtime, feats, blocks):
    """Per decoder block: kv = Linear_kv(norm_context(features)) as [b1, N1, 2 D] in the activation dtype (columns [2][h][hd])."""
    b1 = feats.shape[0]
    ctxf = feats.detach().contiguous().float().view(b1 * rt.n1, rt.dim)
    out = []
    for P in blocks:
        gx, bx, wkv, bkv = P[8], P[9], P[12], P[13]
        hc, _, _ = ops.layernorm_fwd(ctxf, gx, bx, LN_EPS, rt.act_dtype)
        out.append(ops.gemm(hc, rt.weight(wkv), bias=bkv).view(b1, rt.n1, 2 * rt.dim))
    return out


@torch.no_grad()
def decoder_cached(rt: Runtime, tokens2, j_idx, kvs, i_idx, params, q0=None):
    """Logits [P, C] of the pairs (image-1 row i_idx[p] of the cached block, image j_idx[p]): forward_second_part + forward_head
    (vision_transformer.py:397-405,417) on cached image-2 tokens and cached cross-attention keys / values.  With ``q0`` the cache
    already holds block 0's self-attention branch and cross-attention queries (image2_tokens with block0)."""
    gN, bN, wh, bh = params[4:8]
    ns, nb = 8, len(DEC_BLOCK_KEYS)
    d, n = rt.dim, rt.n2
    batch = j_idx.numel()
    x = tokens2.index_select(0, j_idx).view(batch * n, d)
    for l in range(rt.c_depth):
        g1, b1, wqkv, bqkv, wproj, bproj, gc, bc, gx, bx, wq, bq, wkv, bkv, wcp, bcp, g2, b2, w1, bb1, w2, bb2 = params[ns + l * nb: ns + (l + 1) * nb]
        cls_only = rt.cls_tail and l == rt.c_depth - 1          # see _dec_block_fwd: the last block runs on the cls row alone
        lnq = None
        if l == 0 and q0 is not None:
            xa, nq = x, n                                         # cached: x IS x + attn(norm1(x)) of block 0
        elif not cls_only:
            xa, _, lnq = _attn_branch_fwd(rt, x, g1, b1, wqkv, bqkv, wproj, bproj, batch, n, next_ln=(gc, bc))
            nq = n
        else:
            h1, _, _ = ops.layernorm_fwd(x, g1, b1, LN_EPS, rt.act_dtype)
            qkv3 = ops.gemm(h1, rt.weight(wqkv), bias=bqkv).view(batch, n, 3 * d)
            o0, _ = ops.attention_fwd(qkv3[:, 0:1, 0:d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:3 * d], rt.heads, rt.scale)
            xa = ops.gemm(o0.view(batch, d), rt.weight(wproj), epilogue=EPI_RESIDUAL, bias=bproj, residual=_dense_rows(x.view(batch, n, d)[:, 0, :]))
            nq = 1
        if l == 0 and q0 is not None:
            q = q0.index_select(0, j_idx)
        else:
            hq = lnq[0] if lnq is not None else ops.layernorm_fwd(xa, gc, bc, LN_EPS, rt.act_dtype)[0]
            q = ops.gemm(hq, rt.weight(wq), bias=bq)
        kv3 = kvs[l]
        oc, _ = ops.attention_fwd(q.view(batch, nq, d), kv3[:, :, 0:d], kv3[:, :, d:2 * d], rt.heads, rt.scale, kv_index=i_idx)
        xb = ops.gemm(oc.view(batch * nq, d), rt.weight(wcp), epilogue=EPI_RESIDUAL, bias=bcp, residual=xa)
        x, _, _ = _mlp_fwd(rt, xb, g2, b2, w1, bb1, w2, bb2, grad=False)
    xcls = x if (rt.cls_tail and rt.c_depth > 0) else x.view(batch, n, d)[:, 0, :]
    y, _, _ = ops.layernorm_fwd(xcls, gN, bN, LN_EPS, rt.act_dtype)
    return ops.gemm(y, rt.weight(wh), epilogue=EPI_STORE_F32, bias=bh)


# ---------------------------------------------------------------------------------------------
# decoder + head: forward_second_part + forward_head (vision_transformer.py:390-405,417)
# ---------------------------------------------------------------------------------------------
def _dense_rows(t):
    """[rows, D] copy with row stride D (``.contiguous()`` keeps the strides of a one-row view, and the GEMM epilogue needs ldo)."""
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def _dec_block_fwd(rt, x, ctxf, P, batch, n, grad, cls_only, index=0, ln1=None, next_ln=None, kv3=None):
    """One CrossBlock forward (vision_transformer.py:268-272).  ``cls_only`` (the LAST decoder block): only x[:, 0] of the
    block's output reaches the head (:400, :417 - the final norm and the head are row-wise), and within a CrossBlock the
    token rows only mix in the self-attention, as keys / values.  So after the block's qkv projection everything runs on the
    cls row alone: self-attention for query 0, proj, the whole cross-attention query side and the MLP - 1 row instead of
    N2 = 65 / 1025 per pair - with identical logits and identical gradients (the dropped rows' outputs are dead, their
    gradients exactly zero).  ``ln1`` = (h, mean, rstd) of this block's norm1 when the previous block's fc2 kernel produced it;
    ``next_ln`` = (gamma, beta) of the NEXT block's norm1.  Returns (block output, tape entry, next block's ln1 | None)."""
    g1, b1, wqkv, bqkv, wproj, bproj, gc, bc, gx, bx, wq, bq, wkv, bkv, wcp, bcp, g2, b2, w1, bb1, w2, bb2 = P
    d = rt.dim
    if not cls_only:
        xa, sa, lnq = _attn_branch_fwd(rt, x, g1, b1, wqkv, bqkv, wproj, bproj, batch, n, key=('cross_blocks', index, 'attn'), ln=ln1,
                                       next_ln=(gc, bc))
        nq = n
    else:
        h1, m1, r1 = ln1 if ln1 is not None else ops.layernorm_fwd(x, g1, b1, LN_EPS, rt.act_dtype)
        qkv = ops.gemm(h1, rt.weight(wqkv), bias=bqkv)            # K and V of every row feed query 0
        qkv3 = qkv.view(batch, n, 3 * d)
        o0, lse0 = ops.attention_fwd(qkv3[:, 0:1, 0:d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:3 * d], rt.heads, rt.scale)
        o0 = o0.view(batch, d)
        x0 = _dense_rows(x.view(batch, n, d)[:, 0, :])
        xa, lnq = _res_linear(rt, o0, wproj, bproj, x0, (gc, bc))
        sa = (m1, r1, h1, qkv, o0, lse0)
        nq = 1
    # cross attention: q from image-2 tokens, k/v from image-1 features (:174-200)
    hq, mq, rq = lnq
    q = ops.gemm(hq, rt.weight(wq), bias=bq)
    if kv3 is None:
        hc, mc, rc = ops.layernorm_fwd(ctxf, gx, bx, LN_EPS, rt.act_dtype)
        kv = ops.gemm(hc, rt.weight(wkv), bias=bkv)                  # [Mc, 2D], columns [2][h][hd] (:178)
        kv3 = kv.view(batch, rt.n1, 2 * d)
    else:
        hc = mc = rc = None     # this block's keys / values came out of the folded all-blocks GEMM (_context_kv_folded): a strided view
        kv = kv3
    oc, lse_c = ops.attention_fwd(q.view(batch, nq, d), kv3[:, :, 0:d], kv3[:, :, d:2 * d], rt.heads, rt.scale)
    if rt.keep_attn:
        _keep_attention(rt, ('cross_blocks', index, 'cross_attn'), q.view(batch, nq, d), kv3[:, :, 0:d])
    oc = oc.view(batch * nq, d)
    chain = grad or not _fused_mlp_rows(rt, xa, w1, grad)      # the no-grad one-kernel MLP does its own LayerNorm
    xb, ln2 = _res_linear(rt, oc, wcp, bcp, xa, (g2, b2) if chain else None)
    xc, sm, nxt = _mlp_fwd(rt, xb, g2, b2, w1, bb1, w2, bb2, grad, ln=ln2, next_ln=next_ln)
    entry = (x, sa, xa, (mq, rq, hq, mc, rc, hc, q, kv, oc, lse_c), xb, sm) if grad else None
    return xc, entry, nxt


def _dec_block_bwd(rt, dx, dx_lp, ctxf, dctx, P, entry, batch, n, cls_only, tap_index, dkv3=None):
    """Backward of _dec_block_fwd.  dx / dx_lp: gradient w.r.t. the block's output (all rows, or the cls rows when cls_only).
    Returns (d input fp32, its low-precision copy, d context (accumulated in place), the 22 parameter gradients)."""
    g1, b1, wqkv, bqkv, wproj, bproj, gc, bc, gx, bx, wq, bq, wkv, bkv, wcp, bcp, g2, b2, w1, bb1, w2, bb2 = P
    d = rt.dim
    x, sa, xa, sc, xb, sm = entry
    mq, rq, hq, mc, rc, hc, q, kv, oc, lse_c = sc
    nq = 1 if cls_only else n
    dx, dx_lp, (dg2, db2, dw1, dbb1, dw2, dbb2) = _mlp_bwd(rt, dx, dx_lp, xb, g2, b2, w1, bb1, w2, bb2, sm)
    # cross attention
    doc, dwcp, dbcp = _linear_bwd(rt, dx_lp, oc, wcp, bcp)
    dq = torch.empty_like(q)
    folded = hc is None
    if folded:
        kv3, dkv = kv, dkv3                 # views of the all-blocks kv / d(kv) tensors
    else:
        dkv = torch.empty_like(kv)
        kv3, dkv3 = kv.view(batch, rt.n1, 2 * d), dkv.view(batch, rt.n1, 2 * d)
    if rt.keep_attn:
        _keep_attention_grad(rt, ('cross_blocks', tap_index, 'cross_attn'), doc.view(batch, nq, d), kv3[:, :, d:2 * d])
    ops.attention_bwd(q.view(batch, nq, d), kv3[:, :, 0:d], kv3[:, :, d:2 * d], oc.view(batch, nq, d),
                      doc.view(batch, nq, d), lse_c, rt.heads, rt.scale, dq.view(batch, nq, d), dkv3[:, :, 0:d],
                      dkv3[:, :, d:2 * d])
    if rt.tap is not None:
        i = tap_index
        rt.tap[f'dec.doc.{i}'], rt.tap[f'dec.dq.{i}'], rt.tap[f'dec.dkv.{i}'] = doc.clone(), dq.clone(), dkv.clone()
        rt.tap[f'dec.q.{i}'], rt.tap[f'dec.kv.{i}'], rt.tap[f'dec.oc.{i}'] = q.clone(), kv.clone(), oc.clone()
    # q = Linear(norm_cross(x')), kv = Linear(norm_context(features)): input-gradient GEMM + LayerNorm backward fused
    dx, dx_lp, dgc, dbc, dwq, dbq = _linear_ln_bwd(rt, dq, hq, wq, bq, xa, gc, bc, mq, rq, dx_in=dx)
    if rt.exact:
        dx_lp = dx
    if folded:
        dgx = dbx = dwkv = dbkv = None      # filled in by _context_kv_folded_bwd once every block has written its d(kv)
    else:
        # d(context) accumulates over the c_depth blocks in fp32, in place
        dctx, _, dgx, dbx, dwkv, dbkv = _linear_ln_bwd(rt, dkv, hc, wkv, bkv, ctxf, gx, bx, mc, rc, dx_in=dctx, dx_out=dctx, want_lp=False)
    if not cls_only:
        dx, dx_lp, (dg1, db1, dwqkv, dbqkv, dwp, dbp) = _attn_branch_bwd(rt, dx, dx_lp, x, g1, b1, wqkv, bqkv, wproj, bproj, sa, batch, n,
                                                                           key=('cross_blocks', tap_index, 'attn'))
    else:
        m1, r1, h1, qkv, o0, lse0 = sa
        do0, dwp, dbp = _linear_bwd(rt, dx_lp, o0, wproj, bproj)
        qkv3 = qkv.view(batch, n, 3 * d)
        dqkv = torch.zeros_like(qkv)                 # d(q) of the rows that never queried is zero
        dqkv3 = dqkv.view(batch, n, 3 * d)
        ops.attention_bwd(qkv3[:, 0:1, 0:d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:3 * d], o0.view(batch, 1, d), do0.view(batch, 1, d),
                          lse0, rt.heads, rt.scale, dqkv3[:, 0:1, 0:d], dqkv3[:, :, d:2 * d], dqkv3[:, :, 2 * d:3 * d])
        dres = torch.zeros((batch * n, d), dtype=torch.float32, device=dx.device)   # the residual path carries gradient on the cls rows only
        dres.view(batch, n, d)[:, 0, :].copy_(dx)
        dx, dx_lp, dg1, db1, dwqkv, dbqkv = _linear_ln_bwd(rt, dqkv, h1, wqkv, bqkv, x, g1, b1, m1, r1, dx_in=dres)
        if rt.exact:
            dx_lp = dx
    return dx, dx_lp, dctx, [dg1, db1, dwqkv, dbqkv, dwp, dbp, dgc, dbc, dgx, dbx, dwq, dbq, dwkv, dbkv, dwcp, dbcp,
                             dg2, db2, dw1, dbb1, dw2, dbb2]


def _context_kv_folded(rt, ctxf, blocks):
    """Keys / values of EVERY decoder block from one LayerNorm and one GEMM (csrc/context_fold.hip): xhat = LayerNorm(features; 1, 0),
    kv_all = xhat W'^T + b' with W'_l = W_l o gamma_l, b'_l = b_l + W_l beta_l stacked over the blocks.
    Returns (kv_all [L, Mc, 2D], (xhat, mean, rstd), folded buffers)."""
    dev = ctxf.device
    if rt._unit_ln is None or rt._unit_ln[0].device != dev:
        rt._unit_ln = (torch.ones(rt.dim, dtype=torch.float32, device=dev), torch.zeros(rt.dim, dtype=torch.float32, device=dev))
    ws, bs, gs, bes = [P[12] for P in blocks], [P[13] for P in blocks], [P[8] for P in blocks], [P[9] for P in blocks]
    bufs = rt._fold_bufs
    if bufs is not None and (bufs[0].shape[0] != len(blocks) * 2 * rt.dim or bufs[0].device != dev):
        if rt.pinned:
            rt._retired.append(bufs)        # a captured graph still reads them
        bufs = None
    rt._fold_bufs = bufs = ops.fold_context_weights([w.detach() for w in ws], [b.detach() if b is not None else None for b in bs],
                                                    [g.detach() for g in gs], [b.detach() for b in bes], out=bufs)
    xhat, mean, rstd = ops.layernorm_fwd(ctxf, rt._unit_ln[0], rt._unit_ln[1], LN_EPS, rt.act_dtype)
    # one [Mc, 2D] tensor per block (each block's attention then reads dense rows; ONE [Mc, L 2D] product would also push the stacked
    # weights - 4.7 MB at 8 blocks - out of an XCD's L2: measured 690 us for the single GEMM against 8 x 55 us)
    n2 = 2 * rt.dim
    kv_all = torch.empty((len(blocks), ctxf.shape[0], n2), dtype=rt.act_dtype, device=dev)
    for l in range(len(blocks)):
        ops.gemm(xhat, bufs[0][l * n2:(l + 1) * n2], bias=bufs[2][l * n2:(l + 1) * n2], out=kv_all[l])
    return kv_all, (xhat, mean, rstd), bufs


def _context_kv_folded_bwd(rt, dkv_all, ctxf, saved, bufs, blocks):
    """Backward of _context_kv_folded once every block has written its d(kv) slice: d(features) from ONE row-complete kernel
    (input-gradient GEMM with K = L 2D + the affine-free LayerNorm's backward), the folded weights' gradient from one
    weight-gradient GEMM, unfolded into dW_kv, db_kv, d(norm_context.weight / bias) of every block.
    Returns (d features fp32, [(dgx, dbx, dwkv, dbkv) per block] - None entries when accumulated straight into .grad)."""
    xhat, mean, rstd = saved
    ones = rt._unit_ln[0]
    nblk, mc, n2 = dkv_all.shape
    if _row_kernel_ok(rt, mc, bufs[1].shape[0], nblk * n2, dkv_all.dtype, ctxf):
        dctx, _, _, _ = ops.linear_layernorm_bwd(dkv_all, bufs[1], ctxf, ones, mean, rstd)      # contraction over all blocks' d(kv)
    else:
        dh = torch.zeros((mc, rt.dim), dtype=torch.float32, device=ctxf.device)
        for l in range(nblk):
            dh = ops.gemm(dkv_all[l], bufs[1][:, l * n2:(l + 1) * n2], epilogue=EPI_RESIDUAL, residual=dh)
        dctx, _, _, _ = ops.layernorm_bwd(dh, ctxf, ones, mean, rstd)
    dwf = torch.empty((nblk * n2, rt.dim), dtype=torch.float32, device=ctxf.device)
    dbf = torch.empty(nblk * n2, dtype=torch.float32, device=ctxf.device)
    items = [(dkv_all[l], xhat, dwf[l * n2:(l + 1) * n2], dbf[l * n2:(l + 1) * n2]) for l in range(nblk)]
    for i in range(0, nblk, ops.MAX_BATCHED_WEIGHT_GRADS):
        part = items[i: i + ops.MAX_BATCHED_WEIGHT_GRADS]
        if not (len(part) > 1 and ops.linear_bwd_weight_batched(part, False)):
            for dy_, x_, dw_, db_ in part:
                _overwrite_weight_grad(dy_, x_, dw_, db_)
    ws, bs, gs, bes = [P[12] for P in blocks], [P[13] for P in blocks], [P[8] for P in blocks], [P[9] for P in blocks]
    targets = [(_gtarget(rt, w), _gtarget(rt, b) if b is not None else None, _gtarget(rt, g), _gtarget(rt, be)) for w, b, g, be in zip(ws, bs, gs, bes)]
    direct = all(t[0] is not None and t[2] is not None and t[3] is not None and (b is None or t[1] is not None) for t, b in zip(targets, bs))
    if not direct:
        targets = [(torch.empty_like(w), torch.empty_like(b) if b is not None else None, torch.empty_like(g), torch.empty_like(be))
                   for w, b, g, be in zip(ws, bs, gs, bes)]
    ops.unfold_context_grads(dwf, dbf, [w.detach() for w in ws], [g.detach() for g in gs], [b.detach() for b in bes], [t[0] for t in targets], [t[1] for t in targets],
                             [t[2] for t in targets], [t[3] for t in targets], accumulate=direct)
    return dctx, [(None, None, None, None) if direct else (t[2], t[3], t[0], t[1]) for t in targets]


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt: Runtime, feats, img2, img2_index, *params):
        pw, pb, pos, cls, gN, bN, wh, bh = params[:8]
        ns = 8
        nb = len(DEC_BLOCK_KEYS)
        blocks = [params[ns + i * nb: ns + (i + 1) * nb] for i in range(rt.c_depth)]
        grad = any(ctx.needs_input_grad)
        x, patches, batch, n = _patch_tokens_fwd(rt, img2, pw, pb, pos, with_cls=True, cls=cls, batch_index=img2_index)
        assert feats.shape == (batch, rt.n1, rt.dim), f'features {tuple(feats.shape)} do not match {batch} image-2 samples'
        ctxf = feats.detach().contiguous().float().view(batch * rt.n1, rt.dim)
        tape = []
        d = rt.dim
        cls_tail = rt.cls_tail and rt.c_depth > 0 and not rt.keep_attn    # the visualisation path wants every query row's map
        ln1 = None
        fold = rt.fold_context and not rt.exact and rt.c_depth > 1 and rt.c_depth <= ops.MAX_FOLDED_BLOCKS and rt.dim % 32 == 0
        kv_all = kv_saved = fold_bufs = None
        if fold:
            kv_all, kv_saved, fold_bufs = _context_kv_folded(rt, ctxf, blocks)
        for i, P in enumerate(blocks):
            nxt = (blocks[i + 1][0], blocks[i + 1][1]) if i + 1 < rt.c_depth else None     # the next block's norm1
            x, entry, ln1 = _dec_block_fwd(rt, x, ctxf, P, batch, n, grad, cls_tail and i == rt.c_depth - 1, index=i, ln1=ln1, next_ln=nxt,
                                           kv3=kv_all[i].view(batch, rt.n1, 2 * d) if fold else None)
            if grad:
                tape.append(entry)
            if rt.tap is not None:
                rt.tap[f'dec.x.{i}'] = x.clone()
        # final norm on the cls rows only (LayerNorm is row-wise; only x[:, 0] reaches the head, :400,:417)
        xcls = x if cls_tail else x.view(batch, n, d)[:, 0, :]
        y, mN, rN = ops.layernorm_fwd(xcls, gN, bN, LN_EPS, rt.act_dtype)
        logits = ops.gemm(y, rt.weight(wh), epilogue=EPI_STORE_F32, bias=bh)
        if grad:
            ctx.rt, ctx.tape, ctx.patches, ctx.batch, ctx.params = rt, tape, patches, batch, params
            ctx.ctxf, ctx.final, ctx.cls_tail = ctxf, (xcls, y, mN, rN), cls_tail
            ctx.feats_needs_grad = feats.requires_grad
            ctx.fold = (kv_all, kv_saved, fold_bufs) if fold else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        rt, batch, n, d = ctx.rt, ctx.batch, ctx.rt.n2, ctx.rt.dim
        params = ctx.params
        pw, pb, pos, cls, gN, bN, wh, bh = params[:8]
        ns, nb = 8, len(DEC_BLOCK_KEYS)
        grads = [None] * len(params)
        xcls, y, mN, rN = ctx.final
        # head: logits = y Wh^T + bh
        dl = _lp(rt, dlogits.contiguous().float())
        wh_act = rt.weight(wh)
        dy = ops.gemm(dl, wh_act, b_layout=B_KN)                         # [B, D]
        dwh, dbh = _weight_grads(rt, dl, y, wh, bh)
        if ctx.cls_tail:
            # the last block ran on the cls rows only: so does its gradient
            dx, dx_lp, dgN, dbN = _ln_bwd(rt, dy, xcls, gN, bN, mN, rN, want_lp=not rt.exact)
        else:
            # final LayerNorm touches the cls rows only; every other row of d(x) is zero
            dx = torch.zeros((batch * n, d), dtype=torch.float32, device=dy.device)
            dx3 = dx.view(batch, n, d)
            dx_lp = None
            dx_lp3 = None
            if not rt.exact:
                dx_lp = torch.zeros((batch * n, d), dtype=rt.act_dtype, device=dy.device)
                dx_lp3 = dx_lp.view(batch, n, d)[:, 0, :]
            _, _, dgN, dbN = _ln_bwd(rt, dy, xcls, gN, bN, mN, rN, dx_out=dx3[:, 0, :], dx_lp=dx_lp3)
        if rt.exact:
            dx_lp = dx
        grads[4], grads[5], grads[6], grads[7] = dgN, dbN, dwh, dbh
        dctx = None
        rt.ln_queue = [] if not rt.exact else None
        dkv_all3 = None
        if ctx.fold is not None:
            dkv_all = torch.empty_like(ctx.fold[0])
            dkv_all3 = dkv_all
        with _DwBatch(rt, blocks=True, kind='dec') as dwg:
            for i in reversed(range(rt.c_depth)):
                P = params[ns + i * nb: ns + (i + 1) * nb]
                entry = ctx.tape[i]
                ctx.tape[i] = None
                with dwg.block():
                    dx, dx_lp, dctx, blk = _dec_block_bwd(rt, dx, dx_lp, ctx.ctxf, dctx, P, entry, batch, n,
                                                          ctx.cls_tail and i == rt.c_depth - 1, i,
                                                          dkv3=dkv_all3[i].view(batch, rt.n1, 2 * d) if dkv_all3 is not None else None)
                if rt.tap is not None:
                    rt.tap[f'dec.dx.{i}'] = dx.clone()      # gradient w.r.t. the INPUT of decoder block i
                    if dctx is not None:
                        rt.tap[f'dec.dctx.{i}'] = dctx.clone()  # running d(features) after blocks c_depth-1 .. i
                base = ns + i * nb
                grads[base: base + nb] = blk
        if ctx.fold is not None:
            blocks = [params[ns + i * nb: ns + (i + 1) * nb] for i in range(rt.c_depth)]
            dctx, per_block = _context_kv_folded_bwd(rt, dkv_all, ctx.ctxf, ctx.fold[1], ctx.fold[2], blocks)
            for i, (dgx, dbx, dwkv, dbkv) in enumerate(per_block):
                base = ns + i * nb
                grads[base + 8], grads[base + 9], grads[base + 12], grads[base + 13] = dgx, dbx, dwkv, dbkv
            ctx.fold = None
        if rt.ln_queue is not None:
            ops.layernorm_bwd_finish(rt.ln_queue)       # the decoder's 4 x c_depth LayerNorm column sums
            rt.ln_queue = None
        dpw, dpb, dpos, dcls = _patch_tokens_bwd(rt, dx, ctx.patches, pw, pb, pos, with_cls=True, batch=batch)
        grads[0], grads[1], grads[2], grads[3] = dpw, dpb, dpos, dcls.view_as(cls)
        dfeats = dctx.view(batch, rt.n1, d) if ctx.feats_needs_grad and dctx is not None else None
        ctx.tape = ctx.patches = ctx.ctxf = ctx.final = None
        return (None, dfeats, None, None, *grads)
