"""Functional wrappers over the C ABI: one Python function per entry point of include/vited.h.

These allocate outputs (PyTorch owns every buffer), validate shape/dtype/contiguity BEFORE the
launch and raise on any non-zero status.  They are not autograd-aware; ``functions.py`` composes
them into the encoder / decoder autograd Functions.  Every op launches on torch's current HIP
stream, never synchronises and never allocates on the device side, so the whole forward+backward
is hipGraph-capturable.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import (B_KN, B_NK, BF16, EPI_GELU, EPI_GELU_GRAD, EPI_MUL, EPI_MUL_GELU_GRAD, EPI_RESIDUAL, EPI_STORE, EPI_STORE_F32, F32)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f'vited ops support float32 and bfloat16 activations, got {dtype}') from None


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('vited ops run on the MI355X HIP kernels only: got a CPU tensor '
                               '(there is no CPU fallback; the fp32 CPU oracle lives in oracle/ for tests)')


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rows2d(t: torch.Tensor):
    """(tensor, row stride) of a [rows, dim] view whose last dim is dense."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'expected a 2-D tensor with a dense last dim, got shape {tuple(t.shape)} stride {t.stride()}')
    return t.stride(0)


_ws_cache = {}
_ws_pinned = False
_ws_retired = []


def pin_workspace():
    """Called when a hipGraph has been captured: its kernels baked the workspace address in, so a later (eager) growth
    must not free that buffer - it is retired (kept alive) and a larger one serves the eager calls."""
    global _ws_pinned
    _ws_pinned = True


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device (fp32 elements).  Contents never outlive one op."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    n = (int(nbytes) + 3) // 4
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('vited workspace must be sized before graph capture: run one eager step first')
        if buf is not None and _ws_pinned:
            _ws_retired.append(buf)
        buf = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


# ---------------------------------------------------------------------------------------------
def cast(src: torch.Tensor, dtype: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    _need_gpu(src, out)
    src = src.contiguous()
    if out is None:
        out = torch.empty_like(src, dtype=dtype)
    assert out.dtype == dtype and out.numel() == src.numel() and out.is_contiguous()
    _lib.check(_lib.load().vited_cast(_ptr(src), _code(src.dtype), _ptr(out), _code(dtype), src.numel(), _stream()), 'vited_cast')
    return out


def cast_transpose(w: torch.Tensor, dtype: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 [R, C] -> dtype [C, R] (transposed weight shadow)."""
    _need_gpu(w, out)
    assert w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
    if out is None:
        out = torch.empty((w.shape[1], w.shape[0]), dtype=dtype, device=w.device)
    assert out.dtype == dtype and out.shape == (w.shape[1], w.shape[0]) and out.is_contiguous()
    _lib.check(_lib.load().vited_cast_transpose(_ptr(w), _ptr(out), _code(dtype), w.shape[0], w.shape[1], _stream()),
               'vited_cast_transpose')
    return out


class WeightShadowPlan:
    """Device descriptor table for ``vited_cast_weights``: built once per set of (weight, shadow, transposed
    shadow) buffers, then every refresh is one launch."""

    def __init__(self, entries):
        """entries: iterable of (w fp32 [rows, cols] contiguous, dst bf16 [rows, cols] | None, dst_t bf16 [cols, rows] | None)."""
        rows_, keep, tile = [], [], 0
        for w, dst, dst_t in entries:
            _need_gpu(w, dst, dst_t)
            assert w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous() and w.data_ptr() % 16 == 0
            r, c = w.shape
            for t, shape in ((dst, (r, c)), (dst_t, (c, r))):
                assert t is None or (t.dtype == torch.bfloat16 and tuple(t.shape) == shape and t.is_contiguous()
                                     and t.data_ptr() % 16 == 0)
            if dst is None and dst_t is None:
                continue
            rows_.append([w.data_ptr(), _ptr(dst), _ptr(dst_t), r, c, tile])
            tile += ((r + 63) // 64) * ((c + 63) // 64)
            keep.append((w, dst, dst_t))
        self.count, self.total_tiles, self._keep = len(rows_), tile, keep
        self.key = tuple(tuple(r[:3]) for r in rows_)
        self.table = torch.tensor(rows_, dtype=torch.int64).to(keep[0][0].device) if rows_ else None

    def run(self):
        if self.count:
            _lib.check(_lib.load().vited_cast_weights(_ptr(self.table), self.count, self.total_tiles, _stream()), 'vited_cast_weights')


def patchify(img: torch.Tensor, patch: int, dtype: torch.dtype, batch_index: torch.Tensor | None = None,
             mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """img fp32 [B, C, S, S] (any batch stride, dense [C, S, S]) -> [B * (S/p)^2, C*p*p].  uint8 images are normalised on the
    fly: (pixel / 255 - mean[c]) / std[c] (``vited_patchify_u8``)."""
    _need_gpu(img, batch_index)
    u8 = img.dtype == torch.uint8
    if not u8 and img.dtype != torch.float32:
        img = img.float()
    assert img.dim() == 4 and img.shape[2] == img.shape[3], 'expected [B, C, S, S]'
    b, c, s, _ = img.shape
    if img.stride()[1:] != (s * s, s, 1):
        img = img.contiguous()
    nb = b
    if batch_index is not None:
        assert batch_index.dtype == torch.int64 and batch_index.is_contiguous()
        nb = batch_index.numel()
    g = s // patch
    out = torch.empty((nb * g * g, c * patch * patch), dtype=dtype, device=img.device)
    if u8:
        import ctypes
        m = (ctypes.c_float * c)(*[float(v) for v in list(mean)[:c]])
        sd = (ctypes.c_float * c)(*[float(v) for v in list(std)[:c]])
        _lib.check(_lib.load().vited_patchify_u8(_ptr(img), img.stride(0), _ptr(batch_index), _ptr(out), _code(dtype), nb, c, s, patch,
                                                 ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), _stream()), 'vited_patchify_u8')
        return out
    _lib.check(_lib.load().vited_patchify(_ptr(img), img.stride(0), _ptr(batch_index), _ptr(out), _code(dtype), nb, c, s,
                                          patch, _stream()), 'vited_patchify')
    return out


def crop_pairs_u8(regions: torch.Tensor, cells: torch.Tensor, erode: torch.Tensor, img_size: int) -> torch.Tensor:
    """regions uint8 [B, C, 2 S, 3 S] (any batch stride), cells int32 [B, 2], erode int32 [B] -> uint8 [B, 2, C, S, S]: the pair
    of eroded grid cells resized back to S x S (``vited_crop_pairs_u8``; div2k_patch.py:108-121,155-162)."""
    _need_gpu(regions, cells, erode)
    s = int(img_size)
    assert regions.dtype == torch.uint8 and regions.dim() == 4 and regions.shape[2] == 2 * s and regions.shape[3] == 3 * s
    b, c = regions.shape[:2]
    if regions.stride()[1:] != (6 * s * s, 3 * s, 1):
        regions = regions.contiguous()
    assert cells.dtype == torch.int32 and cells.shape == (b, 2) and cells.is_contiguous()
    assert erode.dtype == torch.int32 and erode.shape == (b,) and erode.is_contiguous()
    out = torch.empty((b, 2, c, s, s), dtype=torch.uint8, device=regions.device)
    _lib.check(_lib.load().vited_crop_pairs_u8(_ptr(regions), regions.stride(0), _ptr(cells), _ptr(erode), _ptr(out), b, c, s, _stream()),
               'vited_crop_pairs_u8')
    return out


def slice_rows_cast(x: torch.Tensor, row_offset: int, rows: int, dtype: torch.dtype) -> torch.Tensor:
    """fp32 [B, R, D] -> dtype [B * rows, D] taking rows [row_offset, row_offset + rows) of every batch."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
    b, r, d = x.shape
    out = torch.empty((b * rows, d), dtype=dtype, device=x.device)
    _lib.check(_lib.load().vited_slice_rows_cast(_ptr(x), _ptr(out), _code(dtype), b, r, row_offset, rows, d, _stream()),
               'vited_slice_rows_cast')
    return out


def write_cls_row(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor):
    """x fp32 [B, R, D]: x[:, 0] = cls + pos[0]."""
    _need_gpu(x, cls, pos)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    b, r, d = x.shape
    _lib.check(_lib.load().vited_write_cls_row(_ptr(x), _ptr(cls), _ptr(pos), b, r, d, _stream()), 'vited_write_cls_row')


def sum_rows(x: torch.Tensor) -> torch.Tensor:
    """[batch, width] (fp32 or bf16, dense rows) -> fp32 [width] column sums, deterministic."""
    _need_gpu(x)
    ld = _rows2d(x)
    b, w = x.shape
    lib = _lib.load()
    out = torch.empty(w, dtype=torch.float32, device=x.device)
    nbytes = lib.vited_sum_rows_workspace_bytes(b, w)
    ws = workspace(nbytes, x.device)
    _lib.check(lib.vited_sum_rows(_ptr(x), _code(x.dtype), ld, _ptr(out), b, w, _ptr(ws), ws.numel() * 4, _stream()),
               'vited_sum_rows')
    return out


# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype: torch.dtype, out=None):
    """x fp32 [rows, dim] (row-strided ok) -> (y [rows, dim] out_dtype, mean, rstd).  ``out`` = (y, mean, rstd) pre-made
    (e.g. row slices of larger tensors)."""
    _need_gpu(x, gamma, beta)
    assert x.dtype == torch.float32 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    ld = _rows2d(x)
    rows, dim = x.shape
    if out is not None:
        y, mean, rstd = out
        assert y.shape == (rows, dim) and y.dtype == out_dtype and _rows2d(y) == dim and mean.numel() == rows and rstd.numel() == rows
    else:
        y = torch.empty((rows, dim), dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vited_layernorm_fwd(_ptr(x), ld, _ptr(gamma), _ptr(beta), _ptr(y), _code(out_dtype), dim,
                                               _ptr(mean), _ptr(rstd), rows, dim, float(eps), _stream()), 'vited_layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_in=None, dx_out=None, want_lp: bool = False, dx_lp=None, dgamma=None,
                  dbeta=None):
    """Returns (dx fp32, dx_lp bf16 | None, dgamma, dbeta).  When ``dgamma`` / ``dbeta`` are given (fp32,
    contiguous - e.g. views of the flat gradient buffer) the column sums are ADDED onto them.

    dx = (dx_in or 0) + LN'(dy).  ``dx_out`` / ``dx_lp`` may be given as pre-made (row-strided)
    destinations - e.g. the cls rows of a zero-filled token-gradient tensor."""
    _need_gpu(dy, x, gamma, mean, rstd, dx_in, dx_out, dx_lp)
    dy_ld, x_ld = _rows2d(dy), _rows2d(x)
    rows, dim = x.shape
    assert dy.shape == x.shape and x.dtype == torch.float32
    lib = _lib.load()
    if dx_out is None:
        dx_out = torch.empty((rows, dim), dtype=torch.float32, device=x.device)
    if want_lp and dx_lp is None:
        dx_lp = torch.empty((rows, dim), dtype=torch.bfloat16, device=x.device)
    if dx_in is not None:
        assert dx_in.dtype == torch.float32 and dx_in.shape == x.shape
    accumulate = dgamma is not None
    if accumulate:
        assert dbeta is not None and dgamma.dtype == dbeta.dtype == torch.float32 and dgamma.is_contiguous() and dbeta.is_contiguous()
    else:
        dgamma = torch.empty(dim, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(dim, dtype=torch.float32, device=x.device)
    nbytes = lib.vited_layernorm_bwd_workspace_bytes(rows, dim)
    ws = workspace(nbytes, x.device)
    _lib.check(lib.vited_layernorm_bwd(
        _ptr(dy), _code(dy.dtype), dy_ld, _ptr(x), x_ld, _ptr(gamma), _ptr(mean), _ptr(rstd),
        _ptr(dx_in), _rows2d(dx_in) if dx_in is not None else 0, _ptr(dx_out), _rows2d(dx_out),
        _ptr(dx_lp), BF16, _rows2d(dx_lp) if dx_lp is not None else 0, _ptr(dgamma), _ptr(dbeta), int(accumulate), rows, dim,
        _ptr(ws), ws.numel() * 4, _stream()), 'vited_layernorm_bwd')
    return dx_out, dx_lp, dgamma, dbeta


# ---------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, b: torch.Tensor, *, b_layout: int = B_NK, epilogue: int = EPI_STORE, bias=None, aux=None,
         residual=None, out=None, out2=None, rows_per_batch: int = 0, out_rows_per_batch: int = 0, row_offset: int = 0,
         residual_bcast: bool = False, out_rows: int | None = None):
    """acc = a[M,K] . (b[N,K]^T | b[K,N]); see VITED_EPI_* in include/vited.h.

    Returns ``out`` (and ``out2`` for EPI_GELU)."""
    _need_gpu(a, b, bias, aux, residual, out)
    assert a.dtype == b.dtype, f'operand dtypes differ: {a.dtype} vs {b.dtype}'
    lda, ldb = _rows2d(a), _rows2d(b)
    m, k = a.shape
    n = b.shape[0] if b_layout == B_NK else b.shape[1]
    kb = b.shape[1] if b_layout == B_NK else b.shape[0]
    if kb != k:
        raise ValueError(f'contraction mismatch: A is [{m},{k}], B gives K={kb}')
    f32_out = epilogue in (EPI_RESIDUAL, EPI_STORE_F32)
    if out is None:
        rows = m if out_rows is None else out_rows
        out = torch.empty((rows, n), dtype=torch.float32 if f32_out else a.dtype, device=a.device)
    ldo = _rows2d(out)
    if epilogue in (EPI_GELU, EPI_GELU_GRAD):
        if out2 is None:
            out2 = torch.empty_like(out)
        assert out2.dtype == out.dtype and out2.shape == out.shape and _rows2d(out2) == ldo
    else:
        out2 = None
    if aux is not None:
        assert aux.dtype == a.dtype and _rows2d(aux) == ldo
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        assert residual.stride(-2) == ldo if residual.dim() >= 2 else True
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n
    _lib.check(_lib.load().vited_gemm(
        _ptr(a), lda, _ptr(b), ldb, b_layout, _code(a.dtype), m, n, k, epilogue, _ptr(bias), _ptr(aux), _ptr(residual),
        _ptr(out), _ptr(out2), ldo, rows_per_batch, out_rows_per_batch, row_offset, int(bool(residual_bcast)), _stream()),
        'vited_gemm')
    return (out, out2) if epilogue in (EPI_GELU, EPI_GELU_GRAD) else out


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor, want_bias: bool = True, dw_out=None, db_out=None):
    """dW[N,K] = dy[M,N]^T x[M,K] (fp32), dbias[N] = column sums of dy (fp32) or None.  With ``dw_out``
    (and ``db_out``) the results are ADDED onto those fp32 tensors (a parameter's .grad) instead."""
    _need_gpu(dy, x)
    assert dy.dtype == x.dtype and dy.shape[0] == x.shape[0]
    lddy, ldx = _rows2d(dy), _rows2d(x)
    m, n = dy.shape
    k = x.shape[1]
    lib = _lib.load()
    accumulate = dw_out is not None
    if accumulate:
        assert dw_out.dtype == torch.float32 and dw_out.is_contiguous() and dw_out.numel() == n * k
        dw, db = dw_out, (db_out if want_bias else None)
        assert db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == n)
    else:
        dw = torch.empty((n, k), dtype=torch.float32, device=x.device)
        db = torch.empty(n, dtype=torch.float32, device=x.device) if want_bias else None
    nbytes = lib.vited_linear_bwd_weight_workspace_bytes(m, n, k)
    ws = workspace(nbytes, x.device)
    _lib.check(lib.vited_linear_bwd_weight(_ptr(dy), lddy, _ptr(x), ldx, _code(x.dtype), m, n, k, _ptr(dw), _ptr(db),
                                           int(accumulate), _ptr(ws), ws.numel() * 4, _stream()), 'vited_linear_bwd_weight')
    return dw, db


MAX_BATCHED_WEIGHT_GRADS = 40


def linear_bwd_weight_batched(items, accumulate: bool) -> bool:
    """items: list of (dy [M, N], x [M, K], dw fp32 [N, K] contiguous, db fp32 [N] | None).  dW_i (+)= dy_i^T x_i and db_i (+)= column
    sums of dy_i for every item in ONE launch of the wide weight-gradient kernel + one slab-sum launch
    (``vited_linear_bwd_weight_batched``).  Returns False (nothing launched) when the set is not covered - the caller then issues
    ``linear_bwd_weight`` per item."""
    import ctypes as C
    n = len(items)
    if n < 1 or n > MAX_BATCHED_WEIGHT_GRADS:
        return False
    for dy, x, dw, db in items:
        _need_gpu(dy, x, dw, db)
        if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dy.shape[0] != x.shape[0]:
            return False
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == dy.shape[1] * x.shape[1]
        assert db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == dy.shape[1])
        if dw.data_ptr() % 16 or (db is not None and db.data_ptr() % 16) or dy.data_ptr() % 16 or x.data_ptr() % 16:
            return False                 # the batched kernels use 16-byte accesses throughout
    i64, vp = C.c_int64 * n, C.c_void_p * n
    M = i64(*[it[0].shape[0] for it in items])
    N = i64(*[it[0].shape[1] for it in items])
    K = i64(*[it[1].shape[1] for it in items])
    lib = _lib.load()
    if not lib.vited_linear_bwd_weight_batched_supported(n, M, N, K, BF16):
        return False
    lddy = i64(*[_rows2d(it[0]) for it in items])
    ldx = i64(*[_rows2d(it[1]) for it in items])
    dY = vp(*[it[0].data_ptr() for it in items])
    X = vp(*[it[1].data_ptr() for it in items])
    dW = vp(*[it[2].data_ptr() for it in items])
    dB = vp(*[_ptr(it[3]) or None for it in items])
    ws = workspace(lib.vited_linear_bwd_weight_batched_workspace_bytes(n, M, N, K), items[0][0].device)
    _lib.check(lib.vited_linear_bwd_weight_batched(n, dY, lddy, X, ldx, M, N, K, dW, dB, BF16, int(bool(accumulate)), _ptr(ws),
                                                   ws.numel() * 4, _stream()), 'vited_linear_bwd_weight_batched')
    return True


# ---------------------------------------------------------------------------------------------
def linear_layernorm_supported(m: int, n: int, k: int, dtype: torch.dtype) -> bool:
    """Whether the row-complete fused Linear + LayerNorm kernels (gemm_row.hip) cover the shape."""
    return dtype == torch.bfloat16 and bool(_lib.load().vited_linear_layernorm_supported(int(m), int(n), int(k)))


def linear_residual_layernorm_fwd(a, w, bias, residual, gamma=None, beta=None, eps: float = 1e-6, out=None):
    """y = residual + a w^T + bias (fp32) and, with gamma / beta, h = LayerNorm(y) (bf16), mean, rstd - one kernel
    (``vited_linear_residual_layernorm_fwd``).  Returns (y, h | None, mean | None, rstd | None)."""
    _need_gpu(a, w, bias, residual, gamma, beta, out)
    assert a.dtype == w.dtype == torch.bfloat16 and residual.dtype == torch.float32
    lda, ldw, ldr = _rows2d(a), _rows2d(w), _rows2d(residual)
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and residual.shape == (m, n)
    y = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    assert y.dtype == torch.float32 and y.shape == (m, n)
    h = mean = rstd = None
    if gamma is not None:
        h = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
        mean = torch.empty(m, dtype=torch.float32, device=a.device)
        rstd = torch.empty(m, dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().vited_linear_residual_layernorm_fwd(
        _ptr(a), lda, _ptr(w), ldw, _ptr(bias), _ptr(residual), ldr, _ptr(y), _rows2d(y), _ptr(gamma), _ptr(beta), float(eps),
        _ptr(h), n, _ptr(mean), _ptr(rstd), m, n, k, _stream()), 'vited_linear_residual_layernorm_fwd')
    return y, h, mean, rstd


def linear_layernorm_bwd(dy, wt, x, gamma, mean, rstd, dx_in=None, dx_out=None, want_lp: bool = False, dgamma=None, dbeta=None,
                         defer=None):
    """dx = (dx_in or 0) + LN'(dy wt^T; x, mean, rstd, gamma) in one kernel (``vited_linear_layernorm_bwd``): the input
    gradient of ``y = LayerNorm(x) W^T`` without materialising d(LayerNorm output).  ``wt`` = the transposed weight shadow
    [N, K].  Returns (dx fp32, dx_lp bf16 | None, dgamma, dbeta); given ``dgamma`` / ``dbeta`` are ADDED onto.
    ``defer`` (a list): the column sums are NOT finished here - (partials, rows, dgamma, dbeta, accumulate) is appended and
    ``layernorm_bwd_finish(defer)`` later finishes many LayerNorms with one launch."""
    _need_gpu(dy, wt, x, gamma, mean, rstd, dx_in, dx_out)
    assert dy.dtype == wt.dtype == torch.bfloat16 and x.dtype == torch.float32
    segments = 1
    if dy.dim() == 3:
        # [L, M, seg_k]: the contraction dim arrives as L tensors (one per decoder block), wt is [N, L * seg_k]
        assert dy.stride(2) == 1 and dy.stride(0) % 8 == 0
        segments, m, seg_k = dy.shape
        lddy, seg_stride, k = dy.stride(1), dy.stride(0), segments * seg_k
    else:
        lddy = _rows2d(dy)
        m, k = dy.shape
    ldwt, ldx = _rows2d(wt), _rows2d(x)
    n = wt.shape[0]
    assert wt.shape[1] == k and x.shape == (m, n)
    lib = _lib.load()
    if dx_out is None:
        dx_out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    dx_lp = torch.empty((m, n), dtype=torch.bfloat16, device=x.device) if want_lp else None
    accumulate = dgamma is not None
    if accumulate:
        assert dbeta is not None and dgamma.dtype == dbeta.dtype == torch.float32 and dgamma.is_contiguous() and dbeta.is_contiguous()
    else:
        dgamma = torch.empty(n, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(n, dtype=torch.float32, device=x.device)
    if defer is not None:
        rows = int(lib.vited_linear_layernorm_bwd_partial_rows(m))
        ws = torch.empty(rows * 2 * n, dtype=torch.float32, device=x.device)     # lives until the flush
        defer.append((ws, rows, dgamma, dbeta, accumulate))
    else:
        ws = workspace(lib.vited_linear_layernorm_bwd_workspace_bytes(m, n), x.device)
    tail = (_ptr(x), ldx, _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx_in), _rows2d(dx_in) if dx_in is not None else 0, _ptr(dx_out),
            _rows2d(dx_out), _ptr(dx_lp), n, 0 if defer is not None else _ptr(dgamma), 0 if defer is not None else _ptr(dbeta), int(accumulate))
    if segments > 1:
        _lib.check(lib.vited_linear_layernorm_bwd_segmented(_ptr(dy), lddy, seg_k, seg_stride, segments, _ptr(wt), ldwt, *tail, m, n, _ptr(ws),
                                                            ws.numel() * 4, _stream()), 'vited_linear_layernorm_bwd_segmented')
    else:
        _lib.check(lib.vited_linear_layernorm_bwd(_ptr(dy), lddy, _ptr(wt), ldwt, *tail, m, n, k, _ptr(ws), ws.numel() * 4, _stream()),
                   'vited_linear_layernorm_bwd')
    return dx_out, dx_lp, dgamma, dbeta


def layernorm_bwd_finish(entries):
    """Finish the deferred column sums of ``linear_layernorm_bwd(..., defer=entries)``: one launch per 16 LayerNorms."""
    import ctypes as C
    n = len(entries)
    if not n:
        return
    vp, ci = C.c_void_p * n, C.c_int * n
    dim = entries[0][2].numel()
    _lib.check(_lib.load().vited_layernorm_bwd_finish_batched(
        n, vp(*[e[0].data_ptr() for e in entries]), ci(*[e[1] for e in entries]), vp(*[e[2].data_ptr() for e in entries]),
        vp(*[e[3].data_ptr() for e in entries]), ci(*[int(e[4]) for e in entries]), dim, _stream()), 'vited_layernorm_bwd_finish_batched')
    entries.clear()


# ---------------------------------------------------------------------------------------------
MAX_FOLDED_BLOCKS = 16


def fold_context_weights(ws, biases, gammas, betas, out=None):
    """Folded kv weights of several decoder blocks: W'_l = W_l o gamma_l (bf16, stacked [L N, K] and transposed [K, L N]) and
    b'_l = b_l + W_l beta_l (fp32 [L N]) - ``vited_fold_context_weights``.  ``out`` = (w, wt, b) buffers to refresh in place."""
    import ctypes as C
    n_blk = len(ws)
    _need_gpu(*ws, *gammas, *betas)
    n, k = ws[0].shape
    dev = ws[0].device
    if out is None:
        out = (torch.empty((n_blk * n, k), dtype=torch.bfloat16, device=dev), torch.empty((k, n_blk * n), dtype=torch.bfloat16, device=dev),
               torch.empty(n_blk * n, dtype=torch.float32, device=dev))
    vp = C.c_void_p * n_blk
    for t in list(ws) + list(gammas) + list(betas):
        assert t.dtype == torch.float32 and t.is_contiguous()
    _lib.check(_lib.load().vited_fold_context_weights(n_blk, vp(*[w.data_ptr() for w in ws]), vp(*[_ptr(b) or None for b in biases]),
                                                      vp(*[g.data_ptr() for g in gammas]), vp(*[b.data_ptr() for b in betas]), n, k,
                                                      _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()), 'vited_fold_context_weights')
    return out


def unfold_context_grads(dwf, dbf, ws, gammas, betas, dws, dbiases, dgammas, dbetas, accumulate: bool):
    """Gradients of the folded weights / bias -> dW_l, db_l, dgamma_l, dbeta_l (``vited_unfold_context_grads``)."""
    import ctypes as C
    n_blk = len(ws)
    n, k = ws[0].shape
    assert dwf.dtype == dbf.dtype == torch.float32 and dwf.is_contiguous() and dbf.is_contiguous()
    assert dwf.shape == (n_blk * n, k) and dbf.numel() == n_blk * n
    vp = C.c_void_p * n_blk
    _lib.check(_lib.load().vited_unfold_context_grads(n_blk, _ptr(dwf), _ptr(dbf), vp(*[w.data_ptr() for w in ws]),
                                                      vp(*[g.data_ptr() for g in gammas]), vp(*[b.data_ptr() for b in betas]),
                                                      vp(*[t.data_ptr() for t in dws]),
                                                      vp(*[_ptr(t) or None for t in dbiases]), vp(*[t.data_ptr() for t in dgammas]),
                                                      vp(*[t.data_ptr() for t in dbetas]), n, k, int(bool(accumulate)), _stream()),
               'vited_unfold_context_grads')


# ---------------------------------------------------------------------------------------------
def mlp_fused_supported(x: torch.Tensor, w1: torch.Tensor) -> bool:
    """The fused MLP kernel covers bf16, embed dim 384, hidden 1536 (every shipped pjs config)."""
    return w1.dtype == torch.bfloat16 and tuple(w1.shape) == (1536, 384) and x.shape[-1] == 384


def mlp_fwd(x, gamma, beta, w1, b1, w2, b2, eps: float, save: bool = True, out=None):
    """y = x + fc2(gelu(fc1(LayerNorm(x)))) in one kernel (``vited_mlp_fwd``).  x fp32 [rows, 384]; w1 / w2 the bf16 weights.
    Returns (y, saved) with saved = (mean, rstd, h, gd, u) or None.  ``out`` = (y, mean, rstd, h, gd, u) pre-made outputs
    (row slices of larger tensors are fine for y; the others must be dense)."""
    _need_gpu(x, gamma, beta, w1, b1, w2, b2)
    assert x.dtype == torch.float32 and w1.dtype == w2.dtype == torch.bfloat16 and w1.is_contiguous() and w2.is_contiguous()
    ldx = _rows2d(x)
    rows, dim = x.shape
    hidden = w1.shape[0]
    assert w1.shape == (hidden, dim) and w2.shape == (dim, hidden)
    dev = x.device
    if out is not None:
        y, mean, rstd, h, gd, u = out
    else:
        y = torch.empty((rows, dim), dtype=torch.float32, device=dev)
        mean = rstd = h = gd = u = None
        if save:
            mean = torch.empty(rows, dtype=torch.float32, device=dev)
            rstd = torch.empty(rows, dtype=torch.float32, device=dev)
            h = torch.empty((rows, dim), dtype=torch.bfloat16, device=dev)
            gd = torch.empty((rows, hidden), dtype=torch.bfloat16, device=dev)
            u = torch.empty((rows, hidden), dtype=torch.bfloat16, device=dev)
    if save:
        assert h.is_contiguous() and gd.is_contiguous() and u.is_contiguous() and h.shape == (rows, dim) and gd.shape == u.shape == (rows, hidden)
    _lib.check(_lib.load().vited_mlp_fwd(_ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(y), _rows2d(y),
                                         _ptr(h) if save else 0, _ptr(gd) if save else 0, _ptr(u) if save else 0,
                                         _ptr(mean) if save else 0, _ptr(rstd) if save else 0, rows, dim, hidden, float(eps), _stream()),
               'vited_mlp_fwd')
    return y, ((mean, rstd, h, gd, u) if save else None)


def block_fwd(x, heads: int, ln1_g, ln1_b, wqkv, bqkv, wproj, bproj, ln2_g, ln2_b, w1, b1, w2, b2, eps: float = 1e-6):
    """Encoder Block forward through ``vited_block_fwd``: x fp32 [B, N, D] -> y fp32 [B, N, D] (bf16 weights, inference form)."""
    _need_gpu(x, wqkv, wproj, w1, w2)
    assert x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
    assert all(t.dtype == torch.bfloat16 and t.is_contiguous() for t in (wqkv, wproj, w1, w2))
    b, n, d = x.shape
    hidden = w1.shape[0]
    lib = _lib.load()
    y = torch.empty_like(x)
    ws = workspace(lib.vited_block_workspace_bytes(b, n, d, hidden, heads) + 256, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.vited_block_fwd(_ptr(x), _ptr(y), b, n, d, heads, hidden, _ptr(ln1_g), _ptr(ln1_b), _ptr(wqkv), _ptr(bqkv), _ptr(wproj),
                                   _ptr(bproj), _ptr(ln2_g), _ptr(ln2_b), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), float(eps), base,
                                   ws.numel() * 4 - (base - ws.data_ptr()), _stream()), 'vited_block_fwd')
    return y


def cross_block_fwd(x, context, heads: int, ln1, wqkv, bqkv, wproj, bproj, lnq, lnc, wq, bq, wkv, bkv, wcproj, bcproj, ln2, w1, b1, w2, b2,
                    eps: float = 1e-6):
    """Decoder CrossBlock forward through ``vited_cross_block_fwd``: x fp32 [B, N2, D], context fp32 [B, N1, D] -> y fp32 [B, N2, D]
    (bf16 weights, inference form).  ln1 / lnq / lnc / ln2 = (gamma, beta) of norm1 / norm_cross / norm_context / norm2."""
    _need_gpu(x, context, wqkv, wproj, wq, wkv, wcproj, w1, w2)
    assert x.dtype == context.dtype == torch.float32 and x.dim() == context.dim() == 3 and x.is_contiguous() and context.is_contiguous()
    assert all(t.dtype == torch.bfloat16 and t.is_contiguous() for t in (wqkv, wproj, wq, wkv, wcproj, w1, w2))
    b, n, d = x.shape
    nc = context.shape[1]
    assert context.shape == (b, nc, d)
    hidden = w1.shape[0]
    lib = _lib.load()
    y = torch.empty_like(x)
    ws = workspace(lib.vited_cross_block_workspace_bytes(b, n, nc, d, hidden, heads) + 256, x.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.vited_cross_block_fwd(_ptr(x), _ptr(context), _ptr(y), b, n, nc, d, heads, hidden, _ptr(ln1[0]), _ptr(ln1[1]), _ptr(wqkv),
                                         _ptr(bqkv), _ptr(wproj), _ptr(bproj), _ptr(lnq[0]), _ptr(lnq[1]), _ptr(lnc[0]), _ptr(lnc[1]),
                                         _ptr(wq), _ptr(bq), _ptr(wkv), _ptr(bkv), _ptr(wcproj), _ptr(bcproj), _ptr(ln2[0]), _ptr(ln2[1]),
                                         _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), float(eps), base, ws.numel() * 4 - (base - ws.data_ptr()),
                                         _stream()), 'vited_cross_block_fwd')
    return y


# ---------------------------------------------------------------------------------------------
def _head_view(t: torch.Tensor, heads: int, head_dim: int):
    """t is [B, N, heads*head_dim] (a last-dim slice of the packed projection): (ptr-holder, bs, ts)."""
    assert t.dim() == 3 and t.stride(2) == 1 and t.shape[2] == heads * head_dim
    return t.stride(0), t.stride(1)


def attention_fwd(q, k, v, heads: int, scale: float, kv_index=None):
    """q [B,Nq,D], k/v [B,Nk,D] (strided views of the packed qkv / kv projections are fine)
    -> (o [B,Nq,D] contiguous, lse fp32 [B,H,Nq]).  With ``kv_index`` (int64 [B]) batch item b attends over
    k[kv_index[b]] / v[kv_index[b]] and k / v may hold any number of items (inference only)."""
    _need_gpu(q, k, v, kv_index)
    b, nq, d = q.shape
    nk = k.shape[1]
    hd = d // heads
    assert q.dtype == k.dtype == v.dtype and k.shape == v.shape and k.shape[2] == d
    if kv_index is None:
        assert k.shape[0] == b
    else:
        assert kv_index.dtype == torch.int64 and kv_index.is_contiguous() and kv_index.numel() == b
    q_bs, q_ts = _head_view(q, heads, hd)
    k_bs, k_ts = _head_view(k, heads, hd)
    v_bs, v_ts = _head_view(v, heads, hd)
    o = torch.empty((b, nq, d), dtype=q.dtype, device=q.device)
    lse = torch.empty((b, heads, nq), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().vited_attention_fwd_indexed(_ptr(q), q_bs, q_ts, _ptr(k), k_bs, k_ts, _ptr(v), v_bs, v_ts, _ptr(kv_index),
                                                       _ptr(o), nq * d, d, _ptr(lse), _code(q.dtype), b, heads, nq, nk, hd,
                                                       float(scale), _stream()), 'vited_attention_fwd')
    return o, lse


def attention_bwd(q, k, v, o, do, lse, heads: int, scale: float, dq, dk, dv):
    """Writes dq/dk/dv (pre-allocated, same strided conventions as q/k/v)."""
    _need_gpu(q, k, v, o, do, lse, dq, dk, dv)
    b, nq, d = q.shape
    nk = k.shape[1]
    hd = d // heads
    assert o.is_contiguous() and do.is_contiguous() and o.dtype == do.dtype == q.dtype
    q_bs, q_ts = _head_view(q, heads, hd)
    k_bs, k_ts = _head_view(k, heads, hd)
    v_bs, v_ts = _head_view(v, heads, hd)
    dq_bs, dq_ts = _head_view(dq, heads, hd)
    dk_bs, dk_ts = _head_view(dk, heads, hd)
    dv_bs, dv_ts = _head_view(dv, heads, hd)
    delta = torch.empty((b, heads, nq), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().vited_attention_bwd(
        _ptr(q), q_bs, q_ts, _ptr(k), k_bs, k_ts, _ptr(v), v_bs, v_ts, _ptr(o), _ptr(do), nq * d, d, _ptr(lse), _ptr(delta),
        _ptr(dq), dq_bs, dq_ts, _ptr(dk), dk_bs, dk_ts, _ptr(dv), dv_bs, dv_ts, _code(q.dtype), b, heads, nq, nk, hd,
        float(scale), _stream()), 'vited_attention_bwd')
    return dq, dk, dv


def last_paths():
    lib = _lib.load()
    return lib.vited_last_gemm_path(), lib.vited_last_attention_path()
