"""``FlatAdamW``: torch.optim.AdamW's update (misc/optimizer.py:25-27) as ONE HIP multi-tensor pass over the flat
gradient buffer - gradient-norm clip (misc/utils.py:215-217), AdamW, refresh of the bf16 weight shadows the MFMA
kernels read, and ``zero_grad`` (misc/engine.py:231) - ``vited_adamw_step`` of include/vited.h.

It is a ``torch.optim.Optimizer`` (param_groups / state_dict / lr schedulers work as usual); ``engine.TrainStep``
drives it through ``step_flat``.  There is no CPU path: parameters must live on the GPU.
"""
from __future__ import annotations

import torch

from . import _lib

HYPER_HEADER, GROUP_WORDS, DESC_WORDS = 8, 8, 10


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.flat = None
        self._model = None
        self._desc = self._desc_key = None
        self._hyper = self._hyper_host = self._hyper_seen = None
        self._norm = self._ws = None

    # -- wiring --------------------------------------------------------------------------------
    def bind_flat(self, flat, model=None):
        """Use ``flat`` (engine.FlatGradients over the same parameters) as the gradient buffer; ``model`` (optional)
        supplies the bf16 weight shadows to refresh (its ``_runtimes``)."""
        mine = {id(p) for g in self.param_groups for p in g['params'] if p.requires_grad}
        if {id(p) for p in flat.params} != mine:
            raise ValueError('FlatAdamW.bind_flat: the flat gradient buffer does not cover exactly the optimizer\'s trainable parameters')
        dev = flat.flat.device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdamW runs on the MI355X HIP kernel only (no CPU path); use torch.optim.AdamW for CPU parameters')
        self.flat, self._model = flat, model
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        for p, off in zip(flat.params, flat.offsets):
            st = self.state[p]
            old_m, old_v = st.get('exp_avg'), st.get('exp_avg_sq')
            st['exp_avg'] = self.exp_avg[off: off + p.numel()].view_as(p)
            st['exp_avg_sq'] = self.exp_avg_sq[off: off + p.numel()].view_as(p)
            if old_m is not None:
                st['exp_avg'].copy_(old_m)
                st['exp_avg_sq'].copy_(old_v)
        self._hyper = torch.zeros(HYPER_HEADER + GROUP_WORDS * len(self.param_groups), dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros_like(self._hyper, device='cpu').pin_memory()
        self._hyper_seen = None
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(_lib.load().vited_adamw_workspace_bytes() // 4, dtype=torch.float32, device=dev)
        self._desc = self._desc_key = None

    @property
    def num_updates(self) -> int:
        return int(self._hyper[0].item()) if self._hyper is not None else 0

    def sync_hyperparameters(self):
        """Fold ``param_groups`` (what schedulers write) into the device hyper-parameter array when they changed.  The step
        counter (word 0) is owned by the kernel and never overwritten here."""
        vals = []
        for g in self.param_groups:
            vals += [float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), 0., 0., 0.]
        if vals != self._hyper_seen:
            self._hyper_host[HYPER_HEADER:] = torch.tensor(vals, dtype=torch.float32)
            self._hyper[HYPER_HEADER:].copy_(self._hyper_host[HYPER_HEADER:], non_blocking=True)
            self._hyper_seen = vals

    def _shadows(self):
        out = {}
        for rt in getattr(self._model, '_runtimes', {}).values():
            for (pid, tag), ent in rt._shadow.items():
                out.setdefault(pid, {})[tag] = ent[2]
        return out

    def _descriptors(self):
        shadows = self._shadows()
        group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g['params']}
        rows_, tile = [], 0
        for p, v, off in zip(self.flat.params, self.flat.views, self.flat.offsets):
            assert p.is_contiguous() and p.dtype == torch.float32, 'FlatAdamW: parameters must be contiguous fp32'
            r = p.shape[0] if p.dim() >= 2 else 1
            c = p.numel() // r
            sh = shadows.get(id(p), {})
            n, t = sh.get('n'), sh.get('t')
            rows_.append([p.data_ptr(), v.data_ptr(), self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off,
                          0 if n is None else n.data_ptr(), 0 if t is None else t.data_ptr(), r, c, tile, group_of[id(p)]])
            tile += ((r + 63) // 64) * ((c + 63) // 64)
        key = tuple(tuple(r[:6]) for r in rows_)
        if key != self._desc_key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FlatAdamW: the set of weight shadows changed during graph capture (run one eager step first)')
            self._desc = torch.tensor(rows_, dtype=torch.int64).to(self.flat.flat.device)
            self._desc_key, self._tiles = key, tile
        return self._desc

    # -- the update ----------------------------------------------------------------------------
    def step_flat(self, max_norm=None, zero_grad: bool = True):
        """clip(max_norm) + AdamW + shadow refresh (+ zero the gradients).  Returns the pre-clip gradient norm (device scalar)."""
        if self.flat is None:
            raise RuntimeError('FlatAdamW.step_flat: call bind_flat(FlatGradients) first (engine.TrainStep does)')
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.sync_hyperparameters()
        desc = self._descriptors()
        lib = _lib.load()
        _lib.check(lib.vited_adamw_step(desc.data_ptr(), desc.shape[0], self._tiles, self.flat.flat.data_ptr(), self.flat.flat.numel(),
                                        self._hyper.data_ptr(), float(max_norm) if max_norm else 0.0, int(zero_grad),
                                        self._norm.data_ptr(), self._ws.data_ptr(), self._ws.numel() * 4,
                                        torch.cuda.current_stream().cuda_stream), 'vited_adamw_step')
        # the kernel rewrote the parameters and their shadows together: keep the shadow cache entries current
        for rt in getattr(self._model, '_runtimes', {}).values():
            rt.mark_shadows_current()
        return self._norm[0]

    @torch.no_grad()
    def step(self, closure=None):
        """``torch.optim.Optimizer.step`` for callers that own the loop (the reference's NativeScaler path): gradients are
        taken from ``p.grad`` (folded into a flat buffer on first use), no clipping, no zeroing."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.flat is None:
            from .engine import FlatGradients
            grads = {id(p): p.grad for g in self.param_groups for p in g['params']}
            flat = FlatGradients([p for g in self.param_groups for p in g['params']])
            for p, v in zip(flat.params, flat.views):
                if grads[id(p)] is not None:
                    v.copy_(grads[id(p)])
            self.bind_flat(flat, self._model)
        else:
            self.flat.attach()
        self.step_flat(None, zero_grad=False)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        if self.flat is not None:
            self.flat.zero()        # keeps p.grad attached to the flat buffer
        else:
            super().zero_grad(set_to_none=set_to_none)

    # -- checkpoint compatibility with torch.optim.AdamW (misc/utils.py:130-142 saves optimizer.state_dict()) ----
    def state_dict(self):
        sd = super().state_dict()
        n = self.num_updates
        for st in sd['state'].values():
            st['step'] = torch.tensor(float(n))
        return sd

    def load_state_dict(self, state_dict):
        flat, model = self.flat, self._model
        super().load_state_dict(state_dict)
        steps = [float(st['step']) for st in self.state.values() if 'step' in st]
        if flat is not None:
            self.bind_flat(flat, model)          # re-point the moments at the flat buffers (copies the loaded values in)
            if steps:
                self._hyper[0] = max(steps)
