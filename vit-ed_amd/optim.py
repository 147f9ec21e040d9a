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
    manages_weight_shadows = True      # functions._install_optimizer_step_hook: this optimizer publishes its updates itself

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None):
        """``model`` (the HIP ViT-ED whose parameters these are) lets the kernel refresh the model's bf16 weight shadows in the
        same pass; without it the parameters' version counters are bumped after every update instead, so the model recasts its
        shadows on the next forward (correct, one extra launch per weight)."""
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.flat = None
        self._model = model
        self._loaded_step = None       # step count of a state_dict loaded before bind_flat (the reference's resume order)
        self._desc = self._desc_key = None
        self._hyper = self._hyper_seen = None
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
        rebind = self.flat is flat and getattr(self, 'exp_avg', None) is not None and self.exp_avg.numel() == flat.flat.numel()
        self.flat = flat
        if model is not None:
            self._model = model
        if not rebind:
            # first binding: the moment buffers, the hyper-parameter array and the workspace are allocated ONCE - a captured
            # update graph bakes their addresses in, so a later load_state_dict copies INTO them (below) instead of replacing them
            self.exp_avg = torch.zeros_like(flat.flat)
            self.exp_avg_sq = torch.zeros_like(flat.flat)
            self._hyper = torch.zeros(HYPER_HEADER + GROUP_WORDS * len(self.param_groups), dtype=torch.float32, device=dev)
            self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._ws = torch.empty(_lib.load().vited_adamw_workspace_bytes() // 4, dtype=torch.float32, device=dev)
            self._desc = self._desc_key = None
        self._hyper_seen = None
        for p, off in zip(flat.params, flat.offsets):
            st = self.state[p]
            old_m, old_v = st.get('exp_avg'), st.get('exp_avg_sq')
            new_m = self.exp_avg[off: off + p.numel()].view_as(p)
            new_v = self.exp_avg_sq[off: off + p.numel()].view_as(p)
            if old_m is not None and old_m.data_ptr() != new_m.data_ptr():
                new_m.copy_(old_m)
                new_v.copy_(old_v)
            st['exp_avg'], st['exp_avg_sq'] = new_m, new_v
        if self._loaded_step is not None:
            # torch.optim.AdamW keeps one step count per parameter; they advance together, so one counter serves (bias correction)
            self._hyper[0] = float(self._loaded_step)
            self._loaded_step = None

    @property
    def num_updates(self) -> int:
        if self._hyper is not None:
            return int(self._hyper[0].item())
        return int(self._loaded_step or 0)

    def sync_hyperparameters(self):
        """Fold ``param_groups`` (what schedulers write) into the device hyper-parameter array when they changed.  The step
        counter (word 0) is owned by the kernel and never overwritten here."""
        vals = []
        for g in self.param_groups:
            vals += [float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), 0., 0., 0.]
        if vals != self._hyper_seen:
            # a fresh host tensor per change and an ordinary (host-synchronous) copy of ~16 floats: a reused pinned buffer
            # with a non-blocking copy could be rewritten with the NEXT iteration's rate before this copy executed (the host
            # runs ahead of the device under graph replay)
            self._hyper[HYPER_HEADER:].copy_(torch.tensor(vals, dtype=torch.float32))
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
        if not capturing:
            self._publish_update()
        return self._norm[0]

    def shadow_signature(self):
        """What a captured update graph baked in besides this optimizer's own buffers: the set of weight-shadow buffers."""
        return tuple(sorted((pid, tag, ent[2].data_ptr()) for rt in getattr(self._model, '_runtimes', {}).values()
                            for (pid, tag), ent in rt._shadow.items()))

    def _publish_update(self):
        """The kernel raw-wrote the fp32 parameters (and the shadows listed in its descriptor table).  Bump every
        parameter's version counter - anything that caches by version (a Runtime whose shadows the table did NOT cover, e.g.
        an optimizer built without ``model``) then recasts - and mark the shadows the kernel did refresh as current."""
        params = self.flat.params
        torch.autograd.graph.increment_version(params)
        refreshed = {ptr for row in (self._desc_key or ()) for ptr in row[4:6] if ptr}
        by_id = {id(p): p for p in params}
        for rt in getattr(self._model, '_runtimes', {}).values():
            for (pid, tag), ent in list(rt._shadow.items()):
                p = by_id.get(pid)
                if p is not None and ent[2].data_ptr() in refreshed:
                    rt._shadow[(pid, tag)] = (p._version, ent[1], ent[2])

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
        n = self.num_updates                     # the kernel's counter once bound, else the step of a loaded checkpoint
        for st in sd['state'].values():
            st['step'] = torch.tensor(float(n))
        return sd

    def load_state_dict(self, state_dict):
        """Works in either order relative to ``bind_flat`` / ``TrainStep`` (the reference resumes as: build optimizer,
        ``load_checkpoint``, then build the loop - misc/utils.py:57-70): the loaded step count is kept until the flat buffers
        exist.  Once bound the loaded moments are copied INTO the existing flat buffers, so a captured update graph stays valid."""
        flat = self.flat
        super().load_state_dict(state_dict)
        steps = [float(st['step']) for st in self.state.values() if 'step' in st]
        self._loaded_step = max(steps) if steps else None
        if flat is not None:
            self.bind_flat(flat)                 # copies the loaded moments into the flat buffers and applies the step count
