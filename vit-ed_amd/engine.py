"""The driven training step of the reference (misc/engine.py:183-257, misc/utils.py:206-232,319-344)
re-plumbed for MI355X: one process per GPU, RCCL all-reduce of a single flat fp32 gradient buffer
over xGMI instead of c10d's 25 MB DDP buckets, bf16 autocast instead of fp16 + GradScaler, and the
whole forward+backward replayable as one hipGraph.

Only what drives the hot path is here; data loading, logging,
checkpoint rotation and validation stay in the reference's ``Trainer``.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------
# process group (misc/utils.py:319-344)
# ---------------------------------------------------------------------------------------------
def configure_ddp(backend: str | None = None):
    """env:// rendezvous like the reference, but device-aware: 'nccl' (= RCCL on ROCm) when a GPU is
    present, 'gloo' otherwise (the reference hard-codes nccl and cannot run BASELINE config 0 on CPU)."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        kw = {}
        if use_cuda:
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend=backend or ('nccl' if use_cuda else 'gloo'), init_method='env://',
                                world_size=world, rank=rank, **kw)
    dist.barrier()
    return local_rank, rank, world


# ---------------------------------------------------------------------------------------------
# flat gradient buffer + all-reduce (replaces DistributedDataParallel, misc/engine.py:75)
# ---------------------------------------------------------------------------------------------
def _avg_supported(group=None) -> bool:
    """ReduceOp.AVG exists on the nccl (= RCCL) backend only; gloo has SUM."""
    return dist.is_initialized() and dist.get_backend(group) == 'nccl'


class FlatGradients:
    """All parameter gradients as views of ONE contiguous fp32 buffer.

    ``p.grad`` is pre-set to a view, so autograd accumulates in place and the buffer is always the
    gradient; ``zero()`` replaces ``optimizer.zero_grad()``; ``all_reduce_mean()`` is the data-parallel
    exchange: one RCCL all-reduce of the whole buffer (133 MB fp32 at config A - SURVEY.md 2.3 C1 -
    instead of six 25 MB buckets), optionally bf16-compressed on the wire.

    ``early`` (an iterable of parameters) are laid out FIRST: ``buckets()`` then yields two contiguous
    ranges, [early | rest].  TrainStep passes the decoder-only parameters there - their gradients are
    final when the decoder's backward returns, so their all-reduce can run under the encoder's backward
    (the overlap c10d's DDP reducer gives the reference, misc/engine.py:75)."""

    def __init__(self, params, compress_bf16: bool = False, early=None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('no trainable parameters')
        early_ids = {id(p) for p in (early or ())}
        first = [p for p in params if id(p) in early_ids]
        rest = [p for p in params if id(p) not in early_ids]
        self.params = first + rest
        dev = self.params[0].device
        # every view starts on a 64-byte boundary (config H's [1] head bias would otherwise leave everything behind it on an odd
        # word: the kernels that add into these views use 16-byte accesses); the padding words stay zero
        align = lambda n: (n + 15) // 16 * 16
        self.offsets, off = [], 0
        for i, p in enumerate(self.params):
            self.offsets.append(off)
            off += align(p.numel())
            if i + 1 == len(first):
                self.split = off                        # [0, split) = early bucket, [split, total) = the rest
        total = off
        if not first:
            self.split = 0
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.wire = torch.empty(total, dtype=torch.bfloat16, device=dev) if compress_bf16 else None
        self.views = []
        for p, o in zip(self.params, self.offsets):
            v = self.flat[o: o + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
        self._pending = []

    def buckets(self):
        total = self.flat.numel()
        return [(0, self.split), (self.split, total)] if 0 < self.split < total else [(0, total)]

    def attach(self):
        """Make every ``p.grad`` the flat view again.  ``optimizer.zero_grad()`` (the reference's loop,
        misc/engine.py:231; set_to_none=True by default since torch 2.0) detaches them: a parameter whose grad
        is None gets its view back ZEROED, one that received a fresh tensor has it copied into the view."""
        detached = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not v]
        if not detached:
            return 0
        if len(detached) == len(self.params) and all(p.grad is None for p, _ in detached):
            self.flat.zero_()                      # the common case: one fill instead of one per parameter
        else:
            for p, v in detached:
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
        for p, v in detached:
            p.grad = v
        return len(detached)

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:          # someone called zero_grad(set_to_none=True): re-attach
                p.grad = v

    # -- the exchange ------------------------------------------------------------------------
    def start_all_reduce(self, lo: int, hi: int, group=None):
        """Launch the mean all-reduce of flat[lo:hi] WITHOUT waiting for it (RCCL runs it on the process
        group's own stream behind everything already queued on the current stream).  ``finish_all_reduce``
        joins."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        # VITED_FORCE_COLLECTIVE=1: issue the collective on a one-rank group too (a one-GPU box can then exercise the RCCL calls and
        # their ordering against the graph replays; the mean over one rank is the identity)
        force = world == 1 and dist.is_initialized() and os.environ.get('VITED_FORCE_COLLECTIVE') == '1'
        if (world == 1 and not force) or hi <= lo:
            return
        seg = self.flat[lo:hi]
        if self.wire is not None:
            buf = self.wire[lo:hi]
            buf.copy_(seg)
        else:
            buf = seg
        avg = _avg_supported(group)
        work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True)
        self._pending.append((work, seg, buf, None if avg else 1.0 / world))

    def finish_all_reduce(self):
        for work, seg, buf, scale in self._pending:
            work.wait()                       # the current stream now waits for the collective
            if buf is not seg:
                seg.copy_(buf)
            if scale is not None:
                seg.mul_(scale)
        self._pending = []

    def all_reduce_mean(self, group=None):
        self.start_all_reduce(0, self.flat.numel(), group)
        self.finish_all_reduce()

    def clip_(self, max_norm: float):
        """clip_grad_norm_ on the flat buffer: one norm kernel instead of 280 (misc/utils.py:215-217)."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


def broadcast_parameters(model, src: int = 0, group=None):
    """DDP ctor semantics: every rank starts from rank 0's parameters - as ONE
    broadcast of a flattened copy (the reference's DDP ctor also coalesces) instead of one per tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p.data for p in model.parameters()]
    by_dtype = {}
    for p in params:
        by_dtype.setdefault(p.dtype, []).append(p)
    for ps in by_dtype.values():
        flat = torch.cat([p.reshape(-1) for p in ps])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in ps:
            p.copy_(flat[off: off + p.numel()].view_as(p))
            off += p.numel()


# ---------------------------------------------------------------------------------------------
# optimizer (misc/optimizer.py:10-46)
# ---------------------------------------------------------------------------------------------
def param_groups_no_decay_1d(model):
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim == 1 or name.endswith('.bias')) else decay).append(p)
    return [{'params': decay}, {'params': no_decay, 'weight_decay': 0.}]


def build_optimizer(config, model, capturable: bool = False, fused_hip: bool | None = None):
    """misc/optimizer.py:10-46: AdamW / SGD with the no-decay group for 1-D parameters and ``*.bias``.

    On a GPU model AdamW is ``optim.FlatAdamW`` (the HIP multi-tensor kernel: clip + AdamW + bf16 weight-shadow
    refresh in one pass, hipGraph-replayable with a per-iteration learning rate); ``fused_hip=False`` gives
    ``torch.optim.AdamW(fused=True)`` instead, with ``capturable`` forwarded (a torch optimizer captured into a
    hipGraph must be built with capturable=True, and TrainStep keeps its learning rate in a device tensor)."""
    name = config.TRAIN.OPTIMIZER.NAME.lower()
    groups = param_groups_no_decay_1d(model)
    on_gpu = all(p.is_cuda for g in groups for p in g['params'])
    if name == 'adamw':
        kw = dict(eps=config.TRAIN.OPTIMIZER.EPS, betas=tuple(config.TRAIN.OPTIMIZER.BETAS), lr=config.TRAIN.BASE_LR,
                  weight_decay=config.TRAIN.WEIGHT_DECAY)
        if on_gpu and (fused_hip is None or fused_hip):
            from .optim import FlatAdamW
            return FlatAdamW(groups, model=model, **kw)     # the model's bf16 weight shadows are refreshed by the update kernel
        if on_gpu:
            return torch.optim.AdamW(groups, fused=True, capturable=capturable, **kw)
        return torch.optim.AdamW(groups, **kw)
    if name == 'sgd':
        return torch.optim.SGD(groups, momentum=config.TRAIN.OPTIMIZER.MOMENTUM, nesterov=True, lr=config.TRAIN.BASE_LR,
                               weight_decay=config.TRAIN.WEIGHT_DECAY)
    raise ValueError(f'unknown optimizer {name}')


class NativeScalerWithGradNormCount:
    """Call-compatible with misc/utils.py:206-232.  bf16 needs no loss scaling, so ``scale`` is 1;
    the call still does backward -> (all-reduce) -> clip -> step and returns the gradient norm.

    With ``flat`` (a FlatGradients) the gradients live in one buffer that is all-reduced and clipped as a
    whole.  The reference's loop calls ``optimizer.zero_grad()`` after every update (misc/engine.py:231),
    which DETACHES ``p.grad`` from the buffer (set_to_none), so the views are re-attached (and zeroed)
    before every backward - otherwise the exchange and the clip would act on a stale buffer while the
    optimizer consumed un-reduced gradients."""
    state_dict_key = 'amp_scaler'

    def __init__(self, flat: FlatGradients | None = None):
        self.flat = flat

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        if self.flat is not None:
            self.flat.attach()
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if self.flat is not None:
            self.flat.attach()     # a backward that found grad=None would have allocated fresh tensors: fold them in
            self.flat.all_reduce_mean()
            norm = self.flat.clip_(clip_grad) if clip_grad is not None else torch.linalg.vector_norm(self.flat.flat)
        else:
            parameters = list(parameters)
            if clip_grad is not None:
                norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
            else:
                norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in parameters]))
        optimizer.step()
        return norm

    def state_dict(self):
        return {'scale': 1.0}

    def load_state_dict(self, state_dict):
        pass


# ---------------------------------------------------------------------------------------------
# the per-iteration body of train_one_epoch (misc/engine.py:202-231), hipGraph-replayable
# ---------------------------------------------------------------------------------------------
def _decoder_only_parameters(model):
    """Parameters whose gradient is complete once the decoder + head backward has run: everything except the
    encoder blocks and the tensors both image paths share (patch_embed.*, pos_embed get gradient from the
    encoder too, vision_transformer.py:379,391-392)."""
    out = []
    for name, p in model.named_parameters():
        if name.startswith(('cross_blocks.', 'norm.', 'head.')) or name == 'cls_token':
            out.append(p)
    return out


# Graph capture must not make OTHER threads' HIP calls illegal: RCCL's watchdog thread polls the events of finished collectives
# (hipEventQuery) whenever it likes, and under the default "global" capture mode such a query during a capture kills the process
# group ("operation not permitted when stream is capturing").  Only this thread's own calls are restricted.
_CAPTURE_MODE = 'thread_local'


class TrainStep:
    """forward (autocast) -> BCE-with-logits / accumulation_steps -> backward -> flat all-reduce -> clip 5.0 ->
    optimizer step -> lr_scheduler.step_update -> zero   (misc/engine.py:202-231).

    * ``accumulation_steps`` > 1: gradients accumulate in the flat buffer over that many calls and the
      exchange / clip / update happen on the last one (the reference all-reduces on every micro-step because
      it never uses ``no_sync()``; the mean of sums is the same number).
    * The exchange is split in two buckets: the decoder-only gradients are all-reduced while the encoder's
      backward still runs (``overlap=True``; needs a model with the reference's 3-way forward), the rest
      after it.  The backward is driven in two stages for that: decoder + head first, then the encoder from
      the gradient of the features.
    * ``use_graph=True`` replays hipGraphs (forward + decoder backward | encoder backward | update) captured
      after two eager warm-up steps, with the RCCL all-reduces issued between the replays, so the ~900
      launches of a step cost three graph launches on the host.  The learning rate lives in a device scalar,
      so ``lr_scheduler.step_update`` / ``set_lr`` take effect in the replayed update."""

    def __init__(self, model, optimizer, *, clip_grad=5.0, amp=True, criterion=None, use_graph=False,
                 compress_bf16=False, forward_fn=None, accumulation_steps=1, lr_scheduler=None, overlap=True, group=None,
                 start_update=0):
        self.model, self.optimizer, self.clip_grad, self.amp = model, optimizer, clip_grad, amp
        self.criterion = criterion or torch.nn.BCEWithLogitsLoss()
        self.accum = max(int(accumulation_steps), 1)
        self.lr_scheduler, self.group = lr_scheduler, group
        self.split = bool(overlap) and forward_fn is None and hasattr(model, 'cross_blocks')
        self.flat = FlatGradients(model.parameters(), compress_bf16=compress_bf16,
                                  early=_decoder_only_parameters(model) if self.split else None)
        if hasattr(model, 'direct_param_grads') or hasattr(model, 'runtime'):
            model.direct_param_grads = True     # HIP model: weight-gradient kernels add straight into the flat buffer
        self.forward_fn = forward_fn or (lambda m, x: m(x))
        self.use_graph = use_graph and torch.cuda.is_available()
        self.hip_opt = hasattr(optimizer, 'bind_flat')       # optim.FlatAdamW: clip + AdamW + shadow refresh in one kernel
        if self.hip_opt:
            optimizer.bind_flat(self.flat, model)
        elif self.use_graph:
            bad = [g for g in optimizer.param_groups if not g.get('capturable', False)]
            if bad:
                raise ValueError('TrainStep(use_graph=True) captures optimizer.step() into a hipGraph: build the torch optimizer with '
                                 'capturable=True (engine.build_optimizer(config, model, capturable=True)) or use optim.FlatAdamW')
        self._g1 = self._g2 = self._g_opt = None
        self._opt_signature = None
        self.recaptures = 0
        self._static_x = self._static_y = self._static_loss = self._static_norm = None
        self._eager_steps = 0
        self._micro = 0
        # updates done before this TrainStep existed: a resumed run passes epoch * num_steps // accumulation_steps so that the
        # per-iteration schedule continues where it stopped (misc/engine.py:228 counts from the start of training)
        self.num_updates = int(start_update)
        self.last_norm = None
        self._lr_tensors = None
        # a torch optimizer built with capturable=True reads its learning rate from a device scalar: keep it there in eager
        # mode too, so eager and replayed updates see the same (fp32) value
        self._tensor_lr = (not self.hip_opt) and all(g.get('capturable', False) for g in optimizer.param_groups) \
            and next(model.parameters()).is_cuda
        self.device_type = 'cuda' if next(model.parameters()).is_cuda else 'cpu'

    # -- learning rate -----------------------------------------------------------------------
    def set_lr(self, lr: float, group_index: int | None = None):
        """Per-iteration LR (misc/engine.py:228 ``lr_scheduler.step_update``): takes effect in eager and replayed updates."""
        for i, g in enumerate(self.optimizer.param_groups):
            if group_index is None or i == group_index:
                g['lr'] = float(lr) * g.get('lr_scale', 1.0)
        self._sync_lr()

    def _sync_lr(self):
        """Schedulers write Python floats into ``param_groups[i]['lr']``; a captured update reads a device scalar.
        Fold the floats into the per-group device tensors (torch optimizers: the tensors ARE ``group['lr']``)."""
        if self.hip_opt:
            self.optimizer.sync_hyperparameters()
            return
        if not self._tensor_lr:
            return
        if self._lr_tensors is None:
            dev = next(self.model.parameters()).device
            self._lr_tensors = [torch.tensor(float(g['lr']), dtype=torch.float32, device=dev) for g in self.optimizer.param_groups]
        for g, t in zip(self.optimizer.param_groups, self._lr_tensors):
            if g['lr'] is not t:
                t.fill_(float(g['lr']))
                g['lr'] = t

    # -- pieces ------------------------------------------------------------------------------
    def _loss(self, out, y):
        loss = self.criterion(out.float(), y)
        return loss / self.accum if self.accum > 1 else loss

    def _fwd_bwd(self, x, y):
        """One-stage form (any model / forward_fn)."""
        with torch.autocast(self.device_type, dtype=torch.bfloat16, enabled=self.amp):
            out = self.forward_fn(self.model, x)
            loss = self._loss(out, y)
        loss.backward()
        return loss.detach()

    def _fwd_dec_bwd(self, x, y):
        """Stage 1 of the split backward: encoder forward, decoder + head forward, loss, decoder backward.
        Returns (loss, features, d loss / d features)."""
        with torch.autocast(self.device_type, dtype=torch.bfloat16, enabled=self.amp):
            feats = self.model(x[:, 0], forward_first_part=True)
            leaf = feats.detach().requires_grad_(True)
            out = self.model(leaf, x[:, 1])
            loss = self._loss(out, y)
        loss.backward()
        return loss.detach(), feats, leaf.grad

    @staticmethod
    def _enc_bwd(feats, dfeats):
        feats.backward(dfeats)

    def _update(self):
        if self.hip_opt:
            norm = self.optimizer.step_flat(self.clip_grad)      # clip + AdamW + shadow refresh + zero: one pass
        else:
            norm = self.flat.clip_(self.clip_grad) if self.clip_grad is not None else torch.linalg.vector_norm(self.flat.flat)
            self.optimizer.step()
            self._refresh_shadows()       # the bf16 weight shadows follow the update (eval right after training sees them)
            self.flat.zero()
        return norm

    def _refresh_shadows(self):
        rts = getattr(self.model, '_runtimes', None)
        if rts:
            params = list(self.model.parameters())
            for rt in rts.values():
                rt.refresh_shadows(params)

    def _after_update(self):
        # misc/engine.py:228: lr_scheduler.step_update((epoch * num_steps + idx) // ACCUMULATION_STEPS) runs AFTER the update of
        # iteration idx with the count of updates done BEFORE it: 0, 1, 2, ...
        if self.lr_scheduler is not None:
            self.lr_scheduler.step_update(self.num_updates)
        self.num_updates += 1

    # -- public ------------------------------------------------------------------------------
    def step(self, x, y):
        """One call of the loop body.  Returns the (micro-batch) loss; ``last_norm`` holds the gradient norm of the
        last update."""
        last = (self._micro + 1) % self.accum == 0
        self._micro += 1
        if (self._g1 is not None and self.hip_opt and (self._micro - 1) % self.accum == 0
                and self.optimizer.shadow_signature() != self._opt_signature):
            # a bf16 weight shadow was re-created (or added) after capture: the captured forward / backward graphs read the old
            # buffers and the captured update refreshes the old set.  Drop every graph; this cycle runs eagerly, the next re-captures.
            torch.cuda.synchronize()
            self._g1 = self._g2 = self._g_opt = None
            self._eager_steps = 2 * self.accum - self.accum
            self.recaptures += 1
        if self.use_graph and self._g1 is None and self._eager_steps >= 2 * self.accum and (self._micro - 1) % self.accum == 0:
            self._capture(x, y)     # at the start of an accumulation cycle, after two eager updates: the flat buffer is zero
        if self._g1 is None:
            if self.use_graph:             # warm up allocator, workspaces and weight shadows eagerly
                self._eager_steps += 1
            return self._eager(x, y, last)
        self._static_x.copy_(x, non_blocking=True)
        self._static_y.copy_(y, non_blocking=True)
        buckets = self.flat.buckets()
        self._g1.replay()
        if self._g2 is not None:
            if last and len(buckets) == 2:
                self.flat.start_all_reduce(*buckets[0], group=self.group)   # runs under the encoder's backward
            self._g2.replay()
        if last:
            if self._g2 is not None and len(buckets) == 2:
                self.flat.start_all_reduce(*buckets[1], group=self.group)
            else:
                self.flat.start_all_reduce(0, self.flat.flat.numel(), group=self.group)
            self.flat.finish_all_reduce()
            self._sync_lr()
            self._g_opt.replay()
            self.last_norm = self._static_norm
            self._after_update()
        return self._static_loss

    def _eager(self, x, y, last):
        buckets = self.flat.buckets()
        if self.split and torch.is_tensor(x) and x.dim() == 5:
            loss, feats, dfeats = self._fwd_dec_bwd(x, y)
            if last and len(buckets) == 2:
                self.flat.start_all_reduce(*buckets[0], group=self.group)
            self._enc_bwd(feats, dfeats)
            if last:
                if len(buckets) == 2:
                    self.flat.start_all_reduce(*buckets[1], group=self.group)
                else:
                    self.flat.start_all_reduce(0, self.flat.flat.numel(), group=self.group)
        else:
            loss = self._fwd_bwd(x, y)
            if last:
                self.flat.start_all_reduce(0, self.flat.flat.numel(), group=self.group)
        if last:
            self.flat.finish_all_reduce()
            self._sync_lr()
            self.last_norm = self._update()
            self._after_update()
        return loss

    def _capture(self, x, y):
        from . import ops
        if not torch.is_tensor(x):
            raise TypeError('TrainStep(use_graph=True) replays on a static input tensor: pass use_graph=False for structured batches')
        self._static_x, self._static_y = x.clone(), y.clone()
        self._sync_lr()
        torch.cuda.synchronize()
        ops.pin_workspace()               # captured kernels bake buffer addresses in: later growth must not free them
        for rt in getattr(self.model, '_runtimes', {}).values():
            rt.pinned = True
        split = self.split and torch.is_tensor(x) and x.dim() == 5
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1, capture_error_mode=_CAPTURE_MODE):
            if split:
                self._static_loss, self._feats, self._dfeats = self._fwd_dec_bwd(self._static_x, self._static_y)
            else:
                self._static_loss = self._fwd_bwd(self._static_x, self._static_y)
        if split:
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g2, pool=self._g1.pool(), capture_error_mode=_CAPTURE_MODE):
                self._enc_bwd(self._feats, self._dfeats)
        self._capture_update()

    def _capture_update(self):
        """(Re-)capture the update graph.  Its kernels bake in the optimizer's descriptor table, i.e. the set of weight-shadow
        buffers to refresh; ``step`` compares that set before every replay."""
        torch.cuda.synchronize()
        if self.hip_opt:
            self.optimizer._descriptors()          # build the table outside the capture
            self._opt_signature = self.optimizer.shadow_signature()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._g1.pool(), capture_error_mode=_CAPTURE_MODE):
            norm = self._update()
        if self._static_norm is None:
            self._static_norm = norm
        elif norm.data_ptr() != self._static_norm.data_ptr():
            self._static_norm = norm
        self._g_opt = g


# ---------------------------------------------------------------------------------------------
# input pipeline: host -> HBM one batch ahead (misc/engine.py:202-204)
# ---------------------------------------------------------------------------------------------
class DevicePrefetcher:
    """Wraps a loader of (samples, targets) CPU batches.  The reference copies every batch on the compute stream at the top of
    the iteration (``samples.cuda(non_blocking=True)``, misc/engine.py:203-204) as fp32.  Here the batch is staged in pinned host
    memory and copied on a side stream ``depth`` batches ahead of the step that consumes it, and uint8 images stay uint8 all the
    way into the patch-embedding kernel (``vited_patchify_u8`` applies ToTensor + Normalize), so a config-A batch of 1024 pairs
    is 25 MB on PCIe instead of 101 MB.  Yields device tensors; iteration order and contents equal the wrapped loader's."""

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(int(depth), 1)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._pinned = {}

    def __len__(self):
        return len(self.loader)

    def _stage(self, t, slot, name):
        """CPU tensor -> device tensor through a reusable pinned buffer (per ring slot), on the copy stream."""
        if not torch.is_tensor(t):
            return t
        if self.stream is None:
            return t.to(self.device)
        key = (slot, name, tuple(t.shape), t.dtype)
        ent = self._pinned.get(key)
        if ent is None:
            ent = self._pinned[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory(), None]
        buf, ev = ent
        if ev is not None:
            ev.synchronize()               # the previous H2D copy out of this pinned slot must have executed before it is rewritten
        buf.copy_(t)
        out = buf.to(self.device, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record(self.stream)
        return out

    def __iter__(self):
        import collections
        queue = collections.deque()
        slot = 0
        for batch in self.loader:
            samples, targets = batch
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    item = (self._stage(samples, slot, 'x'), self._stage(targets, slot, 'y'))
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
            else:
                item, ev = (self._stage(samples, slot, 'x'), self._stage(targets, slot, 'y')), None
            queue.append((item, ev))
            slot = (slot + 1) % (self.depth + 1)          # a pinned buffer is rewritten only after its batch was handed out
            if len(queue) > self.depth:
                yield self._hand_out(*queue.popleft())
        while queue:
            yield self._hand_out(*queue.popleft())

    def _hand_out(self, item, ev):
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                              # the consumer's stream waits for the copy, the host does not
            for t in item:
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return item


# ---------------------------------------------------------------------------------------------
# patch-pair assembly on the device (data/datasets/div2k_patch.py:108-162)
# ---------------------------------------------------------------------------------------------
def div2k_pair_plan(u: torch.Tensor, img_size: int, erosion_ratio: float, with_negative: bool = True, train: bool = True):
    """The random choices of ``DIV2KPatch.__getitem__`` (div2k_patch.py:114-153) for a whole batch at once, from uniform numbers
    ``u`` [B, 4] in [0, 1) (columns: negative-pair draw, first swap, second swap, erosion):
      cells  int32 [B, 2]   grid cells (3 columns x 2 rows, row-major) of image 1 and image 2
      labels fp32  [B, 4]   the 4-bin target (all-zero for the 30 % negatives)
      erode  int32 [B]      eroded cell size e = ceil(S (1 - r)), r ~ U(erosion_ratio, 2 erosion_ratio) in training
    first = cell 0, second = 1 (right of it), third = 4 (below second), fourth = 3 (below first), spare = 2."""
    dev = u.device
    a, b = u[:, 1] > 0.5, u[:, 2] > 0.5
    neg = (u[:, 0] < 0.3) if with_negative else torch.zeros_like(a)
    second = torch.where(neg, torch.where(a, 4, 2), torch.where(a, 3, 1))      # negative: third / spare; positive: fourth / second
    first = torch.zeros_like(second)
    img1 = torch.where(b, second, first)
    img2 = torch.where(b, first, second)
    cells = torch.stack([img1, img2], dim=1).to(torch.int32)
    bin_ = a.long() + 2 * b.long()                                              # (a, b) -> label bin 0, 1, 2, 3
    labels = torch.nn.functional.one_hot(bin_, 4).float() * (~neg).float().unsqueeze(1)
    r = erosion_ratio * (1.0 + u[:, 3].double()) if train else torch.full_like(u[:, 3], erosion_ratio, dtype=torch.float64)
    erode = torch.ceil(img_size * (1.0 - r)).to(torch.int32).clamp_(1, img_size)
    return cells.contiguous(), labels.to(dev), erode.contiguous()


def assemble_pairs(regions_u8: torch.Tensor, cells: torch.Tensor, erode: torch.Tensor, img_size: int) -> torch.Tensor:
    """uint8 regions [B, C, 2 S, 3 S] on the device -> uint8 pairs [B, 2, C, S, S]: erosion crop + Pillow-exact bilinear resize of
    the two chosen cells in one kernel (``vited_crop_pairs_u8``).  Feed the result straight to the model: ToTensor + Normalize
    are folded into the patch-embedding kernel."""
    from . import ops
    return ops.crop_pairs_u8(regions_u8, cells, erode, img_size)


# ---------------------------------------------------------------------------------------------
# pair mining for the two-stage HisFrag training step (hisfrag.py:117-159)
# ---------------------------------------------------------------------------------------------
def mine_pairs(targets: torch.Tensor, neg_per_pos: float = 2.0, generator=None):
    """(groups int64 [P, 2], labels fp32 [P, 1]): every same-label pair (i, j), j > i, in row-major order, then a
    random subset of the different-label pairs of size min(#neg, int(neg_per_pos * #pos)) - what
    ``HisfragTrainer.prepare_data`` (hisfrag.py:117-145) builds with a Python loop over the batch and 2n
    ``nonzero`` host syncs.  Here: one ``triu_indices`` + two boolean selects on the device (the pair count is
    data-dependent, so one host sync per step remains)."""
    t = targets.reshape(-1)
    n = t.numel()
    i, j = torch.triu_indices(n, n, offset=1, device=t.device)
    same = t[i] == t[j]
    pos = torch.stack([i[same], j[same]], dim=1)
    neg = torch.stack([i[~same], j[~same]], dim=1)
    keep = min(neg.shape[0], int(neg_per_pos * pos.shape[0]))
    perm = torch.randperm(neg.shape[0], generator=generator, device=neg.device if generator is None else generator.device)[:keep]
    neg = neg[perm.to(neg.device)]
    groups = torch.cat([pos, neg], dim=0)
    labels = torch.cat([torch.ones(pos.shape[0], device=t.device), torch.zeros(neg.shape[0], device=t.device)]).view(-1, 1)
    return groups, labels


def hisfrag_prepare_data(model, samples: torch.Tensor, targets: torch.Tensor, amp: bool = True, generator=None):
    """The first half of the reference's two-stage step (hisfrag.py:117-155): mine pairs, run the encoder ONCE per
    image, gather.  Returns ((x, x1_feats), labels) for ``model(x1_feats, x)`` exactly like the reference's
    ``prepare_data`` -> ``train_step`` hand-off (hisfrag.py:153-159)."""
    groups, labels = mine_pairs(targets, generator=generator)
    with torch.autocast(samples.device.type, dtype=torch.bfloat16, enabled=amp):
        feats = model(samples, forward_first_part=True)
    return (samples[groups[:, 0]], feats[groups[:, 1]]), labels


# ---------------------------------------------------------------------------------------------
# pairwise similarity-matrix inference (hisfrag.py:161-302, BASELINE config 5)
# ---------------------------------------------------------------------------------------------
def shard_rows_by_pair_count(n: int, world: int):
    """Contiguous row blocks of the upper triangle (i <= j) with ~equal PAIR counts per rank - the
    balancing rule of data/samplers.py:108-137 in closed form.  Returns world+1 row boundaries."""
    total = n * (n + 1) // 2
    bounds, acc, r = [0], 0, 1
    for i in range(n):
        acc += n - i
        while r < world and acc >= total * r / world:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world + 1:
        bounds.append(n)
    bounds[-1] = n
    return bounds


def _row_block_pairs(a0, a1, c0, c1, dev):
    """Pairs (i, j), i in [a0, a1), j in [c0, c1), j >= i, row-major - the order every rank and the assembler agree on."""
    ii = torch.arange(a0, a1, device=dev).view(-1, 1).expand(a1 - a0, c1 - c0)
    jj = torch.arange(c0, c1, device=dev).view(1, -1).expand(a1 - a0, c1 - c0)
    keep = jj >= ii
    return ii[keep], jj[keep]


class _ImageSource:
    """Images by index range for the streamed similarity run: a tensor [n, C, S, S] (host or device, uint8 or float) or a
    callable ``(lo, hi) -> tensor`` with ``n_images`` (the reference re-opens its dataset with ``lower_bound`` per row block,
    hisfrag.py:201-211).  Host blocks travel through ``DevicePrefetcher`` (pinned staging, side stream, uint8 stays uint8)."""

    def __init__(self, images, n_images, dev):
        self.images, self.dev = images, dev
        if torch.is_tensor(images):
            self.n = images.shape[0]
        else:
            if n_images is None:
                raise ValueError('pairwise_similarity: a callable image source needs n_images')
            self.n = int(n_images)

    def host_block(self, lo, hi):
        return self.images[lo:hi] if torch.is_tensor(self.images) else self.images(lo, hi)

    def block(self, lo, hi):
        t = self.host_block(lo, hi)
        return t if t.device == self.dev else t.to(self.dev, non_blocking=True)

    def column_blocks(self, start, step):
        """(c0, c1, images on the device) for c0 = start, start + step, ...; host blocks are copied one block ahead."""
        spans = [(c0, min(c0 + step, self.n)) for c0 in range(start, self.n, step)]
        if not spans:
            return
        probe = self.host_block(*spans[0])
        if probe.device == self.dev or self.dev.type != 'cuda':
            for i, (c0, c1) in enumerate(spans):
                yield c0, c1, (probe if i == 0 else self.host_block(c0, c1)).to(self.dev)
            return
        loader = ((probe if i == 0 else self.host_block(c0, c1), torch.tensor([c0, c1])) for i, (c0, c1) in enumerate(spans))
        for (imgs, _), (c0, c1) in zip(DevicePrefetcher(loader, self.dev, depth=1), spans):
            yield c0, c1, imgs


def _similarity_scores_streamed(model, src, r0, r1, *, block, col_block, pair_batch, amp, state_path, meta, after_row_block):
    """This rank's score vector (pairs of rows [r0, r1) in _row_block_pairs order, row block by row block, column block by
    column block) with O(block + col_block) images resident: per row block the encoder output and the cross-attention K / V of
    ``block`` images; per column block the image-2 token cache of ``col_block`` images.  Finished row blocks are saved to
    ``state_path`` (what hisfrag.py:181-195,243-246 does with ``*_result_rank{r}.pt``) and skipped on a restart."""
    n, dev = src.n, src.dev
    total = sum(n - i for i in range(r0, r1))
    scores = torch.empty(total, dtype=torch.float32, device=dev)
    done_rows, off = r0, 0
    if state_path is not None and os.path.exists(state_path):
        st = torch.load(state_path, map_location='cpu', weights_only=True)
        if st.get('meta') == meta and r0 <= int(st['done_rows']) <= r1:
            done_rows = int(st['done_rows'])
            off = int(st['scores'].numel())
            scores[:off] = st['scores'].to(dev)
    dtype_ctx = lambda: torch.autocast(dev.type, dtype=torch.bfloat16, enabled=amp)
    for a0 in range(r0, r1, block):
        a1 = min(a0 + block, r1)
        if a1 <= done_rows:
            continue                                                        # finished before the restart
        with dtype_ctx():
            feats = model(src.block(a0, a1), forward_first_part=True)       # encoder once per row block
            kvs = model.cache_context_kv(feats)                             # K / V of every decoder block, once per row block
        del feats
        for c0, c1, imgs2 in src.column_blocks(a0, col_block):
            with dtype_ctx():
                tokens2, q0 = model.cache_image2_tokens(imgs2)              # everything that depends on image 2 alone
            ii, jj = _row_block_pairs(a0, a1, c0, c1, dev)
            for p0 in range(0, ii.numel(), pair_batch):
                with dtype_ctx():
                    out = model.forward_pairs_cached(tokens2, jj[p0:p0 + pair_batch] - c0, kvs, ii[p0:p0 + pair_batch] - a0, q0)
                cnt = out.numel()
                scores[off: off + cnt] = out.float().reshape(-1)
                off += cnt
            del tokens2, q0, imgs2
        del kvs
        if state_path is not None:
            tmp = state_path + '.tmp'
            torch.save({'meta': meta, 'done_rows': a1, 'scores': scores[:off].cpu(), 'is_finished': a1 == r1}, tmp)
            os.replace(tmp, state_path)                                     # a kill between blocks never leaves a torn file
        if after_row_block is not None:
            after_row_block(a0, a1)
    assert off == total, (off, total)
    return scores


@torch.no_grad()
def pairwise_similarity(model, images, *, rank: int = 0, world: int = 1, block: int = 64, pair_batch: int = 512,
                        amp: bool = True, group=None, pair_cache: bool = True, col_block: int = 256, n_images=None,
                        state_path=None, after_row_block=None):
    """similarity[i, j] = similarity[j, i] = fp16(logit(model(features(image_i), image_j))) for i <= j.

    What hisfrag.py:161-302 computes, re-plumbed: the encoder runs ONCE per image of this rank's row
    block, the decoder runs on `pair_batch` pairs at a time, and the ranks exchange their score vectors with ONE
    all-gather (RCCL on GPU) instead of the reference's per-rank files + 120 s polling.  Every rank returns the full
    symmetric [n, n] fp16 matrix of raw logits (callers take 1 - similarity as the distance, hisfrag.py:294-296).

    With the HIP model (``supports_pair_cache``) the run STREAMS: ``images`` may be a host tensor (uint8 or float) or a
    callable ``(lo, hi) -> tensor`` (+ ``n_images``), only ``block`` row images and ``col_block`` column images are
    resident at a time, scores land in one pre-sized buffer, and with ``state_path`` finished row blocks are saved and a
    restarted run skips them (hisfrag.py:181-195,243-246).  Models without the cache (the CPU oracle in the tests) take the
    plain path on a resident image tensor."""
    cached = bool(getattr(model, 'supports_pair_cache', False)) and pair_cache
    was_training = model.training
    model.eval()
    if cached:
        dev = next(model.parameters()).device
        src = _ImageSource(images, n_images, dev)
        n = src.n
        bounds = shard_rows_by_pair_count(n, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        meta = {'n': n, 'r0': r0, 'r1': r1, 'block': block, 'col_block': col_block}
        mine = _similarity_scores_streamed(model, src, r0, r1, block=block, col_block=col_block, pair_batch=pair_batch, amp=amp,
                                           state_path=state_path, meta=meta, after_row_block=after_row_block)
        enumerate_pairs = lambda lo, hi: [(_row_block_pairs(a0, min(a0 + block, hi), c0, min(c0 + col_block, n), dev))
                                          for a0 in range(lo, hi, block) for c0 in range(a0, n, col_block)]
    else:
        n = images.shape[0]
        dev = images.device
        bounds = shard_rows_by_pair_count(n, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        dtype_ctx = torch.autocast(dev.type, dtype=torch.bfloat16, enabled=amp)
        by_index = bool(getattr(model, 'supports_x2_index', False))
        scores = []
        for a0 in range(r0, r1, block):
            a1 = min(a0 + block, r1)
            with dtype_ctx:
                feats = model(images[a0:a1], forward_first_part=True)          # encoder once per row block
            ii, jj = torch.triu_indices(a1 - a0, n - a0, offset=0, device=dev)  # pairs (a0+ii, a0+jj), jj >= ii
            for c0 in range(0, ii.numel(), pair_batch):
                i_sub, j_sub = ii[c0:c0 + pair_batch], (jj[c0:c0 + pair_batch] + a0)
                with dtype_ctx:
                    out = model(feats[i_sub], images, x2_index=j_sub) if by_index else model(feats[i_sub], images[j_sub])
                scores.append(out.float().reshape(-1))
        mine = torch.cat(scores) if scores else torch.zeros(0, device=dev)

        def enumerate_pairs(lo, hi):
            out = []
            for a0 in range(lo, hi, block):
                a1 = min(a0 + block, hi)
                ii, jj = torch.triu_indices(a1 - a0, n - a0, offset=0, device=dev)
                out.append((ii + a0, jj + a0))
            return out
    model.train(was_training)

    # exchange: pad to the largest shard, one all-gather, then every rank rebuilds the matrix
    counts = [sum(n - i for i in range(bounds[r], bounds[r + 1])) for r in range(world)]
    if world > 1:
        pad = torch.zeros(max(counts), dtype=torch.float32, device=dev)
        pad[:mine.numel()] = mine
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad, group=group)
    else:
        gathered = [mine]
    sim = torch.zeros((n, n), dtype=torch.float16, device=dev)
    for r in range(world):
        off = 0
        for ii, jj in enumerate_pairs(bounds[r], bounds[r + 1]):      # the same enumeration order as the compute loop
            vals = gathered[r][off: off + ii.numel()].to(torch.float16)
            sim[ii, jj] = vals
            sim[jj, ii] = vals
            off += ii.numel()
        assert off == counts[r], (r, off, counts[r])
    return sim
