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
class FlatGradients:
    """All parameter gradients as views of ONE contiguous fp32 buffer.

    ``p.grad`` is pre-set to a view, so autograd accumulates in place and the buffer is always the
    gradient; ``zero()`` replaces ``optimizer.zero_grad()``; ``all_reduce_mean()`` is the data-parallel
    exchange: one RCCL all-reduce of the whole buffer (133 MB fp32 at config A - SURVEY.md 2.3 C1 -
    instead of six 25 MB buckets), optionally bf16-compressed on the wire."""

    def __init__(self, params, compress_bf16: bool = False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.wire = torch.empty(total, dtype=torch.bfloat16, device=dev) if compress_bf16 else None
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off: off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:          # someone called zero_grad(set_to_none=True): re-attach
                p.grad = v

    def all_reduce_mean(self, group=None):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world == 1:
            return
        if self.wire is not None:
            self.wire.copy_(self.flat)
            dist.all_reduce(self.wire, op=dist.ReduceOp.SUM, group=group)
            self.flat.copy_(self.wire)
            self.flat.mul_(1.0 / world)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / world)

    def clip_(self, max_norm: float):
        """clip_grad_norm_ on the flat buffer: one norm kernel instead of 280 (misc/utils.py:215-217)."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


def broadcast_parameters(model, src: int = 0, group=None):
    """DDP ctor semantics: every rank starts from rank 0's parameters."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in model.parameters():
        dist.broadcast(p.data, src=src, group=group)


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


def build_optimizer(config, model):
    name = config.TRAIN.OPTIMIZER.NAME.lower()
    groups = param_groups_no_decay_1d(model)
    fused = all(p.is_cuda for g in groups for p in g['params'])
    if name == 'adamw':
        return torch.optim.AdamW(groups, eps=config.TRAIN.OPTIMIZER.EPS, betas=tuple(config.TRAIN.OPTIMIZER.BETAS),
                                 lr=config.TRAIN.BASE_LR, weight_decay=config.TRAIN.WEIGHT_DECAY, fused=fused)
    if name == 'sgd':
        return torch.optim.SGD(groups, momentum=config.TRAIN.OPTIMIZER.MOMENTUM, nesterov=True, lr=config.TRAIN.BASE_LR,
                               weight_decay=config.TRAIN.WEIGHT_DECAY)
    raise ValueError(f'unknown optimizer {name}')


class NativeScalerWithGradNormCount:
    """Call-compatible with misc/utils.py:206-232.  bf16 needs no loss scaling, so ``scale`` is 1;
    the call still does backward -> (all-reduce) -> clip -> step and returns the gradient norm."""
    state_dict_key = 'amp_scaler'

    def __init__(self, flat: FlatGradients | None = None):
        self.flat = flat

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if self.flat is not None:
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
class TrainStep:
    """forward (autocast) -> BCE-with-logits -> backward -> flat all-reduce -> clip 5.0 -> AdamW -> zero.

    ``use_graph=True`` captures forward+backward (and, separately, clip+optimizer) into hipGraphs
    after one eager warm-up step, with the RCCL all-reduce issued eagerly between the two replays,
    so the ~900 launches of a step cost two graph launches on the host."""

    def __init__(self, model, optimizer, *, clip_grad=5.0, amp=True, criterion=None, use_graph=False,
                 compress_bf16=False, forward_fn=None):
        self.model, self.optimizer, self.clip_grad, self.amp = model, optimizer, clip_grad, amp
        self.criterion = criterion or torch.nn.BCEWithLogitsLoss()
        self.flat = FlatGradients(model.parameters(), compress_bf16=compress_bf16)
        if hasattr(model, 'direct_param_grads') or hasattr(model, 'runtime'):
            model.direct_param_grads = True     # HIP model: weight-gradient kernels add straight into the flat buffer
        self.forward_fn = forward_fn or (lambda m, x: m(x))
        self.use_graph = use_graph and torch.cuda.is_available()
        self._g_fb = self._g_opt = None
        self._static_x = self._static_y = self._static_loss = self._static_norm = None
        self._eager_steps = 0
        self.device_type = 'cuda' if next(model.parameters()).is_cuda else 'cpu'

    # -- pieces ------------------------------------------------------------------------------
    def _fwd_bwd(self, x, y):
        with torch.autocast(self.device_type, dtype=torch.bfloat16, enabled=self.amp):
            out = self.forward_fn(self.model, x)
            loss = self.criterion(out.float(), y)
        loss.backward()
        return loss.detach()

    def _update(self):
        norm = self.flat.clip_(self.clip_grad) if self.clip_grad is not None else torch.linalg.vector_norm(self.flat.flat)
        self.optimizer.step()
        return norm

    def _refresh_shadows(self):
        rts = getattr(self.model, '_runtimes', None)
        if rts:
            params = list(self.model.parameters())
            for rt in rts.values():
                rt.refresh_shadows(params)

    # -- public ------------------------------------------------------------------------------
    def step(self, x, y):
        if not self.use_graph:
            self.flat.zero()
            loss = self._fwd_bwd(x, y)
            self.flat.all_reduce_mean()
            self.last_norm = self._update()
            return loss
        if self._g_fb is None:
            if self._eager_steps < 2:      # warm up allocator, workspaces and weight shadows eagerly
                self._eager_steps += 1
                self.flat.zero()
                loss = self._fwd_bwd(x, y)
                self.flat.all_reduce_mean()
                self.last_norm = self._update()
                return loss
            self._capture(x, y)
        self._static_x.copy_(x, non_blocking=True)
        self._static_y.copy_(y, non_blocking=True)
        self._g_fb.replay()
        self.flat.all_reduce_mean()
        self._g_opt.replay()
        self.last_norm = self._static_norm
        return self._static_loss

    def _capture(self, x, y):
        self._static_x, self._static_y = x.clone(), y.clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._refresh_shadows()
        torch.cuda.current_stream().wait_stream(side)
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._refresh_shadows()           # weights changed since the last replay: recast in place
            self.flat.zero()
            self._static_loss = self._fwd_bwd(self._static_x, self._static_y)
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, pool=self._g_fb.pool()):
            self._static_norm = self._update()


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


@torch.no_grad()
def pairwise_similarity(model, images, *, rank: int = 0, world: int = 1, block: int = 64, pair_batch: int = 512,
                        amp: bool = True, group=None):
    """similarity[i, j] = similarity[j, i] = fp16(logit(model(features(image_i), image_j))) for i <= j.

    What hisfrag.py:161-302 computes, re-plumbed: the encoder runs ONCE per image of this rank's row
    block, the decoder runs on `pair_batch` pairs at a time, image-2 gathers happen inside the
    patch-embed kernel (index array, no materialised [P,3,S,S] copy - SURVEY section 7 last bullet), and
    the ranks exchange their score vectors with ONE all-gather (RCCL on GPU) instead of the reference's
    per-rank files + 120 s polling.  Every rank returns the full symmetric [n, n] fp16 matrix of raw
    logits (callers take 1 - similarity as the distance, hisfrag.py:294-296)."""
    n = images.shape[0]
    dev = images.device
    bounds = shard_rows_by_pair_count(n, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    dtype_ctx = torch.autocast(dev.type, dtype=torch.bfloat16, enabled=amp)
    by_index = bool(getattr(model, 'supports_x2_index', False))
    scores = []
    was_training = model.training
    model.eval()
    for a0 in range(r0, r1, block):
        a1 = min(a0 + block, r1)
        with dtype_ctx:
            feats = model(images[a0:a1], forward_first_part=True)          # encoder once per row block
        ii, jj = torch.triu_indices(a1 - a0, n - a0, offset=0, device=dev)  # pairs (a0+ii, a0+jj), jj >= ii
        for c0 in range(0, ii.numel(), pair_batch):
            i_sub, j_sub = ii[c0:c0 + pair_batch], (jj[c0:c0 + pair_batch] + a0)
            with dtype_ctx:
                if by_index:
                    out = model(feats[i_sub], images, x2_index=j_sub)
                else:
                    out = model(feats[i_sub], images[j_sub])
            scores.append(out.float().reshape(-1))
    mine = torch.cat(scores) if scores else torch.zeros(0, device=dev)
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
        rows = torch.arange(bounds[r], bounds[r + 1], device=dev)
        if rows.numel() == 0:
            continue
        # same enumeration order as the compute loop: row blocks of `block`, triu inside each block
        off = 0
        for a0 in range(bounds[r], bounds[r + 1], block):
            a1 = min(a0 + block, bounds[r + 1])
            ii, jj = torch.triu_indices(a1 - a0, n - a0, offset=0, device=dev)
            vals = gathered[r][off: off + ii.numel()].to(torch.float16)
            sim[ii + a0, jj + a0] = vals
            sim[jj + a0, ii + a0] = vals
            off += ii.numel()
    return sim
