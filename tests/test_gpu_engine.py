"""The driven step on the GPU: fused clip + AdamW kernel against torch.optim.AdamW + clip_grad_norm_ (misc/optimizer.py:25-46,
misc/utils.py:215-223), hipGraph replay against eager launches with a per-iteration learning rate and gradient accumulation
(misc/engine.py:212-231), bf16 weight shadows after an update, and two data-parallel ranks driving the HIP model.

Tolerances: the AdamW kernel evaluates the same expression as torch's fused kernel in a different association, so parameters
and moments agree to fp32 rounding (rtol 2e-6 per step on values that are not catastrophically cancelled: atol tied to lr);
graph replay vs eager launches of the SAME kernels must be bit-identical.
"""
import os
import types

import pytest
import torch
import torch.multiprocessing as mp

from oracle import vited_oracle as vo

pytestmark = pytest.mark.gpu
bce = torch.nn.functional.binary_cross_entropy_with_logits


def _hip_model(vited, s, gpu, dtype):
    m = vited.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, in_chans=s.in_chans, num_classes=s.num_classes,
                                      embed_dim=s.embed_dim, depth=s.depth, c_depth=s.c_depth, num_heads=s.num_heads)
    m.compute_dtype = dtype
    return m.to(gpu)


def test_flat_adamw_matches_torch_adamw_and_refreshes_shadows(vited, gpu):
    torch.manual_seed(0)
    shapes = [(384, 1152), (4, 384), (384, 3, 8, 8), (384,), (65, 130), (1536, 384), (1,), (1, 1, 384), (100, 36)]
    mine = [torch.nn.Parameter(torch.randn(sh, device=gpu) * 0.1) for sh in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]

    def groups(ps):
        return [{'params': [p for p in ps if p.ndim > 1]}, {'params': [p for p in ps if p.ndim <= 1], 'weight_decay': 0.0}]

    kw = dict(lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.05)
    opt = vited.optim.FlatAdamW(groups(mine), **kw)
    topt = torch.optim.AdamW(groups(ref), **kw)
    flat = vited.engine.FlatGradients(mine, early=[mine[1], mine[3]])       # any layout of the flat buffer
    rt = vited.functions.Runtime(img_size=64, patch_size=8, in_chans=3, num_classes=4, embed_dim=384, depth=1, c_depth=1, num_heads=12)
    shadow_n = {i: rt.weight(mine[i]) for i in (0, 2, 5, 8)}
    shadow_t = {i: rt.weight_t(mine[i])[0] for i in (0, 4, 5)}
    opt.bind_flat(flat, types.SimpleNamespace(_runtimes={torch.bfloat16: rt}))
    for it in range(4):
        for p, q in zip(mine, ref):
            g = torch.randn_like(p) * (3.0 if it == 1 else 0.01)       # step 1: the clip is active
            p.grad.copy_(g)
            q.grad = g.clone()
        lr = 2e-3 / (1 + it)                                            # per-iteration schedule
        for o in (opt, topt):
            for grp in o.param_groups:
                grp['lr'] = lr
        want_norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        norm = opt.step_flat(1.0)
        torch.testing.assert_close(norm, want_norm, rtol=1e-5, atol=1e-7)
        assert float(flat.flat.abs().max()) == 0.0                      # zero_grad fused in
        for i, (p, q) in enumerate(zip(mine, ref)):
            torch.testing.assert_close(p, q, rtol=2e-6, atol=1e-3 * lr, msg=lambda m: f'step {it} param {i} {tuple(p.shape)}: {m}')
            torch.testing.assert_close(opt.state[p]['exp_avg'], topt.state[q]['exp_avg'], rtol=1e-5, atol=1e-9)
            torch.testing.assert_close(opt.state[p]['exp_avg_sq'], topt.state[q]['exp_avg_sq'], rtol=1e-5, atol=1e-12)
        for i, sh in shadow_n.items():
            assert torch.equal(sh, mine[i].detach().reshape(mine[i].shape[0], -1).to(torch.bfloat16)), f'shadow of param {i}'
            assert rt.weight(mine[i]) is sh                             # the cache entry stays valid: no recast on the next forward
        for i, sh in shadow_t.items():
            assert torch.equal(sh, mine[i].detach().reshape(mine[i].shape[0], -1).t().contiguous().to(torch.bfloat16)), f'shadow_t of param {i}'
    assert opt.num_updates == 4
    sd = opt.state_dict()                                               # torch.optim.AdamW-shaped state (misc/utils.py:130-142)
    assert set(sd['state'][0].keys()) >= {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 4.0


class _Sched:
    """Stand-in for the reference's per-iteration scheduler (misc/lr_scheduler.py via misc/engine.py:228)."""

    def __init__(self, opt, base):
        self.opt, self.base = opt, base

    def step_update(self, n):
        for g in self.opt.param_groups:
            g['lr'] = self.base / (1.0 + 0.5 * n)


@pytest.mark.parametrize('optimizer', ['flat_hip', 'torch_capturable'])
@pytest.mark.parametrize('amp,accum', [(False, 1), (True, 1), (False, 2)])
def test_graph_replay_equals_eager_with_lr_schedule_and_accumulation(vited, gpu, optimizer, amp, accum):
    """TrainStep(use_graph=True) replays three hipGraphs per step.  With a learning rate that changes after every update
    and (accum = 2) two micro-batches per update it must produce, bit for bit, the parameters of the eagerly launched
    TrainStep - so the replayed update really reads the scheduler's value and the accumulation really accumulates.  bf16
    (amp) adds the weight shadows: they are refreshed inside the update, so an eval forward right after training equals
    the forward of a freshly loaded checkpoint."""
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(3)
    init = _hip_model(vited, s, gpu, None).state_dict()
    models, steps = [], []
    for use_graph in (False, True):
        m = _hip_model(vited, s, gpu, None)
        m.load_state_dict(init)
        groups = vited.engine.param_groups_no_decay_1d(m)
        kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
        opt = vited.optim.FlatAdamW(groups, **kw) if optimizer == 'flat_hip' else torch.optim.AdamW(groups, fused=True, capturable=True, **kw)
        steps.append(vited.engine.TrainStep(m, opt, clip_grad=5.0, amp=amp, use_graph=use_graph, accumulation_steps=accum,
                                            lr_scheduler=_Sched(opt, 1e-3)))
        models.append(m)
    g = torch.Generator().manual_seed(9)
    for it in range(7 * accum):
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1).to(gpu)
        y = (torch.rand(8, 4, generator=g) > 0.6).float().to(gpu)
        le, lg = [float(st.step(x, y)) for st in steps]
        assert le == lg, (it, le, lg)
    assert steps[1]._g1 is not None and steps[1]._g2 is not None and steps[0].num_updates == steps[1].num_updates == 7
    assert steps[1].recaptures == 0          # the shadow set is stable after the eager warm-up: the update graph was captured once
    assert float(steps[0].last_norm) == float(steps[1].last_norm)
    for (n, pe), (_, pg) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert torch.equal(pe, pg), f'{n}: graph replay differs from eager launches'
    # the schedule took effect: a frozen lr = 1e-3 run lands elsewhere
    assert not torch.equal(models[0].head.weight, init['head.weight'])
    x = torch.randn(4, 2, 3, 64, 64, generator=g).clamp(-1, 1).to(gpu)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        after = [m.eval()(x) for m in models]
        fresh = _hip_model(vited, s, gpu, None)
        fresh.load_state_dict(models[1].state_dict())
        want = fresh.eval()(x)
    assert torch.equal(after[0], want) and torch.equal(after[1], want), 'stale bf16 weight shadows after the last update'


def test_graph_survives_workspace_growth_after_capture(vited, gpu):
    """A larger eager op after capture grows the shared workspace; the captured kernels keep their (retired, still alive)
    buffer, so replays stay correct."""
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(5)
    models = [_hip_model(vited, s, gpu, torch.float32) for _ in range(2)]
    models[1].load_state_dict(models[0].state_dict())
    steps = [vited.engine.TrainStep(m, vited.optim.FlatAdamW(vited.engine.param_groups_no_decay_1d(m), lr=1e-3), amp=False, use_graph=ug)
             for m, ug in zip(models, (False, True))]
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        x = torch.randn(4, 2, 3, 64, 64, generator=g).clamp(-1, 1).to(gpu)
        y = (torch.rand(4, 4, generator=g) > 0.6).float().to(gpu)
        if it == 4:
            before = vited.ops.workspace(4, gpu).numel()
            big = torch.randn(4096, 3000, device=gpu)
            vited.ops.sum_rows(big)                                    # batch 4096 x width 3000: grows the workspace
            grown = vited.ops.workspace(4, gpu).numel()
            if grown > before:
                scribble = torch.full((grown,), float('nan'), device=gpu)   # whoever owns recycled memory writes to it
                del scribble
        assert float(steps[0].step(x, y)) == float(steps[1].step(x, y))
    for (n, pe), (_, pg) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert torch.equal(pe, pg), n


# ---------------------------------------------------------------------------------------------
# two data-parallel ranks drive the HIP model (gloo moves the CUDA tensors; RCCL needs one GPU per rank)
# ---------------------------------------------------------------------------------------------
def _rank(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import vited_amd as V
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', init_method='env://', world_size=world, rank=rank)
    dev = torch.device('cuda:0')
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(100 + rank)
    m = V.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=s.num_classes, embed_dim=s.embed_dim,
                                  depth=1, c_depth=1, num_heads=s.num_heads).to(dev)
    m.compute_dtype = torch.float32
    V.engine.broadcast_parameters(m)
    opt = V.optim.FlatAdamW(V.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05)
    step = V.engine.TrainStep(m, opt, clip_grad=5.0, amp=False, use_graph=True)
    assert len(step.flat.buckets()) == 2
    g = torch.Generator().manual_seed(7)
    for it in range(4):                         # 2 eager + capture + replays: the exchange sits between the graphs
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
        y = (torch.rand(8, 4, generator=g) > 0.6).float()
        step.step(x[rank::world].to(dev), y[rank::world].to(dev))
    params = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    gathered = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    if rank == 0:
        torch.save({'same': all(torch.equal(t, gathered[0]) for t in gathered), 'state': {k: v.cpu() for k, v in m.state_dict().items()}}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process(vited, gpu, tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_rank, args=(2, 29700 + os.getpid() % 200, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res['same']
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(100)
    m = _hip_model(vited, s, gpu, torch.float32)
    opt = vited.optim.FlatAdamW(vited.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05)
    step = vited.engine.TrainStep(m, opt, clip_grad=5.0, amp=False)
    g = torch.Generator().manual_seed(7)
    for it in range(4):
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
        y = (torch.rand(8, 4, generator=g) > 0.6).float()
        step.step(x.to(gpu), y.to(gpu))
    for k, v in m.state_dict().items():
        # mean of the two shard gradients == full-batch gradient up to fp32 summation order; AdamW amplifies that on ~0 gradients
        torch.testing.assert_close(res['state'][k], v.cpu(), rtol=1e-3, atol=2e-4, msg=lambda msg: f'{k}: {msg}')


# ---------------------------------------------------------------------------------------------
# RCCL itself on the one GPU there is: a one-rank "nccl" group with the collectives forced on, so the calls the 8-GPU run
# will make (ReduceOp.AVG on slices of the flat fp32 buffer, async work handles joined before the update graph, the
# flat broadcast) run through RCCL between the hipGraph replays.  The mean over one rank is the identity: results must be
# bit-identical to the same steps without a process group.
# ---------------------------------------------------------------------------------------------
def _rccl_one_rank(rank, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import vited_amd as V
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    s = vo.ViTEDShape(depth=1, c_depth=1)

    def run(distributed):
        torch.manual_seed(5)
        m = V.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=s.num_classes, embed_dim=s.embed_dim,
                                      depth=1, c_depth=1, num_heads=s.num_heads).to(dev)
        m.compute_dtype = torch.bfloat16
        if distributed:
            V.engine.broadcast_parameters(m)
        opt = V.optim.FlatAdamW(V.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05)
        step = V.engine.TrainStep(m, opt, clip_grad=5.0, amp=True, use_graph=True)
        g = torch.Generator().manual_seed(7)
        losses = []
        for it in range(5):
            x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
            y = (torch.rand(8, 4, generator=g) > 0.6).float()
            losses.append(float(step.step(x.to(dev), y.to(dev))))
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu(), losses

    ref, ref_losses = run(False)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', VITED_FORCE_COLLECTIVE='1')
    dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0, device_id=dev)
    calls = {'n': 0}
    orig = dist.all_reduce

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    dist.all_reduce = counting
    got, losses = run(True)
    dist.all_reduce = orig
    torch.cuda.synchronize()
    dist.destroy_process_group()
    torch.save({'equal': bool(torch.equal(ref, got)), 'losses': losses == ref_losses, 'calls': calls['n']}, out)


def test_rccl_one_rank_collectives_between_graph_replays(vited, gpu, tmp_path):
    out = str(tmp_path / 'rccl.pt')
    mp.spawn(_rccl_one_rank, args=(29900 + os.getpid() % 90, out), nprocs=1, join=True)
    res = torch.load(out, weights_only=True)
    assert res['calls'] == 10           # two buckets x five steps went through RCCL
    assert res['equal'] and res['losses']


# ---------------------------------------------------------------------------------------------
# the reference's own wrap (misc/engine.py:75): torch DistributedDataParallel around the HIP model, driven by the reference's
# loop order (autocast forward, scaler(loss, optimizer, clip_grad, parameters), optimizer.zero_grad() - misc/engine.py:208-231)
# ---------------------------------------------------------------------------------------------
def _ddp_rank(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import vited_amd as V
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', init_method='env://', world_size=world, rank=rank)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(300 + rank)                     # DDP's ctor broadcasts rank 0's parameters
    m = V.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=s.num_classes, embed_dim=s.embed_dim,
                                  depth=1, c_depth=1, num_heads=s.num_heads).to(dev)
    m.compute_dtype = torch.float32
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False)
    opt = torch.optim.AdamW(V.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05)
    scaler = V.engine.NativeScalerWithGradNormCount()
    g = torch.Generator().manual_seed(11)
    for it in range(3):
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
        y = (torch.rand(8, 4, generator=g) > 0.6).float()
        loss = bce(ddp(x[rank::world].to(dev)).float(), y[rank::world].to(dev))
        scaler(loss, opt, clip_grad=5.0, parameters=ddp.parameters())
        opt.zero_grad()
    if rank == 0:
        torch.save({k: v.cpu() for k, v in m.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_hip_model_under_torch_ddp_wrap_matches_single_process(vited, gpu, tmp_path):
    out = str(tmp_path / 'ddp.pt')
    mp.spawn(_ddp_rank, args=(2, 29500 + os.getpid() % 150, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(300)
    m = _hip_model(vited, s, gpu, torch.float32)
    opt = torch.optim.AdamW(vited.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05)
    scaler = vited.engine.NativeScalerWithGradNormCount()
    g = torch.Generator().manual_seed(11)
    for it in range(3):
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
        y = (torch.rand(8, 4, generator=g) > 0.6).float()
        loss = bce(m(x.to(gpu)).float(), y.to(gpu))
        scaler(loss, opt, clip_grad=5.0, parameters=m.parameters())
        opt.zero_grad()
    for k, v in m.state_dict().items():
        # DDP averages the two shard gradients: the full-batch gradient up to fp32 summation order (AdamW amplifies that on ~0 gradients)
        torch.testing.assert_close(got[k], v.cpu(), rtol=1e-3, atol=2e-4, msg=lambda msg: f'{k}: {msg}')


# ---------------------------------------------------------------------------------------------
# similarity-matrix inference sharded over two ranks WITH the HIP model and the pair cache (BASELINE config 5): rank 1's row
# block starts at r0 > 0, so its image-2 token cache starts at image r0 and is indexed j - r0 (engine.pairwise_similarity)
# ---------------------------------------------------------------------------------------------
_SIM_SHAPE = dict(img_size=256, patch_size=16, num_classes=1, num_heads=6, depth=1, c_depth=2)
_SIM_N = 11


def _sim_rank(rank, world, port, out_dir, state_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import vited_amd as V
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', init_method='env://', world_size=world, rank=rank)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    s = vo.ViTEDShape(**_SIM_SHAPE)
    m = V.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=s.num_classes, embed_dim=s.embed_dim,
                                  depth=s.depth, c_depth=s.c_depth, num_heads=s.num_heads).to(dev).eval()
    m.load_state_dict(torch.load(state_path, weights_only=True))
    imgs = torch.randn(_SIM_N, 3, s.img_size, s.img_size, generator=torch.Generator().manual_seed(17)).clamp(-1, 1).to(dev)
    bounds = V.engine.shard_rows_by_pair_count(_SIM_N, world)
    res = {'bounds': bounds}
    for name, dtype in (('f32', torch.float32), ('bf16', torch.bfloat16)):
        m.compute_dtype = dtype
        assert m.supports_pair_cache
        sim = V.engine.pairwise_similarity(m, imgs, rank=rank, world=world, block=3, pair_batch=16, amp=dtype == torch.bfloat16)
        res[name] = sim.cpu()
    torch.save(res, os.path.join(out_dir, f'sim_{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_pair_cached_similarity_matches_the_oracle(vited, gpu, tmp_path):
    s = vo.ViTEDShape(**_SIM_SHAPE)
    torch.manual_seed(31)
    oracle = vo.OracleViTED(s).eval()
    state = str(tmp_path / 'state.pt')
    torch.save(oracle.state_dict(), state)
    mp.spawn(_sim_rank, args=(2, 29300 + os.getpid() % 150, str(tmp_path), state), nprocs=2, join=True)
    r0, r1 = (torch.load(str(tmp_path / f'sim_{r}.pt'), weights_only=True) for r in (0, 1))
    bounds = r0['bounds']
    assert 0 < bounds[1] < _SIM_N, f'rank 1 must own a non-empty row block that does not start at image 0: {bounds}'
    imgs = torch.randn(_SIM_N, 3, s.img_size, s.img_size, generator=torch.Generator().manual_seed(17)).clamp(-1, 1)
    i, j = torch.triu_indices(_SIM_N, _SIM_N)
    with torch.no_grad():
        ref = oracle(torch.stack([imgs[i], imgs[j]], dim=1)).reshape(-1)       # the naive one-shot pairs (hisfrag.py:226-229)
    mine1 = i >= bounds[1]                                                      # pairs computed BY RANK 1 (rows r0..n)
    assert int(mine1.sum()) > 0 and int((~mine1).sum()) > 0
    for name, tol in (('f32', dict(rtol=2e-3, atol=2e-3)), ('bf16', dict(rtol=3e-2, atol=3e-2))):
        assert torch.equal(r0[name], r1[name]), 'both ranks must rebuild the same matrix from the all-gather'
        sim = r0[name]
        assert torch.equal(sim, sim.t())
        torch.testing.assert_close(sim[i, j][mine1].float(), ref[mine1], msg=lambda m: f'{name}, rank 1 block: {m}', **tol)
        torch.testing.assert_close(sim[i, j][~mine1].float(), ref[~mine1], msg=lambda m: f'{name}, rank 0 block: {m}', **tol)


# ---------------------------------------------------------------------------------------------
# the reference's own loop around engine.build_optimizer()'s default optimizer
# ---------------------------------------------------------------------------------------------
def _cfg_a(vited, **train):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = vited.config_from_yaml(os.path.join(here, 'configs', 'puzzle', 'div2k_erosion7_4bin_patch8_64.yaml'))
    return cfg


def test_build_optimizer_default_under_the_reference_loop_keeps_bf16_shadows_fresh(vited, gpu):
    """``engine.build_optimizer`` hands out ``FlatAdamW`` on a GPU model.  Driven the way the reference drives ANY optimizer
    (misc/engine.py:208-231: autocast forward, ``loss_scaler(loss, optimizer, clip_grad, parameters)`` - which calls the plain
    ``optimizer.step()`` - then ``optimizer.zero_grad()``) the update kernel raw-writes the fp32 parameters, so the bf16 weight
    shadows the MFMA kernels read must follow: the losses must move like a torch.optim.AdamW run of the same loop, and an eval
    forward after training must equal the forward of a freshly built model loaded from the checkpoint, bit for bit."""
    s = vo.ViTEDShape(depth=1, c_depth=1)
    cfg = _cfg_a(vited)
    torch.manual_seed(41)
    init = _hip_model(vited, s, gpu, None).state_dict()
    runs = {}
    for kind in ('flat_hip', 'torch'):
        m = _hip_model(vited, s, gpu, None)
        m.load_state_dict(init)
        opt = vited.engine.build_optimizer(cfg, m, fused_hip=(kind == 'flat_hip'))
        assert isinstance(opt, vited.optim.FlatAdamW) == (kind == 'flat_hip')
        for g_ in opt.param_groups:
            g_['lr'] = 1e-3
        scaler = vited.engine.NativeScalerWithGradNormCount()        # no flat buffer: the reference's call shape
        g = torch.Generator().manual_seed(13)
        losses = []
        opt.zero_grad()
        for it in range(4):
            x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1).to(gpu)
            y = (torch.rand(8, 4, generator=g) > 0.6).float().to(gpu)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss = bce(m(x).float(), y)
            scaler(loss, opt, clip_grad=5.0, parameters=m.parameters())
            opt.zero_grad()
            losses.append(float(loss))
        runs[kind] = (m, losses)
    # the same loop on the CPU oracle (fp32): with stale shadows every forward after the first runs on the INITIAL weights, and the
    # losses stay near ln 2 instead of following the (large, lr = 1e-3) first AdamW steps.  This caught torch.optim.AdamW(fused=True)
    # too: it updates parameters without bumping their version counters (functions._install_optimizer_step_hook).
    oracle = vo.OracleViTED(s)
    oracle.load_state_dict({k: v.cpu() for k, v in init.items()})
    oopt = torch.optim.AdamW(vited.engine.param_groups_no_decay_1d(oracle), lr=1e-3, betas=tuple(cfg.TRAIN.OPTIMIZER.BETAS),
                             eps=cfg.TRAIN.OPTIMIZER.EPS, weight_decay=cfg.TRAIN.WEIGHT_DECAY)
    g = torch.Generator().manual_seed(13)
    want_losses = []
    for it in range(4):
        x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1)
        y = (torch.rand(8, 4, generator=g) > 0.6).float()
        loss = bce(oracle(x), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
        oopt.step()
        oopt.zero_grad()
        want_losses.append(float(loss))
    assert max(want_losses) > 1.0, 'the case must move the loss visibly after the first update'
    m, losses = runs['flat_hip']
    for kind in ('flat_hip', 'torch'):
        for a, b in zip(runs[kind][1], want_losses):
            assert abs(a - b) < 3e-2 * max(1.0, abs(b)), (kind, runs[kind][1], want_losses)
    # (parameters are not compared one by one: with lr = 1e-3 AdamW moves every element by ~1e-3 per step in the direction of
    # the gradient's SIGN, which bf16 noise flips on near-zero gradients; the losses above are the functional check)
    x = torch.randn(4, 2, 3, 64, 64, generator=torch.Generator().manual_seed(2)).clamp(-1, 1).to(gpu)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        after = m.eval()(x)
        fresh = _hip_model(vited, s, gpu, None)
        fresh.load_state_dict(m.state_dict())
        want = fresh.eval()(x)
    assert torch.equal(after, want), 'stale bf16 weight shadows: the forward did not see the updated parameters'
    # ... and with an optimizer that was NOT given the model (shadows unknown to the kernel) the version bump makes the model recast
    m2 = _hip_model(vited, s, gpu, None)
    m2.load_state_dict(init)
    opt2 = vited.optim.FlatAdamW(vited.engine.param_groups_no_decay_1d(m2), lr=1e-3, weight_decay=0.05)
    x8 = torch.randn(8, 2, 3, 64, 64, generator=torch.Generator().manual_seed(3)).clamp(-1, 1).to(gpu)
    y8 = (torch.rand(8, 4, generator=torch.Generator().manual_seed(4)) > 0.6).float().to(gpu)
    for it in range(2):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            bce(m2(x8).float(), y8).backward()
        opt2.step()
        opt2.zero_grad()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        fresh2 = _hip_model(vited, s, gpu, None)
        fresh2.load_state_dict(m2.state_dict())
        assert torch.equal(m2.eval()(x), fresh2.eval()(x))


def test_flat_adamw_resume_before_bind_keeps_the_step_count(vited, gpu):
    """The reference resumes as: build optimizer -> load_checkpoint (optimizer.load_state_dict) -> build the loop
    (misc/utils.py:57-70).  ``FlatAdamW.load_state_dict`` BEFORE ``bind_flat`` / ``TrainStep`` must carry the AdamW step
    count (bias correction) as well as the moments: the first resumed update equals torch.optim.AdamW's."""
    torch.manual_seed(0)
    shapes = [(384, 384), (384,), (4, 384)]
    base = [torch.randn(sh, device=gpu) * 0.1 for sh in shapes]

    def groups(ps):
        return [{'params': [p for p in ps if p.ndim > 1]}, {'params': [p for p in ps if p.ndim <= 1], 'weight_decay': 0.0}]

    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ref = [torch.nn.Parameter(b.clone()) for b in base]
    topt = torch.optim.AdamW(groups(ref), **kw)
    gen = torch.Generator(device=gpu).manual_seed(5)
    grads = [[torch.randn(sh, device=gpu, generator=gen) * 0.01 for sh in shapes] for _ in range(6)]
    for it in range(5):
        for p, g in zip(ref, grads[it]):
            p.grad = g.clone()
        topt.step()
    import copy
    ckpt = {'opt': copy.deepcopy(topt.state_dict()), 'params': [p.detach().clone() for p in ref]}    # state_dict() returns live references
    # resume: fresh parameters + FlatAdamW, load BEFORE anything binds a flat buffer
    mine = [torch.nn.Parameter(p.clone()) for p in ckpt['params']]
    opt = vited.optim.FlatAdamW(groups(mine), **kw)
    opt.load_state_dict(ckpt['opt'])
    assert opt.num_updates == 5 and float(opt.state_dict()['state'][0]['step']) == 5.0      # unbound: the loaded step survives a re-save
    flat = vited.engine.FlatGradients(mine)
    opt.bind_flat(flat)
    assert opt.num_updates == 5
    for p, q, g in zip(mine, ref, grads[5]):
        p.grad.copy_(g)
        q.grad = g.clone()
    topt.step()
    opt.step_flat(None)
    for i, (p, q) in enumerate(zip(mine, ref)):
        torch.testing.assert_close(p, q, rtol=2e-6, atol=1e-6, msg=lambda m: f'param {i}: {m}')
    assert opt.num_updates == 6
    # load AFTER binding copies into the existing buffers (a captured update graph keeps reading them)
    ptr = opt.exp_avg.data_ptr()
    opt.load_state_dict(ckpt['opt'])
    assert opt.exp_avg.data_ptr() == ptr and opt.num_updates == 5
    torch.testing.assert_close(opt.state[mine[0]]['exp_avg'], ckpt['opt']['state'][0]['exp_avg'])


def test_graphs_are_recaptured_when_the_shadow_set_changes(vited, gpu):
    """The captured graphs bake in WHICH bf16 weight-shadow buffers the forward / backward read and the update kernel refreshes.
    If a shadow is re-created after capture (here: two cache entries replaced by hand, as a parameter whose storage moved would
    cause), replaying the old graphs would train on stale weights.  TrainStep compares the shadow set at the start of every
    accumulation cycle, drops the graphs, runs that cycle eagerly and captures again: training continues bit-identically to the
    eagerly launched twin."""
    s = vo.ViTEDShape(depth=1, c_depth=1)
    torch.manual_seed(9)
    init = _hip_model(vited, s, gpu, None).state_dict()
    models, steps = [], []
    for use_graph in (False, True):
        m = _hip_model(vited, s, gpu, None)
        m.load_state_dict(init)
        opt = vited.optim.FlatAdamW(vited.engine.param_groups_no_decay_1d(m), lr=1e-3, weight_decay=0.05, model=m)
        steps.append(vited.engine.TrainStep(m, opt, clip_grad=5.0, amp=True, use_graph=use_graph))
        models.append(m)
    g = torch.Generator().manual_seed(4)

    def run(n):
        for _ in range(n):
            x = torch.randn(8, 2, 3, 64, 64, generator=g).clamp(-1, 1).to(gpu)
            y = (torch.rand(8, 4, generator=g) > 0.6).float().to(gpu)
            le, lg = [float(st.step(x, y)) for st in steps]
            assert le == lg

    run(4)
    assert steps[1]._g_opt is not None and steps[1].recaptures == 0
    rt = models[1].runtime(torch.bfloat16)
    w = models[1].blocks[0].mlp.fc1.weight
    before = {k: v[2].data_ptr() for k, v in rt._shadow.items() if k[0] == id(w)}
    assert set(t for _, t in before) == {'n', 't'}
    for k in before:
        old = rt._shadow.pop(k)
        rt._retired.append(old[2])
    rt.weight(w), rt.weight_t(w)
    after = {k: v[2].data_ptr() for k, v in rt._shadow.items() if k[0] == id(w)}
    assert all(after[k] != before[k] for k in before)
    run(1)
    assert steps[1].recaptures == 1 and steps[1]._g1 is None, 'the changed shadow set must drop the graphs (this step ran eagerly)'
    # the update refreshed the NEW buffers: they hold the current weights
    assert torch.equal(rt._shadow[(id(w), 'n')][2], w.detach().to(torch.bfloat16))
    assert torch.equal(rt._shadow[(id(w), 't')][2], w.detach().t().contiguous().to(torch.bfloat16))
    run(3)
    assert steps[1].recaptures == 1 and steps[1]._g1 is not None, 'graphs are captured again and replayed'
    for (n, pe), (_, pg) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert torch.equal(pe, pg), f'{n}: graph path differs from eager launches after the re-capture'
