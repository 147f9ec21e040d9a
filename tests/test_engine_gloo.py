"""CPU, 2 processes over gloo: the flat-gradient data-parallel plumbing (the RCCL path on the GPU
box uses the same code with backend 'nccl').  The CPU oracle stands in for the model - the engine
is model-agnostic."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vited_oracle as vo

SHAPE = vo.ViTEDShape(img_size=32, patch_size=8, embed_dim=64, num_heads=2, depth=1, c_depth=1, num_classes=4)


def _data(n):
    g = torch.Generator().manual_seed(123)
    x = torch.randn(n, 2, 3, 32, 32, generator=g).clamp(-1, 1)
    y = (torch.rand(n, 4, generator=g) > 0.75).float()
    return x, y


def _named_grads(model, flat):
    name_of = {id(p): n for n, p in model.named_parameters()}
    return {name_of[id(p)]: v.clone() for p, v in zip(flat.params, flat.views)}


def _worker(rank, world, port, compress, out, overlap=True, accum=1):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vited_amd
    from vited_amd import engine
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    lr, r, w = engine.configure_ddp()
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    torch.manual_seed(100 + rank)                       # different init per rank ...
    model = vo.OracleViTED(SHAPE)
    engine.broadcast_parameters(model)                  # ... until rank 0's parameters are broadcast (DDP ctor semantics)
    opt = torch.optim.AdamW(engine.param_groups_no_decay_1d(model), lr=1e-3, weight_decay=0.05)
    step = engine.TrainStep(model, opt, clip_grad=5.0, amp=False, compress_bf16=compress, overlap=overlap, accumulation_steps=accum)
    assert len(step.flat.buckets()) == (2 if overlap else 1)
    x, y = _data(8 * accum)
    shard = slice(rank, None, world)                    # strided shards, data/samplers.py:50
    step.flat.zero()
    step._fwd_bwd(x[shard], y[shard])
    step.flat.all_reduce_mean()
    grads = _named_grads(model, step.flat)
    if accum > 1:
        grads = {k: v * accum for k, v in grads.items()}    # _fwd_bwd divides the loss by accumulation_steps
    step.flat.zero()
    xs, ys = x[shard], y[shard]
    per = xs.shape[0] // accum
    for a in range(accum):                              # full step: every rank must end with identical parameters
        loss = step.step(xs[a * per:(a + 1) * per], ys[a * per:(a + 1) * per])
    assert step.num_updates == 1 and float(step.flat.flat.abs().max()) == 0.0
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    if rank == 0:
        torch.save({'grads': grads, 'same_params': all(torch.equal(g, gathered[0]) for g in gathered), 'loss': float(loss),
                    'state': model.state_dict()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('compress', [False, True])
def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path, compress):
    out = str(tmp_path / 'r0.pt')
    port = 29600 + (os.getpid() % 300) + (50 if compress else 0)
    mp.spawn(_worker, args=(2, port, compress, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res['same_params']
    # reference: one process, whole batch, same initial parameters (rank 0's, seed 100)
    torch.manual_seed(100)
    model = vo.OracleViTED(SHAPE)
    x, y = _data(8)
    torch.nn.functional.binary_cross_entropy_with_logits(model(x), y).backward()
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    tol = dict(rtol=2e-2, atol=2e-3 * gmax) if compress else dict(rtol=1e-4, atol=1e-6)
    for n, p in model.named_parameters():               # mean of shard gradients == full-batch gradient
        torch.testing.assert_close(res['grads'][n], p.grad, msg=lambda m: f'{n}: {m}', **tol)


def _reference_step(accum=1):
    """One process, whole batch, rank 0's initial parameters: the parameters after one clip + AdamW update."""
    torch.manual_seed(100)
    model = vo.OracleViTED(SHAPE)
    from vited_amd import engine
    opt = torch.optim.AdamW(engine.param_groups_no_decay_1d(model), lr=1e-3, weight_decay=0.05)
    x, y = _data(8 * accum)
    torch.nn.functional.binary_cross_entropy_with_logits(model(x), y).backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
    return model.state_dict()


@pytest.mark.parametrize('overlap,accum', [(True, 1), (False, 1), (True, 2)])
def test_two_rank_step_matches_single_process(tmp_path, overlap, accum):
    """The bucketed exchange (decoder-only gradients all-reduced under the encoder's backward, the rest after it),
    the single-bucket one and gradient accumulation all land on the parameters one process gets on the whole batch
    (DDP semantics, misc/engine.py:75,202-231)."""
    out = str(tmp_path / 'r0.pt')
    port = 29300 + (os.getpid() % 200) + (7 if overlap else 0) + 13 * accum
    mp.spawn(_worker, args=(2, port, False, out, overlap, accum), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res['same_params']
    want = _reference_step(accum)
    # AdamW's first update is lr * sign-like(g): on mathematically-zero gradients (the key bias under softmax) it amplifies fp32
    # summation-order noise to a fraction of lr = 1e-3, hence the absolute tolerance of 5 % of one update
    for k, v in want.items():
        torch.testing.assert_close(res['state'][k], v, rtol=2e-4, atol=5e-5, msg=lambda m: f'{k}: {m}')


def test_flat_gradients_single_process_semantics():
    import vited_amd
    from vited_amd import engine
    torch.manual_seed(0)
    model = vo.OracleViTED(SHAPE)
    flat = engine.FlatGradients(model.parameters())
    n_params = sum(p.numel() for p in model.parameters())
    # every view starts on a 64-byte boundary (the [1] head bias of config H must not push its successors onto odd words)
    assert n_params <= flat.flat.numel() < n_params + 16 * len(flat.params) and all(o % 16 == 0 for o in flat.offsets)
    x, y = _data(4)
    torch.nn.functional.binary_cross_entropy_with_logits(model(x), y).backward()
    gather = lambda: torch.cat([flat.flat[o: o + p.numel()] for p, o in zip(flat.params, flat.offsets)])
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.equal(ref, gather()) and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat.views))
    assert int(torch.count_nonzero(flat.flat)) == int(torch.count_nonzero(gather()))          # the padding words stay zero
    assert flat.buckets() == [(0, flat.flat.numel())]
    torch.nn.functional.binary_cross_entropy_with_logits(model(x), y).backward()          # accumulation keeps the views
    torch.testing.assert_close(gather(), 2 * ref)
    norm = flat.clip_(0.5 * float(torch.linalg.vector_norm(flat.flat)))
    torch.testing.assert_close(norm, torch.linalg.vector_norm(2 * ref))
    torch.testing.assert_close(torch.linalg.vector_norm(flat.flat), 0.5 * norm, rtol=1e-4, atol=1e-6)
    model.zero_grad(set_to_none=True)                     # the reference's optimizer.zero_grad() default
    flat.zero()
    assert all(p.grad is v for p, v in zip(flat.params, flat.views)) and float(flat.flat.abs().max()) == 0


def test_flat_scaler_survives_the_reference_loop_order():
    """misc/engine.py:217-231: loss_scaler(...) then optimizer.zero_grad() - which detaches p.grad from the flat buffer
    (set_to_none).  The flat scaler must keep clipping / stepping on the LIVE gradients: three steps against plain
    clip_grad_norm_ + AdamW on a twin model, including one accumulation step (update_grad=False)."""
    import vited_amd
    from vited_amd import engine
    torch.manual_seed(0)
    a, b = vo.OracleViTED(SHAPE), vo.OracleViTED(SHAPE)
    b.load_state_dict(a.state_dict())
    oa = torch.optim.AdamW(engine.param_groups_no_decay_1d(a), lr=1e-3, weight_decay=0.05)
    ob = torch.optim.AdamW(engine.param_groups_no_decay_1d(b), lr=1e-3, weight_decay=0.05)
    flat = engine.FlatGradients(a.parameters(), early=engine._decoder_only_parameters(a))
    scaler = engine.NativeScalerWithGradNormCount(flat)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    for it in range(4):
        x, y = _data(4)
        x = x + 0.1 * it
        update = it != 1                          # iteration 1 only accumulates
        na = scaler(bce(a(x), y), oa, clip_grad=0.05, parameters=a.parameters(), update_grad=update)
        bce(b(x), y).backward()
        if update:
            nb = torch.nn.utils.clip_grad_norm_(b.parameters(), 0.05)
            ob.step()
            ob.zero_grad()
            oa.zero_grad()                        # set_to_none=True: detaches every p.grad from the flat buffer
            torch.testing.assert_close(na, nb, rtol=1e-4, atol=1e-6)
            assert float(nb) > 0.05              # the clip is active, so an un-clipped step would show
        for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            torch.testing.assert_close(p, q, rtol=1e-4, atol=5e-5, msg=lambda m: f'step {it} {n}: {m}')


def test_train_step_accumulation_and_lr_schedule_cpu():
    """TrainStep(accumulation_steps=2, lr_scheduler=...) == the reference loop's arithmetic: loss / 2 per micro-step,
    one update per two calls, lr_scheduler.step_update(updates done before this one) after each update (misc/engine.py:212-231)."""
    import vited_amd
    from vited_amd import engine
    torch.manual_seed(0)
    a, b = vo.OracleViTED(SHAPE), vo.OracleViTED(SHAPE)
    b.load_state_dict(a.state_dict())
    oa = torch.optim.AdamW(engine.param_groups_no_decay_1d(a), lr=1e-3, weight_decay=0.05)
    ob = torch.optim.AdamW(engine.param_groups_no_decay_1d(b), lr=1e-3, weight_decay=0.05)

    class Sched:
        def __init__(self, opt):
            self.opt, self.seen = opt, []

        def step_update(self, n):
            self.seen.append(n)
            for g in self.opt.param_groups:
                g['lr'] = 1e-3 / (1 + n)

    sa, sb = Sched(oa), Sched(ob)
    step = engine.TrainStep(a, oa, clip_grad=5.0, amp=False, accumulation_steps=2, lr_scheduler=sa)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    for it in range(6):
        x, y = _data(4)
        x = x - 0.05 * it
        step.step(x, y)
        (bce(b(x), y) / 2).backward()
        if it % 2 == 1:
            torch.nn.utils.clip_grad_norm_(b.parameters(), 5.0)
            ob.step()
            ob.zero_grad()
            sb.step_update(it // 2)               # misc/engine.py:228: (epoch * num_steps + idx) // ACCUMULATION_STEPS = 0, 1, 2
    assert sa.seen == [0, 1, 2] and step.num_updates == 3
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=5e-5, msg=lambda m: f'{n}: {m}')
    # a resumed run continues the schedule: start_update = epoch * num_steps // accumulation_steps
    resumed = engine.TrainStep(a, oa, clip_grad=5.0, amp=False, accumulation_steps=2, lr_scheduler=sa, start_update=40)
    for it in range(2):
        resumed.step(*_data(4))
    assert sa.seen[-1] == 40 and resumed.num_updates == 41


def test_scaler_call_shape_matches_reference():
    """misc/engine.py:217-223 calls loss_scaler(loss, optimizer, clip_grad=..., parameters=..., update_grad=...)."""
    import vited_amd
    from vited_amd import engine
    torch.manual_seed(0)
    model = vo.OracleViTED(SHAPE)
    opt = torch.optim.AdamW(engine.param_groups_no_decay_1d(model), lr=1e-3)
    scaler = engine.NativeScalerWithGradNormCount()
    x, y = _data(4)
    before = model.head.weight.detach().clone()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(model(x), y)
    assert scaler(loss, opt, clip_grad=5.0, parameters=model.parameters(), update_grad=False) is None
    assert torch.equal(before, model.head.weight)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(model(x), y)
    norm = scaler(loss, opt, clip_grad=5.0, parameters=model.parameters(), update_grad=True)
    assert float(norm) > 0 and not torch.equal(before, model.head.weight) and scaler.state_dict()['scale'] == 1.0


# ---------------------------------------------------------------------------------------------
# pairwise similarity-matrix inference (BASELINE config 5; reference invariant: sharded == unsharded,
# two-stage == one-shot, tests/hisfrag_evaluation_test.py:18-99,143)
# ---------------------------------------------------------------------------------------------
SIM_SHAPE = vo.ViTEDShape(img_size=32, patch_size=8, embed_dim=64, num_heads=2, depth=1, c_depth=1, num_classes=1)


def _sim_images(n):
    g = torch.Generator().manual_seed(7)
    return torch.randn(n, 3, 32, 32, generator=g).clamp(-1, 1)


def _naive_similarity(model, imgs):
    n = imgs.shape[0]
    i, j = torch.triu_indices(n, n)
    with torch.no_grad():
        logits = model(torch.stack([imgs[i], imgs[j]], dim=1)).reshape(-1)       # one-shot forward on stacked pairs
    sim = torch.zeros(n, n, dtype=torch.float16)
    sim[i, j] = logits.to(torch.float16)
    sim[j, i] = logits.to(torch.float16)
    return sim


def test_row_sharding_balances_pairs():
    from vited_amd import engine
    for n, world in ((1, 1), (10, 3), (37, 8), (512, 8), (5, 8)):
        b = engine.shard_rows_by_pair_count(n, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
        counts = [sum(n - i for i in range(b[r], b[r + 1])) for r in range(world)]
        assert sum(counts) == n * (n + 1) // 2
        if n >= 4 * world:
            assert max(counts) <= 1.5 * (sum(counts) / world) + n


def _sim_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vited_amd
    from vited_amd import engine
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    engine.configure_ddp()
    torch.manual_seed(3)
    model = vo.OracleViTED(SIM_SHAPE)
    sim = engine.pairwise_similarity(model, _sim_images(11), rank=rank, world=world, block=4, pair_batch=7, amp=False)
    if rank == 1:
        torch.save(sim, out)
    dist.barrier()
    dist.destroy_process_group()


def test_pairwise_similarity_sharded_equals_naive(tmp_path):
    from vited_amd import engine
    torch.manual_seed(3)
    model = vo.OracleViTED(SIM_SHAPE)
    imgs = _sim_images(11)
    ref = _naive_similarity(model, imgs)
    one = engine.pairwise_similarity(model, imgs, block=4, pair_batch=7, amp=False)
    torch.testing.assert_close(one.float(), ref.float(), rtol=2e-3, atol=2e-3)
    assert torch.equal(one, one.t())
    out = str(tmp_path / 'sim.pt')
    mp.spawn(_sim_worker, args=(2, 29900 + os.getpid() % 90, out), nprocs=2, join=True)
    two = torch.load(out, weights_only=True)
    assert torch.equal(two, one)            # 2 ranks + all-gather give bit-identical scores to 1 rank


def test_mine_pairs_matches_the_reference_loop():
    """engine.mine_pairs vs a literal restatement of hisfrag.py:117-145 (loop over i, nonzero per row): same positives
    in the same order, negatives a subset of the reference's candidates with the reference's count."""
    import torch
    import vited_amd
    eng = vited_amd.engine
    g = torch.Generator().manual_seed(5)
    for n, classes in ((24, 8), (7, 2), (5, 5), (6, 1)):
        targets = torch.randint(0, classes, (n,), generator=g)
        pos_ref, neg_ref = [], []
        for i in range(n):
            for j in range(i + 1, n):
                (pos_ref if targets[i] == targets[j] else neg_ref).append((i, j))
        groups, labels = eng.mine_pairs(targets, generator=torch.Generator().manual_seed(1))
        npos = len(pos_ref)
        nneg = min(len(neg_ref), 2 * npos)
        assert groups.shape == (npos + nneg, 2) and labels.shape == (npos + nneg, 1)
        assert [tuple(r) for r in groups[:npos].tolist()] == pos_ref
        got_neg = [tuple(r) for r in groups[npos:].tolist()]
        assert len(set(got_neg)) == nneg and set(got_neg) <= set(neg_ref)
        assert labels[:npos].eq(1).all() and labels[npos:].eq(0).all()
