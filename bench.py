#!/usr/bin/env python3
"""Headline benchmark: patch-pairs/sec of one ViT-ED training step (forward + backward [+ RCCL
gradient all-reduce] + clip + AdamW) on synthetic 64x64 patch pairs, config A
(configs/puzzle/div2k_erosion7_4bin_patch8_64.yaml), batch 1024 per GPU, bf16 MFMA path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0 (contract in the task statement).  Extra objects:
  roofline      dominant kernel class (by summed device time), timed live with HIP events on the
                launch stream during an instrumented pass of the same step; achieved = algorithmic
                FLOPs of those launches / their summed duration; peak = 2500 TFLOP/s dense bf16.
  cpu_baseline  the fp32 CPU oracle (oracle/vited_oracle.py, a port - the reference itself cannot
                travel) on a bounded sample: config A, batch 32, fwd+bwd, median of 5 steps.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (never the 2:1-sparse figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1024, help='pairs per GPU')
    ap.add_argument('--cfg', default=os.path.join(ROOT, 'configs', 'puzzle', 'div2k_erosion7_4bin_patch8_64.yaml'))
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying hipGraphs')
    ap.add_argument('--fp32', action='store_true', help='run the exact fp32 kernels (parity path; not the headline)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--compress-bf16', action='store_true', help='bf16 gradients on the wire')
    return ap.parse_args()


def cpu_baseline(cfg_shape, seconds_budget=40.0):
    """fp32 eager PyTorch oracle on the host cores: batch 32, fwd+bwd."""
    from oracle import vited_oracle as vo
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = vo.OracleViTED(cfg_shape)
    b = 32
    x = torch.randn(b, 2, 3, cfg_shape.img_size, cfg_shape.img_size).clamp(-1, 1)
    y = (torch.rand(b, cfg_shape.num_classes) > 0.75).float()
    crit = torch.nn.BCEWithLogitsLoss()
    times = []
    t_start = time.time()
    for i in range(7):
        t0 = time.time()
        model.zero_grad(set_to_none=True)
        crit(model(x), y).backward()
        dt = time.time() - t0
        if i >= 2:
            times.append(dt)
        if time.time() - t_start > seconds_budget and len(times) >= 2:
            break
    med = statistics.median(times)
    return {'value': round(b / med, 2), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'config A, batch {b}, fp32 eager CPU oracle, fwd+bwd, median of {len(times)} steps after 2 warm-ups'}


def pmc_traffic(kernel_class, path=os.path.join(ROOT, 'profiles', 'r01_hbm_traffic_per_kernel.csv')):
    """HBM bytes per launch of the dominant kernel class from the COMMITTED rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE
    passes cannot run inside this process; profiles/collect.sh + summarize.py regenerate the file): launch-weighted
    mean over the class's template instances.  None when the file or the class is missing."""
    try:
        tot = n = 0.0
        for line in open(path):
            if line.startswith('#') or line.startswith('kernel,'):
                continue
            name, launches, _f, _w, total = line.rsplit(',', 4)
            if kernel_class.split('(')[0] in name:
                tot += float(launches) * float(total)
                n += float(launches)
        if n:
            return {'traffic': round(tot / n * 1e6), 'traffic_unit': 'bytes/launch (rocprofv3 PMC, profiles/r01_hbm_traffic_per_kernel.csv)'}
    except (OSError, ValueError):
        pass
    return {'traffic': None}


class LaunchTimer:
    """Brackets every C-ABI contraction launch with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops, self.records = ops, []
        self._orig = {}

    def _wrap(self, name, flops_fn, label_fn, bytes_fn=None):
        orig = getattr(self.ops, name)
        self._orig[name] = orig

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            gp, ap = self.ops.last_paths()
            self.records.append((label_fn(a, k, gp, ap), flops_fn(a, k), e0, e1, bytes_fn(a, k) if bytes_fn else 0.0))
            return out
        setattr(self.ops, name, wrapped)

    def __enter__(self):
        def gemm_flops(a, k):
            A, B = a[0], a[1]
            n = B.shape[0] if k.get('b_layout', 0) == 0 else B.shape[1]
            return 2.0 * A.shape[0] * A.shape[1] * n

        def gemm_bytes(a, k):   # algorithmic HBM bytes of one contraction + its fused epilogue (operands once, outputs once)
            A, B = a[0], a[1]
            m, kk = A.shape
            n = B.shape[0] if k.get('b_layout', 0) == 0 else B.shape[1]
            es = A.element_size()
            epi = k.get('epilogue', 0)
            out = {0: es, 1: 2 * es, 2: 4 + 4, 3: es + es, 4: 4}.get(epi, es)   # store | z+gelu | residual in + fp32 out | aux in + out | fp32
            return float(m * kk * es + n * kk * es + m * n * out)

        def tn_flops(a, k):
            dy, x = a[0], a[1]
            return 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]

        def attn_fwd_flops(a, k):
            q, kk = a[0], a[1]
            return 4.0 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]

        self._wrap('gemm', gemm_flops, lambda a, k, gp, ap: 'gemm_nt_mfma_kernel' if gp == 2 else 'gemm_portable_kernel', gemm_bytes)
        self._wrap('linear_bwd_weight', tn_flops, lambda a, k, gp, ap: 'gemm_tn_mfma_kernel(+slab/bias sums)' if gp == 2 else 'gemm_tn_portable_kernel')
        self._wrap('attention_fwd', attn_fwd_flops, lambda a, k, gp, ap: 'attn_fwd_mfma' if ap == 2 else 'attn_fwd_portable_kernel')
        self._wrap('attention_bwd', lambda a, k: 2.5 * attn_fwd_flops(a, k), lambda a, k, gp, ap: 'attn_bwd_mfma' if ap == 2 else 'attn_bwd_portable_kernels')
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(self.ops, name, orig)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for label, fl, e0, e1, nbytes in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(label, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            d['launches'] += 1
            d['ms'] += ms
            d['flops'] += fl
            d['bytes'] += nbytes
        return agg


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py measures the MI355X HIP path; no GPU is visible (the CPU oracle is only the baseline leg)')
    # rehearsal knobs (a 1-GPU box cannot run 2 RCCL ranks): VITED_DIST_BACKEND=gloo VITED_FORCE_DEVICE=0
    dev_index = int(os.environ.get('VITED_FORCE_DEVICE', local_rank))
    backend = os.environ.get('VITED_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        kw = {'device_id': dev} if backend == 'nccl' else {}
        dist.init_process_group(backend, init_method='env://', world_size=world, rank=rank, **kw)
    import vited_amd as V
    from vited_amd import engine
    V._lib.load()

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the reference's loaders print; stdout carries only the JSON line
        cfg = V.config_from_yaml(args.cfg)
        torch.manual_seed(cfg.SEED + rank)          # misc/engine.py:28
        model = V.build_model(cfg).to(dev)
    model.compute_dtype = torch.float32 if args.fp32 else torch.bfloat16
    flops_per_pair = 3 * model.flops()          # fwd+bwd = 3 x fwd (13,299,397,632 at config A)
    engine.broadcast_parameters(model)
    B, S, C = args.batch, cfg.DATA.IMG_SIZE, cfg.MODEL.NUM_CLASSES
    x = torch.randn(B, 2, 3, S, S, device=dev).clamp_(-1, 1)
    y = (torch.rand(B, C, device=dev) > 0.75).float()
    use_graph = not args.no_graph
    lr = 1e-4 * B * world / 256.0                # linear LR scaling, misc/engine.py:33-36
    groups = engine.param_groups_no_decay_1d(model)
    opt = torch.optim.AdamW(groups, lr=lr, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999), fused=True, capturable=use_graph)
    step = engine.TrainStep(model, opt, clip_grad=5.0, amp=not args.fp32, use_graph=use_graph, compress_bf16=args.compress_bf16)

    roof = None
    if rank == 0 and not args.no_roofline:
        step.flat.zero()
        step._fwd_bwd(x, y)                  # eager warm-up: sizes workspaces, builds weight shadows (no collective, no update)
        with LaunchTimer(V.ops) as lt:       # same step, launched eagerly so each launch can be bracketed
            for _ in range(2):
                step.flat.zero()
                step._fwd_bwd(x, y)
            agg = lt.summary()
        kernels = {k: {'launches': v['launches'], 'avg_us': round(1e3 * v['ms'] / v['launches'], 2),
                       'total_ms_per_step': round(v['ms'] / 2, 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)}
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
        dom = next(iter(kernels))
        d = agg[dom]
        achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
        roof = {'roofline': {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 1), 'peak': PEAK_BF16_DENSE_TFLOPS,
                             'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_DENSE_TFLOPS, 4),
                             'avg_launch_us': round(1e3 * d['ms'] / d['launches'], 2), 'launches_per_step': d['launches'] // 2,
                             'algorithmic_bytes_per_launch': round(d['bytes'] / d['launches']), **pmc_traffic(dom)},
                'kernels': kernels}
        step.flat.zero()
    if world > 1:
        dist.barrier()
    # setup (eager steps + graph capture) and W warm-up steps, all untimed
    for _ in range(3 if use_graph else 1):
        step.step(x, y)
    for _ in range(args.warmup):
        step.step(x, y)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step.step(x, y)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    ms_per_step = 1e3 * elapsed / args.steps
    pairs_per_s = B * world * args.steps / elapsed

    out = {
        'metric': 'patch-pairs/sec fwd+bwd, div2k patch8_64 ViT-ED', 'value': round(pairs_per_s, 1), 'unit': 'pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.fp32 else 'bf16', 'data': 'synthetic',
        'config': {'workload': f'{os.path.basename(args.cfg)} batch {B}/GPU, {S}x{S} random patch pairs, full train step '
                               f'(fwd+bwd+{"RCCL all-reduce+" if world > 1 else ""}clip+AdamW)',
                   'global_batch': B * world, 'parallelism': f'dp{world}', 'hipgraph': use_graph},
        'step_tflops': round(pairs_per_s * flops_per_pair / 1e12, 2),
        'step_frac_of_bf16_peak': round(pairs_per_s * flops_per_pair / 1e12 / (PEAK_BF16_DENSE_TFLOPS * world), 4),
        'loss': round(loss_val, 5),
    }

    if roof is not None:
        out.update(roof)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import vited_oracle as vo
        pjs = cfg.MODEL.PJS
        shape = vo.ViTEDShape(img_size=S, patch_size=pjs.PATCH_SIZE, in_chans=pjs.IN_CHANS, num_classes=C,
                              embed_dim=pjs.EMBED_DIM, depth=pjs.DEPTH, c_depth=pjs.C_DEPTH, num_heads=pjs.NUM_HEADS)
        out['cpu_baseline'] = cpu_baseline(shape)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
