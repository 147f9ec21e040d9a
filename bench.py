#!/usr/bin/env python3
"""Headline benchmark: patch-pairs/sec of one ViT-ED training step (forward + backward [+ RCCL gradient all-reduce] +
clip + AdamW) on synthetic patch pairs, bf16 MFMA path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload A-train|H-train|H-infer]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

``--gpus N`` with no WORLD_SIZE in the environment starts the N ranks itself (``torch.distributed.run`` as a child; this
parent makes no GPU call) and relays rank 0's JSON line; under an external launcher WORLD_SIZE must equal ``--gpus``.

Workloads (BASELINE.json ``configs``):
  A-train  configs/puzzle/div2k_erosion7_4bin_patch8_64.yaml, 1024 pairs/GPU, full train step (the headline, default)
  H-train  configs/hisfrag/hisfrag20_patch16_512.yaml, 24 images/GPU: the two-stage step of hisfrag.py:117-159 (encoder once
           per image, 24 positive + 48 negative pairs through the decoder, one backward), 72 pairs/step
  H-infer  the same model, pairwise similarity-matrix inference (hisfrag.py:161-302): 96 images/GPU, pair batch 512

One JSON line on rank 0 (contract in the task statement).  Extra objects:
  roofline      dominant kernel class (by summed device time), timed live with HIP events on the launch stream during an
                instrumented eager pass of the same step, against the roof SURVEY.md section 8(d) / BASELINE.md section 2 name
                for this path: dense bf16 MFMA, 2500 TFLOP/s (MI355X_MICROARCH.md).  ``achieved`` = the launches' algorithmic
                FLOPs (2MNK) over their summed duration.  ``other_roof`` carries the HBM side (operand + result bytes counted
                once over the same duration, against 8000 GB/s).  ``traffic`` = measured HBM bytes per launch from the
                committed rocprofv3 PMC passes; ``step_algorithmic_bytes`` = the byte count over every instrumented launch.
  fused_block   north_star's "attention + MLP fused block" figure from THIS run: algorithmic FLOPs of one encoder Block
                (24 N D^2 + 4 N^2 D per image) at the bench batch over the measured device time of that block's kernels
                (HIP events around each block in the instrumented pass), forward and forward + backward, against 2500 TFLOP/s.
  executed_flops_per_unit / step_frac_executed   FLOPs the launches really execute per unit (the cls-row tail of the last
                decoder block and the pair cache of H-infer execute fewer than the reference-equivalent ``flops_per_unit``).
  other_workloads   (A-train, one GPU) short H-train and H-infer legs run after the headline timing: BASELINE configs 3 and 5.
  cpu_baseline  the fp32 CPU oracle (oracle/vited_oracle.py, a port - the reference itself cannot travel) on a bounded
                sample: config A, batch 32, fwd+bwd, median of 5 steps, with the host's CPU model and thread count.
"""
import argparse
import contextlib
import json
import math
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (never the 2:1-sparse figure)
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CFG_A = os.path.join(ROOT, 'configs', 'puzzle', 'div2k_erosion7_4bin_patch8_64.yaml')
CFG_H = os.path.join(ROOT, 'configs', 'hisfrag', 'hisfrag20_patch16_512.yaml')
def _latest_pmc_csv():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_hbm_traffic_per_kernel.csv')))
    return found[-1] if found else os.path.join(ROOT, 'profiles', 'r03_hbm_traffic_per_kernel.csv')


PMC_TRAFFIC_CSV = _latest_pmc_csv()       # the newest committed rocprofv3 PMC summary (profiles/collect.sh + summarize.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='A-train', choices=['A-train', 'H-train', 'H-infer'])
    ap.add_argument('--batch', type=int, default=None, help='pairs (A-train) or images (H-*) per GPU; default 1024 / 24 / 96')
    ap.add_argument('--cfg', default=None, help='override the YAML of the workload')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying hipGraphs')
    ap.add_argument('--fp32', action='store_true', help='run the exact fp32 kernels (parity path; not the headline)')
    ap.add_argument('--torch-optim', action='store_true', help='torch.optim.AdamW(fused) + separate clip instead of the HIP optimizer kernel')
    ap.add_argument('--no-overlap', action='store_true', help='one all-reduce after the whole backward instead of two buckets')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the short H-train / H-infer legs after the A-train headline')
    ap.add_argument('--compress-bf16', action='store_true', help='bf16 gradients on the wire')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher -> start the N ranks (no GPU call in this process)
# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(f'bench.py --gpus {args.gpus}: the launched ranks failed (exit code {proc.returncode})')
    if json.loads(line)['n_gpus'] != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: the job ran on {json.loads(line)["n_gpus"]} ranks')
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------
# host facts
# ---------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def rocminfo_peak():
    """Dense bf16 MFMA peak of THIS device from rocminfo: CUs x 4 SIMDs x 1024 FLOP/clk/SIMD (one 16x16x32 bf16 MFMA = 16,384
    FLOP per 16 cycles) x max clock.  None when rocminfo is unavailable.  (A child process: it does not touch this process's HIP state.)"""
    try:
        txt = subprocess.run(['rocminfo'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    for agent in txt.split('*******')[1:]:
        if 'gfx' not in agent or 'Device Type:             GPU' not in agent:
            continue
        cu = re.search(r'Compute Unit:\s+(\d+)', agent)
        clk = re.search(r'Max Clock Freq\. \(MHz\):\s+(\d+)', agent)
        name = re.search(r'Name:\s+(gfx\w+)', agent)
        if cu and clk:
            cus, mhz = int(cu.group(1)), int(clk.group(1))
            return {'cus': cus, 'max_clock_mhz': mhz, 'arch': name.group(1) if name else '?',
                    'tflops': round(cus * 4 * 1024 * mhz * 1e6 / 1e12, 1)}
    return None


def cpu_baseline(cfg_shape, seconds_budget=40.0):
    """fp32 eager PyTorch oracle on the host cores: batch 32, fwd+bwd."""
    import torch
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
    return {'value': round(b / med, 2), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port', 'cpu': cpu_model_name(),
            'sample': f'config A, batch {b}, fp32 eager CPU oracle, fwd+bwd, median of {len(times)} steps after 2 warm-ups, '
                      f'torch.set_num_threads({cores})'}


def pmc_traffic(kernel_class, path=PMC_TRAFFIC_CSV):
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
            return {'traffic': round(tot / n * 1e6), 'traffic_unit': f'bytes/launch (rocprofv3 PMC, profiles/{os.path.basename(path)})'}
    except (OSError, ValueError):
        pass
    return {'traffic': None}


class LaunchTimer:
    """Brackets every C-ABI contraction launch with HIP events on the launch stream."""

    def __init__(self, ops):
        import torch
        self.torch, self.ops, self.records = torch, ops, []
        self._orig = {}

    def _wrap(self, name, flops_fn, label_fn, bytes_fn=None):
        orig = getattr(self.ops, name)
        self._orig[name] = orig
        torch = self.torch

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            if out is False:             # linear_bwd_weight_batched: the set is not covered, nothing was launched
                return out
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
            # per output element: store | GELU: two outputs | residual in + fp32 out | aux in + out | fp32 out | aux in + out | two outputs
            out = {0: es, 1: 2 * es, 2: 4 + 4, 3: es + es, 4: 4, 5: es + es, 6: 2 * es}.get(epi, es)
            return float(m * kk * es + n * kk * es + m * n * out)

        def tn_bytes(a, k):     # both operands once + the fp32 result (the split-M slabs are workspace traffic, not algorithmic)
            dy, x = a[0], a[1]
            return float(dy.numel() * dy.element_size() + x.numel() * x.element_size() + 4 * dy.shape[1] * x.shape[1])

        def attn_bytes(a, k, bwd=False):   # q, k, v (+ o, do) in, o (or dq, dk, dv) out
            q, kk, v = a[0], a[1], a[2]
            io = sum(t.numel() * t.element_size() for t in (q, kk, v))
            return float(2 * io + q.numel() * q.element_size()) if bwd else float(io + q.numel() * q.element_size())

        def ln_fwd_bytes(a, k):
            x = a[0]
            return float(x.numel() * (4 + 2))

        def ln_bwd_bytes(a, k):   # dy (act dtype), x, dx_in fp32 in; dx fp32 (+ low-precision copy) out
            dy, x = a[0], a[1]
            n = x.numel()
            return float(n * (dy.element_size() + 4 + (4 if k.get('dx_in') is not None else 0) + 4 + (2 if (k.get('want_lp') or k.get('dx_lp') is not None) else 0)))

        def tn_flops(a, k):
            dy, x = a[0], a[1]
            return 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]

        def attn_fwd_flops(a, k):
            q, kk = a[0], a[1]
            return 4.0 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]

        self._wrap('gemm', gemm_flops, lambda a, k, gp, ap: 'gemm_nt_mfma_kernel' if gp == 2 else 'gemm_portable_kernel', gemm_bytes)
        self._wrap('linear_bwd_weight', tn_flops, lambda a, k, gp, ap: 'gemm_tn_mfma_kernel(+slab/bias sums)' if gp == 2 else 'gemm_tn_portable_kernel', tn_bytes)
        # a block's weight gradients in one launch (+ one slab-sum launch): items = [(dy [M, N], x [M, K], dw, db | None), ...]
        self._wrap('linear_bwd_weight_batched', lambda a, k: sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _dw, _db in a[0]),
                   lambda a, k, gp, ap: 'gemm_tn_wide_kernel(batched dW + slab sums)',
                   lambda a, k: float(sum(dy.numel() * dy.element_size() + x.numel() * x.element_size() + 4 * dy.shape[1] * x.shape[1]
                                          for dy, x, _dw, _db in a[0])))
        self._wrap('attention_fwd', attn_fwd_flops, lambda a, k, gp, ap: 'attn_fwd_mfma' if ap == 2 else 'attn_fwd_portable_kernel', attn_bytes)
        self._wrap('attention_bwd', lambda a, k: 2.5 * attn_fwd_flops(a, k), lambda a, k, gp, ap: 'attn_bwd_mfma' if ap == 2 else 'attn_bwd_portable_kernels',
                   lambda a, k: attn_bytes(a, k, True))
        self._wrap('layernorm_fwd', lambda a, k: 0.0, lambda a, k, gp, ap: 'layernorm_fwd_kernel', ln_fwd_bytes)
        self._wrap('layernorm_bwd', lambda a, k: 0.0, lambda a, k, gp, ap: 'layernorm_bwd_kernel', ln_bwd_bytes)
        def row_mk(t):                 # [M, K], or [L, M, seg_k]: the contraction dim as L tensors (all decoder blocks' d(kv))
            return (t.shape[1], t.shape[0] * t.shape[2]) if t.dim() == 3 else (t.shape[0], t.shape[1])

        def row_flops(a, k):           # a . W^T with W [384, K]
            m, kk = row_mk(a[0])
            return 2.0 * m * kk * a[1].shape[0]

        def row_fwd_bytes(a, k):       # a, W in; residual in, y out (fp32); h out (bf16) when the LayerNorm is fused
            m, kk = row_mk(a[0])
            n = a[1].shape[0]
            return float(m * kk * 2 + n * kk * 2 + m * n * (4 + 4 + (2 if len(a) > 4 and a[4] is not None or k.get('gamma') is not None else 0)))

        def row_bwd_bytes(a, k):       # dy, Wt in; x, dx_in in; dx (+ bf16 copy) out
            m, kk = row_mk(a[0])
            n = a[1].shape[0]
            return float(m * kk * 2 + n * kk * 2 + m * n * (4 + (4 if k.get('dx_in') is not None else 0) + 4 + (2 if k.get('want_lp') else 0)))

        self._wrap('linear_residual_layernorm_fwd', row_flops, lambda a, k, gp, ap: 'gemm_row_kernel', row_fwd_bytes)
        self._wrap('linear_layernorm_bwd', row_flops, lambda a, k, gp, ap: 'gemm_row_kernel', row_bwd_bytes)
        if hasattr(self.ops, 'mlp_fwd'):
            self._wrap('mlp_fwd', lambda a, k: 4.0 * a[0].shape[0] * a[0].shape[1] * a[3].shape[0], lambda a, k, gp, ap: 'mlp_fwd_fused_kernel')
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(self.ops, name, orig)

    def summary(self):
        self.torch.cuda.synchronize()
        agg = {}
        for label, fl, e0, e1, nbytes in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(label, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            d['launches'] += 1
            d['ms'] += ms
            d['flops'] += fl
            d['bytes'] += nbytes
        return agg


def roofline_of(V, run_once, passes=2, runtime=None, block_flops=None):
    """Instrumented eager passes of ``run_once`` -> the ``roofline`` / ``kernels`` (/ ``fused_block``) objects of the JSON line
    and the FLOPs the launches executed per pass."""
    import torch
    run_once()                           # eager warm-up: sizes workspaces, builds weight shadows
    spans = []
    if runtime is not None:
        runtime.block_events = spans
    try:
        with LaunchTimer(V.ops) as lt:       # same work, launched eagerly so each launch can be bracketed
            for _ in range(passes):
                run_once()
            agg = lt.summary()
    finally:
        if runtime is not None:
            runtime.block_events = None
    kernels = {k: {'launches': v['launches'], 'launches_per_step': v['launches'] // passes, 'avg_us': round(1e3 * v['ms'] / v['launches'], 2),
                   'total_ms_per_step': round(v['ms'] / passes, 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1),
                   'algorithmic_TBps': round(v['bytes'] / (v['ms'] * 1e-3) / 1e12, 2)}
               for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
    dom = next(iter(kernels))
    d = agg[dom]
    secs = d['ms'] * 1e-3
    tflops, tbps = d['flops'] / secs / 1e12, d['bytes'] / secs / 1e12
    derived = rocminfo_peak()
    # SURVEY.md section 8(d) / BASELINE.md section 2: the path is bounded by the dense bf16 MFMA roof; the HBM side rides along
    mfma = {'bound': 'mfma', 'achieved': round(tflops, 1), 'peak': PEAK_BF16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tflops / PEAK_BF16_DENSE_TFLOPS, 4)}
    hbm = {'bound': 'hbm', 'achieved': round(tbps * 1e3, 1), 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': round(tbps * 1e3 / PEAK_HBM_GBPS, 4)}
    roof = {**mfma, 'kernel': dom, 'other_roof': hbm, 'peak_from_rocminfo': derived,
            'avg_launch_us': round(1e3 * d['ms'] / d['launches'], 2), 'launches_per_step': d['launches'] // passes,
            'algorithmic_bytes_per_launch': round(d['bytes'] / d['launches']),
            'algorithmic_flops_per_launch': round(d['flops'] / d['launches']), **pmc_traffic(dom)}
    # all instrumented launches of one step (GEMMs, dW, attention, LayerNorm): the dataflow's own HBM bytes
    roof['step_algorithmic_bytes'] = round(sum(v['bytes'] for v in agg.values()) / passes)
    out = {'roofline': roof, 'kernels': kernels, '_executed_flops_per_pass': sum(v['flops'] for v in agg.values()) / passes}
    if spans and block_flops:
        torch.cuda.synchronize()
        fb = {}
        for phase in ('fwd', 'bwd'):
            ms = [e0.elapsed_time(e1) for kind, _i, ph, e0, e1 in spans if kind == 'enc' and ph == phase]
            if ms:
                fb[phase] = statistics.median(ms)
        # the encoder blocks' weight-gradient launches go out for several blocks at once (after the blocks' own spans): their time,
        # shared equally by the blocks, belongs to a block's backward
        dw_ms = sum(e0.elapsed_time(e1) for kind, _i, ph, e0, e1 in spans if kind == 'enc' and ph == 'dw')
        nblk = len([1 for kind, _i, ph, _e0, _e1 in spans if kind == 'enc' and ph == 'bwd'])
        if 'bwd' in fb and nblk:
            fb['bwd'] += dw_ms / nblk
        if 'fwd' in fb:
            tf = block_flops / (fb['fwd'] * 1e-3) / 1e12
            obj = {'what': 'one encoder Block (LayerNorm, qkv, attention, proj + residual, LayerNorm, fc1 + GELU, fc2 + residual) at the bench batch, '
                           'median over the encoder\'s blocks of the device time between HIP events around the block\'s launches (backward: plus the block\'s '
                           'share of the weight-gradient launches, which go out for several blocks at once)',
                   'flops_fwd': round(block_flops), 'fwd_us': round(1e3 * fb['fwd'], 1), 'fwd_tflops': round(tf, 1),
                   'fwd_frac_of_bf16_peak': round(tf / PEAK_BF16_DENSE_TFLOPS, 4), 'target_frac': 0.40}
            if 'bwd' in fb:
                tot = fb['fwd'] + fb['bwd']
                tf3 = 3 * block_flops / (tot * 1e-3) / 1e12
                obj.update({'bwd_us': round(1e3 * fb['bwd'], 1), 'fwd_bwd_tflops': round(tf3, 1),
                            'fwd_bwd_frac_of_bf16_peak': round(tf3 / PEAK_BF16_DENSE_TFLOPS, 4)})
            out['fused_block'] = obj
    return out


# ---------------------------------------------------------------------------------------------
class Ctx:
    """Process-wide state of one bench run (rank / device / process group) shared by the workload legs."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get('WORLD_SIZE') or 1)
        self.rank = int(os.environ.get('RANK', 0))
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
        if not torch.cuda.is_available():
            raise SystemExit('bench.py measures the MI355X HIP path; no GPU is visible (the CPU oracle is only the baseline leg)')
        # rehearsal knobs (a 1-GPU box cannot run 2 RCCL ranks): VITED_DIST_BACKEND=gloo VITED_FORCE_DEVICE=0
        dev_index = int(os.environ.get('VITED_FORCE_DEVICE', local_rank))
        backend = os.environ.get('VITED_DIST_BACKEND', 'nccl')
        torch.cuda.set_device(dev_index)
        self.dev = torch.device('cuda', dev_index)
        if self.world > 1:
            kw = {'device_id': self.dev} if backend == 'nccl' else {}
            dist.init_process_group(backend, init_method='env://', world_size=self.world, rank=self.rank, **kw)
        import vited_amd as V
        V._lib.load()
        self.V, self.engine = V, V.engine

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def build(self, cfg_path):
        torch, V = self.torch, self.V
        with contextlib.redirect_stdout(sys.stderr):   # the reference's loaders print; stdout carries only the JSON line
            cfg = V.config_from_yaml(cfg_path)
            torch.manual_seed(cfg.SEED + self.rank)     # misc/engine.py:28
            model = V.build_model(cfg).to(self.dev)
        model.compute_dtype = torch.float32 if self.args.fp32 else torch.bfloat16
        self.engine.broadcast_parameters(model)
        return cfg, model

    def max_over_ranks(self, elapsed):
        if self.world > 1:
            t = self.torch.tensor([elapsed], device=self.dev, dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return elapsed


def _line(ctx, *, metric, what, units, unit_flops, executed_flops, elapsed, steps, warmup, batch, hipgraph, extra, roof):
    """The JSON object of one workload leg.  ``units`` = units THIS rank processed per step (weak scaling: x world)."""
    args, world = ctx.args, ctx.world
    elapsed = ctx.max_over_ranks(elapsed)
    per_s = units * world * steps / elapsed
    out = {
        'metric': metric, 'value': round(per_s, 1), 'unit': 'pairs/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(1e3 * elapsed / steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.fp32 else 'bf16', 'data': 'synthetic',
        'config': {'workload': what, 'global_batch': batch * world, 'parallelism': f'dp{world}', 'hipgraph': hipgraph},
        # reference-equivalent work: what the reference's code path would compute for the same units
        'step_tflops': round(per_s * unit_flops / 1e12, 2),
        'step_frac_of_bf16_peak': round(per_s * unit_flops / 1e12 / (PEAK_BF16_DENSE_TFLOPS * world), 4),
        'flops_per_unit': round(unit_flops),
    }
    if executed_flops is not None:
        # what the launches really execute (cls-row tail of the last decoder block; pair cache at H-infer): the utilisation figure
        out['executed_flops_per_unit'] = round(executed_flops)
        out['step_tflops_executed'] = round(per_s * executed_flops / 1e12, 2)
        out['step_frac_executed'] = round(per_s * executed_flops / 1e12 / (PEAK_BF16_DENSE_TFLOPS * world), 4)
    out.update(extra)
    if roof is not None:
        out.update({k: v for k, v in roof.items() if not k.startswith('_')})
    return out


def run_train(ctx, workload, *, steps, warmup, batch=None, cfg_path=None, want_roofline=True):
    """A-train / H-train: W untimed warm-up steps, then exactly K timed steps between two fences."""
    torch, V, engine, args, world, rank, dev = ctx.torch, ctx.V, ctx.engine, ctx.args, ctx.world, ctx.rank, ctx.dev
    cfg_path = cfg_path or (CFG_A if workload == 'A-train' else CFG_H)
    cfg, model = ctx.build(cfg_path)
    S, C, name = cfg.DATA.IMG_SIZE, cfg.MODEL.NUM_CLASSES, cfg.MODEL.NAME
    enc_flops, dec_flops = model.flops_parts()     # per image / per pair, forward
    use_graph = not args.no_graph
    extra = {}
    B = batch or (1024 if workload == 'A-train' else 24)
    lr = 1e-4 * B * world / 256.0                # linear LR scaling, misc/engine.py:33-36
    groups = engine.param_groups_no_decay_1d(model)
    okw = dict(lr=lr, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999))
    if args.torch_optim:
        opt = torch.optim.AdamW(groups, fused=True, capturable=use_graph and workload == 'A-train', **okw)
    else:
        opt = V.optim.FlatAdamW(groups, model=model, **okw)
    if workload == 'A-train':
        x = torch.randn(B, 2, 3, S, S, device=dev).clamp_(-1, 1)
        y = (torch.rand(B, C, device=dev) > 0.75).float()
        step = engine.TrainStep(model, opt, clip_grad=5.0, amp=not args.fp32, use_graph=use_graph, compress_bf16=args.compress_bf16,
                                overlap=not args.no_overlap)
        units, unit_flops = B, 3 * (enc_flops + dec_flops)     # fwd+bwd = 3 x fwd (13,299,397,632 at config A)
        what = f'{os.path.basename(cfg_path)} batch {B}/GPU, {S}x{S} random patch pairs, full train step'
    else:
        # hisfrag.py:117-159 with MPerClassSampler(m=3): 8 writers x 3 images -> 24 positive pairs + min(252, 2*24) = 48 negatives
        use_graph = False                        # the pair batch is a structured input; ~100 ms of kernels hide the launches
        samples = torch.randn(B, 3, S, S, device=dev).clamp_(-1, 1)
        targets = torch.arange(B // 3, device=dev).repeat_interleave(3)
        groups_idx, labels = engine.mine_pairs(targets, generator=torch.Generator(device=dev).manual_seed(cfg.SEED + rank))
        P = int(groups_idx.shape[0])

        def two_stage(m, batch_):
            imgs, pairs = batch_
            feats = m(imgs, forward_first_part=True)
            return m(feats[pairs[:, 1]], imgs[pairs[:, 0]])

        x, y = (samples, groups_idx), labels
        step = engine.TrainStep(model, opt, clip_grad=5.0, amp=not args.fp32, use_graph=False, compress_bf16=args.compress_bf16,
                                forward_fn=two_stage)
        units, unit_flops = P, 3 * (enc_flops * B + dec_flops * P) / P
        what = (f'{os.path.basename(cfg_path)} {B} images/GPU ({S}x{S}), two-stage step of hisfrag.py:117-159: encoder once per image, '
                f'{P} mined pairs through the decoder, one backward')
        extra['images_per_step'] = B * world

    roof = None
    if rank == 0 and want_roofline:
        def run_once():
            step.flat.zero()
            step._fwd_bwd(x, y)              # no collective, no update
        d, n1 = model.embed_dim, model.patch_embed.num_patches
        nimg = B                                         # encoder images per step (A: one per pair; H: the batch's images)
        block_flops = nimg * (24 * n1 * d * d + 4 * n1 * n1 * d)
        roof = roofline_of(V, run_once, passes=2 if workload == 'A-train' else 1, runtime=model.runtime(), block_flops=block_flops)
        step.flat.zero()
    if world > 1:
        ctx.dist.barrier()
    # setup (eager steps + graph capture) and W warm-up steps, all untimed
    loss_first = float(step.step(x, y))
    for _ in range(2 if use_graph else 0):
        step.step(x, y)
    for _ in range(warmup):
        step.step(x, y)
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step.step(x, y)
    ctx.fence()
    elapsed = time.perf_counter() - t0
    loss_val, norm_val = float(loss), float(step.last_norm)
    # training on one fixed batch must reduce its loss; anything else means the step is not doing its work
    if not (math.isfinite(loss_val) and math.isfinite(norm_val) and loss_val < loss_first):
        raise SystemExit(f'bench.py: the timed steps did not train (loss {loss_first:.5f} -> {loss_val:.5f}, grad norm {norm_val})')
    extra.update({'loss_first': round(loss_first, 5), 'loss': round(loss_val, 5), 'grad_norm': round(norm_val, 5),
                  'optimizer': 'torch.optim.AdamW(fused)' if args.torch_optim else 'vited_adamw_step (HIP, fused clip)'})
    metric = f'patch-pairs/sec fwd+bwd, {name} ViT-ED'
    what += f' (fwd+bwd+{"RCCL all-reduce+" if world > 1 else ""}clip+AdamW)'
    executed = roof['_executed_flops_per_pass'] / units if roof is not None else None
    return _line(ctx, metric=metric, what=what, units=units, unit_flops=unit_flops, executed_flops=executed, elapsed=elapsed, steps=steps,
                 warmup=warmup, batch=B, hipgraph=use_graph and workload == 'A-train', extra=extra, roof=roof)


def run_train_hostfed(ctx, *, steps, warmup, batch=None):
    """The A-train step FED FROM THE HOST (misc/engine.py:202-204 copies every batch at the top of
    the iteration): each step's batch starts as uint8 pixels in pageable host memory (what a DataLoader hands over), goes
    through ``engine.DevicePrefetcher`` (pinned staging, side-stream copy two batches ahead) and enters the patch-embedding
    kernel as uint8 (ToTensor + Normalize folded in).  Same model, optimizer and hipGraph step as the headline; the timed region
    includes the H2D copies."""
    torch, V, engine, args, world, rank, dev = ctx.torch, ctx.V, ctx.engine, ctx.args, ctx.world, ctx.rank, ctx.dev
    cfg, model = ctx.build(CFG_A)
    S, C = cfg.DATA.IMG_SIZE, cfg.MODEL.NUM_CLASSES
    B = batch or 1024
    opt = V.optim.FlatAdamW(engine.param_groups_no_decay_1d(model), model=model, lr=1e-4 * B * world / 256.0, weight_decay=0.05)
    step = engine.TrainStep(model, opt, clip_grad=5.0, amp=not args.fp32, use_graph=not args.no_graph)
    g = torch.Generator().manual_seed(cfg.SEED + rank)
    host = [(torch.randint(0, 256, (B, 2, 3, S, S), generator=g, dtype=torch.uint8), (torch.rand(B, C, generator=g) > 0.75).float())
            for _ in range(4)]                                   # four distinct pageable batches, cycled
    total = warmup + steps + 3

    def loader():
        for i in range(total):
            yield host[i % len(host)]

    it = iter(engine.DevicePrefetcher(loader(), dev, depth=2))
    for _ in range(3 + warmup):                                  # eager steps + capture, then W warm-up steps
        x, y = next(it)
        step.step(x, y)
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        x, y = next(it)
        loss = step.step(x, y)
    ctx.fence()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    if not math.isfinite(float(loss)):
        raise SystemExit('bench.py: the host-fed steps diverged')
    return {'metric': 'patch-pairs/sec fwd+bwd, host-fed uint8 batches', 'value': round(B * world * steps / elapsed, 1), 'unit': 'pairs/s',
            'ms_per_step': round(1e3 * elapsed / steps, 3), 'steps': steps, 'h2d_bytes_per_step': B * 2 * 3 * S * S + B * C * 4,
            'workload': f'{os.path.basename(CFG_A)} batch {B}/GPU: uint8 pixels from pageable host memory -> DevicePrefetcher (pinned, '
                        f'side stream, depth 2) -> vited_patchify_u8; full train step, H2D inside the timed region'}


def run_infer(ctx, *, steps, warmup, batch=None, cfg_path=None, want_roofline=True):
    """H-infer: pairwise similarity-matrix inference (hisfrag.py:161-302): every rank encodes its row block and streams all later
    images; one "step" = one whole similarity matrix."""
    torch, V, engine, args, world, rank, dev = ctx.torch, ctx.V, ctx.engine, ctx.args, ctx.world, ctx.rank, ctx.dev
    cfg_path = cfg_path or CFG_H
    cfg, model = ctx.build(cfg_path)
    S, name = cfg.DATA.IMG_SIZE, cfg.MODEL.NAME
    enc_flops, dec_flops = model.flops_parts()
    n_img = (batch or 96) * world
    g = torch.Generator(device=dev).manual_seed(cfg.SEED)
    images = torch.randn(n_img, 3, S, S, device=dev, generator=g).clamp_(-1, 1)
    pair_batch = 512                              # README.md:63 (--opts DATA.TEST_BATCH_SIZE 512), BASELINE config 5
    bounds = engine.shard_rows_by_pair_count(n_img, world)
    my_rows = bounds[rank + 1] - bounds[rank]
    my_pairs = sum(n_img - i for i in range(bounds[rank], bounds[rank + 1]))
    total_pairs = n_img * (n_img + 1) // 2

    def run():
        return engine.pairwise_similarity(model, images, rank=rank, world=world, block=64, pair_batch=pair_batch, amp=not args.fp32)

    roof = None
    if rank == 0 and want_roofline and world == 1:
        roof = roofline_of(V, run, passes=1)      # the whole matrix once, every launch bracketed: executed FLOPs are counted, not modelled
    for _ in range(max(warmup, 1)):
        run()
    steps = max(1, min(steps, 5))
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim = run()
    ctx.fence()
    elapsed = time.perf_counter() - t0
    if not bool(torch.isfinite(sim.float()).all()) or not torch.equal(sim, sim.t()):
        raise SystemExit('bench.py: the similarity matrix is not finite / symmetric')
    units = total_pairs / world                   # _line multiplies by world again
    # reference-equivalent work: hisfrag.py:226-229 runs the whole decoder (and image 2's patch embedding) for every pair
    unit_flops = (dec_flops * total_pairs + enc_flops * n_img) / total_pairs
    metric = f'pairs/sec similarity-matrix inference, {name} ViT-ED'
    what = (f'{os.path.basename(cfg_path)} {n_img} images ({S}x{S}) -> {total_pairs} pairs (upper triangle incl. diagonal), pair batch '
            f'{pair_batch}, encoder once per image, row blocks sharded by pair count, one all-gather of scores')
    extra = {'pairs_this_rank': my_pairs, 'rows_this_rank': my_rows}
    executed = roof['_executed_flops_per_pass'] / total_pairs if roof is not None else None
    return _line(ctx, metric=metric, what=what, units=units, unit_flops=unit_flops, executed_flops=executed, elapsed=elapsed, steps=steps,
                 warmup=warmup, batch=n_img // world, hipgraph=False, extra=extra, roof=roof)


def _brief(leg):
    keep = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'step_tflops', 'step_frac_of_bf16_peak', 'flops_per_unit',
            'executed_flops_per_unit', 'step_tflops_executed', 'step_frac_executed')
    out = {k: leg[k] for k in keep if k in leg}
    out['workload'] = leg['config']['workload']
    return out


def main():
    args = parse()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args)                  # before anything touches the GPU
    world = int(env_world or 1)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    ctx = Ctx(args)
    torch = ctx.torch
    want_roof = not args.no_roofline
    if args.workload in ('A-train', 'H-train'):
        out = run_train(ctx, args.workload, steps=args.steps, warmup=args.warmup, batch=args.batch, cfg_path=args.cfg, want_roofline=want_roof)
    else:
        out = run_infer(ctx, steps=args.steps, warmup=args.warmup, batch=args.batch, cfg_path=args.cfg, want_roofline=want_roof)
    if args.workload == 'A-train' and world == 1 and not args.no_other_workloads and args.cfg is None:
        # BASELINE configs 3 and 5 on this GPU, AFTER the headline's timed region: short legs, reported beside the headline
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        legs = {}
        legs['A-train-hostfed'] = run_train_hostfed(ctx, steps=10, warmup=2)
        gc.collect()
        torch.cuda.empty_cache()
        legs['H-train'] = _brief(run_train(ctx, 'H-train', steps=5, warmup=2, want_roofline=want_roof))
        gc.collect()
        torch.cuda.empty_cache()
        legs['H-infer'] = _brief(run_infer(ctx, steps=1, warmup=1, batch=48, want_roofline=want_roof))
        out['other_workloads'] = legs
    if ctx.rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import vited_oracle as vo
        out['cpu_baseline'] = cpu_baseline(vo.SHAPE_A)
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == '__main__':
    main()
