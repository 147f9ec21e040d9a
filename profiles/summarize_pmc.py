#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (written by profiles/collect_pmc.sh) -> profiles/<tag>_gemm_pmc.csv: per GEMM kernel variant and
launch geometry, the counters DESIGN.md section 6 argues from.

    python3 profiles/summarize_pmc.py r02

Columns (means over the launches of that variant/geometry):
  avg_us              kernel duration from the --kernel-trace pass
  wait_any_pct        SQ_WAIT_ANY / SQ_WAVE_CYCLES      (waves parked on s_waitcnt / barriers)
  wait_inst_pct       SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls)
  active_inst_pct     SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  mfma_busy_pct       SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (matrix pipe busy, per SIMD)
  lds_active_pct      SQ_LDS_IDX_ACTIVE / SQ_WAVE_CYCLES
  lds_conflict_pct    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2_hit_pct          TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  tcp_pending_stall   TCP_PENDING_STALL_CYCLES_sum per launch (cycles, summed over the 256 CUs)
  l2_to_cu_MB         TCP_TCC_READ_REQ_sum x request size (bytes the CUs requested from L2: the operand volume staged into LDS).
                      A request is one contiguous run of a wave-instruction inside a 128-B line: 64 B for the BK = 32 NT stages
                      (64-B rows), 128 B for the BK = 64 NT stages and the 256-B rows of the TN stages.  With these sizes the
                      column reproduces the tile arithmetic (tiles x (BM + BN) x K x 2 B) of every shape to within 1 %.
  hbm_fetch_MB        2 x FETCH_SIZE (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md), hbm_write_MB = WRITE_SIZE
"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'pmc_{tag}')


def short(name):
    m = re.search(r'(gemm_\w+_kernel<[^>]*>|sum_slabs\w*|gemm_\w+_kernel)', name)
    return m.group(1) if m else None


# the probe launches its shapes in this order, each (2 warm-up + --reps 2) = 4 times: label dispatches by position
SHAPES = ['qkv N=1152 K=384', 'proj+res N=384 K=384', 'fc1+gelu N=1536 K=384', 'fc2+res N=384 K=1536', "dz=dy.W2*gelu' N=1536 K=384",
          'dh=dz.W1 N=384 K=1536', 'dh=dqkv.Wqkv N=384 K=1152', 'do=dx.Wp N=384 K=384', 'kv N=768 K=384',
          'dWqkv N=1152 K=384', 'dWproj N=384 K=384', 'dWfc1 N=1536 K=384', 'dWfc2 N=384 K=1536', 'dWkv N=768 K=384']
LAUNCHES_PER_SHAPE = 4
_order = {}


def label(path):
    """dispatch id -> shape label for one rocprofv3 output file (GEMM main kernels only, in dispatch order)."""
    if path not in _order:
        ids = sorted({int(r['Dispatch_Id']) for r in csv.DictReader(open(path))
                      if (short(r['Kernel_Name']) or '').startswith(('gemm_nt', 'gemm_tn'))})
        _order[path] = {d: SHAPES[min(i // LAUNCHES_PER_SHAPE, len(SHAPES) - 1)] for i, d in enumerate(ids)}
    return _order[path]


def key(r, path):
    n = short(r['Kernel_Name'])
    if n is None or not n.startswith(('gemm_nt', 'gemm_tn')):
        return None
    return (label(path)[int(r['Dispatch_Id'])], n)


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ('sq', 'lds', 'tc', 'fetch', 'write'):
    for f in glob.glob(os.path.join(src, d, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = key(r, f)
            if k is None:
                continue
            a = acc[k][r['Counter_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(src, 'trace', '*', '*_kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        k = key(r, f)
        if k is None:
            continue
        d_ = dur[k]
        d_[0] += 1
        d_[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def mean(k, c):
    a = acc[k].get(c)
    return a[1] / a[0] if a and a[0] else float('nan')


def find_dur(k):
    v = dur.get(k)
    return v[1] / v[0] if v and v[0] else float('nan')


out = os.path.join(root, 'profiles', f'{tag}_gemm_pmc.csv')
with open(out, 'w') as f:
    f.write('# rocprofv3 --pmc passes over profiles/gemm_shapes_probe.py (isolated launches of the config-A step\'s contraction shapes,\n')
    f.write('# M = 65,536 rows, bf16); one counter group per pass, see profiles/collect_pmc.sh.  Column meanings: profiles/summarize_pmc.py.\n')
    f.write('shape,kernel,avg_us,wait_any_pct,wait_inst_pct,active_inst_pct,mfma_busy_pct,lds_active_pct,lds_conflict_pct,'
            'l2_hit_pct,tcp_pending_stall,l2_to_cu_MB,hbm_fetch_MB,hbm_write_MB\n')
    for k in sorted(acc, key=lambda k: SHAPES.index(k[0])):
        wc = mean(k, 'SQ_WAVE_CYCLES')
        hit, miss = mean(k, 'TCC_HIT_sum'), mean(k, 'TCC_MISS_sum')
        lds_act = mean(k, 'SQ_LDS_IDX_ACTIVE')
        f.write(f'"{k[0]}","{k[1]}",{find_dur(k):.1f},{100 * mean(k, "SQ_WAIT_ANY") / wc:.1f},{100 * mean(k, "SQ_WAIT_INST_ANY") / wc:.1f},'
                f'{100 * mean(k, "SQ_ACTIVE_INST_ANY") / wc:.1f},{100 * mean(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * mean(k, "SQ_BUSY_CU_CYCLES")):.1f},'
                f'{100 * lds_act / wc:.1f},{100 * mean(k, "SQ_LDS_BANK_CONFLICT") / max(lds_act, 1):.1f},{100 * hit / (hit + miss):.1f},'
                f'{mean(k, "TCP_PENDING_STALL_CYCLES_sum"):.3e},{mean(k, "TCP_TCC_READ_REQ_sum") * (64 if ", 32, " in k[1] else 128) / 1e6:.0f},'
                f'{2 * mean(k, "FETCH_SIZE") / 1024:.0f},{mean(k, "WRITE_SIZE") / 1024:.0f}\n')
print(open(out).read())
