#!/usr/bin/env python3
"""Isolated launches of every contraction shape of one config-A training step (M = 65,536 / 66,560 rows), the
workload the PMC passes of profiles/collect_pmc.sh run under rocprofv3.  Prints a timing table when run alone.

    python3 profiles/gemm_shapes_probe.py [--rows 65536] [--reps 5] [--no-time]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vited_amd as v  # noqa: E402

ops, L = v.ops, v._lib

NT = [(1152, 384, L.EPI_STORE, 'qkv'), (384, 384, L.EPI_RESIDUAL, 'proj+res'), (1536, 384, L.EPI_GELU_GRAD, "fc1+gelu'+gelu"),
      (384, 1536, L.EPI_RESIDUAL, 'fc2+res'), (1536, 384, L.EPI_MUL, "dz=dy.W2*gelu'"), (384, 1536, L.EPI_STORE, 'dh=dz.W1'),
      (384, 1152, L.EPI_STORE, 'dh=dqkv.Wqkv'), (384, 384, L.EPI_STORE, 'do=dx.Wp'), (768, 384, L.EPI_STORE, 'kv')]
TN = [(1152, 384, 'dWqkv'), (384, 384, 'dWproj'), (1536, 384, 'dWfc1'), (384, 1536, 'dWfc2'), (768, 384, 'dWkv')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=65536)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--no-time', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    M = a.rows
    g = torch.Generator(device='cpu').manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    def timed(fn, label, flops):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        if not a.no_time:
            print(f'  {label:18s} {us:8.1f} us {flops / us / 1e6:7.1f} TFLOP/s')

    print(f'NT (M = {M})')
    for n, k, epi, name in NT:
        x, w, bias = rnd(M, k).bfloat16(), rnd(n, k, scale=0.05).bfloat16(), rnd(n)
        kw = dict(epilogue=epi, bias=bias)
        if epi == L.EPI_RESIDUAL:
            kw['residual'] = rnd(M, n)
        if epi in (L.EPI_MUL_GELU_GRAD, L.EPI_MUL):
            kw['aux'], kw['bias'] = rnd(M, n).bfloat16(), None
        timed(lambda: ops.gemm(x, w, **kw), f'{name} N={n} K={k}', 2.0 * M * n * k)
    print(f'TN (M = {M})')
    for n, k, name in TN:
        dy, x = rnd(M, n).bfloat16(), rnd(M, k).bfloat16()
        timed(lambda: ops.linear_bwd_weight(dy, x), f'{name} N={n} K={k}', 2.0 * M * n * k)


if __name__ == '__main__':
    main()
