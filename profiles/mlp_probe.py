#!/usr/bin/env python3
"""Fused MLP branch (vited_mlp_fwd) against the unfused kernel sequence it replaces, M = 65,536 / 66,560 token rows (config A,
B = 1024), interleaved rounds in one process.   python3 profiles/mlp_probe.py [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vited_amd as v  # noqa: E402

ops, L = v.ops, v._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rows', type=int, nargs='*', default=[65536, 66560, 24576])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    gamma, beta = 1 + 0.1 * rnd(384), 0.1 * rnd(384)
    w1, b1, w2, b2 = rnd(1536, 384, scale=0.05).bfloat16(), 0.1 * rnd(1536), rnd(384, 1536, scale=0.03).bfloat16(), 0.1 * rnd(384)
    for M in a.rows:
        x = rnd(M, 384)

        def unfused():
            h, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, torch.bfloat16)
            gd, u = ops.gemm(h, w1, epilogue=L.EPI_GELU_GRAD, bias=b1)
            return ops.gemm(u, w2, epilogue=L.EPI_RESIDUAL, bias=b2, residual=x)

        variants = {'unfused (LN + fc1/GELU + fc2/res)': unfused,
                    'fused, saving for backward': lambda: ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2, 1e-6, save=True)[0],
                    'fused, inference': lambda: ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2, 1e-6, save=False)[0]}
        times = {k: [] for k in variants}
        for k, fn in variants.items():
            fn()
        torch.cuda.synchronize()
        for _ in range(a.reps):
            for k, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3)
        flops = 4.0 * M * 384 * 1536
        print(f'M = {M}')
        for k, t in times.items():
            t.sort()
            med = t[len(t) // 2]
            print(f'  {k:36s} median {med:8.1f} us  min {t[0]:8.1f} us  {flops / med / 1e6:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()
