#!/usr/bin/env python3
"""Flash attention kernels (config H: 6 heads x 64, 1024 / 1025 tokens) in isolation: timing table, or the workload of a PMC pass.
    python3 profiles/flash_probe.py [--batch 72] [--reps 6] [--no-time]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vited_amd as v  # noqa: E402

ops = v.ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=72)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--tokens', type=int, default=1024, help='1025 = the decoder\'s sequence (cls + 1024 patches): a one-token tail tile')
    ap.add_argument('--no-time', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, H, N, hd = a.batch, 6, a.tokens, 64
    D = H * hd
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 3 * D, generator=g).to(dev).bfloat16()
    q, k, vv = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    o, lse = ops.attention_fwd(q, k, vv, H, hd ** -0.5)
    do = torch.randn(B, N, D, generator=g).to(dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    fwd = lambda: ops.attention_fwd(q, k, vv, H, hd ** -0.5)
    bwd = lambda: ops.attention_bwd(q, k, vv, o, do, lse, H, hd ** -0.5, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:])
    fl = 4.0 * B * H * N * N * hd
    for name, fn, f in (('fwd', fwd, fl), ('bwd', bwd, 2.5 * fl)):
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
            print(f'flash {name} B={B} N={N}: {us:.1f} us {f / us / 1e6:.0f} TFLOP/s')


if __name__ == '__main__':
    main()
