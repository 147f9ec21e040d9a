#!/usr/bin/env python3
"""Short-sequence attention (attention_mfma.hip) forward / backward at the config-A step's shapes (batch 1024, 12 heads x 32).

    python3 profiles/attn_probe.py [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vited_amd as v  # noqa: E402

ops = v.ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    B, H, hd = a.batch, 12, 32
    D = H * hd

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev).bfloat16()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    for nq, nk in ((64, 64), (65, 65), (65, 64), (1, 65), (1, 64)):
        if nq == nk:
            qkv = rnd(B, nq, 3 * D)
            q, k, vv = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
            dqkv = torch.empty_like(qkv)
            dq, dk, dv = dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:]
        else:
            q, kv = rnd(B, nq, D), rnd(B, nk, 2 * D)
            k, vv = kv[:, :, :D], kv[:, :, D:]
            dq, dkv = torch.empty_like(q), torch.empty_like(kv)
            dk, dv = dkv[:, :, :D], dkv[:, :, D:]
        o, lse = ops.attention_fwd(q, k, vv, H, hd ** -0.5)
        do = rnd(B, nq, D)
        tf = timed(lambda: ops.attention_fwd(q, k, vv, H, hd ** -0.5))
        tb = timed(lambda: ops.attention_bwd(q, k, vv, o, do, lse, H, hd ** -0.5, dq, dk, dv))
        mb_f = (q.numel() + k.numel() + vv.numel() + o.numel()) * 2 / 1e6
        mb_b = (2 * (q.numel() + k.numel() + vv.numel()) + do.numel()) * 2 / 1e6
        print(f'Nq={nq:3d} Nk={nk:3d}  fwd {tf:7.1f} us ({mb_f / tf:5.2f} TB/s)   bwd {tb:7.1f} us ({mb_b / tb:5.2f} TB/s)')


if __name__ == '__main__':
    main()
