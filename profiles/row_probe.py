#!/usr/bin/env python3
"""Fused Linear + LayerNorm (gemm_row.hip) against the two-kernel forms it replaces, at the config-A step's shapes.

    python3 profiles/row_probe.py [--reps 10]

forward : y = res + a W^T + b ; h = LN(y)          vs  vited_gemm(RESIDUAL) + vited_layernorm_fwd
backward: dx = dx_in + LN'(dy Wt^T) (+ bf16 copy)  vs  vited_gemm(STORE) + vited_layernorm_bwd
"""
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
    ap.add_argument('--rows', type=int, nargs='*', default=[65536, 66560])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

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

    N = 384
    print(f'{"shape":28s} {"two kernels":>12s} {"fused":>9s}   GB/s (fused, algorithmic)')
    for M in a.rows:
        gamma, beta, bias = 1.0 + rnd(N, scale=0.1), rnd(N, scale=0.1), rnd(N, scale=0.1)
        res = rnd(M, N)
        mean, rstd = rnd(M, scale=0.1), 1.0 + rnd(M, scale=0.1).abs()
        for K in (384, 768, 1152, 1536):
            x_lp, w = rnd(M, K).bfloat16(), rnd(N, K, scale=K ** -0.5).bfloat16()

            def fwd_two():
                y = ops.gemm(x_lp, w, epilogue=L.EPI_RESIDUAL, bias=bias, residual=res)
                return ops.layernorm_fwd(y, gamma, beta, 1e-6, torch.bfloat16)

            def fwd_fused():
                return ops.linear_residual_layernorm_fwd(x_lp, w, bias, res, gamma, beta, 1e-6)

            def bwd_two():
                dh = ops.gemm(x_lp, w)
                return ops.layernorm_bwd(dh, res, gamma, mean, rstd, dx_in=res, want_lp=True)

            def bwd_fused():
                return ops.linear_layernorm_bwd(x_lp, w, res, gamma, mean, rstd, dx_in=res, want_lp=True)

            t2, t1 = timed(fwd_two), timed(fwd_fused)
            byt = M * K * 2 + N * K * 2 + M * N * (4 + 4 + 2)
            print(f'fwd M={M} K={K:5d}          {t2:10.1f}us {t1:8.1f}us   {byt / t1 / 1e3:7.0f}   {2.0 * M * N * K / t1 / 1e6:6.0f} TF/s')
            t2, t1 = timed(bwd_two), timed(bwd_fused)
            byt = M * K * 2 + N * K * 2 + M * N * (4 + 4 + 4 + 2)
            print(f'bwd M={M} K={K:5d}          {t2:10.1f}us {t1:8.1f}us   {byt / t1 / 1e3:7.0f}   {2.0 * M * N * K / t1 / 1e6:6.0f} TF/s')


if __name__ == '__main__':
    main()
