import sys, torch, time
sys.path.insert(0, '.')
import vited_amd as v
ops, L = v.ops, v._lib
dev = torch.device('cuda:0')
import os
M = int(os.environ.get("GB_M", 65536))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('NT gemm (M=%d)' % M)
for (N, K, epi, name) in [(1152, 384, L.EPI_STORE, 'qkv'), (384, 384, L.EPI_RESIDUAL, 'proj+res'), (1536, 384, L.EPI_GELU, 'fc1+gelu'),
                          (384, 1536, L.EPI_RESIDUAL, 'fc2+res'), (1536, 384, L.EPI_MUL_GELU_GRAD, 'dz=dy.W2*gelu\''), (384, 1536, L.EPI_STORE, 'dh=dz.W1'),
                          (384, 1152, L.EPI_STORE, 'dh=dqkv.Wqkv'), (384, 384, L.EPI_STORE, 'do=dx.Wp'), (768, 384, L.EPI_STORE, 'kv'), (1536, 384, L.EPI_STORE, 'fc1 plain')]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    kw = dict(epilogue=epi, bias=bias)
    if epi == L.EPI_RESIDUAL: kw['residual'] = torch.randn(M, N, device=dev)
    if epi == L.EPI_MUL_GELU_GRAD: kw['aux'] = torch.randn(M, N, device=dev).bfloat16(); kw['bias'] = None
    us = timeit(lambda: ops.gemm(a, w, **kw))
    fl = 2.0 * M * N * K
    byt = M * K * 2 + N * K * 2 + M * N * (2 if epi in (L.EPI_STORE,) else 4 if epi == L.EPI_GELU else 8 if epi == L.EPI_RESIDUAL else 4)
    print(f'  {name:18s} N={N:5d} K={K:5d} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {byt/us/1e3:7.1f} GB/s algorithmic')
print('TN dW (M=%d)' % M)
for (N, K, name) in [(1152, 384, 'dWqkv'), (384, 384, 'dWproj'), (1536, 384, 'dWfc1'), (384, 1536, 'dWfc2'), (768, 384, 'dWkv'), (384, 192, 'dWpatch')]:
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    us = timeit(lambda: ops.linear_bwd_weight(dy, x))
    us_nb = timeit(lambda: ops.linear_bwd_weight(dy, x, want_bias=False))
    fl = 2.0 * M * N * K
    print(f'  {name:18s} N={N:5d} K={K:5d} {us:8.1f} us ({us_nb:8.1f} w/o bias) {fl/us/1e6:7.1f} TF/s  {(M*(N+K)*2)/us/1e3:7.1f} GB/s algorithmic')
