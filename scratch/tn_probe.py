import sys, torch, math
sys.path.insert(0, '.')
import vited_amd as v
ops = v.ops
dev = torch.device('cuda:0')
for (M, N, K) in [(12, 32, 3072), (12, 32, 128), (64, 32, 3072), (12, 128, 3072), (12, 32, 256), (16,32,256), (12,128,128), (12, 32, 384)]:
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dw, db = ops.linear_bwd_weight(dy, x)
    ref = dy.double().t() @ x.double()
    err = (dw.double() - ref).abs().max().item()
    # locate wrong columns
    bad = ((dw.double() - ref).abs() > 1e-2).nonzero()
    print(M, N, K, 'path', ops.last_paths()[0], 'maxerr', err, 'nbad', bad.shape[0], bad[:4].tolist(), bad[-2:].tolist())
