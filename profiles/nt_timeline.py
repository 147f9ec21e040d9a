#!/usr/bin/env python3
"""Per-workgroup timeline of the NT GEMM kernel (diagnostic build only):

    make -C vit-ed_amd/csrc VARIANT=tl EXTRA=-DNT_TIMELINE
    VITED_LIB=$PWD/vit-ed_amd/libvited_hip_tl.so [TL_N=1536 TL_K=384 TL_EPI=6 VITED_NT_WM= VITED_NT_BK= VITED_NT_STAGES=] python3 profiles/nt_timeline.py

Every workgroup stamps s_memrealtime at its start, at the end of its K loop and at its end (plus HW_ID / XCC_ID); the script
prints the kernel span, the median K-loop and epilogue time per workgroup and how many workgroups are resident per CU.
Results of round 2: profiles/r02_nt_timeline.txt; what they mean: DESIGN.md section 6."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, '.')
import vited_amd as v
ops, L = v.ops, v._lib
lib = L.load()
dev = torch.device('cuda:0')
M = 65536
g = torch.Generator().manual_seed(0)
N, K, epi = int(os.environ.get('TL_N', 1536)), int(os.environ.get('TL_K', 384)), int(os.environ.get('TL_EPI', 6))
x = torch.randn(M, K, generator=g).to(dev).bfloat16(); w = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16(); b = torch.randn(N, generator=g).to(dev)
kw = dict(epilogue=epi, bias=b)
if epi == L.EPI_RESIDUAL: kw['residual'] = torch.randn(M, N, generator=g).to(dev)
if epi == L.EPI_MUL: kw['aux'] = torch.randn(M, N, generator=g).to(dev).bfloat16()
for _ in range(3): ops.gemm(x, w, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(x, w, **kw); e1.record(); torch.cuda.synchronize()
buf = np.zeros(16384 * 4, dtype=np.uint64)
lib.vited_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.vited_debug_timeline(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(-1, 4)
t = t[t[:, 2] > 0]
t = t[t[:, 0] >= t[:, 0].max() - 100000]        # this launch only (stale rows of earlier, larger grids are older)
t0 = t[:, 0].min()
start, mid, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
hw = (t[:, 3] & 0xffffffff).astype(np.int64); xcc = (t[:, 3] >> 32).astype(np.int64) & 0xf
cu = (xcc << 16) | (hw & 0xff00)
ids, counts = np.unique(cu, return_counts=True)
ts = np.linspace(0, end.max(), 300)
resident = np.array([((start <= x_) & (end > x_)).sum() for x_ in ts]) / len(ids)
print(f'{os.environ.get("TL_TAG","")}: N={N} K={K} epi={epi} WGs={len(t)} event {e0.elapsed_time(e1)*1e3:.1f} us span {end.max():.1f} us | per-WG loop {np.median(mid-start):.2f} epilogue {np.median(end-mid):.2f} us | resident WGs/CU {np.median(resident):.2f}')
