"""dhconv time as a function of the contraction length (fixed-overhead vs per-k-tile cost of the complex GEMM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


L, M = 240, 241
for cin, cout in ((384, 48), (384, 96), (384, 192), (384, 384), (384, 768)):
    G = torch.randn(L, M, 2, cout, device=dev)
    S = torch.randn(L, M, 2, cin, device=dev)
    w = torch.randn(1, cin, cout, L, dtype=torch.complex64, device=dev)
    W = ops.weight_to_wlayout(w)
    td = timeit(lambda: ops.dhconv_dgrad(G, W, 1, cin, cout))          # K = cout
    print(f"dgrad  N=cin={cin} K=cout={cout:4d}: {td:.3f} ms   ({cout // 16} k-tiles)")
for cin, cout in ((48, 384), (96, 384), (192, 384), (384, 384), (768, 384)):
    S = torch.randn(L, M, 2, cin, device=dev)
    w = torch.randn(1, cin, cout, L, dtype=torch.complex64, device=dev)
    W = ops.weight_to_wlayout(w)
    tf = timeit(lambda: ops.dhconv_fwd(S, W, 1, cin))                   # K = cin
    print(f"fwd    N=cout={cout} K=cin={cin:4d}: {tf:.3f} ms   ({cin // 16} k-tiles)")
