"""Weight-gradient launches at the per-rank pixel counts of the h x w split (14 400 = 60 x 240, 128 160 = 178 x 720) against the
full grids: time per launch for several values of the kernel's minimum pixels per split (MK_WGRAD_MINPX).
    python tools/wgrad_shard_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from makani_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (H, W) in ((60, 240), (60, 480), (178, 720), (240, 480)):
    for (M, K) in ((768, 384), (384, 768), (384, 384)):
        torch.manual_seed(M + K)
        x = (torch.rand(1, K, H, W, device=dev) - 0.3).bfloat16()
        g = (torch.randn(1, M, H, W, device=dev) * 0.5).bfloat16()
        ref = None
        row = []
        for minpx in ("1024", "512", "256", "128"):
            os.environ["MK_WGRAD_MINPX"] = minpx      # (read by a build with the knob compiled in: git show HEAD~1:makani_amd/csrc/conv1x1.hip)
            dW, db = ops.conv1x1_wgrad(g, x, want_bias=True)
            if ref is None:
                ref = dW.clone()
            err = float((dW - ref).norm() / ref.norm())
            us = timeit(lambda: ops.conv1x1_wgrad(g, x, want_bias=True))
            row.append(f"{minpx}: {us:6.1f} us (vs 1024: {err:.1e})")
        print(f"wgrad+bias M={M} K={K} {H}x{W} ({H * W} px): " + "  ".join(row), flush=True)
os.environ.pop("MK_WGRAD_MINPX", None)
