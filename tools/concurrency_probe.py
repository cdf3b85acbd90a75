"""Do weight-gradient kernels overlap with the data-gradient chain when issued on a second stream?
(one-off probe; prints sequential vs two-stream time for the pairs that are independent in backward)"""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makani_amd as ma  # noqa: E402
from makani_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


side = torch.cuda.Stream()


def both(f_main, f_side):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f_side()
        f_main()
        main.wait_stream(side)
    return run


for H, W in ((240, 480), (721, 1440)):
    N = H * W
    gy = torch.randn(1, 768, H, W, device=dev).bfloat16()
    x = torch.randn(1, 384, H, W, device=dev).bfloat16()
    w = torch.randn(768, 384, device=dev).bfloat16()
    f_wgrad = lambda: ops.conv1x1_wgrad(gy, x)
    f_dgrad = lambda: torch.mm(w.t(), gy.view(768, N))
    a, b = timeit(f_wgrad), timeit(f_dgrad)
    c = timeit(both(f_dgrad, f_wgrad))
    print(f"conv {H}x{W}: wgrad {a:.3f} + dgrad(lib) {b:.3f} = {a+b:.3f} ms sequential;  two streams {c:.3f} ms")

C, L, M = 384, 240, 241
S = torch.randn(L, M, 2, C, device=dev)
G = torch.randn(L, M, 2, C, device=dev)
wt = torch.randn(1, C, C, L, dtype=torch.complex64, device=dev)
Wl = ops.weight_to_wlayout(wt)
f_w = lambda: ops.dhconv_wgrad(S, G, 1)
f_d = lambda: ops.dhconv_dgrad(G, Wl, 1, C, C)
a, b = timeit(f_w), timeit(f_d)
c = timeit(both(f_d, f_w))
print(f"dhconv: wgrad {a:.3f} + dgrad {b:.3f} = {a+b:.3f} ms sequential;  two streams {c:.3f} ms")
