"""Yardstick only (NOT a product path): how long does the vendor library's bf16 GEMM (torch.mm -> hipBLASLt) take on the channel-GEMM
shapes of the train step, next to this package's hand-written kernels?  Says how much headroom the kernels of csrc/conv1x1.hip have.
    python tools/gemm_yardstick.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from makani_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (M, K, H, W) in ((768, 384, 240, 480), (384, 768, 240, 480), (384, 384, 240, 480), (768, 384, 721, 1440), (384, 768, 721, 1440),
                     (384, 384, 721, 1440)):
    N = H * W
    x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
    w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    g = torch.randn(1, M, H, W, device=dev).bfloat16()
    A = ops.pad_weight_bf16(w)
    x2, g2 = x.view(K, N), g.view(M, N)
    t_ours = timeit(lambda: ops.conv1x1_nn(A, K, x))
    t_lib = timeit(lambda: torch.mm(w, x2))
    t_wg = timeit(lambda: ops.conv1x1_wgrad(g, x))
    t_wg_lib = timeit(lambda: torch.mm(g2, x2.t()))
    gf = 2.0 * M * K * N / 1e9
    mb = 2.0 * N * (M + K) / 1e6
    print(f"m{M} k{K} n{N}: forward ours {t_ours:7.1f} us ({gf / t_ours * 1e3:6.0f} TF, {mb / t_ours / 1e3:5.2f} TB/s)  library {t_lib:7.1f} us ({gf / t_lib * 1e3:6.0f} TF) | "
          f"weight gradient ours {t_wg:7.1f} us  library (bf16 output) {t_wg_lib:7.1f} us", flush=True)
