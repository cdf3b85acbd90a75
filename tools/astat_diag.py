"""Where the waves of the weight-stationary channel GEMM spend their cycles (needs a library built with -DMK_ASTAT_DIAG=1:
`tools/ab_fast.sh conv1x1 adiag:-DMK_ASTAT_DIAG=1`, run with MAKANI_AMD_LIB pointing at it): s_memtime stamps at the segment
boundaries of every pixel tile, summed over all waves (csrc/conv1x1.hip: MK_AS_STAMP)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.lib()
lib.mk_astat_diag_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
NAMES = ["wait for chunk + barrier", "fragment reads + MFMAs", "DMA issue", "epilogue: convert + stage", "epilogue: barrier",
         "epilogue: read back + math + stores", "epilogue: end barrier", "prologue (weights)"]
shapes = ((384, 384, 721, 1440), (768, 384, 721, 1440), (768, 384, 240, 480), (384, 384, 240, 480), (384, 73, 721, 1440))
if os.environ.get("MAKANI_AMD_ASTAT2", "0") != "0":      # the two-group kernel: segments 0 wait + barrier, 1 multiply, 2 requests, 3 stage, 5 read back
    shapes = shapes[:3]
for (M, K, H, W) in shapes:
    torch.manual_seed(M + K)
    x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
    w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(M, device=dev)
    A = ops.pad_weight_bf16(w)
    g = torch.randn(1, M, H, W, device=dev).bfloat16()
    for name, fn in (("plain", lambda: ops.conv1x1_nn(A, K, x)),
                     ("+bias+gelu+pre", lambda: ops.conv1x1_nn(A, K, x, bias=bias, act=True, want_pre=True)),
                     ("*gelu'(G)", lambda: ops.conv1x1_nn(A, K, x, gelu_grad_of=g)),
                     ("+R", lambda: ops.conv1x1_nn(A, K, x, residual=g))):
        fn()
        torch.cuda.synchronize()
        lib.mk_astat_diag_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        lib.mk_astat_diag_read(buf, 1)
        v = [buf[k] for k in range(11)]
        waves, tiles = max(v[9], 1), max(v[10], 1)
        print(f"M={M} K={K} {H}x{W} {name}: {e0.elapsed_time(e1) * 1e3:.0f} us (instrumented), {waves} waves, "
              f"{v[8] / waves:.0f} cycles per wave, {v[8] / tiles:.0f} per tile")
        print("    " + ", ".join(f"{NAMES[k].split(' (')[0]} {100.0 * v[k] / max(v[8], 1):.1f}%" for k in range(8)))
    del x, g
