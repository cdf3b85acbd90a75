"""Where the waves of the complex split-bf16 kernel spend their cycles (needs a library built with -DMK_X2_DIAG=32, e.g.
`python tools/ab.py build t32:-DMK_X2_DIAG=32`, run with MAKANI_AMD_LIB pointing at it): s_memtime stamps at the segment
boundaries of every k-step, summed per wave group (group 0 splits first and multiplies second, group 1 the other way round)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import _lib, ops

dev = torch.device("cuda:0")
C, L, M = 384, 240, 241
S = torch.randn(L, M, 2, C, device=dev)
G = torch.randn(L, M, 2, C, device=dev)
w = ops.native_w_empty(C, C, L, dev)
w.copy_(torch.randn(1, C, C, L, dtype=torch.complex64, device=dev))
lib = _lib.lib()
lib.mk_x2_diag_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
NAMES = ["produce (split + LDS stores + global loads)", "fragment reads (wait)", "MFMA segment", "barrier", "prologue", "epilogue",
         "whole kernel", "waves"]
for name, fn in (("dhconv fwd", lambda: ops.dhconv_fwd(S, w, 1, C)), ("dhconv dgrad", lambda: ops.dhconv_dgrad(G, w, 1, C, C)),
                 ("dhconv wgrad", lambda: ops.dhconv_wgrad(S, G, 1, native=True))):
    fn()
    torch.cuda.synchronize()
    lib.mk_x2_diag_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    lib.mk_x2_diag_read(buf, 1)
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:.0f} us (instrumented)")
    for g in range(2):
        v = [buf[g * 8 + k] for k in range(8)]
        waves = max(v[7], 1)
        print(f"  group {g}: {waves} waves, mean cycles per wave {v[6] / waves:9.0f}: " +
              ", ".join(f"{NAMES[k].split(' (')[0]} {100.0 * v[k] / max(v[6], 1):.1f}%" for k in range(6)))
