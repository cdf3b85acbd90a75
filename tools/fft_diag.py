"""Where the waves of the forward FFT kernels spend their cycles (library built with -DMK_FFT_DIAG=1: `bash tools/ab_fast.sh fft_fast
diag:-DMK_FFT_DIAG=1`, run with MAKANI_AMD_LIB pointing at it): s_memtime stamps at the phase boundaries, summed over all waves."""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.lib()
lib.mk_fft_diag_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
NAMES = ["prefetch issue", "pass loads (LDS reads landed)", "barriers", "twiddle + butterflies + LDS stores", "untangle + F stores issued",
         "commit (wait for the row vectors, convert, LDS stores)"]
for nlat, nlon, mmax, dt in ((721, 1440, 241, torch.bfloat16), (721, 1440, 241, torch.float32), (240, 480, 241, torch.bfloat16)):
    c = 2 * math.pi / nlon
    x = torch.rand(1, 384, nlat, nlon, device=dev).to(dt)
    ops.rfft_rows(x, mmax, 384, (c, c, c))
    torch.cuda.synchronize()
    lib.mk_fft_diag_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.rfft_rows(x, mmax, 384, (c, c, c))
    e1.record()
    torch.cuda.synchronize()
    lib.mk_fft_diag_read(buf, 1)
    v = [buf[k] for k in range(8)]
    waves = max(v[7], 1)
    print(f"rfft {nlat}x{nlon} {str(dt)[6:]}: {e0.elapsed_time(e1) * 1e3:.0f} us (instrumented), {waves} waves, mean cycles per wave {v[6] / waves:.0f}")
    for k in range(6):
        print(f"   {NAMES[k]:58s} {100.0 * v[k] / max(v[6], 1):5.1f} %")
