import os, sys, math
sys.path.insert(0, "/root/repo")
import torch
from makani_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
C = 384
for nlat, nlon in ((721, 1440), (240, 480)):
    c = 2 * math.pi / nlon
    x = torch.rand(1, C, nlat, nlon, device=dev)
    for mmax in (241, 64, 8):
        ms = timeit(lambda: ops.rfft_rows(x, mmax, C, (c, c, c)))
        F = ops.rfft_rows(x, mmax, C, (c, c, c))
        ms2 = timeit(lambda: ops.irfft_rows(F, 1, C, nlon, torch.float32, (1.0, 2.0, 1.0)))
        print(f"{nlat}x{nlon} mmax={mmax:3d}: rfft {ms:.3f} ms  irfft {ms2:.3f} ms")
