import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from makani_amd import ops
dev = torch.device("cuda:0")
C = 384
for nlat, nlon in ((721, 1440), (240, 480)):
    c = 2 * math.pi / nlon
    x = torch.rand(1, C, nlat, nlon, device=dev)
    for _ in range(3):
        F = ops.rfft_rows(x, 241, C, (c, c, c))
        y = ops.irfft_rows(F, 1, C, nlon, torch.float32, (1.0, 2.0, 1.0))
torch.cuda.synchronize()
