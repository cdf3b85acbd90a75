"""HBM rate of the fused AdamW kernel on spectral-weight sized tensors (8 x 70.8 M floats, as the SFNO's dhconv weights),
cycling over the eight so that nothing is cache-warm: python tools/adamw_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import _lib

dev = torch.device("cuda:0")
n = 384 * 384 * 240 * 2
sets = [[torch.randn(n, device=dev) for _ in range(4)] for _ in range(8)]
for s in sets:
    s[3].abs_()
lib = _lib.lib()


def step(s):
    _lib.check(lib.mk_adamw_step(_lib.c_vp(s[0].data_ptr()), _lib.c_vp(s[1].data_ptr()), _lib.c_vp(s[2].data_ptr()), _lib.c_vp(s[3].data_ptr()),
                                 n, None, 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, None, _lib.stream()), "adamw")


for s in sets:
    step(s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 5
for _ in range(reps):
    for s in sets:
        step(s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (reps * 8)
print(f"adamw {n / 1e6:.1f} M floats: {ms * 1e3:7.1f} us  {28.0 * n / ms / 1e6:7.1f} GB/s")
