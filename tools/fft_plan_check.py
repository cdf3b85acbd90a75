"""Outputs of the specialised FFT kernels on seeded inputs, written to a file: run once per library build (MAKANI_AMD_LIB) and
compare — a change of the LDS layouts (csrc/fft_fast.hip LdsPlan, MK_FFT_LDSPLAN=0 = one layout) must leave every bit alone.
   python tools/fft_plan_check.py out.pt            # writes
   python tools/fft_plan_check.py a.pt b.pt         # compares two files"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print(f"{len(a)} tensors compared, {len(bad)} differ" + (": " + ", ".join(bad) if bad else " (bit-identical)"))
    sys.exit(1 if bad else 0)

from makani_amd import ops

dev = torch.device("cuda:0")
out = {}
torch.manual_seed(11)
for nlat, nlon, mmax, B, C in ((37, 1440, 241, 1, 52), (19, 1440, 721, 1, 24), (5, 1440, 241, 2, 9), (33, 480, 241, 1, 96), (21, 480, 81, 1, 40),
                               (7, 480, 241, 2, 7), (9, 720, 361, 1, 20), (8, 360, 181, 1, 16)):
    c = 2 * math.pi / nlon
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, C, nlat, nlon, device=dev).to(dt)
        Cp = (C + 3) // 4 * 4
        F = ops.rfft_rows(x, mmax, Cp, (c, 0.7 * c, 1.3 * c))
        y = ops.irfft_rows(F, B, C, nlon, dt, (1.0, 2.0, 0.5))
        tag = f"{nlat}x{nlon}_m{mmax}_B{B}C{C}_{str(dt)[6:]}"
        cols = torch.tensor([b * Cp + ch for b in range(B) for ch in range(C)], device=dev)      # (padding columns are never written)
        out["F_" + tag] = F[..., cols].cpu()
        out["y_" + tag] = y.float().cpu()
        ref = torch.fft.rfft(x.double(), dim=-1)[..., :mmax].permute(3, 2, 0, 1).reshape(mmax, nlat, B * C)
        got = torch.complex(F[:, :, 0, :][..., cols].double(), F[:, :, 1, :][..., cols].double())
        wv = torch.full((mmax,), 0.7 * c, dtype=torch.float64, device=dev)
        wv[0] = c
        if mmax == nlon // 2 + 1:
            wv[-1] = 1.3 * c
        err = ((got - ref * wv[:, None, None]).norm() / (ref * wv[:, None, None]).norm()).item()
        worst = max(globals().get("worst", 0.0), err)                               # (bf16 input: the rounded input IS the operand)
torch.save(out, sys.argv[1])
print(f"wrote {len(out)} tensors to {sys.argv[1]}; forward transforms vs torch.fft (fp64): worst rel-L2 {worst:.2e}")
