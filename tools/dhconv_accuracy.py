"""rel-L2 of the dhconv forward / data gradient against a complex128 einsum (run once per MAKANI_AMD_X2_TALL setting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(3)
B, C, L, M = 1, 384, 24, 241
w = torch.randn(1, C, C, L, dtype=torch.complex64, device=dev)
wn = ops.native_w_empty(C, C, L, dev).copy_(w)
live = (torch.arange(L, device=dev)[:, None] + 217 >= torch.arange(M, device=dev)[None, :])[:, :, None]      # degrees 217 .. 240
S = torch.randn(L, M, 2, C, device=dev) * live[..., None]
x = torch.complex(S[:, :, 0].double(), S[:, :, 1].double())
cplx = lambda T: torch.where(live, torch.complex(T[:, :, 0], T[:, :, 1]), torch.zeros((), dtype=torch.complex64, device=dev))
y = cplx(ops.dhconv_fwd(S, wn, B, C, tri_off=217))
gx = cplx(ops.dhconv_dgrad(S, wn, B, C, C, tri_off=217))
ref = torch.einsum("lmi,iol->lmo", x, w[0].to(torch.complex128))
refg = torch.einsum("lmo,iol->lmi", x, w[0].conj().to(torch.complex128))
e = lambda a, b: ((a - b).norm() / b.norm()).item()
print(f"MAKANI_AMD_X2_TALL={os.environ.get('MAKANI_AMD_X2_TALL', '(default)')}: fwd {e(y, ref):.3e}  dgrad {e(gx, refg):.3e}  (rel-L2 vs complex128)")
