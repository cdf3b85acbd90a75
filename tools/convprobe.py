"""A few launches of the forward channel GEMM at one shape (for rocprofv3 --pmc passes):
   python tools/convprobe.py M K H W [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from makani_amd import ops  # noqa: E402

M, K, H, W = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda", 0)
x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
A = ops.pad_weight_bf16(w)
g = torch.randn(1, M, H, W, device=dev).bfloat16()
for _ in range(reps):
    ops.conv1x1_nn(A, K, x)
    ops.conv1x1_wgrad(g, x)
torch.cuda.synchronize()
