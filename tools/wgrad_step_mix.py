"""The 29 channel-GEMM weight-gradient launches of one sfno_sc3_layers8_edim384 train step, in their per-step mix
(7 internal-grid blocks x 3 shapes, the full-resolution block, encoder / decoder / skips), for the PMC passes:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/wgrad_step_mix.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python tools/wgrad_step_mix.py

Mean bytes per dispatch of conv_wgrad_kernel + reduce_splits = HBM traffic per `conv1x1_wgrad` launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from makani_amd import ops

MIX = [  # (M = output channels, K = input channels, H, W, launches per step)
    (384, 768, 240, 480, 7), (768, 384, 240, 480, 7), (384, 384, 240, 480, 7),
    (384, 384, 721, 1440, 3), (384, 768, 721, 1440, 1), (768, 384, 721, 1440, 1),
    (73, 384, 721, 1440, 1), (384, 73, 721, 1440, 1), (73, 73, 721, 1440, 1),
]


def main():
    dev = torch.device("cuda:0")
    total = 0.0
    n = 0
    for M, K, H, W, mult in MIX:
        g = torch.randn(1, M, H, W, device=dev).bfloat16()
        x = torch.randn(1, K, H, W, device=dev).bfloat16()
        for _ in range(mult):
            ops.conv1x1_wgrad(g, x)
        total += mult * (2.0 * H * W * (M + K) + 4.0 * M * K)
        n += mult
        del g, x
    torch.cuda.synchronize()
    print(f"{n} launches, algorithmic bytes per launch (mean): {total / n / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
