"""Forward / data-gradient channel GEMMs at the per-rank pixel counts of the h x w split against the kernel choices of
mk_conv1x1_nn (weight-stationary default, MAKANI_AMD_CONV_NN=ring, =tile): time per launch, plain and with the fused epilogues.
    python tools/conv_shard_ab.py        (one process per setting: the choice is read once per process)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from makani_amd import ops
    dev = torch.device("cuda:0")

    def timeit(fn, reps=30, warm=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for (H, W) in ((60, 240), (60, 480), (178, 720), (240, 480)):
        for (M, K) in ((768, 384), (384, 768), (384, 384)):
            torch.manual_seed(M + K)
            x = (torch.rand(1, K, H, W, device=dev) - 0.5).bfloat16()
            w = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
            bias = torch.randn(M, device=dev)
            A = ops.pad_weight_bf16(w)
            gsrc = torch.randn(1, M, H, W, device=dev).bfloat16()
            t0 = timeit(lambda: ops.conv1x1_nn(A, K, x))
            t1 = timeit(lambda: ops.conv1x1_nn(A, K, x, bias=bias, act=True, want_pre=True))
            t2 = timeit(lambda: ops.conv1x1_nn(A, K, x, gelu_grad_of=gsrc))
            t3 = timeit(lambda: ops.conv1x1_nn(A, K, x, residual=gsrc))
            print(f"{os.environ.get('MAKANI_AMD_CONV_NN', 'default'):8s} M={M} K={K} {H * W:6d} px: plain {t0:6.1f}  +bias+gelu+pre {t1:6.1f}  *gelu' {t2:6.1f}  +R {t3:6.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for mode in ("", "ring", "tile"):
            env = dict(os.environ)
            if mode:
                env["MAKANI_AMD_CONV_NN"] = mode
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
