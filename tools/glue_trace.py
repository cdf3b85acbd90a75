"""Where do the torch kernels of the train step (adds, copies, fills: ~1 ms per step) come from?  torch.profiler with Python
stacks over ONE eager step of the benchmark's workload; prints, per torch GPU kernel, calls / time and the innermost frames of
this repository that launched it.
    python tools/glue_trace.py"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS["sfno_sc3_layers8_edim384"]
    H, W = cfg["inp_shape"]
    model = bench.build_model("sfno_sc3_layers8_edim384", dev, seed=333)
    opt = bench.make_optimizer(model)
    inp = torch.rand(1, cfg["inp_chans"], H, W, device=dev)
    tar = torch.rand(1, cfg["out_chans"], H, W, device=dev)
    loss_fn = bench.make_loss(H, W, cfg["out_chans"], dev, False)
    for _ in range(2):
        bench.train_step(model, opt, inp, tar, loss_fn, True, False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        bench.train_step(model, opt, inp, tar, loss_fn, True, False)
        torch.cuda.synchronize()
    rows = []
    for ka in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        dt = getattr(ka, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ka, "self_cuda_time_total", 0)
        if not dt or not ka.key.startswith("aten::"):
            continue
        frames = [f for f in (ka.stack or []) if "/repo/" in f and "glue_trace" not in f][:3]
        rows.append((dt, ka.count, ka.key, str(ka.input_shapes)[:90], " <- ".join(f.split("/repo/")[-1] for f in frames) or "(autograd engine)"))
    rows.sort(key=lambda r: -r[0])
    print(f"torch (aten) GPU work in one eager step: {sum(r[0] for r in rows) / 1e3:.3f} ms")
    for dt, n, name, shapes, where in rows[:45]:
        print(f"{dt / 1e3:7.3f} ms  {n:3d} x  {name:26s} {shapes}")
        print(f"              {where[:240]}")


if __name__ == "__main__":
    main()
