"""wall time of the first train steps (allocator / heuristic warm-up profile of bench.py's step)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["sfno_sc3_layers8_edim384"]
model = bench.build_model("sfno_sc3_layers8_edim384", dev, 333)
opt = bench.make_optimizer(model)
red = bench.GradReducer(model)
H, W = cfg["inp_shape"]
inp = torch.rand(1, cfg["inp_chans"], H, W, device=dev)
tar = torch.rand(1, cfg["out_chans"], H, W, device=dev)
loss_fn = bench.make_loss(H, W, cfg["out_chans"], dev, False)
clip = bench.ClipState(model, None)
import gc
if os.environ.get('NOGC') == '1':
    gc.disable()
for i in range(9):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.train_step(model, opt, red, inp, tar, loss_fn, True, clip)
    torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms, reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB")
