"""One training step (forward + backward, bf16 autocast) of FourCastNet3 at BASELINE config 4's architecture
(config/fourcastnet3.yaml: fcn3_sc2_edim45_layers10, 721x1440, 72 channels, B = 1) on ONE MI355X, per-kernel-family
times from HIP events.  The reference trains this configuration with h=2 x w=2 model parallelism on 80 GB devices; one
MI355X holds it whole.   python tools/fcn3_step.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import makani_amd as ma
from makani_amd import ops

LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
CHANS = ["u10m", "v10m", "u100m", "v100m", "t2m", "msl", "tcwv"] + [f"{v}{l}" for v in "uvztq" for l in LEVELS]
AUX = ["xzen", "xoro", "xlsml", "xlsms"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(333)
    t0 = time.time()
    model = ma.AtmoSphericNeuralOperatorNet(
        inp_shape=(721, 1440), out_shape=(721, 1440), scale_factor=2, filter_basis_type="morlet", kernel_shape=(3, 3),
        channel_names=CHANS, aux_channel_names=AUX, atmo_embed_dim=45, surf_embed_dim=56, aux_embed_dim=36, num_layers=10,
        sfno_block_frequency=5, normalization_layer="none", use_mlp=True, mlp_ratio=2, activation_function="gelu",
        big_skip=False, bias=False, encoder_mlp=False).to(dev)
    build_s = time.time() - t0
    nparam = sum(p.numel() * (2 if p.is_complex() else 1) for p in model.parameters())
    x = torch.rand(1, len(CHANS) + len(AUX), 721, 1440, device=dev)
    tar = torch.rand(1, len(CHANS), 721, 1440, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        loss = ((y.float() - tar) ** 2).mean()
        loss.backward()
        for p in model.parameters():
            p.grad = None
        return loss

    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    ops.PROFILER.enabled = True
    ops.PROFILER.reset()
    step()
    torch.cuda.synchronize()
    fam = {k: dict(launches=v["launches"], ms_total=round(v["ms_total"], 3)) for k, v in
           sorted(ops.PROFILER.summary().items(), key=lambda kv: -kv[1]["ms_total"])}
    print(json.dumps(dict(workload="fcn3_sc2_edim45_layers10 (config 4 architecture), B=1, one GPU, fwd+bwd, bf16 autocast",
                          ms_per_step=round(ms, 2), samples_per_s=round(1000.0 / ms, 3), loss=float(loss), real_params=nparam,
                          build_s=round(build_s, 1), peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 1e9, 2),
                          hip_kernel_families_ms=fam)))


if __name__ == "__main__":
    main()
