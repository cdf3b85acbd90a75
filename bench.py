"""bench.py — SFNO train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = forward + backward + gradient clipping + AdamW update of
``sfno_sc3_layers8_edim384`` (config/sfnonet.yaml: scale_factor 3, 8 layers, embed_dim 384,
dhconv, instance norm, mlp_ratio 2) at 721 x 1440, 73 -> 73 channels, batch 1 per GPU,
bf16 autocast (fp32 spectral path), synthetic DummyLoader-shaped data resident in HBM.
N > 1: one process per GPU (torchrun env), data parallel over RCCL with the gradient
all-reduce overlapped with backward; "scaling": "weak".

Rank 0 prints ONE JSON line (see DESIGN.md §7 for every field).
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver (already exported on the boxes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

CONFIGS = {
    # BASELINE.json configs[1]
    "sfno_sc3_layers8_edim384": dict(inp_shape=(721, 1440), out_shape=(721, 1440), inp_chans=73, out_chans=73,
                                     scale_factor=3, embed_dim=384, num_layers=8, mlp_ratio=2, operator_type="dhconv",
                                     normalization_layer="instance_norm", activation_function="gelu", big_skip=True,
                                     model_grid_type="equiangular", sht_grid_type="legendre-gauss"),
    # small stand-in with the same structure (plumbing / CI on small GPUs)
    "sfno_debug": dict(inp_shape=(91, 180), out_shape=(91, 180), inp_chans=8, out_chans=8, scale_factor=3,
                       embed_dim=64, num_layers=4, mlp_ratio=2),
}
PEAK_F32_MFMA_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TF = 2500.0    # dense
PEAK_HBM_GBS = 8000.0


def make_loss(H, W, channels, device, spatial):
    """The config's training loss (config/sfnonet.yaml:43-48: type "l2", squared): GeometricLpLoss(p=2, squared=True)
    on the equiangular grid (makani/utils/losses/lp_loss.py:28-107), uniform channel weights, mean over (B, C) —
    on the HIP path (makani_amd.GeometricLpLoss: one fused kernel forward, one backward)."""
    import makani_amd as ma
    fn = ma.GeometricLpLoss(img_shape=(H, W), crop_shape=(H, W), crop_offset=(0, 0),
                            channel_names=[f"c{i}" for i in range(channels)], p=2.0, squared=True,
                            grid_type="equiangular", spatial_distributed=spatial).to(device)
    return lambda pred, tar: fn(pred, tar).mean()


class GradReducer:
    """Gradient reduction over RCCL, overlapped with backward: every parameter's gradient is all-reduced
    (async) from its post-accumulate hook — the eight 283 MB spectral weights are natural large messages,
    small parameters are flushed in one bucket per group.

    data parallel:   mean over the data group (all ranks when no model parallelism).
    spatial h x w:   SUM over the groups a parameter is shared across, as makani's hook does
                     (makani/mpu/mappings.py:460-523): spectral weights (l-sharded over h) over "w",
                     everything else over "spatial"; then the data-parallel mean."""

    def __init__(self, model, data_group=None, data_size=1, spatial_group=None, w_group=None):
        self.data_group, self.data_size = data_group, data_size
        self.spatial_group, self.w_group = spatial_group, w_group
        self.handles = []
        self.small = {}
        self.big_bytes = 8 << 20
        self.active = data_size > 1 or spatial_group is not None
        # RCCL averages inside the collective (ReduceOp.AVG): no scaling pass over the 2.3 GB of gradients afterwards.
        # Probed once on one element; gloo (CPU tests, N ranks on one GPU) has no AVG and keeps SUM + scale.
        self.avg = False
        if data_size > 1:
            try:
                probe = torch.ones(1, device=next(model.parameters()).device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=data_group)
                self.avg = abs(float(probe) - 1.0) < 1e-6
            except (RuntimeError, ValueError, NotImplementedError):
                self.avg = False
        if self.active:
            for name, p in model.named_parameters():
                p.register_post_accumulate_grad_hook(lambda q, n=name: self._hook(q, n))

    def _groups_for(self, name):
        out = []
        if self.spatial_group is not None:
            if name.endswith("filter.filter.weight"):
                if self.w_group is not None:
                    out.append((self.w_group, 1.0))
            else:
                out.append((self.spatial_group, 1.0))
        if self.data_size > 1:
            out.append((self.data_group, 1.0 / self.data_size))
        return out

    def _hook(self, p, name):
        g = torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad
        groups = self._groups_for(name)
        if not groups:
            return
        if g.numel() * g.element_size() >= self.big_bytes and len(groups) == 1:
            grp, scale = groups[0]
            op = dist.ReduceOp.SUM
            if self.avg and grp is self.data_group:
                op, scale = dist.ReduceOp.AVG, 1.0
            self.handles.append((dist.all_reduce(g, op=op, group=grp, async_op=True), g, scale))
        else:
            self.small.setdefault(tuple(id(x[0]) for x in groups), (groups, []))[1].append(p)

    def finish(self):
        if not self.active:
            return
        for groups, params in self.small.values():
            flat = torch.cat([(torch.view_as_real(q.grad) if q.grad.is_complex() else q.grad).reshape(-1) for q in params])
            for grp, scale in groups:
                if self.avg and grp is self.data_group:
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=grp)
                    continue
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=grp)
                if scale != 1.0:
                    flat.mul_(scale)
            off = 0
            for q in params:            # the reduced bucket becomes the gradients (views, no copy-back kernels)
                n = q.grad.numel() * (2 if q.grad.is_complex() else 1)
                piece = flat[off:off + n]
                q.grad = torch.view_as_complex(piece.view(*q.grad.shape, 2)) if q.grad.is_complex() else piece.view_as(q.grad)
                off += n
        self.small = {}
        for h, g, scale in self.handles:
            h.wait()
            if scale != 1.0:
                g.mul_(scale)
        self.handles = []


def kernel_family(name):
    """Launch names of the channel GEMMs carry their shape (``conv1x1_wgrad_m384_k768_n115200``): one kernel symbol,
    nine shapes per step.  The family is what rocprofv3 aggregates under one kernel name."""
    import re
    return re.sub(r"_m\d+_k\d+_n\d+$", "", name)


def dominant_family(summary):
    """(family, member launch names) with the largest accumulated time in a LaunchProfiler summary"""
    fam = {}
    for k, d in summary.items():
        fam.setdefault(kernel_family(k), []).append(k)
    if not fam:
        return None, []
    best = max(fam, key=lambda f: sum(summary[k]["ms_total"] for k in fam[f]))
    return best, sorted(fam[best])


def kernel_table(warm_prof, prof, steps):
    """per launch name: launches/step, mean duration, ms/step, dense TFLOP/s, algorithmic GB/s.  Numbers come from the
    timed region where a kernel was instrumented there, else from the fully profiled last warm-up step."""
    table, table_steps = (warm_prof, 1) if warm_prof else (prof, steps)
    kernels = {}
    for k, d in sorted(table.items(), key=lambda kv: -kv[1]["ms_total"]):
        nsteps = table_steps
        if k in prof:
            d, nsteps = prof[k], steps
        tf = d["flops"] / d["launches"] / (d["ms_avg"] * 1e-3) / 1e12 if d["flops"] else None
        gb = d["bytes"] / d["launches"] / (d["ms_avg"] * 1e-3) / 1e9
        kernels[k] = dict(launches_per_step=d["launches"] / nsteps, ms_avg=round(d["ms_avg"], 4),
                          ms_per_step=round(d["ms_total"] / nsteps, 3),
                          TFLOPs_dense=round(tf, 2) if tf else None, GBps_algorithmic=round(gb, 1))
    return kernels


def roofline_of(family, members, prof, gemm_mode, traffic):
    """The ``roofline`` object for one kernel family: algorithmic flops / bytes per launch (summed over the family's
    launches in ``prof``) over its mean launch duration.  Bound: HBM for kernels without matrix work; for the bf16
    channel GEMMs whichever roof the shape mix sits under (arithmetic intensity vs 2500 TF / 8 TB/s = 312 flop/B);
    the fp32 spectral GEMMs are priced against the matrix rate of the engine they run on."""
    ds = [prof[m] for m in members if m in prof]
    launches = sum(d["launches"] for d in ds)
    if not launches:
        return None
    ms_avg = sum(d["ms_total"] for d in ds) / launches
    flops = sum(d["flops"] for d in ds) / launches
    nbytes = sum(d["bytes"] for d in ds) / launches
    tf = flops / (ms_avg * 1e-3) / 1e12
    gb = nbytes / (ms_avg * 1e-3) / 1e9
    base = dict(kernel=family, launches=launches, ms_avg=round(ms_avg, 4), algorithmic_bytes=int(nbytes),
                algorithmic_flops=int(flops), traffic=traffic)
    if len(members) > 1:
        base["shapes"] = members
    if not flops:
        return dict(base, bound="hbm", achieved=round(gb, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gb / PEAK_HBM_GBS, 4))
    if family.startswith("conv1x1"):
        if flops / (PEAK_BF16_MFMA_TF * 1e12) < nbytes / (PEAK_HBM_GBS * 1e9):      # below the ridge: HBM is the roof
            return dict(base, bound="hbm", achieved=round(gb, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(gb / PEAK_HBM_GBS, 4), TFLOPs=round(tf, 1),
                        note=f"bf16 channel GEMM, {flops / nbytes:.0f} flop/B < 312 flop/B ridge: HBM-bound; "
                             f"{tf:.0f} TF = {tf / PEAK_BF16_MFMA_TF:.3f} of the bf16 MFMA peak")
        return dict(base, bound="mfma", achieved=round(tf, 2), peak=PEAK_BF16_MFMA_TF, unit="TFLOP/s",
                    frac=round(tf / PEAK_BF16_MFMA_TF, 4), GBps=round(gb, 1), note="bf16 MFMA")
    peak, eng = {"fp32": (PEAK_F32_MFMA_TF, "exact-fp32 MFMA"),
                 "x6": (PEAK_BF16_MFMA_TF / 6, "bf16 MFMA, 6 limb products per fp32 product"),
                 "x3": (PEAK_BF16_MFMA_TF / 3, "bf16 MFMA, 3 limb products per fp32 product")}[gemm_mode]
    return dict(base, bound="mfma", achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4),
                note=f"dense-formulation fp32-equivalent flops per launch / HIP-event launch time; engine: {eng}; "
                     f"the kernel skips the structurally-zero l<m half (DESIGN.md §4)")


def load_pmc_traffic():
    """HBM bytes per launch from the committed PMC passes (rocprofv3 counters cannot be collected from inside the timed
    run; profiles/r01_pmc_hbm_traffic.json says how they were taken)"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as fh:
            return {k: v["hbm_bytes"] for k, v in json.load(fh).items() if isinstance(v, dict) and "hbm_bytes" in v}
    except (OSError, ValueError):
        return {}


def parse_parallelism(par):
    """'dp' -> (1, 1); 'hHwW' -> (H, W)"""
    if par == "dp":
        return 1, 1
    import re
    mt = re.fullmatch(r"h(\d+)w(\d+)", par)
    if not mt:
        raise SystemExit(f"--parallelism {par!r}: expected 'dp' or 'hHwW'")
    return int(mt.group(1)), int(mt.group(2))


def build_groups(world, rank, ph, pw):
    """Process-group tree world -> data x (h x w) (makani/utils/comm.py:114-201): model instance d owns the ranks
    [d*h*w, (d+1)*h*w), laid out h-major; every rank creates every group (torch.distributed requires it) and keeps its
    own.  Returns (d_idx, ih, iw, data_group, spatial_group, h_group, w_group); groups of size 1 are None."""
    msize = ph * pw
    dsize = world // msize
    d_idx, m_idx = rank // msize, rank % msize
    ih, iw = m_idx // pw, m_idx % pw
    data_group = spatial_group = h_group = w_group = None
    if world > 1:
        for d in range(dsize):
            base = d * msize
            if msize > 1:
                g = dist.new_group(list(range(base, base + msize)))
                if d == d_idx:
                    spatial_group = g
                for j in range(pw):
                    g = dist.new_group([base + i * pw + j for i in range(ph)])
                    if d == d_idx and j == iw:
                        h_group = g
                for i in range(ph):
                    g = dist.new_group([base + i * pw + j for j in range(pw)])
                    if d == d_idx and i == ih:
                        w_group = g
        if dsize > 1:
            for m in range(msize):
                g = dist.new_group([d * msize + m for d in range(dsize)])
                if m == m_idx:
                    data_group = g
    return d_idx, ih, iw, data_group, spatial_group, h_group, w_group


def build_model(cfg_name, device, seed):
    import makani_amd as ma
    torch.manual_seed(seed)
    model = ma.SphericalFourierNeuralOperatorNet(**CONFIGS[cfg_name]).to(device)
    return model


def make_optimizer(model):
    # AdamW, betas (0.9, 0.95), lr 1e-3, weight decay 0 (config/sfnonet.yaml:50-54,114-117) as one fused
    # HIP pass per tensor; complex64 spectral weights are updated through their real view.
    from makani_amd.optim import FusedAdamW
    params = [p for p in model.parameters() if p.requires_grad]
    return FusedAdamW(params, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0)


def clip_grads(model, max_norm):
    """global-norm clipping (makani/utils/training/training_helpers.py:123-165), complex grads viewed as real"""
    grads = [torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad for p in model.parameters() if p.grad is not None]
    norms = torch._foreach_norm(grads)
    total = torch.linalg.vector_norm(torch.stack(norms))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


class ClipState:
    """global gradient norm when the spectral weights are sharded over the h group: their squared norms
    are summed over h, replicated parameters count once (training_helpers.py:123-165)."""

    def __init__(self, model, h_group):
        self.h_group = h_group
        self.sharded = [p for n, p in model.named_parameters() if n.endswith("filter.filter.weight")]
        self.repl = [p for n, p in model.named_parameters() if not n.endswith("filter.filter.weight")]

    def scale(self, max_norm):
        def sq(ps):
            gs = [torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad for p in ps if p.grad is not None]
            return torch.stack(torch._foreach_norm(gs)).square().sum() if gs else torch.zeros((), device="cuda")
        s = sq(self.sharded)
        dist.all_reduce(s, group=self.h_group)
        total = torch.sqrt(s + sq(self.repl))
        return torch.clamp(max_norm / (total + 1e-6), max=1.0).float().reshape(1)


def train_step(model, opt, reducer, inp, tar, loss_fn, amp, clip=None):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        pred = model(inp)
    loss = loss_fn(pred, tar)
    loss.backward()
    reducer.finish()
    if clip is not None and clip.h_group is not None:
        opt.step(grad_scale=clip.scale(32.0))
    else:
        opt.step(max_grad_norm=32.0)      # global-norm clipping folded into the AdamW pass
    return loss


def sht_bandwidth(device, reps=5):
    """Secondary metric 'fwd SHT GB/s' (BASELINE.md §2): S1 ERA5-shaped, S2 model-shaped."""
    import makani_amd as ma
    out = {}
    for name, C, lmax, mmax in (("S1_c73_L721_M721", 73, 721, 721), ("S2_c384_L240_M241", 384, 240, 241)):
        S = ma.RealSHT(721, 1440, lmax=lmax, mmax=mmax, grid="equiangular").to(device)
        x = torch.rand(1, C, 721, 1440, device=device)
        S(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            S(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nbytes = C * 721 * 1440 * 4 + C * lmax * mmax * 8
        out[name] = dict(ms=dt * 1e3, GBps=nbytes / dt / 1e9)
        del S, x
        torch.cuda.empty_cache()
    return out


# forward GFLOP per stage of sfno_sc3_layers8_edim384 at 721x1440 (BASELINE.md §2 / SURVEY.md Appendix B)
_STAGE_GF = {"mid_block": 42.6 + 68.23 + 135.9 + 34.0, "total": 4517.7}


def _cpu_worker(cfg_name):
    """child process: time ONE forward+backward of one internal-grid block of the oracle on the host cores"""
    from oracle import sfno as osf
    from oracle import sht as osht
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 32))
    torch.set_num_threads(threads)
    cfg = CONFIGS[cfg_name]
    H, W = cfg["inp_shape"]
    E, sf = cfg["embed_dim"], cfg["scale_factor"]
    h, w = H // sf, W // sf
    torch.manual_seed(333)
    trans = osht.RealSHT(h, w, lmax=h, mmax=w // 2 + 1, grid="legendre-gauss").float()
    itrans = osht.InverseRealSHT(h, w, lmax=h, mmax=w // 2 + 1, grid="legendre-gauss").float()
    blk = osf.NeuralOperatorBlock(trans, itrans, E, "dhconv", cfg["mlp_ratio"], torch.nn.GELU, False)
    x = torch.rand(1, E, h, w, requires_grad=True)
    blk(x).square().mean().backward()                      # warm-up (thread pools, allocator)
    reps = 3                                               # ~10-12 s of CPU work in total
    t0 = time.perf_counter()
    for _ in range(reps):
        x.grad = None
        blk(x).square().mean().backward()
    t = (time.perf_counter() - t0) / reps
    print(json.dumps(dict(t_mid=t, threads=threads, cores=cores, h=h, w=w, reps=reps)), flush=True)


def cpu_baseline(cfg_name, timeout_s=240):
    """Reference-equivalent CPU path: the oracle (the reference's model code restated over the restated
    torch-harmonics SHT), fp32, timed in a child process on this host's cores on a BOUNDED sample: one
    forward+backward of ONE internal-grid block (the unit the network repeats 6x; 280.7 of the 4517.7
    forward GFLOP of the step).  The step time is that measurement scaled by the FLOP ratio (x16.1);
    optimizer time is excluded (which favours the CPU number)."""
    import subprocess
    if cfg_name != "sfno_sc3_layers8_edim384":
        return None
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", cfg_name], capture_output=True,
                             text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:   # timeout or failure: report it, never stall the GPU benchmark
        return dict(value=None, unit="samples/s", cores=None, kind="port", sample=f"CPU baseline unavailable: {type(e).__name__}")
    scale = _STAGE_GF["total"] / _STAGE_GF["mid_block"]
    step = rec["t_mid"] * scale
    return dict(value=1.0 / step, unit="samples/s", cores=rec["threads"], kind="port",
                sample=f"oracle fp32 on {rec['threads']} host threads ({rec['cores']} cores visible): fwd+bwd of one "
                       f"internal-grid block ({rec['h']}x{rec['w']}, 384 ch), mean of {rec.get('reps', 1)} = {rec['t_mid']:.2f} s, scaled by the step/block "
                       f"FLOP ratio {scale:.1f} -> {step:.1f} s per step (optimizer excluded)",
                ms_per_step=step * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sfno_sc3_layers8_edim384", choices=list(CONFIGS))
    ap.add_argument("--fp32", action="store_true", help="disable bf16 autocast (not the BASELINE metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sht-metric", action="store_true")
    ap.add_argument("--parallelism", default=os.environ.get("MAKANI_AMD_PARALLELISM", "dp"),
                    help="'dp' (default: one sample per GPU, weak scaling) or 'hHwW' e.g. h4w2: spatial model "
                         "parallelism over H x W GPUs per model instance (strong scaling), remaining ranks data parallel")
    ap.add_argument("--multistep-count", type=int, default=1,
                    help="autoregressive rollout length per sample (makani's --multistep_count, BASELINE configs[4]); "
                         "1 = the headline single-step metric")
    ap.add_argument("--multistep-checkpoint", action="store_true",
                    help="recompute each rollout step's network call in backward (makani's --multistep_checkpoint)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        _cpu_worker(args.cpu_worker)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    backend = os.environ.get("MAKANI_AMD_BENCH_BACKEND", "nccl")    # "gloo": functional test of N ranks on one GPU
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from makani_amd import ops
    import makani_amd.distributed as thd

    # ---- process-group tree: world -> data x (h x w), as makani/utils/comm.py:114-201 ----
    ph, pw = parse_parallelism(args.parallelism)
    msize = ph * pw
    if world % msize:
        raise SystemExit(f"world size {world} is not a multiple of h*w = {msize}")
    dsize = world // msize
    d_idx, ih, iw, data_group, spatial_group, h_group, w_group = build_groups(world, rank, ph, pw)
    if msize > 1:
        thd.init(h_group if ph > 1 else None, w_group if pw > 1 else None, spatial_group)

    cfg = CONFIGS[args.config]
    H, W = cfg["inp_shape"]
    B = 1
    model = build_model(args.config, device, seed=333)            # same seed on every rank -> same init
    opt = make_optimizer(model)
    net = model
    if args.multistep_count > 1:                                   # makani/models/stepper.py:176-345
        from makani_amd.stepper import MultiStepWrapper
        net = MultiStepWrapper(model, n_future=args.multistep_count - 1, multistep_checkpoint=args.multistep_checkpoint).train()
    reducer = GradReducer(model, data_group, dsize, spatial_group if msize > 1 else None, w_group if pw > 1 else None)
    torch.manual_seed(333 + d_idx)                                 # DummyLoader: fixed U[0,1) tensors on device
    inp = torch.rand(B, cfg["inp_chans"], H, W, device=device)
    tar = torch.rand(B, cfg["out_chans"] * args.multistep_count, H, W, device=device)
    loss_fn = make_loss(H, W, cfg["out_chans"] * args.multistep_count, device, msize > 1)
    if msize > 1:                                                  # this rank's lat/lon shard (dataloaders shard likewise)
        lat0, lon0 = sum(model.trans_down.lat_shapes[:ih]), sum(model.trans_down.lon_shapes[:iw])
        hl, wl = model.inp_shape_loc
        inp = inp[..., lat0:lat0 + hl, lon0:lon0 + wl].contiguous()
        tar = tar[..., lat0:lat0 + hl, lon0:lon0 + wl].contiguous()
    amp = not args.fp32
    clip = ClipState(model, h_group if ph > 1 else None)

    # Per-kernel HIP events cost ~1.2 ms/step (2 events x ~300 C-ABI launches).  The LAST warm-up step is
    # profiled in full: it yields the per-kernel table and names the dominant HIP kernel; inside the timed region
    # only that kernel carries events (its roofline numbers therefore come from the timed steps).  With
    # --warmup 0 every launch of the timed region is instrumented instead.
    # Python's cyclic collector is paused for the warm-up and the timed loop: a generation-2 collection over the
    # autograd graphs stalls the launching thread for ~100 ms (measured with tools/step_times.py: one step of 154 ms
    # among 54 ms steps), enough to drain the GPU queue.  Reference counting still frees every tensor immediately.
    import gc
    gc.collect()
    gc.disable()
    warm_prof, dom_family, dom_members = {}, None, []
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last:
            torch.cuda.synchronize()
            ops.PROFILER.reset()
            ops.PROFILER.enabled = True
        train_step(net, opt, reducer, inp, tar, loss_fn, amp, clip)
        if last:
            torch.cuda.synchronize()
            ops.PROFILER.enabled = False
            warm_prof = ops.PROFILER.summary()
            dom_family, dom_members = dominant_family(warm_prof)
    torch.cuda.synchronize()

    ops.PROFILER.reset()
    ops.PROFILER.enabled = True
    ops.PROFILER.only = set(dom_members) if dom_members else None      # HIP events on the dominant kernel only (see above)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(net, opt, reducer, inp, tar, loss_fn, amp, clip)
    torch.cuda.synchronize()
    gc.enable()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.PROFILER.enabled = False

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = float(loss.detach())

    if rank == 0:
        prof = ops.PROFILER.summary()           # timed region: the dominant kernel (or everything with --warmup 0)
        # dominant HIP kernel = the kernel family with the largest accumulated time among our launches (what the
        # rocprofv3 --stats table of the same command shows on top); its numbers come from the timed steps
        if not (dom_family and any(m in prof for m in dom_members)):
            dom_family, dom_members = dominant_family(prof)
        kernels = kernel_table(warm_prof, prof, args.steps)
        pmc = load_pmc_traffic() if (args.config == "sfno_sc3_layers8_edim384" and msize == 1) else {}
        roofline = roofline_of(dom_family, dom_members, prof, ops.GEMM_MODE, pmc.get(dom_family)) if dom_family else None
        # the runners-up, from the fully profiled warm-up step (one launch set, not an average over the timed steps)
        others = []
        if warm_prof:
            fams = {}
            for k in warm_prof:
                fams.setdefault(kernel_family(k), []).append(k)
            rank_f = sorted(fams, key=lambda f: -sum(warm_prof[k]["ms_total"] for k in fams[f]))
            for f in [f for f in rank_f if f != dom_family][:4]:
                others.append(roofline_of(f, sorted(fams[f]), warm_prof, ops.GEMM_MODE, pmc.get(f)))
        hip_ms = sum(v["ms_per_step"] for v in kernels.values())
        out = {
            "metric": f"SFNO train samples/sec at {H}x{W}x{cfg['inp_chans']}ch",
            "value": dsize * B * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if msize == 1 else "strong",
            "vs_baseline": None,
            "dtype": "bf16" if amp else "f32",
            "data": "synthetic",
            "config": {"workload": args.config, "grid": f"{H}x{W}", "channels": cfg["inp_chans"],
                       "global_batch": dsize * B, "parallelism": f"dp{dsize}" + (f"_h{ph}w{pw}" if msize > 1 else ""),
                       "amp": "bf16 autocast, fp32 SHT/contraction" if amp else "fp32",
                       "multistep_count": args.multistep_count,
                       "multistep_checkpoint": bool(args.multistep_checkpoint)},
            "roofline": roofline,
            "roofline_runners_up": others,
            "peak_hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
            "hip_kernels": kernels,
            "hip_kernel_ms_per_step": round(hip_ms, 2),
            "final_loss": final_loss,
        }
        if world == 1 and not args.no_sht_metric and args.config == "sfno_sc3_layers8_edim384":
            del model, net, opt
            torch.cuda.empty_cache()
            print("[bench] train loop done; measuring fwd SHT", file=sys.stderr, flush=True)
            out["fwd_sht"] = sht_bandwidth(device)
            print("[bench] fwd SHT done; CPU baseline", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
