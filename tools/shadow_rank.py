"""ONE rank of an h x w model instance ALONE on the GPU: the per-rank kernels of BASELINE configs[2] / [4] at their real
shard shapes, timed without a second GPU.

    python tools/shadow_rank.py --h 4 --w 2 [--ih 3 --iw 0] [--steps 3] [--fp32] [--multistep-count 4] [--json out.json]

The rank builds the distributed network exactly as ``bench.py`` does (process-group tree -> ``thd.init`` -> l-sharded
spectral weights, m-sharded Legendre matrices, fused exchange schedule, gradient hooks, fused AdamW) but every collective
is a PHANTOM: the group objects report the real sizes / ranks, an all-to-all copies the rank's own slab and leaves the
peers' slabs as they are, reductions and gathers return at once.  The numbers that come out of the kernels mean nothing;
their launch shapes, counts and durations are exactly the rank's (the kernels are data-independent), and nothing else
runs on the GPU — unlike N ranks time-slicing one GPU over gloo.  Reported per step:

  * the per-kernel table (HIP events around every C-ABI launch: makani_amd.ops.PROFILER) and its sum = GPU compute;
  * wall time of eager steps with phantom collectives = the HOST's launch path (Python + ctypes + torch glue): the step
    cannot be faster than this however fast the links are;
  * what this rank puts on the links (``thd.COMM_STATS``: bytes sent and all-to-alls per group size) and the time that
    takes at 153 GB/s per xGMI link with all peers' links in parallel (SURVEY.md §8e's budget model).

h = w = 1 runs the serial model through the same harness (no phantoms): the baseline the shard efficiencies divide by.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
import torch.distributed as real_dist  # noqa: E402

LINK_GBS = 153.0          # one xGMI link, one direction


class PhantomGroup:
    def __init__(self, name, size, rank):
        self.name, self.size, self.rank = name, size, rank

    def __repr__(self):
        return f"PhantomGroup({self.name}, size {self.size}, rank {self.rank})"


class _Done:
    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


def phantom_dist(world, rank):
    d = types.SimpleNamespace()
    d.ReduceOp = real_dist.ReduceOp
    d.group = types.SimpleNamespace(WORLD=PhantomGroup("world", world, rank))
    d.is_available = lambda: True
    d.is_initialized = lambda: True
    d.get_world_size = lambda group=None: (group or d.group.WORLD).size
    d.get_rank = lambda group=None: (group or d.group.WORLD).rank
    d.get_backend = lambda group=None: "phantom"
    d.get_global_rank = lambda group, peer: peer
    d.barrier = lambda *a, **k: None

    def all_to_all(recv, send, group=None, async_op=False):
        me = (group or d.group.WORLD).rank
        if recv[me].numel():
            recv[me].copy_(send[me])
        return _Done()

    def all_reduce(t, op=None, group=None, async_op=False):
        return _Done()

    def all_gather(out, t, group=None, async_op=False):
        for o in out:
            o.copy_(t)
        return _Done()

    def all_gather_into_tensor(out, t, group=None, async_op=False):
        out.view(-1, t.numel())[:] = t.reshape(1, -1)
        return _Done()

    def reduce_scatter_tensor(out, t, op=None, group=None, async_op=False):
        n = out.numel()
        r = (group or d.group.WORLD).rank
        out.copy_(t.reshape(-1)[r * n:(r + 1) * n].view_as(out))
        return _Done()

    d.all_to_all, d.all_reduce, d.all_gather = all_to_all, all_reduce, all_gather
    d.all_gather_into_tensor, d.reduce_scatter_tensor = all_gather_into_tensor, reduce_scatter_tensor
    return d


def install_phantoms(h, w, ih, iw):
    import makani_amd.comm as mcomm
    import makani_amd.distributed as thd
    from makani_amd import dist_pipeline, ops, optim, losses, disco
    world, rank = h * w, ih * w + iw
    pd = phantom_dist(world, rank)
    for mod in (mcomm, thd, dist_pipeline, ops, optim, losses, disco):
        if hasattr(mod, "dist"):
            mod.dist = pd
    # functions that import torch.distributed locally
    import torch.distributed as td
    for name in ("get_world_size", "get_rank", "get_backend", "all_reduce", "all_gather", "all_to_all", "barrier",
                 "all_gather_into_tensor", "reduce_scatter_tensor", "is_initialized", "get_global_rank"):
        setattr(td, name, getattr(pd, name))
    mcomm.reset()
    sp = PhantomGroup("spatial", world, rank)
    groups = {"h": (PhantomGroup("h", h, ih) if h > 1 else None, h, ih), "w": (PhantomGroup("w", w, iw) if w > 1 else None, w, iw),
              "spatial": (sp, world, rank), "data": (None, 1, 0), "ensemble": (None, 1, 0), "batch": (None, 1, 0)}
    mcomm._GROUPS.update(groups)
    mcomm._GROUPS["model"] = mcomm._GROUPS["spatial"]
    for name in ("matmul", "fin", "fout"):
        mcomm._GROUPS[name] = (None, 1, 0)
    mcomm._SOURCE = "init"
    return pd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=4)
    ap.add_argument("--w", type=int, default=2)
    ap.add_argument("--ih", type=int, default=None, help="polar rank (default: the last = all 60 degrees x all orders live)")
    ap.add_argument("--iw", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--multistep-count", type=int, default=1)
    ap.add_argument("--config", default="sfno_sc3_layers8_edim384")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    h, w = a.h, a.w
    ih = a.ih if a.ih is not None else h - 1
    iw = a.iw
    import bench
    from makani_amd import ops
    import makani_amd.distributed as thd
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if h * w > 1:
        install_phantoms(h, w, ih, iw)
    cfg = bench.CONFIGS[a.config]
    H, W = cfg["inp_shape"]
    model = bench.build_model(a.config, dev, seed=333)
    assert model.spatial_parallel == (h * w > 1)
    opt = bench.make_optimizer(model)
    net = thd.init_gradient_reduction_hooks(model, dev) if h * w > 1 else model
    if a.multistep_count > 1:
        from makani_amd.stepper import MultiStepWrapper
        net = MultiStepWrapper(net, n_future=a.multistep_count - 1, multistep_checkpoint=False).train()
    lats, lons = thd.compute_split_shapes(H, h), thd.compute_split_shapes(W, w)
    hl, wl = lats[ih], lons[iw]
    torch.manual_seed(333)
    inp = torch.rand(1, cfg["inp_chans"], hl, wl, device=dev)
    tar = torch.rand(1, cfg["out_chans"] * a.multistep_count, hl, wl, device=dev)
    loss_fn = bench.make_loss(H, W, cfg["out_chans"] * a.multistep_count, dev, h * w > 1)
    amp, sharded_clip = not a.fp32, h > 1
    torch.backends.cuda.matmul.allow_tf32 = bool(amp)        # as bench.py: the reference's training flag (makani/train.py:87-88)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(a.warmup):
        bench.train_step(net, opt, inp, tar, loss_fn, amp, sharded_clip)
    torch.cuda.synchronize()
    # (1) wall time of eager steps, no events: the host's launch path
    thd.COMM_STATS.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.train_step(net, opt, inp, tar, loss_fn, amp, sharded_clip)
    t_host = time.perf_counter() - t0            # the launching thread is done (the GPU may still be draining)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    comm = {f"{k[0]}_group_of_{k[1]}": dict(MB_sent=round(v["bytes_sent"] / a.steps / 1e6, 2), all_to_alls=v["all_to_alls"] / a.steps)
            for k, v in sorted(thd.COMM_STATS.items())}
    # (1b) the same step captured as a hipGraph (the phantom collectives are plain copies): the GPU's time for the step with
    # no launch path at all = what a captured step with FREE links would take on this rank
    graph_ms = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bench.train_step(net, opt, inp, tar, loss_fn, amp, sharded_clip)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            bench.train_step(net, opt, inp, tar, loss_fn, amp, sharded_clip)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            graph.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / a.steps * 1e3
        del graph
    except Exception as e:
        print(f"[shadow] graph capture failed: {type(e).__name__}: {str(e)[:300]}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    # (2) per-kernel durations
    ops.PROFILER.reset()
    ops.PROFILER.enabled = True
    for _ in range(a.steps):
        bench.train_step(net, opt, inp, tar, loss_fn, amp, sharded_clip)
    torch.cuda.synchronize()
    ops.PROFILER.enabled = False
    prof = ops.PROFILER.summary()
    kernels = bench.kernel_table({}, prof, a.steps)
    hip_ms = sum(k["ms_per_step"] for k in kernels.values())
    fam = {}
    for k, d in kernels.items():
        f = fam.setdefault(bench.kernel_family(k), dict(ms_per_step=0.0, launches_per_step=0.0))
        f["ms_per_step"] += d["ms_per_step"]
        f["launches_per_step"] += d["launches_per_step"]
    # link time: each exchange over a group of k ranks sends (k - 1) pieces on (k - 1) links in parallel
    link_ms = sum(v["MB_sent"] / max(1, int(k.split("_")[-1]) - 1) / LINK_GBS for k, v in comm.items())
    from makani_amd import dist_pipeline as dp
    out = dict(parallelism=f"h{h}w{w}", rank=dict(ih=ih, iw=iw), local_grid=f"{hl}x{wl}", dtype="fp32" if a.fp32 else "bf16 autocast",
               multistep_count=a.multistep_count, steps=a.steps,
               wall_ms_per_step_eager_phantom=round(wall * 1e3, 2), host_launch_ms_per_step=round(t_host / a.steps * 1e3, 2),
               graph_ms_per_step_phantom=(round(graph_ms, 2) if graph_ms is not None else None),
               hip_kernel_ms_per_step=round(hip_ms, 3), launches_per_step=sum(k["launches_per_step"] for k in kernels.values()),
               exchange_per_step=comm, link_ms_per_step_at_153GBs=round(link_ms, 3),
               fused_fallbacks=dp.FALLBACKS, peak_hbm_GB=round(torch.cuda.max_memory_allocated() / 1e9, 2),
               families={k: dict(ms_per_step=round(v["ms_per_step"], 3), launches_per_step=v["launches_per_step"])
                         for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])},
               kernels=kernels)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    brief = {k: v for k, v in out.items() if k != "kernels"}
    print(json.dumps(brief))


if __name__ == "__main__":
    main()
