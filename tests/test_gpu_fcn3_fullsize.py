"""BASELINE configs[3] in ITS OWN layout at FULL SIZE: FourCastNet3 ``fcn3_sc2_edim45_layers10``
(config/fourcastnet3.yaml:24-46: 721 x 1440 input grid, 360 x 720 internal grid, 641 + 36 = 677 channels, 10 layers, Morlet
DISCO filters) under h2 w2 spatial model parallelism — FOUR ranks share ONE GPU (all compute units, no masks) and exchange
through gloo —, forward + backward in fp32 with ensemble size 1: every rank's output shard, input-gradient shard and REDUCED
parameter gradients against the serial HIP network (which tests/test_fcn3.py pins to fixtures generated from the reference's
own module).  What this runs that the 33 x 64 tests cannot: the distributed DISCO halo exchange with ragged 361 / 360 latitude
shards and 720-column longitude shards, the distributed bilinear resampling 721 x 1440 <-> 360 x 720, the l-sharded global
(spectral) convolutions at L = 360, the 677-channel grouped mixes on shard-sized pixel counts.
Reference twins: tests/distributed/tests_distributed_model.py:218-330, makani/models/networks/fourcastnet3.py:339-381.
Tolerance: fp32 <= 1e-4 end to end (tests/distributed/tests_distributed_layers.py:71-76)."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist

from _fullsize import CACHE_DIR, log_line, spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
H, W = 2, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _r(t):
    t = t.detach()
    if t.is_complex():
        t = torch.view_as_real(t.resolve_conj())
    return t


def _rel(a, b):
    a, b = _r(a).cpu().double(), _r(b).cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _config():
    sys.path.insert(0, ROOT)
    import bench
    return bench.FCN3_CONFIGS["fcn3_sc2_edim45_layers10"]["model"], bench.CONFIGS["fcn3_sc2_edim45_layers10"]


def _serial_worker(rank, path):
    """the SERIAL HIP network in a process of its own (its ~100 GB of fp32 activations are gone when it ends)"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    import makani_amd as ma
    kw, cfg = _config()
    torch.manual_seed(333)
    model = ma.AtmoSphericNeuralOperatorNet(**kw)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to("cuda:0")
    x = torch.rand(1, cfg["inp_chans"], *kw["inp_shape"])                     # DummyLoader-shaped U[0, 1)
    g = torch.randn(1, cfg["out_chans"], *kw["out_shape"], generator=torch.Generator().manual_seed(99))
    xd = x.to("cuda:0").requires_grad_(True)
    t0 = time.time()
    y = model(xd)
    (y * g.to("cuda:0")).sum().backward()
    torch.cuda.synchronize()
    out = dict(state=state, x=x, g=g, y=y.detach().cpu(), gx=xd.grad.cpu(),
               grads={n: _r(p.grad).cpu().contiguous() for n, p in model.named_parameters()},
               peak_gib=torch.cuda.max_memory_allocated() / 2 ** 30, seconds=time.time() - t0)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(out, path)


@pytest.fixture(scope="module")
def serial_fcn3():
    path = os.path.join(CACHE_DIR, f"fcn3_hip_serial_{os.getpid()}.pt")
    spawn(_serial_worker, (path,), 1, timeout_s=900)
    d = torch.load(path, mmap=True, weights_only=True)
    log_line(f"--- FourCastNet3 fcn3_sc2_edim45_layers10 serial HIP fp32 forward + backward at 721x1440: peak {float(d['peak_gib']):.1f} GiB, "
             f"{float(d['seconds']):.1f} s ---")
    yield path
    try:
        os.remove(path)
    except OSError:
        pass


def _shard(t, lat, lon, ih, iw):
    a, b = sum(lat[:ih]), sum(lon[:iw])
    return t[..., a:a + lat[ih], b:b + lon[iw]]


def _worker(rank, world, port, path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    torch.set_num_threads(8)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd as ma
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        _, ih, iw = mcomm.init(H, W)
        kw, cfg = _config()
        ref = torch.load(path, mmap=True, weights_only=True)
        t0 = time.time()
        model = ma.AtmoSphericNeuralOperatorNet(**kw)
        assert isinstance(model.sht, thd.DistributedRealSHT)
        l0, ll = sum(model.isht.l_shapes[:ih]), model.isht.l_shapes[ih]
        own = model.state_dict()
        with torch.no_grad():
            for k in own:
                src = ref["state"][k]
                if k.endswith("global_conv.weight"):
                    src = src[..., l0:l0 + ll]
                assert own[k].shape == src.shape, (k, own[k].shape, src.shape)
                own[k].copy_(src)
        model = model.to("cuda:0")
        lat = thd.compute_split_shapes(kw["inp_shape"][0], H)
        lon = thd.compute_split_shapes(kw["inp_shape"][1], W)
        xl = _shard(ref["x"], lat, lon, ih, iw).to("cuda:0").clone().requires_grad_(True)
        gl = _shard(ref["g"], lat, lon, ih, iw).to("cuda:0")
        torch.cuda.synchronize()
        t1 = time.time()
        yl = model(xl)
        (yl * gl).sum().backward()
        torch.cuda.synchronize()
        t2 = time.time()
        e_y = _rel(yl, _shard(ref["y"], lat, lon, ih, iw))
        e_gx = _rel(xl.grad, _shard(ref["gx"], lat, lon, ih, iw))
        gmax = max(float(v.abs().max()) for v in ref["grads"].values())
        hgrp, wgrp = mcomm.get_group("h") if H > 1 else None, mcomm.get_group("w") if W > 1 else None
        worst, bad = ("", 0.0), {}
        for k, p in model.named_parameters():
            gp = _r(p.grad).cpu().contiguous()
            if k.endswith("global_conv.weight"):                     # sharded over h (degrees), shared over w
                if W > 1:
                    dist.all_reduce(gp, group=wgrp)
                r = ref["grads"][k]
                r = r[..., l0:l0 + ll, :] if p.is_complex() else r[..., l0:l0 + ll]          # (complex gradients are kept as real views)
            else:                                                    # replicated: the shards' gradients sum over h x w
                dist.all_reduce(gp)
                r = ref["grads"][k]
            e, a = _rel(gp, r), float((gp.double() - r.double()).abs().max())
            if not (e < TOL or a < 1e-5 * gmax):
                bad[k] = (e, a)
            if e > worst[1] and a >= 1e-5 * gmax:
                worst = (k, e)
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        log_line(f"fcn3 h{H}w{W} rank {rank} (ih {ih}, iw {iw}; {lat[ih]}x{lon[iw]} px, l {l0}..{l0 + ll}) fp32 vs serial HIP:  y {e_y:.2e}  gx {e_gx:.2e}  "
                 f"worst parameter gradient {worst[0]} {worst[1]:.2e} (of {len(ref['grads'])})  peak {peak:.1f} GiB, setup {t1 - t0:.0f} s, "
                 f"fwd+bwd over gloo {t2 - t1:.1f} s")
        assert e_y < TOL and e_gx < TOL, (rank, e_y, e_gx)
        assert not bad, (rank, bad)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_fcn3_fullsize_h2w2_fwd_bwd_matches_serial(serial_fcn3):
    log_line("--- test_fcn3_fullsize_h2w2_fwd_bwd_matches_serial: fcn3_sc2_edim45_layers10, 721x1440 / 360x720 / 677 ch, 4 ranks on one GPU ---")
    spawn(_worker, (H * W, _free_port(), serial_fcn3), H * W, timeout_s=1500)
