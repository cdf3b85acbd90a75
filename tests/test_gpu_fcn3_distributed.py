"""DISCO convolution, bilinear resampling and FourCastNet3 under h x w spatial model parallelism, validated on ONE GPU:
several processes share cuda:0 and exchange through gloo (host staged), each holds its latitude / longitude shard, and the
results are compared with the serial HIP operators (the pattern of the reference's
tests/distributed/tests_distributed_layers.py and tests_distributed_model.py; torch-harmonics' own distributed DISCO tests
compare with the serial operator the same way)."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _spawn(fn, nprocs, *args):
    """mp.spawn with a fresh rendezvous port; one retry when another process grabbed the port between probing and binding"""
    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=(nprocs, _free_port()) + args, nprocs=nprocs, join=True)
            return
        except Exception as e:
            if attempt == 1 or "EADDRINUSE" not in str(e):
                raise


def _setup(rank, world, port, h, w):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _fullsize import share_gpu
    share_gpu(rank, world)              # ranks sharing ONE GPU get disjoint compute units, set before the first GPU call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ih, iw = rank // w, rank % w
    hg = wg = None
    for j in range(w):
        g = dist.new_group([i * w + j for i in range(h)])
        if j == iw:
            hg = g
    for i in range(h):
        g = dist.new_group([i * w + j for j in range(w)])
        if i == ih:
            wg = g
    return ih, iw, hg, wg


def _shard(t, lat_shapes, lon_shapes, ih, iw):
    a, b = sum(lat_shapes[:ih]), sum(lon_shapes[:iw])
    return t[..., a:a + lat_shapes[ih], b:b + lon_shapes[iw]]


def _worker_ops(rank, world, port, h, w):
    ih, iw, hg, wg = _setup(rank, world, port, h, w)
    try:
        import makani_amd.distributed as thd
        from makani_amd import disco
        dev = torch.device("cuda:0")
        cases = [dict(cin=6, cout=8, in_shape=(33, 64), out_shape=(17, 32), groups=2, bias=True, grid_in="equiangular",
                      grid_out="legendre-gauss", dtype=torch.float32),
                 dict(cin=8, cout=6, in_shape=(17, 32), out_shape=(17, 32), groups=1, bias=False, grid_in="legendre-gauss",
                      grid_out="legendre-gauss", dtype=torch.float32),
                 dict(cin=8, cout=16, in_shape=(24, 48), out_shape=(24, 48), groups=1, bias=False, grid_in="equiangular",
                      grid_out="equiangular", dtype=torch.bfloat16)]
        serial, xs, gs, ys = [], [], [], []
        for c in cases:                         # serial operators first (thd not initialised yet)
            torch.manual_seed(5)
            m = disco.DiscreteContinuousConvS2(c["cin"], c["cout"], c["in_shape"], c["out_shape"], (3, 3), basis_type="morlet",
                                               groups=c["groups"], bias=c["bias"], grid_in=c["grid_in"], grid_out=c["grid_out"],
                                               theta_cutoff=4.0 * torch.pi / (c["in_shape"][0] - 1)).to(dev)
            x = torch.randn(2, c["cin"], *c["in_shape"], device=dev, dtype=c["dtype"]).requires_grad_(True)
            g = torch.randn(2, c["cout"], *c["out_shape"], device=dev, dtype=c["dtype"])
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=c["dtype"] == torch.bfloat16):
                y = m(x)
            (y * g).sum().backward()
            serial.append(m), xs.append(x), gs.append(g), ys.append(y)
        rs = disco.ResampleS2(17, 32, 33, 64, grid_in="legendre-gauss", grid_out="equiangular").to(dev)
        xr = torch.randn(2, 5, 17, 32, device=dev).requires_grad_(True)
        gr = torch.randn(2, 5, 33, 64, device=dev)
        yr = rs(xr)
        (yr * gr).sum().backward()

        thd.init(hg if h > 1 else None, wg if w > 1 else None, dist.group.WORLD)
        for c, m, x, g, y in zip(cases, serial, xs, gs, ys):
            d = disco.DistributedDiscreteContinuousConvS2(c["cin"], c["cout"], c["in_shape"], c["out_shape"], (3, 3),
                                                          basis_type="morlet", groups=c["groups"], bias=c["bias"],
                                                          grid_in=c["grid_in"], grid_out=c["grid_out"],
                                                          theta_cutoff=4.0 * torch.pi / (c["in_shape"][0] - 1)).to(dev)
            d.load_state_dict(m.state_dict())
            xl = _shard(x.detach(), d.lat_in_shapes, d.lon_in_shapes, ih, iw).clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=c["dtype"] == torch.bfloat16):
                yl = d(xl)
            gl = _shard(g, d.lat_out_shapes, d.lon_out_shapes, ih, iw)
            assert yl.shape == gl.shape, (yl.shape, gl.shape)
            (yl * gl).sum().backward()
            tol = 2e-2 if c["dtype"] == torch.bfloat16 else 2e-6
            e_y = _rel(yl, _shard(y, d.lat_out_shapes, d.lon_out_shapes, ih, iw))
            e_x = _rel(xl.grad, _shard(x.grad, d.lat_in_shapes, d.lon_in_shapes, ih, iw))
            assert e_y < tol and e_x < tol, (rank, c, e_y, e_x)
            for k, p in d.named_parameters():
                gp = p.grad.detach().float().cpu()
                dist.all_reduce(gp)                                   # replicated weight: shard gradients sum
                e = _rel(gp, dict(m.named_parameters())[k].grad.float().cpu())
                assert e < (3e-2 if c["dtype"] == torch.bfloat16 else 1e-5), (rank, k, e)
        dr = disco.DistributedResampleS2(17, 32, 33, 64, grid_in="legendre-gauss", grid_out="equiangular").to(dev)
        xl = _shard(xr.detach(), dr.lat_in_shapes, dr.lon_in_shapes, ih, iw).clone().requires_grad_(True)
        yl = dr(xl)
        (yl * _shard(gr, dr.lat_out_shapes, dr.lon_out_shapes, ih, iw)).sum().backward()
        assert _rel(yl, _shard(yr, dr.lat_out_shapes, dr.lon_out_shapes, ih, iw)) < 1e-6
        assert _rel(xl.grad, _shard(xr.grad, dr.lat_in_shapes, dr.lon_in_shapes, ih, iw)) < 1e-6
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2), (3, 1)])
def test_distributed_disco_and_resample_match_serial(h, w):
    _spawn(_worker_ops, h * w, h, w)


def _worker_fcn3(rank, world, port, h, w, name):
    ih, iw, hg, wg = _setup(rank, world, port, h, w)
    try:
        import numpy as np
        import makani_amd as ma
        import makani_amd.distributed as thd
        dev = torch.device("cuda:0")
        g = np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=True)
        kwargs = json.loads(str(g["kwargs"]))
        sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
        serial = ma.AtmoSphericNeuralOperatorNet(**kwargs)
        serial.load_state_dict(sd, strict=True)
        serial = serial.to(dev)
        x = torch.from_numpy(g["x"]).to(dev)
        G = torch.from_numpy(g["g"]).to(dev)
        xs = x.clone().requires_grad_(True)
        ys = serial(xs)
        (ys * G).sum().backward()

        thd.init(hg if h > 1 else None, wg if w > 1 else None, dist.group.WORLD)
        model = ma.AtmoSphericNeuralOperatorNet(**kwargs).to(dev)
        assert isinstance(model.sht, thd.DistributedRealSHT)
        lat = thd.compute_split_shapes(kwargs["inp_shape"][0], h)
        lon = thd.compute_split_shapes(kwargs["inp_shape"][1], w)
        l0, ll = sum(model.isht.l_shapes[:ih]), model.isht.l_shapes[ih]
        own = model.state_dict()
        for k in own:
            src = sd[k].to(dev)
            if k.endswith("global_conv.weight"):
                src = src[..., l0:l0 + ll]
            assert own[k].shape == src.shape, (k, own[k].shape, src.shape)
            own[k].copy_(src)
        xl = _shard(x, lat, lon, ih, iw).clone().requires_grad_(True)
        yl = model(xl)
        (yl * _shard(G, lat, lon, ih, iw)).sum().backward()
        e_y = _rel(yl, _shard(ys, lat, lon, ih, iw))
        e_gx = _rel(xl.grad, _shard(xs.grad, lat, lon, ih, iw))
        assert e_y < 1e-4 and e_gx < 2e-4, (rank, e_y, e_gx)
        sref = dict(serial.named_parameters())
        gmax = max(float(p.grad.abs().max()) for p in sref.values())
        for k, p in model.named_parameters():
            gp = (torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad).detach().cpu().contiguous()
            if k.endswith("global_conv.weight"):                     # sharded over h, shared over w
                if w > 1:
                    dist.all_reduce(gp, group=wg)
                ref = sref[k].grad[..., l0:l0 + ll].contiguous()
                ref = (torch.view_as_real(ref) if ref.is_complex() else ref).cpu()
            else:
                dist.all_reduce(gp)
                ref = sref[k].grad.cpu()
            e, a = _rel(gp, ref), (gp - ref).abs().max().item()
            assert e < 5e-4 or a < 1e-4 * gmax, (rank, k, e, a)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,name", [(2, 1, "fcn3_small_33x64.npz"), (1, 2, "fcn3_small_33x64.npz"),
                                      (2, 2, "fcn3_small_33x64.npz"), (2, 2, "fcn3_options_24x48.npz")])
def test_spatial_parallel_fcn3_matches_serial(h, w, name):
    _spawn(_worker_fcn3, h * w, h, w, name)
