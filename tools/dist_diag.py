"""Where does the h x w distributed network's BACKWARD pass leave the serial one?  N ranks share the GPU (gloo), every rank
runs the serial HIP model AND its shard of the distributed one on the same weights / input / cotangent and prints the
relative error of the output, the input gradient and every parameter gradient in network order, for a list of variants
(number of layers, normalisation off, exchange schedule) — the diagnostic behind the full-size tests of
tests/test_gpu_dist_fullsize.py.

    python tools/dist_diag.py --h 4 --w 1 [--variants base,nonorm,fused0,l2,l2nonorm]
"""
import argparse
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

VARIANTS = {
    "base": dict(),
    "nonorm": dict(normalization_layer="none"),
    "l2": dict(num_layers=2),
    "l2nonorm": dict(num_layers=2, normalization_layer="none"),
    "l1": dict(num_layers=1),
    "nomlp": dict(use_mlp=False),
    "noskip": dict(big_skip=False),
    "fused0": dict(_env=dict(MAKANI_AMD_DIST_FUSED="0")),
    "small": dict(inp_shape=(361, 720), out_shape=(361, 720), embed_dim=128),
}


def _r(t):
    t = t.detach()
    return torch.view_as_real(t.resolve_conj()) if t.is_complex() else t


def _rel(a, b):
    a, b = _r(a).double(), _r(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def worker(rank, world, port, h, w, names, amp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(8)
    if os.environ.get("DIST_DIAG_SHARE_CUS", "0") != "1":
        from _fullsize import share_gpu
        share_gpu(rank, world)                # disjoint compute units per rank (before the first GPU call)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import makani_amd as ma
    import makani_amd.comm as mcomm
    import makani_amd.distributed as thd
    from _fullsize import CONFIG2, perturb_affine
    dev = torch.device("cuda:0")
    for name in names:
        var = dict(VARIANTS[name])
        env = var.pop("_env", {})
        cfg = {**CONFIG2, **var}
        for k, v in env.items():
            os.environ[k] = v
        # serial
        mcomm.reset()
        thd._INIT = False
        thd._POLAR = thd._AZIMUTH = thd._SPATIAL = None
        torch.manual_seed(333)
        serial = ma.SphericalFourierNeuralOperatorNet(**cfg)
        assert not serial.spatial_parallel
        perturb_affine(serial, 7)
        H, W = cfg["inp_shape"]
        x = torch.rand(1, 73, H, W)
        g = torch.randn(1, 73, H, W, generator=torch.Generator().manual_seed(99))
        sd = {k: v.clone() for k, v in serial.state_dict().items()}
        serial = serial.to(dev)
        xs = x.to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            ys = serial(xs)
        (ys.float() * g.to(dev)).sum().backward()
        sref = {n: p.grad.detach().clone() for n, p in serial.named_parameters()}
        ys, gxs = ys.detach().float(), xs.grad.detach().clone()
        del serial
        torch.cuda.empty_cache()
        # distributed
        _, ih, iw = mcomm.init(h, w)
        model = ma.SphericalFourierNeuralOperatorNet(**cfg)
        assert model.spatial_parallel
        td = model.trans_down
        l0, ll = sum(td.l_shapes[:ih]), td.l_shapes[ih]
        own = model.state_dict()
        with torch.no_grad():
            for k in own:
                src = sd[k][..., l0:l0 + ll] if k.endswith("filter.filter.weight") else sd[k]
                own[k].copy_(src)
        model = model.to(dev)
        net = thd.init_gradient_reduction_hooks(model, dev)
        lat0, lon0 = sum(td.lat_shapes[:ih]), sum(td.lon_shapes[:iw])
        hl, wl = td.lat_shapes[ih], td.lon_shapes[iw]
        sl = (..., slice(lat0, lat0 + hl), slice(lon0, lon0 + wl))
        xl = x[sl].to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yl = net(xl)
        (yl.float() * g[sl].to(dev)).sum().backward()
        torch.cuda.synchronize()
        lines = [f"[{name}] rank {rank}: y {_rel(yl.float(), ys[sl]):.2e}  gx {_rel(xl.grad, gxs[sl]):.2e}"]
        for n, p in model.named_parameters():
            ref = sref[n][..., l0:l0 + ll] if n.endswith("filter.filter.weight") else sref[n]
            lines.append(f"    {n:40s} {_rel(p.grad, ref):.2e}   |ref| {float(_r(ref).norm()):.2e}")
        dist.barrier()
        if rank == world - 1:
            print("\n".join(lines), flush=True)
        else:
            print(lines[0], flush=True)
        for k in env:
            os.environ.pop(k, None)
        del model, net
        torch.cuda.empty_cache()
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=4)
    ap.add_argument("--w", type=int, default=1)
    ap.add_argument("--variants", default="base,nonorm,l2,l2nonorm")
    ap.add_argument("--amp", action="store_true")
    a = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(a.h * a.w, port, a.h, a.w, a.variants.split(","), a.amp), nprocs=a.h * a.w, join=True)
