"""Spatial (h x w) model parallelism on the HIP path, validated on ONE GPU: several processes share cuda:0
and exchange through gloo (host-staged), so the full distributed SFNO — distributed SHT with the HIP FFT /
Legendre kernels, l-sharded dhconv weights with triangular shard offsets, distributed instance norm —
runs end to end and is compared with the serial HIP model (the pattern of the reference's
tests/distributed/tests_distributed_layers.py:69-223 and tests_distributed_model.py:218-330)."""
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


# distributed against serial HIP in fp32, 16-channel toy network.  Rounds 2-4 needed 1e-4 / 2e-4 / 5e-4 here because the ranks
# shared compute units (docs/LAB_NOTEBOOK.md 5.1); with a compute-unit range per rank the two schedules agree to summation order:
# measured y 4e-7 ... 7e-7, gx 8e-7 ... 1.4e-6, worst parameter gradient 2.4e-6 ... 4.2e-6 (gpurun_out/r05s)
TOL_Y, TOL_G = 5e-6, 2e-5


def _worker(rank, world, port, h, w, norm="instance_norm"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _fullsize import share_gpu
    share_gpu(rank, world)              # ranks sharing ONE GPU get disjoint compute units, set before the first GPU call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd as ma
        import makani_amd.distributed as thd
        dev = torch.device("cuda:0")
        cfg = dict(inp_shape=(37, 72), out_shape=(37, 72), inp_chans=4, out_chans=4, scale_factor=3, embed_dim=16,
                   num_layers=3, mlp_ratio=2, normalization_layer=norm)
        B = 2
        torch.manual_seed(11)
        serial = ma.SphericalFourierNeuralOperatorNet(**cfg).to(dev)
        x = torch.rand(B, 4, 37, 72, device=dev)
        G = torch.randn(B, 4, 37, 72, device=dev)
        xs = x.clone().requires_grad_(True)
        ys = serial(xs)
        (ys * G).sum().backward()

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
        thd.init(hg if h > 1 else None, wg if w > 1 else None, dist.group.WORLD)
        model = ma.SphericalFourierNeuralOperatorNet(**cfg).to(dev)
        assert model.spatial_parallel
        td = model.trans_down
        lat0, lon0 = sum(td.lat_shapes[:ih]), sum(td.lon_shapes[:iw])
        hl, wl = td.lat_shapes[ih], td.lon_shapes[iw]
        l0, ll = sum(td.l_shapes[:ih]), td.l_shapes[ih]
        sd = serial.state_dict()
        own = model.state_dict()
        for k in own:
            src = sd[k]
            if k.endswith("filter.filter.weight"):
                src = src[..., l0:l0 + ll]
            assert own[k].shape == src.shape, (k, own[k].shape, src.shape)
            own[k].copy_(src)
        xl = x[..., lat0:lat0 + hl, lon0:lon0 + wl].clone().requires_grad_(True)
        yl = model(xl)
        (yl * G[..., lat0:lat0 + hl, lon0:lon0 + wl]).sum().backward()
        assert yl.shape == (B, 4, hl, wl)
        e_y = _rel(yl, ys[..., lat0:lat0 + hl, lon0:lon0 + wl])
        e_gx = _rel(xl.grad, xs.grad[..., lat0:lat0 + hl, lon0:lon0 + wl])
        assert e_y < TOL_Y and e_gx < TOL_G, (rank, e_y, e_gx)
        sref = dict(serial.named_parameters())
        worst = ("", 0.0)
        for k, p in model.named_parameters():
            g = (torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad).detach().cpu().contiguous()
            if k.endswith("filter.filter.weight"):           # sharded over h, shared over w
                if wg is not None and w > 1:
                    dist.all_reduce(g, group=wg)
                ref = torch.view_as_real(sref[k].grad[..., l0:l0 + ll].contiguous()).cpu()
            else:                                            # replicated: partial gradients sum over spatial
                dist.all_reduce(g)
                ref = sref[k].grad.cpu()
            if k.endswith("mlp.fwd.3.bias"):                 # (exactly zero by construction: a constant in front of an instance norm)
                continue
            e = _rel(g, ref)
            worst = max(worst, (k, e), key=lambda t: t[1])
            assert e < TOL_G, (rank, k, e)
        print(f"h{h}w{w} {norm} rank {rank}: y {e_y:.1e} gx {e_gx:.1e} worst parameter gradient {worst[0]} {worst[1]:.1e}", flush=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2)])
def test_spatial_parallel_sfno_matches_serial(h, w):
    world = h * w
    mp.spawn(_worker, args=(world, _free_port(), h, w), nprocs=world, join=True)


@pytest.mark.parametrize("h,w", [(2, 1), (2, 2)])
def test_spatial_parallel_sfno_with_geometric_instance_norm_matches_serial(h, w):
    """normalization_layer="instance_norm_s2" under h x w parallelism: DistributedGeometricInstanceNormS2
    (makani/mpu/layer_norm.py:173-253: area-weighted local moments merged over the spatial group) against the serial
    GeometricInstanceNormS2 network, forward, input gradient and every parameter gradient"""
    world = h * w
    mp.spawn(_worker, args=(world, _free_port(), h, w, "instance_norm_s2"), nprocs=world, join=True)


class _OneRankTree:
    """a process-group tree that names the (one-rank) world as a split "spatial" group: every gradient takes the SUM
    stage of the reducer, which over one rank is the identity"""

    @staticmethod
    def get_comm_names():
        return ["spatial", "data"]

    @staticmethod
    def get_size(name):
        return 2 if name == "spatial" else 1

    @staticmethod
    def get_group(name):
        return dist.group.WORLD if name == "spatial" else None

    @staticmethod
    def is_initialized():
        return True


def _worker_rccl(rank, world, port):
    """RCCL itself on the one GPU of the box (world size 1): the collectives makani_amd.distributed.GradReducer issues —
    an async in-place all-reduce on the real view of a complex gradient, the flattened small-gradient bucket whose
    pieces become the gradients, and ReduceOp.AVG (or its fallback)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import makani_amd.distributed as thd
        model = torch.nn.Module()
        model.c = torch.nn.Parameter(torch.randn(1200, 1024, dtype=torch.complex64, device=dev))     # 9.8 MB: async path
        model.a = torch.nn.Parameter(torch.randn(7, 5, device=dev))
        model.sc = torch.nn.Parameter(torch.randn(3, 3, dtype=torch.complex64, device=dev))            # small complex: bucket
        for p in model.parameters():
            p.is_shared_mp = ["spatial"]
        red = thd.GradReducer(model, comm=_OneRankTree)
        assert red.active and all(len(st) == 1 for st in red.plan.values())
        (torch.view_as_real(model.c).sum() * 3.0 + model.a.sum() * 2.0 + torch.view_as_real(model.sc).sum() * 5.0).backward()
        assert not red.pending and not red.small                    # finished by the autograd engine callback
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(model.c.grad), torch.full((1200, 1024, 2), 3.0, device=dev))
        assert torch.equal(model.a.grad, torch.full((7, 5), 2.0, device=dev))
        assert torch.equal(torch.view_as_real(model.sc.grad), torch.full((3, 3, 2), 5.0, device=dev))
        assert thd.GradReducer._probe_avg(dist.group.WORLD, dev) in (True, False)
        t = torch.full((4,), 5.0, device=dev)
        try:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
            assert torch.equal(t, torch.full((4,), 5.0, device=dev))
        except RuntimeError:
            pass                                      # GradReducer then keeps SUM + scale
        # the all-to-all primitive of the distributed transforms on the RCCL branch (one rank: a copy)
        send = [torch.arange(12.0, device=dev).view(3, 4)]
        recv = [torch.empty(3, 4, device=dev)]
        thd._exchange(recv, send, dist.group.WORLD)
        torch.cuda.synchronize()
        assert torch.equal(recv[0], send[0])
        # ... and as the fused schedule issues it: asynchronous, contiguous views of differing shapes
        from makani_amd import dist_pipeline as dp
        big = torch.arange(24.0, device=dev)
        out = torch.zeros(2, 3, 4, device=dev)
        dp._exchange_async([out[0:2]], [big], dist.group.WORLD).wait()
        torch.cuda.synchronize()
        assert torch.equal(out.reshape(-1), big)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_the_grad_reducer_world1():
    mp.spawn(_worker_rccl, args=(1, _free_port()), nprocs=1, join=True)


def _worker_ragged_gpu(rank, world, port, h, w, C, fused=True):
    """BASELINE configs[2] / [4] split sizes at the real grid THROUGH THE HIP BACKEND: 721 x 1440, lmax 240, mmax 241 over
    h = 4 (lat [181, 181, 181, 178], l [60] * 4) and h4 w2 (lon [720, 720], m [121, 120]) with ragged plane counts; every
    rank's shard of the distributed transform and of its gradient against the serial HIP transform AND the fp64 oracle
    (split rule: makani/mpu/fft.py:50-51; tolerance of the reference's distributed tests: tests_distributed_layers.py:71-76)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(4)
    os.environ["MAKANI_AMD_DIST_FUSED"] = "1" if fused else "0"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _fullsize import share_gpu
    share_gpu(rank, world)              # ranks sharing ONE GPU get disjoint compute units, set before the first GPU call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd as ma
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        from oracle import sht as osht
        dev = torch.device("cuda:0")
        _, ih, iw = mcomm.init(h, w)
        assert thd.ensure_initialized() and thd._BACKEND is thd.HipBackend
        nlat, nlon, lmax, mmax, B = 721, 1440, 240, 241, 1
        kw = dict(lmax=lmax, mmax=mmax, grid="equiangular")
        fwd = thd.DistributedRealSHT(nlat, nlon, **kw).to(dev)
        inv = thd.DistributedInverseRealSHT(nlat, nlon, **kw).to(dev)
        if h == 4:
            assert fwd.lat_shapes == [181, 181, 181, 178] and fwd.l_shapes == [60, 60, 60, 60]
        if w == 2:
            assert fwd.lon_shapes == [720, 720] and fwd.m_shapes == [121, 120]
        from makani_amd import dist_pipeline as dp
        assert dp.eligible(fwd, torch.float32) == fused and dp.eligible(inv, torch.float32) == fused      # which schedule runs
        lat0, lon0 = sum(fwd.lat_shapes[:ih]), sum(fwd.lon_shapes[:iw])
        l0, m0 = sum(fwd.l_shapes[:ih]), sum(fwd.m_shapes[:iw])
        hl, wl, ll, ml = fwd.lat_shapes[ih], fwd.lon_shapes[iw], fwd.l_shapes[ih], fwd.m_shapes[iw]
        torch.manual_seed(7)
        x = torch.randn(B, C, nlat, nlon)
        G = torch.randn(B, C, lmax, mmax, dtype=torch.complex64)
        coef = torch.tril(torch.randn(B, C, lmax, mmax, dtype=torch.complex64))
        gy = torch.randn(B, C, nlat, nlon)
        # serial HIP transform and fp64 oracle on the whole field
        S, I = ma.RealSHT(nlat, nlon, **kw).to(dev), ma.InverseRealSHT(nlat, nlon, **kw).to(dev)
        xs = x.to(dev).requires_grad_(True)
        cs = S(xs)
        (torch.view_as_real(cs) * torch.view_as_real(G.to(dev))).sum().backward()
        cfs = coef.to(dev).requires_grad_(True)
        ys = I(cfs)
        (ys * gy.to(dev)).sum().backward()
        xo = x.double().requires_grad_(True)
        co = osht.RealSHT(nlat, nlon, **kw)(xo)
        (torch.view_as_real(co) * torch.view_as_real(G.to(torch.complex128))).sum().backward()
        cfo = coef.to(torch.complex128).requires_grad_(True)
        yo = osht.InverseRealSHT(nlat, nlon, **kw)(cfo)
        (yo * gy.double()).sum().backward()
        # this rank's shard through the distributed HIP transform
        xl = x[..., lat0:lat0 + hl, lon0:lon0 + wl].to(dev).requires_grad_(True)
        c = fwd(xl)
        assert c.shape == (B, C, ll, ml) and c.dtype == torch.complex64
        (torch.view_as_real(c) * torch.view_as_real(G[..., l0:l0 + ll, m0:m0 + ml].to(dev))).sum().backward()
        cl = coef[..., l0:l0 + ll, m0:m0 + ml].to(dev).requires_grad_(True)
        y = inv(cl)
        assert y.shape == (B, C, hl, wl)
        (y * gy[..., lat0:lat0 + hl, lon0:lon0 + wl].to(dev)).sum().backward()
        sl_s, sl_g = (..., slice(l0, l0 + ll), slice(m0, m0 + ml)), (..., slice(lat0, lat0 + hl), slice(lon0, lon0 + wl))
        tri = torch.tril(torch.ones(lmax, mmax))[l0:l0 + ll, m0:m0 + ml]           # gradients of the l < m entries are unspecified
        pairs = {"fwd": (c, cs[sl_s], co[sl_s]), "fwd_gx": (xl.grad, xs.grad[sl_g], xo.grad[sl_g]),
                 "inv": (y, ys[sl_g], yo[sl_g]),
                 "inv_gc": (cl.grad.cpu() * tri, cfs.grad[sl_s].cpu() * tri, cfo.grad[sl_s] * tri)}
        for k, (a, b_hip, b_or) in pairs.items():
            e1, e2 = _rel(torch.view_as_real(a.detach().cpu()) if a.is_complex() else a.detach().cpu(),
                          torch.view_as_real(b_hip.detach().cpu()) if b_hip.is_complex() else b_hip.detach().cpu()), \
                     _rel(torch.view_as_real(a.detach().cpu()) if a.is_complex() else a.detach().cpu(),
                          torch.view_as_real(b_or.detach()) if b_or.is_complex() else b_or.detach())
            assert e1 < 1e-5 and e2 < 1e-5, (rank, k, e1, e2)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,C,fused", [(4, 1, 6, True), (4, 2, 5, True), (4, 2, 5, False), (2, 2, 48, True)])
def test_distributed_sht_ragged_config3_splits_on_the_hip_backend(h, w, C, fused):
    """fused = the schedule of makani_amd/dist_pipeline.py (segmented FFT kernels, one h x w exchange, latitude-major Legendre
    operand, two latitude chunks); not fused = transpose by transpose"""
    mp.spawn(_worker_ragged_gpu, args=(h * w, _free_port(), h, w, C, fused), nprocs=h * w, join=True)


# --------------------------------------------------------------------------- #
# BASELINE configs[4] as a composition: MultiStepWrapper (n_future = 3 = multistep_count 4) around the spatially parallel
# network with the package's own gradient reduction — the reference's makani/models/stepper.py:224-284 over makani/mpu layers
# --------------------------------------------------------------------------- #
def _worker_multistep(rank, world, port, h, w, checkpointed, amp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _fullsize import share_gpu
    share_gpu(rank, world)              # ranks sharing ONE GPU get disjoint compute units, set before the first GPU call
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd as ma
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        from makani_amd.stepper import MultiStepWrapper
        dev = torch.device("cuda:0")
        cfg = dict(inp_shape=(37, 72), out_shape=(37, 72), inp_chans=4, out_chans=4, scale_factor=3, embed_dim=16,
                   num_layers=2, mlp_ratio=2)
        B, NF = 1, 3
        torch.manual_seed(23)
        serial = ma.SphericalFourierNeuralOperatorNet(**cfg).to(dev)
        x = torch.rand(B, 4, 37, 72, device=dev)
        G = torch.randn(B, 4 * (NF + 1), 37, 72, device=dev)
        xs = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            ys = MultiStepWrapper(serial, n_future=NF).train()(xs)
        (ys.float() * G).sum().backward()
        e_ser = None
        if amp:
            # bf16: a 4-step rollout of a 16-channel random network amplifies rounding differences (the two schedules sum in
            # different orders), so the reference is the serial FP32 rollout and the yardstick the serial bf16 rollout's own
            # distance from it
            def _r(t):
                return torch.view_as_real(t) if t.is_complex() else t
            g16 = {k: p.grad.clone() for k, p in serial.named_parameters()}
            y16, gx16 = ys.float().detach(), xs.grad.clone()
            serial.zero_grad(set_to_none=True)
            xs = x.clone().requires_grad_(True)
            ys = MultiStepWrapper(serial, n_future=NF).train()(xs)
            (ys * G).sum().backward()
            e_ser = (_rel(y16, ys), _rel(gx16, xs.grad),
                     max(_rel(_r(g16[k]), _r(p.grad)) for k, p in serial.named_parameters() if not k.endswith("mlp.fwd.3.bias")))

        _, ih, iw = mcomm.init(h, w)                       # the tree bench.py builds; the network finds h / w / spatial in it
        model = ma.SphericalFourierNeuralOperatorNet(**cfg).to(dev)
        assert model.spatial_parallel
        td = model.trans_down
        lat0, lon0 = sum(td.lat_shapes[:ih]), sum(td.lon_shapes[:iw])
        hl, wl = td.lat_shapes[ih], td.lon_shapes[iw]
        l0, ll = sum(td.l_shapes[:ih]), td.l_shapes[ih]
        sd, own = serial.state_dict(), model.state_dict()
        for k in own:
            src = sd[k][..., l0:l0 + ll] if k.endswith("filter.filter.weight") else sd[k]
            own[k].copy_(src)
        net = thd.init_gradient_reduction_hooks(model, dev)            # mappings.py:321-525: the sums over h x w / w complete in backward()
        net = MultiStepWrapper(net, n_future=NF, multistep_checkpoint=checkpointed).train()
        xl = x[..., lat0:lat0 + hl, lon0:lon0 + wl].clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yl = net(xl)
        assert yl.shape == (B, 4 * (NF + 1), hl, wl)
        (yl.float() * G[..., lat0:lat0 + hl, lon0:lon0 + wl]).sum().backward()
        # fp32: against the serial rollout; bf16: against the serial FP32 rollout, no further from it than 2 x the serial bf16
        # rollout is (a shard's relative error scatters around the whole field's)
        tol_y, tol_g = (2 * e_ser[0], 2 * e_ser[1]) if amp else (2e-4, 1e-3)
        e_y = _rel(yl.float(), ys.float()[..., lat0:lat0 + hl, lon0:lon0 + wl])
        e_gx = _rel(xl.grad, xs.grad[..., lat0:lat0 + hl, lon0:lon0 + wl])
        assert e_y < tol_y and e_gx < tol_g, (rank, e_y, e_gx, e_ser)
        sref = dict(serial.named_parameters())
        for k, p in model.named_parameters():
            # bf16: the parameter gradients of this 16-channel random network through a 4-step rollout are dominated by rounding
            # (the serial bf16 rollout's own gradients sit e_ser[2] ~ 1 from the fp32 ones; measured 2.7 between the schedules on
            # the encoder weight), so they are compared in fp32 only — outputs and input gradient above are compared in both
            if amp or k.endswith("mlp.fwd.3.bias"):
                continue
            ref = sref[k].grad[..., l0:l0 + ll] if k.endswith("filter.filter.weight") else sref[k].grad
            e = _rel(torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad,
                     torch.view_as_real(ref.contiguous()) if ref.is_complex() else ref)
            assert e < tol_g, (rank, k, e)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,checkpointed,amp", [(2, 2, False, False), (2, 2, True, False), (2, 2, False, True), (4, 2, False, False)])
def test_multistep4_around_the_spatially_parallel_network_matches_serial(h, w, checkpointed, amp):
    """BASELINE configs[4] (multistep_count 4 under h x w spatial parallelism): a 4-step rollout (MultiStepWrapper,
    n_future = 3) of the h x w distributed network — every step's output re-enters the distributed transforms as a shard — with
    the package's gradient reduction hooks, N ranks on one GPU, against the same rollout of the serial HIP network: all four
    outputs, the input gradient through the rollout, every parameter gradient AFTER the reductions (so p.grad is what the
    optimizer would see), with and without rollout checkpointing, fp32 and bf16 autocast"""
    world = h * w
    mp.spawn(_worker_multistep, args=(world, _free_port(), h, w, checkpointed, amp), nprocs=world, join=True)


# --------------------------------------------------------------------------- #
# the fused schedule's collectives on RCCL itself (one rank): VERDICT r3 item 5a
# --------------------------------------------------------------------------- #
def _worker_rccl_fused(rank, world, port):
    """MAKANI_AMD_DIST_FORCE_FUSED=1 with h = w = 1 on the RCCL backend: the distributed transforms run the FUSED pipeline
    (segmented FFT kernels writing per-peer slabs, the list all_to_all(async_op=True) on contiguous slab views per latitude
    chunk, the waits that order the compute stream behind RCCL's stream, the latitude-major Legendre GEMMs) with a process
    group of one rank — the call signatures, view contiguity and stream ordering the first multi-GPU run will meet —
    against the serial HIP transform, forward and gradients, fp32 and bf16 input, full size 721 x 1440 included"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MAKANI_AMD_DIST_FORCE_FUSED"] = "1"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import makani_amd as ma
        import makani_amd.distributed as thd
        from makani_amd import dist_pipeline as dp
        thd.init(None, None)
        for nlat, nlon, lmax, mmax, C, dt in ((37, 72, 12, 13, 6, torch.float32), (91, 360, 30, 31, 8, torch.float32),
                                              (721, 1440, 240, 241, 24, torch.float32), (240, 480, 240, 241, 16, torch.bfloat16)):
            kw = dict(lmax=lmax, mmax=mmax, grid="equiangular" if nlat % 2 else "legendre-gauss")
            fwd, inv = thd.DistributedRealSHT(nlat, nlon, **kw).to(dev), thd.DistributedInverseRealSHT(nlat, nlon, **kw).to(dev)
            assert dp.eligible(fwd, dt) and dp.eligible(inv, dt)
            S, I = ma.RealSHT(nlat, nlon, **kw).to(dev), ma.InverseRealSHT(nlat, nlon, **kw).to(dev)
            torch.manual_seed(3)
            x = torch.randn(1, C, nlat, nlon, device=dev).to(dt)
            G = torch.randn(1, C, lmax, mmax, dtype=torch.complex64, device=dev)
            coef = torch.tril(torch.randn(1, C, lmax, mmax, dtype=torch.complex64, device=dev))
            gy = torch.randn(1, C, nlat, nlon, device=dev)
            thd.COMM_STATS.clear()
            res = []
            for A, B_ in ((fwd, inv), (S, I)):
                xs = x.clone().requires_grad_(True)
                c = A(xs.float() if dt == torch.float32 else xs)
                (torch.view_as_real(c) * torch.view_as_real(G)).sum().backward()
                cf = coef.clone().requires_grad_(True)
                y = B_(cf)
                (y * gy).sum().backward()
                res.append((c.detach(), xs.grad.float(), y.detach(), cf.grad * torch.tril(torch.ones(lmax, mmax, device=dev))))
            torch.cuda.synchronize()
            assert thd.COMM_STATS and sum(v["all_to_alls"] for v in thd.COMM_STATS.values()) >= 4      # the collectives were issued
            for k, (a, b) in enumerate(zip(*res)):
                e = _rel(torch.view_as_real(a.cpu()) if a.is_complex() else a.cpu(), torch.view_as_real(b.cpu()) if b.is_complex() else b.cpu())
                assert e < (1e-5 if dt == torch.float32 else 1e-2), (nlat, nlon, k, e)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_fused_schedule_collectives_on_rccl_world1():
    mp.spawn(_worker_rccl_fused, args=(1, _free_port()), nprocs=1, join=True)


# --------------------------------------------------------------------------- #
# hipGraph capture of the distributed step's building blocks WITH their RCCL collectives (one rank): what bench.py does for the
# h x w step at N > 1 (the eager h4 w2 step is bound by the host: tools/shadow_rank.py, profiles/r05_shard_shapes.md)
# --------------------------------------------------------------------------- #
def _worker_rccl_graph(rank, world, port):
    """one captured "step" on the RCCL backend with a process group of one rank: forward + backward of the FUSED distributed
    SHT pair (segmented FFT kernels, the list all_to_all(async_op=True) per latitude chunk, its waits), of the distributed
    instance norm (all_gather of the statistics, all_reduce of the backward sums, no host read after the first call) and the
    gradient reduction hooks (async all_reduce issued from post-accumulate hooks, finished by the autograd engine's
    callback) — captured once with torch.cuda.graph, replayed twice on fresh inputs, against the eager results"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MAKANI_AMD_DIST_FORCE_FUSED"] = "1"
    # torch 2.10's process group hands the work of a CAPTURED send / recv collective to its watchdog thread, whose event query
    # then fails ("operation not permitted on an event last recorded in a capturing stream") and, by default, takes the process
    # down; with these two settings the watchdog thread just ends (bench.py sets them for its captured N > 1 step)
    os.environ["TORCH_NCCL_RETHROW_CUDA_ERRORS"] = "0"
    os.environ["TORCH_NCCL_ENABLE_MONITORING"] = "0"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import makani_amd.distributed as thd
        from makani_amd import dist_pipeline as dp
        from makani_amd import ops
        thd.init(None, None)
        nlat, nlon, lmax, mmax, C = 91, 360, 30, 31, 8
        kw = dict(lmax=lmax, mmax=mmax, grid="equiangular")
        fwd, inv = thd.DistributedRealSHT(nlat, nlon, **kw).to(dev), thd.DistributedInverseRealSHT(nlat, nlon, **kw).to(dev)
        assert dp.eligible(fwd, torch.float32) and dp.eligible(inv, torch.float32)
        model = torch.nn.Module()
        model.gamma = torch.nn.Parameter(torch.rand(C, device=dev) + 0.5)
        model.beta = torch.nn.Parameter(torch.randn(C, device=dev) * 0.1)
        model.big = torch.nn.Parameter(torch.randn(C, 1200, 300, device=dev) * 0.01)          # 11.5 MB: the async path of the reducer
        for p in model.parameters():
            p.is_shared_mp = ["spatial"]
        red = thd.GradReducer(model, comm=_OneRankTree)
        assert red.active
        torch.manual_seed(5)
        xs = torch.randn(1, C, nlat, nlon, device=dev)             # static input / cotangent buffers of the graph
        gs = torch.randn(1, C, nlat, nlon, device=dev)

        def step():
            for p in model.parameters():
                p.grad = None
            x = xs.clone().requires_grad_(True)
            h = ops.DistInstanceNormFn.apply(x, model.gamma, model.beta, 1e-6, True, dist.group.WORLD)
            y = inv(fwd(h)) + model.big[0, 0, :8].sum() * 1e-3
            (y * gs).sum().backward()
            return y.detach(), x.grad, model.gamma.grad, model.beta.grad, model.big.grad

        # two data sets, their eager results computed BEFORE the capture: no eager step runs between replays (an eager torch
        # reduction between two replays of a graph that contains the same reduction disturbed the replayed one here — torch's
        # two-stage sum, nothing of this package; the benchmark never interleaves the two either)
        data = [(torch.randn(1, C, nlat, nlon, device=dev), torch.randn(1, C, nlat, nlon, device=dev)) for _ in range(2)]
        refs = []
        for xa, ga in data:                                         # (also the warm-up: plans, shard counts, RCCL communicator)
            xs.copy_(xa)
            gs.copy_(ga)
            step()
            refs.append([t.clone() for t in step()])
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            outs = step()
        torch.cuda.synchronize()
        for it in (1, 0, 1):                                        # the replay follows the static buffers
            xs.copy_(data[it][0])
            gs.copy_(data[it][1])
            graph.replay()
            torch.cuda.synchronize()
            diffs = [float((a - b).abs().max()) for a, b in zip(outs, refs[it])]
            print(f"replay on data set {it}: max |graph - eager| of (y, gx, dgamma, dbeta, dbig) = {diffs}", flush=True)
            assert all(d == 0.0 for d in diffs), (it, diffs)
        torch.cuda.synchronize()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
    # no destroy_process_group(): tearing the communicator down under a live graph that holds its send / recv kernels hangs
    # (tools/probes/rccl_graph_probe.py); the process ends here
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def test_hipgraph_capture_of_the_distributed_blocks_with_rccl_world1():
    mp.spawn(_worker_rccl_graph, args=(1, _free_port()), nprocs=1, join=True)
