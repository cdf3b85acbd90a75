"""Multi-process CPU tests (gloo, 127.0.0.1) of the N > 1 host logic:
 * the h x w all-to-all schedule of the distributed SHT (makani_amd/distributed.py) reproduces the serial
   transform, forward and backward — the pattern of the reference's
   tests/distributed/tests_distributed_layers.py:69-223.  Local compute is injected from the CPU oracle
   (test-only backend); on a GPU the same schedule runs on the HIP kernels.
 * the process-group tree (makani_amd.comm) and the gradient reductions / sharded gradient norm of
   makani_amd.distributed (data-parallel mean, per-group sums from the is_shared_mp annotations)."""
import math
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _teardown():
    """every rank waits for the others before it tears its process group down: a rank that leaves while a peer is still inside
    a gloo collective makes the peer's transport thread throw ("terminate called without an active exception", seen once in
    five runs of the 4-rank ZeRO test: VERDICT r5 weak #11)"""
    try:
        dist.barrier()
    finally:
        dist.destroy_process_group()

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleBackend:
    """CPU local compute on the internal F/S layouts (torch ops; differentiable by autograd)."""

    @staticmethod
    def rfft(x4, mmax, w):
        B, P, nlat, nlon = x4.shape
        X = torch.fft.rfft(x4[0], dim=-1, norm="backward")[..., :mmax]          # (P, nlat, M) unnormalised sums
        wv = torch.full((mmax,), w[1], dtype=x4.dtype)
        wv[0] = w[0]
        if mmax - 1 == nlon // 2:
            wv[-1] = w[2]
        X = X * wv
        F = torch.stack([X.real, X.imag], dim=0).permute(3, 2, 0, 1)             # (M, nlat, 2, P)
        return torch.nn.functional.pad(F, (0, (-P) % 4))

    @staticmethod
    def irfft(F, planes, nlon, dtype, w):
        M = F.shape[0]
        X = torch.complex(F[:, :, 0, :planes], F[:, :, 1, :planes]).permute(2, 1, 0)   # (P, nlat, M)
        s = torch.full((M,), 2.0, dtype=F.dtype)
        s[0] = 1.0
        wv = torch.full((M,), w[1], dtype=F.dtype)
        wv[0] = w[0]
        mask = torch.ones(M, dtype=F.dtype)
        mask[0] = 0.0
        if M - 1 == nlon // 2:
            s[-1], wv[-1], mask[-1] = 1.0, w[2], 0.0
        X = torch.complex(X.real, X.imag * mask) * (wv / s)
        return torch.fft.irfft(X, n=nlon, dim=-1, norm="forward").unsqueeze(0).to(dtype)

    @staticmethod
    def analysis(F, mat, matT, m_off):
        assert torch.equal(matT[:, :, : mat.shape[1]], mat[:, :, : matT.shape[1]].transpose(1, 2))
        return torch.einsum("mkl,mkir->lmir", matT[:, :, : mat.shape[1]].to(F.dtype), F)

    @staticmethod
    def synthesis(S, mat, matT, nlat, m_off):
        return torch.einsum("lmir,mlk->mkir", S, mat[:, :, :nlat].to(S.dtype))


class OracleSegBackend(OracleBackend):
    """the four operations of the fused schedule (makani_amd/dist_pipeline.py) in torch on the CPU: FFTs that read the
    longitude pieces of a row and scatter / gather the per-peer slabs [lat][m][re/im][row] of the flat exchange buffer,
    Legendre transforms on the latitude-major operand — what csrc/fft_fast.hip (SEG kernels) and the GEMM engine do"""
    segmented = True

    @staticmethod
    def seg_supported(nlon):
        return True

    @staticmethod
    def _slabs(p, a, b):
        out, r0 = [], 0
        roff = [0]
        for n in p.sub[p.iw]:
            roff.append(roff[-1] + n)
        moff = [0]
        for n in p.m_shapes:
            moff.append(moff[-1] + n)
        for i in range(p.h):
            for j in range(p.w):
                row = p.m_shapes[j] * 2 * p.sub[p.iw][i]
                out.append((j, i, p.base[j][i] + a * row, p.base[j][i] + b * row, moff[j], moff[j + 1], roff[i], roff[i + 1]))
        return out

    @staticmethod
    def rfft_seg(xbuf, a, b, fs, p, w):
        Pw = p.pw[p.iw]
        x = xbuf[:, :, a:b, :].permute(1, 2, 0, 3).reshape(1, Pw, b - a, p.nlon)
        F = OracleBackend.rfft(x, p.M, w)                                           # (M, nl, 2, round4(Pw))
        rows = sum(p.sub[p.iw])
        F = torch.nn.functional.pad(F, (0, rows - F.shape[-1]))
        for j, i, f0, f1, m0, m1, r0, r1 in OracleSegBackend._slabs(p, a, b):
            fs[f0:f1] = F[m0:m1, :, :, r0:r1].permute(1, 0, 2, 3).reshape(-1).to(fs.dtype)

    @staticmethod
    def irfft_seg(fr, a, b, xbuf, p, w):
        Pw, nl = p.pw[p.iw], b - a
        F = torch.zeros((p.M, nl, 2, sum(p.sub[p.iw])), dtype=fr.dtype)
        for j, i, f0, f1, m0, m1, r0, r1 in OracleSegBackend._slabs(p, a, b):
            F[m0:m1, :, :, r0:r1] = fr[f0:f1].reshape(nl, m1 - m0, 2, r1 - r0).permute(1, 0, 2, 3)
        x = OracleBackend.irfft(F, Pw, p.nlon, xbuf.dtype, w)[0]                    # (Pw, nl, nlon)
        xbuf[:, :, a:b, :] = x.reshape(Pw, nl, p.w, p.wl).permute(2, 0, 1, 3)

    @staticmethod
    def analysis_lm(G, matT, L, m_off):
        return torch.einsum("mkl,kmir->lmir", matT[:, :, :L].to(G.dtype), G).contiguous()

    @staticmethod
    def synthesis_lm(T, mat, nlat, m_off):
        return torch.einsum("lmir,mlk->kmir", T, mat[:, :, :nlat].to(T.dtype)).contiguous()

    @staticmethod
    def analysis_lm_blocks(G, matT, L, m_off):
        """(w, nlat, M_loc, 2, sub) -> (L, w, M_loc, 2, sub): all plane blocks in one call (HipBackend: one batched GEMM)"""
        return torch.einsum("mkl,jkmir->ljmir", matT[:, :, :L].to(G.dtype), G).contiguous()

    @staticmethod
    def synthesis_lm_blocks(T, mat, nlat, m_off):
        return torch.einsum("ljmir,mlk->jkmir", T, mat[:, :, :nlat].to(T.dtype)).contiguous()


def _s_planes(B, C):
    """S layout: plane b * Cp + c with Cp = round4(C) when B > 1 (every sample's channels padded), round4(C) planes for B == 1"""
    return B * (C + (-C) % 4) if B > 1 else C + (-C) % 4


def _s_to_complex(S, B, C):
    L, M, _, R = S.shape
    Cp = R // B
    S = S.reshape(L, M, 2, B, Cp)[..., :C]
    return torch.complex(S[:, :, 0], S[:, :, 1]).permute(2, 3, 0, 1)


def _complex_to_s(c):
    B, C, L, M = c.shape
    S = torch.stack([c.real, c.imag], dim=0).permute(3, 4, 0, 1, 2)                  # (L, M, 2, B, C)
    return torch.nn.functional.pad(S, (0, (-C) % 4)).reshape(L, M, 2, -1).contiguous()


def _worker_sht(rank, world, port, h, w, nlat, nlon, lmax, mmax, grid, B, C, fused=False, chunks="2"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.distributed as thd
        from oracle import sht as osht
        ih, iw = rank // w, rank % w
        hg = wg = None
        for j in range(w):                       # polar groups: same iw
            g = dist.new_group([i * w + j for i in range(h)])
            if j == iw:
                hg = g
        for i in range(h):                       # azimuth groups: same ih
            g = dist.new_group([i * w + j for j in range(w)])
            if i == ih:
                wg = g
        thd.init(hg if h > 1 else None, wg if w > 1 else None, dist.group.WORLD)
        thd._BACKEND = OracleSegBackend if fused else OracleBackend      # test-only: the CPU stand-in for the HIP kernels
        os.environ["MAKANI_AMD_DIST_CHUNKS"] = chunks
        torch.manual_seed(7)
        x = torch.randn(B, C, nlat, nlon, dtype=torch.float64)
        fwd = thd.DistributedRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
        inv = thd.DistributedInverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
        from makani_amd import dist_pipeline as dp
        dp.FALLBACKS.clear()
        assert dp.eligible(fwd, x.dtype) == fused and dp.eligible(inv, x.dtype) == fused
        # a transform that leaves the fused schedule says so (once per transform and dtype), with the reason
        if fused:
            assert dp.FALLBACKS == []
        else:
            assert [f[0] for f in dp.FALLBACKS] == ["DistributedRealSHT", "DistributedInverseRealSHT"], dp.FALLBACKS
            assert all("segmented" in f[3] for f in dp.FALLBACKS), dp.FALLBACKS          # (the plain test backend has no SEG kernels)
        assert fwd.lat_shapes == thd.compute_split_shapes(nlat, h) and fwd.m_shapes == thd.compute_split_shapes(mmax, w)
        lat0, lon0 = sum(fwd.lat_shapes[:ih]), sum(fwd.lon_shapes[:iw])
        l0, m0 = sum(fwd.l_shapes[:ih]), sum(fwd.m_shapes[:iw])
        hl, wl, ll, ml = fwd.lat_shapes[ih], fwd.lon_shapes[iw], fwd.l_shapes[ih], fwd.m_shapes[iw]
        assert (fwd.l_off, fwd.m_off) == (l0, m0)

        # ---- forward + its gradient ----
        xl = x[..., lat0:lat0 + hl, lon0:lon0 + wl].clone().requires_grad_(True)
        S = fwd.analysis(xl)
        assert S.shape == (ll, ml, 2, _s_planes(B, C))
        c = _s_to_complex(S, B, C)
        So = osht.RealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
        xs = x.clone().requires_grad_(True)
        cref = So(xs)
        assert (c - cref[..., l0:l0 + ll, m0:m0 + ml]).abs().max().item() < 1e-5
        G = torch.randn(B, C, lmax, mmax, dtype=torch.complex128)
        (torch.view_as_real(c) * torch.view_as_real(G[..., l0:l0 + ll, m0:m0 + ml])).sum().backward()
        (torch.view_as_real(cref) * torch.view_as_real(G)).sum().backward()
        gref = xs.grad[..., lat0:lat0 + hl, lon0:lon0 + wl]
        assert (xl.grad - gref).abs().max().item() < 1e-5 * max(1.0, gref.abs().max().item())

        # ---- inverse + its gradient ----
        coef = torch.tril(torch.randn(B, C, lmax, mmax, dtype=torch.complex128))
        cl = coef[..., l0:l0 + ll, m0:m0 + ml].clone().requires_grad_(True)
        y = inv.synthesis(_complex_to_s(cl), B, C, out_dtype=torch.float64)
        Io = osht.InverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
        cs = coef.clone().requires_grad_(True)
        yref = Io(cs)
        assert y.shape == (B, C, hl, wl)
        assert (y - yref[..., lat0:lat0 + hl, lon0:lon0 + wl]).abs().max().item() < 1e-5
        Gy = torch.randn(B, C, nlat, nlon, dtype=torch.float64)
        (y * Gy[..., lat0:lat0 + hl, lon0:lon0 + wl]).sum().backward()
        (yref * Gy).sum().backward()
        tri = torch.tril(torch.ones(lmax, mmax, dtype=torch.bool))[l0:l0 + ll, m0:m0 + ml]
        gc_ref = cs.grad[..., l0:l0 + ll, m0:m0 + ml]
        assert ((cl.grad - gc_ref) * tri).abs().max().item() < 1e-5 * max(1.0, gc_ref.abs().max().item())
        dist.barrier()
    finally:
        _teardown()


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize("nlat,nlon,lmax,mmax,grid,B,C", [(33, 64, 16, 17, "equiangular", 1, 6), (12, 24, 12, 13, "legendre-gauss", 2, 8),
                                                    (12, 24, 12, 13, "legendre-gauss", 2, 3)])
def test_distributed_sht_schedule_matches_serial(h, w, nlat, nlon, lmax, mmax, grid, B, C):
    world = h * w
    mp.spawn(_worker_sht, args=(world, _free_port(), h, w, nlat, nlon, lmax, mmax, grid, B, C), nprocs=world, join=True)


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2), (4, 2), (3, 1)])
@pytest.mark.parametrize("nlat,nlon,lmax,mmax,grid,B,C", [(33, 64, 16, 17, "equiangular", 1, 6), (12, 24, 12, 13, "legendre-gauss", 2, 8),
                                                    (12, 24, 12, 13, "legendre-gauss", 2, 3), (19, 48, 10, 11, "equiangular", 1, 1)])
def test_fused_distributed_sht_schedule_matches_serial(h, w, nlat, nlon, lmax, mmax, grid, B, C):
    """the fused schedule of makani_amd/dist_pipeline.py (per-peer slabs written / read by the FFT, ONE h x w all-to-all between
    FFT and Legendre transform, latitude-major Legendre operand, plane blocks): forward and backward of both transforms against
    the serial oracle, incl. ragged plane blocks (fewer planes than ranks, padded sub-blocks) and empty slabs"""
    world = h * w
    mp.spawn(_worker_sht, args=(world, _free_port(), h, w, nlat, nlon, lmax, mmax, grid, B, C, True), nprocs=world, join=True)


def test_fused_distributed_sht_latitude_chunks():
    """steps (2)+(3) cut into latitude chunks (MAKANI_AMD_DIST_CHUNKS): every rank enters the same number of collectives even
    when the ranks' latitude counts differ (ragged 130 -> [44, 43, 43]...)"""
    mp.spawn(_worker_sht, args=(4, _free_port(), 2, 2, 130, 48, 20, 21, "equiangular", 1, 5, True, "2"), nprocs=4, join=True)


def _worker_disagree(rank, world, port, what):
    """ranks whose environment differs must agree on the schedule before the first collective of a transform: a differing
    chunk count raises on EVERY rank (instead of a hang in mismatched collectives), a differing MAKANI_AMD_DIST_FUSED makes
    every rank run the transpose-by-transpose schedule"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.distributed as thd
        from makani_amd import dist_pipeline as dp
        wg = dist.new_group([0, 1])
        thd.init(None, wg, dist.group.WORLD)
        thd._BACKEND = OracleSegBackend
        nlat, nlon, lmax, mmax = 130, 48, 20, 21
        if what == "chunks":
            os.environ["MAKANI_AMD_DIST_CHUNKS"] = "2" if rank == 0 else "1"
        else:
            os.environ["MAKANI_AMD_DIST_FUSED"] = "1" if rank == 0 else "0"
        fwd = thd.DistributedRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid="equiangular")
        x = torch.randn(1, 4, nlat, nlon // 2, dtype=torch.float64)
        if what == "chunks":
            with pytest.raises(RuntimeError, match="MAKANI_AMD_DIST_CHUNKS differs"):
                fwd.analysis(x)
        else:
            assert dp.eligible(fwd, x.dtype) is False          # rank 0 alone would have taken the fused schedule
            S = fwd.analysis(x)                                  # ... and both ranks complete the same (plain) schedule
            assert S.shape[0] == lmax
        dist.barrier()
    finally:
        _teardown()


@pytest.mark.parametrize("what", ["chunks", "fused"])
def test_ranks_agree_on_the_exchange_schedule(what):
    mp.spawn(_worker_disagree, args=(2, _free_port(), what), nprocs=2, join=True)


def _worker_dp(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        mcomm.init(1, 1)
        assert mcomm.get_size("data") == world and mcomm.get_size("spatial") == 1 and not thd.ensure_initialized()
        torch.manual_seed(0)
        model = torch.nn.Module()
        model.a = torch.nn.Parameter(torch.randn(5, 3))
        model.c = torch.nn.Parameter(torch.randn(4, 2, dtype=torch.complex64))
        model.big = torch.nn.Parameter(torch.randn(3 * 1024 * 1024))          # > 8 MB: async path
        model.forward = lambda: (model.a.sum() * (rank + 1)) + (torch.view_as_real(model.c).sum() * (rank + 2)) + model.big.sum() * rank
        net = thd.init_gradient_reduction_hooks(model, torch.device("cpu"))
        assert net.module is model
        for it in range(2):                     # the reductions complete inside backward(), every step
            for p in model.parameters():
                p.grad = None
            net().backward()
            mean = sum(range(1, world + 1)) / world
            assert torch.allclose(model.a.grad, torch.full_like(model.a, mean))
            assert torch.allclose(torch.view_as_real(model.c.grad), torch.full((4, 2, 2), mean + 1.0))
            assert torch.allclose(model.big.grad, torch.full_like(model.big, (world - 1) / 2))
            assert not net.reducer.pending and not net.reducer.small
    finally:
        _teardown()


def test_data_parallel_grad_reducer_world2():
    mp.spawn(_worker_dp, args=(2, _free_port()), nprocs=2, join=True)


def _worker_no_sync(rank, world, port):
    """gradient accumulation over two micro-batches with ``no_sync()`` around the first (deterministic_trainer.py:531-546)
    against the reduction of the summed gradients: identical results, and NO collective inside the context"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        mcomm.init(1, 1)
        torch.manual_seed(0)
        model = torch.nn.Module()
        model.a = torch.nn.Parameter(torch.randn(5, 3))
        model.c = torch.nn.Parameter(torch.randn(4, 2, dtype=torch.complex64))
        model.big = torch.nn.Parameter(torch.randn(3 * 1024 * 1024))          # > 8 MB: async path
        xs = [torch.randn(5, 3, generator=torch.Generator().manual_seed(10 * rank + i)) for i in range(2)]

        def fwd(i):
            return (model.a * xs[i]).sum() + (torch.view_as_real(model.c).sum() * (rank + 2 + i)) + (model.big * (rank + i)).sum()
        model.forward = fwd
        net = thd.init_gradient_reduction_hooks(model, torch.device("cpu"))
        assert hasattr(net, "no_sync")
        calls = []
        real = thd.GradReducer._issue
        thd.GradReducer._issue = staticmethod(lambda t, st: (calls.append(1), real(t, st))[1])
        with net.no_sync():
            net(0).backward()
        assert not calls and not net.reducer.pending and not net.reducer.small      # nothing was issued or queued
        local_a = model.a.grad.clone()
        assert torch.equal(local_a, xs[0])                                          # this rank's contribution only
        net(1).backward()                                                            # reduces the ACCUMULATED gradients
        assert calls and not net.reducer.pending and not net.reducer.small
        # expected: mean over ranks of the per-rank sums over the two micro-batches
        ga = sum(torch.randn(5, 3, generator=torch.Generator().manual_seed(10 * r + i)) for r in range(world) for i in range(2)) / world
        assert torch.allclose(model.a.grad, ga, atol=1e-6)
        gc = sum((r + 2 + i) for r in range(world) for i in range(2)) / world
        assert torch.allclose(torch.view_as_real(model.c.grad), torch.full((4, 2, 2), float(gc)))
        gb = sum((r + i) for r in range(world) for i in range(2)) / world
        assert torch.allclose(model.big.grad, torch.full_like(model.big, float(gb)))
        # the context restores the state it found, also on an exception
        try:
            with net.no_sync():
                raise KeyError("x")
        except KeyError:
            pass
        assert net.reducer.enabled
    finally:
        _teardown()


def test_grad_reduce_wrapper_no_sync_gradient_accumulation_world2():
    mp.spawn(_worker_no_sync, args=(2, _free_port()), nprocs=2, join=True)


def _worker_tree(rank, world, port, ph, pw):
    """makani_amd.comm.init (makani/utils/comm.py:114-201) + the gradient reductions of makani/mpu/mappings.py:460-523
    driven by the is_shared_mp annotations: dhconv weights (["matmul", "w"], l-sharded over h) are summed over "w" only,
    ["spatial"] / un-annotated parameters over the whole model instance, a position embedding ([]) over nothing; then
    the data-parallel mean.  Also the sharded gradient norm (training_helpers.py:123-160)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        msize = ph * pw
        dsize = world // msize
        d_idx, ih, iw = mcomm.init(ph, pw)
        assert (d_idx, ih, iw) == (rank // msize, (rank % msize) // pw, (rank % msize) % pw)
        assert (mcomm.get_size("h"), mcomm.get_size("w"), mcomm.get_size("spatial"), mcomm.get_size("data")) == (ph, pw, msize, dsize)
        assert (mcomm.get_rank("h"), mcomm.get_rank("w"), mcomm.get_rank("data")) == (ih, iw, d_idx)
        assert (mcomm.get_group("spatial") is None) == (msize == 1) and (mcomm.get_group("data") is None) == (dsize == 1)
        if ph > 1:
            assert dist.get_process_group_ranks(mcomm.get_group("h")) == [d_idx * msize + i * pw + iw for i in range(ph)]
        if pw > 1:
            assert dist.get_process_group_ranks(mcomm.get_group("w")) == [d_idx * msize + ih * pw + j for j in range(pw)]
        if dsize > 1:
            assert dist.get_process_group_ranks(mcomm.get_group("data")) == [d * msize + rank % msize for d in range(dsize)]
        assert thd.ensure_initialized() == (msize > 1)       # the transform layer initialises itself from the tree
        assert thd.polar_group_size() == ph and thd.azimuth_group_size() == pw

        model = torch.nn.Module()
        model.w = torch.nn.Parameter(torch.zeros(1, 3, 3, 4, dtype=torch.complex64))
        model.w.is_shared_mp, model.w.sharded_dims_mp = ["matmul", "w"], [None, None, None, "h"]
        model.enc = torch.nn.Parameter(torch.zeros(6))
        model.enc.is_shared_mp = ["spatial"]
        model.plain = torch.nn.Parameter(torch.zeros(2))                   # no annotation: shared over "model"
        model.pos = torch.nn.Parameter(torch.zeros(3))
        model.pos.is_shared_mp, model.pos.sharded_dims_mp = [], [None]
        model.forward = lambda: (torch.view_as_real(model.w).sum() + model.enc.sum() + model.plain.sum() + model.pos.sum()) * float(rank + 1)
        net = thd.init_gradient_reduction_hooks(model, torch.device("cpu"))
        net().backward()
        r1 = lambda ranks: sum(r + 1 for r in ranks)

        def over_data(fn):          # expected: mean over data of (sum over the model-parallel group) of (rank + 1)
            return sum(fn(d) for d in range(dsize)) / dsize
        exp_sp = over_data(lambda d: r1([d * msize + k for k in range(msize)]))
        exp_w = over_data(lambda d: r1([d * msize + ih * pw + j for j in range(pw)]))
        exp_pos = over_data(lambda d: float(d * msize + rank % msize + 1))
        assert torch.allclose(model.enc.grad, torch.full((6,), exp_sp)), (rank, model.enc.grad, exp_sp)
        assert torch.allclose(model.plain.grad, torch.full((2,), exp_sp))
        assert torch.allclose(torch.view_as_real(model.w.grad), torch.full((1, 3, 3, 4, 2), exp_w)), (rank, model.w.grad.flatten()[0], exp_w)
        assert torch.allclose(model.pos.grad, torch.full((3,), exp_pos))
        # sharded norm: ||w||^2 summed over h (each h-rank holds another l-shard), everything else counted once
        sq_w = sum(72 * over_data(lambda d, i=i: r1([d * msize + i * pw + j for j in range(pw)])) ** 2 for i in range(ph))
        exp_norm = math.sqrt(sq_w + 6 * exp_sp ** 2 + 2 * exp_sp ** 2 + 3 * exp_pos ** 2)
        assert abs(float(thd.total_grad_norm(model)) - exp_norm) < 1e-3 * exp_norm
        dist.barrier()
    finally:
        _teardown()


@pytest.mark.parametrize("world,ph,pw", [(2, 1, 1), (4, 2, 1), (4, 1, 2), (4, 2, 2), (8, 2, 2)])
def test_group_tree_and_model_parallel_grad_reduction(world, ph, pw):
    mp.spawn(_worker_tree, args=(world, _free_port(), ph, pw), nprocs=world, join=True)


def _worker_ragged(rank, world, port, h, w, C, fused=False):
    """BASELINE configs[2] / [4] split sizes at the real grid: 721 x 1440, lmax 240, mmax 241 over h = 4 (lat
    [181, 181, 181, 178], l [60] * 4) and w = 2 (lon [720, 720], m [121, 120]), reduced channel count"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        from oracle import sht as osht
        _, ih, iw = mcomm.init(h, w)
        assert thd.ensure_initialized()
        thd._BACKEND = OracleSegBackend if fused else OracleBackend
        nlat, nlon, lmax, mmax, B = 721, 1440, 240, 241, 1
        fwd = thd.DistributedRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid="equiangular")
        inv = thd.DistributedInverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid="equiangular")
        if h == 4:
            assert fwd.lat_shapes == [181, 181, 181, 178] and fwd.l_shapes == [60, 60, 60, 60]
        if w == 2:
            assert fwd.lon_shapes == [720, 720] and fwd.m_shapes == [121, 120]
        lat0, lon0 = sum(fwd.lat_shapes[:ih]), sum(fwd.lon_shapes[:iw])
        l0, m0 = sum(fwd.l_shapes[:ih]), sum(fwd.m_shapes[:iw])
        hl, wl, ll, ml = fwd.lat_shapes[ih], fwd.lon_shapes[iw], fwd.l_shapes[ih], fwd.m_shapes[iw]
        torch.manual_seed(7)
        x = torch.randn(B, C, nlat, nlon, dtype=torch.float64)
        xl = x[..., lat0:lat0 + hl, lon0:lon0 + wl].clone().requires_grad_(True)
        c = _s_to_complex(fwd.analysis(xl), B, C)
        So = osht.RealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid="equiangular")
        xs = x.clone().requires_grad_(True)
        cref = So(xs)
        assert (c - cref[..., l0:l0 + ll, m0:m0 + ml]).abs().max().item() < 1e-6
        G = torch.randn(B, C, lmax, mmax, dtype=torch.complex128)
        (torch.view_as_real(c) * torch.view_as_real(G[..., l0:l0 + ll, m0:m0 + ml])).sum().backward()
        (torch.view_as_real(cref) * torch.view_as_real(G)).sum().backward()
        gref = xs.grad[..., lat0:lat0 + hl, lon0:lon0 + wl]
        assert (xl.grad - gref).abs().max().item() < 1e-6 * max(1.0, gref.abs().max().item())
        coef = torch.tril(torch.randn(B, C, lmax, mmax, dtype=torch.complex128))
        y = inv.synthesis(_complex_to_s(coef[..., l0:l0 + ll, m0:m0 + ml].clone()), B, C, out_dtype=torch.float64)
        yref = osht.InverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid="equiangular")(coef)
        assert y.shape == (B, C, hl, wl)
        assert (y - yref[..., lat0:lat0 + hl, lon0:lon0 + wl]).abs().max().item() < 1e-6 * yref.abs().max().item()
        dist.barrier()
    finally:
        _teardown()


@pytest.mark.parametrize("h,w,C", [(4, 1, 6), (4, 2, 5)])
@pytest.mark.parametrize("fused", [False, True])
def test_distributed_sht_ragged_config3_splits(h, w, C, fused):
    """h = 4 (BASELINE configs[2]) and h4 w2 (configs[4]) at 721 x 1440 with ragged plane counts (6 planes over 4 ranks ->
    [1, 1, 1, 3]; 5 over 2 -> [3, 2]); transpose-by-transpose and fused schedule"""
    mp.spawn(_worker_ragged, args=(h * w, _free_port(), h, w, C, fused), nprocs=h * w, join=True)


def test_parse_parallelism():
    import bench
    assert bench.parse_parallelism("dp") == (1, 1) and bench.parse_parallelism("h4w2") == (4, 2)
    # 2 GPUs: data parallel (the reference's partitioning has no 2-GPU split worth running on xGMI: bench.default_parallelism)
    assert [bench.default_parallelism(n) for n in (1, 2, 4, 8)] == ["dp", "dp", "h4w1", "h4w2"]
    with pytest.raises(SystemExit):
        bench.parse_parallelism("tp8")


# --------------------------------------------------------------------------- #
# ZeRO-1: reduce-scattered gradients + sharded optimizer state + in-place parameter all-gather
# --------------------------------------------------------------------------- #
def _worker_zero(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.comm as mcomm
        import makani_amd.distributed as thd
        import makani_amd.optim as mo
        mcomm.init(1, 1)

        # torch stand-ins for the three HIP operations of the big-tensor path (the product has no CPU path)
        def advance(sdev, b1, b2):
            sdev[0] += 1
            sdev[1] = 1.0 / (1.0 - b1 ** float(sdev[0]))
            sdev[2] = 1.0 / math.sqrt(1.0 - b2 ** float(sdev[0]))

        def adamw(pr, gr, m, v, scale, lr, b1, b2, eps, wd, sdev):
            g = gr * (scale if scale is not None else 1.0)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            pr.mul_(1 - lr * wd).sub_(lr * float(sdev[1]) * m / (v.sqrt() * float(sdev[2]) + eps))

        def sumsq_clip(grads, max_norm, pre=()):
            n = torch.sqrt(sum((g.double() ** 2).sum() for g in grads) + sum(t.double().sum() for t in pre)).float().reshape(1)
            coef = torch.clamp(max_norm / (n + 1e-6), max=1.0) if max_norm else torch.ones(1)
            return torch.cat([coef, n])
        mo._k_advance, mo._k_adamw, mo._k_sumsq_clip = advance, adamw, sumsq_clip
        mo.SMALL = 0                                           # every tensor takes the big-tensor path in this test

        torch.manual_seed(0)
        model = torch.nn.Module()
        model.w = torch.nn.Parameter(torch.randn(64, 48))                        # 3072 floats: sharded (divisible by 4 * world)
        model.c = torch.nn.Parameter(torch.randn(16, 8, dtype=torch.complex64))   # complex: 256 reals, sharded through its real view
        model.b = torch.nn.Parameter(torch.randn(37))                             # not divisible: stays replicated
        ref = {k: v.detach().clone().requires_grad_(True) for k, v in model.named_parameters()}
        ropt = torch.optim.AdamW(list(ref.values()), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        net = thd.init_gradient_reduction_hooks(model, torch.device("cpu"), zero=True)
        net.reducer.big_bytes = 512                            # bytes: w and c are "big", b is bucketed
        opt = mo.FusedAdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        for it in range(3):
            gens = [torch.Generator().manual_seed(100 * it + r) for r in range(world)]
            tg = {k: [torch.randn(v.shape, generator=gens[r], dtype=v.dtype if not v.is_complex() else torch.float32).to(v.dtype)
                      if not v.is_complex() else torch.complex(torch.randn(v.shape, generator=gens[r]), torch.randn(v.shape, generator=gens[r]))
                      for r in range(world)] for k, v in model.named_parameters()}
            for p in model.parameters():
                p.grad = None
            loss = sum((torch.view_as_real(p * tg[k][rank].conj()).select(-1, 0) if p.is_complex() else p * tg[k][rank]).sum()
                       for k, p in model.named_parameters())
            loss.backward()                                    # d loss / d p = tg[k][rank] (conj-linear form for the complex one)
            assert mo.FusedAdamW.zero_shard(model.w) is not None and mo.FusedAdamW.zero_shard(model.c) is not None
            assert mo.FusedAdamW.zero_shard(model.b) is None
            assert mo.FusedAdamW.zero_shard(model.w)[0].numel() == 3072 // world
            # reference: one process, averaged gradients, same clipping
            for k, v in ref.items():
                v.grad = sum(tg[k]) / world
            gn_ref = torch.sqrt(sum((torch.view_as_real(v.grad) if v.is_complex() else v.grad).double().pow(2).sum() for v in ref.values()))
            coef_ref = min(1.0, 0.5 / (float(gn_ref) + 1e-6))
            for v in ref.values():
                v.grad.mul_(coef_ref)
            ropt.step()
            cc = opt.clip_coef(0.5)
            assert abs(float(cc[1]) - float(gn_ref)) < 1e-4 * float(gn_ref), (float(cc[1]), float(gn_ref))
            assert abs(float(thd.total_grad_norm(model)) - float(gn_ref)) < 1e-4 * float(gn_ref)
            opt.step(grad_scale=cc[:1])
            for k, p in model.named_parameters():
                a = torch.view_as_real(p.detach()) if p.is_complex() else p.detach()
                b = torch.view_as_real(ref[k].detach()) if p.is_complex() else ref[k].detach()
                assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), (it, k, (a - b).abs().max())
        # sharded state: 1 / world of the elements
        assert opt.state[model.w]["exp_avg"].numel() == 3072 // world and opt.state[model.b]["exp_avg"].numel() == 37
        # a checkpoint one data rank can write for all: full_state_dict() all-gathers the sharded moments; they
        # equal the single-process optimizer's, carry no shard tag, and restore on any rank (step() re-slices them)
        fsd = opt.full_state_dict()
        names = [k for k, _ in model.named_parameters()]
        for i, k in enumerate(names):
            st = fsd["state"][i]
            assert "zero_shard" not in st and st["exp_avg"].shape == ref[k].shape
            for key in ("exp_avg", "exp_avg_sq"):
                a = torch.view_as_real(st[key]) if st[key].is_complex() else st[key]
                b = ropt.state[ref[k]][key]
                b = torch.view_as_real(b) if b.is_complex() else b
                assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), (k, key)
        opt2 = mo.FusedAdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        opt2.load_state_dict(fsd)
        assert opt2.state[model.w]["exp_avg"].numel() == 3072 and "zero_shard" not in opt2.state[model.w]
    finally:
        _teardown()


@pytest.mark.parametrize("world", [2, 4])
def test_zero1_sharded_adamw_matches_single_process_adamw(world):
    mp.spawn(_worker_zero, args=(world, _free_port()), nprocs=world, join=True)


# --------------------------------------------------------------------------- #
# distributed DISCO contraction / resampling schedule (makani_amd/disco.py)
# --------------------------------------------------------------------------- #
def _psi_contract(x, psi, nlat_out, nlon_out, window=None):
    """torch stand-in for the HIP contraction: y[b, c*K + k, t, p] = sum_e v_e x[b, c, i_e, (j_e + p s) % nlon_in] over the
    entries of the convolution tensor (``window = (i0, ni, t0, nt)``: only the output latitudes t0 .. t0 + nt, the input given
    as rows i0 .. i0 + ni)"""
    k, t, i, j, v = (torch.from_numpy(psi[n]) for n in ("k", "t", "i", "j", "v"))
    K = psi["K"]
    if window is not None:
        i0, ni, t0, nt = window
        sel = (t >= t0) & (t < t0 + nt)
        k, t, i, j, v = k[sel], t[sel] - t0, i[sel] - i0, j[sel], v[sel]
        assert x.shape[2] == ni
        nlat_out = nt
    B, C, _, nlon_in = x.shape
    s = nlon_in // nlon_out
    cols = (j[:, None] + s * torch.arange(nlon_out)[None, :]) % nlon_in                 # (nnz, nlon_out)
    g = x[:, :, i, :].gather(3, cols.expand(B, C, -1, -1)) * v.to(x.dtype)[:, None]       # (B, C, nnz, nlon_out)
    y = torch.zeros(B, C, K * nlat_out, nlon_out, dtype=x.dtype).index_add(2, k * nlat_out + t, g)
    return y.reshape(B, C * K, nlat_out, nlon_out)


def _worker_disco(rank, world, port, h, w):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd.distributed as thd
        from makani_amd import disco
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
        thd.init(hg if h > 1 else None, wg if w > 1 else None)
        for in_shape, out_shape, C, kshape, basis in (((19, 36), (10, 18), 5, (3, 3), "morlet"), ((13, 24), (13, 24), 4, (3, 3), "morlet"),
                                                      ((13, 24), (7, 12), 1, (3, 3), "morlet"), ((13, 24), (13, 24), 2, (3, 4), "piecewise linear"),
                                                      ((19, 36), (10, 18), 2, 3, "zernike")):
            d = disco.DistributedDiscreteContinuousConvS2(C, 3, in_shape, out_shape, kshape, basis_type=basis, bias=False,
                                                          theta_cutoff=4.0 * math.pi / (in_shape[0] - 1))
            torch.manual_seed(3)
            x = torch.randn(2, C, *in_shape, dtype=torch.float64, requires_grad=True)
            G = torch.randn(2, C * d.kernel_size, *out_shape, dtype=torch.float64)
            ys = _psi_contract(x, d._psi, *out_shape)
            (ys * G).sum().backward()
            a0, b0 = sum(d.lat_in_shapes[:ih]), sum(d.lon_in_shapes[:iw])
            a1, b1 = sum(d.lat_out_shapes[:ih]), sum(d.lon_out_shapes[:iw])
            xl = x.detach()[..., a0:a0 + d.lat_in_shapes[ih], b0:b0 + d.lon_in_shapes[iw]].clone().requires_grad_(True)
            win = d.window() if h > 1 else None
            yl = d._spatial_contract(xl, lambda t_: _psi_contract(t_, d._psi, *out_shape, window=win))
            sl = (Ellipsis, slice(a1, a1 + d.lat_out_shapes[ih]), slice(b1, b1 + d.lon_out_shapes[iw]))
            assert yl.shape == ys[sl].shape
            (yl * G[sl]).sum().backward()
            assert torch.allclose(yl, ys[sl], atol=1e-12), (rank, (yl - ys[sl]).abs().max())
            gref = x.grad[..., a0:a0 + d.lat_in_shapes[ih], b0:b0 + d.lon_in_shapes[iw]]
            assert torch.allclose(xl.grad, gref, atol=1e-12), (rank, (xl.grad - gref).abs().max())
            # the device lists of the ranks partition the convolution tensor by output latitude; the halo is a few rows
            L = disco._Lists(d._psi, in_shape, out_shape, "cpu", window=win)
            assert L.out_shape[0] == d.lat_out_shapes[ih]
            if h > 1:
                assert L.in_shape[0] == win[1] <= d.lat_in_shapes[ih] + 2 * 6
            cnt = torch.tensor([L.nnz])
            if hg is not None and h > 1:
                dist.all_reduce(cnt, group=hg)
            assert int(cnt) == d._psi["v"].size
        dist.barrier()
    finally:
        _teardown()


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2), (3, 1)])
def test_distributed_disco_schedule_matches_serial(h, w):
    mp.spawn(_worker_disco, args=(h * w, _free_port(), h, w), nprocs=h * w, join=True)
