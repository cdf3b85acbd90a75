"""GPU parity tests for the raw HIP kernels, called through the C ABI (ctypes),
against fp64 CPU references (torch / numpy) on seeded inputs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


# --------------------------------------------------------------------------- #
# batched GEMM engine
# --------------------------------------------------------------------------- #
def _store_operand(x, kc):
    """x (batch, rows, K) -> storage + (row_stride, k_stride, batch_stride); unit stride along k (kc) or rows."""
    b, r, k = x.shape
    if kc:
        kp = (k + 3) // 4 * 4 + 4
        t = torch.full((b, r, kp), 3.0)          # finite garbage inside the float4 reach: must be masked by K
        t[:, :, :k] = x
        return t, kp, 1, r * kp
    assert r % 4 == 0
    t = x.transpose(1, 2).contiguous()           # (b, k, r)
    return t, 1, r, k * r


def _tri_mask(mode, batch, M, K, inner):
    from makani_amd import _lib
    rows = torch.ones(batch, M, dtype=torch.bool)
    ks = torch.ones(batch, K, dtype=torch.bool)
    t = torch.arange(batch) // inner
    i = torch.arange(M)[None, :]
    k = torch.arange(K)[None, :]
    if mode == _lib.TRI_ROW_GE:
        pass  # rows i < t are either skipped (tile) or computed; compared only where i >= t
    if mode == _lib.TRI_ROW_LE:
        rows = i <= t[:, None]
    if mode == _lib.TRI_K_GE:
        ks = k >= t[:, None]
    if mode == _lib.TRI_K_LE:
        ks = k <= t[:, None]
    return rows, ks


def _launch_gemm(g, cplx, engine, a_planes=None):
    from makani_amd._lib import lib, check
    if engine == "fp32":
        rc = (lib().mk_cgemm_batched if cplx else lib().mk_sgemm_batched)(C.byref(g), C.c_void_p(0))
    elif engine.endswith("v2"):       # second-generation kernels (csrc/xgemm2.hip)
        limbs = 3 if engine == "x6v2" else 2
        if cplx:
            rc = lib().mk_cgemm_split2_batched(C.byref(g), limbs, C.c_void_p(0))
        else:
            pl = a_planes
            rc = lib().mk_sgemm_presplit_batched(C.byref(g), C.c_void_p(pl.data_ptr()), pl.stride(0), pl.stride(1), pl.stride(2),
                                                 limbs, None, None, 0, C.c_void_p(0))
    else:
        rc = (lib().mk_cgemm_split_batched if cplx else lib().mk_sgemm_split_batched)(C.byref(g), 3 if engine == "x6" else 2, C.c_void_p(0))
    check(rc, "gemm")


ENGINE_TOL = {"fp32": 2e-6, "x6": 2e-6, "x3": 2e-5, "x6v2": 2e-6, "x3v2": 2e-5}


def _sgemm_cases():
    """every combination an engine actually serves (the wide shapes exist for the big-tile kernels of the default three-limb
    engines; the pre-split kernel takes row-contiguous operands only — the Legendre layouts)"""
    out = []
    for (M, N, K, batch) in [(68, 132, 37, 11), (200, 72, 129, 5), (12, 300, 16, 9),
                             # wide problems: the 256 x 256-tile kernel of the split-bf16 engine (Legendre shapes)
                             (240, 768, 100, 250), (300, 520, 70, 4)]:
        for tri in (0, 1, 2, 3, 4):
            for (a_kc, b_kc) in [(True, True), (True, False), (False, True), (False, False)]:
                for engine in ("fp32", "x6", "x3", "x6v2", "x3v2"):
                    if N >= 512 and engine not in ("x6", "x6v2"):
                        continue
                    if engine.endswith("v2") and (a_kc or b_kc):
                        continue
                    out.append((engine, a_kc, b_kc, tri, M, N, K, batch))
    return out


@pytest.mark.parametrize("engine,a_kc,b_kc,tri,M,N,K,batch", _sgemm_cases())
def test_sgemm_batched(engine, a_kc, b_kc, tri, M, N, K, batch):
    from makani_amd import _lib
    from makani_amd._lib import MkGemm, lib, check
    torch.manual_seed(M * 1000 + N + tri)
    A = torch.randn(batch, M, K)
    B = torch.randn(batch, N, K)
    rows_ok, k_ok = _tri_mask(tri, batch, M, K, 1)
    ref = torch.einsum("bik,bjk->bij", A.double() * k_ok[:, None, :], B.double())
    At, a_row, a_k, a_b = _store_operand(A, a_kc)
    Bt, b_col, b_k, b_b = _store_operand(B, b_kc)
    Ad, Bd = At.contiguous().to(_dev()), Bt.contiguous().to(_dev())
    Cd = torch.full((batch, M, N + 4), -123.0, device=_dev())
    g = MkGemm()
    for f, _ in MkGemm._fields_:
        setattr(g, f, 0)
    g.A, g.B, g.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
    g.a_batch, g.a_row, g.a_k = a_b, a_row, a_k
    g.b_batch, g.b_col, g.b_k = b_b, b_col, b_k
    g.c_batch, g.c_row, g.c_col = M * (N + 4), N + 4, 1
    g.M, g.N, g.K, g.batch, g.inner, g.tri_mode = M, N, K, batch, 1, tri
    planes = None
    if engine.endswith("v2"):
        from makani_amd import ops
        planes = ops.limb_planes(Ad)              # (3, batch, K, round8(M)) bf16
        g.A = 0                                   # the fp32 A is not read
    _launch_gemm(g, False, engine, planes)
    torch.cuda.synchronize()
    out = Cd.cpu()
    assert (out[:, :, N:] == -123.0).all(), "wrote outside the N extent"
    got = out[:, :, :N].double()
    if tri == _lib.TRI_ROW_GE:
        valid = (torch.arange(M)[None, :] >= torch.arange(batch)[:, None])
    else:
        valid = rows_ok
    vm = valid[:, :, None].expand_as(ref)
    err = ((got - ref)[vm]).norm() / ref[vm].norm()
    assert torch.isfinite(got[vm]).all()
    assert err < ENGINE_TOL[engine], err
    if tri == _lib.TRI_ROW_LE:
        assert (out[:, :, :N][~vm] == -123.0).all(), "rows beyond the triangular bound must not be written"


def _cgemm_cases():
    out = []
    for (M, N, K, outer, inner) in [(36, 140, 24, 7, 2), (100, 64, 52, 9, 1), (200, 132, 40, 210, 1)]:
        for (tri, conj_a, conj_b, beta) in [(3, 0, 0, 0), (3, 0, 1, 0), (4, 1, 0, 0), (4, 1, 0, 1), (0, 1, 1, 1)]:
            for (a_kc, b_kc) in [(True, False), (True, True), (False, False), (False, True)]:
                for engine in ("fp32", "x6", "x3", "x6v2", "x3v2"):
                    if outer > 100 and engine not in ("x6", "x6v2"):      # the many-row-tile shape exists for the default engines
                        continue
                    out.append((engine, a_kc, b_kc, tri, conj_a, conj_b, beta, M, N, K, outer, inner))
    return out


@pytest.mark.parametrize("engine,a_kc,b_kc,tri,conj_a,conj_b,beta,M,N,K,outer,inner", _cgemm_cases())
def test_cgemm_batched(engine, a_kc, b_kc, tri, conj_a, conj_b, beta, M, N, K, outer, inner):
    from makani_amd import _lib
    from makani_amd._lib import MkGemm, lib, check
    torch.manual_seed(17 + M + tri)
    batch = outer * inner
    A = torch.randn(batch, M, K, dtype=torch.complex128)
    B = torch.randn(batch, N, K, dtype=torch.complex128)
    C0 = torch.randn(batch, M, N, dtype=torch.complex128)
    rows_ok, k_ok = _tri_mask(tri, batch, M, K, inner)
    Ae = A.conj() if conj_a else A
    Be = B.conj() if conj_b else B
    ref = torch.einsum("bik,bjk->bij", Ae * k_ok[:, None, :], Be)
    if beta:
        ref = ref + C0

    def planar(x, kc):           # (batch, rows, K) complex -> (batch, 2, ...) float32 planes
        xr = torch.stack([x.real, x.imag], dim=1).float()
        if not kc:
            xr = xr.transpose(2, 3)
        return xr.contiguous()

    Ap, Bp = planar(A, a_kc), planar(B, b_kc)
    Cp = torch.stack([C0.real, C0.imag], dim=1).float().contiguous()
    for t, kc in ((Ap, a_kc), (Bp, b_kc)):
        assert t.shape[-1] % 4 == 0 and t.shape[-2] % 4 == 0
    Ad, Bd, Cd = Ap.to(_dev()), Bp.to(_dev()), Cp.to(_dev())
    g = MkGemm()
    for f, _ in MkGemm._fields_:
        setattr(g, f, 0)
    g.A, g.B, g.C = Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr()
    pa, pb = M * K, N * K
    # batch index = outer * inner + in  ->  linear: use a_batch = inner * 2*pa, a_inner = 2*pa
    g.a_batch, g.a_inner, g.a_im = inner * 2 * pa, 2 * pa, pa
    g.b_batch, g.b_inner, g.b_im = inner * 2 * pb, 2 * pb, pb
    g.c_batch, g.c_inner, g.c_im = inner * 2 * M * N, 2 * M * N, M * N
    g.a_row, g.a_k = (K, 1) if a_kc else (1, M)
    g.b_col, g.b_k = (K, 1) if b_kc else (1, N)
    g.c_row, g.c_col = N, 1
    g.M, g.N, g.K, g.batch, g.inner, g.tri_mode = M, N, K, batch, inner, tri
    g.conj_a, g.conj_b, g.beta = conj_a, conj_b, beta
    _launch_gemm(g, True, engine)
    torch.cuda.synchronize()
    out = Cd.cpu().double()
    got = torch.complex(out[:, 0], out[:, 1])
    vm = rows_ok[:, :, None].expand_as(ref)
    err = (got - ref)[vm].abs().pow(2).sum().sqrt() / ref[vm].abs().pow(2).sum().sqrt()
    assert err < ENGINE_TOL[engine], err
    if tri == _lib.TRI_ROW_LE:   # untouched rows keep their input
        assert ((got - C0)[~vm].abs() < 1e-6).all()


# --------------------------------------------------------------------------- #
# FFT
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,nlat,nlon,mmax", [(1, 3, 5, 16, 9), (2, 5, 9, 24, 13), (1, 4, 19, 72, 20),
                                              (1, 2, 33, 128, 65), (1, 2, 8, 480, 241), (1, 1, 9, 1440, 241),
                                              (1, 2, 7, 28, 15), (1, 1, 20, 360, 181), (2, 3, 19, 128, 40), (1, 5, 33, 480, 100)])
def test_rfft_rows(dtype, B, C, nlat, nlon, mmax):
    from makani_amd import ops
    torch.manual_seed(nlon)
    x = torch.randn(B, C, nlat, nlon).to(dtype)
    Cp = ops.round4(C)
    c = 2 * math.pi / nlon
    F = ops.rfft_rows(x.to(_dev()), mmax, Cp, (c, c, c))
    torch.cuda.synchronize()
    F = F.cpu()
    ref = 2 * math.pi * torch.fft.rfft(x.double(), dim=-1, norm="forward")[..., :mmax]     # (B,C,nlat,M)
    assert F.shape == (mmax, nlat, 2, B * Cp)
    got = torch.complex(F[:, :, 0], F[:, :, 1]).view(mmax, nlat, B, Cp)[..., :C].permute(2, 3, 1, 0)
    assert rel_l2(got, ref) < 3e-6
    # weighted variant = adjoint of irfft: w = (1, 2, 1)
    F2 = ops.rfft_rows(x.to(_dev()), mmax, Cp, (1.0, 2.0, 1.0)).cpu()
    got2 = torch.complex(F2[:, :, 0], F2[:, :, 1]).view(mmax, nlat, B, Cp)[..., :C].permute(2, 3, 1, 0)
    ref2 = torch.fft.rfft(x.double(), dim=-1)[..., :mmax] * 2
    ref2[..., 0] /= 2
    if mmax - 1 == nlon // 2:
        ref2[..., -1] /= 2
    assert rel_l2(got2, ref2) < 3e-6
    assert (got2[..., 0].imag == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,nlat,nlon,mmax", [(1, 3, 5, 16, 9), (2, 5, 9, 24, 10), (1, 4, 19, 72, 20),
                                              (1, 2, 33, 128, 65), (1, 2, 8, 480, 241), (1, 1, 9, 1440, 241),
                                              (1, 3, 21, 360, 100), (2, 2, 17, 72, 37), (1, 2, 12, 28, 15)])
def test_irfft_rows(dtype, B, C, nlat, nlon, mmax):
    from makani_amd import ops
    torch.manual_seed(nlon + 1)
    Cp = ops.round4(C)
    X = torch.randn(B, C, nlat, mmax, dtype=torch.complex128)
    F = torch.full((mmax, nlat, 2, B, Cp), float("nan"))          # pad rows are never read
    F[:, :, 0, :, :C] = X.real.permute(3, 2, 0, 1).float()
    F[:, :, 1, :, :C] = X.imag.permute(3, 2, 0, 1).float()
    F = F.view(mmax, nlat, 2, B * Cp).contiguous()
    x = ops.irfft_rows(F.to(_dev()), B, C, nlon, dtype, (1.0, 2.0, 1.0))
    torch.cuda.synchronize()
    Xr = X.clone()
    Xr[..., 0] = Xr[..., 0].real.to(torch.complex128)
    if mmax - 1 == nlon // 2:
        Xr[..., -1] = Xr[..., -1].real.to(torch.complex128)
    ref = torch.fft.irfft(Xr, n=nlon, dim=-1, norm="forward")
    tol = 3e-6 if dtype == torch.float32 else 6e-3
    assert rel_l2(x.cpu(), ref) < tol
    assert x.dtype == dtype


def test_fft_adjoint_pair_fullsize():
    """<rfft(x), Y> == <x, rfft^T(Y)> at the BASELINE row length (size-independent property)."""
    from makani_amd import ops
    torch.manual_seed(5)
    B, C, nlat, nlon, mmax = 1, 4, 16, 1440, 241
    c = 2 * math.pi / nlon
    x = torch.randn(B, C, nlat, nlon, device=_dev())
    F = ops.rfft_rows(x, mmax, 4, (c, c, c))
    Y = torch.randn_like(F)
    xt = ops.irfft_rows(Y, B, C, nlon, torch.float32, (c, c, c))
    Yv = Y.clone()
    Yv[0, :, 1] = 0                   # Im of m=0 carries no information
    lhs = (F.double() * Yv.double()).sum().item()
    rhs = (x.double() * xt.double()).sum().item()
    assert abs(lhs - rhs) / abs(lhs) < 1e-5


# --------------------------------------------------------------------------- #
# layout changes
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("cin,cout,L", [(6, 5, 9), (32, 32, 8), (40, 70, 33)])
def test_weight_layout_roundtrip(cin, cout, L):
    from makani_amd import ops
    w = torch.randn(1, cin, cout, L, dtype=torch.complex64)
    W = ops.weight_to_wlayout(w.to(_dev())).cpu()
    cip, cop = ops.round4(cin), ops.round4(cout)
    assert W.shape == (L, 2, cip, cop)
    ref = torch.zeros(L, 2, cip, cop)
    ref[:, 0, :cin, :cout] = w[0].real.permute(2, 0, 1)
    ref[:, 1, :cin, :cout] = w[0].imag.permute(2, 0, 1)
    assert torch.equal(W, ref)
    back = ops.wlayout_to_weight_grad(W.to(_dev()), cin, cout).cpu()
    assert torch.equal(back, w)


@pytest.mark.parametrize("B,cin,cout,L,M", [(1, 8, 12, 9, 10), (2, 36, 140, 20, 21), (1, 128, 64, 40, 41), (1, 384, 384, 6, 241)])
@pytest.mark.parametrize("engine", ["x6", "x3"])
def test_dhconv_native_weight_order_matches_planar(B, cin, cout, L, M, engine, monkeypatch):
    """the dhconv weight used IN PLACE (complex64 (1, Cin, Cout, L) whose memory order is [l][i][o], interleaved B operand
    of the split engine; gradient written interleaved by the GEMM epilogue) against the planar W-layout path: the same
    limbs meet in the same order, so forward, data gradient and weight gradient agree bit for bit"""
    from makani_amd import ops
    monkeypatch.setattr(ops, "GEMM_MODE", engine)
    torch.manual_seed(L * 7 + cin)
    w = torch.randn(1, cin, cout, L, dtype=torch.complex64, device=_dev())
    wn = ops.native_w_empty(cin, cout, L, _dev()).copy_(w)
    assert ops.is_native_w(wn) and not ops.is_native_w(w) and wn.shape == w.shape and torch.equal(wn, w)
    tri = torch.tril(torch.ones(L, M, device=_dev()))
    x = torch.randn(B, cin, L, M, dtype=torch.complex64, device=_dev()) * tri
    gy = torch.randn(B, cout, L, M, dtype=torch.complex64, device=_dev()) * tri
    res = []
    for wt in (w, wn):
        wt = wt.detach().requires_grad_(True)
        xs = x.clone().requires_grad_(True)
        T = ops.DhconvFn.apply(ops.ComplexToSFn.apply(xs), wt, B)
        y = ops.SToComplexFn.apply(T, B, cout)
        torch.view_as_real(y).mul(torch.view_as_real(gy)).sum().backward()
        res.append((y.detach(), xs.grad, wt.grad))
    (y0, gx0, gw0), (y1, gx1, gw1) = res
    assert gw1.stride() == wn.stride() and ops.is_native_w(gw1)            # autograd keeps it without a copy
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1) and torch.equal(gw0, gw1)
    ref = torch.einsum("bilm,iol->bolm", x.to(torch.complex128), w[0].to(torch.complex128))
    assert rel_l2(y1, ref) < (1e-5 if engine == "x6" else 1e-4)


def test_fused_adamw_on_native_order_weight():
    """FusedAdamW on a dense, non-C-contiguous complex parameter (gradient and state in the same strides)"""
    from makani_amd import ops
    from makani_amd.optim import FusedAdamW
    torch.manual_seed(4)
    w = torch.randn(1, 8, 12, 5, dtype=torch.complex64, device=_dev())
    a = torch.nn.Parameter(ops.native_w_empty(8, 12, 5, _dev()).copy_(w))
    b = torch.nn.Parameter(w.clone())
    big = torch.nn.Parameter(ops.native_w_empty(256, 256, 9, _dev()).copy_(torch.randn(1, 256, 256, 9, dtype=torch.complex64)))
    bigr = torch.nn.Parameter(big.detach().contiguous().clone())
    oa = FusedAdamW([a, big], lr=1e-2, weight_decay=0.01)
    ob = torch.optim.AdamW([b, bigr], lr=1e-2, weight_decay=0.01)
    for it in range(3):
        torch.manual_seed(20 + it)
        g, gb = torch.randn_like(w), torch.randn(1, 256, 256, 9, dtype=torch.complex64, device=_dev())
        a.grad = ops.native_w_empty(8, 12, 5, _dev()).copy_(g)
        b.grad = g.clone()
        big.grad = ops.native_w_empty(256, 256, 9, _dev()).copy_(gb)
        bigr.grad = gb.clone()
        torch.nn.utils.clip_grad_norm_([b, bigr], 1.5)
        ob.step()
        oa.step(max_grad_norm=1.5)
    assert a.stride() == ops.native_w_empty(8, 12, 5).stride() and oa.state[a]["exp_avg"].stride() == a.stride()
    assert rel_l2(a, b) < 2e-6 and rel_l2(big, bigr) < 2e-6


@pytest.mark.parametrize("B,C,L,M", [(1, 3, 5, 6), (2, 33, 40, 35), (2, 8, 12, 13)])
def test_s_layout_roundtrip(B, C, L, M):
    from makani_amd import ops
    c = torch.randn(B, C, L, M, dtype=torch.complex64)
    S = ops.complex_to_s(c.to(_dev()))
    Cp = ops.round4(C)
    assert S.shape == (L, M, 2, B * Cp)
    Sc = S.cpu().view(L, M, 2, B, Cp)
    assert torch.equal(Sc[:, :, 0, :, :C].permute(2, 3, 0, 1), c.real)
    assert (Sc[..., C:] == 0).all()
    back = ops.s_to_complex(S, B, C).cpu()
    tri = torch.tril(torch.ones(L, M, dtype=torch.bool))     # l >= m kept, rest exact zeros
    assert torch.equal(back, torch.where(tri, c, torch.zeros_like(c)))


# --------------------------------------------------------------------------- #
# pointwise
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("fuse_gelu", [False, True])
@pytest.mark.parametrize("B,C,H,W", [(2, 5, 12, 24), (1, 3, 37, 72), (1, 4, 7, 9), (1, 2, 240, 480)])
def test_instance_norm(dtype, tol, fuse_gelu, B, C, H, W):
    from makani_amd import ops
    torch.manual_seed(B * C + H)
    x = (torch.randn(B, C, H, W) * 2 + 3).to(dtype)
    gamma = torch.randn(C) + 1
    beta = torch.randn(C)
    gy = torch.randn(B, C, H, W).to(dtype)
    xd = x.to(_dev()).requires_grad_(True)
    gd, bd = gamma.to(_dev()).requires_grad_(True), beta.to(_dev()).requires_grad_(True)
    y = ops.InstanceNormFn.apply(xd, gd, bd, 1e-6, fuse_gelu)
    y.backward(gy.to(_dev()))
    torch.cuda.synchronize()
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.instance_norm(xr, weight=gr, bias=br, eps=1e-6)
    if fuse_gelu:
        yr = torch.nn.functional.gelu(yr)
    yr.backward(gy.double())
    assert y.dtype == dtype
    assert rel_l2(y, yr) < tol
    assert rel_l2(xd.grad, xr.grad) < (tol * 5 if dtype == torch.float32 else 2e-2)
    assert rel_l2(gd.grad, gr.grad) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel_l2(bd.grad, br.grad) < (1e-5 if dtype == torch.float32 else 2e-2)


def _norm_fwd_bwd(x, gy, gamma, beta, fuse_gelu, pre_bias=None):
    from makani_amd import ops
    xd = x.clone().requires_grad_(True)
    gd, bd = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = ops.InstanceNormFn.apply(xd, gd, bd, 1e-6, fuse_gelu, pre_bias)
    y.backward(gy)
    return y.detach(), xd.grad, gd.grad, bd.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fuse_gelu", [False, True])
@pytest.mark.parametrize("B,C,H,W,with_pb", [(2, 5, 12, 24, False), (1, 40, 240, 480, True), (1, 3, 721, 1440, False), (2, 7, 181, 1440, True)])
def test_instance_norm_one_pass_matches_two_kernel_path(monkeypatch, dtype, fuse_gelu, B, C, H, W, with_pb):
    """round 6: the one-pass kernels (a block keeps its chunk of the plane in registers, partial sums handed over through
    64-bit agent-scope atomics) against the two-kernel path on the same inputs: same statistics arithmetic, so bit-identical
    where the chunking is the same (small planes) and to fp32 round-off of the partial sums otherwise; and against fp64"""
    from makani_amd import ops
    from makani_amd._lib import lib, dtype_code
    torch.manual_seed(B * C + H)
    x = (torch.randn(B, C, H, W, device=_dev()) * 2 + 3).to(dtype)
    gy = torch.randn(B, C, H, W, device=_dev()).to(dtype)
    gamma, beta = torch.randn(C, device=_dev()) + 1, torch.randn(C, device=_dev())
    pb = torch.randn(C, device=_dev()) * 0.3 if with_pb else None
    assert lib().mk_instnorm_fused_chunks(H * W, dtype_code(x), B * C, 0) > 0
    monkeypatch.setenv("MAKANI_AMD_NORM_FUSED", "1")
    one = _norm_fwd_bwd(x, gy, gamma, beta, fuse_gelu, pb)
    monkeypatch.setenv("MAKANI_AMD_NORM_FUSED", "0")
    two = _norm_fwd_bwd(x, gy, gamma, beta, fuse_gelu, pb)
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float32 else 4e-3          # bf16: one rounding step of a few elements when a statistic moves by an ulp
    for a, b, name in zip(one, two, ("y", "gx", "dgamma", "dbeta")):
        assert rel_l2(a, b) < (tol if name in ("y", "gx") else 1e-5), (name, rel_l2(a, b))
    xr = (x.double() + (pb.double().view(1, -1, 1, 1) if with_pb else 0)).to(dtype).double().cpu().requires_grad_(True)
    gr, br = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    yr = torch.nn.functional.instance_norm(xr, weight=gr, bias=br, eps=1e-6)
    if fuse_gelu:
        yr = torch.nn.functional.gelu(yr)
    yr.backward(gy.double().cpu())
    t = 2e-6 if dtype == torch.float32 else 6e-3
    assert rel_l2(one[0], yr) < t
    assert rel_l2(one[1], xr.grad) < (t * 5 if dtype == torch.float32 else 2e-2)
    assert rel_l2(one[2], gr.grad) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel_l2(one[3], br.grad) < (1e-5 if dtype == torch.float32 else 2e-2)


def test_instance_norm_one_pass_falls_back_on_unserved_planes():
    from makani_amd._lib import lib, MK_BF16, MK_F32
    assert lib().mk_instnorm_fused_chunks(7 * 9, MK_BF16, 4, 0) == 0          # not a multiple of the 8-element vector
    assert lib().mk_instnorm_fused_chunks(7 * 9 * 4, MK_F32, 4, 0) > 0
    assert lib().mk_instnorm_fused_chunks(721 * 1440, MK_BF16, 384, 0) == 64   # 8 slots: 16 384 elements per block
    assert lib().mk_instnorm_fused_chunks(721 * 1440, MK_BF16, 384, 2) == 102  # 5 slots (backward through the GELU)
    assert lib().mk_instnorm_fused_chunks(240 * 480, MK_BF16, 384, 1) == 8
    assert lib().mk_instnorm_fused_chunks(12 * 24, MK_BF16, 10, 0) == 1        # small planes: the two-kernel chunking fits


def test_instance_norm_one_pass_handover_survives_repetition_and_graph_replay():
    """the hand-over buffers are re-armed by the kernels themselves: 200 back-to-back launches (forward + backward, two plane
    shapes interleaved on one buffer pair) stay bit-identical, and so do replays of a captured forward + backward"""
    torch.manual_seed(5)
    shapes = [(1, 96, 60, 480), (1, 6, 721, 1440)]
    data = []
    for (B, C, H, W) in shapes:
        x = torch.randn(B, C, H, W, device=_dev()).bfloat16()
        gy = torch.randn(B, C, H, W, device=_dev()).bfloat16()
        data.append((x, gy, torch.rand(C, device=_dev()) + 0.5, torch.randn(C, device=_dev())))
    refs = [[t.clone() for t in _norm_fwd_bwd(*d, True)] for d in data]
    for it in range(100):
        for d, ref in zip(data, refs):
            out = _norm_fwd_bwd(*d, True)
            assert all(torch.equal(a.float(), b.float()) for a, b in zip(out, ref)), it
    # graph: static tensors, capture one forward + backward of the first shape, replay on changing data
    from makani_amd import ops
    x, gy, gam, bet = data[0]
    xs, gs = x.clone().requires_grad_(True), gy.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.InstanceNormFn.apply(xs, gam, bet, 1e-6, True).backward(gs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    xs.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ys = ops.InstanceNormFn.apply(xs, gam, bet, 1e-6, True)
        ys.backward(gs)
    for k in range(4):
        with torch.no_grad():
            xs.copy_(x * (1.0 + 0.25 * k))
        graph.replay()
        torch.cuda.synchronize()
        want = _norm_fwd_bwd((x * (1.0 + 0.25 * k)).bfloat16(), gy, gam, bet, True)
        assert torch.equal(ys.detach().float(), want[0].float()) and torch.equal(xs.grad.float(), want[1].float()), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 5, 12, 24), (1, 3, 37, 71), (1, 768, 240, 480), (3, 700, 9, 16),
                                     # block 7's MLP hidden gradient at the benchmark's size: 768 planes of
                                     # 1 038 240 points, non-zero mean, against the fp64 sum
                                     (1, 768, 721, 1440)])
def test_plane_sums_bias_gradient(dtype, B, C, H, W):
    from makani_amd import ops
    torch.manual_seed(B * C + H)
    x = (torch.randn(B, C, H, W) + 0.25).to(dtype)
    got = ops._sum_planes(x.to(_dev()))
    ref = x.double().sum(dim=(0, 2, 3))
    assert got.shape == (C,) and got.dtype == torch.float32
    assert float((got.cpu().double() - ref).abs().max()) < 2e-6 * float(x.double().abs().sum(dim=(0, 2, 3)).max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_bias_gelu(dtype, tol, with_bias):
    from makani_amd import ops
    B, C, H, W = 2, 6, 9, 20
    x = torch.randn(B, C, H, W).to(dtype)
    b = torch.randn(C) if with_bias else None
    gy = torch.randn(B, C, H, W).to(dtype)
    xd = x.to(_dev()).requires_grad_(True)
    bd = b.to(_dev()).requires_grad_(True) if with_bias else None
    y = ops.BiasGeluFn.apply(xd, bd)
    y.backward(gy.to(_dev()))
    xr = x.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if with_bias else None
    yr = torch.nn.functional.gelu(xr + (br.view(1, -1, 1, 1) if with_bias else 0))
    yr.backward(gy.double())
    assert rel_l2(y, yr) < tol
    assert rel_l2(xd.grad, xr.grad) < (1e-5 if dtype == torch.float32 else 2e-2)
    if with_bias:
        assert rel_l2(bd.grad, br.grad) < (1e-5 if dtype == torch.float32 else 2e-2)


def test_fused_adamw_matches_torch():
    from makani_amd.optim import FusedAdamW
    torch.manual_seed(3)
    # one complex tensor, 60 small tensors (two multi-tensor launches, odd sizes and offsets) and one above the
    # multi-tensor threshold (single-tensor kernel)
    shapes = [(1, 6, 5, 9), (33,), (7, 13, 1, 1)] + [(3 + 5 * i,) for i in range(58)] + [(1100, 1024)]
    def make():
        torch.manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(*shapes[0], dtype=torch.complex64, device=_dev()))]
        ps += [torch.nn.Parameter(torch.randn(*s_, device=_dev())) for s_ in shapes[1:]]
        return ps
    a, b = make(), make()
    oa = FusedAdamW(a, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01)
    ob = torch.optim.AdamW(b, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, foreach=True)
    for it in range(4):
        torch.manual_seed(10 + it)
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa) * 3
            pa.grad, pb.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(b, 2.0, foreach=True)
        ob.step()
        oa.step(max_grad_norm=2.0)
    for pa, pb in zip(a, b):
        assert rel_l2(pa, pb) < 2e-6
    assert set(oa.state[a[0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert a[1]._version > 0                       # raw-pointer updates are visible to autograd's version counter


# --------------------------------------------------------------------------- #
# bf16 channel GEMMs
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("B,M,K,H,W", [(1, 64, 32, 8, 32), (2, 73, 40, 12, 24), (1, 40, 73, 37, 72), (1, 384, 768, 16, 40),
                                     (2, 130, 260, 10, 52),
                                     # K = 384: the weight-stationary kernel (one slab, two slabs + ragged pixel tail + batch,
                                     # a slab with rows past M, more pixel tiles than the DMA look-ahead)
                                     (1, 384, 384, 16, 40), (2, 768, 384, 10, 52), (1, 300, 384, 37, 72), (1, 768, 384, 91, 184),
                                     # FourCastNet3's channel counts: ring kernel with a ragged last k-tile (input channels past K
                                     # read zeros), ragged channel slabs, one k-tile plus a remainder
                                     (1, 677, 1354, 16, 40), (1, 641, 677, 20, 36), (2, 200, 100, 10, 52), (1, 1354, 641, 8, 40),
                                     # ... and enough pixels for the weight-gradient ring kernel with both channel counts above 384
                                     # (two-dimensional tiling: slabs of P x 384-row slabs of Q, either operand as Q, batch)
                                     (1, 677, 1354, 48, 48), (1, 1354, 641, 48, 48), (2, 450, 400, 32, 72),
                                     # FourCastNet3's local-block channel mix itself: 677 <- 6093 = 677 x 9 (96 k-tiles, the last
                                     # one ragged: 13 of 64 input channels), forward / fused epilogues / weight gradient
                                     (1, 677, 6093, 24, 48), (1, 6093, 677, 16, 24),
                                     # the 73-channel edges on the weight-stationary kernel (round 4: K padded to lda = 80, one 96-row
                                     # chunk per tile whose rows >= K arrive as zeros): ragged last pixel tile, batch, slab rows past M,
                                     # two slabs with K = 80 exactly, fewer tiles than workgroups
                                     (1, 384, 73, 37, 72), (2, 384, 73, 16, 40), (1, 300, 73, 91, 184), (1, 768, 80, 12, 24),
                                     (1, 384, 73, 8, 16)])
def test_conv1x1_nn_and_wgrad(B, M, K, H, W):
    from makani_amd import ops
    torch.manual_seed(M + K)
    x = torch.randn(B, K, H, W).bfloat16()
    w = (torch.randn(M, K) / math.sqrt(K)).bfloat16()
    bias = torch.randn(M)
    res = torch.randn(B, M, H, W).bfloat16()
    ref = torch.einsum("mk,bkhw->bmhw", w.double(), x.double())
    A = ops.pad_weight_bf16(w.to(_dev()))
    y, _ = ops.conv1x1_nn(A, K, x.to(_dev()))
    assert rel_l2(y, ref) < 4e-3                                    # bf16 output rounding only
    y, pre = ops.conv1x1_nn(A, K, x.to(_dev()), bias=bias.to(_dev()), act=True, want_pre=True, residual=res.to(_dev()))
    pre_ref = ref + bias.double().view(1, -1, 1, 1)
    assert rel_l2(pre, pre_ref) < 4e-3
    assert rel_l2(y, torch.nn.functional.gelu(pre_ref) + res.double()) < 5e-3
    y, pre = ops.conv1x1_nn(A, K, x.to(_dev()), bias=bias.to(_dev()), act=True, want_pre=True)      # the MLP's / encoder's first layer
    assert rel_l2(pre, pre_ref) < 4e-3 and rel_l2(y, torch.nn.functional.gelu(pre_ref)) < 5e-3
    y, _ = ops.conv1x1_nn(A, K, x.to(_dev()), residual=res.to(_dev()))                               # the skip connections
    assert rel_l2(y, ref + res.double()) < 5e-3
    gsrc = torch.randn(B, M, H, W).bfloat16()
    y, _ = ops.conv1x1_nn(A, K, x.to(_dev()), gelu_grad_of=gsrc.to(_dev()))
    gd = gsrc.double().requires_grad_(True)
    torch.nn.functional.gelu(gd).sum().backward()
    assert rel_l2(y, ref * gd.grad) < 5e-3
    # weight gradient: dW[m][k] = sum g[b][m][n] x[b][k][n]
    g = torch.randn(B, M, H, W).bfloat16()
    dW = ops.conv1x1_wgrad(g.to(_dev()), x.to(_dev()))
    dref = torch.einsum("bmhw,bkhw->mk", g.double(), x.double())
    assert dW.dtype == torch.float32 and rel_l2(dW, dref) < 1e-5
    # ... with the bias gradient from the same pass over g (round 4: row sums inside the ring kernel; plane sums elsewhere)
    gm = (g.float() + 0.25).bfloat16()                               # non-zero mean: the sums do not cancel
    dW2, db = ops.conv1x1_wgrad(gm.to(_dev()), x.to(_dev()), want_bias=True)
    assert db.shape == (M,) and db.dtype == torch.float32
    bref = gm.double().sum(dim=(0, 2, 3))
    assert float((db.cpu().double() - bref).abs().max()) < 2e-6 * float(gm.double().abs().sum(dim=(0, 2, 3)).max())
    assert rel_l2(dW2, torch.einsum("bmhw,bkhw->mk", gm.double(), x.double())) < 1e-5


def _bf16_ulps(got, want):
    """distance in bf16 steps between two bf16 tensors of the same sign pattern (bit patterns are monotonic in magnitude)"""
    a = got.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0x7FFF
    b = want.detach().cpu().contiguous().view(torch.int16).to(torch.int32) & 0x7FFF
    same_sign = (got.cpu().float() >= 0) == (want.cpu().float() >= 0)
    return torch.where(same_sign | ((a == 0) & (b == 0)), (a - b).abs(), a + b)


def _gelu_erfc(x):
    """exact GELU in fp64 WITHOUT the cancellation of x (1 + erf(x / sqrt 2)) / 2, which has no digit left below x = -8 even in fp64"""
    x = x.double()
    return 0.5 * x * torch.special.erfc(-x / math.sqrt(2.0))


def _worst(ulps, live, arg, got, want, n=6):
    u = torch.where(live, ulps, torch.zeros_like(ulps)).flatten()
    idx = torch.argsort(u, descending=True)[:n]
    return [(float(arg.flatten()[i]), float(got.cpu().float().flatten()[i]), float(want.float().flatten()[i]), int(u[i])) for i in idx]


@pytest.mark.parametrize("where", ["gemm_epilogue_768", "gemm_epilogue_73", "instance_norm"])
def test_bf16_forward_gelu_is_exact_to_the_rounding_including_the_tails(where):
    """The forward GELU of the bf16 kernels (csrc/common.h gelu_exp2_x2: x Phi(x) = relu(x) - |x| Phi(-|x|), Phi(-a) = 2^-(1 + a S(a)))
    element by element against the exact-erf GELU (nn.GELU(approximate='none'), makani/models/common/layers.py:768) of the SAME
    bf16 argument, evaluated in fp64 and rounded to bf16: at most ONE bf16 step anywhere — including x < -4 where x Phi(x) is
    1e-4 ... 1e-17 and a formula of the form x (1 + erf) / 2 has no correct digit left — and the identical value in >= 99.9 %."""
    from makani_amd import ops
    torch.manual_seed(5)
    if where == "instance_norm":
        B, Cc, H, W = 1, 8, 64, 128
        x = (torch.randn(B, Cc, H, W) * torch.tensor([0.3, 1, 1, 2, 3, 4, 5, 6.0]).view(1, Cc, 1, 1)).bfloat16()
        gamma, beta = torch.tensor([0.3, 1, 1, 2, 3, 4, 5, 6.0]), torch.linspace(-1, 1, Cc)
        y = ops.InstanceNormFn.apply(x.to(_dev()), gamma.to(_dev()), beta.to(_dev()), 1e-6, True)
        xd = x.double()
        mean, var = xd.mean(dim=(2, 3), keepdim=True), xd.var(dim=(2, 3), keepdim=True, unbiased=False)
        arg = (xd - mean) / torch.sqrt(var + 1e-6) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
        want = _gelu_erfc(arg)
        # the argument itself is formed in fp32 inside the kernel (not rounded to bf16): compare where fp32 vs fp64 of the ARGUMENT
        # cannot move the result by a bf16 step, i.e. everywhere but a sliver around rounding boundaries -> allow 1 step, count equal
        ulps = _bf16_ulps(y, want.bfloat16())
        live = want.abs() >= 1e-17                                              # beyond |x| = 9: Phi(-9) stands in (values ~ 1e-18)
        assert float(arg.min()) < -8 and int(ulps[live].max()) <= 1, _worst(ulps, live, arg, y, want.bfloat16())
        assert float((ulps[live] == 0).double().mean()) > 0.995
        return
    # pre-activations that are EXACT in bf16 and in the fp32 accumulators (one power-of-two weight per row, integer inputs, bias in
    # quarters), so that every kernel form — GELU of the fp32 accumulator or of its bf16 rounding — evaluates the same argument
    M, K = (768, 384) if where == "gemm_epilogue_768" else (384, 73)
    H, W = 24, 64
    x = torch.randint(-16, 17, (1, K, H, W)).bfloat16()
    w = torch.zeros(M, K)
    w[torch.arange(M), torch.arange(M) % K] = 2.0 ** -(torch.arange(M) % 4).float()
    bias = (torch.arange(M) % 7 - 3).float() * 0.25
    A = ops.pad_weight_bf16(w.bfloat16().to(_dev()))
    y, pre = ops.conv1x1_nn(A, K, x.to(_dev()), bias=bias.to(_dev()), act=True, want_pre=True)
    arg = torch.einsum("mk,bkhw->bmhw", w.double(), x.double()) + bias.double().view(1, -1, 1, 1)
    assert torch.equal(pre.cpu().double(), arg)
    want = _gelu_erfc(arg).bfloat16()
    assert float(arg.min()) < -8 and float(arg.max()) > 8
    ulps = _bf16_ulps(y, want)
    tiny = want.float().abs() < 1e-17                                        # beyond |x| = 9: Phi(-9) stands in (values ~ 1e-18)
    assert int(ulps[~tiny].max()) <= 1, _worst(ulps, ~tiny, arg, y, want)
    assert float((ulps[~tiny] == 0).double().mean()) > 0.999
    tail = (pre.cpu().float() < -4) & ~tiny
    assert int(tail.sum()) > 100 and int(ulps[tail].max()) <= 1


@pytest.mark.parametrize("form", ["0", "1"])
def test_weight_stationary_kernel_forms(form):
    """the K = 384 launches pick the one-group or the two-group weight-stationary kernel per epilogue variant (csrc/conv1x1.hip:
    MAKANI_AMD_ASTAT2 unset); the switch is read once per process, so both forms of EVERY variant are run here in child
    processes: "0" = one wave group everywhere, "1" = two wave groups everywhere (incl. bias + GELU + pre-activation)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x", "-k",
                          "test_conv1x1_nn_and_wgrad and 384 and not 73"], env=dict(os.environ, MAKANI_AMD_ASTAT2=form),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.parametrize("policy", ["0", "1"])
def test_channel_gemm_store_policies(policy):
    """the output stores of the channel GEMMs take the streaming (nt) cache policy for outputs of at most 256 MiB and the default one
    above (csrc/conv1x1.hip: mk_st16 / MK_BUF_ST16, MAKANI_AMD_CONV_NT unset): the test shapes are all small, so the default-policy
    branch of every kernel would never run here.  The switch is read once per process: both policies of every variant in child
    processes ("0" = never, "1" = always); a cache policy must not change a single value."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x", "-k",
                          "test_conv1x1_nn_and_wgrad or test_bf16_forward_gelu"], env=dict(os.environ, MAKANI_AMD_CONV_NT=policy),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@pytest.mark.parametrize("engine", ["x6", "x3"])
@pytest.mark.parametrize("native", [True, False])
def test_dhconv_high_degrees_all_row_tiles(engine, native, monkeypatch):
    """degrees 217 ... 240 of the 241-order contraction (tri_off: the degree of the first batch entry): all eight 32-row tiles
    of the complex split kernel's 256 x 128 form are live, the last one with 17 of its 32 rows; forward and data gradient, native
    (interleaved B operand) and planar weights, against a complex128 einsum"""
    from makani_amd import ops
    monkeypatch.setattr(ops, "GEMM_MODE", engine)
    torch.manual_seed(3)
    B, C, L, M, off = 1, 384, 24, 241, 217
    w = torch.randn(1, C, C, L, dtype=torch.complex64, device=_dev())
    W = ops.native_w_empty(C, C, L, _dev()).copy_(w) if native else ops.weight_to_wlayout(w)
    live = (torch.arange(L, device=_dev())[:, None] + off >= torch.arange(M, device=_dev())[None, :])[:, :, None]
    S = torch.randn(L, M, 2, C, device=_dev()) * live[..., None]           # the contraction's own layout [l][m][re / im][channel]
    x = torch.complex(S[:, :, 0].double(), S[:, :, 1].double())
    cplx = lambda T: torch.where(live, torch.complex(T[:, :, 0], T[:, :, 1]), torch.zeros((), dtype=torch.complex64, device=_dev()))
    y = cplx(ops.dhconv_fwd(S, W, B, C, tri_off=off))                      # (orders above the degree are not written)
    gx = cplx(ops.dhconv_dgrad(S, W, B, C, C, tri_off=off))
    ref = torch.einsum("lmi,iol->lmo", x, w[0].to(torch.complex128))
    refg = torch.einsum("lmo,iol->lmi", x, w[0].conj().to(torch.complex128))
    tol = 1e-5 if engine == "x6" else 1e-4
    assert rel_l2(y, ref) < tol and rel_l2(gx, refg) < tol


def test_complex_split_kernel_square_tile_form():
    """the dhconv forward / data gradient with more than 128 orders run on the 256 x 128 tile of the complex split kernel
    (csrc/xgemm2.hip, RT = 4); MAKANI_AMD_X2_TALL=0 (read once per process) keeps the 128 x 128 tile: the 241-order cases of the
    dhconv tests in a child process on that form"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x", "-k",
                          "(test_dhconv_native_weight_order_matches_planar and 241) or test_dhconv_high_degrees"],
                         env=dict(os.environ, MAKANI_AMD_X2_TALL="0"),
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


def test_conv_gelu_conv_autograd():
    from makani_amd import ops
    torch.manual_seed(4)
    B, K, Hd, M, H, W = 2, 24, 48, 20, 6, 16
    x = torch.randn(B, K, H, W)
    w1 = torch.randn(Hd, K, 1, 1) / math.sqrt(K)
    b1 = torch.randn(Hd) * 0.1
    w2 = torch.randn(M, Hd, 1, 1) / math.sqrt(Hd)
    b2 = torch.randn(M) * 0.1
    gy = torch.randn(B, M, H, W)
    dev_t = [t.to(_dev()).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = ops.ConvGeluConvFn.apply(dev_t[0].bfloat16(), *dev_t[1:])
    y.backward(gy.to(_dev()).bfloat16())
    ref_t = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    xr = ref_t[0].bfloat16().double()          # same rounded input
    h = torch.nn.functional.gelu(torch.nn.functional.conv2d(xr, ref_t[1], ref_t[2]))
    yr = torch.nn.functional.conv2d(h, ref_t[3], ref_t[4])
    yr.backward(gy.double())
    assert rel_l2(y, yr) < 1e-2
    for d, r in zip(dev_t, ref_t):
        assert rel_l2(d.grad, r.grad) < 2e-2, d.shape


def test_bf16_weight_shadow_is_exact_and_invalidated_by_inplace_updates():
    """FusedAdamW writes bf16(p) — and its transpose — for the channel-GEMM weights (zero-padded to multiples of 8
    columns); the GEMMs use them only for that exact parameter version"""
    import makani_amd as ma
    from makani_amd import ops
    from makani_amd.optim import FusedAdamW
    torch.manual_seed(4)
    for cin, cout in ((16, 24), (73, 20), (12, 73)):
        conv = ma.PointwiseConv(cin, cout, bias=False).to(_dev())
        opt = FusedAdamW(conv.parameters(), lr=1e-2, weight_decay=0.0)
        x = torch.randn(1, cin, 8, 16, device=_dev())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            conv(x).float().square().mean().backward()
        assert getattr(conv.weight, "_mk_shadow", None) is None
        opt.step()
        wb = conv.weight.detach().view(cout, cin).to(torch.bfloat16)
        sh, sht = conv.weight._mk_shadow, conv.weight._mk_shadow_t
        assert sh.dtype == torch.bfloat16 and sh.shape == (cout, (cin + 7) // 8 * 8) and sht.shape == (cin, (cout + 7) // 8 * 8)
        assert torch.equal(sh[:, :cin], wb) and torch.equal(sht[:, :cout], wb.t())
        assert (sh[:, cin:] == 0).all() and (sht[:, cout:] == 0).all()
        A, At = ops.weight_operands(conv.weight, need_t=True)
        assert A.data_ptr() == sh.data_ptr() and At.data_ptr() == sht.data_ptr()
        assert ops.cast_weight(conv.weight, torch.bfloat16).data_ptr() == sh.data_ptr()
        with torch.no_grad():
            conv.weight.mul_(2.0)                                   # any in-place change through torch
        w = ops.cast_weight(conv.weight, torch.bfloat16)
        assert w.data_ptr() != sh.data_ptr() and torch.equal(w, conv.weight.detach().to(torch.bfloat16))
        A, At = ops.weight_operands(conv.weight, need_t=True)
        assert A.data_ptr() != sh.data_ptr() and torch.equal(At[:, :cout], conv.weight.detach().view(cout, cin).to(torch.bfloat16).t())
        xr = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(xr)
        y.float().sum().backward()
        wf = conv.weight.detach().view(cout, cin).to(torch.bfloat16).float()
        ref = torch.einsum("oi,bihw->bohw", wf, x.to(torch.bfloat16).float())
        assert rel_l2(y, ref) < 1e-2
        assert rel_l2(xr.grad, wf.sum(0).view(1, -1, 1, 1).expand_as(x)) < 1e-2
        # a write THROUGH param.data is invisible to the version counter: the images stay "valid" until they are dropped by hand
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            conv(x).float().square().mean().backward()
        opt.step()
        sh2 = conv.weight._mk_shadow
        conv.weight.data.mul_(0.5)
        assert ops.weight_operands(conv.weight)[0].data_ptr() == sh2.data_ptr()            # stale, undetected
        from makani_amd.optim import invalidate_weight_shadows
        invalidate_weight_shadows(conv)
        A, _ = ops.weight_operands(conv.weight)
        assert A.data_ptr() != sh2.data_ptr() and torch.equal(A[:, :cin], conv.weight.detach().view(cout, cin).to(torch.bfloat16))


def test_fused_adamw_device_step_counter_matches_host_steps():
    """the step number lives in device memory (mk_adamw_advance): five steps give torch.optim.AdamW's result, i.e. the
    bias corrections follow the counter, not a launch argument"""
    from makani_amd.optim import FusedAdamW
    torch.manual_seed(1)
    a = [torch.nn.Parameter(torch.randn(300, 40, device=_dev())), torch.nn.Parameter(torch.randn(1200, 1024, device=_dev()))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = FusedAdamW(a, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.0)
    ob = torch.optim.AdamW(b, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.0)
    for it in range(5):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    st = oa.param_groups[0]["_mk_step_state"].cpu()
    assert st[0].item() == 5.0 and abs(st[1].item() - 1.0 / (1 - 0.9 ** 5)) < 1e-5
    for pa, pb in zip(a, b):
        assert rel_l2(pa, pb) < 2e-6


@pytest.mark.parametrize("B,Cc,H,W", [(1, 384, 240, 480), (2, 45, 33, 64), (2, 7, 5, 9), (1, 384, 31, 45)])
@pytest.mark.parametrize("mode", ["fp32", "bf16_autocast", "bf16"])
def test_channel_layernorm_matches_torch_layernorm(B, Cc, H, W, mode):
    """csrc/chan_layernorm.hip (no transposes, no library layer norm) against nn.LayerNorm over the channels in fp64 — the
    reference module (makani/mpu/layer_norm.py:256-290): output, input gradient, dgamma, dbeta; vector (P % 4 == 0) and scalar
    pixel paths; fp32, bf16 under autocast (fp32 output, as torch's layer_norm autocast policy) and plain bf16"""
    from makani_amd.layers import ChannelLayerNorm
    DEV = _dev()
    torch.manual_seed(B * 100 + Cc)
    m = ChannelLayerNorm(Cc, eps=1e-6).to(DEV)
    with torch.no_grad():
        m.norm.weight.normal_(1.0, 0.3)
        m.norm.bias.normal_(0.0, 0.3)
    x = torch.randn(B, Cc, H, W, device=DEV) * 2.0 + 5.0            # a large common mean: the shifted moments matter
    g = torch.randn(B, Cc, H, W, device=DEV)
    xin = (x if mode == "fp32" else x.bfloat16()).detach().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16_autocast"):
        y = m(xin)
    assert y.dtype == (torch.bfloat16 if mode == "bf16" else torch.float32)
    (y.float() * g).sum().backward()
    ref = torch.nn.LayerNorm(Cc, eps=1e-6).to(DEV).double()
    ref.load_state_dict({k: v.double() for k, v in m.norm.state_dict().items()})
    xr = xin.detach().double().requires_grad_(True)
    yr = torch.transpose(ref(torch.transpose(xr, 1, 3)), 1, 3)
    (yr * g.double()).sum().backward()
    tol = 1e-5 if mode == "fp32" else (2e-5 if mode == "bf16_autocast" else 6e-3)
    assert rel_l2(y, yr) < tol
    assert rel_l2(xin.grad, xr.grad) < (tol if mode == "fp32" else 6e-3)
    assert rel_l2(m.norm.weight.grad, ref.weight.grad) < max(tol, 2e-5) and rel_l2(m.norm.bias.grad, ref.bias.grad) < max(tol, 2e-5)


@pytest.mark.parametrize("B,G,M,K,N", [(1, 1, 384, 384, 115200), (2, 1, 73, 384, 2664), (1, 1, 384, 73, 16380), (2, 3, 10, 27, 2048),
                                       (1, 1, 16, 5, 45), (2, 2, 9, 7, 333)])
def test_fp32_channel_gemm_engine_matches_fp64(B, G, M, K, N):
    """ops.chan_gemm_f32 / chan_wgrad_f32 / ConvMmFn / GroupMmFn: the fp32 1x1 convolutions and grouped channel mixes on the
    package's own GEMM engine (no library GEMM in the product): forward, accumulate-into, data gradient
    (transposed weight), weight gradient with the pixel sum split into slabs; channel counts that are not multiples of 4,
    pixel counts that are not multiples of 4 (zero-padded planes); fp32 op tolerance 1e-5 against fp64"""
    from makani_amd import ops
    DEV = _dev()
    torch.manual_seed(M + K)
    grouped = G > 1
    W = (torch.randn(G, M, K, device=DEV) / math.sqrt(K))
    x = torch.randn(B, G, K, N, device=DEV)
    gy = torch.randn(B, G, M, N, device=DEV)
    yr = torch.matmul(W.double().unsqueeze(0), x.double())
    w_, x_, g_ = (W, x, gy) if grouped else (W[0], x[:, 0], gy[:, 0])
    y = ops.chan_gemm_f32(w_, x_)
    assert rel_l2(y.reshape(yr.shape), yr) < 1e-5
    acc = gy.clone()
    a_ = acc if grouped else acc[:, 0].contiguous()
    ops.chan_gemm_f32(w_, x_, a_, accumulate=True)
    assert rel_l2(a_.reshape(yr.shape), yr + gy.double()) < 1e-5
    gx = ops.chan_gemm_f32(w_, g_, transposed=True)
    assert rel_l2(gx.reshape(x.shape), torch.matmul(W.double().transpose(1, 2).unsqueeze(0), gy.double())) < 1e-5
    dW = ops.chan_wgrad_f32(g_, x_)
    assert rel_l2(dW.reshape(W.shape), torch.einsum("bgmn,bgkn->gmk", gy.double(), x.double())) < 1e-5
    if grouped:
        xg, Wg = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        z = ops.GroupMmFn.apply(xg, Wg)
        (z * gy).sum().backward()
        assert rel_l2(z, yr) < 1e-5 and rel_l2(xg.grad, torch.matmul(W.double().transpose(1, 2).unsqueeze(0), gy.double())) < 1e-5
        assert rel_l2(Wg.grad, torch.einsum("bgmn,bgkn->gmk", gy.double(), x.double())) < 1e-5
    else:
        H = 9 if N % 9 == 0 else 1
        x4 = x[:, 0].reshape(B, K, H, N // H).clone().requires_grad_(True)
        w4 = W[0].reshape(M, K, 1, 1).clone().requires_grad_(True)
        r4 = gy[:, 0].reshape(B, M, H, N // H).clone().requires_grad_(True)
        y4 = ops.ConvMmFn.apply(x4, w4, r4, False)
        (y4 * gy[:, 0].reshape(y4.shape)).sum().backward()
        assert rel_l2(y4.reshape(B, 1, M, N), yr + gy.double()) < 1e-5 and torch.equal(r4.grad, gy[:, 0].reshape(r4.shape))
        assert rel_l2(x4.grad.reshape(B, 1, K, N), torch.matmul(W.double().transpose(1, 2).unsqueeze(0), gy.double())) < 1e-5
        assert rel_l2(w4.grad.reshape(1, M, K), torch.einsum("bgmn,bgkn->gmk", gy.double(), x.double())) < 1e-5


@pytest.mark.parametrize("nlat,nlon,mmax,Cc,m_shapes,r_shapes,xseg", [
    (181, 1440, 241, 192, [121, 120], [48, 48, 48, 48], 2),        # BASELINE configs[4] on one rank: h4 w2, 192 planes
    (45, 1440, 241, 37, [241], [12, 12, 12, 4], 1),                # h4 w1, ragged planes (37 -> 40 rows in sub-blocks of 4)
    (60, 480, 241, 96, [121, 120], [48, 48], 2),                   # internal grid, full spectrum
    (23, 720, 361, 20, [91, 90, 90, 90], [8, 12], 2),              # FourCastNet3's internal grid
    (9, 72, 13, 6, [7, 6], [4, 4], 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segmented_fft_kernels_match_plain_kernels(nlat, nlon, mmax, Cc, m_shapes, r_shapes, xseg, dtype):
    """mk_rfft_rows_seg / mk_irfft_rows_seg (csrc/fft_fast.hip, SEG kernels: rows read from / written to `xseg` longitude
    pieces, the F side addressed as per-peer slabs [lat][m][re/im][row]) against the plain kernels on the same data: bit-identical
    values, only the addresses differ; incl. a latitude sub-range call (chunked overlap)"""
    from makani_amd import ops
    DEV = _dev()
    torch.manual_seed(nlat + Cc)
    w = (2.0 * math.pi / nlon, 2.0 * math.pi / nlon, 2.0 * math.pi / nlon)
    x = torch.randn(1, Cc, nlat, nlon, device=DEV).to(dtype)
    Cp = ops.round4(Cc)
    F = ops.rfft_rows(x, mmax, Cp, w)                                                  # (M, nlat, 2, Cp)
    wl = nlon // xseg
    xbuf = x[0].reshape(Cc, nlat, xseg, wl).permute(2, 0, 1, 3).contiguous()          # pieces [j][plane][lat][wl]
    rows = sum(r_shapes)
    assert rows >= Cc and len(m_shapes) <= 8 and sum(m_shapes) == mmax
    moff, roff = [0], [0]
    for n in m_shapes:
        moff.append(moff[-1] + n)
    for n in r_shapes:
        roff.append(roff[-1] + n)

    def desc(a, b):
        base, off = [[0] * len(r_shapes) for _ in m_shapes], 0
        for i in range(len(r_shapes)):
            for j in range(len(m_shapes)):
                base[j][i] = off + a * m_shapes[j] * 2 * r_shapes[i]
                off += nlat * m_shapes[j] * 2 * r_shapes[i]
        return ops.fft_seg_desc(m_shapes, r_shapes, base, xseg=xseg, x_stride=Cc * nlat * wl, x_nlat=nlat), off

    def slabs(fs):
        """flat slab buffer -> (M, nlat, 2, rows)"""
        out, off = torch.zeros(mmax, nlat, 2, rows, device=DEV), 0
        for i in range(len(r_shapes)):
            for j in range(len(m_shapes)):
                n = nlat * m_shapes[j] * 2 * r_shapes[i]
                out[moff[j]:moff[j + 1], :, :, roff[i]:roff[i + 1]] = fs[off:off + n].reshape(nlat, m_shapes[j], 2, r_shapes[i]).permute(1, 0, 2, 3)
                off += n
        return out

    _, total = desc(0, nlat)
    fs = torch.full((total,), float("nan"), device=DEV)
    cuts = [0, nlat // 3, nlat]                                                       # two latitude chunks
    for a, b in zip(cuts[:-1], cuts[1:]):
        sg, _ = desc(a, b)
        ops.rfft_rows_seg(xbuf, a * wl, fs, Cc, b - a, nlon, mmax, w, sg)
    got = slabs(fs)
    assert torch.equal(got[..., :Cc], F[..., :Cc])
    assert not torch.isnan(got[..., :Cp]).any()                                       # pad rows of a started group of 4 are zeros
    # inverse: slabs -> longitude pieces
    xr = ops.irfft_rows(F, 1, Cc, nlon, dtype, w)
    fr = torch.zeros((total,), device=DEV)
    off = 0
    for i in range(len(r_shapes)):
        for j in range(len(m_shapes)):
            n = nlat * m_shapes[j] * 2 * r_shapes[i]
            src = torch.zeros(m_shapes[j], nlat, 2, r_shapes[i], device=DEV)
            hi = min(roff[i + 1], Cp)
            if hi > roff[i]:
                src[..., :hi - roff[i]] = F[moff[j]:moff[j + 1], :, :, roff[i]:hi]
            fr[off:off + n] = src.permute(1, 0, 2, 3).reshape(-1)
            off += n
    xo = torch.full((xseg, Cc, nlat, wl), float("nan"), device=DEV).to(dtype)
    for a, b in zip(cuts[:-1], cuts[1:]):
        sg, _ = desc(a, b)
        ops.irfft_rows_seg(fr, xo, a * wl, Cc, b - a, nlon, mmax, w, sg)
    assert torch.equal(xo.permute(1, 2, 0, 3).reshape(1, Cc, nlat, nlon), xr)
