"""Geometric L^p loss: oracle vs the reference's golden vectors (CPU), the module's
quadrature buffers vs the same vectors (CPU), HIP kernels vs oracle and golden vectors (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "geometric_lp_loss.npz")


def _cases():
    d = np.load(GOLDEN)
    return d, json.loads(str(d["cases"]))


def test_oracle_loss_matches_reference_golden():
    from oracle import losses as ol
    d, cases = _cases()
    for i, c in enumerate(cases):
        q = ol.quadrature_weights(ol.GRID_TO_RULE[c["grid"]], c["img"], c["crop"], c["off"], normalize=True)
        assert np.allclose(q.float().numpy(), d[f"{i}_q"][0, 0], rtol=0, atol=1e-9)
        prd = torch.tensor(d[f"{i}_prd"], requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"]) if c["wgt"] else None
        out = ol.geometric_lp_loss(prd, tar, q, c["p"], c["relative"], c["squared"], wgt=w)
        (out * torch.tensor(d[f"{i}_g"])).sum().backward()
        assert np.allclose(out.detach().numpy(), d[f"{i}_out"], rtol=1e-6, atol=1e-8)
        assert np.allclose(prd.grad.numpy(), d[f"{i}_dprd"], rtol=1e-5, atol=1e-9)
        assert np.allclose(tar.grad.numpy(), d[f"{i}_dtar"], rtol=1e-5, atol=1e-9)


def test_module_quadrature_buffers_match_reference():
    """host logic: the (non-persistent) quad_weight buffer equals the reference's for every rule, with crop"""
    import makani_amd as ma
    d, cases = _cases()
    for i, c in enumerate(cases):
        mod = ma.GeometricLpLoss(img_shape=c["img"], crop_shape=c["crop"], crop_offset=c["off"],
                                 channel_names=["a", "b", "c"], p=c["p"], relative=c["relative"], squared=c["squared"],
                                 grid_type=c["grid"])
        q = mod.quadrature.quad_weight
        assert q.shape == (1, 1, *c["crop"]) and q.dtype == torch.float32
        assert np.allclose(q.numpy(), d[f"{i}_q"], rtol=2e-7, atol=1e-12), c["grid"]
        assert len(mod.state_dict()) == 0 and mod.n_channels == 3
    with pytest.raises(NotImplementedError):
        ma.GeometricLpLoss((8, 16), (8, 16), (0, 0), ["a"], grid_type="healpix")
    with pytest.raises(RuntimeError):                   # no CPU implementation behind the module
        ma.GeometricLpLoss((8, 16), (8, 16), (0, 0), ["a"])(torch.zeros(1, 1, 8, 16), torch.zeros(1, 1, 8, 16))


@pytest.mark.gpu
def test_hip_loss_matches_reference_golden():
    import makani_amd as ma
    d, cases = _cases()
    dev = torch.device("cuda", 0)
    for i, c in enumerate(cases):
        mod = ma.GeometricLpLoss(img_shape=c["img"], crop_shape=c["crop"], crop_offset=c["off"],
                                 channel_names=["a", "b", "c"], p=c["p"], relative=c["relative"], squared=c["squared"],
                                 grid_type=c["grid"]).to(dev)
        prd = torch.tensor(d[f"{i}_prd"], device=dev, requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], device=dev, requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"], device=dev) if c["wgt"] else None
        out = mod(prd, tar, w)
        (out * torch.tensor(d[f"{i}_g"], device=dev)).sum().backward()
        assert np.allclose(out.detach().cpu().numpy(), d[f"{i}_out"], rtol=2e-5, atol=1e-7), c
        assert np.allclose(prd.grad.cpu().numpy(), d[f"{i}_dprd"], rtol=2e-4, atol=1e-8), c
        assert np.allclose(tar.grad.cpu().numpy(), d[f"{i}_dtar"], rtol=2e-4, atol=1e-8), c


@pytest.mark.gpu
@pytest.mark.parametrize("dt_prd,dt_tar", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                           (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("H,W,p,relative,squared", [(721, 1440, 2.0, False, True), (37, 73, 1.5, True, False),
                                                    (24, 48, 1.0, False, False)])
def test_hip_loss_vs_oracle(dt_prd, dt_tar, H, W, p, relative, squared):
    """seeded inputs incl. the BASELINE grid, mixed dtypes, odd plane sizes (scalar tail path)"""
    import makani_amd as ma
    from oracle import losses as ol
    torch.manual_seed(H + int(10 * p))
    dev = torch.device("cuda", 0)
    B, C = 1, 3
    prd = torch.randn(B, C, H, W).to(dt_prd)
    tar = torch.randn(B, C, H, W).to(dt_tar)
    g = torch.randn(B, C)
    mod = ma.GeometricLpLoss((H, W), (H, W), (0, 0), ["a"] * C, p=p, relative=relative, squared=squared).to(dev)
    pd, td = prd.to(dev).requires_grad_(True), tar.to(dev).requires_grad_(True)
    out = mod(pd, td)
    (out * g.to(dev)).sum().backward()
    pr, tr = prd.double().requires_grad_(True), tar.double().requires_grad_(True)
    q = ol.quadrature_weights("naive", (H, W), normalize=True).double()
    ref = ol.geometric_lp_loss(pr, tr, q, p, relative, squared)
    (ref * g.double()).sum().backward()
    assert out.dtype == torch.float32 and pd.grad.dtype == dt_prd and td.grad.dtype == dt_tar
    assert torch.allclose(out.cpu().double(), ref.detach(), rtol=2e-5, atol=1e-8)
    for got, want, dt in ((pd.grad, pr.grad, dt_prd), (td.grad, tr.grad, dt_tar)):
        tol = 1e-4 if dt == torch.float32 else 1e-2
        err = (got.cpu().double() - want).norm() / want.norm()
        assert err < tol, err


@pytest.mark.gpu
def test_hip_grid_quadrature_is_differentiable_sum():
    import makani_amd as ma
    dev = torch.device("cuda", 0)
    quad = ma.GridQuadrature("legendre-gauss", (24, 48), normalize=False).to(dev)
    x = torch.randn(2, 5, 24, 48, device=dev, requires_grad=True)
    y = quad(x)
    ref = torch.sum(x.detach() * quad.quad_weight, dim=(-2, -1))
    assert y.shape == (2, 5) and torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    y.sum().backward()
    assert torch.allclose(x.grad, quad.quad_weight.expand_as(x), rtol=1e-6, atol=0)
    one = quad(torch.ones(1, 1, 24, 48, device=dev))
    assert abs(one.item() - 4 * np.pi) < 1e-4          # tests/test_grids.py:136-220: the rule integrates 1 to 4 pi


# --------------------------------------------------------------------------- #
# SpectralLpLoss
# --------------------------------------------------------------------------- #
SGOLDEN = os.path.join(os.path.dirname(__file__), "golden", "spectral_lp_loss.npz")


def _scases():
    d = np.load(SGOLDEN)
    return d, json.loads(str(d["cases"]))


def test_oracle_spectral_loss_matches_reference_golden():
    from oracle import losses as ol
    d, cases = _scases()
    for i, c in enumerate(cases):
        prd = torch.tensor(d[f"{i}_prd"], requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"]) if c["wgt"] else None
        out = ol.spectral_lp_loss(prd, tar, c["img"], c["grid"], c["p"], c["relative"], c["squared"], wgt=w)
        (out * torch.tensor(d[f"{i}_g"])).sum().backward()
        assert np.allclose(out.detach().numpy(), d[f"{i}_out"], rtol=1e-6, atol=1e-9)
        assert np.allclose(prd.grad.numpy(), d[f"{i}_dprd"], rtol=1e-5, atol=1e-10)
        assert np.allclose(tar.grad.numpy(), d[f"{i}_dtar"], rtol=1e-5, atol=1e-10)


def test_spectral_loss_module_attributes_match_reference():
    import makani_amd as ma
    d, cases = _scases()
    for i, c in enumerate(cases):
        mod = ma.SpectralLpLoss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0), channel_names=["a", "b", "c"],
                                grid_type=c["grid"], p=c["p"], relative=c["relative"], squared=c["squared"])
        assert mod.lm_weights.shape == d[f"{i}_lm"].shape and np.allclose(mod.lm_weights.numpy(), d[f"{i}_lm"], rtol=1e-7)
        assert (mod.sht.lmax, mod.sht.mmax) == d[f"{i}_lm"].shape and len(mod.state_dict()) == 0
    with pytest.raises(NotImplementedError):
        ma.SpectralLpLoss((8, 16), (8, 16), (0, 0), ["a"], grid_type="healpix")


@pytest.mark.gpu
def test_hip_spectral_loss_matches_reference_golden():
    import makani_amd as ma
    d, cases = _scases()
    dev = torch.device("cuda", 0)
    for i, c in enumerate(cases):
        mod = ma.SpectralLpLoss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0), channel_names=["a", "b", "c"],
                                grid_type=c["grid"], p=c["p"], relative=c["relative"], squared=c["squared"]).to(dev)
        prd = torch.tensor(d[f"{i}_prd"], device=dev, requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], device=dev, requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"], device=dev) if c["wgt"] else None
        out = mod(prd, tar, w)
        (out * torch.tensor(d[f"{i}_g"], device=dev)).sum().backward()
        assert np.allclose(out.detach().cpu().numpy(), d[f"{i}_out"], rtol=3e-5, atol=1e-7), c
        for got, want in ((prd.grad, d[f"{i}_dprd"]), (tar.grad, d[f"{i}_dtar"])):
            err = np.linalg.norm(got.cpu().numpy() - want) / np.linalg.norm(want)
            assert err < 2e-5, (c, err)


@pytest.mark.gpu
def test_hip_spectral_loss_parseval_fullsize():
    """size-independent property at the BASELINE grid (tests/test_losses.py:440-452 of the reference): for a
    band-limited field the spectral L2 norm equals the quadrature L2 norm (Clenshaw-Curtis exact for the integrand)"""
    import makani_amd as ma
    dev = torch.device("cuda", 0)
    H, W, C = 721, 1440, 2
    torch.manual_seed(11)
    lmax = 120
    isht = ma.InverseRealSHT(H, W, lmax=lmax, mmax=lmax, grid="equiangular").to(dev)
    coef = torch.tril(torch.randn(1, C, lmax, lmax, dtype=torch.complex64, device=dev))
    x = isht(coef)
    spec = ma.SpectralLpLoss((H, W), (H, W), (0, 0), ["a"] * C, grid_type="equiangular", p=2.0, squared=True).to(dev)
    geo = ma.GridQuadrature("clenshaw-curtiss", (H, W), normalize=True).to(dev)
    a = spec(x, torch.zeros_like(x))
    b = geo(x * x)
    assert torch.allclose(a, b, rtol=2e-4), (a, b)


# --------------------------------------------------------------------------- #
# GeometricInstanceNormS2
# --------------------------------------------------------------------------- #
NGOLDEN = os.path.join(os.path.dirname(__file__), "golden", "geometric_instance_norm_s2.npz")


def _ncases():
    d = np.load(NGOLDEN)
    return d, json.loads(str(d["cases"]))


def test_oracle_s2_norm_matches_reference_golden():
    from oracle import losses as ol
    d, cases = _ncases()
    for i, c in enumerate(cases):
        q = ol.quadrature_weights(ol.GRID_TO_RULE[c["grid"]], c["img"], c["crop"], c["off"], normalize=True)
        x = torch.tensor(d[f"{i}_x"], requires_grad=True)
        w = torch.tensor(d[f"{i}_w"], requires_grad=True) if c["affine"] else None
        b = torch.tensor(d[f"{i}_b"], requires_grad=True) if c["affine"] else None
        y = ol.geometric_instance_norm_s2(x, q, w, b, eps=1e-5)
        (y * torch.tensor(d[f"{i}_g"])).sum().backward()
        assert np.allclose(y.detach().numpy(), d[f"{i}_y"], rtol=1e-5, atol=1e-6)
        assert np.allclose(x.grad.numpy(), d[f"{i}_dx"], rtol=1e-4, atol=1e-6)
        if c["affine"]:
            assert np.allclose(w.grad.numpy(), d[f"{i}_dw"], rtol=1e-4, atol=1e-5)
            assert np.allclose(b.grad.numpy(), d[f"{i}_db"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_hip_s2_norm_matches_reference_golden():
    import makani_amd as ma
    d, cases = _ncases()
    dev = torch.device("cuda", 0)
    for i, c in enumerate(cases):
        mod = ma.GeometricInstanceNormS2(c["img"], c["crop"], c["off"], c["grid"], num_features=5, eps=1e-5, affine=c["affine"]).to(dev)
        assert np.allclose(mod.quadrature.quad_weight.cpu().numpy(), d[f"{i}_q"], rtol=2e-7, atol=1e-12)
        if c["affine"]:
            with torch.no_grad():
                mod.weight.copy_(torch.tensor(d[f"{i}_w"]))
                mod.bias.copy_(torch.tensor(d[f"{i}_b"]))
        x = torch.tensor(d[f"{i}_x"], device=dev, requires_grad=True)
        y = mod(x)
        (y * torch.tensor(d[f"{i}_g"], device=dev)).sum().backward()

        def rel(a, b):
            return np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel(y.detach().cpu().numpy(), d[f"{i}_y"]) < 2e-6, c
        assert rel(x.grad.cpu().numpy(), d[f"{i}_dx"]) < 2e-5, c
        if c["affine"]:
            assert rel(mod.weight.grad.cpu().numpy(), d[f"{i}_dw"]) < 2e-5
            assert rel(mod.bias.grad.cpu().numpy(), d[f"{i}_db"]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_s2_norm_fused_gelu_matches_oracle(dtype):
    """norm + exact GELU in one pass (what the SFNO block uses for normalization_layer="instance_norm_s2"), forward and
    all gradients, against gelu(oracle S2 norm) in fp64"""
    import makani_amd as ma
    from oracle import losses as ol
    dev = torch.device("cuda", 0)
    B, C, H, W = 2, 5, 33, 64
    torch.manual_seed(9)
    mod = ma.GeometricInstanceNormS2((H, W), (H, W), (0, 0), "equiangular", num_features=C, eps=1e-6, affine=True).to(dev)
    with torch.no_grad():
        mod.weight.copy_(torch.rand(C) + 0.5)
        mod.bias.copy_(torch.randn(C) * 0.3)
    x0 = (torch.randn(B, C, H, W) * 2 + 1).to(dtype)
    g0 = torch.randn(B, C, H, W).to(dtype)
    x = x0.to(dev).requires_grad_(True)
    y = mod(x, fuse_gelu=True)
    assert y.dtype == dtype
    (y.float() * g0.to(dev).float()).sum().backward()
    q = ol.quadrature_weights("naive", (H, W), normalize=True).double()
    xr = x0.double().requires_grad_(True)
    wr, br = mod.weight.detach().cpu().double().requires_grad_(True), mod.bias.detach().cpu().double().requires_grad_(True)
    mean = (xr * q).sum((-2, -1), keepdim=True)
    var = ((xr - mean) ** 2 * q).sum((-2, -1), keepdim=True)
    yr = torch.nn.functional.gelu((xr - mean) / torch.sqrt(var + 1e-6) * wr.view(1, -1, 1, 1) + br.view(1, -1, 1, 1))
    (yr * g0.double()).sum().backward()

    def rel(a, b):
        return ((a.detach().cpu().double() - b.detach()).norm() / b.detach().norm()).item()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel(y, yr) < tol and rel(x.grad, xr.grad) < tol
    assert rel(mod.weight.grad, wr.grad) < tol and rel(mod.bias.grad, br.grad) < tol


@pytest.mark.gpu
def test_hip_s2_norm_fullsize_properties():
    """BASELINE grid, bf16: quadrature-weighted mean 0 / variance 1 of the output, and a constant field maps to beta"""
    import makani_amd as ma
    dev = torch.device("cuda", 0)
    H, W, C = 721, 1440, 3
    torch.manual_seed(2)
    mod = ma.GeometricInstanceNormS2((H, W), (H, W), (0, 0), "equiangular", num_features=C, affine=False).to(dev)
    x = (torch.randn(1, C, H, W, device=dev) * 3 + 5)
    y = mod(x).double()
    q = mod.quadrature.quad_weight.double()
    mean = (y * q).sum((-2, -1))
    var = (y * y * q).sum((-2, -1)) - mean ** 2
    assert mean.abs().max().item() < 1e-4 and (var - 1).abs().max().item() < 1e-3
    yb = mod(x.to(torch.bfloat16))
    assert yb.dtype == torch.bfloat16 and (yb.float() - y.float()).abs().max().item() < 0.1


# --------------------------------------------------------------------------- #
# SpectralH1Loss
# --------------------------------------------------------------------------- #
HGOLDEN = os.path.join(os.path.dirname(__file__), "golden", "spectral_h1_loss.npz")


def test_oracle_h1_loss_matches_reference_golden():
    from oracle import losses as ol
    d = np.load(HGOLDEN)
    for i, c in enumerate(json.loads(str(d["cases"]))):
        prd = torch.tensor(d[f"{i}_prd"], requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"]) if c["wgt"] else None
        out = ol.spectral_h1_loss(prd, tar, c["img"], c["grid"], c["relative"], c["squared"], wgt=w)
        (out * torch.tensor(d[f"{i}_g"])).sum().backward()
        assert np.allclose(out.detach().numpy(), d[f"{i}_out"], rtol=1e-6, atol=1e-9)
        assert np.allclose(prd.grad.numpy(), d[f"{i}_dprd"], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_hip_h1_loss_matches_reference_golden():
    import makani_amd as ma
    d = np.load(HGOLDEN)
    dev = torch.device("cuda", 0)
    for i, c in enumerate(json.loads(str(d["cases"]))):
        mod = ma.SpectralH1Loss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0), channel_names=["a", "b", "c"],
                                grid_type=c["grid"], relative=c["relative"], squared=c["squared"]).to(dev)
        prd = torch.tensor(d[f"{i}_prd"], device=dev, requires_grad=True)
        tar = torch.tensor(d[f"{i}_tar"], device=dev, requires_grad=True)
        w = torch.tensor(d[f"{i}_wgt"], device=dev) if c["wgt"] else None
        out = mod(prd, tar, w)
        (out * torch.tensor(d[f"{i}_g"], device=dev)).sum().backward()
        assert np.allclose(out.detach().cpu().numpy(), d[f"{i}_out"], rtol=5e-5, atol=1e-6), c
        for got, want in ((prd.grad, d[f"{i}_dprd"]), (tar.grad, d[f"{i}_dtar"])):
            err = np.linalg.norm(got.cpu().numpy() - want) / np.linalg.norm(want)
            assert err < 3e-5, (c, err)
    # H1 = l (l + 1) L2 for a single degree (reference tests/test_losses.py:470-509)
    lmax = 16
    isht = ma.InverseRealSHT(33, 64, lmax=lmax, mmax=lmax, grid="equiangular").to(dev)
    coef = torch.zeros(1, 1, lmax, lmax, dtype=torch.complex64, device=dev)
    coef[0, 0, 5, 2] = 1.0 + 0.5j
    x = isht(coef)
    h1 = ma.SpectralH1Loss((33, 64), (33, 64), (0, 0), ["a"], "equiangular", squared=True).to(dev)
    l2 = ma.SpectralLpLoss((33, 64), (33, 64), (0, 0), ["a"], "equiangular", p=2.0, squared=True).to(dev)
    z = torch.zeros_like(x)
    assert torch.allclose(h1(x, z), 5 * 6 * l2(x, z), rtol=1e-4)
