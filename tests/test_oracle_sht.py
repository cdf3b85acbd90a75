"""Pin the SHT restatement (oracle/sht.py) against mathematics and against the
relational tests the reference holds: scipy spherical
harmonics, exact fp64 round trips, quadrature sums, Parseval, H1 = l(l+1) L2,
and the split rule."""
import math

import numpy as np
import pytest
import torch
from scipy.special import sph_harm_y

from oracle import sht as o


@pytest.mark.parametrize("grid,nlat,nlon,lmax", [("legendre-gauss", 24, 48, 24), ("equiangular", 33, 64, 16),
                                                  ("legendre-gauss", 12, 24, 12), ("equiangular", 37, 72, 12)])
def test_roundtrip_fp64(grid, nlat, nlon, lmax):
    mmax = min(lmax, nlon // 2 + 1)
    S = o.RealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
    I = o.InverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
    c = torch.tril(torch.randn(3, lmax, mmax, dtype=torch.complex128))
    c[..., 0] = c[..., 0].real.to(torch.complex128)
    assert (S(I(c)) - c).abs().max().item() < 1e-12


@pytest.mark.parametrize("grid,nlat,nlon", [("legendre-gauss", 24, 48), ("equiangular", 33, 64)])
@pytest.mark.parametrize("l,m", [(0, 0), (4, 0), (3, 2), (7, 7), (9, 4)])
def test_single_mode_matches_scipy(grid, nlat, nlon, l, m):
    lmax, mmax = 12, 12
    I = o.InverseRealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
    theta, _ = o.precompute_latitudes(nlat, grid)
    phi = np.linspace(0, 2 * np.pi, nlon, endpoint=False)
    e = torch.zeros(lmax, mmax, dtype=torch.complex128)
    e[l, m] = 1
    Y = sph_harm_y(l, m, theta[:, None], phi[None, :])
    want = Y.real if m == 0 else 2 * Y.real   # m>0 modes carry their conjugate partner
    assert np.abs(I(e).numpy() - want).max() < 1e-13


@pytest.mark.parametrize("grid", ["legendre-gauss", "equiangular", "lobatto"])
@pytest.mark.parametrize("nlat", [8, 33, 64, 181])
def test_quadrature_weights(grid, nlat):
    # reference tests/test_grids.py:136-220: non-negative (CC/LG), sum = 2 (=> 4*pi after dlambda)
    _, w = o.precompute_latitudes(nlat, grid)
    assert abs(w.sum() - 2.0) < 1e-12
    assert (w >= 0).all()
    theta, _ = o.precompute_latitudes(nlat, grid)
    assert (np.diff(theta) > 0).all() and theta[0] < 0.5 and theta[-1] > math.pi - 0.5   # north pole first


def test_constant_field_is_l0_only_and_parseval():
    # reference tests/test_losses.py:440-452,500-509
    nlat, nlon = 32, 64
    S = o.RealSHT(nlat, nlon, grid="legendre-gauss")
    c = S(torch.full((nlat, nlon), 3.0, dtype=torch.float64))
    assert abs(c[0, 0].real.item() - 3.0 * math.sqrt(4 * math.pi)) < 1e-12
    c[0, 0] = 0
    assert c.abs().max().item() < 1e-12
    # Parseval for a band-limited field: quadrature L2 == spectral L2 with m>0 counted twice
    I = o.InverseRealSHT(nlat, nlon, lmax=20, mmax=20, grid="legendre-gauss")
    S = o.RealSHT(nlat, nlon, lmax=20, mmax=20, grid="legendre-gauss")
    coef = torch.tril(torch.randn(20, 20, dtype=torch.complex128))
    coef[:, 0] = coef[:, 0].real.to(torch.complex128)
    x = I(coef)
    _, w = o.precompute_latitudes(nlat, "legendre-gauss")
    quad = (x**2 * torch.from_numpy(w)[:, None]).sum().item() * 2 * math.pi / nlon
    mult = torch.ones(20, dtype=torch.float64) * 2
    mult[0] = 1
    spec = (coef.abs() ** 2 * mult).sum().item()
    assert abs(quad - spec) / spec < 1e-12


def test_h1_is_l_lplus1_l2_for_single_mode():
    # reference tests/test_losses.py:470-498 builds the field with InverseRealSHT(grid="equiangular")
    nlat, nlon, l, m = 33, 64, 5, 3
    I = o.InverseRealSHT(nlat, nlon, lmax=16, mmax=16, grid="equiangular")
    S = o.RealSHT(nlat, nlon, lmax=16, mmax=16, grid="equiangular")
    e = torch.zeros(16, 16, dtype=torch.complex128)
    e[l, m] = 1.0
    c = S(I(e))
    ls = torch.arange(16, dtype=torch.float64)[:, None]
    l2 = (c.abs() ** 2).sum()
    h1 = ((ls * (ls + 1)) * c.abs() ** 2).sum()
    assert abs(h1 / l2 - l * (l + 1)) < 1e-9


def test_split_shapes():
    assert o.compute_split_shapes(721, 4) == [181, 181, 181, 178]
    assert o.compute_split_shapes(241, 2) == [121, 120]
    assert o.compute_split_shapes(240, 4) == [60] * 4
    assert o.compute_split_shapes(5, 4) == [1, 1, 1, 2]         # floor fallback when the last would be empty
    assert o.compute_split_shapes(9, 1) == [9]
    for size, n in [(721, 4), (1440, 2), (13, 5), (8, 8)]:
        assert sum(o.compute_split_shapes(size, n)) == size


@pytest.mark.parametrize("h,w", [(2, 1), (1, 2), (2, 2), (4, 2)])
def test_simulated_distributed_sht_equals_serial(h, w):
    # the pattern of the reference's tests/distributed/tests_distributed_layers.py:69-223
    nlat, nlon, C = 33, 64, 8
    D = o.SimulatedDistributedRealSHT(nlat, nlon, lmax=16, mmax=17, grid="equiangular", h=h, w=w)
    x = torch.randn(2, C, nlat, nlon, dtype=torch.float64)
    want = D.serial(x)
    rows = o.split_tensor_along_dim(x, -2, h)
    shards = [list(o.split_tensor_along_dim(r, -1, w)) for r in rows]
    got = D(shards)
    full = torch.cat([torch.cat(r, dim=-1) for r in got], dim=-2)
    assert (full - want).abs().max().item() < 1e-12
