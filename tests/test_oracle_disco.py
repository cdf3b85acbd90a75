"""CPU tests of the DISCO / ResampleS2 restatement (oracle/disco.py) and of the product's host-side precompute
(makani_amd/disco.py: convolution tensor lists, interpolation tables).  torch-harmonics is absent from the image and the
reference tree holds no vectors for these operators, so the oracle is pinned against mathematics (see its header)."""
import math

import numpy as np
import pytest
import torch

from oracle import disco as od
from oracle.sht import precompute_latitudes

CASES = [((33, 64), (17, 32), "equiangular", "equiangular", 1.0),
         ((24, 48), (24, 48), "legendre-gauss", "legendre-gauss", 2.0),
         ((19, 36), (12, 36), "equiangular", "legendre-gauss", 1.0)]


def _cutoff(nlat, factor):
    return factor * (3 + 1) * 0.5 * math.pi / float(nlat - 1)      # fourcastnet3.py:46-50


def test_rotation_gives_great_circle_distance_and_bearing():
    lats, _ = precompute_latitudes(19, "equiangular")
    lats = torch.from_numpy(lats)
    lons = torch.linspace(0, 2 * math.pi, 37, dtype=torch.float64)[:-1]
    for t in (0, 3, 9, 18):
        theta, phi = od.rotated_coordinates(lats[t], lats, lons)
        # great-circle distance between (colat_t, lon 0) and (colat_i, lon_j)
        cosd = torch.cos(lats[t]) * torch.cos(lats)[:, None] + torch.sin(lats[t]) * torch.sin(lats)[:, None] * torch.cos(lons)[None, :]
        assert torch.allclose(theta, torch.arccos(cosd.clamp(-1, 1)), atol=1e-7)
        assert (phi >= 0).all() and (phi < 2 * math.pi + 1e-12).all()
    # a point due south of the centre (same longitude, larger colatitude) has bearing phi = 0; due east phi = pi / 2
    theta, phi = od.rotated_coordinates(lats[5], lats, lons)
    assert abs(phi[8, 0].item()) < 1e-9 or abs(phi[8, 0].item() - 2 * math.pi) < 1e-9
    assert 0.0 < phi[5, 1].item() < math.pi


@pytest.mark.parametrize("mode", ["none", "individual", "mean", "support"])
def test_normalisation_identities(mode):
    in_shape, out_shape = (33, 64), (17, 32)
    fb = od.MorletFilterBasis([3, 3])
    idx, vals = od.precompute_convolution_tensor(in_shape, out_shape, fb, theta_cutoff=_cutoff(33, 1.0), basis_norm_mode=mode)
    K, T = 9, 17
    s = torch.zeros(K * T, dtype=torch.float64).index_add_(0, idx[0] * T + idx[1], vals.abs()).reshape(K, T)
    if mode == "individual":                 # every (k, t) filter has unit 1-norm under the quadrature
        assert torch.allclose(s, torch.ones_like(s), atol=1e-6)
    if mode == "mean":                       # unit 1-norm on average over the output latitudes
        assert torch.allclose(s.mean(dim=1), torch.ones(K, dtype=torch.float64), atol=1e-6)
    if mode == "support":                    # the constant basis function (k = 0 is cos(0) cos(0) = window >= 0) integrates to <= 1
        assert (s[0] <= 1.0 + 1e-9).all()
    if mode == "none":                       # plain quadrature of the window: bounded by the area fraction of the disk
        frac = 0.5 * (1.0 - math.cos(1.001 * _cutoff(33, 1.0)))
        assert (s[0] <= frac * 1.2).all()


@pytest.mark.parametrize("in_shape,out_shape,gi,go,fac", CASES)
def test_dense_contraction_equals_direct_sum_and_is_equivariant(in_shape, out_shape, gi, go, fac):
    torch.manual_seed(3)
    m = od.DiscreteContinuousConvS2(2, 3, in_shape, out_shape, (3, 3), basis_type="morlet", grid_in=gi, grid_out=go, bias=True,
                                    theta_cutoff=_cutoff(in_shape[0], fac)).double()
    x = torch.randn(2, 2, *in_shape, dtype=torch.float64)
    y = od.disco_contraction_dense(x, m.get_psi(torch.float64), out_shape[1])
    # direct evaluation of the defining sum
    pscale = in_shape[1] // out_shape[1]
    k, t, ij = m.psi_idx
    i, j = ij // in_shape[1], ij % in_shape[1]
    ref = torch.zeros_like(y)
    for p in range(out_shape[1]):
        contrib = m.psi_vals.double() * x[:, :, i, (j + p * pscale) % in_shape[1]]
        ref[:, :, :, :, p] = torch.zeros(2, 2, 9 * out_shape[0], dtype=torch.float64).index_add_(2, k * out_shape[0] + t, contrib).reshape(2, 2, 9, out_shape[0])
    assert torch.allclose(y, ref, atol=1e-12)
    out = m(x)
    shifted = m(torch.roll(x, 2 * pscale, dims=-1))
    assert torch.allclose(shifted, torch.roll(out, 2, dims=-1), atol=1e-12)


@pytest.mark.parametrize("in_shape,out_shape,gi,go,fac", CASES)
def test_direct_contraction_equals_dense_forward_and_gradient(in_shape, out_shape, gi, go, fac):
    """oracle.disco.disco_contraction_direct (used at FourCastNet3's real grids, where the dense form is intractable on the
    CPU) is the dense roll / bmm form of torch-harmonics' CPU path: outputs and input gradients agree to fp64 round-off,
    and the module gives the same output, input gradient and weight gradient with either"""
    torch.manual_seed(4)
    m = od.DiscreteContinuousConvS2(2, 3, in_shape, out_shape, (3, 3), basis_type="morlet", grid_in=gi, grid_out=go, bias=True,
                                    theta_cutoff=_cutoff(in_shape[0], fac)).double()
    x = torch.randn(2, 2, *in_shape, dtype=torch.float64)
    g = torch.randn(2, 3, *out_shape, dtype=torch.float64)
    res = {}
    for mode in ("dense", "direct"):
        m.contraction = mode
        m.zero_grad()
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        (y * g).sum().backward()
        res[mode] = (y.detach(), xr.grad, m.weight.grad.clone())
    for a, b in zip(res["dense"], res["direct"]):
        assert torch.allclose(a, b, atol=1e-12, rtol=1e-10)


BASES = [("morlet", [3, 3]), ("morlet", [2, 4]), ("piecewise linear", [3, 4]), ("piecewise linear", [4, 3]), ("piecewise linear", [3]),
         ("piecewise linear", [2]), ("zernike", [3, 3]), ("zernike", 4)]


def _check_product_tensor(in_shape, out_shape, gi, go, cut, mode, basis, kshape):
    from makani_amd import disco as pd
    fb = od.get_filter_basis(kshape, basis)
    K = fb.kernel_size
    idx, vals = od.precompute_convolution_tensor(in_shape, out_shape, fb, grid_in=gi, grid_out=go, theta_cutoff=cut, basis_norm_mode=mode)
    psi = pd.convolution_tensor(in_shape, out_shape, kshape, basis_type=basis, grid_in=gi, grid_out=go, theta_cutoff=cut, basis_norm_mode=mode)
    assert psi["K"] == K == pd.basis_layout(basis, kshape)[1]
    size = (K, out_shape[0], in_shape[0] * in_shape[1])
    A = torch.sparse_coo_tensor(idx, vals, size=size).to_dense()
    pidx = torch.from_numpy(np.stack([psi["k"], psi["t"], psi["i"] * in_shape[1] + psi["j"]]))
    B = torch.sparse_coo_tensor(pidx, torch.from_numpy(psi["v"]), size=size).to_dense()
    # theta = arccos(z) near z = 1 carries sqrt(eps) conditioning: two fp64 evaluations agree to ~1e-8, not 1e-16
    if basis == "piecewise linear":          # the product stores every basis function on the whole disk (zeros off its support)
        assert pidx.shape[1] >= idx.shape[1] and pidx.shape[1] == K * int((psi["k"] == 0).sum())
    else:
        assert idx.shape == pidx.shape                  # same support
        assert psi["live"].all()
    if basis == "piecewise linear" and mode == "support":
        # the support of a hat includes its rim, where the value is 0 and rounding decides membership (the grid's central meridian
        # sits exactly on the sector rims): the two support sets may differ there and ONLY there, and where they do, the
        # normalisation constant (the quadrature sum over the support) differs with them — compare with it taken out again
        i0, v0 = od.precompute_convolution_tensor(in_shape, out_shape, fb, grid_in=gi, grid_out=go, theta_cutoff=cut, basis_norm_mode="none",
                                                  merge_quadrature=False)
        A0 = torch.sparse_coo_tensor(i0, v0, size=size).to_dense()
        Mo = torch.sparse_coo_tensor(i0, torch.ones_like(v0), size=size).to_dense() > 0
        Mp = torch.sparse_coo_tensor(pidx, torch.from_numpy(psi["live"].astype(np.float64)), size=size).to_dense() > 0
        rim = Mo ^ Mp
        assert A0[rim].abs().max() < 1e-9 if rim.any() else True
        _, w = precompute_latitudes(in_shape[0], gi)
        q = (torch.from_numpy(w) / in_shape[1] / 2.0).repeat_interleave(in_shape[1])
        So, Sp = (Mo * q).sum(-1, keepdim=True), (Mp * q).sum(-1, keepdim=True)
        assert (So - Sp).abs().max() <= 1.05 * (rim * q).sum(-1).max()
        A, B = A * (So + 1e-9), B * (Sp + 1e-9)
    return psi


@pytest.mark.parametrize("in_shape,out_shape,gi,go,fac", CASES)
@pytest.mark.parametrize("mode", ["mean", "individual"])
def test_product_convolution_tensor_matches_oracle(in_shape, out_shape, gi, go, fac, mode):
    from makani_amd import disco as pd
    psi = _check_product_tensor(in_shape, out_shape, gi, go, _cutoff(in_shape[0], fac), mode, "morlet", [3, 3])
    # list form: every forward list entry points inside the staged rows, the transposed lists hold the same entries
    L = pd._Lists(psi, in_shape, out_shape, "cpu")
    assert int(L.f_off[-1]) == psi["v"].size == int(L.b_off[-1])
    assert (L.f_row >= 0).all() and (L.f_row < L.max_rows).all()
    assert torch.allclose(L.f_val.double().sum(), L.b_val.double().sum(), rtol=1e-6)


@pytest.mark.parametrize("basis,kshape", BASES[1:])
@pytest.mark.parametrize("mode", ["none", "individual", "mean", "support"])
def test_product_convolution_tensor_matches_oracle_every_basis_and_norm_mode(basis, kshape, mode):
    """the host-side precompute of makani_amd/disco.py (numpy, written on its own) against the oracle's restatement of
    torch-harmonics' filter bases: morlet with a non-square shape, piecewise linear (odd / even radial counts, isotropic and
    not), zernike; all four ``basis_norm_mode``s ("support" is where the piecewise linear supports enter); FourCastNet3's
    cutoff heuristic per basis (fourcastnet3.py:46-50)"""
    in_shape, out_shape, gi, go = (25, 48), (13, 24), "equiangular", "legendre-gauss"
    n0 = kshape if isinstance(kshape, int) else kshape[0]
    factor = {"piecewise linear": 0.5, "morlet": 0.5, "zernike": math.sqrt(2.0)}[basis]
    cut = (n0 + 1) * factor * math.pi / float(in_shape[0] - 1)
    _check_product_tensor(in_shape, out_shape, gi, go, cut, mode, basis, kshape)


def test_piecewise_linear_basis_is_a_nodal_partition_of_unity():
    """pins of the piecewise linear basis that do not depend on anybody's memory of the package: every function is a hat (values
    in [0, 1]) that is 1 at its own collocation point (ring q, sector s) and 0 at every other one, and inside the outermost ring
    the functions sum to one — across the centre too, where the even radial counts fold (-r, phi + pi) onto (r, phi)"""
    rng = np.random.default_rng(5)
    for kshape in ([3], [5], [2], [4], [3, 4], [5, 3], [2, 4], [4, 3], [6, 5]):
        fb = od.PiecewiseLinearFilterBasis(kshape)
        nr, nphi = fb.kernel_shape
        K, dr = fb.kernel_size, 2.0 / (nr + 1)
        assert K == (nr // 2) * nphi + nr % 2
        # collocation points
        if nr % 2:
            pts = [(0.0, 0.0)] + [(q * dr, s * 2 * math.pi / nphi) for q in range(1, nr // 2 + 1) for s in range(nphi)]
        else:
            pts = [((q + 0.5) * dr, s * 2 * math.pi / nphi) for q in range(nr // 2) for s in range(nphi)]
        assert len(pts) == K
        r = torch.tensor([[p[0] for p in pts]], dtype=torch.float64)
        phi = torch.tensor([[p[1] for p in pts]], dtype=torch.float64)
        iidx, vals = fb.compute_support_vals(r, phi, r_cutoff=1.0)
        M = torch.zeros(K, K, dtype=torch.float64)
        M[iidx[:, 0], iidx[:, 2]] = vals
        assert torch.allclose(M, torch.eye(K, dtype=torch.float64), atol=1e-12), kshape
        # partition of unity inside the outermost ring (the isotropic even case has no fold: from its first ring outwards)
        r_hi = (nr // 2) * dr if nr % 2 else (nr // 2 - 0.5) * dr
        r_lo = 0.5 * dr if (nphi == 1 and nr % 2 == 0) else 0.0
        rr = torch.from_numpy(rng.uniform(r_lo, r_hi, (1, 400)))
        pp = torch.from_numpy(rng.uniform(0.0, 2 * math.pi, (1, 400)))
        iidx, vals = fb.compute_support_vals(rr, pp, r_cutoff=1.0)
        assert vals.min() >= -1e-12 and vals.max() <= 1.0 + 1e-12
        tot = torch.zeros(400, dtype=torch.float64).index_add_(0, iidx[:, 2], vals)
        assert torch.allclose(tot, torch.ones(400, dtype=torch.float64), atol=1e-12), kshape


def test_zernike_basis_is_orthogonal_with_the_textbook_norms():
    """int_disk Z_k Z_k' r dr dphi = eps_m pi / (2 n + 2) delta_kk' (eps_0 = 2, else 1) by Gauss-Legendre x trapezoid quadrature of
    the oracle's own values, the pyramid indexing k = n (n + 1) / 2 + l, m = 2 l - n, and three closed forms"""
    fb = od.ZernikeFilterBasis([5, 5])
    K = fb.kernel_size
    assert K == 15
    x, w = np.polynomial.legendre.leggauss(24)
    r, wr = 0.5 * (x + 1.0), 0.5 * w
    phi = 2.0 * math.pi * np.arange(64) / 64
    R, PHI = np.meshgrid(r, phi, indexing="ij")
    iidx, vals = fb.compute_support_vals(torch.from_numpy(R), torch.from_numpy(PHI), r_cutoff=1.0)
    Z = np.zeros((K,) + R.shape)
    Z[iidx[:, 0].numpy(), iidx[:, 1].numpy(), iidx[:, 2].numpy()] = vals.numpy()
    G = np.einsum("kij,lij,i->kl", Z, Z, wr * r) * (2.0 * math.pi / 64)
    want = np.zeros(K)
    for n in range(5):
        for l in range(n + 1):
            want[n * (n + 1) // 2 + l] = (2.0 if 2 * l - n == 0 else 1.0) * math.pi / (2 * n + 2)
    assert np.abs(G - np.diag(want)).max() < 1e-12
    assert np.allclose(Z[4], 2 * R ** 2 - 1)                          # (n, m) = (2, 0)
    assert np.allclose(Z[2], R * np.cos(PHI)) and np.allclose(Z[1], R * np.sin(-PHI))      # (1, 1) and (1, -1): sin(m phi), m = -1
    assert np.allclose(Z[12], 6 * R ** 4 - 6 * R ** 2 + 1)            # (4, 0)


def test_unknown_basis_and_argument_errors():
    with pytest.raises(NotImplementedError):
        od.get_filter_basis([3, 3], "harmonic")         # the reference classes' default argument: in no known release, not invented
    with pytest.raises(ValueError):
        od.DiscreteContinuousConvS2(3, 4, (9, 16), (9, 16), (3, 3), groups=2)
    from makani_amd import disco as pd
    with pytest.raises(ValueError):
        pd.DiscreteContinuousConvS2(3, 4, (9, 16), (9, 16), (3, 3), groups=2)
    with pytest.raises(ValueError):
        pd.DiscreteContinuousConvS2(4, 4, (9, 16), (9, 12), (3, 3))
    with pytest.raises(NotImplementedError):
        pd.DiscreteContinuousConvS2(4, 4, (9, 16), (9, 16), (3, 3), basis_type="harmonic")


@pytest.mark.parametrize("nin,nout,gi,go", [((12, 24), (23, 48), "legendre-gauss", "equiangular"),
                                            ((17, 32), (33, 64), "equiangular", "equiangular"),
                                            ((24, 48), (12, 24), "equiangular", "legendre-gauss"),
                                            ((9, 16), (9, 16), "equiangular", "equiangular")])
def test_resample_properties_and_product_tables(nin, nout, gi, go):
    from makani_amd import disco as pd
    m = od.ResampleS2(*nin, *nout, grid_in=gi, grid_out=go)
    lats_in, _ = precompute_latitudes(nin[0], gi)
    lats_out, _ = precompute_latitudes(nout[0], go)
    # constants are preserved; a function linear in colatitude is reproduced wherever no pole extension is involved
    assert torch.allclose(m(torch.ones(1, 2, *nin)), torch.ones(1, 2, *nout), atol=1e-6)
    f = torch.from_numpy(lats_in).float().view(1, 1, -1, 1).expand(1, 1, nin[0], nin[1]).contiguous()
    g = m(f)
    inside = (lats_out >= lats_in[0]) & (lats_out <= lats_in[-1])
    assert torch.allclose(g[0, 0, inside, 0], torch.from_numpy(lats_out[inside]).float(), atol=1e-5)
    if nin == nout and gi == go:
        x = torch.randn(1, 1, *nin)
        assert m(x) is x
    p = pd.ResampleS2(*nin, *nout, grid_in=gi, grid_out=go)
    assert p.expand_poles == m.expand_poles and p.skip_resampling == m.skip_resampling
    assert torch.equal(p.lon_l.long(), m.lon_idx_left) and torch.equal(p.lon_r.long(), m.lon_idx_right)
    assert torch.allclose(p.lon_w, m.lon_weights) and torch.allclose(p.lat_w, m.lat_weights.view(-1))
    off = 1 if m.expand_poles else 0
    a = p.lat_a.long()
    exp_a = m.lat_idx - off
    assert torch.equal(torch.where(a >= 0, a, exp_a), exp_a)      # non-polar sources agree with the (extended) row index


# --------------------------------------------------------------------------- #
# the discrete operator against the CONTINUOUS convolution integral it discretises
# --------------------------------------------------------------------------- #
def _smooth_field(px, py, pz):
    """a smooth function on the sphere, given through the Cartesian coordinates of the unit vector"""
    return 1.0 + 0.5 * px + 0.3 * py * pz + 0.2 * pz * pz - 0.4 * px * py


def _continuous_convolution(fb, k, colat_t, r_cut, n_r=160, n_phi=256):
    """(1 / 4 pi) int_0^{r_cut} int_0^{2 pi} kappa_k(r, phi) u(p(r, phi)) sin r dphi dr, evaluated in the frame of the output
    point WITHOUT the latitude-longitude grid: Gauss-Legendre nodes in the geodesic radius r, the (spectrally accurate)
    trapezoid rule in the bearing phi; p(r, phi) = R_y(colat_t) (sin r cos phi, sin r sin phi, cos r) is the closed-form inverse
    of oracle.disco.rotated_coordinates (checked in the test)."""
    x, w = np.polynomial.legendre.leggauss(n_r)
    r = 0.5 * r_cut * (x + 1.0)
    wr = 0.5 * r_cut * w
    phi = 2.0 * math.pi * np.arange(n_phi) / n_phi
    R, PHI = np.meshgrid(r, phi, indexing="ij")
    qx, qy, qz = np.sin(R) * np.cos(PHI), np.sin(R) * np.sin(PHI), np.cos(R)
    ct, st = math.cos(colat_t), math.sin(colat_t)
    px, py, pz = ct * qx + st * qz, qy, -st * qx + ct * qz
    # kernel values through the oracle's own basis class on the (r, phi) nodes
    iidx, vals = fb.compute_support_vals(torch.from_numpy(R), torch.from_numpy(PHI), r_cutoff=r_cut)
    sel = iidx[:, 0] == k
    kap = np.zeros_like(R)
    kap[iidx[sel, 1].numpy(), iidx[sel, 2].numpy()] = vals[sel].numpy()
    integrand = kap * _smooth_field(px, py, pz) * np.sin(R)
    return float((integrand.sum(axis=1) * (2.0 * math.pi / n_phi) * wr).sum() / (4.0 * math.pi))


def _discrete_convolution(fb, colat_t, nlat, nlon, r_cut, grid):
    """the quadrature sum the oracle's convolution tensor encodes (basis_norm_mode "none", quadrature merged), for all k"""
    lats, w = precompute_latitudes(nlat, grid)
    lats_t, lons = torch.from_numpy(lats), torch.linspace(0, 2 * math.pi, nlon + 1, dtype=torch.float64)[:-1]
    theta, phi = od.rotated_coordinates(torch.tensor(colat_t, dtype=torch.float64), lats_t, lons)
    iidx, vals = fb.compute_support_vals(theta, phi, r_cutoff=r_cut)
    q = torch.from_numpy(w)[iidx[:, 1]] / nlon / 2.0
    la, lo = lats_t[iidx[:, 1]], lons[iidx[:, 2]]
    u = _smooth_field(torch.sin(la) * torch.cos(lo), torch.sin(la) * torch.sin(lo), torch.cos(la))
    return torch.zeros(fb.kernel_size, dtype=torch.float64).index_add_(0, iidx[:, 0], vals * q * u).numpy()


def test_rotated_coordinates_have_the_closed_form_inverse_used_below():
    rng = np.random.default_rng(0)
    colat_t = 0.7
    r, phi = rng.uniform(0.01, 1.0, 50), rng.uniform(0, 2 * math.pi, 50)
    qx, qy, qz = np.sin(r) * np.cos(phi), np.sin(r) * np.sin(phi), np.cos(r)
    ct, st = math.cos(colat_t), math.sin(colat_t)
    px, py, pz = ct * qx + st * qz, qy, -st * qx + ct * qz
    for i in range(50):
        th, ph = od.rotated_coordinates(torch.tensor(colat_t, dtype=torch.float64), torch.tensor([math.acos(pz[i])], dtype=torch.float64),
                                        torch.tensor([math.atan2(py[i], px[i])], dtype=torch.float64))
        assert abs(th.item() - r[i]) < 1e-9 and abs(((ph.item() - phi[i] + math.pi) % (2 * math.pi)) - math.pi) < 1e-8


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
def test_disco_quadrature_converges_to_the_continuous_convolution_integral(grid):
    """An independent pin of geometry + support + quadrature weights: on a smooth field, the sum the convolution tensor encodes
    approaches (1 / 4 pi) int kappa_k u dOmega over the filter's disc, computed in the output point's own polar frame (Gauss-
    Legendre x trapezoid, no latitude-longitude grid involved), and the error falls by two orders of magnitude over two grid
    refinements, for every basis function and at polar, mid and equatorial output latitudes.  (Normalisation MODES are conventions of the package and cannot be pinned this way: mode "none".)"""
    fb = od.MorletFilterBasis([3, 3])
    r_cut = 0.3
    for colat_t in (0.12, 0.9, 0.5 * math.pi):
        exact = np.array([_continuous_convolution(fb, k, colat_t, r_cut) for k in range(fb.kernel_size)])
        scale = np.abs(exact).max()
        errs = []
        for nlat, nlon in ((91, 180), (181, 360), (361, 720)):
            got = _discrete_convolution(fb, colat_t, nlat, nlon, r_cut, grid)
            errs.append(np.abs(got - exact).max() / scale)
        assert errs[2] < 2e-6, (colat_t, errs)          # measured 1.4e-7 ... 7.1e-7 at 361 x 720 (1.2e-5 ... 5.7e-5 at 91 x 180)
        assert errs[2] < errs[0] / 20.0, (colat_t, errs)


@pytest.mark.parametrize("basis,kshape", [("piecewise linear", [3, 4]), ("piecewise linear", [2, 3]), ("zernike", 3)])
def test_disco_quadrature_converges_for_the_other_bases(basis, kshape):
    """the same pin for the bases that are not smooth: hats have kinks (second-order convergence of the grid's quadrature), Zernike
    polynomials jump to zero at the rim of the disk (first order, not monotone) — the error still falls with the grid and is
    < 2e-3 at 361 x 720 (measured: piecewise linear 2e-5 ... 8e-4, zernike 5e-4 ... 1.2e-3)"""
    fb = od.get_filter_basis(kshape, basis)
    r_cut = 0.3
    for colat_t in (0.12, 0.9, 0.5 * math.pi):
        exact = np.array([_continuous_convolution(fb, k, colat_t, r_cut, n_r=320, n_phi=1024) for k in range(fb.kernel_size)])
        scale = np.abs(exact).max()
        errs = [np.abs(_discrete_convolution(fb, colat_t, nlat, nlon, r_cut, "equiangular") - exact).max() / scale
                for nlat, nlon in ((91, 180), (361, 720))]
        assert errs[1] < 2e-3 and errs[1] < errs[0] / 3.0, (colat_t, errs)


def test_precomputed_convolution_tensor_is_that_quadrature_sum():
    """precompute_convolution_tensor (mode "none", quadrature merged) contracted with a field = the sum tested above"""
    in_shape = out_shape = (46, 90)
    fb = od.MorletFilterBasis([3, 3])
    cutoff = 0.3
    idx, vals = od.precompute_convolution_tensor(in_shape, out_shape, fb, theta_cutoff=cutoff, basis_norm_mode="none")
    lats, _ = precompute_latitudes(46, "equiangular")
    lats_t = torch.from_numpy(lats)
    lons = torch.linspace(0, 2 * math.pi, 91, dtype=torch.float64)[:-1]
    u = _smooth_field(torch.sin(lats_t)[:, None] * torch.cos(lons)[None, :], torch.sin(lats_t)[:, None] * torch.sin(lons)[None, :],
                      torch.cos(lats_t)[:, None].expand(-1, 90)).reshape(-1)
    for t in (3, 20, 40):
        sel = idx[1] == t
        got = torch.zeros(9, dtype=torch.float64).index_add_(0, idx[0][sel], vals[sel] * u[idx[2][sel]]).numpy()
        ref = _discrete_convolution(fb, float(lats[t]), 46, 90, 1.001 * cutoff, "equiangular")
        assert np.abs(got - ref).max() < 1e-12


@pytest.mark.parametrize("gi,go", [("legendre-gauss", "equiangular"), ("equiangular", "legendre-gauss")])
def test_resample_converges_to_the_field_it_interpolates(gi, go):
    """ResampleS2 (bilinear) of a smooth analytic field sampled on the input grid approaches the field's values on the output
    grid with second order in the input spacing (error / 4 per refinement), poles included (the Gauss grid has no polar rows:
    the pole extension supplies them).  FourCastNet3's decoder direction is Gauss 360 x 720 -> equiangular 721 x 1440."""
    def field(nlat, nlon, grid):
        lats, _ = precompute_latitudes(nlat, grid)
        la = torch.from_numpy(lats)[:, None]
        lo = torch.linspace(0, 2 * math.pi, nlon + 1, dtype=torch.float64)[:-1][None, :]
        return _smooth_field(torch.sin(la) * torch.cos(lo), torch.sin(la) * torch.sin(lo), torch.cos(la).expand(-1, nlon))
    errs = []
    for n in (23, 45, 90):
        nin, nout = (n, 2 * n), (2 * n + 1, 4 * n)
        m = od.ResampleS2(*nin, *nout, grid_in=gi, grid_out=go).double()
        got = m(field(*nin, gi).view(1, 1, *nin))[0, 0]
        errs.append(float((got - field(*nout, go)).abs().max()))
    assert errs[2] < 2e-3 and errs[1] < errs[0] / 3.0 and errs[2] < errs[1] / 3.0, errs
