"""CPU restatement of torch-harmonics' discrete-continuous (DISCO) convolution on the sphere and of ``ResampleS2``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  torch-harmonics is an un-vendored dependency of the reference (pin:
commit 887006c640f1d61c3f80590ecc2b207bbb647072, absent from this image), so this file restates the PUBLISHED algorithm
(``torch_harmonics/convolution.py``, ``filter_basis.py``, ``resample.py`` as of the 0.7 / 0.8 releases) and is
**parity unpinned** against the package itself: the reference tree holds no golden vectors for these operators.  What it
is pinned against is mathematics (``tests/test_oracle_disco.py``): the rotation geometry against great-circle distances,
the normalisation identities of every ``basis_norm_mode``, rotation equivariance in longitude, the equivalence of
the two contraction forms (dense roll / bmm, as torch-harmonics' CPU path does it, and the direct quadrature sum), and —
the counterpart of the SHT's ``scipy.special.sph_harm_y`` pin — the CONTINUOUS operator itself: on a smooth field the sum the
convolution tensor encodes (mode "none") converges to (1 / 4 pi) int kappa_k u dOmega over the filter's disc, evaluated in
the output point's own polar frame by Gauss-Legendre x trapezoid quadrature without the latitude-longitude grid: relative
error 1.2e-5 ... 5.7e-5 at 91 x 180, 1.4e-7 ... 7.1e-7 at 361 x 720, all nine basis functions, polar / mid / equatorial
output latitudes, both grids.  That pins geometry, support and quadrature weights; the normalisation MODES and the
orientation convention of the basis (x = r sin phi, y = r cos phi) are conventions of the package and stay unpinned.

Reference call sites (all of FourCastNet3's local operators):
  * ``makani/models/networks/fourcastnet3.py:189-205``  encoder   ``th.DiscreteContinuousConvS2(inp, out, in_shape=,
    out_shape=, kernel_shape=, basis_type=, basis_norm_mode=, grid_in=, grid_out=, groups=, bias=, theta_cutoff=)``
  * ``fourcastnet3.py:356-381``  decoder: ``th.ResampleS2(*inp_shape, *out_shape, grid_in=, grid_out=, mode="bilinear")``
    followed by a DISCO convolution on the output grid
  * ``fourcastnet3.py:518-534``  the "local" blocks (theta_cutoff doubled)
  * ``fourcastnet3.py:46-50``    the cutoff heuristic ``(kernel_shape[0] + 1) * 0.5 * pi / (nlat - 1)``

Which torch-harmonics release each function restates, and what is known to have changed up to the reference's pin
----------------------------------------------------------------------------------------------------------------
The reference requires ``torch-harmonics >= 0.9.0`` at commit 887006c ("main as of 2026-08-04").  Neither that
commit nor any release of the package is in this image and there is no network, so NOTHING below could be diffed against the
pinned sources: the restatement follows the published algorithm as the author of this file knows it from the public 0.7.4 -
0.8.0 releases, function by function:

  ``MorletFilterBasis.compute_support_vals``   filter_basis.py as introduced in 0.7.4 (Hann window cos^2(pi r / 2 r_cutoff) times
      sin / cos products in x = r sin(phi), y = r cos(phi), basis index k = m * kernel_shape[1] + n); unchanged in 0.7.5 / 0.8.0.
  ``PiecewiseLinearFilterBasis``, ``ZernikeFilterBasis``   filter_basis.py of 0.7.4 - 0.8.0 (their docstrings give the indexing).  What
      cannot be pinned from here is convention, not mathematics: which collocation angle sector 0 sits at (0, with phi in
      [0, 2 pi)), the sign inside sin(m phi) of the Zernike functions with m < 0, and the float32 rounding the package's collocation
      points carry (see ``_isotropic``); the hats' nodal / partition-of-unity properties and the Zernike orthogonality relations are
      pinned (tests/test_oracle_disco.py).
  ``rotated_coordinates``                      convolution.py::_precompute_convolution_tensor_s2, 0.7.x - 0.8.0: YZY Euler rotation
      (alpha = -theta_out, beta = lon_in, gamma = theta_in), normalisation of (x, y, z) before arccos / atan2 (added in 0.7.4),
      phi wrapped to [0, 2 pi).
  ``precompute_convolution_tensor``            the same function: support ``theta <= theta_cutoff * (1 + theta_eps)``, theta_eps = 1e-3
      (0.7.4+; earlier releases compared against theta_cutoff itself), quadrature weights ``w / nlon_in / 2`` (0.7.5 / 0.8.0;
      0.7.3 and earlier used ``2 pi w / nlon_in`` and normalised differently: NOT restated).
  ``normalize_convolution_tensor``             convolution.py::_normalize_convolution_tensor_s2 of 0.7.5 / 0.8.0: modes "none",
      "individual", "mean", "support", ``eps = 1e-9``, ``merge_quadrature`` multiplies the weights into psi.  (0.7.4 had the
      "individual" / "mean" / "none" modes only; "support" came with 0.7.5.)
  ``disco_contraction_dense``                  _disco_s2_contraction_torch (0.6 - 0.8.0): roll by ``pscale`` columns + bmm per output
      longitude.  ``disco_contraction_direct`` is this file's equivalent form (see its docstring).
  ``DiscreteContinuousConvS2``                 convolution.py of 0.7.5 / 0.8.0: weight (out, in / groups, kernel_size) with scale
      sqrt(1 / groupsize / kernel_size), bias zeros, default theta_cutoff pi / (nlat_out - 1), einsum "bgckxy,gock->bgoxy".
  ``ResampleS2``                               resample.py of 0.7.5 / 0.8.0, mode "bilinear": pole extension by the longitude mean
      of the polar rows, ``searchsorted(side="right") - 1`` indices, lerp in latitude then longitude.

Known or suspected changes between 0.8.0 and the pin that this file does NOT reflect because they could not be read here:
additional filter bases ("harmonic" is the reference classes' DEFAULT argument, makani/models/networks/fourcastnet3.py:175,315,495,660; the
FourCastNet3 recipe itself uses "morlet", config/fourcastnet3.yaml:34 — "morlet", "piecewise linear" and "zernike", the three
bases of the 0.7.4 - 0.8.0 releases, are restated), a "bilinear-
spherical" resampling mode, optimised CUDA contraction kernels (which do not change results), and possibly further
``basis_norm_mode`` values.  Whether the normalisation constants of the restated modes changed after 0.8.0 CANNOT be known
from inside this container; if they did, every fixture built on this file (``tests/golden/fcn3_*.npz``) inherits the
difference as one per-basis-function scale of psi — which FourCastNet3's learned weights absorb, but a checkpoint trained
on the reference would not transfer bit-for-bit.  This is the open end of the "parity unpinned" statement above.

Conventions of the restated algorithm:
  * ``psi[k, t, i * nlon_in + j]`` = value of filter basis function k, centred on the output point (latitude t,
    longitude 0), at input point (latitude i, longitude j); the centre is moved to longitude p by rolling the input by
    ``p * (nlon_in // nlon_out)`` columns;
  * the centre is taken to the north pole by the YZY Euler rotation (alpha = -theta_t passive, beta = input longitude,
    gamma = input colatitude); theta = arccos z, phi = atan2(y, x) in [0, 2 pi);
  * support: theta <= (1 + theta_eps) * theta_cutoff, theta_eps = 1e-3;
  * quadrature weights q_i = w_i / (2 nlon_in) (they integrate to one over the sphere) are merged into psi after the
    normalisation of ``basis_norm_mode`` ("none", "individual", "mean", "support");
  * forward: y[b, c, k, t, p] = sum psi[k, t, (i, j)] x[b, c, i, (j + p * pscale) mod nlon_in], then
    out[b, o, t, p] = sum_{c, k} weight[o, c, k] y[b, c, k, t, p] (+ bias), groups as in a grouped convolution.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .sht import precompute_latitudes


# --------------------------------------------------------------------------- #
# filter basis (torch_harmonics.filter_basis.MorletFilterBasis)
# --------------------------------------------------------------------------- #
class MorletFilterBasis:
    """Hann-windowed sine / cosine products on the disk of radius ``r_cutoff``: basis function k = (m, n) with
    n = k % kernel_shape[1], m = k // kernel_shape[1]; even index -> cos(ceil(n/2) pi x), odd -> sin(ceil(n/2) pi x)
    on x = r sin(phi), y = r cos(phi), r scaled to the unit disk."""

    def __init__(self, kernel_shape):
        if isinstance(kernel_shape, int):
            kernel_shape = [kernel_shape, kernel_shape]
        if len(kernel_shape) != 2:
            raise ValueError("expected kernel_shape to be a list or tuple of length 2")
        self.kernel_shape = list(kernel_shape)

    @property
    def kernel_size(self):
        return self.kernel_shape[0] * self.kernel_shape[1]

    @staticmethod
    def hann_window(r, width=1.0):
        return torch.cos(0.5 * math.pi * r / width) ** 2

    def compute_support_vals(self, r, phi, r_cutoff, width=1.0):
        """r, phi: (nlat_in, nlon_in) -> (iidx (nnz, 3) = [k, i, j], vals (nnz,))"""
        ikernel = torch.arange(self.kernel_size).reshape(-1, 1, 1)
        nkernel = ikernel % self.kernel_shape[1]
        mkernel = ikernel // self.kernel_shape[1]
        iidx = torch.argwhere((r <= r_cutoff) & torch.full_like(ikernel, True, dtype=torch.bool))
        rs = r[iidx[:, 1], iidx[:, 2]] / r_cutoff
        ph = phi[iidx[:, 1], iidx[:, 2]]
        x = rs * torch.sin(ph)
        y = rs * torch.cos(ph)
        n = nkernel[iidx[:, 0], 0, 0]
        m = mkernel[iidx[:, 0], 0, 0]
        harmonic = torch.where(n % 2 == 1, torch.sin(torch.ceil(n / 2) * math.pi * x / width),
                               torch.cos(torch.ceil(n / 2) * math.pi * x / width))
        harmonic = harmonic * torch.where(m % 2 == 1, torch.sin(torch.ceil(m / 2) * math.pi * y / width),
                                          torch.cos(torch.ceil(m / 2) * math.pi * y / width))
        vals = self.hann_window(rs, width=width) * harmonic
        return iidx, vals


class PiecewiseLinearFilterBasis:
    """Tensor-product hat functions on the disk (torch_harmonics.filter_basis.PiecewiseLinearFilterBasis, 0.7.4 - 0.8.0).
    ``kernel_shape = [nr, nphi]``: nr collocation points ACROSS the diameter (spacing dr = 2 r_cutoff / (nr + 1)), nphi
    around the circle (spacing 2 pi / nphi); ``kernel_size = (nr // 2) * nphi + nr % 2`` — odd nr has the centre function
    k = 0 (isotropic) and rings at dr, 2 dr, ...; even nr has rings at dr / 2, 3 dr / 2, ... whose innermost hats reach
    across the centre (a point at (r, phi) is also the point (-r, phi + pi)).  nphi = 1: isotropic rings."""

    def __init__(self, kernel_shape):
        if isinstance(kernel_shape, int):
            kernel_shape = [kernel_shape]
        if len(kernel_shape) == 1:
            kernel_shape = [kernel_shape[0], 1]
        elif len(kernel_shape) != 2:
            raise ValueError("expected kernel_shape to be a list or tuple of length 1 or 2")
        self.kernel_shape = list(kernel_shape)

    @property
    def kernel_size(self):
        return (self.kernel_shape[0] // 2) * self.kernel_shape[1] + self.kernel_shape[0] % 2

    def _isotropic(self, r, phi, r_cutoff):
        # (the published code keeps the enumerator as an int64 tensor, whose product with a Python float is a float32 tensor: its
        # collocation radii / angles carry float32 rounding, 1e-8 relative.  Here they are float64 — the values differ at that level)
        ikernel = torch.arange(self.kernel_size, dtype=r.dtype).reshape(-1, 1, 1)
        nr = self.kernel_shape[0]
        dr = 2 * r_cutoff / (nr + 1)
        ir = ikernel * dr if nr % 2 == 1 else (ikernel + 0.5) * dr
        iidx = torch.argwhere(((r - ir).abs() <= dr) & (r <= r_cutoff))
        vals = 1 - (r[iidx[:, 1], iidx[:, 2]] - ir[iidx[:, 0], 0, 0]).abs() / dr
        return iidx, vals

    def _anisotropic(self, r, phi, r_cutoff):
        ikernel = torch.arange(self.kernel_size, dtype=r.dtype).reshape(-1, 1, 1)
        nr, nphi = self.kernel_shape
        dr = 2 * r_cutoff / (nr + 1)
        dphi = 2.0 * math.pi / nphi
        two_pi = 2.0 * math.pi

        def hat_phi(d):                                    # periodic distance in angle
            return torch.minimum(d, two_pi - d)
        if nr % 2 == 1:
            ir = ((ikernel - 1) // nphi + 1) * dr
            iphi = ((ikernel - 1) % nphi) * dphi
            cond_r = ((r - ir).abs() <= dr) & (r <= r_cutoff)
            cond_phi = (ikernel == 0) | ((phi - iphi).abs() <= dphi) | ((two_pi - (phi - iphi).abs()) <= dphi)
            iidx = torch.argwhere(cond_r & cond_phi)
            k, i, j = iidx[:, 0], iidx[:, 1], iidx[:, 2]
            dist_r = (r[i, j] - ir[k, 0, 0]).abs()
            dist_phi = (phi[i, j] - iphi[k, 0, 0]).abs()
            vals = 1 - dist_r / dr
            vals = vals * torch.where(k > 0, 1 - hat_phi(dist_phi) / dphi, torch.ones_like(vals))
        else:
            ir = (ikernel // nphi + 0.5) * dr
            iphi = (ikernel % nphi) * dphi
            rn = -r                                        # the same point seen from across the centre
            phin = torch.where(phi + math.pi >= two_pi, phi - math.pi, phi + math.pi)
            cond_r = ((r - ir).abs() <= dr) & (r <= r_cutoff)
            cond_phi = ((phi - iphi).abs() <= dphi) | ((two_pi - (phi - iphi).abs()) <= dphi)
            cond_rn = ((rn - ir).abs() <= dr) & (rn <= r_cutoff)
            cond_phin = ((phin - iphi).abs() <= dphi) | ((two_pi - (phin - iphi).abs()) <= dphi)
            iidx = torch.argwhere((cond_r & cond_phi) | (cond_rn & cond_phin))
            k, i, j = iidx[:, 0], iidx[:, 1], iidx[:, 2]
            dist_r = (r[i, j] - ir[k, 0, 0]).abs()
            dist_phi = (phi[i, j] - iphi[k, 0, 0]).abs()
            dist_rn = (rn[i, j] - ir[k, 0, 0]).abs()
            dist_phin = (phin[i, j] - iphi[k, 0, 0]).abs()
            vals = cond_r[k, i, j] * (1 - dist_r / dr) * cond_phi[k, i, j] * (1 - hat_phi(dist_phi) / dphi)
            vals = vals + cond_rn[k, i, j] * (1 - dist_rn / dr) * cond_phin[k, i, j] * (1 - hat_phi(dist_phin) / dphi)
        return iidx, vals

    def compute_support_vals(self, r, phi, r_cutoff):
        if self.kernel_shape[1] > 1:
            return self._anisotropic(r, phi, r_cutoff)
        return self._isotropic(r, phi, r_cutoff)


class ZernikeFilterBasis:
    """Zernike polynomials on the disk of radius ``r_cutoff`` (torch_harmonics.filter_basis.ZernikeFilterBasis, 0.7.4 -
    0.8.0): ``kernel_shape`` = number of radial degrees n = 0 .. kernel_shape - 1 (a tuple contributes its first entry),
    ``kernel_size = kernel_shape (kernel_shape + 1) / 2``; basis function k sits at level n, position l = 0 .. n of the
    pyramid (k = n (n + 1) / 2 + l), azimuthal order m = 2 l - n:  R_n^|m|(r) cos(m phi) for m >= 0, R_n^|m|(r) sin(m phi)
    for m < 0 (with the sign of m inside the sine, as published)."""

    def __init__(self, kernel_shape):
        if isinstance(kernel_shape, (tuple, list)):
            kernel_shape = kernel_shape[0]
        if not isinstance(kernel_shape, int):
            raise ValueError("expected kernel_shape to be an integer")
        self.kernel_shape = kernel_shape

    @property
    def kernel_size(self):
        return (self.kernel_shape * (self.kernel_shape + 1)) // 2

    @staticmethod
    def zernikeradial(r, n, m):
        """R_n^m(r) = sum_k (-1)^k (n - k)! / (k! ((n + m) / 2 - k)! ((n - m) / 2 - k)!) r^(n - 2 k), n / m integer tensors"""
        out = torch.zeros_like(r)
        bound = (n - m) // 2 + 1
        fact = lambda t: torch.exp(torch.lgamma(t.to(r.dtype) + 1.0)).round()
        for k in range(int(bound.max().item())):
            kk = torch.full_like(n, k)
            ok = kk < bound
            a, b, c = n - kk, (n + m) // 2 - kk, (n - m) // 2 - kk
            a, b, c = (torch.where(ok, t, torch.zeros_like(t)) for t in (a, b, c))      # keep the factorials defined where unused
            inc = (-1) ** k * fact(a) / (math.factorial(k) * fact(b) * fact(c)) * r ** torch.where(ok, n - 2 * kk, torch.zeros_like(n)).to(r.dtype)
            out = out + torch.where(ok, inc, torch.zeros_like(inc))
        return out

    def zernikepoly(self, r, phi, n, l):
        m = 2 * l - n
        mf = m.to(r.dtype)
        return torch.where(m < 0, self.zernikeradial(r, n, -m) * torch.sin(mf * phi), self.zernikeradial(r, n, m) * torch.cos(mf * phi))

    def compute_support_vals(self, r, phi, r_cutoff):
        ikernel = torch.arange(self.kernel_size).reshape(-1, 1, 1)
        iidx = torch.argwhere((r <= r_cutoff) & torch.full_like(ikernel, True, dtype=torch.bool))
        nshifts = torch.arange(self.kernel_shape)
        nshifts = (nshifts + 1) * nshifts // 2                   # first index of each level of the pyramid
        nkernel = torch.searchsorted(nshifts, ikernel.reshape(-1), right=True) - 1
        lkernel = ikernel.reshape(-1) - nshifts[nkernel]
        rs = r[iidx[:, 1], iidx[:, 2]] / r_cutoff
        ph = phi[iidx[:, 1], iidx[:, 2]]
        vals = self.zernikepoly(rs, ph, nkernel[iidx[:, 0]], lkernel[iidx[:, 0]])
        return iidx, vals


def get_filter_basis(kernel_shape, basis_type):
    """torch_harmonics.filter_basis.get_filter_basis of 0.7.4 - 0.8.0: "piecewise linear", "morlet", "zernike".  "harmonic" —
    the default ARGUMENT of the reference's FourCastNet3 classes (fourcastnet3.py:175,315,495,660; its recipe passes
    "morlet", config/fourcastnet3.yaml:34) — exists in no release known to the author of this file and is not invented here."""
    if basis_type == "morlet":
        return MorletFilterBasis(kernel_shape)
    if basis_type == "piecewise linear":
        return PiecewiseLinearFilterBasis(kernel_shape)
    if basis_type == "zernike":
        return ZernikeFilterBasis(kernel_shape)
    raise NotImplementedError(f"filter basis {basis_type!r} is not restated: 'morlet' (FourCastNet3's recipe, config/fourcastnet3.yaml:34), "
                              "'piecewise linear' and 'zernike' are")


# --------------------------------------------------------------------------- #
# convolution tensor (torch_harmonics.convolution._precompute_convolution_tensor_s2 / _normalize_...)
# --------------------------------------------------------------------------- #
def rotated_coordinates(lat_out, lats_in, lons_in):
    """(theta, phi) of every input point in the frame whose north pole is the output point (colatitude lat_out,
    longitude 0)."""
    alpha = -lat_out
    beta = lons_in.reshape(1, -1)
    gamma = lats_in.reshape(-1, 1)
    x = torch.cos(alpha) * torch.cos(beta) * torch.sin(gamma) + torch.cos(gamma) * torch.sin(alpha)
    y = torch.sin(beta) * torch.sin(gamma)
    z = -torch.cos(beta) * torch.sin(alpha) * torch.sin(gamma) + torch.cos(alpha) * torch.cos(gamma)
    norm = torch.sqrt(x * x + y * y + z * z)
    x, y, z = x / norm, y / norm, z / norm
    theta = torch.arccos(z)
    phi = torch.arctan2(y, x)
    phi = torch.where(phi < 0.0, phi + 2 * math.pi, phi)
    return theta, phi


def normalize_convolution_tensor(psi_idx, psi_vals, in_shape, out_shape, kernel_size, quad_weights,
                                 basis_norm_mode="mean", merge_quadrature=True, eps=1e-9):
    if basis_norm_mode == "none" and not merge_quadrature:
        return psi_vals
    ikernel, ilat_out = psi_idx[0], psi_idx[1]
    ilat_in = psi_idx[2] // in_shape[1]
    nlat_out = out_shape[0]
    q = quad_weights[ilat_in].reshape(-1)
    vnorm = torch.zeros(kernel_size, nlat_out, dtype=torch.float64)
    support = torch.zeros(kernel_size, nlat_out, dtype=torch.float64)
    flat = ikernel * nlat_out + ilat_out
    vnorm.view(-1).index_add_(0, flat, psi_vals.abs() * q)
    support.view(-1).index_add_(0, flat, q)
    if basis_norm_mode == "individual":
        val = vnorm[ikernel, ilat_out]
    elif basis_norm_mode == "mean":
        val = vnorm.mean(dim=1)[ikernel]
    elif basis_norm_mode == "support":
        val = support[ikernel, ilat_out]
    elif basis_norm_mode == "none":
        val = None
    else:
        raise ValueError(f"Unknown basis normalization mode {basis_norm_mode}.")
    out = psi_vals if val is None else psi_vals / (val + eps)
    if merge_quadrature:
        out = out * q
    return out


def precompute_convolution_tensor(in_shape, out_shape, filter_basis, grid_in="equiangular", grid_out="equiangular",
                                  theta_cutoff=0.01 * math.pi, theta_eps=1e-3, basis_norm_mode="mean",
                                  merge_quadrature=True):
    """-> (idx (3, nnz) int64 = [k, t, i * nlon_in + j], vals (nnz,) float64)"""
    nlat_in, nlon_in = in_shape
    nlat_out, nlon_out = out_shape
    lats_in, win = precompute_latitudes(nlat_in, grid=grid_in)
    lats_out, _ = precompute_latitudes(nlat_out, grid=grid_out)
    lats_in, win, lats_out = torch.from_numpy(lats_in), torch.from_numpy(win), torch.from_numpy(lats_out)
    lons_in = torch.linspace(0, 2 * math.pi, nlon_in + 1, dtype=torch.float64)[:-1]
    quad_weights = win.reshape(-1, 1) / nlon_in / 2.0
    cutoff = (1.0 + theta_eps) * theta_cutoff
    out_idx, out_vals = [], []
    for t in range(nlat_out):
        theta, phi = rotated_coordinates(lats_out[t], lats_in, lons_in)
        iidx, vals = filter_basis.compute_support_vals(theta, phi, r_cutoff=cutoff)
        out_idx.append(torch.stack([iidx[:, 0], t * torch.ones_like(iidx[:, 0]), iidx[:, 1] * nlon_in + iidx[:, 2]], dim=0))
        out_vals.append(vals)
    out_idx = torch.cat(out_idx, dim=-1).contiguous()
    out_vals = torch.cat(out_vals, dim=-1)
    out_vals = normalize_convolution_tensor(out_idx, out_vals, in_shape, out_shape, filter_basis.kernel_size, quad_weights,
                                            basis_norm_mode=basis_norm_mode, merge_quadrature=merge_quadrature)
    return out_idx, out_vals.contiguous()


def disco_contraction_dense(x, psi, nlon_out):
    """torch-harmonics' CPU form (``_disco_s2_contraction_torch``): psi (K, nlat_out, nlat_in * nlon_in) dense or sparse,
    one bmm per output longitude with the input rolled in between.  x (B, C, nlat_in, nlon_in) -> (B, C, K, nlat_out, nlon_out)"""
    B, C, nlat_in, nlon_in = x.shape
    K, nlat_out, _ = psi.shape
    assert psi.shape[-1] == nlat_in * nlon_in and nlon_in % nlon_out == 0
    pscale = nlon_in // nlon_out
    xe = x.reshape(1, B * C, nlat_in, nlon_in).permute(0, 2, 3, 1).expand(K, -1, -1, -1)
    y = torch.zeros(nlon_out, K, nlat_out, B * C, dtype=x.dtype)
    for p in range(nlon_out):
        y[p] = torch.bmm(psi, xe.reshape(K, nlat_in * nlon_in, -1))
        xe = torch.roll(xe, -pscale, dims=2)
    return y.permute(3, 1, 2, 0).reshape(B, C, K, nlat_out, nlon_out)


def _direct_fwd(x, idx, vals, in_shape, out_shape, K):
    """y[b, c, k, t, p] = sum_e vals[e] x[b, c, i_e, (j_e + p * pscale) mod nlon_in] over the entries e = (k, t, i, j) of psi:
    the defining quadrature sum, evaluated one output longitude at a time as ONE sparse (CSR) x dense product — rows (k, t),
    columns = the entries' input points shifted by p, against the input with the channels last (no dense psi, no roll of the
    input: torch-harmonics' CPU path rolls a copy of the input per longitude and multiplies by the same sparse psi)."""
    import warnings
    nlat_in, nlon_in = in_shape
    nlat_out, nlon_out = out_shape
    pscale = nlon_in // nlon_out
    B, C = x.shape[:2]
    k, t, ij = idx
    seg = k * nlat_out + t
    order = torch.argsort(seg, stable=True)
    crow = torch.zeros(K * nlat_out + 1, dtype=torch.long)
    crow[1:] = torch.cumsum(torch.bincount(seg, minlength=K * nlat_out), 0)
    i, j, v = (ij // nlon_in)[order], (ij % nlon_in)[order], vals[order].to(x.dtype)
    xT = x.reshape(B * C, nlat_in * nlon_in).t().contiguous()
    yT = torch.zeros(nlon_out, K * nlat_out, B * C, dtype=x.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                 # (torch announces its CSR support as beta, once per process)
        for p in range(nlon_out):
            psi_p = torch.sparse_csr_tensor(crow, i * nlon_in + (j + p * pscale) % nlon_in, v, size=(K * nlat_out, nlat_in * nlon_in))
            yT[p] = psi_p @ xT
    return yT.permute(2, 1, 0).reshape(B, C, K, nlat_out, nlon_out)


def _direct_adj(g, idx, vals, in_shape, out_shape, K):
    """the adjoint of ``_direct_fwd``: gx[b, c, i_e, (j_e + p * pscale) mod nlon_in] += vals[e] g[b, c, k_e, t_e, p].  Per output
    longitude the transposed tensor restricted to the input points it touches at p = 0 (a fixed sparse matrix: the shift only moves
    where its rows are added) times the gradient of that longitude, channels last."""
    import warnings
    nlat_in, nlon_in = in_shape
    nlat_out, nlon_out = out_shape
    pscale = nlon_in // nlon_out
    B, C = g.shape[:2]
    k, t, ij = idx
    seg = k * nlat_out + t
    uniq, inv = torch.unique(ij, return_inverse=True)
    order = torch.argsort(inv, stable=True)
    crow = torch.zeros(uniq.numel() + 1, dtype=torch.long)
    crow[1:] = torch.cumsum(torch.bincount(inv, minlength=uniq.numel()), 0)
    ui, uj = uniq // nlon_in, uniq % nlon_in
    gT = g.reshape(B * C, K * nlat_out, nlon_out).permute(2, 1, 0).contiguous()
    gxT = torch.zeros(nlat_in * nlon_in, B * C, dtype=g.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        psi_t = torch.sparse_csr_tensor(crow, seg[order], vals[order].to(g.dtype), size=(uniq.numel(), K * nlat_out))
        for p in range(nlon_out):
            gxT.index_add_(0, ui * nlon_in + (uj + p * pscale) % nlon_in, psi_t @ gT[p])
    return gxT.t().reshape(B, C, nlat_in, nlon_in)


class _DirectContraction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, vals, in_shape, out_shape, K):
        ctx.save_for_backward(idx, vals)
        ctx.meta = (in_shape, out_shape, K)
        return _direct_fwd(x, idx, vals, in_shape, out_shape, K)

    @staticmethod
    def backward(ctx, g):
        idx, vals = ctx.saved_tensors
        return _direct_adj(g.contiguous(), idx, vals, *ctx.meta), None, None, None, None, None


def disco_contraction_direct(x, psi_idx, psi_vals, in_shape, out_shape, kernel_size):
    """The same contraction as ``disco_contraction_dense`` written as the defining sum over the entries of psi, with a
    hand-written adjoint (no autograd graph per output longitude).  This is what makes FourCastNet3's real grids
    (721 x 1440 -> 360 x 720, 360 x 720 with the doubled cutoff, 721 x 1440) tractable on the CPU: the dense form rolls a
    (K, nlat_in, nlon_in, B * C) tensor once per output longitude and autograd keeps every rolled copy.  Pinned against the
    dense form, forward and gradient, in ``tests/test_oracle_disco.py``."""
    return _DirectContraction.apply(x, psi_idx, psi_vals.to(x.dtype), tuple(in_shape), tuple(out_shape), kernel_size)


class DiscreteContinuousConvS2(nn.Module):
    """``th.DiscreteContinuousConvS2`` (constructor, parameters and forward as published).  ``contraction``: "dense" — the
    form of torch-harmonics' CPU path — or "direct" (``disco_contraction_direct``, the same sum, for large grids); a class
    attribute so that test harnesses can switch every instance of a network at once."""

    contraction = "dense"

    def __init__(self, in_channels, out_channels, in_shape, out_shape, kernel_shape, basis_type="morlet",
                 basis_norm_mode="mean", groups=1, grid_in="equiangular", grid_out="equiangular", bias=True,
                 theta_cutoff=None):
        super().__init__()
        self.nlat_in, self.nlon_in = in_shape
        self.nlat_out, self.nlon_out = out_shape
        self.filter_basis = get_filter_basis(kernel_shape, basis_type)
        self.kernel_size = self.filter_basis.kernel_size
        if theta_cutoff is None:
            theta_cutoff = math.pi / float(self.nlat_out - 1)
        if theta_cutoff <= 0.0:
            raise ValueError("Error, theta_cutoff has to be positive.")
        self.groups = groups
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError("Error, the number of input and output channels have to be an integer multiple of the group size")
        self.groupsize = in_channels // groups
        scale = math.sqrt(1.0 / self.groupsize / self.kernel_size)
        self.weight = nn.Parameter(scale * torch.randn(out_channels, self.groupsize, self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        idx, vals = precompute_convolution_tensor(in_shape, out_shape, self.filter_basis, grid_in=grid_in, grid_out=grid_out,
                                                  theta_cutoff=theta_cutoff, basis_norm_mode=basis_norm_mode,
                                                  merge_quadrature=True)
        self.register_buffer("psi_idx", idx, persistent=False)
        self.register_buffer("psi_vals", vals.float(), persistent=False)

    def get_psi(self, dtype=torch.float32):
        """sparse COO (K, nlat_out, nlat_in * nlon_in), as torch-harmonics keeps it for its torch contraction path"""
        return torch.sparse_coo_tensor(self.psi_idx, self.psi_vals.to(dtype),
                                       size=(self.kernel_size, self.nlat_out, self.nlat_in * self.nlon_in)).coalesce()

    def forward(self, x):
        if self.contraction == "direct":
            y = disco_contraction_direct(x, self.psi_idx, self.psi_vals, (self.nlat_in, self.nlon_in),
                                         (self.nlat_out, self.nlon_out), self.kernel_size)
        else:
            y = disco_contraction_dense(x, self.get_psi(x.dtype), self.nlon_out)
        B, C, K, H, W = y.shape
        y = y.reshape(B, self.groups, self.groupsize, K, H, W)
        w = self.weight.reshape(self.groups, -1, self.weight.shape[1], self.weight.shape[2]).to(x.dtype)
        out = torch.einsum("bgckxy,gock->bgoxy", y, w).reshape(B, -1, H, W)
        if self.bias is not None:
            out = out + self.bias.reshape(1, -1, 1, 1).to(x.dtype)
        return out


# --------------------------------------------------------------------------- #
# ResampleS2 (torch_harmonics.resample.ResampleS2, mode = "bilinear")
# --------------------------------------------------------------------------- #
class ResampleS2(nn.Module):
    def __init__(self, nlat_in, nlon_in, nlat_out, nlon_out, grid_in="equiangular", grid_out="equiangular", mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError(f"unknown interpolation mode {mode}")
        self.nlat_in, self.nlon_in, self.nlat_out, self.nlon_out = nlat_in, nlon_in, nlat_out, nlon_out
        self.skip_resampling = (nlat_in == nlat_out) and (nlon_in == nlon_out) and (grid_in == grid_out)
        lats_in, _ = precompute_latitudes(nlat_in, grid=grid_in)
        lats_out, _ = precompute_latitudes(nlat_out, grid=grid_out)
        lons_in = np.linspace(0, 2 * math.pi, nlon_in, endpoint=False)
        lons_out = np.linspace(0, 2 * math.pi, nlon_out, endpoint=False)
        # points outside the latitude range of the input grid: extend the input to the poles (mean of the polar rows)
        self.expand_poles = bool((lats_out > lats_in[-1]).any() or (lats_out < lats_in[0]).any())
        if self.expand_poles:
            lats_in = np.append(np.insert(lats_in, 0, 0.0), math.pi)
        lat_idx = np.searchsorted(lats_in, lats_out, side="right") - 1
        lat_idx = np.where(lats_out == lats_in[-1], lat_idx - 1, lat_idx)
        lat_w = (lats_out - lats_in[lat_idx]) / np.diff(lats_in)[lat_idx]
        self.register_buffer("lat_idx", torch.from_numpy(lat_idx).long(), persistent=False)
        self.register_buffer("lat_weights", torch.from_numpy(lat_w).float().unsqueeze(-1), persistent=False)
        left = np.searchsorted(lons_in, lons_out, side="right") - 1
        right = np.where(lons_out >= lons_in[-1], np.zeros_like(left), left + 1)
        diff = lons_in[right] - lons_in[left]
        diff = np.where(diff < 0.0, diff + 2 * math.pi, diff)
        lon_w = (lons_out - lons_in[left]) / diff
        self.register_buffer("lon_idx_left", torch.from_numpy(left).long(), persistent=False)
        self.register_buffer("lon_idx_right", torch.from_numpy(right).long(), persistent=False)
        self.register_buffer("lon_weights", torch.from_numpy(lon_w).float(), persistent=False)

    def forward(self, x):
        if self.skip_resampling:
            return x
        if self.expand_poles:
            north = x[..., 0:1, :].mean(dim=-1, keepdim=True).expand(*x.shape[:-2], 1, x.shape[-1])
            south = x[..., -1:, :].mean(dim=-1, keepdim=True).expand(*x.shape[:-2], 1, x.shape[-1])
            x = torch.cat([north, x, south], dim=-2)
        w = self.lat_weights.to(x.dtype)
        x = torch.lerp(x[..., self.lat_idx, :], x[..., self.lat_idx + 1, :], w)
        x = torch.lerp(x[..., self.lon_idx_left], x[..., self.lon_idx_right], self.lon_weights.to(x.dtype))
        return x
