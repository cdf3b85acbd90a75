"""CPU restatement of the torch-harmonics spherical harmonic transform.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  torch-harmonics is an
un-vendored dependency of the reference (pin: commit
887006c640f1d61c3f80590ecc2b207bbb647072).  The reference call sites this file
serves are

* ``makani/models/networks/sfnonet.py:786-805``  (``th.RealSHT`` /
  ``th.InverseRealSHT`` construction: positional ``(nlat, nlon)``, keywords
  ``lmax=``, ``mmax=``, ``grid=``),
* ``makani/models/common/spectral_convolution.py:169-173,239-253`` (forward /
  inverse calls, attribute reads ``.nlat .nlon .lmax .mmax .grid``),
* ``makani/utils/grids.py:20-23,205-220`` (quadrature helpers),
* ``makani/mpu/fft.py:50-51,143-182,214-249`` (split shapes and the
  all-to-all schedule of the distributed transforms).

The recipe follows the published torch-harmonics algorithm (SURVEY.md
Appendix A): ortho-normalised associated Legendre functions with
Condon-Shortley phase built by the standard three-term recurrence in fp64,
``X = 2*pi*rfft(x, norm="forward")[..., :mmax]`` followed by a quadrature sum
over latitude, and the transposed operations for the inverse.

Everything here is plain torch/numpy on the CPU and works in fp32 or fp64.
"""

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn


# --------------------------------------------------------------------------- #
# quadrature  (torch_harmonics.quadrature)
# --------------------------------------------------------------------------- #
def legendre_gauss_weights(n: int, a: float = -1.0, b: float = 1.0):
    """Gauss-Legendre nodes (ascending in cos(theta)) and weights on [a, b]."""
    xlg, wlg = np.polynomial.legendre.leggauss(n)
    xlg = (b - a) * 0.5 * xlg + (b + a) * 0.5
    wlg = wlg * (b - a) * 0.5
    return xlg, wlg


def lobatto_weights(n: int, a: float = -1.0, b: float = 1.0, tol: float = 1e-16, maxiter: int = 100):
    """Gauss-Lobatto nodes and weights (Newton iteration on the Legendre Vandermonde)."""
    wlg = np.zeros((n,))
    tlg = np.zeros((n,))
    tmp = np.zeros((n,))
    # Chebyshev nodes as the initial guess
    for i in range(n):
        tlg[i] = -np.cos(np.pi * i / (n - 1))
    tmp = 2.0
    vdm = np.zeros((n, n))
    for _ in range(maxiter):
        tmp = tlg
        vdm[:, 0] = 1.0
        vdm[:, 1] = tlg
        for k in range(2, n):
            vdm[:, k] = ((2 * k - 1) * tlg * vdm[:, k - 1] - (k - 1) * vdm[:, k - 2]) / k
        tlg = tmp - (tlg * vdm[:, n - 1] - vdm[:, n - 2]) / (n * vdm[:, n - 1])
        if max(abs(tlg - tmp).flatten()) < tol:
            break
    wlg = 2.0 / ((n * (n - 1)) * (vdm[:, n - 1] ** 2))
    tlg = (b - a) * 0.5 * tlg + (b + a) * 0.5
    wlg = wlg * (b - a) * 0.5
    return tlg, wlg


def clenshaw_curtiss_weights(n: int, a: float = -1.0, b: float = 1.0):
    """Clenshaw-Curtis nodes cos(linspace(pi, 0, n)) (poles included) and weights.

    Classic FFT construction (Waldvogel 2006), as used by torch-harmonics for
    the ``equiangular`` grid.
    """
    assert n > 1
    tcc = np.cos(np.linspace(np.pi, 0, n))
    if n == 2:
        wcc = np.array([1.0, 1.0])
    else:
        n1 = n - 1
        N = np.arange(1, n1, 2)
        ll = len(N)
        m = n1 - ll
        v = np.concatenate([2 / N / (N - 2), 1 / N[-1:], np.zeros(m)])
        v = 0 - v[:-1] - v[-1:0:-1]
        g0 = -np.ones(n1)
        g0[ll] = g0[ll] + n1
        g0[m] = g0[m] + n1
        g = g0 / (n1**2 - 1 + (n1 % 2))
        wcc = np.fft.ifft(v + g).real
        wcc = np.concatenate((wcc, wcc[:1]))
    tcc = (b - a) * 0.5 * tcc + (b + a) * 0.5
    wcc = wcc * (b - a) * 0.5
    return tcc, wcc


def precompute_latitudes(nlat: int, grid: str = "equiangular"):
    """Colatitudes (ascending, north pole first) and quadrature weights."""
    cost, w = _grid_nodes(nlat, grid)
    lats = np.flip(np.arccos(cost)).copy()
    wts = np.flip(w).copy()
    return lats, wts


def _grid_nodes(nlat: int, grid: str):
    if grid == "legendre-gauss":
        return legendre_gauss_weights(nlat, -1, 1)
    if grid == "lobatto":
        return lobatto_weights(nlat, -1, 1)
    if grid == "equiangular":
        return clenshaw_curtiss_weights(nlat, -1, 1)
    raise ValueError(f"Unknown quadrature mode {grid}")


# --------------------------------------------------------------------------- #
# Legendre functions  (torch_harmonics.legendre._precompute_legpoly)
# --------------------------------------------------------------------------- #
def precompute_legpoly(mmax: int, lmax: int, t: np.ndarray, norm: str = "ortho", inverse: bool = False, csphase: bool = True):
    """``P[m, l, k]`` = normalised associated Legendre function of degree l,
    order m at colatitude ``t[k]`` (fp64).  Zero for ``l < m``."""
    nmax = max(mmax, lmax)
    cost = np.cos(t)
    vdm = np.zeros((nmax, nmax, len(t)), dtype=np.float64)

    norm_factor = 1.0 if norm == "ortho" else np.sqrt(4 * np.pi)
    norm_factor = 1.0 / norm_factor if inverse else norm_factor

    vdm[0, 0, :] = norm_factor / np.sqrt(4 * np.pi)

    # diagonal and first off-diagonal
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * cost * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = np.sqrt((2 * l + 1) * (1 + cost) * (1 - cost) / 2 / l) * vdm[l - 1, l - 1, :]

    # three-term recurrence for the rest
    for l in range(2, nmax):
        for m in range(0, l - 1):
            vdm[m, l, :] = (
                cost * np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m)) * vdm[m, l - 1, :]
                - np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m)) * vdm[m, l - 2, :]
            )

    if norm == "schmidt":
        for l in range(0, nmax):
            if inverse:
                vdm[:, l, :] = vdm[:, l, :] * np.sqrt(2 * l + 1)
            else:
                vdm[:, l, :] = vdm[:, l, :] / np.sqrt(2 * l + 1)

    vdm = vdm[:mmax, :lmax]

    if csphase:
        for m in range(1, mmax, 2):
            vdm[m] *= -1

    return vdm


# --------------------------------------------------------------------------- #
# serial transforms  (torch_harmonics.RealSHT / InverseRealSHT)
# --------------------------------------------------------------------------- #
class RealSHT(nn.Module):
    """``(..., nlat, nlon)`` real -> ``(..., lmax, mmax)`` complex."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__()
        self.nlat = nlat
        self.nlon = nlon
        self.grid = grid
        self.norm = norm
        self.csphase = csphase

        cost, w = _grid_nodes(nlat, grid)
        if grid == "lobatto":
            self.lmax = lmax or self.nlat - 1
        else:
            self.lmax = lmax or self.nlat

        tq = np.flip(np.arccos(cost))
        self.mmax = mmax or self.nlon // 2 + 1

        pct = precompute_legpoly(self.mmax, self.lmax, tq, norm=norm, csphase=csphase)
        weights = torch.from_numpy(np.einsum("mlk,k->mlk", pct, w))
        self.register_buffer("weights", weights, persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.nlat and x.shape[-1] == self.nlon
        x = 2.0 * math.pi * torch.fft.rfft(x, dim=-1, norm="forward")
        x = torch.view_as_real(x)
        w = self.weights.to(x.dtype)
        re = torch.einsum("...km,mlk->...lm", x[..., : self.mmax, 0], w)
        im = torch.einsum("...km,mlk->...lm", x[..., : self.mmax, 1], w)
        return torch.view_as_complex(torch.stack((re, im), dim=-1).contiguous())


class InverseRealSHT(nn.Module):
    """``(..., lmax, mmax)`` complex -> ``(..., nlat, nlon)`` real."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__()
        self.nlat = nlat
        self.nlon = nlon
        self.grid = grid
        self.norm = norm
        self.csphase = csphase

        cost, _ = _grid_nodes(nlat, grid)
        if grid == "lobatto":
            self.lmax = lmax or self.nlat - 1
        else:
            self.lmax = lmax or self.nlat

        t = np.flip(np.arccos(cost))
        self.mmax = mmax or self.nlon // 2 + 1

        pct = precompute_legpoly(self.mmax, self.lmax, t, norm=norm, inverse=True, csphase=csphase)
        self.register_buffer("pct", torch.from_numpy(pct), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.lmax and x.shape[-1] == self.mmax
        x = torch.view_as_real(x)
        p = self.pct.to(x.dtype)
        rl = torch.einsum("...lm,mlk->...km", x[..., 0], p)
        im = torch.einsum("...lm,mlk->...km", x[..., 1], p)
        # imaginary parts of the m = 0 and Nyquist columns are dropped
        mask = torch.ones(self.mmax, dtype=x.dtype)
        mask[0] = 0.0
        if (self.nlon % 2 == 0) and (self.nlon // 2 < self.mmax):
            mask[self.nlon // 2] = 0.0
        im = im * mask
        xs = torch.view_as_complex(torch.stack((rl, im), dim=-1).contiguous())
        return torch.fft.irfft(xs, n=self.nlon, dim=-1, norm="forward")


# --------------------------------------------------------------------------- #
# distributed helpers  (torch_harmonics.distributed.*) - single-process
# simulation used to check the all-to-all schedule without GPUs
# --------------------------------------------------------------------------- #
def compute_split_shapes(size: int, num_chunks: int) -> List[int]:
    """ceil-div chunks, last one smaller; falls back to floor-div when the
    last chunk would be empty (reference uses at
    ``makani/mpu/fft.py:50-51`` and ``makani/utils/grids.py:154-165``)."""
    if num_chunks == 1:
        return [size]
    chunk_size = (size + num_chunks - 1) // num_chunks
    last = max(0, size - chunk_size * (num_chunks - 1))
    if last == 0:
        chunk_size = size // num_chunks
        last = size - chunk_size * (num_chunks - 1)
    return [chunk_size for _ in range(num_chunks - 1)] + [last]


def split_tensor_along_dim(tensor: torch.Tensor, dim: int, num_chunks: int):
    assert dim < tensor.dim()
    assert tensor.shape[dim] >= num_chunks
    sections = compute_split_shapes(tensor.shape[dim], num_chunks)
    return torch.split(tensor, sections, dim=dim)


def simulated_transpose(shards: List[torch.Tensor], dim0: int, dim1: int, dim1_split_sizes=None) -> List[torch.Tensor]:
    """Single-process model of ``distributed_transpose`` over one group:
    every rank splits its tensor along ``dim0`` into P chunks, chunk j goes to
    rank j, and every rank concatenates what it received along ``dim1``
    (``makani/mpu/mappings.py:38-67``)."""
    P = len(shards)
    send = [split_tensor_along_dim(s, dim0, P) for s in shards]
    return [torch.cat([send[src][dst] for src in range(P)], dim=dim1) for dst in range(P)]


class SimulatedDistributedRealSHT:
    """List-of-shards model of ``thd.DistributedRealSHT`` on an h x w grid of
    ranks (rank-major order ``[ih][iw]``).  Schedule follows the in-tree twin
    ``makani/mpu/fft.py:148-182``: (w) channels<->lon, rfft + truncate,
    (w) m<->channels, (h) channels<->lat, Legendre, (h) l<->channels."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", h=1, w=1, dtype=torch.float64):
        self.serial = RealSHT(nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
        self.h, self.w = h, w
        self.nlat, self.nlon = nlat, nlon
        self.lmax, self.mmax = self.serial.lmax, self.serial.mmax
        self.lat_shapes = compute_split_shapes(nlat, h)
        self.lon_shapes = compute_split_shapes(nlon, w)
        self.l_shapes = compute_split_shapes(self.lmax, h)
        self.m_shapes = compute_split_shapes(self.mmax, w)
        self.dtype = dtype

    def __call__(self, shards):
        h, w = self.h, self.w
        wts = self.serial.weights.to(self.dtype)
        m_off = np.cumsum([0] + self.m_shapes)
        # azimuth group: one group per ih
        out = [[None] * w for _ in range(h)]
        for ih in range(h):
            row = [shards[ih][iw] for iw in range(w)]
            if w > 1:
                row = simulated_transpose(row, -3, -1)
            row = [2.0 * math.pi * torch.fft.rfft(r, dim=-1, norm="forward")[..., : self.mmax] for r in row]
            if w > 1:
                row = simulated_transpose(row, -1, -3)
            for iw in range(w):
                out[ih][iw] = row[iw]
        # polar group: one group per iw
        res = [[None] * w for _ in range(h)]
        for iw in range(w):
            col = [out[ih][iw] for ih in range(h)]
            if h > 1:
                col = simulated_transpose(col, -3, -2)
            wl = wts[m_off[iw] : m_off[iw + 1]]
            col = [
                torch.complex(
                    torch.einsum("...km,mlk->...lm", c.real, wl),
                    torch.einsum("...km,mlk->...lm", c.imag, wl),
                )
                for c in col
            ]
            if h > 1:
                col = simulated_transpose(col, -2, -3)
            for ih in range(h):
                res[ih][iw] = col[ih]
        return res
