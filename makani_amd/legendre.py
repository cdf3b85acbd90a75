"""Host-side (numpy, fp64) precompute for the spherical harmonic transforms:
quadrature nodes / weights and the normalised associated Legendre functions.

Same mathematical objects as torch-harmonics' ``quadrature`` /
``legendre._precompute_legpoly`` [un-vendored; used by the reference at
``makani/models/networks/sfnonet.py:802-805``], computed here with a recurrence
vectorised over (m, latitude) per degree.  Runs once per transform object.
"""
import math

import numpy as np


def gauss_legendre(n):
    """nodes cos(theta) ascending in [-1, 1] and weights (sum = 2)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def clenshaw_curtis(n):
    """Clenshaw-Curtis rule on the equiangular grid cos(j*pi/(n-1)) incl. both poles.

    Weights from the closed form  w_j = c_j/(n-1) * (1 - sum_k b_k/(4k^2-1) cos(2 k j pi/(n-1)))
    evaluated through one FFT (Waldvogel's construction)."""
    if n < 2:
        raise ValueError("equiangular grid needs nlat >= 2")
    x = np.cos(np.linspace(np.pi, 0.0, n))
    if n == 2:
        return x, np.array([1.0, 1.0])
    n1 = n - 1
    odd = np.arange(1, n1, 2)
    nodd = len(odd)
    rest = n1 - nodd
    v = np.concatenate([2.0 / odd / (odd - 2), 1.0 / odd[-1:], np.zeros(rest)])
    v = -v[:-1] - v[-1:0:-1]
    g = -np.ones(n1)
    g[nodd] += n1
    g[rest] += n1
    g /= n1 * n1 - 1 + (n1 % 2)
    w = np.fft.ifft(v + g).real
    return x, np.concatenate([w, w[:1]])


def gauss_lobatto(n, tol=1e-16, maxiter=100):
    x = -np.cos(np.pi * np.arange(n) / (n - 1))
    P = np.zeros((n, n))
    for _ in range(maxiter):
        xo = x
        P[:, 0] = 1.0
        P[:, 1] = x
        for k in range(2, n):
            P[:, k] = ((2 * k - 1) * x * P[:, k - 1] - (k - 1) * P[:, k - 2]) / k
        x = xo - (x * P[:, n - 1] - P[:, n - 2]) / (n * P[:, n - 1])
        if np.max(np.abs(x - xo)) < tol:
            break
    w = 2.0 / (n * (n - 1) * P[:, n - 1] ** 2)
    return x, w


def grid_nodes(nlat, grid):
    if grid == "legendre-gauss":
        return gauss_legendre(nlat)
    if grid == "equiangular":
        return clenshaw_curtis(nlat)
    if grid == "lobatto":
        return gauss_lobatto(nlat)
    raise ValueError(f"Unknown quadrature mode {grid}")


def colatitudes(nlat, grid):
    """(theta ascending from the north pole, quadrature weights in the same order)."""
    x, w = grid_nodes(nlat, grid)
    return np.flip(np.arccos(x)).copy(), np.flip(w).copy()


def legendre_matrix(mmax, lmax, theta, norm="ortho", inverse=False, csphase=True):
    """P[m, l, k] (fp64): orthonormal associated Legendre functions at colatitudes theta[k]."""
    n = max(mmax, lmax)
    x = np.cos(theta)
    nk = len(theta)
    P = np.zeros((n, n, nk))
    nf = 1.0 if norm == "ortho" else math.sqrt(4 * math.pi)
    if inverse:
        nf = 1.0 / nf
    P[0, 0] = nf / math.sqrt(4 * math.pi)
    s2 = (1.0 + x) * (1.0 - x)
    for l in range(1, n):
        P[l, l] = np.sqrt((2 * l + 1) * s2 / (2 * l)) * P[l - 1, l - 1]
        P[l - 1, l] = math.sqrt(2 * l + 1) * x * P[l - 1, l - 1]
    for l in range(2, n):
        m = np.arange(0, l - 1, dtype=np.float64)[:, None]
        a = np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m))
        b = np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m))
        P[: l - 1, l] = a * x[None, :] * P[: l - 1, l - 1] - b * P[: l - 1, l - 2]
    if norm == "schmidt":
        f = np.sqrt(2.0 * np.arange(n) + 1.0)[None, :, None]
        P = P * f if inverse else P / f
    P = P[:mmax, :lmax]
    if csphase:
        P[1::2] *= -1.0
    return P


def factorize_half(nlon):
    """Radix list (4s first, then 2, 3, 5, small primes) whose product is nlon // 2."""
    if nlon % 2:
        raise NotImplementedError(f"nlon={nlon}: odd longitude counts are not supported by the HIP FFT")
    n = nlon // 2
    out = []
    while n % 4 == 0:
        out.append(4)
        n //= 4
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
        while n % p == 0:
            out.append(p)
            n //= p
    if n != 1:
        raise NotImplementedError(f"nlon={nlon}: prime factor > 31 is not supported by the HIP FFT")
    return out or [1]


def twiddle_table(nlon):
    """exp(-2 pi i q / nlon), q < nlon, as (nlon, 2) float32 (rounded from fp64)."""
    q = np.arange(nlon, dtype=np.float64)
    a = -2.0 * np.pi * q / nlon
    return np.stack([np.cos(a), np.sin(a)], axis=-1).astype(np.float32)
