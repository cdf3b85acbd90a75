"""Discrete-continuous (DISCO) convolution on the sphere and bilinear grid resampling: the local operators of
FourCastNet3.

Boundary: ``th.DiscreteContinuousConvS2`` / ``th.ResampleS2`` [torch-harmonics, un-vendored; pin
887006c640f1d61c3f80590ecc2b207bbb647072] as constructed at ``makani/models/networks/fourcastnet3.py:189-205`` (encoder),
``:356-381`` (decoder: resample + convolution on the output grid) and ``:518-534`` (local blocks): same constructor
arguments, ``weight`` (out_channels, in_channels / groups, kernel_size) and ``bias`` (out_channels) parameters with the
published initialisation, ``psi_idx`` / ``psi_vals`` buffers, ``forward((B, C, nlat_in, nlon_in)) -> (B, O, nlat_out,
nlon_out)``.

MI355X design.  The published operator is a sparse contraction ``y[b,c,k,t,p] = sum psi[k,t,(i,j)] x[b,c,i,(j + p s) mod
nlon_in]`` (s = nlon_in / nlon_out) followed by a dense channel mix ``out[o] = sum_{c,k} w[o,c,k] y[c,k]``.  Its sparsity
pattern is the same for every output longitude, so the convolution tensor is kept as per-(output latitude, basis
function) lists of (input latitude, input longitude, value) — 9 short lists per latitude instead of a COO tensor over
the whole grid:
  * equal longitude counts (local blocks, decoder; ``csrc/disco_runs.hip``): per (k, t, input row) the non-zeros form one
    circular run in longitude, so the contraction is a 1-D circular correlation; a lane owns 4 consecutive output longitudes
    and slides a register window over the row (de-interleaved LDS row image, filter values through the scalar cache, packed
    fp32 FMAs); the forward kernel shares the window between all nine basis functions.  The result is written as (B, C * K,
    nlat_out, nlon_out), i.e. as the NCHW activation the channel GEMM kernels of ``csrc/conv1x1.hip`` consume in place — the
    channel mix IS a 1x1 convolution with C * K input channels and runs on those kernels with their weight-gradient kernel
    in backward.  The two linear maps are evaluated in the cheaper order per layer: contraction first (O >= C), channel mix
    first with ``sum_k psi_k (*) z_k`` afterwards (O < C: FourCastNet3's decoders), and the data gradient as
    ``W regrouped x (psi^T (*) g)`` (``DiscoConvFn``), each through the same kernels on the tensor or its transpose;
  * strided case (encoder, nlon_in = 2 nlon_out) and shapes the run form does not cover (``csrc/disco.hip``): lists in LDS,
    a lane owns output longitudes 256 apart; adjoint as a deterministic gather (no atomics).
The convolution tensor is computed in fp64 numpy at construction (vectorised over the input grid).  Filter bases: "morlet"
(the one FourCastNet3's recipe selects, ``config/fourcastnet3.yaml:34``), "piecewise linear" and "zernike" (``basis_layout``);
the kernels take the tensor as data and do not know the basis.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import legendre as _leg
from . import ops
from ._lib import check, device_guard, dtype_code, lib, ptr, stream
from .layers import hip_conv_eligible


# --------------------------------------------------------------------------- #
# convolution tensor
# --------------------------------------------------------------------------- #
def _morlet_vals(kernel_shape, r, phi):
    """values of the kernel_shape[0] * kernel_shape[1] basis functions at unit-disk polar coordinates (r, phi): (K, n)"""
    x, y = r * np.sin(phi), r * np.cos(phi)
    window = np.cos(0.5 * math.pi * r) ** 2
    out = np.empty((kernel_shape[0] * kernel_shape[1], r.shape[0]))
    for k in range(out.shape[0]):
        n, m = k % kernel_shape[1], k // kernel_shape[1]
        hx = np.sin(math.ceil(n / 2) * math.pi * x) if n % 2 else np.cos(math.ceil(n / 2) * math.pi * x)
        hy = np.sin(math.ceil(m / 2) * math.pi * y) if m % 2 else np.cos(math.ceil(m / 2) * math.pi * y)
        out[k] = window * hx * hy
    return out


def _piecewise_linear_vals(kernel_shape, r, phi):
    """hat functions on the unit disk: nr = kernel_shape[0] collocation points across the DIAMETER (spacing dr = 2 / (nr + 1)),
    nphi = kernel_shape[1] around the circle.  Odd nr: k = 0 is the centre function, ring q = 1 .. nr // 2 at radius q dr holds
    k = 1 + (q - 1) nphi + s; even nr: ring q = 0 .. nr / 2 - 1 at (q + 1/2) dr holds k = q nphi + s, and the innermost hats reach
    across the centre, where the point (r, phi) is seen as (-r, phi + pi).  nphi = 1: rings without an angular factor.
    -> (values (K, n), support mask (K, n): the "support" normalisation integrates over it, zeros on its rim included)"""
    nr, nphi = kernel_shape
    K = (nr // 2) * nphi + nr % 2
    dr = 2.0 / (nr + 1)
    vals, live = np.zeros((K, r.shape[0])), np.zeros((K, r.shape[0]), dtype=bool)
    two_pi = 2.0 * math.pi

    def radial(rr, centre):
        d = np.abs(rr - centre)
        return 1.0 - d / dr, d <= dr

    def angular(ph, centre):
        d = np.abs(ph - centre)
        d = np.minimum(d, two_pi - d)
        return 1.0 - d / (two_pi / nphi), d <= two_pi / nphi
    for k in range(K):
        if nphi == 1:
            vals[k], live[k] = radial(r, k * dr if nr % 2 else (k + 0.5) * dr)
        elif nr % 2:
            if k == 0:
                vals[k], live[k] = radial(r, 0.0)
            else:
                hr, lr = radial(r, ((k - 1) // nphi + 1) * dr)
                hp, lp = angular(phi, ((k - 1) % nphi) * two_pi / nphi)
                vals[k], live[k] = hr * hp, lr & lp
        else:
            centre_r, centre_p = (k // nphi + 0.5) * dr, (k % nphi) * two_pi / nphi
            hr, lr = radial(r, centre_r)
            hp, lp = angular(phi, centre_p)
            hrn, lrn = radial(-r, centre_r)
            hpn, lpn = angular(np.where(phi + math.pi >= two_pi, phi - math.pi, phi + math.pi), centre_p)
            vals[k] = np.where(lr & lp, hr * hp, 0.0) + np.where(lrn & lpn, hrn * hpn, 0.0)
            live[k] = (lr & lp) | (lrn & lpn)
    return np.where(live, vals, 0.0), live


def _zernike_vals(levels, r, phi):
    """Zernike polynomials on the unit disk, radial degrees n = 0 .. levels - 1: k = n (n + 1) / 2 + l, l = 0 .. n, azimuthal
    order m = 2 l - n; R_n^|m|(r) cos(m phi) for m >= 0 and R_n^|m|(r) sin(m phi) (m negative inside the sine) otherwise"""
    out = np.empty((levels * (levels + 1) // 2, r.shape[0]))
    for n in range(levels):
        for l in range(n + 1):
            m = 2 * l - n
            a = abs(m)
            rad = np.zeros_like(r)
            for s in range((n - a) // 2 + 1):
                c = (-1) ** s * math.factorial(n - s) / (math.factorial(s) * math.factorial((n + a) // 2 - s) * math.factorial((n - a) // 2 - s))
                rad += c * r ** (n - 2 * s)
            out[n * (n + 1) // 2 + l] = rad * (np.sin(m * phi) if m < 0 else np.cos(m * phi))
    return out


BASES = ("morlet", "piecewise linear", "zernike")


def basis_layout(basis_type, kernel_shape):
    """-> (kernel_shape as the basis keeps it, number of basis functions).  torch-harmonics' ``get_filter_basis`` names of the
    0.7.4 - 0.8.0 releases; "harmonic", the default ARGUMENT of the reference's FourCastNet3 classes (fourcastnet3.py:175; its
    recipe passes "morlet"), is in no release whose source is known here and is refused rather than invented."""
    if basis_type == "morlet":
        ks = [kernel_shape, kernel_shape] if isinstance(kernel_shape, int) else list(kernel_shape)
        if len(ks) != 2:
            raise ValueError("expected kernel_shape to be a list or tuple of length 2")
        return ks, ks[0] * ks[1]
    if basis_type == "piecewise linear":
        ks = [kernel_shape] if isinstance(kernel_shape, int) else list(kernel_shape)
        if len(ks) == 1:
            ks = [ks[0], 1]
        if len(ks) != 2:
            raise ValueError("expected kernel_shape to be a list or tuple of length 1 or 2")
        return ks, (ks[0] // 2) * ks[1] + ks[0] % 2
    if basis_type == "zernike":
        n = kernel_shape if isinstance(kernel_shape, int) else int(kernel_shape[0])
        return [n], n * (n + 1) // 2
    raise NotImplementedError(f"filter basis {basis_type!r}: built are {BASES} (FourCastNet3's recipe: 'morlet', config/fourcastnet3.yaml:34)")


def convolution_tensor(in_shape, out_shape, kernel_shape, basis_type="morlet", grid_in="equiangular", grid_out="equiangular",
                       theta_cutoff=0.01 * math.pi, theta_eps=1e-3, basis_norm_mode="mean", eps=1e-9):
    """-> dict(k, t, i, j: int arrays of the non-zeros, v: float64 values with normalisation and quadrature merged).  Every
    basis function is stored at every grid point of the filter's disk (zeros outside its own support, piecewise linear basis):
    per (k, t, input row) the entries stay ONE circular run in longitude, which is what the run kernels consume."""
    kernel_shape, K = basis_layout(basis_type, kernel_shape)
    if basis_norm_mode not in ("none", "individual", "mean", "support"):
        raise ValueError(f"Unknown basis normalization mode {basis_norm_mode}.")
    nlat_in, nlon_in = in_shape
    nlat_out, _ = out_shape
    th_in, w_in = _leg.colatitudes(nlat_in, grid_in)
    th_out, _ = _leg.colatitudes(nlat_out, grid_out)
    lon = 2.0 * math.pi * np.arange(nlon_in) / nlon_in
    q_lat = w_in / nlon_in / 2.0                            # quadrature weights that integrate to one over the sphere
    cutoff = (1.0 + theta_eps) * theta_cutoff
    cb, sb = np.cos(lon)[None, :], np.sin(lon)[None, :]
    ks, ts, is_, js, vs, ms = [], [], [], [], [], []
    for t in range(nlat_out):
        # only latitudes within the cutoff of the centre can fall inside the disk
        rows = np.nonzero(np.abs(th_in - th_out[t]) <= cutoff)[0]
        if rows.size == 0:
            continue
        cg, sg = np.cos(th_in[rows])[:, None], np.sin(th_in[rows])[:, None]
        ca, sa = math.cos(-th_out[t]), math.sin(-th_out[t])
        x = ca * cb * sg + cg * sa
        y = sb * sg
        z = -cb * sa * sg + ca * cg
        nrm = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / nrm, y / nrm, z / nrm
        theta = np.arccos(np.clip(z, -1.0, 1.0))
        phi = np.arctan2(y, x)
        phi = np.where(phi < 0.0, phi + 2.0 * math.pi, phi)
        ri, jj = np.nonzero(theta <= cutoff)
        if ri.size == 0:
            continue
        live = None
        if basis_type == "morlet":
            vals = _morlet_vals(kernel_shape, theta[ri, jj] / cutoff, phi[ri, jj])          # (K, n)
        elif basis_type == "zernike":
            vals = _zernike_vals(kernel_shape[0], theta[ri, jj] / cutoff, phi[ri, jj])
        else:
            vals, live = _piecewise_linear_vals(kernel_shape, theta[ri, jj] / cutoff, phi[ri, jj])
        n = ri.size
        ms.append(np.ones(K * n, dtype=bool) if live is None else live.reshape(-1))
        ks.append(np.repeat(np.arange(K), n))
        ts.append(np.full(K * n, t))
        is_.append(np.tile(rows[ri], K))
        js.append(np.tile(jj, K))
        vs.append(vals.reshape(-1))
    k, t, i, j, v, live = (np.concatenate(a) for a in (ks, ts, is_, js, vs, ms))
    q = q_lat[i]
    flat = k * nlat_out + t
    vnorm = np.bincount(flat, weights=np.abs(v) * q, minlength=K * nlat_out).reshape(K, nlat_out)
    support = np.bincount(flat, weights=q * live, minlength=K * nlat_out).reshape(K, nlat_out)
    if basis_norm_mode == "individual":
        v = v / (vnorm[k, t] + eps)
    elif basis_norm_mode == "mean":
        v = v / (vnorm.mean(axis=1)[k] + eps)
    elif basis_norm_mode == "support":
        v = v / (support[k, t] + eps)
    v = v * q
    return dict(k=k.astype(np.int64), t=t.astype(np.int64), i=i.astype(np.int64), j=j.astype(np.int64), v=v, K=K, live=live)


_PSI_CACHE = {}          # constructor arguments -> convolution tensor (FourCastNet3's eight local blocks share one)
_LIST_CACHE = {}         # (constructor arguments, device) -> device lists


def _psi_key(in_shape, out_shape, kernel_shape, basis_type, grid_in, grid_out, theta_cutoff, basis_norm_mode):
    return (tuple(in_shape), tuple(out_shape), tuple(kernel_shape), basis_type, grid_in, grid_out, float(theta_cutoff),
            basis_norm_mode)


class _Lists:
    """device-side list form of one convolution tensor (forward lists per (t, k), transposed lists per input latitude).
    ``window = (in_first, in_count, out_first, out_count)``: the operator of a rank that owns those OUTPUT latitudes and has
    gathered the input latitudes ``in_first .. in_first + in_count`` they touch (own rows + halo), re-indexed to the window."""

    def __init__(self, psi, in_shape, out_shape, device, window=None):
        nlat_in, nlon_in = in_shape
        nlat_out, nlon_out = out_shape
        K = psi["K"]
        k, t, i, j, v = psi["k"], psi["t"], psi["i"], psi["j"], psi["v"]
        if window is not None:
            i0, ni, t0, nt = window
            sel = (t >= t0) & (t < t0 + nt)
            k, t, i, j, v = k[sel], t[sel] - t0, i[sel] - i0, j[sel], v[sel]
            assert i.size == 0 or (i.min() >= 0 and i.max() < ni), "window does not cover the stencil"
            nlat_in, nlat_out = ni, nt
            in_shape, out_shape = (ni, nlon_in), (nt, nlon_out)
        # forward: sorted by (t, k); input rows relative to the first row the output latitude touches
        order = np.lexsort((j, i, k, t))
        kf, tf, if_, jf, vf = k[order], t[order], i[order], j[order], v[order]
        lat_lo = np.full(nlat_out, 0, np.int64)
        lat_n = np.zeros(nlat_out, np.int64)
        for tt in range(nlat_out):
            sel = if_[tf == tt]
            if sel.size:
                lat_lo[tt], lat_n[tt] = sel.min(), sel.max() - sel.min() + 1
        off = np.zeros(nlat_out * K + 1, np.int64)
        np.add.at(off, tf * K + kf + 1, 1)
        off = np.cumsum(off)
        to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(device)
        self.f_off, self.f_row, self.f_lon = to(off, np.int32), to(if_ - lat_lo[tf], np.int32), to(jf, np.int32)
        self.f_val = to(vf, np.float32)
        self.lat_lo, self.lat_n = to(lat_lo, np.int32), to(lat_n, np.int32)
        self.max_rows = max(1, int(lat_n.max()))
        # transposed: sorted by input latitude
        order = np.lexsort((j, k, t, i))
        kb, tb, ib, jb, vb = k[order], t[order], i[order], j[order], v[order]
        boff = np.zeros(nlat_in + 1, np.int64)
        np.add.at(boff, ib + 1, 1)
        boff = np.cumsum(boff)
        self.b_off, self.b_k, self.b_t, self.b_lon = to(boff, np.int32), to(kb, np.int32), to(tb, np.int32), to(jb, np.int32)
        self.b_val = to(vb, np.float32)
        self.K, self.in_shape, self.out_shape = K, tuple(in_shape), tuple(out_shape)
        self.nnz = int(v.size)
        # same longitude count on both grids: the adjoint is the forward correlation over the lists transposed per
        # (input latitude, basis function), longitudes negated, rows = output latitudes relative to the first one touched
        self.same_lon = nlon_in == nlon_out
        if self.same_lon:
            order = np.lexsort((j, t, k, i))
            ks, ts, is2, js, vs2 = k[order], t[order], i[order], j[order], v[order]
            seg = is2 * K + ks
            t_lo = np.zeros(nlat_in * K, np.int64)
            t_n = np.zeros(nlat_in * K, np.int64)
            lo = np.full(nlat_in * K, nlat_out, np.int64)
            hi = np.full(nlat_in * K, -1, np.int64)
            np.minimum.at(lo, seg, ts)
            np.maximum.at(hi, seg, ts)
            live = hi >= 0
            t_lo[live], t_n[live] = lo[live], hi[live] - lo[live] + 1
            soff = np.zeros(nlat_in * K + 1, np.int64)
            np.add.at(soff, seg + 1, 1)
            soff = np.cumsum(soff)
            self.s_off, self.s_row = to(soff, np.int32), to(ts - t_lo[seg], np.int32)
            self.s_lon, self.s_val = to((-js) % nlon_in, np.int32), to(vs2, np.float32)
            self.t_lo, self.t_n = to(t_lo, np.int32), to(t_n, np.int32)
            self.max_rows_b = max(1, int(t_n.max()))
        # the run form of the same tensor (sliding-window kernels) where it applies
        self.runs = None
        self._transposed = None
        if self.same_lon and runs_radix(nlon_in) is not None and v.size:
            self.runs = _RunLists(dict(k=k, t=t, i=i, j=j, v=v, K=K), ((nlat_in, nlon_in), (nlat_out, nlon_out)), device)
            self._tsrc = (dict(k=k, t=i, i=t, j=(-j) % nlon_in, v=v, K=K), ((nlat_out, nlon_out), (nlat_in, nlon_in)), device)

    def transposed(self):
        """run lists of the TRANSPOSED tensor (roles of input and output latitude exchanged, longitudes negated): with them the
        adjoint-shaped kernel computes sum_k psi_k (*) z_k (K planes in, one out) and the forward-shaped one its adjoint"""
        if self._transposed is None and self.runs is not None:
            psi_t, shape, device = self._tsrc
            self._transposed = _TLists(_RunLists(psi_t, shape, device), self.K, shape[0], shape[1], self.nnz)
        return self._transposed


class _TLists:
    """what ``_contract_fwd`` / ``_contract_bwd`` read from a list object, for the transposed tensor (run form only)"""

    def __init__(self, runs, K, in_shape, out_shape, nnz):
        self.runs, self.K, self.in_shape, self.out_shape, self.nnz = runs, K, tuple(in_shape), tuple(out_shape), nnz
        self.same_lon = True


def runs_radix(nlon):
    """longitudes a lane owns in the run-form kernels (csrc/disco_runs.hip: N / R lanes in at most three waves), or None"""
    if nlon % 4 == 0 and nlon // 4 >= 2 and ((nlon // 4 + 63) // 64 <= 3 or (nlon // 4 + 63) // 64 == 6):
        return 4
    if nlon % 8 == 0 and 128 < nlon // 8 <= 192:
        return 8
    return None


def _build_runs(seg, row, lon, val, nseg, nlon, R):
    """entries (segment, image row, longitude, value) -> the run form of csrc/disco_runs.hip: for every (segment, row) the
    longitudes are split into circular runs of consecutive longitudes (one run when the filter support is a disc); a run is
    {row, first longitude, offset of its values, groups of R values}, values zero-padded to whole groups.
    Returns seg_off (nseg + 1), runs (n, 4) int32, vals float32 (R trailing zeros: the kernels prefetch one group ahead)."""
    order = np.lexsort((lon, row, seg))
    seg, row, lon, val = seg[order], row[order], lon[order], val[order]
    n = seg.size
    key = seg * (int(row.max()) + 1 if n else 1) + row
    first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]]) if n else np.zeros(0, np.int64)
    last = np.r_[first[1:], n]
    out_runs, out_vals, counts = [], [], np.zeros(nseg, np.int64)
    voff = 0
    for a, b in zip(first, last):
        l, v = lon[a:b], val[a:b]
        if b - a == nlon:
            pieces = [(0, v)]
        else:
            cuts = np.flatnonzero(np.diff(l) != 1) + 1
            idx = np.split(np.arange(b - a), cuts)
            pieces = [(int(l[i[0]]), v[i]) for i in idx]
            if len(pieces) > 1 and l[0] == 0 and l[-1] == nlon - 1:             # the run that crosses longitude 0
                pieces = [(pieces[-1][0], np.concatenate([pieces[-1][1], pieces[0][1]]))] + pieces[1:-1]
        for js, pv in pieces:
            ng = (pv.size + R - 1) // R
            out_runs.append((int(row[a]), js, voff, ng))
            pad = np.zeros(ng * R, np.float32)
            pad[:pv.size] = pv
            out_vals.append(pad)
            voff += ng * R
        counts[seg[a]] += len(pieces)
    seg_off = np.zeros(nseg + 1, np.int64)
    seg_off[1:] = np.cumsum(counts)
    runs = np.array(out_runs, np.int32).reshape(-1, 4)
    vals = np.concatenate(out_vals + [np.zeros(R, np.float32)])
    return seg_off.astype(np.int32), runs, vals


def _build_fused(t, k, row, lon, val, nlat_out, K, nlon):
    """entries (output latitude, basis function, image row, longitude, value) -> the streams of the fused forward kernel
    (csrc/disco_runs.hip: disco_fused_fwd_kernel): per (t, row) the UNION over k of the longitudes is cut into circular runs,
    each aligned down to a multiple of 4 longitudes; a run is {row, first slot = first longitude / 4, value offset, groups}
    and carries groups x K x 4 values ([group][k][tau], zeros where a basis function has no tap)."""
    R = 4
    order = np.lexsort((lon, row, t))
    t, k, row, lon, val = t[order], k[order], row[order], lon[order], val[order]
    n = t.size
    key = t * (int(row.max()) + 1 if n else 1) + row
    first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]]) if n else np.zeros(0, np.int64)
    last = np.r_[first[1:], n]
    out_runs, out_vals, counts = [], [], np.zeros(nlat_out, np.int64)
    voff = 0
    for a, b in zip(first, last):
        l, kk, v = lon[a:b], k[a:b], val[a:b]
        ul = np.unique(l)
        if ul.size == nlon:
            pieces = [(0, nlon)]
        else:
            cuts = np.flatnonzero(np.diff(ul) != 1) + 1
            segs = np.split(ul, cuts)
            pieces = [(int(sg[0]), sg.size) for sg in segs]
            if len(pieces) > 1 and ul[0] == 0 and ul[-1] == nlon - 1:             # the run that crosses longitude 0
                pieces = [(pieces[-1][0], pieces[-1][1] + pieces[0][1])] + pieces[1:-1]
        for js, cnt in pieces:
            ja = js - js % R
            ng = (js - ja + cnt + R - 1) // R
            blk = np.zeros((ng, K, R), np.float32)
            off = (l - ja) % nlon
            sel = off < (js - ja) + cnt                                          # the entries of this run
            sel &= off >= (js - ja)
            blk[off[sel] // R, kk[sel], off[sel] % R] = v[sel]
            out_runs.append((int(row[a]), ja // R, voff, ng))
            out_vals.append(blk.reshape(-1))
            voff += blk.size
        counts[t[a]] += len(pieces)
    seg_off = np.zeros(nlat_out + 1, np.int64)
    seg_off[1:] = np.cumsum(counts)
    runs = np.array(out_runs, np.int32).reshape(-1, 4)
    vals = np.concatenate(out_vals + [np.zeros(K * R, np.float32)])
    return seg_off.astype(np.int32), runs, vals


class _RunLists:
    """run-form lists of one convolution tensor with nlon_in == nlon_out (forward: segments (t, k), image rows = input
    latitudes relative to the first one latitude t touches; adjoint: segments (i, k), image rows = output latitudes relative to
    the first one the latitude group (i // LG) touches for basis function k, longitudes negated; built per LG on demand)"""

    def __init__(self, psi, shape, device):
        (nlat_in, nlon), (nlat_out, _) = shape
        K = psi["K"]
        k, t, i, j, v = psi["k"], psi["t"], psi["i"], psi["j"], psi["v"]
        self.R = R = runs_radix(nlon)
        self._dev = device
        self._e = (k, t, i, j, v.astype(np.float32))
        to = self._to
        lat_lo = np.zeros(nlat_out, np.int64)
        lat_hi = np.full(nlat_out, -1, np.int64)
        lo = np.full(nlat_out, nlat_in, np.int64)
        np.minimum.at(lo, t, i)
        np.maximum.at(lat_hi, t, i)
        live = lat_hi >= 0
        lat_lo[live] = lo[live]
        lat_n = np.where(live, lat_hi - lat_lo + 1, 0)
        so, rn, vl = _build_runs(t * K + k, i - lat_lo[t], j, self._e[4], nlat_out * K, nlon, R)
        self.f_seg, self.f_runs, self.f_vals = to(so), to(rn), to(vl)
        self.lat_lo, self.lat_n = to(lat_lo.astype(np.int32)), to(lat_n.astype(np.int32))
        self.max_rows = max(1, int(lat_n.max()))
        self.K, self.in_shape, self.out_shape, self.nnz = K, (nlat_in, nlon), (nlat_out, nlon), int(v.size)
        self._groups, self._bwd, self._fused, self._frows = {}, {}, {}, {}
        for LG in (4, 2):                      # the image rows of a latitude group: cheap, decides which LG fits the LDS
            gseg = (i // LG) * K + k
            ngrp = (nlat_in + LG - 1) // LG
            t_lo = np.full(ngrp * K, nlat_out, np.int64)
            t_hi = np.full(ngrp * K, -1, np.int64)
            np.minimum.at(t_lo, gseg, t)
            np.maximum.at(t_hi, gseg, t)
            liveg = t_hi >= 0
            t_n = np.where(liveg, t_hi - t_lo + 1, 0)
            self._groups[LG] = (np.where(liveg, t_lo, 0), t_n, max(1, int(t_n.max())))

    def _to(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self._dev)

    def fused(self, LG):
        """(seg_off, runs, vals, lat_lo, lat_n, max_rows) of the fused forward kernel for groups of LG output latitudes"""
        if LG not in self._fused:
            k, t, i, j, v = self._e
            nlat_out, nlon = self.out_shape
            ngrp = (nlat_out + LG - 1) // LG
            g_lo = np.full(ngrp, self.in_shape[0], np.int64)
            g_hi = np.full(ngrp, -1, np.int64)
            np.minimum.at(g_lo, t // LG, i)
            np.maximum.at(g_hi, t // LG, i)
            live = g_hi >= 0
            g_lo = np.where(live, g_lo, 0)
            g_n = np.where(live, g_hi - g_lo + 1, 0)
            so, rn, vl = _build_fused(t, k, i - g_lo[t // LG], j, v, nlat_out, self.K, nlon)
            self._fused[LG] = (self._to(so), self._to(rn), self._to(vl), self._to(g_lo.astype(np.int32)), self._to(g_n.astype(np.int32)),
                               max(1, int(g_n.max())))
        return self._fused[LG]

    def fused_rows(self, LG):
        """image rows of the fused forward kernel (without building its lists)"""
        if LG not in self._frows:
            _, t, i, _, _ = self._e
            ngrp = (self.out_shape[0] + LG - 1) // LG
            g_lo = np.full(ngrp, self.in_shape[0], np.int64)
            g_hi = np.full(ngrp, -1, np.int64)
            np.minimum.at(g_lo, t // LG, i)
            np.maximum.at(g_hi, t // LG, i)
            self._frows[LG] = max(1, int((g_hi - g_lo + 1).max()))
        return self._frows[LG]

    def max_rows_b(self, LG):
        return self._groups[LG][2]

    def bwd(self, LG):
        """(seg_off, runs, vals, t_lo, t_n, max_rows) of the adjoint for latitude groups of LG"""
        if LG not in self._bwd:
            k, t, i, j, v = self._e
            K, (nlat_in, nlon) = self.K, self.in_shape
            t_lo, t_n, mr = self._groups[LG]
            so, rn, vl = _build_runs(i * K + k, t - t_lo[(i // LG) * K + k], (-j) % nlon, v, nlat_in * K, nlon, self.R)
            self._bwd[LG] = (self._to(so), self._to(rn), self._to(vl), self._to(t_lo.astype(np.int32)), self._to(t_n.astype(np.int32)), mr)
        return self._bwd[LG]


def _runs_enabled(L):
    return getattr(L, "runs", None) is not None and os.environ.get("MAKANI_AMD_DISCO", "runs") == "runs"


def _runs_shape(L, planes, dtype, max_rows, img_bf16, pb=0):
    R, PB = C.c_int(0), C.c_int(pb)
    code = _lib.MK_BF16 if dtype == torch.bfloat16 else _lib.MK_F32
    ok = lib().mk_disco_runs_shape(L.in_shape[1], max_rows, planes, code, img_bf16, C.byref(R), C.byref(PB))
    return (R.value, PB.value) if ok and R.value == L.runs.R else None


def _fused_plan(L, planes):
    """LG (output latitudes per workgroup) when the fused forward kernel (all K basis functions per stream) takes the launch"""
    if not _runs_enabled(L) or os.environ.get("MAKANI_AMD_DISCO_FUSED", "1") != "1":
        return None
    LG, PB = C.c_int(0), C.c_int(0)
    for lg in (4, 2):            # the library names the LG of this longitude count; the row count belongs to that LG
        if lib().mk_disco_fused_shape(L.in_shape[1], L.K, L.runs.fused_rows(lg), planes, C.byref(LG), C.byref(PB)) and LG.value == lg:
            return lg
    return None


def _runs_plan_fwd(L, planes, dtype):
    """(R, PB, img_bf16) when the run-form forward kernel takes this launch, else None.  bf16 tensors are widened to fp32 in
    the LDS image when four planes of it fit (one conversion when staged instead of one per read: the kernel is bound by
    VALU issue), MAKANI_AMD_DISCO_IMG=b forces the bf16 image."""
    if not _runs_enabled(L):
        return None
    force_b = os.environ.get("MAKANI_AMD_DISCO_IMG", "") == "b"
    for img, pb in ((0, 4), (1, 4), (0, 2), (1, 2)):
        if (img and dtype != torch.bfloat16) or (force_b and not img and dtype == torch.bfloat16):
            continue
        got = _runs_shape(L, planes, dtype, L.runs.max_rows, img, pb)
        if got is not None:
            return got[0], got[1], img
    return None


def _runs_plan_bwd(L, planes, dtype):
    """(R, PB, img_bf16, LG) for the run-form adjoint: four latitudes per workgroup (12 waves: the four SIMDs evenly loaded)
    with an fp32 image of four planes when that fits the LDS, then two latitudes, then the bf16 image, then two planes"""
    if not _runs_enabled(L):
        return None
    force_b = os.environ.get("MAKANI_AMD_DISCO_IMG", "") == "b"
    pref = os.environ.get("MAKANI_AMD_DISCO_BWD", "")             # "LG,img,PB" (experiments)
    order = [tuple(int(c) for c in pref.split(","))] if pref else [(4, 0, 4), (2, 0, 4), (4, 1, 4), (2, 1, 4), (4, 0, 2), (4, 1, 2)]
    nw = (L.in_shape[1] // L.runs.R + 63) // 64
    for LG, img, pb in order:
        if (img and dtype != torch.bfloat16) or (force_b and not img and dtype == torch.bfloat16) or nw * LG > 12:
            continue
        got = _runs_shape(L, planes, dtype, L.runs.max_rows_b(LG), img, pb)
        if got is not None:
            return got[0], got[1], img, LG
    return None


def _contract_fwd(x, L: _Lists):
    B, Cc, nlat_in, nlon_in = x.shape
    nlat_out, nlon_out = L.out_shape
    y = torch.empty((B, Cc * L.K, nlat_out, nlon_out), dtype=x.dtype, device=x.device)
    if B * Cc == 0:            # a rank of the azimuth group that got no channel (fewer channels than ranks)
        return y
    fplan = _fused_plan(L, B * Cc)
    if fplan is not None:
        f_seg, f_runs, f_vals, g_lo, g_n, mr = L.runs.fused(fplan)
        with ops._timed(f"disco_fwd_{nlat_in}x{nlon_in}_p{B * Cc}", flops=2.0 * B * Cc * nlon_out * L.nnz,
                        nbytes=float(x.element_size()) * (x.numel() + y.numel())):
            check(lib().mk_disco_fwd_fused(ptr(x), ptr(y), dtype_code(x), ptr(f_seg), ptr(f_runs), ptr(f_vals), ptr(g_lo), ptr(g_n), mr,
                                           B * Cc, L.K, nlat_in, nlon_in, nlat_out, stream()), "mk_disco_fwd_fused")
        return y
    plan = _runs_plan_fwd(L, B * Cc, x.dtype)
    if plan is not None:
        RL = L.runs
        with ops._timed(f"disco_fwd_{nlat_in}x{nlon_in}_p{B * Cc}", flops=2.0 * B * Cc * nlon_out * L.nnz,
                        nbytes=float(x.element_size()) * (x.numel() + y.numel())):
            check(lib().mk_disco_fwd_runs(ptr(x), ptr(y), dtype_code(x), ptr(RL.f_seg), ptr(RL.f_runs), ptr(RL.f_vals), ptr(RL.lat_lo),
                                          ptr(RL.lat_n), RL.max_rows, B * Cc, L.K, nlat_in, nlon_in, nlat_out, plan[0], plan[1],
                                          plan[2], stream()), "mk_disco_fwd_runs")
        return y
    with ops._timed(f"disco_fwd_{nlat_in}x{nlon_in}_p{B * Cc}", flops=2.0 * B * Cc * nlon_out * L.nnz,
                    nbytes=float(x.element_size()) * (x.numel() + y.numel())):
        check(lib().mk_disco_fwd(ptr(x), ptr(y), dtype_code(x), ptr(L.f_off), ptr(L.f_row), ptr(L.f_lon), ptr(L.f_val),
                                 ptr(L.lat_lo), ptr(L.lat_n), L.max_rows, B * Cc, L.K, nlat_in, nlon_in, nlat_out, nlon_out,
                                 stream()), "mk_disco_fwd")
    return y


def _contract_bwd(gy, L: _Lists):
    B, CK, nlat_out, nlon_out = gy.shape
    nlat_in, nlon_in = L.in_shape
    Cc = CK // L.K
    gx = torch.empty((B, Cc, nlat_in, nlon_in), dtype=gy.dtype, device=gy.device)
    if B * Cc == 0:
        return gx
    plan = _runs_plan_bwd(L, B * Cc, gy.dtype)
    if plan is not None:
        b_seg, b_runs, b_vals, t_lo, t_n, mrb = L.runs.bwd(plan[3])
        with ops._timed(f"disco_bwd_{nlat_in}x{nlon_in}_p{B * Cc}", flops=2.0 * B * Cc * nlon_out * L.nnz,
                        nbytes=float(gy.element_size()) * (gx.numel() + gy.numel())):
            check(lib().mk_disco_bwd_runs(ptr(gy), ptr(gx), dtype_code(gy), ptr(b_seg), ptr(b_runs), ptr(b_vals), ptr(t_lo), ptr(t_n), mrb,
                                          B * Cc, L.K, nlat_in, nlon_in, nlat_out, plan[0], plan[1], plan[2], plan[3], stream()),
                  "mk_disco_bwd_runs")
        return gx
    with ops._timed(f"disco_bwd_{nlat_in}x{nlon_in}_p{B * Cc}", flops=2.0 * B * Cc * nlon_out * L.nnz,
                    nbytes=float(gy.element_size()) * (gx.numel() + gy.numel())):
        if L.same_lon:
            check(lib().mk_disco_bwd_same(ptr(gy), ptr(gx), dtype_code(gy), ptr(L.s_off), ptr(L.s_row), ptr(L.s_lon), ptr(L.s_val),
                                          ptr(L.t_lo), ptr(L.t_n), L.max_rows_b, B * Cc, L.K, nlat_in, nlon_in, nlat_out,
                                          stream()), "mk_disco_bwd_same")
        else:
            check(lib().mk_disco_bwd(ptr(gy), ptr(gx), dtype_code(gy), ptr(L.b_off), ptr(L.b_k), ptr(L.b_t), ptr(L.b_lon),
                                     ptr(L.b_val), B * Cc, L.K, nlat_in, nlon_in, nlat_out, nlon_out, stream()), "mk_disco_bwd")
    return gx


class DiscoContractFn(torch.autograd.Function):
    """y = psi (*) x, (B, C, nlat_in, nlon_in) -> (B, C * K, nlat_out, nlon_out); linear with a constant tensor"""

    @staticmethod
    def forward(ctx, x, lists):
        ctx.lists = lists
        return _contract_fwd(x.contiguous(), lists)

    @staticmethod
    def backward(ctx, gy):
        return _contract_bwd(gy.contiguous(), ctx.lists), None


class GroupMixFn(torch.autograd.Function):
    """z[b, g, r, n] = sum_c W[g, r, c] x[b, g, c, n] for 8-9 planes per group (``csrc/groupmix.hip``: one streaming pass instead
    of M = 9 batched library GEMMs); x (B, G, CG, N) f32 | bf16, W (G, RG, CG)"""

    @staticmethod
    def supported(x, W):
        vec = 8 if x.dtype == torch.bfloat16 else 4
        return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % vec == 0 and x.shape[0] * x.shape[1] <= 65535
                and bool(lib().mk_group_mix_supported(W.shape[2], W.shape[1])))

    @staticmethod
    def forward(ctx, x, W):
        B, G, CG, N = x.shape
        RG = W.shape[1]
        x = x.contiguous()
        Wf = W.detach().to(x.dtype).float().contiguous()        # under bf16 autocast the reference multiplies bf16-rounded weights
        z = torch.empty((B, G, RG, N), dtype=x.dtype, device=x.device)
        with ops._timed(f"group_mix_c{CG}_r{RG}_n{N}", nbytes=float(x.element_size()) * (x.numel() + z.numel())):
            check(lib().mk_group_mix(ptr(x), ptr(Wf), ptr(z), dtype_code(x), B, G, CG, RG, N, stream()), "mk_group_mix")
        ctx.save_for_backward(x, Wf)
        ctx.wdtype = W.dtype
        return z

    @staticmethod
    def backward(ctx, gz):
        x, Wf = ctx.saved_tensors
        B, G, CG, N = x.shape
        RG = Wf.shape[1]
        gz = gz.contiguous()
        gx = gW = None
        if ctx.needs_input_grad[0]:
            Wt = Wf.transpose(1, 2).contiguous()
            gx = torch.empty_like(x)
            with ops._timed(f"group_mix_c{RG}_r{CG}_n{N}", nbytes=float(x.element_size()) * (x.numel() + gz.numel())):
                check(lib().mk_group_mix(ptr(gz), ptr(Wt), ptr(gx), dtype_code(x), B, G, RG, CG, N, stream()), "mk_group_mix")
        if ctx.needs_input_grad[1]:
            nb = lib().mk_group_mix_blocks(N, dtype_code(x), B * G)
            part = torch.empty((B, G, nb, RG, CG), dtype=torch.float32, device=x.device)
            with ops._timed(f"group_mix_wgrad_c{CG}_r{RG}_n{N}", nbytes=float(x.element_size()) * (x.numel() + gz.numel())):
                check(lib().mk_group_mix_wgrad(ptr(x), ptr(gz), ptr(part), dtype_code(x), B, G, CG, RG, N, stream()), "mk_group_mix_wgrad")
            gW = part.sum(dim=(0, 2)).to(ctx.wdtype)
        return gx, gW


def _group_mix(W, x):
    """W (G, RG, CG) applied to x (B, G, CG, N): the HIP streaming kernel for FourCastNet3's group sizes (8-9 planes per
    group), else the package's fp32 GEMM engine batched over (sample, group) — never a library GEMM"""
    if GroupMixFn.supported(x, W):
        return GroupMixFn.apply(x, W)
    return ops.GroupMmFn.apply(x, W)


class DiscoSumFn(torch.autograd.Function):
    """out = sum_k psi_k (*) z_k, (B, O * K, nlat_in, nlon) -> (B, O, nlat_out, nlon): the contraction AFTER the channel mix
    (``lists_t``: the transposed tensor's run lists, whose adjoint-shaped kernel is this map and whose forward-shaped kernel is
    its adjoint)"""

    @staticmethod
    def forward(ctx, z, lists_t):
        ctx.lists_t = lists_t
        return _contract_bwd(z.contiguous(), lists_t)

    @staticmethod
    def backward(ctx, g):
        return _contract_fwd(g.contiguous(), ctx.lists_t), None


class DiscoConvFn(torch.autograd.Function):
    """out = W (psi (*) x) (+ bias) for groups = 1 on bf16 planes, contraction first (one plane in, K planes out: the fused
    forward kernel), then the channel GEMM.  The data gradient is evaluated in the OTHER order of the same adjoint,
    gx[c] = sum_{o, k} W[o, c, k] (psi_k^T (*) g[o]): the transposed tensor's one-in-K-out kernel on the output gradient, then a
    GEMM with the weight regrouped as (C, O K) — instead of W^T g followed by the K-in-one-out adjoint kernel, which shares no
    row window between basis functions and runs at half the rate (FourCastNet3 local block: 9.4 instead of 16.2 ms)."""

    @staticmethod
    def forward(ctx, x, weight, bias, L):
        O, Cc, K = weight.shape
        y = _contract_fwd(x.contiguous(), L)                                      # (B, C * K, H, W)
        out, _ = ops.conv1x1_nn(ops.pad_weight_bf16(weight.reshape(O, Cc * K)), Cc * K, y, bias=bias)
        ctx.save_for_backward(y, weight)
        ctx.L, ctx.has_bias = L, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        y, weight = ctx.saved_tensors
        O, Cc, K = weight.shape
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = ops.conv1x1_wgrad(g, y).view(O, Cc, K)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = ops._sum_planes(g)
        if ctx.needs_input_grad[0]:
            u = _contract_fwd(g, ctx.L.transposed())                              # (B, O * K, H_in, W): u[o K + k] = psi_k^T (*) g[o]
            gx, _ = ops.conv1x1_nn(ops.pad_weight_bf16(weight.permute(1, 0, 2).reshape(Cc, O * K)), O * K, u)
        return gx, gw, gb, None


class DiscreteContinuousConvS2(nn.Module):
    def __init__(self, in_channels, out_channels, in_shape, out_shape, kernel_shape, basis_type="morlet",
                 basis_norm_mode="mean", groups=1, grid_in="equiangular", grid_out="equiangular", bias=True,
                 theta_cutoff=None):
        super().__init__()
        self.nlat_in, self.nlon_in = in_shape
        self.nlat_out, self.nlon_out = out_shape
        self.kernel_shape, self.kernel_size = basis_layout(basis_type, kernel_shape)
        if self.nlon_in % self.nlon_out != 0:
            raise ValueError("nlon_in must be an integer multiple of nlon_out")
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
        self._key = _psi_key(in_shape, out_shape, self.kernel_shape, basis_type, grid_in, grid_out, theta_cutoff, basis_norm_mode)
        if self._key not in _PSI_CACHE:
            _PSI_CACHE[self._key] = convolution_tensor(in_shape, out_shape, self.kernel_shape, basis_type=basis_type,
                                                       grid_in=grid_in, grid_out=grid_out, theta_cutoff=theta_cutoff,
                                                       basis_norm_mode=basis_norm_mode)
        psi = self._psi = _PSI_CACHE[self._key]
        idx = np.stack([psi["k"], psi["t"], psi["i"] * self.nlon_in + psi["j"]])
        self.register_buffer("psi_idx", torch.from_numpy(idx), persistent=False)
        self.register_buffer("psi_vals", torch.from_numpy(psi["v"]).float(), persistent=False)

    def _device_lists(self, device):
        key = (self._key, str(device))
        if key not in _LIST_CACHE:
            _LIST_CACHE[key] = _Lists(self._psi, (self.nlat_in, self.nlon_in), (self.nlat_out, self.nlon_out), device)
        return _LIST_CACHE[key]

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        bf16 = hip_conv_eligible(x)
        with torch.autocast(device_type="cuda", enabled=False):
            xc = x.to(torch.bfloat16) if bf16 else x.float()
            L = self._device_lists(x.device)
            if self._mix_first(L, xc):
                return self._mix_then_contract(xc, L)
            if bf16 and self._fused_adjoint(L, xc):
                return DiscoConvFn.apply(xc, self.weight, self.bias, L)
            y = DiscoContractFn.apply(xc, L)                                      # (B, C * K, H, W)
            return self._channel_mix(y, bf16)

    def _fused_adjoint(self, L, xc):
        """groups = 1 on bf16 planes with the fused one-in-K-out kernel available in both directions (see DiscoConvFn)"""
        if self.groups != 1 or os.environ.get("MAKANI_AMD_DISCO_ADJ", "fused") != "fused" or not _runs_enabled(L):
            return False
        if (L.out_shape[0] * L.out_shape[1]) % 8 or (L.in_shape[0] * L.in_shape[1]) % 8:
            return False
        Lt = L.transposed()
        return (Lt is not None and _fused_plan(L, xc.shape[0] * xc.shape[1]) is not None
                and _fused_plan(Lt, xc.shape[0] * self.weight.shape[0]) is not None)

    def _mix_first(self, L, xc):
        """fewer output than input channels (FourCastNet3's decoders: 45 -> 5 per pressure level): mixing the channels first,
        z[o, k] = sum_c w[o, c, k] x[c], and contracting afterwards, out[o] = sum_k psi_k (*) z[o, k], is the same linear map
        with O * K instead of C * K plane contractions and without the (B, C * K, H, W) intermediate (22 GB fp32 in the decoder)"""
        O, gs, _ = self.weight.shape
        if O >= gs * self.groups or os.environ.get("MAKANI_AMD_DISCO_MIXFIRST", "1") != "1" or not _runs_enabled(L):
            return False
        Lt = L.transposed()
        planes = xc.shape[0] * O
        return Lt is not None and _runs_plan_bwd(Lt, planes, xc.dtype) is not None and (
            _fused_plan(Lt, planes) is not None or _runs_plan_fwd(Lt, planes, xc.dtype) is not None)

    def _mix_then_contract(self, xc, L):
        B, Cc, H, W = xc.shape
        O, gs, K = self.weight.shape
        G = self.groups
        # z[b, g, (o, k), n] = sum_c w[g, o, c, k] x[b, g, c, n]: a batched GEMM that leaves z in place as planes o * K + k (an
        # einsum would transpose the activations to put the batch last: two copies of the largest tensors of the decoder)
        wm = self.weight.reshape(G, O // G, gs, K).permute(0, 1, 3, 2).reshape(G, (O // G) * K, gs)
        z = _group_mix(wm, xc.reshape(B, G, gs, H * W)).reshape(B, O * K, H, W)
        out = DiscoSumFn.apply(z, L.transposed())
        if self.bias is not None:
            out = out + self.bias.to(out.dtype).view(1, -1, 1, 1)
        return out

    def _channel_mix(self, y, bf16):
        """out[o] = sum_{c, k} weight[o, c, k] y[c * K + k] (+ bias): a 1x1 convolution over the C * K channels"""
        O = self.weight.shape[0]
        if self.groups == 1:
            w4 = self.weight.reshape(O, -1, 1, 1)
            if bf16 and (y.shape[-1] * y.shape[-2]) % 8 == 0:
                return ops.Conv1x1Fn.apply(y, w4, self.bias, None)
            out = ops.ConvMmFn.apply(y.float() if not bf16 else y, w4, None, False)
        else:
            B, _, H, W = y.shape
            yg = y.reshape(B, self.groups, self.groupsize * self.kernel_size, H * W)
            wg = self.weight.reshape(self.groups, O // self.groups, self.groupsize * self.kernel_size)
            out = _group_mix(wg, yg).reshape(B, O, H, W)                          # per (batch, group), output in place
        if self.bias is not None:
            out = out + self.bias.to(out.dtype).view(1, -1, 1, 1)
        return out


# --------------------------------------------------------------------------- #
# h x w spatial model parallelism (thd.DistributedDiscreteContinuousConvS2 / thd.DistributedResampleS2)
# --------------------------------------------------------------------------- #
def _p2p(send, recv, group):
    """one batched point-to-point round: ``send`` / ``recv`` map a group rank to a contiguous tensor.  RCCL runs the batch as
    one grouped launch (neighbour traffic rides single xGMI links); gloo (tests) moves host memory, so GPU tensors are staged."""
    import torch.distributed as dist
    host = dist.get_backend(group) == "gloo" and any(t_.is_cuda for t_ in list(send.values()) + list(recv.values()))
    hs = {p_: (t_.cpu() if host else t_) for p_, t_ in send.items()}
    hr = {p_: (torch.empty(t_.shape, dtype=t_.dtype) if host else t_) for p_, t_ in recv.items()}
    ops_ = []
    for peer in sorted(set(hs) | set(hr)):
        gp = dist.get_global_rank(group, peer)
        if peer in hs:
            ops_.append(dist.P2POp(dist.isend, hs[peer], gp, group=group))
        if peer in hr:
            ops_.append(dist.P2POp(dist.irecv, hr[peer], gp, group=group))
    if ops_:
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
    if host:
        for p_, t_ in recv.items():
            t_.copy_(hr[p_])


class _HaloPlan:
    """which input latitudes every polar rank needs for ITS output latitudes, from the support of the convolution tensor:
    ``win[r] = (first, count)`` and, for the pair (r, p), the rows of p's shard inside r's window."""

    def __init__(self, psi, lat_in_shapes, lat_out_shapes):
        P = len(lat_in_shapes)
        self.in_off = np.concatenate([[0], np.cumsum(lat_in_shapes)]).astype(int)
        self.out_off = np.concatenate([[0], np.cumsum(lat_out_shapes)]).astype(int)
        self.win = []
        for r in range(P):
            sel = (psi["t"] >= self.out_off[r]) & (psi["t"] < self.out_off[r + 1])
            if sel.any():
                lo, hi = int(psi["i"][sel].min()), int(psi["i"][sel].max()) + 1
            else:
                lo, hi = int(self.in_off[r]), int(self.in_off[r]) + 1
            self.win.append((lo, hi - lo))

    def rows(self, r, p):
        """global input rows [a, b) of rank p's shard that lie in rank r's window (empty: a >= b)"""
        lo, n = self.win[r]
        return max(lo, int(self.in_off[p])), min(lo + n, int(self.in_off[p + 1]))


class _HaloGatherFn(torch.autograd.Function):
    """local latitude rows -> this rank's input window (own rows + the halo rows the neighbours hold); the adjoint sends the
    halo parts of the window gradient back to their owners, which add them in rank order (deterministic)."""

    @staticmethod
    def forward(ctx, x, plan, group, me):
        P = len(plan.win)
        send, recv, parts = {}, {}, []
        for p_ in range(P):
            a, b = plan.rows(p_, me)                       # what peer p_ needs from my shard
            if p_ != me and a < b:
                send[p_] = x[..., a - plan.in_off[me]:b - plan.in_off[me], :].contiguous()
            a, b = plan.rows(me, p_)                       # what I need from peer p_
            if a < b:
                if p_ == me:
                    parts.append(x[..., a - plan.in_off[me]:b - plan.in_off[me], :])
                else:
                    recv[p_] = x.new_empty((*x.shape[:-2], b - a, x.shape[-1]))
                    parts.append(recv[p_])
        _p2p(send, recv, group)
        ctx.meta = (plan, group, me, x.shape)
        return torch.cat(parts, dim=-2) if len(parts) > 1 else parts[0].contiguous()

    @staticmethod
    def backward(ctx, g):
        plan, group, me, shape = ctx.meta
        P = len(plan.win)
        lo = plan.win[me][0]
        gx = g.new_zeros(shape)
        send, recv = {}, {}
        for p_ in range(P):
            a, b = plan.rows(me, p_)                       # window rows owned by p_: their gradient goes home
            if a < b:
                if p_ == me:
                    gx[..., a - plan.in_off[me]:b - plan.in_off[me], :] = g[..., a - lo:b - lo, :]
                else:
                    send[p_] = g[..., a - lo:b - lo, :].contiguous()
            a, b = plan.rows(p_, me)
            if p_ != me and a < b:
                recv[p_] = g.new_empty((*shape[:-2], b - a, shape[-1]))
        _p2p(send, recv, group)
        for p_ in sorted(recv):
            a, b = plan.rows(p_, me)
            gx[..., a - plan.in_off[me]:b - plan.in_off[me], :] += recv[p_]
        return gx, None, None, None


class DistributedDiscreteContinuousConvS2(DiscreteContinuousConvS2):
    """``thd.DistributedDiscreteContinuousConvS2`` (constructed at ``fourcastnet3.py:189-205,356-381,518-534`` when
    ``comm.get_size("spatial") > 1``): local ``(B, C, nlat_in_loc, nlon_in_loc) -> (B, O, nlat_out_loc, nlon_out_loc)``, same
    results as the serial operator on the gathered field.  The schedule differs from the published one on purpose.  The
    reference lets every rank contract its input latitudes into partial sums for ALL output latitudes and reduce-scatters that
    (B, C K, nlat_out, nlon_out) tensor over the polar group — K = 9 times the activation through a ring collective.  The
    convolution tensor has compact support (``theta_cutoff``: a few latitude rows), so here each rank computes ITS output
    latitudes only and first fetches the few input rows beyond its shard from the neighbours that hold them: one batched
    point-to-point exchange of (B, C, halo, nlon) — point-to-point neighbour traffic is what xGMI links are, and it is ~K x
    nlat_loc / halo times fewer bytes.  Steps: (1) all-to-all over the azimuth group trades the longitude split for a channel
    split (whole latitude circles: the contraction is a correlation in longitude); (2) halo gather over the polar group; (3) local
    contraction (HIP kernels of the serial operator with the window's lists); (4) the second all-to-all restores the longitude
    split; (5) the channel mix is local.  The weight is replicated (``is_shared_mp = ["spatial"]`` is set by the caller)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from . import distributed as thd
        if not thd.is_initialized():
            raise RuntimeError("makani_amd.distributed.init(polar_group, azimuth_group) has not been called")
        self.comm_size_polar, self.comm_rank_polar = thd.polar_group_size(), thd.polar_group_rank()
        self.comm_size_azimuth, self.comm_rank_azimuth = thd.azimuth_group_size(), thd.azimuth_group_rank()
        self.lat_in_shapes = thd.compute_split_shapes(self.nlat_in, self.comm_size_polar)
        self.lon_in_shapes = thd.compute_split_shapes(self.nlon_in, self.comm_size_azimuth)
        self.lat_out_shapes = thd.compute_split_shapes(self.nlat_out, self.comm_size_polar)
        self.lon_out_shapes = thd.compute_split_shapes(self.nlon_out, self.comm_size_azimuth)
        self.nlat_in_local = self.lat_in_shapes[self.comm_rank_polar]
        self.nlat_out_local = self.lat_out_shapes[self.comm_rank_polar]
        self._plan = _HaloPlan(self._psi, self.lat_in_shapes, self.lat_out_shapes)

    def window(self):
        """(first input row, input rows, first output row, output rows) of this rank's local operator"""
        r = self.comm_rank_polar
        return (*self._plan.win[r], int(self._plan.out_off[r]), self.nlat_out_local)

    def _device_lists(self, device):
        if self.comm_size_polar == 1:
            return super()._device_lists(device)
        key = (self._key, str(device), self.window())
        if key not in _LIST_CACHE:
            _LIST_CACHE[key] = _Lists(self._psi, (self.nlat_in, self.nlon_in), (self.nlat_out, self.nlon_out), device,
                                      window=self.window())
        return _LIST_CACHE[key]

    def _spatial_contract(self, xc, contract):
        """steps (1)-(4): ``contract`` is the local contraction (B, C_loc, window rows, nlon_in) -> (B, C_loc * K, nlat_out_loc,
        nlon_out) (the HIP kernels; the CPU schedule test passes a torch stand-in)"""
        from . import distributed as thd
        chan_shapes = thd.compute_split_shapes(xc.shape[1], self.comm_size_azimuth)
        if self.comm_size_azimuth > 1:          # channels <-> longitude
            xc = thd.transpose(xc, 1, chan_shapes, 3, self.lon_in_shapes, thd.azimuth_group())
        if self.comm_size_polar > 1:
            xc = _HaloGatherFn.apply(xc, self._plan, thd.polar_group(), self.comm_rank_polar)
        y = contract(xc)
        if self.comm_size_azimuth > 1:
            y = thd.transpose(y, 3, self.lon_out_shapes, 1, [c * self.kernel_size for c in chan_shapes], thd.azimuth_group())
        return y.contiguous()

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        bf16 = hip_conv_eligible(x)
        with torch.autocast(device_type="cuda", enabled=False):
            xc = x.to(torch.bfloat16) if bf16 else x.float()
            lists = self._device_lists(x.device)
            y = self._spatial_contract(xc, lambda t: DiscoContractFn.apply(t, lists))
            return self._channel_mix(y, bf16)


# --------------------------------------------------------------------------- #
# ResampleS2 (bilinear)
# --------------------------------------------------------------------------- #
class ResampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        ctx.mod = mod
        ctx.in_shape = x.shape
        return mod._launch(x.contiguous(), False)

    @staticmethod
    def backward(ctx, gy):
        return ctx.mod._launch(gy.contiguous(), True), None


class ResampleS2(nn.Module):
    """``th.ResampleS2(nlat_in, nlon_in, nlat_out, nlon_out, grid_in=, grid_out=, mode="bilinear")``: linear
    interpolation in colatitude (the input is extended to the poles by the mean of its first / last row when the output
    grid reaches beyond it), then periodic linear interpolation in longitude.  One HIP kernel for the forward map (gather
    of four input points) and one for its adjoint (deterministic gather over the precomputed inverse stencils); float32
    or bfloat16 tensors of any leading shape."""

    def __init__(self, nlat_in, nlon_in, nlat_out, nlon_out, grid_in="equiangular", grid_out="equiangular", mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError(f"unknown interpolation mode {mode}")
        self.nlat_in, self.nlon_in, self.nlat_out, self.nlon_out = nlat_in, nlon_in, nlat_out, nlon_out
        self.mode = mode
        self.skip_resampling = (nlat_in == nlat_out) and (nlon_in == nlon_out) and (grid_in == grid_out)
        lats_in, _ = _leg.colatitudes(nlat_in, grid_in)
        lats_out, _ = _leg.colatitudes(nlat_out, grid_out)
        lons_in = np.linspace(0, 2 * math.pi, nlon_in, endpoint=False)
        lons_out = np.linspace(0, 2 * math.pi, nlon_out, endpoint=False)
        self.expand_poles = bool((lats_out > lats_in[-1]).any() or (lats_out < lats_in[0]).any())
        if self.expand_poles:
            lats_in = np.append(np.insert(lats_in, 0, 0.0), math.pi)
        lat_idx = np.searchsorted(lats_in, lats_out, side="right") - 1
        lat_idx = np.where(lats_out == lats_in[-1], lat_idx - 1, lat_idx)
        lat_w = ((lats_out - lats_in[lat_idx]) / np.diff(lats_in)[lat_idx]).astype(np.float32)
        left = np.searchsorted(lons_in, lons_out, side="right") - 1
        right = np.where(lons_out >= lons_in[-1], np.zeros_like(left), left + 1)
        diff = lons_in[right] - lons_in[left]
        diff = np.where(diff < 0.0, diff + 2 * math.pi, diff)
        lon_w = ((lons_out - lons_in[left]) / diff).astype(np.float32)
        # the operator as two sparse 1-D maps.  Latitude: out row t = (1 - w) * R[a] + w * R[a + 1] over the (possibly pole
        # extended) rows R; an extended row is the mean over longitude of input row 0 / nlat_in - 1, flagged by index -1 / -2.
        off = 1 if self.expand_poles else 0

        def src(r):                      # extended row index -> input row (or -1 north mean, -2 south mean)
            if not self.expand_poles:
                return r
            return -1 if r == 0 else (-2 if r == nlat_in + 1 else r - off)
        la = np.array([src(r) for r in lat_idx], np.int32)
        lb = np.array([src(r + 1) for r in lat_idx], np.int32)
        self.register_buffer("lat_a", torch.from_numpy(la), persistent=False)
        self.register_buffer("lat_b", torch.from_numpy(lb), persistent=False)
        self.register_buffer("lat_w", torch.from_numpy(lat_w), persistent=False)
        self.register_buffer("lon_l", torch.from_numpy(left.astype(np.int32)), persistent=False)
        self.register_buffer("lon_r", torch.from_numpy(right.astype(np.int32)), persistent=False)
        self.register_buffer("lon_w", torch.from_numpy(lon_w), persistent=False)

    def _inverse_stencils(self, device):
        """adjoint operator as gather lists (CSR over input rows / input columns, plus the rows fed by a polar mean)"""
        key = str(device)
        if getattr(self, "_ikey", None) != key:
            la, lb, lw = self.lat_a.cpu().numpy(), self.lat_b.cpu().numpy(), self.lat_w.cpu().numpy()
            lat_lists = [[] for _ in range(self.nlat_in)]
            pole_lists = [[], []]
            for t in range(self.nlat_out):
                for r, w in ((int(la[t]), 1.0 - float(lw[t])), (int(lb[t]), float(lw[t]))):
                    (lat_lists[r] if r >= 0 else pole_lists[-r - 1]).append((t, w))
            ll, lr, pw = self.lon_l.cpu().numpy(), self.lon_r.cpu().numpy(), self.lon_w.cpu().numpy()
            lon_lists = [[] for _ in range(self.nlon_in)]
            for p_ in range(self.nlon_out):
                lon_lists[int(ll[p_])].append((p_, 1.0 - float(pw[p_])))
                lon_lists[int(lr[p_])].append((p_, float(pw[p_])))

            def csr(lists):
                off = np.zeros(len(lists) + 1, np.int32)
                off[1:] = np.cumsum([len(l) for l in lists])
                idx = np.array([e[0] for l in lists for e in l] + [0], np.int32)
                wts = np.array([e[1] for l in lists for e in l] + [0.0], np.float32)
                return [torch.from_numpy(a_).to(device) for a_ in (off, idx, wts)]
            self._inv = csr(lat_lists) + csr(lon_lists) + csr(pole_lists)
            self._ikey = key
        return self._inv

    def _launch(self, x, adjoint):
        lead = x.shape[:-2]
        planes = int(np.prod(lead)) if len(lead) else 1
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        if not adjoint:
            y = torch.empty((*lead, self.nlat_out, self.nlon_out), dtype=x.dtype, device=x.device)
            with ops._timed("resample_fwd", nbytes=float(x.element_size()) * (x.numel() + y.numel())):
                check(lib().mk_resample_fwd(ptr(x), ptr(y), dtype_code(x), ptr(self.lat_a), ptr(self.lat_b), ptr(self.lat_w),
                                            ptr(self.lon_l), ptr(self.lon_r), ptr(self.lon_w), planes, self.nlat_in,
                                            self.nlon_in, self.nlat_out, self.nlon_out, stream()), "mk_resample_fwd")
            return y
        inv = self._inverse_stencils(x.device)
        gx = torch.empty((*lead, self.nlat_in, self.nlon_in), dtype=x.dtype, device=x.device)
        with ops._timed("resample_bwd", nbytes=float(x.element_size()) * (x.numel() + gx.numel())):
            check(lib().mk_resample_bwd(ptr(x), ptr(gx), dtype_code(x), *[ptr(a_) for a_ in inv], planes, self.nlat_in,
                                        self.nlon_in, self.nlat_out, self.nlon_out, stream()), "mk_resample_bwd")
        return gx

    @device_guard
    def forward(self, x):
        if self.skip_resampling:
            return x
        return ResampleFn.apply(x, self)


class DistributedResampleS2(ResampleS2):
    """``thd.DistributedResampleS2``: local ``(B, C, nlat_in_loc, nlon_in_loc) -> (B, C, nlat_out_loc, nlon_out_loc)``.  Two
    all-to-alls (azimuth, then polar) trade the spatial split for a channel split, the serial HIP kernel interpolates
    whole planes of this rank's channels, two all-to-alls restore the spatial split on the output grid."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from . import distributed as thd
        if not thd.is_initialized():
            raise RuntimeError("makani_amd.distributed.init(polar_group, azimuth_group) has not been called")
        self.comm_size_polar, self.comm_size_azimuth = thd.polar_group_size(), thd.azimuth_group_size()
        self.lat_in_shapes = thd.compute_split_shapes(self.nlat_in, self.comm_size_polar)
        self.lon_in_shapes = thd.compute_split_shapes(self.nlon_in, self.comm_size_azimuth)
        self.lat_out_shapes = thd.compute_split_shapes(self.nlat_out, self.comm_size_polar)
        self.lon_out_shapes = thd.compute_split_shapes(self.nlon_out, self.comm_size_azimuth)

    @device_guard
    def forward(self, x):
        from . import distributed as thd
        if self.skip_resampling:
            return x
        lead = x.shape[:-3]
        x4 = x.reshape(-1, *x.shape[-3:]) if x.dim() != 4 else x
        Cc = x4.shape[1]
        ca = thd.compute_split_shapes(Cc, self.comm_size_azimuth)
        if self.comm_size_azimuth > 1:
            x4 = thd.transpose(x4, 1, ca, 3, self.lon_in_shapes, thd.azimuth_group())
        cp = thd.compute_split_shapes(x4.shape[1], self.comm_size_polar)
        if self.comm_size_polar > 1:
            x4 = thd.transpose(x4, 1, cp, 2, self.lat_in_shapes, thd.polar_group())
        y = ResampleFn.apply(x4.contiguous(), self)
        if self.comm_size_polar > 1:
            y = thd.transpose(y, 2, self.lat_out_shapes, 1, cp, thd.polar_group())
        if self.comm_size_azimuth > 1:
            y = thd.transpose(y, 3, self.lon_out_shapes, 1, ca, thd.azimuth_group())
        return y.reshape(*lead, *y.shape[-3:]) if x.dim() != 4 else y
