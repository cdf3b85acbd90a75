"""Host-side launchers and autograd Functions over the internal spectral layouts.

Layouts (all fp32, see include/makani_amd.h):
  F-layout  (M, nlat, 2, R) longitude spectrum, k-major: rows (b, c) contiguous    R = B * Cp
  S-layout  (L, M, 2, R)    spherical harmonic coefficients, channel contiguous
  W-layout  (L, 2, Cip, Cop) dhconv weights
Entries of an S tensor with l < m are NEVER read by any kernel (P_l^m = 0 there) and may hold
garbage; only the conversion to complex64 at the API boundary materialises them as zeros.

Everything here launches on ``torch.cuda.current_stream()`` through the C ABI — there is no
CPU implementation behind these functions.
"""
import ctypes as C
import os
import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import MkGemm, check, dtype_code, lib, ptr, stream
from . import legendre as _leg


def round4(n: int) -> int:
    return (n + 3) // 4 * 4


# Arithmetic of the fp32 spectral GEMMs (Legendre, dhconv):
#   "x6"   fp32 operands split into 3 bf16 limbs, 6 bf16 MFMAs per product, fp32 accumulate:
#          fp32 round-off class (rel-L2 1.8e-7 vs fp64; the exact-fp32 MFMA gives 2.8e-7), 2.7x the fp32 MFMA rate
#   "fp32" exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
#   "x3"   2 limbs, 3 MFMAs: rel-L2 ~4e-6 (16 mantissa bits per factor), 5.3x
#   "auto" (default) follows torch's own switch for fp32 matrix products, exactly as the reference's path does on its GPUs:
#          ``torch.backends.cuda.matmul.allow_tf32`` False (torch's default; what the reference's TESTS set, tests/testutils.py:
#          disable_tf32) -> "x6"; True (what the reference's TRAINING entry points set: makani/train.py:87-88, train_stochastic.py:
#          102-103, inference.py:162-163 — its fp32 einsums of the SHT and of the spectral contraction then run in TF32,
#          10 mantissa bits) -> "x3", which is still 64x more accurate than TF32.  gfx950 has no TF32 mode; the two-limb split
#          is this hardware's counterpart of that flag.
GEMM_MODE = os.environ.get("MAKANI_AMD_GEMM", "auto")
if GEMM_MODE not in ("auto", "x6", "x3", "fp32"):
    raise ValueError(f"MAKANI_AMD_GEMM={GEMM_MODE!r}: expected auto, x6, x3 or fp32")


def gemm_mode() -> str:
    """the arithmetic the spectral GEMMs run in right now ("x6" / "x3" / "fp32"): GEMM_MODE, with "auto" resolved through
    torch's fp32-matmul switch"""
    if GEMM_MODE != "auto":
        return GEMM_MODE
    try:
        return "x3" if torch.backends.cuda.matmul.allow_tf32 else "x6"
    except Exception:                       # (a torch build without the attribute: exact arithmetic)
        return "x6"


# Kernel generation of the split engine: "2" = the ping-pong kernels of csrc/xgemm2.hip (512-thread workgroups, double-buffered
# limb images, pre-split Legendre matrices) wherever they apply; the real kernel of csrc/xgemm.hip serves the rest (operands
# that are both split on the fly: the fp32 channel GEMMs of the parity runs).
GEMM_GEN = "2"          # (module attribute, not an environment switch: tools may set "1" to time the first-generation real kernel)


def _run_gemm(g, cplx, what, mode=None, a_limbs=None, band=None, ssq=None):
    """``a_limbs``: the constant A operand of a real GEMM already split into bf16 limb planes (``limb_planes``);
    ``band`` = (lo, hi, mode): its numerical band per batch (``polar_band``); ``ssq``: buffer for the per-workgroup sums of
    squares of the result (``grad_ssq_buffer``: only the second-generation complex engine forms them)."""
    mode = mode or gemm_mode()
    L = lib()
    if mode == "fp32":
        assert ssq is None
        rc = (L.mk_cgemm_batched if cplx else L.mk_sgemm_batched)(C.byref(g), stream())
    else:
        limbs = 3 if mode == "x6" else 2
        if GEMM_GEN == "2" and cplx:
            if ssq is not None:
                rc = L.mk_cgemm_split2_batched_ssq(C.byref(g), limbs, ptr(ssq), stream())
            else:
                rc = L.mk_cgemm_split2_batched(C.byref(g), limbs, stream())
        elif GEMM_GEN == "2" and a_limbs is not None:
            pl = a_limbs
            lo, hi, bm = band if band is not None else (None, None, 0)
            rc = L.mk_sgemm_presplit_batched(C.byref(g), ptr(pl), pl.stride(0), pl.stride(1), pl.stride(2), limbs,
                                             ptr(lo), ptr(hi), bm, stream())
        else:
            assert ssq is None
            rc = (L.mk_cgemm_split_batched if cplx else L.mk_sgemm_split_batched)(C.byref(g), limbs, stream())
    check(rc, what)


# --------------------------------------------------------------------------- #
# Sums of squares formed by the kernel that WROTE a gradient (the dhconv weight gradient: 283 MB per layer of the SFNO), for
# the global-norm clipping of FusedAdamW (makani/utils/training/training_helpers.py:123-165) — which would otherwise read every
# gradient once more only to square it (0.37 ms of the 37 ms step).  An entry says: "the tensor that starts at this address, has
# this many fp32 elements and this autograd version was written by one kernel launch whose per-workgroup sums of squares are in
# `part`".  It is used only while all three still match (any torch-level write to the gradient — accumulation into an existing
# .grad, hooks, unscaling, all-reduce through copy_ — bumps the version or changes the address), every weight-gradient call that
# does not form sums drops the entry of its address, the optimizer's step() and the gradient reducers drop all of them.
# MAKANI_AMD_GRAD_SSQ=0: never form them.
# --------------------------------------------------------------------------- #
GRAD_SSQ = os.environ.get("MAKANI_AMD_GRAD_SSQ", "1") != "0"
_GRAD_SSQ = {}


def grad_ssq_register(t: torch.Tensor, part: torch.Tensor):
    if len(_GRAD_SSQ) > 256:            # never consumed (no FusedAdamW in the loop): do not grow with the allocator's addresses
        _GRAD_SSQ.clear()
    r = torch.view_as_real(t) if t.is_complex() else t
    _GRAD_SSQ[r.data_ptr()] = (part, r.numel(), t._version, t.device)


def grad_ssq_drop(t: torch.Tensor = None):
    if t is None:
        _GRAD_SSQ.clear()
    else:
        _GRAD_SSQ.pop(t.data_ptr(), None)


def grad_ssq_lookup(flat: torch.Tensor, version: int):
    """the partial sums of squares of the fp32 gradient whose memory-order view is ``flat``, or None"""
    e = _GRAD_SSQ.get(flat.data_ptr())
    if e is None or e[1] != flat.numel() or e[2] != version or e[3] != flat.device:
        return None
    return e[0]


_LIMB_PLANES = {}      # id(matrix tensor) -> (weak reference to it, version, limb planes); entries die with the matrix


def limb_planes(mat: torch.Tensor) -> torch.Tensor:
    """The three bf16 limbs ``hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)`` (round to nearest even, the
    split the GEMM kernels apply on the fly) of a constant (batch, K, rows) fp32 matrix as one (3, batch, K, rows8)
    bf16 tensor, rows padded with zeros to a multiple of 8.  Computed once per matrix object (the Legendre matrices are
    module buffers that never change; an in-place write bumps ``_version`` and triggers a re-split) and consumed by
    ``mk_sgemm_presplit_batched``.  The planes die with the matrix tensor."""
    hit = _LIMB_PLANES.get(id(mat))
    if hit is not None and hit[0]() is mat and hit[1] == mat._version:
        return hit[2]
    nb, K, rows = mat.shape
    r8 = (rows + 7) // 8 * 8
    pl = torch.zeros((3, nb, K, r8), dtype=torch.bfloat16, device=mat.device)
    r = mat.detach().to(torch.float32).clone()
    for q in range(3):
        h = r.to(torch.bfloat16)
        pl[q, :, :, :rows] = h
        r -= h.to(torch.float32)
    key = id(mat)
    _LIMB_PLANES[key] = (weakref.ref(mat, lambda _r, key=key: _LIMB_PLANES.pop(key, None)), mat._version, pl)
    return pl


# Polar band of the Legendre matrices: P_l^m(theta) vanishes towards the poles like sin^m(theta), so for every order m
# there is a latitude band outside of which max_l |P_l^m| is below BAND_EPS times the largest entry of that order — for
# the benchmark's transforms the band covers 78 % of the (m, latitude) pairs.  Outside it the GEMMs neither read the
# matrix nor the data and write exact zeros.  BAND_EPS = 1e-18: the dropped terms are 11 orders of magnitude below the
# fp32 rounding of the terms that are kept (MAKANI_AMD_BAND_EPS=0 switches the optimisation off).
BAND_EPS = float(os.environ.get("MAKANI_AMD_BAND_EPS", "1e-18"))
_POLAR_BANDS = {}


def polar_band(mat: torch.Tensor, lat_dim: int):
    """(lo, hi, lo_host, hi_host): int32 device vectors (and their host lists) with, per batch entry (order m), the half-open
    range of latitude indices along ``lat_dim`` (1 or 2 of the (batch, ., .) matrix) where the matrix has entries above
    ``BAND_EPS`` * its largest entry; None when the optimisation is off.  Cached per matrix object like ``limb_planes``."""
    if BAND_EPS <= 0.0:
        return None
    key = (id(mat), lat_dim)
    hit = _POLAR_BANDS.get(key)
    if hit is not None and hit[0]() is mat and hit[1] == mat._version:
        return hit[2]
    a = mat.detach().abs().amax(dim=3 - lat_dim)                 # (batch, nlat): largest entry per latitude
    live = a > BAND_EPS * a.amax(dim=1, keepdim=True).clamp_min(1e-300)
    n = a.shape[1]
    idx = torch.arange(n, device=mat.device)
    lo = torch.where(live, idx, n).amin(dim=1)
    hi = torch.where(live, idx + 1, 0).amax(dim=1)
    lo = torch.minimum(lo, hi)                                   # an all-dead order gives the empty range [0, 0)
    out = (lo.to(torch.int32).contiguous(), hi.to(torch.int32).contiguous(), lo.tolist(), hi.tolist())     # device + host copies
    _POLAR_BANDS[key] = (weakref.ref(mat, lambda _r, key=key: _POLAR_BANDS.pop(key, None)), mat._version, out)
    return out


# --------------------------------------------------------------------------- #
# optional per-launch timing (used by bench.py): HIP events on the launch stream
# --------------------------------------------------------------------------- #
class LaunchProfiler:
    """Records one (start, end) HIP-event pair per C-ABI launch on the stream the kernel is
    launched on (torch's current stream) together with its algorithmic work."""

    def __init__(self):
        self.enabled = False
        self.only = None           # optional set of launch names to instrument (None = all)
        self.records = []          # (name, start_event, end_event, flops, bytes, executed bf16-MFMA flops)

    def reset(self):
        self.records = []

    def summary(self):
        """name -> dict(launches, ms_total, ms_avg, flops, bytes); call after a device synchronize."""
        out = {}
        for name, e0, e1, fl, by, mf in self.records:
            d = out.setdefault(name, dict(launches=0, ms_total=0.0, flops=0.0, bytes=0.0, mfma_flops=0.0))
            d["launches"] += 1
            d["ms_total"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
            d["mfma_flops"] += mf
        for d in out.values():
            d["ms_avg"] = d["ms_total"] / d["launches"]
        return out


PROFILER = LaunchProfiler()


class _timed:
    """``mfma_flops``: the bf16 matrix-core flops the launch actually EXECUTES (limb products and tile padding included,
    structurally-zero tiles excluded) — what MFMA utilisation is measured with; ``flops`` is the dense formulation."""

    def __init__(self, name, flops=0.0, nbytes=0.0, mfma_flops=0.0):
        self.name, self.flops, self.nbytes, self.mfma_flops = name, flops, nbytes, mfma_flops

    def __enter__(self):
        self.on = PROFILER.enabled and (PROFILER.only is None or self.name in PROFILER.only)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            mf = self.mfma_flops() if callable(self.mfma_flops) else self.mfma_flops       # evaluated only when profiling
            PROFILER.records.append((self.name, self.e0, self.e1, self.flops, self.nbytes, mf))
        return False


# --------------------------------------------------------------------------- #
# FFT plans
# --------------------------------------------------------------------------- #
@dataclass
class FftPlan:
    nlon: int
    radix: object          # ctypes int array
    nradix: int
    twiddle: torch.Tensor  # (nlon, 2) float32 on the device


_PLANS = {}


def fft_plan(nlon: int, device) -> FftPlan:
    key = (nlon, str(device))
    p = _PLANS.get(key)
    if p is None:
        radix = _leg.factorize_half(nlon)
        arr = (C.c_int * len(radix))(*radix)
        tw = torch.from_numpy(_leg.twiddle_table(nlon)).to(device)
        p = FftPlan(nlon, arr, len(radix), tw)
        _PLANS[key] = p
    return p


def rfft_rows(x: torch.Tensor, mmax: int, Cp: int, w) -> torch.Tensor:
    """x (B, C, nlat, nlon) f32|bf16 -> F (mmax, nlat, 2, B*Cp);  X_m = w_m sum_n x_n e^{-i m n 2pi/N}."""
    B, Cc, nlat, nlon = x.shape
    x = x.contiguous()
    plan = fft_plan(nlon, x.device)
    F = torch.empty((mmax, nlat, 2, B * Cp), dtype=torch.float32, device=x.device)
    nbytes = B * Cc * nlat * (nlon * x.element_size() + mmax * 8)
    with _timed(f"rfft_{nlon}", nbytes=nbytes):
        check(lib().mk_rfft_rows(ptr(x), dtype_code(x), ptr(F), ptr(plan.twiddle), plan.radix, plan.nradix, B, Cc, Cp,
                                 nlat, nlon, mmax, w[0], w[1], w[2], stream()), "mk_rfft_rows")
    return F


def irfft_rows(F: torch.Tensor, B: int, Cc: int, nlon: int, out_dtype, w) -> torch.Tensor:
    """F (mmax, nlat, 2, B*Cp) -> x (B, C, nlat, nlon);  x_n = sum_m w_m (Re X_m cos - Im X_m sin)."""
    mmax, nlat, _, R = F.shape
    Cp = R // B
    plan = fft_plan(nlon, F.device)
    x = torch.empty((B, Cc, nlat, nlon), dtype=out_dtype, device=F.device)
    nbytes = B * Cc * nlat * (nlon * x.element_size() + mmax * 8)
    with _timed(f"irfft_{nlon}", nbytes=nbytes):
        check(lib().mk_irfft_rows(ptr(F), ptr(x), dtype_code(x), ptr(plan.twiddle), plan.radix, plan.nradix, B, Cc,
                                  Cp, nlat, nlon, mmax, w[0], w[1], w[2], stream()), "mk_irfft_rows")
    return x


def fft_seg_desc(m_shapes, r_shapes, base, xseg=1, x_stride=0, x_nlat=0):
    """MkFftSeg (include/makani_amd.h): per-peer F slabs (m ranges x row ranges, ``base[jw][ih]`` float offsets) and x rows
    cut into ``xseg`` pieces ``x_stride`` elements apart"""
    sg = _lib.MkFftSeg()
    sg.nw, sg.nh = len(m_shapes), len(r_shapes)
    if sg.nw > _lib.MK_FFT_SEG_MAX or sg.nh > _lib.MK_FFT_SEG_MAX:
        raise ValueError(f"at most {_lib.MK_FFT_SEG_MAX} peers per direction")
    off = 0
    for j, n in enumerate(m_shapes):
        sg.m_off[j] = off
        off += n
    for j in range(len(m_shapes), _lib.MK_FFT_SEG_MAX + 1):
        sg.m_off[j] = off
    off = 0
    for i, n in enumerate(r_shapes):
        sg.r_off[i] = off
        off += n
    for i in range(len(r_shapes), _lib.MK_FFT_SEG_MAX + 1):
        sg.r_off[i] = off
    for j in range(sg.nw):
        for i in range(sg.nh):
            sg.base[j][i] = int(base[j][i])
    sg.xseg, sg.x_stride, sg.x_nlat = int(xseg), int(x_stride), int(x_nlat)
    return sg


def fft_seg_supported(nlon: int) -> bool:
    return bool(lib().mk_fft_seg_supported(int(nlon)))


def rfft_rows_seg(x_ptr_tensor: torch.Tensor, x_off: int, F: torch.Tensor, planes: int, nlat: int, nlon: int, mmax: int, w, sg):
    """truncated rFFT of ``planes`` rows x ``nlat`` latitudes read from the pieces of ``x`` (element offset ``x_off`` = first
    latitude of the call) into the per-peer slabs of the flat fp32 buffer ``F`` (csrc/fft_fast.hip, SEG kernels)"""
    plan = fft_plan(nlon, F.device)
    xa = C.c_void_p(x_ptr_tensor.data_ptr() + x_off * x_ptr_tensor.element_size())
    nbytes = planes * nlat * (nlon * x_ptr_tensor.element_size() + mmax * 8)
    with _timed(f"rfft_{nlon}", nbytes=nbytes):
        check(lib().mk_rfft_rows_seg(xa, dtype_code(x_ptr_tensor), ptr(F), ptr(plan.twiddle), planes, nlat, nlon, mmax,
                                     w[0], w[1], w[2], C.byref(sg), stream()), "mk_rfft_rows_seg")


def irfft_rows_seg(F: torch.Tensor, x_ptr_tensor: torch.Tensor, x_off: int, planes: int, nlat: int, nlon: int, mmax: int, w, sg):
    """the inverse: per-peer slabs of ``F`` -> pieces of ``x`` (zero-padded spectrum, Im of m = 0 dropped)"""
    plan = fft_plan(nlon, F.device)
    xa = C.c_void_p(x_ptr_tensor.data_ptr() + x_off * x_ptr_tensor.element_size())
    nbytes = planes * nlat * (nlon * x_ptr_tensor.element_size() + mmax * 8)
    with _timed(f"irfft_{nlon}", nbytes=nbytes):
        check(lib().mk_irfft_rows_seg(ptr(F), xa, dtype_code(x_ptr_tensor), ptr(plan.twiddle), planes, nlat, nlon, mmax,
                                      w[0], w[1], w[2], C.byref(sg), stream()), "mk_irfft_rows_seg")


# --------------------------------------------------------------------------- #
# Legendre GEMMs (real, batched over m).  Both operands are row-contiguous ("k-major"):
#   analysis   S[l][j]   = sum_k  matT[m][k][l] * F[m][k][j]          j = (ri, row)
#   synthesis  F[m][k][j] = sum_l  mat[m][l][k]  * S[l][m][j]
# mat  = (M, L, Kp)  natural torch-harmonics layout, latitude padded to 4
# matT = (M, nlat, Lp) its transpose, degree padded to 4
# --------------------------------------------------------------------------- #
def _gemm(**kw) -> MkGemm:
    g = MkGemm()
    for f, _ in MkGemm._fields_:
        setattr(g, f, 0)
    g.inner = 1
    g.c_col = 1
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def _limb_products() -> int:
    return {"x6": 6, "x3": 3, "fp32": 16}[gemm_mode()]       # fp32 MFMA = 1/16 of the bf16 rate: counted as 16 bf16-equivalents


def _up(n, q):
    return (n + q - 1) // q * q


def _exec_rows_ge(nrows, nbatch, off=0, g=32):
    """sum over batches b of the rows a kernel that skips dead row tiles of g rows executes (rows >= b + off are live)"""
    return sum(max(0, _up(nrows, g) - min(_up(nrows, g), max(0, b + off) // g * g)) for b in range(nbatch))


def _exec_le(n, nbatch, off=0, g=32):
    """sum over batches b of min(n, b + off + 1) rounded up to the tile granularity g (rows / k <= b + off are live)"""
    return sum(min(_up(n, g), _up(max(0, min(n, b + off + 1)), g)) for b in range(nbatch))


def _exec_band(L, M, m_off, nlat, band, lat_gran, rows_tri):
    """sum over orders m of (degrees l executed) x (latitudes executed): the triangle skips l < m in steps of 32 (rows of
    the analysis) or 16 (k-steps of the synthesis), the polar band clips the latitudes in steps of ``lat_gran``"""
    tot = 0
    for b in range(M):
        g = 32 if rows_tri else 16
        ldeg = max(0, _up(L, g) - min(_up(L, g), max(0, b + m_off) // g * g))
        if band is None:
            nl = _up(nlat, lat_gran)
        else:
            lo, hi = band[2][b], band[3][b]
            nl = max(0, _up(hi, lat_gran) - lo // lat_gran * lat_gran) if hi > lo else 0
        tot += ldeg * nl
    return tot


def _presplit_ok() -> bool:
    return GEMM_GEN == "2" and gemm_mode() != "fp32"


def legendre_analysis(F: torch.Tensor, matT: torch.Tensor, L: int, m_off: int = 0, lat_major: bool = False,
                      blocks: bool = False) -> torch.Tensor:
    """S[l][m][ri][row] = sum_k matT[m][k][l] F[m][k][ri][row]      (rows l >= m only).
    ``lat_major``: F is (nlat, M, 2, R) — latitude outermost, the layout in which the slabs the ranks of a polar group send
    concatenate by landing next to each other (makani_amd/dist_pipeline.py); only the operand strides change.
    ``blocks`` (with ``lat_major``): F is (nb, nlat, M, 2, R) — nb column blocks in separate buffers (the plane blocks of the
    h x w distributed transform) — transformed by ONE launch (the block index is the GEMM's inner batch index: same matrix,
    band and triangle per order m) into S (L, nb, M, 2, R), degree outermost: the l-slabs the polar exchange sends are
    contiguous over all blocks."""
    if blocks:
        assert lat_major
        nb, nlat, M, _, R = F.shape
    elif lat_major:
        nlat, M, _, R = F.shape
        nb = 1
    else:
        M, nlat, _, R = F.shape
        nb = 1
    Mm, nk, Lp = matT.shape
    assert Mm == M and nk == nlat and Lp >= L and F.is_contiguous() and matT.is_contiguous()
    S = torch.empty((L, nb, M, 2, R) if blocks else (L, M, 2, R), dtype=torch.float32, device=F.device)
    g = _gemm(A=matT.data_ptr(), B=F.data_ptr(), C=S.data_ptr(),
              a_batch=nlat * Lp, a_row=1, a_k=Lp,
              b_batch=2 * R if lat_major else nlat * 2 * R, b_inner=nlat * M * 2 * R, b_col=1, b_k=M * 2 * R if lat_major else 2 * R,
              c_batch=2 * R, c_inner=M * 2 * R, c_row=nb * M * 2 * R,
              M=L, N=2 * R, K=nlat, batch=M * nb, inner=nb, tri_mode=_lib.TRI_ROW_GE, tri_off=m_off)
    # dense-formulation work: 2 * (2R) * nlat * L * M flops
    pre = _presplit_ok()
    b = polar_band(matT, 1) if pre else None                     # matT = (m, latitude, l): the band clips the k-loop
    with _timed(f"legendre_analysis_k{nlat}", flops=2.0 * nb * 2 * R * nlat * L * M,
                nbytes=4.0 * (nb * 2 * R * nlat * M + nb * 2 * R * L * M + M * L * nlat),
                mfma_flops=lambda: 2.0 * nb * _limb_products() * _up(2 * R, 32) * _exec_band(L, M, m_off, nlat, b, 16, rows_tri=True)):
        _run_gemm(g, False, "legendre_analysis", a_limbs=limb_planes(matT) if pre else None,
                  band=(b[0], b[1], 1) if b is not None else None)
    return S


def legendre_synthesis(S: torch.Tensor, mat: torch.Tensor, nlat: int, m_off: int = 0, lat_major: bool = False,
                       blocks: bool = False) -> torch.Tensor:
    """F[m][k][ri][row] = sum_{l >= m} mat[m][l][k] S[l][m][ri][row].  ``lat_major``: F is written as (nlat, M, 2, R).
    ``blocks`` (with ``lat_major``): S is (L, nb, M, 2, R), F is written as (nb, nlat, M, 2, R) by one launch (see
    ``legendre_analysis``)."""
    if blocks:
        assert lat_major
        L, nb, M, _, R = S.shape
    else:
        L, M, _, R = S.shape
        nb = 1
    Mm, Lm, kp = mat.shape
    assert Mm == M and Lm == L and kp >= nlat and S.is_contiguous() and mat.is_contiguous()
    F = torch.empty((nb, nlat, M, 2, R) if blocks else ((nlat, M, 2, R) if lat_major else (M, nlat, 2, R)),
                    dtype=torch.float32, device=S.device)
    g = _gemm(A=mat.data_ptr(), B=S.data_ptr(), C=F.data_ptr(),
              a_batch=L * kp, a_row=1, a_k=kp,
              b_batch=2 * R, b_inner=M * 2 * R, b_col=1, b_k=nb * M * 2 * R,
              c_batch=2 * R if lat_major else nlat * 2 * R, c_inner=nlat * M * 2 * R, c_row=M * 2 * R if lat_major else 2 * R,
              M=nlat, N=2 * R, K=L, batch=M * nb, inner=nb, tri_mode=_lib.TRI_K_GE, tri_off=m_off)
    pre = _presplit_ok()
    b = polar_band(mat, 2) if pre else None                      # mat = (m, l, latitude): the band clips the output rows
    with _timed(f"legendre_synthesis_k{nlat}", flops=2.0 * nb * 2 * R * nlat * L * M,
                nbytes=4.0 * (nb * 2 * R * nlat * M + nb * 2 * R * L * M + M * L * nlat),
                mfma_flops=lambda: 2.0 * nb * _limb_products() * _up(2 * R, 32) * _exec_band(L, M, m_off, nlat, b, 32, rows_tri=False)):
        _run_gemm(g, False, "legendre_synthesis", a_limbs=limb_planes(mat) if pre else None,
                  band=(b[0], b[1], 2) if b is not None else None)
    return F


# --------------------------------------------------------------------------- #
# dhconv (complex, batched over l)
# --------------------------------------------------------------------------- #
def native_w_empty(cin: int, cout: int, L: int, device=None) -> torch.Tensor:
    """uninitialised complex64 (1, Cin, Cout, L) tensor whose MEMORY order is [l][i][o] (re, im interleaved): the order
    the dhconv GEMMs read their weight in.  ``SpectralConv`` allocates its dhconv parameter like this, autograd hands
    the weight gradient back in the same strides, and no layout kernel runs per step (round 1 re-laid 2.26 GB out
    twice per step).  Shape, values, ``state_dict`` and ``load_state_dict`` are those of the reference parameter."""
    return torch.empty((L, cin, cout), dtype=torch.complex64, device=device).permute(1, 2, 0).unsqueeze(0)


def is_native_w(weight: torch.Tensor) -> bool:
    if weight.dim() != 4 or weight.shape[0] != 1 or weight.dtype != torch.complex64:
        return False
    _, cin, cout, L = weight.shape
    return (cin % 4 == 0 and cout % 4 == 0 and (cout == 1 or weight.stride(2) == 1) and (cin == 1 or weight.stride(1) == cout)
            and (L == 1 or weight.stride(3) == cin * cout) and weight.data_ptr() % 16 == 0 and gemm_mode() != "fp32")


def _w_operand(W, transposed):
    """B-operand fields of the dhconv GEMMs for W = planar (L, 2, Cip, Cop) fp32 or a native-order complex64 weight"""
    if W.is_complex():
        _, cip, cop, L = W.shape
        d = dict(b_batch=2 * cip * cop, b_inner=0, b_im=1, b_col=2 * cop if transposed else 2, b_k=2 if transposed else 2 * cop)
    else:
        L, _, cip, cop = W.shape
        d = dict(b_batch=2 * cip * cop, b_inner=0, b_im=cip * cop, b_col=cop if transposed else 1, b_k=1 if transposed else cop)
    return L, cip, cop, d


def weight_to_wlayout(weight: torch.Tensor) -> torch.Tensor:
    """complex64 (1, Cin, Cout, L) -> W (L, 2, Cip, Cop), zero padded."""
    G, cin, cout, L = weight.shape
    assert G == 1 and weight.dtype == torch.complex64
    cip, cop = round4(cin), round4(cout)
    wr = torch.view_as_real(weight.contiguous())
    alloc = torch.zeros if (cip != cin or cop != cout) else torch.empty
    W = alloc((L, 2, cip, cop), dtype=torch.float32, device=weight.device)
    check(lib().mk_weight_to_wlayout(ptr(wr), ptr(W), cin, cout, cip, cop, L, stream()), "weight_to_wlayout")
    return W


def wlayout_to_weight_grad(gW: torch.Tensor, cin: int, cout: int) -> torch.Tensor:
    L, _, cip, cop = gW.shape
    out = torch.empty((1, cin, cout, L, 2), dtype=torch.float32, device=gW.device)
    check(lib().mk_wlayout_to_weight_grad(ptr(gW), ptr(out), cin, cout, cip, cop, L, stream()), "wlayout_to_weight_grad")
    return torch.view_as_complex(out)


def dhconv_fwd(S: torch.Tensor, W: torch.Tensor, B: int, cin: int, tri_off: int = 0, out=None, grp=None) -> torch.Tensor:
    """T[l][m][ri][b][o] = sum_i S[l][m][.][b][i] * W[l][.][i][o]   (complex; rows m <= l only).
    ``grp = (first input channel, first output channel, x_ld, y_ld)`` runs one channel group of a grouped operator:
    S and ``out`` then hold all groups (x_ld / y_ld channels per batch entry), W this group's matrices."""
    L, M, _, R = S.shape
    Lw, cip, cop, bdesc = _w_operand(W, False)
    a_off, c_off, xld, yld = grp if grp is not None else (0, 0, cip, cop)
    assert Lw == L and R == B * xld
    Ro = B * yld
    T = out if out is not None else torch.empty((L, M, 2, Ro), dtype=torch.float32, device=S.device)
    g = _gemm(A=S.data_ptr() + 4 * a_off, B=W.data_ptr(), C=T.data_ptr() + 4 * c_off,
              a_batch=M * 2 * R, a_inner=xld, a_row=2 * R, a_k=1, a_im=R, **bdesc,
              c_batch=M * 2 * Ro, c_inner=yld, c_row=2 * Ro, c_im=Ro,
              M=M, N=cop, K=cin, batch=L * B, inner=B, tri_mode=_lib.TRI_ROW_LE, tri_off=tri_off)
    # dense-formulation work: 8 * B * Cin * Cout * L * M flops (complex MAC = 8 real flops)
    with _timed("dhconv_fwd", flops=8.0 * B * cin * cop * L * M,
                nbytes=4.0 * (2 * B * cip * L * M + 2 * B * cop * L * M + 2 * cip * cop * L),
                mfma_flops=lambda: 8.0 * _limb_products() * B * _up(cin, 16) * _up(cop, 32) * _exec_le(M, L, tri_off)):
        _run_gemm(g, True, "dhconv_fwd")
    return T


def dhconv_dgrad(gT: torch.Tensor, W: torch.Tensor, B: int, cin: int, cout: int, tri_off: int = 0, out=None, grp=None) -> torch.Tensor:
    """gS[l][m][b][i] = sum_o gT[l][m][b][o] * conj(W[l][i][o])."""
    L, M, _, Ro = gT.shape
    _, cip, cop, bdesc = _w_operand(W, True)
    a_off, c_off, xld, yld = grp if grp is not None else (0, 0, cip, cop)      # (input ch., output ch., x_ld, y_ld)
    assert Ro == B * yld
    R = B * xld
    gS = out if out is not None else torch.empty((L, M, 2, R), dtype=torch.float32, device=gT.device)
    g = _gemm(A=gT.data_ptr() + 4 * c_off, B=W.data_ptr(), C=gS.data_ptr() + 4 * a_off,
              a_batch=M * 2 * Ro, a_inner=yld, a_row=2 * Ro, a_k=1, a_im=Ro, **bdesc,
              c_batch=M * 2 * R, c_inner=xld, c_row=2 * R, c_im=R,
              M=M, N=cin, K=cout, batch=L * B, inner=B, tri_mode=_lib.TRI_ROW_LE, tri_off=tri_off, conj_b=1)
    with _timed("dhconv_dgrad", flops=8.0 * B * cin * cout * L * M,
                nbytes=4.0 * (2 * B * cip * L * M + 2 * B * cop * L * M + 2 * cip * cop * L),
                mfma_flops=lambda: 8.0 * _limb_products() * B * _up(cout, 16) * _up(cin, 32) * _exec_le(M, L, tri_off)):
        _run_gemm(g, True, "dhconv_dgrad")
    return gS


def dhconv_wgrad(S: torch.Tensor, gT: torch.Tensor, B: int, tri_off: int = 0, grp=None, native=False) -> torch.Tensor:
    """gW[l][i][o] = sum_{b, m <= l} conj(S[l][m][b][i]) * gT[l][m][b][o].
    ``grp = (first input channel, first output channel, group inputs, group outputs)`` for one group of a grouped operator.
    ``native``: the result is the complex64 (1, Cin, Cout, L) gradient in the memory order of ``native_w_empty``
    (written interleaved by the GEMM epilogue) instead of the planar (L, 2, Cip, Cop) W-layout."""
    L, M, _, R = S.shape
    Ro = gT.shape[-1]
    xld, yld = R // B, Ro // B
    a_off, c_off, cip, cop = grp if grp is not None else (0, 0, xld, yld)
    if native:
        gW = native_w_empty(cip, cop, L, S.device)
        cdesc = dict(c_batch=2 * cip * cop, c_row=2 * cop, c_col=2, c_im=1)
    else:
        gW = torch.empty((L, 2, cip, cop), dtype=torch.float32, device=S.device)
        cdesc = dict(c_batch=2 * cip * cop, c_row=cop, c_im=cip * cop)
    # the whole gradient from ONE launch (native order, one sample, one group): the launch also sums its squares
    want_ssq = GRAD_SSQ and native and B == 1 and grp is None and GEMM_GEN == "2" and gemm_mode() != "fp32"
    grad_ssq_drop(gW)
    for b in range(B):
        g = _gemm(A=S.data_ptr() + 4 * (b * xld + a_off), B=gT.data_ptr() + 4 * (b * yld + c_off), C=gW.data_ptr(),
                  a_batch=M * 2 * R, a_row=1, a_k=2 * R, a_im=R,
                  b_batch=M * 2 * Ro, b_col=1, b_k=2 * Ro, b_im=Ro, **cdesc,
                  M=cip, N=cop, K=M, batch=L, inner=1, tri_mode=_lib.TRI_K_LE, tri_off=tri_off, conj_a=1,
                  beta=1 if b > 0 else 0)
        part = None
        if want_ssq:
            n = lib().mk_cgemm_split2_ssq_count(C.byref(g))
            part = torch.empty((n,), dtype=torch.float32, device=S.device) if n > 0 else None
        with _timed("dhconv_wgrad", flops=8.0 * cip * cop * L * M,
                    nbytes=4.0 * (2 * cip * L * M + 2 * cop * L * M + 2 * cip * cop * L),
                    mfma_flops=lambda: 8.0 * _limb_products() * _up(cip, 32) * _up(cop, 32) * _exec_le(M, L, tri_off, 16)):
            _run_gemm(g, True, "dhconv_wgrad", ssq=part)
        if part is not None:
            grad_ssq_register(gW, part)
    return gW


# --------------------------------------------------------------------------- #
# API-boundary layout changes
# --------------------------------------------------------------------------- #
def s_to_complex(S: torch.Tensor, B: int, Cc: int, l_off: int = 0, m_off: int = 0) -> torch.Tensor:
    L, M, _, R = S.shape
    out = torch.empty((B, Cc, L, M, 2), dtype=torch.float32, device=S.device)
    check(lib().mk_slayout_to_complex(ptr(S), ptr(out), B, Cc, R // B, L, M, l_off, m_off, stream()), "slayout_to_complex")
    return torch.view_as_complex(out)


def complex_to_s(c: torch.Tensor) -> torch.Tensor:
    B, Cc, L, M = c.shape
    Cp = round4(Cc)
    cr = torch.view_as_real(c.contiguous())
    S = torch.empty((L, M, 2, B * Cp), dtype=torch.float32, device=c.device)
    check(lib().mk_complex_to_slayout(ptr(cr), ptr(S), B, Cc, Cp, L, M, stream()), "complex_to_slayout")
    return S


# --------------------------------------------------------------------------- #
# autograd Functions (linear maps with constant matrices: nothing is saved but the matrix)
# --------------------------------------------------------------------------- #
class RfftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mmax, Cp, w):
        ctx.meta = (x.shape, x.dtype, w)
        return rfft_rows(x, mmax, Cp, w)

    @staticmethod
    def backward(ctx, gF):
        (B, Cc, nlat, nlon), dt, w = ctx.meta
        return irfft_rows(gF.contiguous(), B, Cc, nlon, dt, w), None, None, None


class IrfftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F, B, Cc, nlon, out_dtype, w):
        ctx.meta = (F.shape[0], F.shape[3] // B, w)
        return irfft_rows(F, B, Cc, nlon, out_dtype, w)

    @staticmethod
    def backward(ctx, gx):
        mmax, Cp, w = ctx.meta
        return rfft_rows(gx, mmax, Cp, w), None, None, None, None, None


class AnalysisFn(torch.autograd.Function):
    """S = analysis(F) with the (constant) matrix given in both orientations (mat: (M, L, Kp), matT: (M, nlat, Lp))."""

    @staticmethod
    def forward(ctx, F, mat, matT, m_off=0):
        ctx.mat, ctx.m_off, ctx.nlat = mat, m_off, F.shape[1]
        return legendre_analysis(F, matT, mat.shape[1], m_off)

    @staticmethod
    def backward(ctx, gS):
        return legendre_synthesis(gS.contiguous(), ctx.mat, ctx.nlat, ctx.m_off), None, None, None


class SynthesisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, mat, matT, nlat, m_off=0):
        ctx.matT, ctx.L, ctx.m_off = matT, mat.shape[1], m_off
        return legendre_synthesis(S, mat, nlat, m_off)

    @staticmethod
    def backward(ctx, gF):
        return legendre_analysis(gF.contiguous(), ctx.matT, ctx.L, ctx.m_off), None, None, None, None


class DhconvFn(torch.autograd.Function):
    """y[b,o,l,m] = sum_i x[b,i,l,m] w[i,o,l] on S-layout tensors
    (``_contract_lwise``, makani/models/common/contractions.py:23-24)."""

    @staticmethod
    def forward(ctx, S, weight, B, tri_off=0):
        """tri_off = (first l of this shard) - (first m of this shard); 0 when not sharded."""
        _, cin, cout, _ = weight.shape
        native = is_native_w(weight)          # the parameter is already in the GEMM's order: used (and saved) in place
        W = weight.detach() if native else weight_to_wlayout(weight)
        ctx.save_for_backward(S, W)
        ctx.meta = (B, cin, cout, tri_off, native)
        return dhconv_fwd(S, W, B, cin, tri_off)

    @staticmethod
    def backward(ctx, gT):
        S, W = ctx.saved_tensors
        B, cin, cout, tri_off, native = ctx.meta
        gT = gT.contiguous()
        gS = dhconv_dgrad(gT, W, B, cin, cout, tri_off) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            if native and S.shape[-1] == B * cin and gT.shape[-1] == B * cout:
                gw = dhconv_wgrad(S, gT, B, tri_off, native=True)
            else:
                gw = wlayout_to_weight_grad(dhconv_wgrad(S, gT, B, tri_off), cin, cout)
        return gS, gw, None, None


def _addr(t, off_floats=0):
    """device address of element ``off_floats`` of an fp32 GPU tensor (channel-group slices of the S-layout)"""
    if not t.is_cuda:
        raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
    return C.c_void_p(t.data_ptr() + 4 * off_floats)


class GroupedDhconvFn(torch.autograd.Function):
    """``_contract_lwise`` with num_groups > 1: ``einsum("bgixy,giox->bgoxy")`` as G runs of the same MFMA engine on
    channel slices of the S-layout (group sizes must be multiples of 4: the slices start on 16-byte boundaries)."""

    @staticmethod
    def forward(ctx, S, weight, B, tri_off=0):
        G, cgi, cgo, L = weight.shape
        cin, cout = G * cgi, G * cgo
        Ws = [weight_to_wlayout(weight[g:g + 1]) for g in range(G)]
        T = torch.empty((L, S.shape[1], 2, B * cout), dtype=torch.float32, device=S.device)
        for g in range(G):
            dhconv_fwd(S, Ws[g], B, cgi, tri_off, out=T, grp=(g * cgi, g * cgo, cin, cout))
        ctx.save_for_backward(S, *Ws)
        ctx.meta = (B, G, cgi, cgo, tri_off)
        return T

    @staticmethod
    def backward(ctx, gT):
        S, *Ws = ctx.saved_tensors
        B, G, cgi, cgo, tri_off = ctx.meta
        cin, cout = G * cgi, G * cgo
        gT = gT.contiguous()
        gS = gw = None
        if ctx.needs_input_grad[0]:
            gS = torch.empty_like(S)
            for g in range(G):
                dhconv_dgrad(gT, Ws[g], B, cgi, cgo, tri_off, out=gS, grp=(g * cgi, g * cgo, cin, cout))
        if ctx.needs_input_grad[1]:
            gw = torch.cat([wlayout_to_weight_grad(dhconv_wgrad(S, gT, B, tri_off, grp=(g * cgi, g * cgo, cgi, cgo)), cgi, cgo)
                            for g in range(G)], dim=0)
        return gS, gw, None, None


class WeightToSFn(torch.autograd.Function):
    """complex64 weight (C, L, Mw) -> S-layout (L, Mw, 2, round4(C)) and back for its gradient (no triangle mask: the
    contraction kernels already write exact zeros at dead positions, in shard-local or global indexing alike)"""

    @staticmethod
    def forward(ctx, w3):
        ctx.C = w3.shape[0]
        return complex_to_s(w3.unsqueeze(0))

    @staticmethod
    def backward(ctx, gS):
        L, Mw = gS.shape[:2]
        return s_to_complex(gS.contiguous(), 1, ctx.C, l_off=Mw, m_off=0)[0]


class SepContractFn(torch.autograd.Function):
    """``_contract_sep_lmwise`` / ``_contract_sep_lwise`` (makani/models/common/contractions.py:26-31) on the S-layout:
    y[l][m][b][c] = x[l][m][b][c] * w[l][m or 0][c]; one streaming kernel each for y, gx and gw."""

    @staticmethod
    def forward(ctx, S, Ws, B, tri_off=0):
        L, M, _, R = S.shape
        Lw, Mw, _, Cp = Ws.shape
        assert Lw == L and Mw in (1, M) and R == B * Cp
        S = S.contiguous()
        y = torch.empty_like(S)
        with _timed("spec_sep_mul", nbytes=4.0 * (4 * R * L * M + 2 * Cp * L * Mw)):
            check(lib().mk_spec_sep_mul(ptr(S), ptr(Ws), ptr(y), L, M, Mw, B, Cp, tri_off, 0, stream()), "spec_sep_mul")
        ctx.save_for_backward(S, Ws)
        ctx.meta = (B, tri_off)
        return y

    @staticmethod
    def backward(ctx, gy):
        S, Ws = ctx.saved_tensors
        B, tri_off = ctx.meta
        L, M, _, R = S.shape
        _, Mw, _, Cp = Ws.shape
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(S)
            with _timed("spec_sep_mul", nbytes=4.0 * (4 * R * L * M + 2 * Cp * L * Mw)):
                check(lib().mk_spec_sep_mul(ptr(gy), ptr(Ws), ptr(gx), L, M, Mw, B, Cp, tri_off, 1, stream()), "spec_sep_mul")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(Ws)
            with _timed("spec_sep_wgrad", nbytes=4.0 * (4 * R * L * M + 2 * Cp * L * Mw)):
                check(lib().mk_spec_sep_wgrad(ptr(S), ptr(gy), ptr(gw), L, M, Mw, B, Cp, tri_off, stream()), "spec_sep_wgrad")
        return gx, gw, None, None


class DiagContractFn(torch.autograd.Function):
    """``_contract_lmwise`` (contractions.py:17-18), ``einsum("bgixy,gioxy->bgoxy")``: one Cin/G x Cout/G matrix per
    (l, m).  The weight is streamed once from its parameter layout (G, Cin/G, Cout/G, L, M) — no re-layout, and its
    gradient is written straight into that layout."""

    @staticmethod
    def forward(ctx, S, weight, B, tri_off=0):
        G, cgi, cgo, L, M = weight.shape
        cin, cout = G * cgi, G * cgo
        cip, cop = round4(cin), round4(cout)
        S = S.contiguous()
        assert S.shape == (L, M, 2, B * cip)
        wr = torch.view_as_real(weight.detach().contiguous())
        T = torch.empty((L, M, 2, B * cop), dtype=torch.float32, device=S.device)
        nb = 4.0 * (2 * B * cgi * L * M + 2 * B * cgo * L * M + 2 * cgi * cgo * L * M)
        for g in range(G):
            with _timed("spec_diag_fwd", flops=8.0 * B * cgi * cgo * L * M, nbytes=nb):
                check(lib().mk_spec_diag_apply(_addr(S, g * cgi), _addr(wr[g]), _addr(T, g * cgo), L, M, B, cgi, cgo,
                                               cip, cop, cop - cout if g == G - 1 else 0, tri_off, 0, stream()), "spec_diag_apply")
        ctx.save_for_backward(S, wr)
        ctx.meta = (B, tri_off)
        return T

    @staticmethod
    def backward(ctx, gT):
        S, wr = ctx.saved_tensors
        B, tri_off = ctx.meta
        G, cgi, cgo, L, M, _ = wr.shape
        cin, cout = G * cgi, G * cgo
        cip, cop = round4(cin), round4(cout)
        gT = gT.contiguous()
        gS = gw = None
        nb = 4.0 * (2 * B * cgi * L * M + 2 * B * cgo * L * M + 2 * cgi * cgo * L * M)
        if ctx.needs_input_grad[0]:
            gS = torch.empty_like(S)
            for g in range(G):
                with _timed("spec_diag_dgrad", flops=8.0 * B * cgi * cgo * L * M, nbytes=nb):
                    check(lib().mk_spec_diag_apply(_addr(gT, g * cgo), _addr(wr[g]), _addr(gS, g * cgi), L, M, B, cgi,
                                                   cgo, cop, cip, cip - cin if g == G - 1 else 0, tri_off, 1, stream()), "spec_diag_apply")
        if ctx.needs_input_grad[1]:
            gwr = torch.empty_like(wr)
            for g in range(G):
                with _timed("spec_diag_wgrad", flops=8.0 * B * cgi * cgo * L * M, nbytes=nb):
                    check(lib().mk_spec_diag_wgrad(_addr(S, g * cgi), _addr(gT, g * cgo), _addr(gwr[g]), L, M, B, cgi,
                                                   cgo, cip, cop, tri_off, stream()), "spec_diag_wgrad")
            gw = torch.view_as_complex(gwr)
        return gS, gw, None, None


class SToComplexFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, B, Cc, l_off=0, m_off=0):
        return s_to_complex(S, B, Cc, l_off, m_off)

    @staticmethod
    def backward(ctx, gc):
        return complex_to_s(gc), None, None, None, None


class ComplexToSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, l_off=0, m_off=0):
        """l_off / m_off: first degree / order of this rank's shard (the gradient of the l < m entries is an exact zero)"""
        ctx.meta = (*c.shape[:2], l_off, m_off)
        return complex_to_s(c)

    @staticmethod
    def backward(ctx, gS):
        B, Cc, l_off, m_off = ctx.meta
        return s_to_complex(gS.contiguous(), B, Cc, l_off, m_off), None, None


# --------------------------------------------------------------------------- #
# pointwise
# --------------------------------------------------------------------------- #
def _ws(planes, hw, dtype, device):
    ch = lib().mk_pointwise_chunks(hw, _lib.MK_BF16 if dtype == torch.bfloat16 else _lib.MK_F32, planes)
    return torch.empty((planes * ch * 2,), dtype=torch.float32, device=device)


def _batch_sum(sums, B, Cc):
    """(2, B*C) per-plane sums -> (2, C) per-channel sums; a view (no kernel) when B == 1"""
    return sums if B == 1 else sums.view(2, B, Cc).sum(1)


_FUSED_SYNC = {}        # device -> (slots, depart): hand-over buffers of the one-pass instance norm


def _fused_sync(nslots: int, planes: int, device):
    """``slots`` (all ones) / ``depart`` (zero) of csrc/pointwise.hip's one-pass norm kernels.  One pair per device: the kernels
    restore both before they finish, so every later launch may use them — as long as no two norm launches of one device run
    CONCURRENTLY (two streams); the package launches its norms on one stream at a time (eager: the current stream; captured: the
    capture stream).  The pair is allocated (generously: 8 MB) at the first call and only ever grown outside a hipGraph capture:
    a warm-up step pays for it."""
    key = str(device)
    hit = _FUSED_SYNC.get(key)
    if hit is None or hit[0].numel() < nslots or hit[1].numel() < planes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("one-pass instance norm: its hand-over buffers must exist before the step is captured (run a warm-up step)")
        torch.cuda.synchronize(device)                      # (growing: nothing may still be using the old pair)
        slots = torch.full((max(nslots, 1 << 20),), -1, dtype=torch.int64, device=device)
        depart = torch.zeros((max(planes, 1 << 16),), dtype=torch.int32, device=device)
        hit = _FUSED_SYNC[key] = (slots, depart)
    return hit


def _fused_norm_chunks(x, planes, hw, quad, kind):
    """chunks per plane when the one-pass kernels serve this call (kind: 0 forward, 1 backward, 2 backward through the fused
    GELU), else 0 (MAKANI_AMD_NORM_FUSED=0: the two-kernel A/B path)"""
    if quad is not None or os.environ.get("MAKANI_AMD_NORM_FUSED", "1") != "1":
        return 0
    return int(lib().mk_instnorm_fused_chunks(hw, dtype_code(x), planes, kind))


class InstanceNormFn(torch.autograd.Function):
    """nn.InstanceNorm2d(affine) (+ fused exact GELU); fp32 statistics, io in the input dtype.
    ``pre_bias`` (C,) folds a per-channel constant added right in front of the norm (the bias of the MLP's output
    convolution) into the kernels: they see (x + pre_bias[c]) rounded to x's dtype, the tensor the reference
    materialises.  Its gradient is the plane sum of the norm's input gradient, which is exactly zero."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fuse_gelu, pre_bias=None, quad=None, quad_sum=0.0):
        """quad: (H*W,) fp32 quadrature weights, quad_sum their sum -> area-weighted statistics (GeometricInstanceNormS2)"""
        x = x.contiguous()
        B, Cc, H, W = x.shape
        planes, hw = B * Cc, H * W
        dt = dtype_code(x)
        stats = torch.empty((planes, 2), dtype=torch.float32, device=x.device)
        g = gamma.float().contiguous() if gamma is not None else None
        b = beta.float().contiguous() if beta is not None else None
        pb = pre_bias.float().contiguous() if pre_bias is not None else None
        y = torch.empty_like(x)
        ch = _fused_norm_chunks(x, planes, hw, quad, 0)
        if ch:
            slots, depart = _fused_sync(planes * ch, planes, x.device)
            with _timed(f"instnorm_fwd{'_gelu' if fuse_gelu else ''}_n{hw}", nbytes=2.0 * x.numel() * x.element_size()):
                check(lib().mk_instnorm_fwd_fused(ptr(x), ptr(y), dt, ptr(stats), ptr(g), ptr(b), ptr(pb), ptr(slots), ptr(depart), planes, Cc, hw,
                                                  eps, 1 if fuse_gelu else 0, stream()), "instnorm_fwd_fused")
        else:
            ws = _ws(planes, hw, x.dtype, x.device)
            with _timed(f"instnorm_fwd{'_gelu' if fuse_gelu else ''}_n{hw}", nbytes=3.0 * x.numel() * x.element_size()):
                check(lib().mk_instnorm_fwd(ptr(x), ptr(y), dt, ptr(stats), ptr(ws), ptr(g), ptr(b), ptr(pb), ptr(quad), float(quad_sum),
                                            planes, Cc, hw, eps, 1 if fuse_gelu else 0, stream()), "instnorm_fwd")
        ctx.save_for_backward(x, stats, g, b, pb, quad)
        ctx.quad_sum = float(quad_sum)
        ctx.fuse_gelu = fuse_gelu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, g, b, pb, quad = ctx.saved_tensors
        B, Cc, H, W = x.shape
        planes, hw = B * Cc, H * W
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = torch.empty_like(x)
        sums = torch.empty((2, planes), dtype=torch.float32, device=x.device)
        ch = _fused_norm_chunks(x, planes, hw, quad, 2 if ctx.fuse_gelu else 1)
        if ch:
            slots, depart = _fused_sync(planes * ch, planes, x.device)
            with _timed(f"instnorm_bwd{'_gelu' if ctx.fuse_gelu else ''}_n{hw}", nbytes=3.0 * x.numel() * x.element_size()):
                check(lib().mk_instnorm_bwd_fused(ptr(x), ptr(gy), ptr(gx), dtype_code(x), ptr(stats), ptr(g), ptr(b), ptr(pb), ptr(sums),
                                                  ptr(slots), ptr(depart), planes, Cc, hw, 1 if ctx.fuse_gelu else 0, stream()), "instnorm_bwd_fused")
        else:
            ws = _ws(planes, hw, x.dtype, x.device)
            with _timed(f"instnorm_bwd{'_gelu' if ctx.fuse_gelu else ''}_n{hw}", nbytes=5.0 * x.numel() * x.element_size()):
                check(lib().mk_instnorm_bwd(ptr(x), ptr(gy), ptr(gx), dtype_code(x), ptr(stats), ptr(g), ptr(b), ptr(pb), ptr(quad),
                                            ctx.quad_sum, ptr(sums), ptr(ws), planes, Cc, hw, hw, 0, 1 if ctx.fuse_gelu else 0,
                                            stream()), "instnorm_bwd")
        s = _batch_sum(sums, B, Cc)
        gpb = torch.zeros_like(pb) if (pb is not None and ctx.needs_input_grad[5]) else None
        return gx, (s[1] if g is not None else None), (s[0] if b is not None else None), None, None, gpb, None, None


class ChannelLayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the channel dimension of an NCHW tensor (DistributedLayerNorm, makani/mpu/layer_norm.py:256-290)
    without the two transposes: csrc/chan_layernorm.hip.  ``out_dtype``: float32 under autocast (layer_norm is an fp32 op of
    torch's autocast policy), else the input dtype."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        x = x.contiguous()
        B, Cc, H, W = x.shape
        P = H * W
        y = torch.empty((B, Cc, H, W), dtype=out_dtype, device=x.device)
        stats = torch.empty((B, 2, P), dtype=torch.float32, device=x.device)
        g = gamma.float().contiguous() if gamma is not None else None
        b = beta.float().contiguous() if beta is not None else None
        with _timed(f"chan_layernorm_fwd_n{P}", nbytes=float(x.numel()) * (2 * x.element_size() + y.element_size())):
            check(lib().mk_chan_layernorm_fwd(ptr(x), dtype_code(x), ptr(y), dtype_code(y), ptr(stats), ptr(g), ptr(b), B, Cc, P,
                                              float(eps), stream()), "mk_chan_layernorm_fwd")
        ctx.save_for_backward(x, stats, g)
        ctx.has = (gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, g = ctx.saved_tensors
        B, Cc, H, W = x.shape
        P = H * W
        gy = gy.contiguous()
        if gy.dtype not in (torch.float32, x.dtype):
            gy = gy.float()
        gx = dg = db = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            with _timed(f"chan_layernorm_bwd_n{P}", nbytes=float(x.numel()) * (3 * x.element_size() + 2 * gy.element_size())):
                check(lib().mk_chan_layernorm_bwd(ptr(x), dtype_code(x), ptr(gy), dtype_code(gy), ptr(gx), ptr(stats), ptr(g), B, Cc, P,
                                                  stream()), "mk_chan_layernorm_bwd")
        if (ctx.has[0] and ctx.needs_input_grad[1]) or (ctx.has[1] and ctx.needs_input_grad[2]):
            ch = lib().mk_chan_layernorm_chunks(Cc, P)
            part = torch.empty((2, Cc, ch), dtype=torch.float32, device=x.device)
            check(lib().mk_chan_layernorm_wgrad(ptr(x), dtype_code(x), ptr(gy), dtype_code(gy), ptr(stats), ptr(part), B, Cc, P,
                                                stream()), "mk_chan_layernorm_wgrad")
            sums = part.sum(dim=2)
            dg = sums[0] if ctx.has[0] else None
            db = sums[1] if ctx.has[1] else None
        return gx, dg, db, None, None


class BiasGeluFn(torch.autograd.Function):
    """y = gelu(x + bias[c]) on NCHW (bias may be None)."""

    @staticmethod
    def forward(ctx, x, bias):
        x = x.contiguous()
        B, Cc, H, W = x.shape
        bf = bias.float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        check(lib().mk_bias_gelu_fwd(ptr(x), ptr(bf), ptr(y), dtype_code(x), B * Cc, Cc, H * W, stream()), "bias_gelu_fwd")
        ctx.save_for_backward(x, bf)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, bf = ctx.saved_tensors
        B, Cc, H, W = x.shape
        planes, hw = B * Cc, H * W
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = torch.empty_like(x)
        need_b = bf is not None and ctx.needs_input_grad[1]
        sums = torch.empty((2, planes), dtype=torch.float32, device=x.device) if need_b else None
        ws = _ws(planes, hw, x.dtype, x.device) if need_b else None
        check(lib().mk_bias_gelu_bwd(ptr(x), ptr(bf), ptr(gy), ptr(gx), ptr(sums), ptr(ws), dtype_code(x), planes, Cc,
                                     hw, stream()), "bias_gelu_bwd")
        gb = _batch_sum(sums, B, Cc)[0] if need_b else None
        return gx, gb


# --------------------------------------------------------------------------- #
# bf16 channel GEMMs (1x1 convolutions on NCHW planes)
# --------------------------------------------------------------------------- #
def round8(n: int) -> int:
    return (n + 7) // 8 * 8


def pad_weight_bf16(w2d: torch.Tensor) -> torch.Tensor:
    """(M, K) any float dtype -> (M, round8(K)) bf16, zero padded (the A operand of mk_conv1x1_nn)."""
    M, K = w2d.shape
    lda = round8(K)
    if lda == K:
        return w2d.to(torch.bfloat16).contiguous()
    out = torch.zeros((M, lda), dtype=torch.bfloat16, device=w2d.device)
    out[:, :K] = w2d
    return out


def conv1x1_nn(A: torch.Tensor, K: int, x: torch.Tensor, bias=None, act=False, want_pre=False, residual=None,
               gelu_grad_of=None):
    """y[b] = epi(A[:, :K] @ x[b]);  x (B, K, H, W) bf16, A (M, lda) bf16 -> y (B, M, H, W) bf16 (and pre-activation)."""
    B, Kx, H, W = x.shape
    assert Kx == K and x.dtype == torch.bfloat16 and A.dtype == torch.bfloat16
    x = x.contiguous()
    M, lda = A.shape
    N = H * W
    y = torch.empty((B, M, H, W), dtype=torch.bfloat16, device=x.device)
    ypre = torch.empty_like(y) if (act and want_pre) else None
    bf = bias.float().contiguous() if bias is not None else None
    r = residual.contiguous() if residual is not None else None
    g = gelu_grad_of.contiguous() if gelu_grad_of is not None else None
    # algorithmic bytes: read x, write y, plus every fused operand that is read (residual, gelu' argument) or written
    # (pre-activation) in the same pass
    extra = (ypre is not None) + (r is not None) + (g is not None)
    with _timed(f"conv1x1_nn_m{M}_k{K}_n{N}", flops=2.0 * B * M * K * N, nbytes=2.0 * B * N * (M * (1 + extra) + K)):
        check(lib().mk_conv1x1_nn(ptr(A), ptr(x), ptr(y), ptr(ypre), ptr(bf), ptr(r), ptr(g), M, K, lda, B, N,
                                  1 if act else 0, stream()), "mk_conv1x1_nn")
    return y, ypre


def conv1x1_wgrad(g: torch.Tensor, x: torch.Tensor, want_bias: bool = False):
    """dW[m][k] = sum_{b,n} g[b][m][n] x[b][k][n]  -> (M, K) fp32.  ``want_bias``: returns (dW, db) with the bias gradient
    db[m] = sum_{b,n} g[b][m][n] — formed inside the weight-gradient kernel from the rows it streams anyway where the library
    can (the ring kernel: every shape of the train step), by a separate plane-sum pass elsewhere."""
    B, M, H, W = g.shape
    K = x.shape[1]
    N = H * W
    g, x = g.contiguous(), x.contiguous()
    nws = lib().mk_conv1x1_wgrad_workspace(M, K, B, N)
    part = torch.empty((nws,), dtype=torch.float32, device=g.device)
    dW = torch.empty((M, K), dtype=torch.float32, device=g.device)
    fused = want_bias and bool(lib().mk_conv1x1_wgrad_fuses_bias(M, K, B, N))
    with _timed(f"conv1x1_wgrad_m{M}_k{K}_n{N}", flops=2.0 * B * M * K * N, nbytes=2.0 * B * N * (M + K)):
        if fused:
            db = torch.empty((M,), dtype=torch.float32, device=g.device)
            check(lib().mk_conv1x1_wgrad_bias(ptr(g), ptr(x), ptr(dW), ptr(db), ptr(part), M, K, B, N, 0, stream()), "mk_conv1x1_wgrad_bias")
        else:
            check(lib().mk_conv1x1_wgrad(ptr(g), ptr(x), ptr(dW), ptr(part), M, K, B, N, 0, stream()), "mk_conv1x1_wgrad")
    if not want_bias:
        return dW
    return dW, (db if fused else _sum_planes(g))


def _sum_planes(t: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) -> (C,) fp32 sums over batch and pixels (bias gradient of a 1x1 convolution)"""
    if not (t.is_cuda and t.dim() == 4 and t.dtype in (torch.bfloat16, torch.float32)):
        raise RuntimeError("makani_amd ops need 4-d bf16 / fp32 GPU tensors (the HIP path has no CPU fallback)")
    t = t.contiguous()
    B, Cc, H, W = t.shape
    sums = torch.empty((2, B * Cc), dtype=torch.float32, device=t.device)
    check(lib().mk_plane_sums(ptr(t), dtype_code(t), ptr(sums), ptr(_ws(B * Cc, H * W, t.dtype, t.device)), B * Cc, H * W,
                              stream()), "mk_plane_sums")
    return _batch_sum(sums, B, Cc)[0]


class Conv1x1Fn(torch.autograd.Function):
    """y = W x (+ bias) (+ residual) on NCHW bf16; weight is the fp32 (M, K, 1, 1) parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        M, K = weight.shape[0], weight.shape[1]
        A, At = weight_operands(weight, need_t=ctx.needs_input_grad[0])
        y, _ = conv1x1_nn(A, K, x, bias=bias, residual=residual)
        ctx.save_for_backward(x, weight, At)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, At = ctx.saved_tensors
        M, K = weight.shape[0], weight.shape[1]
        gy = gy.contiguous()
        gx = gw = gb = gr = None
        if ctx.needs_input_grad[0]:
            gx, _ = conv1x1_nn(At, M, gy)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:                                  # the bias gradient rides on the weight gradient's pass over gy
                gw, gb = conv1x1_wgrad(gy, x, want_bias=True)
                gw = gw.view_as(weight)
            else:
                gw = conv1x1_wgrad(gy, x).view_as(weight)
        elif want_b:
            gb = _sum_planes(gy)
        if ctx.has_res and ctx.needs_input_grad[3]:
            gr = gy
        return gx, gw, gb, gr


class ConvGeluConvFn(torch.autograd.Function):
    """y = W2 gelu(W1 x + b1) (+ b2) (+ residual): the MLP / EncoderDecoder pattern
    (makani/models/common/layers.py:603-643,768-823) with bias+GELU fused into the first GEMM's
    epilogue and gelu' fused into the epilogue of the second GEMM's data-gradient."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        H1, K = w1.shape[0], w1.shape[1]
        M = w2.shape[0]
        A1, A1t = weight_operands(w1, need_t=ctx.needs_input_grad[0])
        A2, A2t = weight_operands(w2, need_t=True)
        h, a1 = conv1x1_nn(A1, K, x, bias=b1, act=True, want_pre=True)
        y, _ = conv1x1_nn(A2, H1, h, bias=b2)
        ctx.save_for_backward(x, w1, w2, a1, h, A1t, A2t)
        ctx.has_b = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, w2, a1, h, A1t, A2t = ctx.saved_tensors
        H1, K = w1.shape[0], w1.shape[1]
        M = w2.shape[0]
        gy = gy.contiguous()
        # ga1 = (W2^T gy) * gelu'(a1)
        ga1, _ = conv1x1_nn(A2t, M, gy, gelu_grad_of=a1)
        # (the bias gradients ride on the weight gradients' passes over gy / ga1 where both are wanted)
        gw2 = gb2 = gw1 = gb1 = None
        want_b2, want_b1 = ctx.has_b[1] and ctx.needs_input_grad[4], ctx.has_b[0] and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[3] and want_b2:
            gw2, gb2 = conv1x1_wgrad(gy, h, want_bias=True)
            gw2 = gw2.view_as(w2)
        else:
            gw2 = conv1x1_wgrad(gy, h).view_as(w2) if ctx.needs_input_grad[3] else None
            gb2 = _sum_planes(gy) if want_b2 else None
        if ctx.needs_input_grad[1] and want_b1:
            gw1, gb1 = conv1x1_wgrad(ga1, x, want_bias=True)
            gw1 = gw1.view_as(w1)
        else:
            gw1 = conv1x1_wgrad(ga1, x).view_as(w1) if ctx.needs_input_grad[1] else None
            gb1 = _sum_planes(ga1) if want_b1 else None
        gx = None
        if ctx.needs_input_grad[0]:
            gx, _ = conv1x1_nn(A1t, H1, ga1)
        return gx, gw1, gb1, gw2, gb2


def want_bf16_shadow(param):
    """mark a fp32 parameter whose bf16 copy ``makani_amd.optim.FusedAdamW`` should emit with every update"""
    param._mk_want_bf16 = True


def _shadow_valid(weight):
    return (getattr(weight, "_mk_shadow", None) is not None and getattr(weight, "_mk_shadow_version", -1) == weight._version
            and getattr(weight, "_mk_shadow_ptr", 0) == weight.data_ptr() and weight._mk_shadow.device == weight.device)


def cast_weight(weight, dtype):
    """weight in the GEMM dtype: the optimizer's bf16 shadow when it belongs to exactly this version of the
    parameter (any in-place change through torch bumps ``_version``, re-pointing ``param.data`` changes the storage
    address: both invalidate it), else a cast.  In-place writes THROUGH ``param.data`` (``p.data.mul_()``) bypass the
    version counter and are not supported while shadows are in use; ``load_state_dict`` / ``copy_`` / optimizers are."""
    if dtype == weight.dtype:
        return weight
    if dtype == torch.bfloat16 and _shadow_valid(weight):
        K = weight.numel() // weight.shape[0]
        sh = weight._mk_shadow                      # (M, round8(K)): a strided (M, K) slice when K is not a multiple of 8
        return sh.view(weight.shape) if sh.shape[1] == K else sh[:, :K]
    return weight.to(dtype)


def weight_operands(weight, need_t=False):
    """bf16 operands of the HIP channel GEMMs for a (M, K, 1, 1) weight: A = (M, round8(K)) zero padded and, if asked
    for, its transpose At = (K, round8(M)).  FusedAdamW writes both with every update (valid for exactly that version
    of the parameter); otherwise they are produced here."""
    M = weight.shape[0]
    K = weight.numel() // M
    if _shadow_valid(weight):
        return weight._mk_shadow, (weight._mk_shadow_t if need_t else None)
    w2 = weight.detach().view(M, K)
    return pad_weight_bf16(w2), (pad_weight_bf16(w2.t()) if need_t else None)


def _pad_last(t, mult):
    """zero-pad the last dimension to a multiple of ``mult`` (a copy only when it is ragged)"""
    r = (-t.shape[-1]) % mult
    return t if r == 0 else torch.nn.functional.pad(t, (0, r))


def chan_gemm_f32(w: torch.Tensor, x: torch.Tensor, out: torch.Tensor = None, accumulate: bool = False, transposed: bool = False):
    """out[b, g] (+)= W[g] x[b, g] (or W[g]^T x[b, g]) in fp32 on the package's own GEMM engine (the split-bf16 / exact-fp32
    MFMA kernels of csrc/xgemm.hip / sgemm.hip, the engine of the Legendre transforms; ``MAKANI_AMD_GEMM`` picks the
    arithmetic): the fp32 form of the 1x1 convolutions and of the grouped channel mixes (parity runs without autocast; under
    bf16 autocast the LDS-DMA kernels of csrc/conv1x1.hip / the streaming kernel of csrc/groupmix.hip run instead).
    w (M, K) with x (B, K | M, N), or w (G, M, K) with x (B, G, K | M, N); fp32, pixel index contiguous."""
    grouped = w.dim() == 3
    w3 = w if grouped else w.unsqueeze(0)
    x4 = x if grouped else x.unsqueeze(1)
    o4 = None if out is None else (out if grouped else out.unsqueeze(1))
    B, G, Kx, N = x4.shape
    Gw, M, K = w3.shape
    rows, kdim = (K, M) if transposed else (M, K)
    assert Gw == G and Kx == kdim and x4.dtype == torch.float32 and w3.dtype == torch.float32
    if N % 4:                                   # ragged pixel count (toy grids): operate on zero-padded planes
        r = chan_gemm_f32(w3, _pad_last(x4, 4), None if o4 is None else _pad_last(o4, 4), accumulate, transposed)[..., :N]
        if o4 is None:
            r = r.contiguous()
            return r if grouped else r[:, 0]
        o4.copy_(r)
        return out
    wp = _pad_last(w3.contiguous(), 4)          # (G, M, K4): 16-byte rows
    K4 = wp.shape[2]
    x4 = x4.contiguous()
    if o4 is None:
        o4 = torch.empty((B, G, rows, N), dtype=torch.float32, device=x4.device)
        accumulate = False
    assert o4.is_contiguous()
    a = dict(a_row=1, a_k=K4) if transposed else dict(a_row=K4, a_k=1)
    g = _gemm(A=wp.data_ptr(), B=x4.data_ptr(), C=o4.data_ptr(), a_batch=0, a_inner=M * K4, **a,
              b_batch=G * kdim * N, b_inner=kdim * N, b_col=1, b_k=N, c_batch=G * rows * N, c_inner=rows * N, c_row=N,
              M=rows, N=N, K=kdim, batch=B * G, inner=G, beta=1 if accumulate else 0)
    with _timed(f"conv1x1_f32_m{rows}_k{kdim}_n{N}", flops=2.0 * B * G * M * K * N,
                nbytes=4.0 * B * G * N * (M + K) * (1 + int(accumulate))):
        _run_gemm(g, False, "chan_gemm_f32")
    if out is not None:
        return out
    return o4 if grouped else o4[:, 0]


def chan_wgrad_f32(gy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW[g][m][k] = sum_{b, n} gy[b, g, m, n] x[b, g, k, n] in fp32 on the same engine: the pixel sum is cut into slabs that
    run as batch entries (each a k-major / k-major GEMM), their partial products are added up afterwards.
    gy (B, M, N), x (B, K, N) -> (M, K); or gy (B, G, M, N), x (B, G, K, N) -> (G, M, K)."""
    grouped = gy.dim() == 4
    gy4 = gy if grouped else gy.unsqueeze(1)
    x4 = x if grouped else x.unsqueeze(1)
    B, G, M, N = gy4.shape
    K = x4.shape[2]
    if N % 4:
        gy4, x4 = _pad_last(gy4, 4), _pad_last(x4, 4)
        N = gy4.shape[-1]
    gy4, x4 = gy4.contiguous(), x4.contiguous()
    ns = 1
    for cand in range(min(256, max(1, N // 2048)), 0, -1):      # slabs of >= 2048 pixels whose length is a multiple of 4
        if N % (4 * cand) == 0:
            ns = cand
            break
    chunk = N // ns
    Kp = round4(K)
    part = torch.empty((B, G, ns, M, Kp), dtype=torch.float32, device=gy4.device)
    for b in range(B):
        g = _gemm(A=gy4[b].data_ptr(), B=x4[b].data_ptr(), C=part[b].data_ptr(), a_batch=M * N, a_inner=chunk, a_row=N, a_k=1,
                  b_batch=K * N, b_inner=chunk, b_col=N, b_k=1, c_batch=ns * M * Kp, c_inner=M * Kp, c_row=Kp,
                  M=M, N=K, K=chunk, batch=G * ns, inner=ns)
        with _timed(f"conv1x1_f32_wgrad_m{M}_k{K}_n{N}", flops=2.0 * G * M * K * N, nbytes=4.0 * G * N * (M + K)):
            _run_gemm(g, False, "chan_wgrad_f32")
    dW = part.sum(dim=(0, 2))[..., :K]
    return dW if grouped else dW[0]


class GroupMmFn(torch.autograd.Function):
    """z[b, g] = W[g] x[b, g] for arbitrary group sizes in fp32 (the group sizes the streaming kernel of csrc/groupmix.hip is
    not instantiated for): forward, data gradient and weight gradient on the fp32 GEMM engine."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return chan_gemm_f32(W.detach().float(), x.float()).to(x.dtype)

    @staticmethod
    def backward(ctx, gz):
        x, W = ctx.saved_tensors
        gz = gz.contiguous().float()
        gx = chan_gemm_f32(W.detach().float(), gz, transposed=True).to(x.dtype) if ctx.needs_input_grad[0] else None
        gW = chan_wgrad_f32(gz, x.float()).to(W.dtype) if ctx.needs_input_grad[1] else None
        return gx, gW


class ConvMmFn(torch.autograd.Function):
    """y = W x (+ residual) for everything the bf16 LDS-DMA kernels do not take (fp32 tensors: parity runs without
    autocast; bf16 planes whose pixel count is not a multiple of 8): forward, data gradient and weight gradient on the
    package's fp32 GEMM engine (``chan_gemm_f32`` / ``chan_wgrad_f32``) — no library GEMM anywhere in the product."""

    @staticmethod
    def forward(ctx, x, weight, residual, residual_is_fresh=False):
        B, K, H, W = x.shape
        M = weight.shape[0]
        N = H * W
        w2 = weight.detach().reshape(M, K).float()
        xf = x if x.dtype == torch.float32 else x.float()
        # with a residual the product may accumulate INTO it (beta = 1, no 88-800 MB copy of the residual first), which
        # autograd is told through mark_dirty.  That is only legal when no upstream node saved that tensor for its own
        # backward (a ReLU saves its output, an instance norm or a GEMM of this package does not), which this function
        # cannot see: the caller has to vouch for it with ``residual_is_fresh``; otherwise the output is a new tensor.
        inplace = (residual_is_fresh and residual is not None and residual.is_contiguous() and residual.dtype == torch.float32
                   and x.dtype == torch.float32 and N % 4 == 0 and not residual._is_view()
                   and not (residual.is_leaf and residual.requires_grad))
        if inplace:
            out = residual
            chan_gemm_f32(w2, xf.reshape(B, K, N), out.view(B, M, N), accumulate=True)
        elif residual is not None:
            out = residual.float().clone(memory_format=torch.contiguous_format)
            chan_gemm_f32(w2, xf.reshape(B, K, N), out.view(B, M, N), accumulate=True)
            out = out.to(x.dtype)
        else:
            out = chan_gemm_f32(w2, xf.reshape(B, K, N)).view(B, M, H, W).to(x.dtype)
        ctx.save_for_backward(x, weight)
        ctx.has_res = residual is not None
        if inplace:
            ctx.mark_dirty(residual)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, K, H, W = x.shape
        M = weight.shape[0]
        N = H * W
        gy = gy.contiguous()
        gx = gw = gr = None
        if ctx.needs_input_grad[0]:
            gx = chan_gemm_f32(weight.detach().reshape(M, K).float(), gy.float().reshape(B, M, N), transposed=True)
            gx = gx.view(B, K, H, W).to(x.dtype)
        if ctx.needs_input_grad[1]:
            if gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and N % 8 == 0:
                gw = conv1x1_wgrad(gy, x).view_as(weight)
            else:
                gw = chan_wgrad_f32(gy.float().reshape(B, M, N), x.float().reshape(B, K, N)).reshape(weight.shape).to(weight.dtype)
        if ctx.has_res and ctx.needs_input_grad[2]:
            gr = gy
        return gx, gw, gr, None


# --------------------------------------------------------------------------- #
# distributed instance norm (spatial model parallelism)
# --------------------------------------------------------------------------- #
def merge_moments(means: torch.Tensor, variances: torch.Tensor, counts: torch.Tensor):
    """Combine per-shard (mean, biased variance, count) along dim 0 into the moments of the union
    (Chan et al.; same result as the sequential Welford merge of makani/mpu/layer_norm.py:54-81)."""
    n = counts.sum(0)
    mean = (means * counts).sum(0) / n
    m2 = (variances * counts + counts * (means - mean) ** 2).sum(0)
    return mean, m2 / n, n


def _all_gather_stack(t, group):
    import torch.distributed as dist
    P = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and t.is_cuda:       # CPU-staged (gloo test runs on one GPU)
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(P)]
        dist.all_gather(out, h, group=group)
        return torch.stack(out, dim=0).to(t.device)
    out = [torch.empty_like(t) for _ in range(P)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out, dim=0)


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


_COUNT_ROWS = {}        # (local count, device) -> a (1, 2) device row {count, 0}: a per-rank constant, no collective involved


def _count_row(count: float, device):
    key = (float(count), str(device))
    row = _COUNT_ROWS.get(key)
    if row is None:
        row = _COUNT_ROWS[key] = torch.tensor([[float(count), 0.0]], dtype=torch.float32, device=device)
    return row


class DistInstanceNormFn(torch.autograd.Function):
    """Instance norm over a plane that is sharded across the ``spatial`` process group
    (``DistributedInstanceNorm2d``, makani/mpu/layer_norm.py:108-170): local fp32 statistics (HIP) ->
    all-gather of the ranks' (mean, rstd) -> merge with the ranks' counts (HIP, ``mk_instnorm_merge``) -> normalise (+GELU)
    with the merged statistics (HIP).  Backward all-reduces the two per-plane sums between the HIP reduce and apply phases.
    With ``quad`` (this shard's quadrature weights, ``quad_sum`` their sum) the moments are area-weighted and the counts are
    the weight sums: ``DistributedGeometricInstanceNormS2`` (makani/mpu/layer_norm.py:173-253).
    No host synchronisation at all: the ranks' shard counts ride in the all-gathered statistics tensor and stay on the device."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fuse_gelu, group, quad=None, quad_sum=0.0):
        import torch.distributed as dist
        x = x.contiguous()
        B, Cc, H, W = x.shape
        planes, hw = B * Cc, H * W
        dt = dtype_code(x)
        # the ranks' shard sizes (pixel counts, or quadrature-weight sums) travel WITH the statistics as one extra row of the
        # all-gathered tensor: every call sees the counts of THIS call's split on every rank — no cache that could go stale or
        # let one rank skip a collective the others enter, no extra collective, and no device -> host read
        # (what a hipGraph capture of the h x w step requires)
        stats = torch.empty((planes + 1, 2), dtype=torch.float32, device=x.device)
        stats[planes:].copy_(_count_row(float(quad_sum) if quad is not None else float(hw), x.device))
        ws = _ws(planes, hw, x.dtype, x.device)
        nb = float(x.numel() * x.element_size())
        tag = f"{'_gelu' if fuse_gelu else ''}_n{hw}"
        with _timed(f"instnorm_dist_stats{tag}", nbytes=nb):
            check(lib().mk_instnorm_stats(ptr(x), dt, ptr(stats), ptr(ws), planes, hw, eps, ptr(quad), float(quad_sum), stream()), "instnorm_stats")
        allst = _all_gather_stack(stats, group)                  # (P, planes + 1, 2): every rank's local {mean, rstd} + its count
        counts = allst[:, planes, 0].contiguous()
        # what the backward pass scales its all-reduced sums by (see there): hw / total, resp. 1 / Q — one device scalar, computed here
        bscale = ((1.0 if quad is not None else float(hw)) / counts.double().sum()).float()
        allst = allst[:, :planes].contiguous()
        mstats = torch.empty((planes, 2), dtype=torch.float32, device=x.device)
        check(lib().mk_instnorm_merge(ptr(allst), ptr(counts), ptr(mstats), planes, allst.shape[0], eps, stream()), "instnorm_merge")
        g = gamma.float().contiguous() if gamma is not None else None
        b = beta.float().contiguous() if beta is not None else None
        y = torch.empty_like(x)
        with _timed(f"instnorm_dist_apply{tag}", nbytes=2.0 * nb):
            check(lib().mk_instnorm_apply(ptr(x), ptr(y), dt, ptr(mstats), ptr(g), ptr(b), planes, Cc, hw,
                                          1 if fuse_gelu else 0, stream()), "instnorm_apply")
        ctx.save_for_backward(x, mstats, g, b, quad, bscale)
        ctx.meta = (fuse_gelu, group)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mstats, g, b, quad, bscale = ctx.saved_tensors
        fuse_gelu, group = ctx.meta
        B, Cc, H, W = x.shape
        planes, hw = B * Cc, H * W
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = torch.empty_like(x)
        sums = torch.empty((2, planes), dtype=torch.float32, device=x.device)
        ws = _ws(planes, hw, x.dtype, x.device)
        fg = 1 if fuse_gelu else 0
        nb = float(x.numel() * x.element_size())
        tag = f"{'_gelu' if fuse_gelu else ''}_n{hw}"
        with _timed(f"instnorm_dist_bwd_reduce{tag}", nbytes=2.0 * nb):
            check(lib().mk_instnorm_bwd(ptr(x), ptr(gy), ptr(gx), dtype_code(x), ptr(mstats), ptr(g), ptr(b), None, None, 0.0, ptr(sums),
                                        ptr(ws), planes, Cc, hw, hw, 1, fg, stream()), "instnorm_bwd(reduce)")
        local = _batch_sum(sums, B, Cc).clone()                   # this rank's share of dgamma / dbeta
        _all_reduce_sum(sums, group)
        # the kernel divides the sums by its `hw_total` argument (unweighted) or multiplies them by q_i per element (weighted);
        # the true total — the sum of the ranks' counts along possibly ragged splits, exact small integers in fp32 — lives on the
        # device: the sums are pre-scaled there (hw / total, resp. 1 / Q for the normalised weights p_i = q_i / Q with the
        # kernel's crop term switched off; the factor was formed in fp64 in the forward pass) and the kernel is told the LOCAL count
        sums = sums * bscale
        with _timed(f"instnorm_dist_bwd_apply{tag}", nbytes=3.0 * nb):
            check(lib().mk_instnorm_bwd(ptr(x), ptr(gy), ptr(gx), dtype_code(x), ptr(mstats), ptr(g), ptr(b), None, ptr(quad),
                                        1.0 if quad is not None else 0.0, ptr(sums), ptr(ws), planes, Cc, hw, hw, 2, fg, stream()),
                  "instnorm_bwd(apply)")
        dgamma = local[1] if g is not None else None
        dbeta = local[0] if b is not None else None
        return gx, dgamma, dbeta, None, None, None, None, None
