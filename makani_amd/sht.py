"""Drop-in spherical harmonic transforms (boundary B1 of SURVEY.md §8b).

``RealSHT`` / ``InverseRealSHT`` keep the constructor, attributes and call
convention of ``torch_harmonics.RealSHT`` / ``InverseRealSHT`` as makani uses
them (``makani/models/networks/sfnonet.py:802-805``,
``makani/models/common/spectral_convolution.py:142-149,239-253``):
``forward(x: (..., nlat, nlon) real) -> (..., lmax, mmax) complex64`` and the
reverse, differentiable, arbitrary leading dims.  The arithmetic runs in the HIP
library: truncated rFFT -> fp32-MFMA Legendre GEMM (and the mirror image).

In addition to the complex64 API the modules expose the internal S-layout
(``analysis`` / ``synthesis``) so that ``SpectralConv`` can chain
SHT -> contraction -> iSHT without ever materialising complex64 tensors.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ._lib import device_guard

from . import legendre as _leg
from . import ops


def _as4d(x, trailing):
    lead = x.shape[: x.dim() - trailing]
    if len(lead) == 0:
        return x.reshape(1, 1, *x.shape[-trailing:]), lead
    if len(lead) == 1:
        return x.reshape(1, lead[0], *x.shape[-trailing:]), lead
    Bn = int(np.prod(lead[:-1]))
    return x.reshape(Bn, lead[-1], *x.shape[-trailing:]), lead


class _SHTBase(nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__()
        self.nlat, self.nlon, self.grid, self.norm, self.csphase = nlat, nlon, grid, norm, csphase
        self._theta, self._wq = _leg.colatitudes(nlat, grid)     # validates `grid`
        if grid == "lobatto":
            self.lmax = lmax or nlat - 1
        else:
            self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        if self.mmax > nlon // 2 + 1:
            raise ValueError(f"mmax={self.mmax} exceeds nlon//2+1={nlon // 2 + 1}")
        _leg.factorize_half(nlon)                                   # raises for unsupported nlon
        self.kp = ops.round4(nlat)

    def _padded(self, P):
        """(mmax, lmax, nlat) fp64 -> fp32 in both orientations, unit-stride dim zero-padded to a multiple of 4:
        natural (mmax, lmax, kp) and transposed (mmax, nlat, lp)."""
        nat = np.zeros((self.mmax, self.lmax, self.kp), dtype=np.float32)
        nat[:, :, : self.nlat] = P
        lp = ops.round4(self.lmax)
        tr = np.zeros((self.mmax, self.nlat, lp), dtype=np.float32)
        tr[:, :, : self.lmax] = np.transpose(P, (0, 2, 1))
        return torch.from_numpy(nat), torch.from_numpy(tr)

    def extra_repr(self):
        return f"nlat={self.nlat}, nlon={self.nlon}, lmax={self.lmax}, mmax={self.mmax}, grid={self.grid}"


class RealSHT(_SHTBase):
    """Forward transform.  Buffers ``weights`` (mmax, lmax, kp) / ``weights_t`` (mmax, nlat, lp), fp32 =
    Legendre functions times quadrature weights in both orientations (unit-stride dim padded to 4)."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        P = _leg.legendre_matrix(self.mmax, self.lmax, self._theta, norm=norm, inverse=False, csphase=csphase)
        nat, tr = self._padded(P * self._wq[None, None, :])
        self.register_buffer("weights", nat, persistent=False)        # (mmax, lmax, kp): backward (synthesis-shaped)
        self.register_buffer("weights_t", tr, persistent=False)       # (mmax, nlat, lp): forward
        c = 2.0 * math.pi / nlon
        self._w = (c, c, c)

    def analysis(self, x4: torch.Tensor) -> torch.Tensor:
        """(B, C, nlat, nlon) f32|bf16 -> S-layout (lmax, mmax, 2, B*Cp)."""
        if x4.shape[-2] != self.nlat or x4.shape[-1] != self.nlon:
            raise ValueError(f"expected (..., {self.nlat}, {self.nlon}), got {tuple(x4.shape)}")
        F = ops.RfftFn.apply(x4, self.mmax, ops.round4(x4.shape[1]), self._w)
        return ops.AnalysisFn.apply(F, self.weights, self.weights_t)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"RealSHT (HIP) supports float32 / bfloat16 input, got {x.dtype}")
        x4, lead = _as4d(x, 2)
        S = self.analysis(x4)
        c = ops.SToComplexFn.apply(S, x4.shape[0], x4.shape[1])
        return c.reshape(*lead, self.lmax, self.mmax)


class InverseRealSHT(_SHTBase):
    """Inverse transform.  Buffer ``pct``: (mmax, lmax, kp) fp32 Legendre functions."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        P = _leg.legendre_matrix(self.mmax, self.lmax, self._theta, norm=norm, inverse=True, csphase=csphase)
        nat, tr = self._padded(P)
        self.register_buffer("pct", nat, persistent=False)            # (mmax, lmax, kp): forward
        self.register_buffer("pct_t", tr, persistent=False)           # (mmax, nlat, lp): backward (analysis-shaped)
        self._w = (1.0, 2.0, 1.0)

    def synthesis(self, S: torch.Tensor, B: int, C: int, out_dtype=torch.float32) -> torch.Tensor:
        """S-layout (lmax, mmax, 2, B*Cp) -> (B, C, nlat, nlon)."""
        F = ops.SynthesisFn.apply(S, self.pct, self.pct_t, self.nlat)
        return ops.IrfftFn.apply(F, B, C, self.nlon, out_dtype, self._w)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, c: torch.Tensor) -> torch.Tensor:
        if c.dtype != torch.complex64:
            raise TypeError(f"InverseRealSHT (HIP) supports complex64 input, got {c.dtype}")
        if c.shape[-2] != self.lmax or c.shape[-1] != self.mmax:
            raise ValueError(f"expected (..., {self.lmax}, {self.mmax}), got {tuple(c.shape)}")
        c4, lead = _as4d(c, 2)
        S = ops.ComplexToSFn.apply(c4)
        x = self.synthesis(S, c4.shape[0], c4.shape[1])
        return x.reshape(*lead, self.nlat, self.nlon)
