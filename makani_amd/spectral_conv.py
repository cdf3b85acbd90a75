"""Drop-in ``SpectralConv`` (boundary B2 of SURVEY.md §8b).

Mirrors ``makani/models/common/spectral_convolution.py:116-264``: same
constructor, same ``weight`` parameter (complex64, ``(G, Cin/G, Cout/G, L)``
for ``dhconv``) with the ``is_shared_mp`` / ``sharded_dims_mp`` annotations,
same ``forward(x) -> (y, residual)`` and the same ``ValueError`` conditions.

The arithmetic is one HIP pipeline in fp32 that never leaves the internal layouts:
  x --rFFT--> F --Legendre--> S --dhconv (complex MFMA GEMM)--> T --Legendre^T--> F' --irFFT--> y
with the bf16<->fp32 casts of the reference (``:237-239,252-256``) fused into
the FFT kernels' loads and stores.
"""
import math
import os
from functools import partial

import torch
import torch.nn as nn

from ._lib import device_guard

from . import ops
from .sht import RealSHT, InverseRealSHT


class SpectralConv(nn.Module):
    def __init__(self, forward_transform, inverse_transform, in_channels, out_channels, num_groups=1,
                 operator_type="dhconv", separable=False, bias=False, gain=1.0):
        super().__init__()
        if in_channels % num_groups != 0:
            raise ValueError(f"in_channels ({in_channels}) must be divisible by num_groups ({num_groups})")
        if out_channels % num_groups != 0:
            raise ValueError(f"out_channels ({out_channels}) must be divisible by num_groups ({num_groups})")

        self.forward_transform = forward_transform
        self.inverse_transform = inverse_transform
        self.in_channels, self.out_channels, self.num_groups = in_channels, out_channels, num_groups
        self.modes_lat = inverse_transform.lmax
        self.modes_lon = inverse_transform.mmax
        self.scale_residual = (forward_transform.nlat != inverse_transform.nlat) or (
            forward_transform.nlon != inverse_transform.nlon
        )
        if hasattr(forward_transform, "grid"):
            self.scale_residual = self.scale_residual or (forward_transform.grid != inverse_transform.grid)
        self.operator_type = operator_type
        self.separable = separable

        if forward_transform.lmax != self.modes_lat:
            raise ValueError(f"forward transform lmax ({forward_transform.lmax}) must match modes_lat ({self.modes_lat})")
        if forward_transform.mmax != self.modes_lon:
            raise ValueError(f"forward transform mmax ({forward_transform.mmax}) must match modes_lon ({self.modes_lon})")
        if not isinstance(forward_transform, RealSHT) or not isinstance(inverse_transform, InverseRealSHT):
            raise TypeError("makani_amd.SpectralConv needs makani_amd RealSHT / InverseRealSHT transforms")

        # spatial model parallelism: the weight is sharded along l over the "h" group and shared over "w"
        # (spectral_convolution.py:169-173,195-198)
        self._tri_off = 0
        if hasattr(inverse_transform, "l_shapes"):
            ih, iw = inverse_transform.comm_rank_polar, inverse_transform.comm_rank_azimuth
            self.modes_lat_local = inverse_transform.l_shapes[ih]
            self.modes_lon_local = inverse_transform.m_shapes[iw]
            self.nlat_local = inverse_transform.lat_shapes[ih]
            self.nlon_local = inverse_transform.lon_shapes[iw]
            self._tri_off = inverse_transform.l_off - inverse_transform.m_off
        else:
            self.modes_lat_local, self.modes_lon_local = self.modes_lat, self.modes_lon
            self.nlat_local, self.nlon_local = inverse_transform.nlat, inverse_transform.nlon

        weight_shape = [num_groups, in_channels // num_groups]
        if not separable:
            weight_shape += [out_channels // num_groups]
        if operator_type == "diagonal":
            weight_shape += [self.modes_lat_local, self.modes_lon_local]
        elif operator_type == "dhconv":
            weight_shape += [self.modes_lat_local]
        else:
            raise ValueError(f"Unsupported operator type f{operator_type}")
        if separable and in_channels != out_channels:
            raise ValueError(f"separable operators keep the channel count: in_channels ({in_channels}) != out_channels ({out_channels})")
        if operator_type == "dhconv" and not separable and num_groups > 1 and (
                (in_channels // num_groups) % 4 or (out_channels // num_groups) % 4):
            raise NotImplementedError("grouped dhconv on the MFMA engine needs group sizes that are multiples of 4 "
                                      f"(got {in_channels // num_groups} -> {out_channels // num_groups})")

        # same initialisation as the reference (spectral_convolution.py:184-193), including its broadcast of the
        # per-l scale against the LAST weight axis (which is m for the "diagonal" operator)
        scale = math.sqrt(gain / (in_channels // num_groups)) * torch.ones(self.modes_lat_local, dtype=torch.complex64)
        scale[0] *= math.sqrt(2.0)
        w0 = scale * torch.randn(*weight_shape, dtype=torch.complex64)
        cgi, cgo = in_channels // num_groups, out_channels // num_groups
        if operator_type == "dhconv" and not separable and num_groups == 1 and cgi % 4 == 0 and cgo % 4 == 0:
            # same shape and values, memory in the order the dhconv GEMMs read ([l][i][o]): no re-layout per step
            w0 = ops.native_w_empty(cgi, cgo, self.modes_lat_local).copy_(w0)
        self.weight = nn.Parameter(w0)
        if operator_type == "dhconv":
            self.weight.is_shared_mp = ["matmul", "w"]
            self.weight.sharded_dims_mp = [None for _ in weight_shape]
            self.weight.sharded_dims_mp[-1] = "h"
        else:
            self.weight.is_shared_mp = ["matmul"]
            self.weight.sharded_dims_mp = [None for _ in weight_shape]
            self.weight.sharded_dims_mp[-1] = "w"
            self.weight.sharded_dims_mp[-2] = "h"

        if bias:
            self.bias = nn.Parameter(torch.zeros(1, out_channels, 1, 1))
            self.bias.is_shared_mp = ["model"]
            self.bias.sharded_dims_mp = [None, None, None, None]

    def _contract(self, S, B):
        """the four contractions of makani/models/common/contractions.py:17-54 on the S-layout"""
        w = self.weight
        if self.separable:                                   # "bgixy,gixy->bgixy" / "bgixy,gix->bgixy"
            w3 = w.reshape(self.in_channels, self.modes_lat_local, -1)          # (C, L, M) or (C, L, 1)
            return ops.SepContractFn.apply(S, ops.WeightToSFn.apply(w3), B, self._tri_off)
        if self.operator_type == "diagonal":                  # "bgixy,gioxy->bgoxy"
            return ops.DiagContractFn.apply(S, w, B, self._tri_off)
        if self.num_groups > 1:                               # "bgixy,giox->bgoxy", G > 1
            return ops.GroupedDhconvFn.apply(S, w, B, self._tri_off)
        return ops.DhconvFn.apply(S, w, B, self._tri_off)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, x):
        if x.dim() != 4:
            raise ValueError(f"expected (B, C, H, W), got {tuple(x.shape)}")
        dtype = x.dtype
        B, C = x.shape[:2]
        residual = x
        S = self.forward_transform.analysis(x)                    # fp32 coefficients, bf16 read fused
        if self.scale_residual:
            residual = self.inverse_transform.synthesis(S, B, C, out_dtype=dtype)
        T = self._contract(S, B)
        y = self.inverse_transform.synthesis(T, B, self.out_channels, out_dtype=dtype)
        if hasattr(self, "bias"):
            y = y + self.bias.to(dtype=y.dtype)
        return y, residual
