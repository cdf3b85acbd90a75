"""Spatial (h x w) model-parallel spherical harmonic transforms over RCCL all-to-all.

Counterpart of ``torch_harmonics.distributed`` as makani uses it
(``makani/models/networks/sfnonet.py:786-805,823-838``; schedule mirrored in-tree by
``makani/mpu/fft.py:148-182,214-249`` and ``makani/mpu/mappings.py:38-67``):

forward   (w) planes<->lon all-to-all -> rFFT over full longitude, truncated
          (w) m<->planes all-to-all   -> (h) planes<->lat all-to-all
          Legendre analysis over full latitude with the local m-slice of the matrix
          (h) l<->planes all-to-all
inverse   the same four exchanges in reverse order around synthesis and irFFT.

Differences from the reference's NCCL pattern, chosen for xGMI (point-to-point links, every
peer pair has its own link): each exchange is ONE ``all_to_all`` on the internal F/S layouts
(fp32 planar, the layouts the HIP kernels consume), the sharded index is always the OUTERMOST or
a plane index, so send chunks are contiguous slices where possible; only kept modes travel
(truncation happens before the first spectral exchange); the triangular structure is carried
through the shards (``tri_off``) so the GEMMs skip the structurally-zero half on every rank.

Local compute goes through a small backend object.  The product backend is the HIP library
(``HipBackend``) — there is no CPU fallback; tests inject the CPU oracle to verify the
exchange schedule with gloo.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from . import comm as _comm
from . import legendre as _leg
from . import ops
from .sht import RealSHT, InverseRealSHT

def _size(group) -> int:
    return 1 if group is None else dist.get_world_size(group)


def _rank(group) -> int:
    return 0 if group is None else dist.get_rank(group)


_POLAR = None      # "h" group: latitude / degree l
_AZIMUTH = None    # "w" group: longitude / order m
_SPATIAL = None    # "spatial" group: all h x w ranks of one model instance
_INIT = False


def init(polar_group, azimuth_group, spatial_group=None):
    """``thd.init(polar_group, azimuth_group)`` (``sfnonet.py:786-789``); ``None`` = not split.
    ``spatial_group`` (makani's "spatial" = h x w, ``makani/utils/comm.py:114-201``) is needed by the
    distributed instance norm; it defaults to whichever of the two groups is split when only one is."""
    global _POLAR, _AZIMUTH, _SPATIAL, _INIT
    _POLAR, _AZIMUTH, _INIT = polar_group, azimuth_group, True
    if spatial_group is None and not (_size(polar_group) > 1 and _size(azimuth_group) > 1):
        spatial_group = polar_group if _size(polar_group) > 1 else azimuth_group
    _SPATIAL = spatial_group


def spatial_group():
    """The h x w group (needed by the instance-norm statistics, not by the transforms)."""
    if _INIT and _SPATIAL is None and _size(_POLAR) > 1 and _size(_AZIMUTH) > 1:
        raise ValueError("h and w are both split: init() needs the spatial (h x w) process group")
    return _SPATIAL


def spatial_size() -> int:
    return _size(_POLAR) * _size(_AZIMUTH) if _INIT else 1


def is_initialized() -> bool:
    return _INIT


def ensure_initialized() -> bool:
    """``sfnonet.py:786-789``: when the process-group tree (``makani_amd.comm``: built by ``comm.init(h, w)`` or adopted
    from ``makani.utils.comm``) names a split ``spatial`` group and nobody called ``init`` yet, do it from the tree's
    ``h`` / ``w`` groups.  Returns whether spatial model parallelism is active."""
    if not _INIT:
        _comm.autodetect()
        if _comm.get_size("spatial") > 1:
            init(_comm.get_group("h") if _comm.get_size("h") > 1 else None,
                 _comm.get_group("w") if _comm.get_size("w") > 1 else None, _comm.get_group("spatial"))
    return _INIT and spatial_size() > 1


def polar_group():
    return _POLAR


def azimuth_group():
    return _AZIMUTH




def polar_group_size():
    return _size(_POLAR)


def azimuth_group_size():
    return _size(_AZIMUTH)


def polar_group_rank():
    return _rank(_POLAR)


def azimuth_group_rank():
    return _rank(_AZIMUTH)


def compute_split_shapes(size: int, num_chunks: int) -> List[int]:
    """ceil-div chunks, the last one smaller; floor-div fallback when the last would be empty
    (used by ``makani/mpu/fft.py:50-51``, ``makani/utils/grids.py:154-165``)."""
    if num_chunks == 1:
        return [size]
    chunk = (size + num_chunks - 1) // num_chunks
    last = max(0, size - chunk * (num_chunks - 1))
    if last == 0:
        chunk = size // num_chunks
        last = size - chunk * (num_chunks - 1)
    return [chunk] * (num_chunks - 1) + [last]


def split_tensor_along_dim(tensor, dim, num_chunks):
    if tensor.shape[dim] < num_chunks:
        raise ValueError(f"cannot split dimension {dim} of size {tensor.shape[dim]} into {num_chunks} chunks")
    return torch.split(tensor, compute_split_shapes(tensor.shape[dim], num_chunks), dim=dim)


# --------------------------------------------------------------------------- #
# the exchange primitive
# --------------------------------------------------------------------------- #
def _pad_to4(t: torch.Tensor, dim: int) -> torch.Tensor:
    n = t.shape[dim]
    if n % 4 == 0:
        return t
    shape = list(t.shape)
    shape[dim] = 4 - n % 4
    return torch.cat([t, t.new_zeros(shape)], dim=dim)


# exchange accounting (bench.py reports it per step and rank at N > 1, so that the first scaling curve can be read against
# SURVEY.md §8(e)'s communication budget): bytes this rank SENDS to other ranks and the number of all-to-alls, per
# (direction, group size): "polar" (h), "azimuth" (w), "spatial" (the fused schedule's one h x w exchange)
COMM_STATS = {}


def _group_label(group) -> str:
    """which direction of the h x w block an exchange runs over (with h == w the sizes alone cannot tell)"""
    if group is _SPATIAL and not (group is _POLAR or group is _AZIMUTH):
        return "spatial"
    if group is _POLAR:
        return "polar"
    if group is _AZIMUTH:
        return "azimuth"
    return "other"


def _count_exchange(send, group):
    me = dist.get_rank(group)
    n = sum(t.numel() * t.element_size() for i, t in enumerate(send) if i != me)
    st = COMM_STATS.setdefault((_group_label(group), dist.get_world_size(group)), {"bytes_sent": 0, "all_to_alls": 0})
    st["bytes_sent"] += n
    st["all_to_alls"] += 1


def _exchange(recv, send, group, count=False):
    """one all-to-all; RCCL ("nccl") has it natively, gloo (CPU tests) is served by paired isend/irecv"""
    if count:
        _count_exchange(send, group)
    if dist.get_backend(group) != "gloo":
        dist.all_to_all(recv, send, group=group)
        return
    me = dist.get_rank(group)
    recv[me].copy_(send[me])
    on_gpu = send[me].is_cuda                 # gloo moves host memory only: stage through the host
    hs = [t.cpu() if on_gpu else t for t in send]
    hr = [torch.empty(t.shape, dtype=t.dtype) if on_gpu else t for t in recv]
    ops_ = []
    for peer in range(len(send)):
        if peer == me or (hs[peer].numel() == 0 and hr[peer].numel() == 0):      # (empty slabs: nothing to post)
            continue
        gp = dist.get_global_rank(group, peer)
        ops_.append(dist.P2POp(dist.isend, hs[peer], gp, group=group))
        ops_.append(dist.P2POp(dist.irecv, hr[peer], gp, group=group))
    for req in (dist.batch_isend_irecv(ops_) if ops_ else ()):
        req.wait()
    if on_gpu:
        for peer in range(len(send)):
            if peer != me:
                recv[peer].copy_(hr[peer])


def _a2a(x, sdim, ssizes, cdim, csizes, group, me):
    """split ``x`` along ``sdim`` into one chunk per peer (``ssizes``), exchange, join what arrives along ``cdim``
    (``csizes[src]`` wide from peer ``src``).  One copy per exchange instead of the reference's two: when the chunks
    that arrive are slabs of the OUTERMOST dimension they are received straight into slices of the result (no ``cat``),
    and when the chunks that leave are slabs of the outermost dimension they are sent as views (no pack copy)."""
    P = len(ssizes)
    send = [c if c.is_contiguous() else c.contiguous() for c in torch.split(x, ssizes, dim=sdim)]
    shape = list(x.shape)
    shape[sdim] = ssizes[me]
    shape[cdim] = sum(csizes)
    if cdim == 0:                                   # arriving slabs are contiguous ranges of the result
        y = torch.empty(shape, dtype=x.dtype, device=x.device)
        recv = list(torch.split(y, csizes, dim=0))
        _exchange(recv, send, group, count=True)
        return y
    recv = []
    for src in range(P):
        sh = list(shape)
        sh[cdim] = csizes[src]
        recv.append(torch.empty(sh, dtype=x.dtype, device=x.device))
    _exchange(recv, send, group, count=True)
    return torch.cat(recv, dim=cdim)


class _TransposeFn(torch.autograd.Function):
    """Split ``x`` along ``sdim`` into one chunk per peer (sizes ``ssizes``), exchange, concatenate what
    arrives along ``cdim`` (``makani/mpu/mappings.py:38-67``).  ``svalid``/padded dims: a padded dim is
    narrowed to its valid extent before splitting and re-padded to a multiple of 4 after concatenation.
    Backward is the reverse exchange."""

    @staticmethod
    def forward(ctx, x, sdim, ssizes, cdim, csizes, group):
        P = dist.get_world_size(group)
        me = dist.get_rank(group)
        assert len(ssizes) == P and len(csizes) == P
        svalid = sum(ssizes)
        xs = x.narrow(sdim, 0, svalid) if x.shape[sdim] != svalid else x
        xc = xs.narrow(cdim, 0, csizes[me]) if xs.shape[cdim] != csizes[me] else xs
        y = _a2a(xc, sdim, ssizes, cdim, csizes, group, me)
        ctx.meta = (sdim, ssizes, cdim, csizes, group, x.shape, me)
        return y

    @staticmethod
    def backward(ctx, gy):
        sdim, ssizes, cdim, csizes, group, xshape, me = ctx.meta
        g = gy.narrow(cdim, 0, sum(csizes)) if gy.shape[cdim] != sum(csizes) else gy
        gx = _a2a(g, cdim, csizes, sdim, ssizes, group, me)
        # restore the (padded) input extents
        for d in (sdim, cdim):
            if gx.shape[d] != xshape[d]:
                shape = list(gx.shape)
                shape[d] = xshape[d] - gx.shape[d]
                gx = torch.cat([gx, gx.new_zeros(shape)], dim=d)
        return gx, None, None, None, None, None


def transpose(x, sdim, ssizes, cdim, csizes, group, pad_dims=()):
    """exchange (no-op for a group of one), then zero-pad ``pad_dims`` to multiples of 4 (kernel layouts)."""
    if group is not None and dist.get_world_size(group) > 1:
        x = _TransposeFn.apply(x, sdim, list(ssizes), cdim, list(csizes), group)
    for d in pad_dims:
        x = _pad_to4(x, d)
    return x


# --------------------------------------------------------------------------- #
# local compute backends
# --------------------------------------------------------------------------- #
class _ReduceFromSpatialFn(torch.autograd.Function):
    """``reduce_from_parallel_region(x, "spatial")`` (makani/mpu/mappings.py): SUM all-reduce forward,
    identity backward."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.clone()
        ops._all_reduce_sum(y, group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


def reduce_from_spatial_region(x):
    return _ReduceFromSpatialFn.apply(x, spatial_group()) if spatial_size() > 1 else x


class HipBackend:
    """Local compute on the HIP library (the product path).  ``segmented``: the four operations of the fused schedule
    (makani_amd/dist_pipeline.py) are available — FFTs that address their operands per peer, Legendre GEMMs on the
    latitude-major operand."""
    segmented = True

    @staticmethod
    def seg_supported(nlon):
        return ops.fft_seg_supported(nlon)

    @staticmethod
    def _seg(p, a):
        base = [[p.base[j][i] + a * p.m_shapes[j] * 2 * p.sub[p.iw][i] for i in range(p.h)] for j in range(p.w)]
        return ops.fft_seg_desc(p.m_shapes, p.sub[p.iw], base, xseg=p.w, x_stride=p.pw[p.iw] * p.hl * p.wl, x_nlat=p.hl)

    @staticmethod
    def rfft_seg(xbuf, a, b, fs, p, w):
        """latitudes [a, b) of the pieces ``xbuf`` (w, P_w, lat_loc, lon piece) -> the per-peer slabs of the flat buffer ``fs``"""
        ops.rfft_rows_seg(xbuf, a * p.wl, fs, p.pw[p.iw], b - a, p.nlon, p.M, w, HipBackend._seg(p, a))

    @staticmethod
    def irfft_seg(fr, a, b, xbuf, p, w):
        ops.irfft_rows_seg(fr, xbuf, a * p.wl, p.pw[p.iw], b - a, p.nlon, p.M, w, HipBackend._seg(p, a))

    @staticmethod
    def analysis_lm(G, matT, L, m_off):
        return ops.legendre_analysis(G, matT, L, m_off, lat_major=True)

    @staticmethod
    def synthesis_lm(T, mat, nlat, m_off):
        return ops.legendre_synthesis(T, mat, nlat, m_off, lat_major=True)

    @staticmethod
    def analysis_lm_blocks(G, matT, L, m_off):
        """G (w, nlat, M_loc, 2, sub): all plane blocks of the Legendre phase in ONE launch -> (L, w, M_loc, 2, sub)"""
        return ops.legendre_analysis(G, matT, L, m_off, lat_major=True, blocks=True)

    @staticmethod
    def synthesis_lm_blocks(T, mat, nlat, m_off):
        """T (L, w, M_loc, 2, sub) -> (w, nlat, M_loc, 2, sub)"""
        return ops.legendre_synthesis(T, mat, nlat, m_off, lat_major=True, blocks=True)

    @staticmethod
    def rfft(x4, mmax, w):
        return ops.RfftFn.apply(x4, mmax, ops.round4(x4.shape[1]), w)

    @staticmethod
    def irfft(F, planes, nlon, dtype, w):
        return ops.IrfftFn.apply(F, 1, planes, nlon, dtype, w)

    @staticmethod
    def analysis(F, mat, matT, m_off):
        return ops.AnalysisFn.apply(F.contiguous(), mat, matT, m_off)

    @staticmethod
    def synthesis(S, mat, matT, nlat, m_off):
        return ops.SynthesisFn.apply(S.contiguous(), mat, matT, nlat, m_off)


_BACKEND = HipBackend


def _offsets(sizes):
    out = [0]
    for s in sizes:
        out.append(out[-1] + s)
    return out


class _DistBase:
    def _setup_dist(self):
        if not is_initialized():
            raise RuntimeError("makani_amd.distributed.init(polar_group, azimuth_group) has not been called")
        self.comm_size_polar, self.comm_rank_polar = polar_group_size(), polar_group_rank()
        self.comm_size_azimuth, self.comm_rank_azimuth = azimuth_group_size(), azimuth_group_rank()
        self.lat_shapes = compute_split_shapes(self.nlat, self.comm_size_polar)
        self.lon_shapes = compute_split_shapes(self.nlon, self.comm_size_azimuth)
        self.l_shapes = compute_split_shapes(self.lmax, self.comm_size_polar)
        self.m_shapes = compute_split_shapes(self.mmax, self.comm_size_azimuth)
        self.m_off = _offsets(self.m_shapes)[self.comm_rank_azimuth]
        self.l_off = _offsets(self.l_shapes)[self.comm_rank_polar]

    def _plane_shapes(self, planes):
        return compute_split_shapes(planes, self.comm_size_azimuth), compute_split_shapes(planes, self.comm_size_polar)

    def _plan(self, planes):
        """sizes and slab offsets of the fused schedule for this plane count (makani_amd/dist_pipeline.py), cached"""
        from . import dist_pipeline as dp
        plans = self.__dict__.setdefault("_plans", {})
        key = (planes, os.environ.get("MAKANI_AMD_DIST_CHUNKS", "2"))
        if key not in plans:
            plans[key] = dp.Plan(self, planes)
        return plans[key]


class DistributedRealSHT(RealSHT, _DistBase):
    """``thd.DistributedRealSHT``: local ``(B, C, nlat_loc, nlon_loc)`` -> local ``(B, C, l_loc, m_loc)``.
    Keeps only the m-slice of the Legendre matrix this azimuth rank needs."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self._setup_dist()
        m0, m1 = self.m_off, self.m_off + self.m_shapes[self.comm_rank_azimuth]
        self.weights = self.weights[m0:m1].contiguous()
        self.weights_t = self.weights_t[m0:m1].contiguous()

    def analysis(self, x4: torch.Tensor) -> torch.Tensor:
        """(B, C, nlat_loc, nlon_loc) -> S-layout (l_loc, m_loc, 2, round4(B*C)), planes = b*C + c."""
        B, C = x4.shape[:2]
        if B > 1 and C % 4:            # the S layout pads every sample's channels to a multiple of 4: zero planes
            x4 = torch.nn.functional.pad(x4, (0, 0, 0, 0, 0, (-C) % 4))
            C = x4.shape[1]
        hl, wl = self.lat_shapes[self.comm_rank_polar], self.lon_shapes[self.comm_rank_azimuth]
        if x4.shape[-2] != hl or x4.shape[-1] != wl:
            raise ValueError(f"expected local shape (..., {hl}, {wl}), got {tuple(x4.shape)}")
        P = B * C
        from . import dist_pipeline as dp
        if dp.eligible(self, x4.dtype):              # the fused schedule: per-peer addressing in the FFTs, one h x w exchange
            return dp.DistAnalysisFn.apply(x4.reshape(P, hl, wl), self, self._plan(P))
        pw, ph = self._plane_shapes(P)
        x = x4.reshape(1, P, hl, wl)
        # (w) planes <-> lon
        x = transpose(x, 1, pw, 3, self.lon_shapes, azimuth_group())
        F = _BACKEND.rfft(x.contiguous(), self.mmax, self._w)                       # (M, hl, 2, round4(P_w))
        # (w) m <-> planes
        F = transpose(F, 0, self.m_shapes, 3, pw, azimuth_group(), pad_dims=(3,))      # (M_loc, hl, 2, round4(P))
        # (h) planes <-> lat
        F = transpose(F, 3, ph, 1, self.lat_shapes, polar_group(), pad_dims=(3,))      # (M_loc, nlat, 2, round4(P_h))
        S = _BACKEND.analysis(F, self.weights, self.weights_t, self.m_off)            # (L, M_loc, 2, round4(P_h))
        # (h) l <-> planes
        S = transpose(S, 0, self.l_shapes, 3, ph, polar_group(), pad_dims=(3,))        # (L_loc, M_loc, 2, round4(P))
        return S

    @torch.compiler.disable(recursive=True)
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .sht import _as4d
        x4, lead = _as4d(x, 2)
        S = self.analysis(x4)
        c = ops.SToComplexFn.apply(S, x4.shape[0], x4.shape[1], self.l_off, self.m_off)
        return c.reshape(*lead, c.shape[-2], c.shape[-1])


class DistributedInverseRealSHT(InverseRealSHT, _DistBase):
    """``thd.DistributedInverseRealSHT``: local ``(B, C, l_loc, m_loc)`` -> local ``(B, C, nlat_loc, nlon_loc)``."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True):
        super().__init__(nlat, nlon, lmax, mmax, grid, norm, csphase)
        self._setup_dist()
        m0, m1 = self.m_off, self.m_off + self.m_shapes[self.comm_rank_azimuth]
        self.pct = self.pct[m0:m1].contiguous()
        self.pct_t = self.pct_t[m0:m1].contiguous()

    def synthesis(self, S: torch.Tensor, B: int, C: int, out_dtype=torch.float32) -> torch.Tensor:
        Cc = C
        if B > 1 and C % 4:            # S holds round4(C) planes per sample: transform the zero planes too, drop them at the end
            C = C + (-C) % 4
        P = B * C
        hl, wl = self.lat_shapes[self.comm_rank_polar], self.lon_shapes[self.comm_rank_azimuth]
        from . import dist_pipeline as dp
        if dp.eligible(self, out_dtype):
            x = dp.DistSynthesisFn.apply(S, self, self._plan(P), out_dtype)
            return x.reshape(B, C, hl, wl)[:, :Cc]
        pw, ph = self._plane_shapes(P)
        # (h) planes <-> l
        S = transpose(S, 3, ph, 0, self.l_shapes, polar_group(), pad_dims=(3,))       # (L, M_loc, 2, round4(P_h))
        F = _BACKEND.synthesis(S, self.pct, self.pct_t, self.nlat, self.m_off)        # (M_loc, nlat, 2, round4(P_h))
        # (h) lat <-> planes
        F = transpose(F, 1, self.lat_shapes, 3, ph, polar_group(), pad_dims=(3,))      # (M_loc, hl, 2, round4(P))
        # (w) planes <-> m
        F = transpose(F, 3, pw, 0, self.m_shapes, azimuth_group(), pad_dims=(3,))      # (M, hl, 2, round4(P_w))
        x = _BACKEND.irfft(F.contiguous(), pw[self.comm_rank_azimuth], self.nlon, out_dtype, self._w)
        # (w) lon <-> planes
        x = transpose(x, 3, self.lon_shapes, 1, pw, azimuth_group())                  # (1, P, hl, wl)
        return x.reshape(B, C, hl, wl)[:, :Cc]

    @torch.compiler.disable(recursive=True)
    def forward(self, c: torch.Tensor) -> torch.Tensor:
        from .sht import _as4d
        c4, lead = _as4d(c, 2)
        S = ops.ComplexToSFn.apply(c4, self.l_off, self.m_off)
        x = self.synthesis(S, c4.shape[0], c4.shape[1])
        return x.reshape(*lead, x.shape[-2], x.shape[-1])


class DistributedInstanceNorm2d(nn.Module):
    """``makani/mpu/layer_norm.py:108-170``: instance norm over planes sharded across the spatial group."""

    def __init__(self, num_features, eps=1e-5, affine=False):
        super().__init__()
        self.num_features, self.eps, self.affine = num_features, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.weight.is_shared_mp = ["spatial"]
            self.bias.is_shared_mp = ["spatial"]
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    @torch.compiler.disable(recursive=True)
    def forward(self, x, fuse_gelu=False):
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise ValueError(f"expected (B, {self.num_features}, H, W), got {tuple(x.shape)}")
        return ops.DistInstanceNormFn.apply(x, self.weight, self.bias, self.eps, fuse_gelu, spatial_group())


class DistributedGeometricInstanceNormS2(DistributedInstanceNorm2d):
    """``makani/mpu/layer_norm.py:173-253``: instance norm with the (normalised) quadrature weights of the sphere grid over
    planes sharded across the spatial group — local area-weighted moments with count = sum of the local weights, merged
    over the group, then the usual normalise (+ affine, + fused GELU) in the HIP kernels."""

    def __init__(self, img_shape, crop_shape, crop_offset, grid_type, num_features, eps=1e-05, affine=False):
        super().__init__(num_features, eps, affine)
        from .losses import GridQuadrature, grid_to_quadrature_rule
        quad = GridQuadrature(grid_to_quadrature_rule(grid_type), img_shape=img_shape, crop_shape=crop_shape,
                              crop_offset=crop_offset, normalize=True, distributed=True).quad_weight
        self.register_buffer("quad_weight", quad.reshape(-1).contiguous().float(), persistent=False)
        self._qsum = float(quad.double().sum())          # this shard's share of the total weight

    @torch.compiler.disable(recursive=True)
    def forward(self, x, fuse_gelu=False):
        if x.dim() != 4 or x.shape[1] != self.num_features or x.shape[-2] * x.shape[-1] != self.quad_weight.numel():
            raise ValueError(f"expected (B, {self.num_features}, H_loc, W_loc) with {self.quad_weight.numel()} local grid points, got {tuple(x.shape)}")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return ops.DistInstanceNormFn.apply(x, self.weight, self.bias, self.eps, fuse_gelu, spatial_group(), self.quad_weight, self._qsum)


# --------------------------------------------------------------------------- #
# gradient reduction (makani/mpu/mappings.py:321-525) and the sharded gradient norm
# (makani/utils/training/training_helpers.py:123-165)
# --------------------------------------------------------------------------- #
def _real(g):
    """real view of a gradient, in memory order when the tensor is dense but not C-contiguous (native-order dhconv
    weight gradients): reductions and norms do not care about the order, collectives need contiguity"""
    r = torch.view_as_real(g) if g.is_complex() else g
    d = _lib.dense_view(r)
    return r if d is None else d


class GradReducer:
    """The reductions of makani's communication hook (``mappings.py:460-523``) issued from post-accumulate-grad hooks
    so that they overlap with the rest of backward:

      * MEAN over the ``data`` group for every parameter;
      * SUM over every model-parallel group named in ``param.is_shared_mp`` (a parameter without the annotation counts
        as shared over ``model``, ``mappings.py:398-401``): the l-sharded dhconv weights carry ``["matmul", "w"]``, the
        pointwise weights and norm parameters ``["spatial"]``, a position embedding ``[]``.

    A gradient of >= ``big_bytes`` is all-reduced on its own as soon as it is final (the eight 283 MB spectral weights are
    natural large xGMI messages); when it belongs to several groups the later stages are chained in ``finish()`` without
    any copy.  Smaller gradients are flattened into one bucket per reduction signature; the reduced bucket then BECOMES
    the gradients (views), so nothing is copied back.  ``finish()`` runs automatically at the end of ``backward()``
    (autograd engine callback, the mechanism DDP uses) and is idempotent, so calling it explicitly is harmless."""

    def __init__(self, model, comm=None, big_bytes=8 << 20, zero=False):
        """``zero``: ZeRO-1 over the data group — the data-parallel stage of every gradient of at least ``big_bytes``
        whose size divides evenly is a REDUCE-SCATTER: rank r of the data group ends up with the mean of slice r of the
        flattened gradient in ``param._mk_zero[0]`` (``p.grad`` keeps the local, unreduced values), which
        ``makani_amd.optim.FusedAdamW`` consumes (sharded optimizer state, in-place all-gather of the parameter)."""
        self.comm = comm or _comm
        self.big_bytes = big_bytes
        self.zero = bool(zero)
        self.enabled = True        # False inside GradReduceWrapper.no_sync()
        self.pending = []          # (work, real-view gradient, remaining stages)
        self.small = {}            # signature -> (stages, [params])
        self._armed = False
        self.plan = {}
        data_size = self.comm.get_size("data")
        names = [n for n in self.comm.get_comm_names() if n != "data" and self.comm.get_size(n) > 1]
        # "model" and "spatial" may be the same ranks under two names (no matmul parallelism): reduce once
        seen_members = {}
        self._avg = False
        if data_size > 1:
            self._avg = self._probe_avg(self.comm.get_group("data"), next(model.parameters()).device)
        for pname, p in model.named_parameters():
            if not p.requires_grad:
                continue
            shared = getattr(p, "is_shared_mp", None)
            if shared is None:
                shared = ["model"]
            stages, used = [], set()
            for n in names:
                if n in shared:
                    g = self.comm.get_group(n)
                    if id(g) in used:
                        continue
                    used.add(id(g))
                    stages.append((g, "sum", 1.0))
            if data_size > 1:
                stages.append((self.comm.get_group("data"), "avg" if self._avg else "sum", 1.0 if self._avg else 1.0 / data_size))
            self.plan[pname] = stages
            if stages:
                p.register_post_accumulate_grad_hook(lambda q, st=tuple(stages): self._hook(q, st))
        self.active = any(self.plan.values())

    @staticmethod
    def _probe_avg(group, device):
        """RCCL averages inside the collective (ReduceOp.AVG): no scaling pass afterwards.  gloo has no AVG."""
        try:
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
            return abs(float(probe) - 1.0) < 1e-6
        except (RuntimeError, ValueError, NotImplementedError):
            return False

    @staticmethod
    def _issue(t, stage):
        g, kind, _ = stage
        return dist.all_reduce(t, op=dist.ReduceOp.AVG if kind == "avg" else dist.ReduceOp.SUM, group=g, async_op=True)

    def _arm(self):
        if not self._armed:
            self._armed = True
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(self.finish)

    def _scatter_ok(self, g, stages):
        n = dist.get_world_size(stages[-1][0])
        return (self.zero and stages[-1][0] is self.comm.get_group("data") and n > 1 and g.numel() % (4 * n) == 0
                and g.dtype == torch.float32)

    def _issue_stage(self, p, g, stages):
        """the next stage of a big gradient: an all-reduce, or (ZeRO, last stage = data) a reduce-scatter into the shard"""
        if len(stages) == 1 and self._scatter_ok(g, stages):
            grp, kind, _ = stages[0]
            n = dist.get_world_size(grp)
            flat = g.reshape(-1)
            z = getattr(p, "_mk_zero", None)
            shard = z[0] if z is not None and z[0] is not None and z[0].numel() == flat.numel() // n else None
            if shard is None:
                old = getattr(p, "_mk_zero_buf", None)
                shard = old if old is not None and old.numel() == flat.numel() // n else torch.empty(flat.numel() // n, dtype=g.dtype, device=g.device)
                p._mk_zero_buf = shard
            p._mk_zero = (shard, grp, n, dist.get_rank(grp))
            op = dist.ReduceOp.AVG if kind == "avg" else dist.ReduceOp.SUM
            if dist.get_backend(grp) == "gloo":              # (CPU tests) gloo has no reduce_scatter_tensor
                tmp = flat.clone()
                work = dist.all_reduce(tmp, op=op, group=grp, async_op=True)
                return _ScatterAfter(work, tmp, shard, dist.get_rank(grp))
            return dist.reduce_scatter_tensor(shard, flat, op=op, group=grp, async_op=True)
        return self._issue(g, stages[0])

    def _hook(self, p, stages):
        if not self.enabled:           # GradReduceWrapper.no_sync(): the gradient keeps accumulating locally
            return
        self._arm()
        g = _real(p.grad)
        ops.grad_ssq_drop(g)           # the reduced gradient is not the tensor whose squares its producer summed (in-place collectives do not bump versions)
        dense = g if g.is_contiguous() else _dense(g)
        if dense is not None and dense.numel() * dense.element_size() >= self.big_bytes:
            self.pending.append([self._issue_stage(p, dense, list(stages)), dense, list(stages), p])
        else:
            self.small.setdefault(tuple((id(s[0]), s[1]) for s in stages), (stages, []))[1].append(p)

    def finish(self):
        self._armed = False
        for stages, params in self.small.values():
            params = sorted(params, key=lambda q: not q.grad.is_complex())     # complex first: their views need even offsets
            flat = torch.cat([_real(q.grad).reshape(-1) for q in params])
            for st in stages:
                self._issue(flat, st).wait()
                if st[2] != 1.0:
                    flat.mul_(st[2])
            off = 0
            for q in params:            # the reduced bucket becomes the gradients (views, no copy-back kernels)
                r = _real(q.grad)
                n = r.numel()
                piece = flat[off:off + n]
                if not q.grad.is_contiguous():          # dense in another order (native dhconv weight): keep its strides
                    r.copy_(piece.view(r.shape))
                else:
                    q.grad = torch.view_as_complex(piece.view(*q.grad.shape, 2)) if q.grad.is_complex() else piece.view_as(q.grad)
                off += n
        self.small = {}
        while self.pending:
            nxt = []
            for work, g, stages, p in self.pending:
                work.wait()
                scattered = len(stages) == 1 and self._scatter_ok(g, stages)
                if stages[0][2] != 1.0:
                    (p._mk_zero[0] if scattered else g).mul_(stages[0][2])
                if len(stages) > 1:
                    nxt.append([self._issue_stage(p, g, stages[1:]), g, stages[1:], p])
            self.pending = nxt


class _ScatterAfter:
    """gloo stand-in for an asynchronous reduce-scatter (CPU tests): all-reduce a copy, keep this rank's slice"""

    def __init__(self, work, tmp, shard, rank):
        self.work, self.tmp, self.shard, self.rank = work, tmp, shard, rank

    def wait(self):
        self.work.wait()
        n = self.shard.numel()
        self.shard.copy_(self.tmp[self.rank * n:(self.rank + 1) * n])


def _dense(t):
    from ._lib import dense_view
    return dense_view(t)


class GradReduceWrapper(nn.Module):
    """What ``init_gradient_reduction_hooks`` returns: the model under ``.module`` (as DDP exposes it) with the
    reductions armed; ``backward()`` alone completes them."""

    def __init__(self, module, reducer):
        super().__init__()
        self.module = module
        self.reducer = reducer

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        """``DistributedDataParallel.no_sync()`` as makani's trainer uses it under gradient accumulation
        (``makani/utils/training/deterministic_trainer.py:531-546``: every micro-batch but the last runs inside it): backward
        passes inside the context issue NO collective — the hooks return at once and ``p.grad`` accumulates this rank's
        contributions —, the first backward pass outside it reduces the accumulated gradients (the post-accumulate hooks see
        the sums).  As with DDP, forward AND backward of a micro-batch belong inside the context."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.reducer.enabled
            self.reducer.enabled = False
            try:
                yield
            finally:
                self.reducer.enabled = old
        return ctx()


def init_gradient_reduction_hooks(model, device=None, reduction_buffer_count=1, broadcast_buffers=True,
                                  find_unused_parameters=False, gradient_as_bucket_view=True, static_graph=False,
                                  verbose=None, comm=None, zero=False):
    """Signature and semantics of ``makani/mpu/mappings.py:321-525``: returns the model unchanged when
    ``torch.distributed`` is not initialised, otherwise a wrapper (``.module`` = the model) whose backward pass performs
    the data-parallel mean and the per-group sums of the ``is_shared_mp`` annotations.  The DDP-specific knobs are
    accepted for call compatibility; buffers are never broadcast (sharded Legendre buffers legitimately differ)."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    c = comm or _comm
    if not c.is_initialized():
        c.autodetect()
    red = GradReducer(model, c, zero=zero)
    if verbose:
        for n, st in red.plan.items():
            print(f"[grad reduction] {n}: {[(k, dist.get_world_size(g)) for g, k, _ in st]}")
    return GradReduceWrapper(model, red)


def total_grad_norm(model, comm=None):
    """Global 2-norm of the gradients with sharded parameters counted once (``training_helpers.py:123-160``): the
    squared norms of parameters sharded over a model-parallel group (``sharded_dims_mp``) are summed over that group,
    replicated parameters count once."""
    c = comm or _comm
    groups = {}
    from .optim import _check_live_shard
    for p in model.parameters():
        z = getattr(p, "_mk_zero", None)
        _check_live_shard(p)
        if z is not None and z[0] is not None:          # ZeRO: this rank holds the reduced slice; the slices add up over "data"
            key = tuple(g for g in getattr(p, "sharded_dims_mp", []) if g is not None and c.get_size(g) > 1) + ("data",)
            groups.setdefault(key, []).append(z[0])
            continue
        if p.grad is None:
            continue
        key = tuple(g for g in getattr(p, "sharded_dims_mp", []) if g is not None and c.get_size(g) > 1)
        groups.setdefault(key, []).append(_real(p.grad))
    partials = []
    for key, grads in groups.items():
        part = torch.stack(torch._foreach_norm(grads)).square().sum()
        for g in key:
            ops._all_reduce_sum(part.view(1), c.get_group(g))
        partials.append(part)
    if not partials:
        return torch.zeros(())
    return torch.stack(partials).sum().sqrt()


def clip_coefficient(model, max_norm, comm=None):
    """(1,) fp32 device tensor ``min(1, max_norm / (||g|| + 1e-6))`` for ``FusedAdamW.step(grad_scale=...)``"""
    total = total_grad_norm(model, comm)
    return torch.clamp(max_norm / (total + 1e-6), max=1.0).float().reshape(1)
