"""Drop-in geometric losses on the HIP path.

``GridQuadrature`` mirrors ``makani/utils/grids.py:102-191`` (same constructor, same non-persistent
``quad_weight`` buffer, ``forward(x)`` = quadrature over the last two axes) and ``GeometricLpLoss`` mirrors
``makani/utils/losses/lp_loss.py:28-107`` (``abs`` / ``rel`` / ``forward(prd, tar, wgt)`` -> (B, C) norms).
The elementwise chain ``|prd - tar|^p * wgt * q`` and the plane reduction are ONE HIP kernel
(``mk_quad_lp_fwd``), its autograd one more (``mk_quad_lp_bwd``); prediction and target may have different
dtypes (bf16 prediction, fp32 target) without a cast pass.  Spatially distributed quadrature sums over the
"spatial" group of ``makani_amd.distributed``.
"""
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, legendre
from ._lib import device_guard, check, dtype_code, lib, ptr, stream

GRID_TO_QUADRATURE_RULE = {
    "euclidean": "uniform",
    "equiangular": "naive",
    "legendre-gauss": "legendre-gauss",
    "clenshaw-curtiss": "clenshaw-curtiss",
    "weatherbench2": "weatherbench2",
}


def grid_to_quadrature_rule(grid_type: str) -> str:
    """``makani/utils/grids.py:27-40``."""
    if grid_type not in GRID_TO_QUADRATURE_RULE:
        raise NotImplementedError(f"Grid type {grid_type} does not have a quadrature rule")
    return GRID_TO_QUADRATURE_RULE[grid_type]


def _quad_launch(a, b, wgt, q, mode, p):
    """sums (planes,) f32 of q * (mode ? |a-b|^p : a) * wgt over the trailing (H, W) plane."""
    planes = a.numel() // q.numel()
    hw = q.numel()
    ch = lib().mk_quad_lp_chunks(hw)
    sums = torch.empty((2, planes), dtype=torch.float32, device=a.device)
    ws = torch.empty((planes * ch * 2,), dtype=torch.float32, device=a.device)
    check(lib().mk_quad_lp_fwd(ptr(a), dtype_code(a), ptr(b) if b is not None else None,
                               dtype_code(b) if b is not None else _lib.MK_F32, ptr(wgt) if wgt is not None else None,
                               ptr(q), ptr(sums), ptr(ws), planes, hw, mode, float(p), stream()), "mk_quad_lp_fwd")
    return sums[0]


def _prep(t, ref_shape=None):
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    if ref_shape is not None and tuple(t.shape) != tuple(ref_shape):
        t = t.expand(ref_shape)
    return t.contiguous()


class QuadLpFn(torch.autograd.Function):
    """out[...] = sum_{h,w} q[h,w] * (mode ? |a - b|^p : a) * wgt;  a, b: (..., H, W)."""

    @staticmethod
    def forward(ctx, a, b, wgt, q, mode, p):
        if not a.is_cuda:
            raise RuntimeError("makani_amd losses run on the GPU (HIP) path only")
        a = _prep(a)
        b = _prep(b, a.shape) if b is not None else None
        w = _prep(wgt, a.shape).float() if wgt is not None else None
        ctx.save_for_backward(a, b, w, q)
        ctx.meta = (mode, float(p))
        return _quad_launch(a, b, w, q, mode, p).view(a.shape[:-2])

    @staticmethod
    def backward(ctx, g):
        a, b, w, q = ctx.saved_tensors
        mode, p = ctx.meta
        g = g.contiguous().float().view(-1)
        need_a, need_b = ctx.needs_input_grad[0], (b is not None and ctx.needs_input_grad[1])
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        if need_a or need_b:
            check(lib().mk_quad_lp_bwd(ptr(a), dtype_code(a), ptr(b) if b is not None else None,
                                       dtype_code(b) if b is not None else _lib.MK_F32,
                                       ptr(w) if w is not None else None, ptr(q), ptr(g),
                                       ptr(da) if need_a else None, ptr(db) if need_b else None,
                                       g.numel(), q.numel(), mode, p, stream()), "mk_quad_lp_bwd")
        return da, db, None, None, None, None


def _rule_weights(rule: str, img_shape) -> torch.Tensor:
    """(H, W) weights summing to 4 pi (before normalisation), ``grids.py:111-144``; same torch-fp32 arithmetic
    for the closed-form rules so the buffer is bit-identical to the reference's."""
    H, W = img_shape
    dlambda = 2 * math.pi / W
    if rule == "naive":
        jac = torch.clamp(torch.sin(torch.linspace(0, math.pi, H)), min=0.0)
        q = ((dlambda * (math.pi / H)) * jac.unsqueeze(1)).tile(1, W)
        return q * (4.0 * math.pi) / torch.sum(q)
    if rule in ("clenshaw-curtiss", "legendre-gauss"):
        _, w = legendre.clenshaw_curtis(H) if rule == "clenshaw-curtiss" else legendre.gauss_legendre(H)
        return (dlambda * torch.from_numpy(w.copy()).unsqueeze(1)).tile(1, W)
    if rule == "weatherbench2":
        lats = torch.linspace(0, math.pi, H)
        bounds = torch.cat([torch.zeros(1), (lats[:-1] + lats[1:]) / 2, torch.full((1,), math.pi)])
        jac = torch.cos(bounds[:-1]) - torch.cos(bounds[1:])
        return (dlambda * jac.unsqueeze(1)).tile(1, W)
    if rule == "uniform":
        q = torch.ones((H, W))
        return 4.0 * math.pi * q / torch.sum(q)
    raise ValueError(f"Unknown quadrature rule {rule}")


class GridQuadrature(nn.Module):
    def __init__(self, quadrature_rule, img_shape, crop_shape=None, crop_offset=(0, 0), normalize=False, distributed=False):
        super().__init__()
        from . import distributed as thd
        self.distributed = bool(distributed) and thd.ensure_initialized()
        crop_shape = img_shape if crop_shape is None else crop_shape
        q = _rule_weights(quadrature_rule, img_shape)
        if normalize:
            q = q / (4.0 * math.pi)
        h0, hl = crop_offset[0], crop_shape[0]
        w0, wl = crop_offset[1], crop_shape[1]
        if self.distributed:        # this rank's lat/lon shard of the crop (grids.py:150-168)
            if thd.polar_group_size() > 1:
                sh = thd.compute_split_shapes(crop_shape[0], thd.polar_group_size())
                h0, hl = h0 + sum(sh[: thd.polar_group_rank()]), sh[thd.polar_group_rank()]
            if thd.azimuth_group_size() > 1:
                sw = thd.compute_split_shapes(crop_shape[1], thd.azimuth_group_size())
                w0, wl = w0 + sum(sw[: thd.azimuth_group_rank()]), sw[thd.azimuth_group_rank()]
        q = q[h0:h0 + hl, w0:w0 + wl].contiguous()
        H, W = q.shape
        self.register_buffer("quad_weight", q.float().reshape(1, 1, H, W), persistent=False)

    def _reduce(self, quad):
        if self.distributed:
            from . import distributed as thd
            quad = thd.reduce_from_spatial_region(quad.contiguous())
        return quad

    @torch.compiler.disable(recursive=True)
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._reduce(QuadLpFn.apply(x, None, None, self.quad_weight, 0, 1.0).to(x.dtype))

    def lp(self, a, b, wgt, p):
        """quadrature of |a - b|^p * wgt without materialising the integrand (b None = 0)."""
        return self._reduce(QuadLpFn.apply(a, b, wgt, self.quad_weight, 1, p))


def compute_spherical_bandlimit(img_shape, grid_type):
    """``makani/utils/grids.py:43-55``."""
    if grid_type == "equiangular":
        return min((img_shape[0] - 1) // 2, img_shape[1] // 2)
    if grid_type == "legendre-gauss":
        return min(img_shape[0] - 1, img_shape[1] // 2)
    raise NotImplementedError(f"Unknown type {grid_type} not implemented")


class SpecLpFn(torch.autograd.Function):
    """out[row] = sum_{l >= m} w(m) |c_lm|^p (* wgt) over an S-layout tensor (L, M, 2, R)."""

    @staticmethod
    def forward(ctx, S, wgt, p, w0, w1, tri_off, m_off):
        S = S.contiguous()
        L, M, _, R = S.shape
        nb = lib().mk_spec_lp_blocks(L, M)
        partial = torch.empty((nb, R), dtype=torch.float32, device=S.device)
        check(lib().mk_spec_lp_fwd(ptr(S), ptr(wgt) if wgt is not None else None, ptr(partial), L, M, R, tri_off, m_off,
                                   float(p), float(w0), float(w1), stream()), "mk_spec_lp_fwd")
        ctx.save_for_backward(S, wgt)
        ctx.meta = (float(p), float(w0), float(w1), tri_off, m_off)
        return partial.sum(0)

    @staticmethod
    def backward(ctx, g):
        S, wgt = ctx.saved_tensors
        p, w0, w1, tri_off, m_off = ctx.meta
        L, M, _, R = S.shape
        dS = torch.empty_like(S)
        g = g.contiguous().float()
        check(lib().mk_spec_lp_bwd(ptr(S), ptr(wgt) if wgt is not None else None, ptr(g), ptr(dS), L, M, R, tri_off, m_off,
                                   p, w0, w1, stream()), "mk_spec_lp_bwd")
        return dS, None, None, None, None, None, None


class SpectralLpLoss(nn.Module):
    """Computes the Lp loss in spectral (SH coefficient) space (``lp_loss.py:110-259``, base class
    ``base_loss.py:345-404``): SHT of the difference on the HIP path, then ONE kernel for
    ``|c|^p``, the Parseval weights (m = 0 once, m > 0 twice, 1/4pi) and the sum over (l, m) — on the SHT's
    internal layout, without materialising complex coefficients."""

    def __init__(self, img_shape: Tuple[int, int], crop_shape: Tuple[int, int], crop_offset: Tuple[int, int],
                 channel_names: List[str], grid_type: str, p: Optional[float] = 2.0, relative: Optional[bool] = False,
                 squared: Optional[bool] = False, spatial_distributed: Optional[bool] = False,
                 eps: Optional[float] = 1.0e-6, lmax: Optional[int] = None, **kwargs):
        super().__init__()
        from . import distributed as thd
        from .sht import RealSHT
        self.img_shape, self.crop_shape, self.crop_offset = img_shape, crop_shape, crop_offset
        self.channel_names = channel_names
        self.spatial_distributed = bool(spatial_distributed) and thd.ensure_initialized()
        bandlimit = compute_spherical_bandlimit(img_shape, grid_type)
        if lmax is None or lmax > bandlimit:
            lmax = bandlimit
        if self.spatial_distributed:
            self.sht = thd.DistributedRealSHT(*img_shape, lmax=lmax, mmax=lmax, grid=grid_type)
            self._l_off, self._m_off = self.sht.l_off, self.sht.m_off
            l_loc = self.sht.l_shapes[self.sht.comm_rank_polar]
            m_loc = self.sht.m_shapes[self.sht.comm_rank_azimuth]
        else:
            self.sht = RealSHT(*img_shape, lmax=lmax, mmax=lmax, grid=grid_type).float()
            self._l_off = self._m_off = 0
            l_loc, m_loc = self.sht.lmax, self.sht.mmax
        m_weights = 2 * torch.ones(self.sht.mmax, dtype=torch.float32)
        m_weights[0] = 1.0
        m_weights = m_weights / (4.0 * math.pi)
        lm = torch.ones(self.sht.lmax, dtype=torch.float32)[:, None] * m_weights[None, :]
        lm = lm[self._l_off:self._l_off + l_loc, self._m_off:self._m_off + m_loc].contiguous()
        self.register_buffer("lm_weights", lm, persistent=False)
        self.p, self.relative, self.squared, self.eps = p, relative, squared, eps

    @property
    def n_channels(self):
        return len(self.channel_names)

    def _s_weights(self, wgt, B, C, L, M, R, device):
        """broadcastable (B, C, L, M) weights -> (L, M, R) in the S-layout's row order"""
        if wgt is None:
            return None
        w = wgt.to(device=device, dtype=torch.float32).expand(B, C, L, M)
        Cp = R // B
        out = torch.zeros((L, M, B, Cp), dtype=torch.float32, device=device)
        out[:, :, :, :C] = w.permute(2, 3, 0, 1)
        return out.view(L, M, R)

    def _norm_p(self, x, wgt, w0, w1):
        """sum_{l,m} w(m) |sht(x)|^p per (b, c)"""
        if x.dim() != 4:
            raise ValueError(f"expected (B, C, H, W), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("makani_amd losses run on the GPU (HIP) path only")
        B, C = x.shape[:2]
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        S = self.sht.analysis(x.contiguous())
        L, M, _, R = S.shape
        w3 = self._s_weights(wgt, B, C, L, M, R, x.device)
        rows = SpecLpFn.apply(S, w3, self.p, w0, w1, self._l_off - self._m_off, self._m_off)
        out = rows.view(B, R // B)[:, :C]
        if self.spatial_distributed:
            from . import distributed as thd
            out = thd.reduce_from_spatial_region(out.contiguous())
        return out

    def abs(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None):
        inv_area = 1.0 / (4.0 * math.pi)
        normp = self._norm_p(prd - tar, wgt, inv_area, 2.0 * inv_area)
        return normp if self.squared else normp.pow(1.0 / self.p)

    def rel(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None):
        normp = self._norm_p(prd - tar, wgt, 1.0, 2.0)
        tar_normp = self._norm_p(tar, wgt, 1.0, 2.0)
        if not self.squared:
            normp, tar_normp = normp.pow(1.0 / self.p), tar_normp.pow(1.0 / self.p)
        return normp / (tar_normp + self.eps)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None, **kwargs):
        return self.rel(prd, tar, wgt) if self.relative else self.abs(prd, tar, wgt)


class SpectralH1Loss(SpectralLpLoss):
    """H1 seminorm loss on the sphere (``makani/utils/losses/h1_loss.py:30-180``): the p = 2 spectral loss with every
    degree weighted by l (l + 1).  Same kernels as ``SpectralLpLoss``; the degree weights ride in its weight operand."""

    def __init__(self, img_shape: Tuple[int, int], crop_shape: Tuple[int, int], crop_offset: Tuple[int, int],
                 channel_names: List[str], grid_type: str, relative: Optional[bool] = False,
                 squared: Optional[bool] = False, spatial_distributed: Optional[bool] = False,
                 eps: Optional[float] = 1.0e-6, **kwargs):
        super().__init__(img_shape, crop_shape, crop_offset, channel_names, grid_type, p=2.0, relative=relative,
                         squared=squared, spatial_distributed=spatial_distributed, eps=eps)
        l = torch.arange(self.sht.lmax).float()
        h1 = (l * (l + 1))[self._l_off:self._l_off + self.lm_weights.shape[0]]
        self.register_buffer("h1_weights", h1.reshape(1, 1, -1), persistent=False)

    def _norm_p(self, x, wgt, w0, w1):
        h1 = self.h1_weights.reshape(1, 1, -1, 1)
        return super()._norm_p(x, h1 if wgt is None else wgt * h1, w0, w1)

    def abs(self, prd, tar, wgt=None):
        inv_area = 1.0 / (4.0 * math.pi)
        n2 = self._norm_p(prd - tar, wgt, inv_area, 2.0 * inv_area)
        return n2 if self.squared else torch.sqrt(n2)

    def rel(self, prd, tar, wgt=None):
        n2 = self._norm_p(prd - tar, wgt, 1.0, 2.0)
        t2 = self._norm_p(tar, wgt, 1.0, 2.0)
        if not self.squared:
            n2, t2 = torch.sqrt(n2), torch.sqrt(t2)
        return n2 / (t2 + self.eps)


class GeometricLpLoss(nn.Module):
    """Computes the Lp loss on the sphere (``lp_loss.py:28-107``)."""

    def __init__(self, img_shape: Tuple[int, int], crop_shape: Tuple[int, int], crop_offset: Tuple[int, int],
                 channel_names: List[str], p: Optional[float] = 2.0, relative: Optional[bool] = False,
                 squared: Optional[bool] = False, jacobian: Optional[str] = "s2",
                 grid_type: Optional[str] = "equiangular", spatial_distributed: Optional[bool] = False,
                 eps: Optional[float] = 1.0e-6, **kwargs):
        super().__init__()
        self.img_shape, self.crop_shape, self.crop_offset = img_shape, crop_shape, crop_offset
        self.channel_names = channel_names
        self.quadrature = GridQuadrature(grid_to_quadrature_rule(grid_type), img_shape=img_shape, crop_shape=crop_shape,
                                         crop_offset=crop_offset, normalize=True, distributed=spatial_distributed)
        self.spatial_distributed = self.quadrature.distributed
        self.p, self.relative, self.squared, self.eps = p, relative, squared, eps

    @property
    def n_channels(self):
        return len(self.channel_names)

    def abs(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None):
        n = prd.shape[0]
        norms = self.quadrature.lp(prd, tar, wgt, self.p).reshape(n, -1)
        if not self.squared:
            norms = norms.pow(1.0 / self.p)
        return norms

    def rel(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None):
        n = prd.shape[0]
        diff = self.quadrature.lp(prd, tar, wgt, self.p).reshape(n, -1)
        tarn = self.quadrature.lp(tar, None, wgt, self.p).reshape(n, -1)
        norms = diff / (tarn + self.eps)
        if not self.squared:
            norms = norms.pow(1.0 / self.p)
        return norms

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, prd: torch.Tensor, tar: torch.Tensor, wgt: Optional[torch.Tensor] = None, **kwargs):
        return self.rel(prd, tar, wgt) if self.relative else self.abs(prd, tar, wgt)


# --------------------------------------------------------------------------- #
# ensemble CRPS (makani/utils/losses/crps_loss.py:277-452)
# --------------------------------------------------------------------------- #
_CRPS_TYPES = {"skillspread": 0, "probability weighted moment": 1, "naive skillspread": 2, "gauss": 3, "cdf": 4}
MAX_ENSEMBLE = 32          # members of a grid point live in registers (csrc/crps.hip)


class CrpsFn(torch.autograd.Function):
    """out[b, c] = sum_p q[p] * w[b, c, p] * crps(obs[b, c, p], forecasts[b, :, c, p]); gradient w.r.t. the forecasts"""

    @staticmethod
    def forward(ctx, forecasts, obs, q, wgt, ctype, alpha, eps, ens_w=None):
        B, E, Cc, H, W = forecasts.shape
        if E > MAX_ENSEMBLE:
            raise NotImplementedError(f"ensemble size {E}: the HIP CRPS kernels hold the members of a point in registers "
                                      f"(2 <= E <= {MAX_ENSEMBLE})")
        hw = H * W
        f, o = _prep(forecasts), _prep(obs)
        w = wgt.float().contiguous() if wgt is not None else None
        ch = lib().mk_crps_chunks(hw)
        partial = torch.empty((B * Cc, ch), dtype=torch.float32, device=f.device)
        check(lib().mk_crps(ptr(f), dtype_code(f), ptr(o), dtype_code(o), ptr(q), ptr(w), None, ptr(partial), None, B, E, Cc, hw,
                            ctype, float(alpha), float(eps), 0, ptr(ens_w), stream()), "mk_crps")
        ctx.save_for_backward(f, o, q, w if w is not None else torch.empty(0, device=f.device))
        ctx.meta = (ctype, alpha, eps, w is not None, forecasts.dtype)
        ctx.ens_w = ens_w
        return partial.sum(dim=1).reshape(B, Cc)

    @staticmethod
    def backward(ctx, g):
        f, o, q, w = ctx.saved_tensors
        ctype, alpha, eps, has_w, dt = ctx.meta
        B, E, Cc, H, W = f.shape
        gf = torch.empty_like(f)
        go = g.float().contiguous()
        check(lib().mk_crps(ptr(f), dtype_code(f), ptr(o), dtype_code(o), ptr(q), ptr(w) if has_w else None, ptr(go), None,
                            ptr(gf), B, E, Cc, H * W, ctype, float(alpha), float(eps), 1, ptr(ctx.ens_w), stream()), "mk_crps")
        return gf.to(dt), None, None, None, None, None, None, None


class CrpsComplexFn(torch.autograd.Function):
    """the "naive skillspread" score of COMPLEX members (``mk_crps_complex``): forecasts (B, E, C, L, M) complex64,
    obs (B, C, L, M) complex64 -> (B, C); gradient with respect to the forecasts"""

    @staticmethod
    def forward(ctx, forecasts, obs, q, wgt, alpha):
        B, E, Cc, H, W = forecasts.shape
        if E > MAX_ENSEMBLE:
            raise NotImplementedError(f"ensemble size {E}: the HIP CRPS kernels hold the members of a point in registers "
                                      f"(2 <= E <= {MAX_ENSEMBLE})")
        hw = H * W
        f = torch.view_as_real(forecasts.to(torch.complex64).contiguous())
        o = torch.view_as_real(obs.to(torch.complex64).contiguous())
        w = wgt.float().contiguous() if wgt is not None else None
        ch = lib().mk_crps_chunks(hw)
        partial = torch.empty((B * Cc, ch), dtype=torch.float32, device=f.device)
        check(lib().mk_crps_complex(ptr(f), ptr(o), ptr(q), ptr(w), None, ptr(partial), None, B, E, Cc, hw, float(alpha), 0, stream()),
              "mk_crps_complex")
        ctx.save_for_backward(f, o, q, w if w is not None else torch.empty(0, device=f.device))
        ctx.meta = (alpha, w is not None)
        return partial.sum(dim=1).reshape(B, Cc)

    @staticmethod
    def backward(ctx, g):
        f, o, q, w = ctx.saved_tensors
        alpha, has_w = ctx.meta
        B, E, Cc, H, W, _ = f.shape
        gf = torch.empty_like(f)
        go = g.float().contiguous()
        check(lib().mk_crps_complex(ptr(f), ptr(o), ptr(q), ptr(w) if has_w else None, ptr(go), None, ptr(gf), B, E, Cc, H * W,
                                    float(alpha), 1, stream()), "mk_crps_complex")
        return torch.view_as_complex(gf), None, None, None, None


class _EnsembleTransposeFn(torch.autograd.Function):
    """``distributed_transpose(forecasts, (-1, 0), ensemble_shapes, "ensemble")`` of ``crps_loss.py:362-366,566-570``: every
    rank of the ensemble group holds E_loc members on all N points and ends up with ALL members on its share of the points
    (``compute_split_shapes(N, n)``).  x (B, E_loc, C, N) -> (B, E_loc * n, C, N_loc); backward is the reverse exchange."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        from . import distributed as thd
        n, me = dist.get_world_size(group), dist.get_rank(group)
        B, El, Cc, N = x.shape
        sizes = thd.compute_split_shapes(N, n)
        off = [0]
        for v in sizes:
            off.append(off[-1] + v)
        send = [x[..., off[r]:off[r + 1]].contiguous() for r in range(n)]
        recv = [torch.empty((B, El, Cc, sizes[me]), dtype=x.dtype, device=x.device) for _ in range(n)]
        thd._exchange(recv, send, group)
        ctx.meta = (group, n, me, sizes, off, N)
        return torch.cat(recv, dim=1)

    @staticmethod
    def backward(ctx, g):
        from . import distributed as thd
        group, n, me, sizes, off, N = ctx.meta
        B, E, Cc, Nl = g.shape
        El = E // n
        send = [g[:, r * El:(r + 1) * El].contiguous() for r in range(n)]
        recv = [torch.empty((B, El, Cc, sizes[r]), dtype=g.dtype, device=g.device) for r in range(n)]
        thd._exchange(recv, send, group)
        return torch.cat(recv, dim=3), None


def _ensemble_split(forecasts, obs, q, wgt):
    """the ensemble-parallel path of the CRPS losses: (forecasts with ALL members on this rank's share of the points, that
    share of the observations / quadrature weights / spatial weights, the group).  forecasts (B, E_loc, C, N) etc."""
    import torch.distributed as dist
    from . import comm as _comm
    from . import distributed as thd
    group = _comm.get_group("ensemble")
    n, me = dist.get_world_size(group), dist.get_rank(group)
    N = forecasts.shape[-1]
    sizes = thd.compute_split_shapes(N, n)
    a = sum(sizes[:me])
    b = a + sizes[me]
    f = _EnsembleTransposeFn.apply(forecasts, group)
    return f, obs[..., a:b].contiguous(), q[..., a:b].contiguous(), (wgt[..., a:b].contiguous() if wgt is not None else None), group


class _ReduceFromGroupFn(torch.autograd.Function):
    """``reduce_from_parallel_region``: SUM all-reduce forward, identity backward"""

    @staticmethod
    def forward(ctx, x, group):
        from . import ops
        y = x.clone()
        ops._all_reduce_sum(y, group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


def _ensemble_active(flag) -> bool:
    """``ensemble_distributed`` is honoured when the process-group tree names a split "ensemble" group.  The tree may not have
    been looked at yet (a loss constructed before any makani_amd network under makani's own driver): adopt makani's tree first;
    a set flag without such a group in a multi-rank job is reported — each rank would otherwise silently score its local
    members only."""
    if not flag:
        return False
    from . import comm as _comm
    _comm.autodetect()
    active = _comm.is_distributed("ensemble") and _comm.get_size("ensemble") > 1
    if not active and _comm.get_world_size() > 1:
        import warnings
        warnings.warn("ensemble_distributed=True, but the process-group tree has no split 'ensemble' group: the loss scores the "
                      "members of this rank only (makani_amd.comm.init(h, w, ensemble=n) or makani's own tree provides the group)")
    return active


def _check_finite_weights(w):
    """Non-finite ``ensemble_weights`` are rejected at construction.  Deviation from the reference, stated: its kernels fold
    ``isnan(weights)`` into their NaN mask (``crps_loss.py:66-73,176-177``), so a NaN weight — one entry per member, broadcast
    over every grid point — silently zeroes the whole score (PWM) or drops the member everywhere (cdf); here that is an error."""
    if w is not None and not bool(torch.isfinite(w).all()):
        raise ValueError("ensemble_weights must be finite (the reference would mask every grid point of a member with a NaN weight)")


def _ens_w(w, E, crps_type="cdf"):
    if w is not None and w.numel() != E:
        raise ValueError(f"ensemble_weights holds {w.numel()} entries for an ensemble of {E}")
    return w if crps_type == "cdf" else None          # only the cdf kernel reads the values (PWM accepts and ignores them)


class CRPSLoss(nn.Module):
    """``CRPSLoss`` of ``makani/utils/losses/crps_loss.py:277-452``: ``forward(forecasts (B, E, C, H, W), observations
    (B, C, H, W), spatial_weights=None) -> (B, C)``, the quadrature-weighted ensemble CRPS.  Score and quadrature are one HIP
    kernel (``csrc/crps.hip``), the gradient with respect to the forecasts one more.  Built: ``crps_type`` "skillspread"
    (default, with the almost-fair factor ``alpha``), "naive skillspread", "probability weighted moment", "gauss" and "cdf"
    (:55-122, with optional per-member ``ensemble_weights``; the "probability weighted moment" form accepts them too and, like the reference's kernel, ignores their values — finite values: non-finite weights raise, see ``_check_finite_weights``); any ensemble
    size 2..32; ``ensemble_distributed=True`` with a split "ensemble" group (``makani_amd.comm.init(h, w, ensemble=n)`` or
    makani's own tree) trades the members for a share of the grid points before scoring, as the reference does."""

    def __init__(self, img_shape: Tuple[int, int], crop_shape: Tuple[int, int], crop_offset: Tuple[int, int],
                 channel_names: List[str], grid_type: str, crps_type: str = "skillspread",
                 spatial_distributed: Optional[bool] = False, ensemble_distributed: Optional[bool] = False,
                 ensemble_weights: Optional[torch.Tensor] = None, alpha: Optional[float] = 1.0, eps: Optional[float] = 1.0e-6,
                 **kwargs):
        super().__init__()
        self.img_shape, self.crop_shape, self.crop_offset = img_shape, crop_shape, crop_offset
        self.channel_names = channel_names
        self.quadrature = GridQuadrature(grid_to_quadrature_rule(grid_type), img_shape=img_shape, crop_shape=crop_shape,
                                         crop_offset=crop_offset, normalize=True, distributed=spatial_distributed)
        self.spatial_distributed = self.quadrature.distributed
        self.ensemble_distributed = _ensemble_active(ensemble_distributed)                # crps_loss.py:305-307
        # the reference hands ensemble_weights to the "cdf" kernel (:392-396) and to the "probability weighted moment" kernel
        # (:404-409), which ignores their values: accepted for both (and ignored by the latter, as there); other forms raise
        if ensemble_weights is not None and crps_type not in ("cdf", "probability weighted moment"):
            raise NotImplementedError("currently only constant ensemble weights are supported")
        _check_finite_weights(ensemble_weights)
        if crps_type not in _CRPS_TYPES:
            raise ValueError(f"Unknown CRPS crps_type {crps_type}")
        if crps_type not in ("skillspread", "naive skillspread") and alpha < 1.0:
            raise NotImplementedError("The alpha parameter (almost fair CRPS factor) is only supported for the skillspread kernels.")
        self.crps_type, self.alpha, self.eps = crps_type, alpha, eps
        self.register_buffer("ensemble_weights", None if ensemble_weights is None else ensemble_weights.float().reshape(-1).contiguous(),
                             persistent=False)
        self.register_buffer("quad_weight_split", self.quadrature.quad_weight.reshape(1, 1, -1).contiguous(), persistent=False)

    @property
    def n_channels(self):
        return len(self.channel_names)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, forecasts: torch.Tensor, observations: torch.Tensor, spatial_weights: Optional[torch.Tensor] = None,
                **kwargs) -> torch.Tensor:
        if forecasts.dim() != 5:
            raise ValueError(f"Error, forecasts tensor expected to have 5 dimensions but found {forecasts.dim()}.")
        if spatial_weights is not None and spatial_weights.dim() != observations.dim():
            raise ValueError(f"the weights have to have the same number of dimensions (found {spatial_weights.dim()}) as "
                             f"observations (found {observations.dim()}).")
        B, E, Cc, H, W = forecasts.shape
        if E == 1 and not self.ensemble_distributed:            # |obs - forecast| under the quadrature (crps_loss.py:375-377)
            crps = self.quadrature.lp(forecasts.squeeze(1), observations, spatial_weights, 1.0).reshape(B, Cc)
            return crps
        w = spatial_weights.expand(B, Cc, H, W) if spatial_weights is not None else None
        q = self.quad_weight_split.reshape(-1)
        if self.ensemble_distributed:
            # members are spread over the ensemble group: trade them for a share of the grid points (crps_loss.py:362-373),
            # score that share with all members, sum the shares (:441-442)
            f, o, q, w, group = _ensemble_split(forecasts.reshape(B, E, Cc, H * W), observations.reshape(B, Cc, H * W), q,
                                                w.reshape(B, Cc, H * W) if w is not None else None)
            E = f.shape[1]
            crps = CrpsFn.apply(f.unsqueeze(-1), o.unsqueeze(-1), q, w.unsqueeze(-1) if w is not None else None,
                                _CRPS_TYPES[self.crps_type], self.alpha, self.eps, _ens_w(self.ensemble_weights, E, self.crps_type))
            return self.quadrature._reduce(_ReduceFromGroupFn.apply(crps, group))
        crps = CrpsFn.apply(forecasts, observations, q, w,
                            _CRPS_TYPES[self.crps_type], self.alpha, self.eps, _ens_w(self.ensemble_weights, E, self.crps_type))
        return self.quadrature._reduce(crps)


class SpectralCRPSLoss(SpectralLpLoss):
    """``SpectralCRPSLoss`` of ``makani/utils/losses/crps_loss.py:454-637`` (registered as "ensemble_spectral_crps"): the
    ensemble CRPS of the ABSOLUTE VALUES of the spherical-harmonic coefficients, summed with the Parseval weights of
    ``SpectralBaseLoss`` (m = 0 once, m > 0 twice, 1 / 4 pi).  ``forward(forecasts (B, E, C, H, W), observations (B, C, H, W),
    spectral_weights=None) -> (B, C)``.  The transforms are the HIP SHT (fp32, autocast off, as the reference), the per-(l, m)
    ensemble score and its weighted sum the HIP kernel of ``CRPSLoss`` (``csrc/crps.hip``) with the (l, m) plane in the place of
    the grid; ``absolute=False`` scores the complex coefficients themselves with the naive skill / spread kernel
    (``mk_crps_complex``); ``ensemble_distributed=True`` as in ``CRPSLoss`` (absolute values only)."""

    def __init__(self, img_shape: Tuple[int, int], crop_shape: Tuple[int, int], crop_offset: Tuple[int, int],
                 channel_names: List[str], grid_type: str, lmax: Optional[int] = None, crps_type: str = "skillspread",
                 spatial_distributed: Optional[bool] = False, ensemble_distributed: Optional[bool] = False,
                 ensemble_weights: Optional[torch.Tensor] = None, absolute: Optional[bool] = True, alpha: Optional[float] = 1.0,
                 eps: Optional[float] = 1.0e-6, **kwargs):
        super().__init__(img_shape, crop_shape, crop_offset, channel_names, grid_type, spatial_distributed=spatial_distributed,
                         lmax=lmax)
        self.ensemble_distributed = _ensemble_active(ensemble_distributed)
        if ensemble_weights is not None and crps_type != "cdf":
            raise NotImplementedError("currently only constant ensemble weights are supported")
        _check_finite_weights(ensemble_weights)
        if crps_type not in ("cdf", "skillspread", "probability weighted moment", "gauss"):     # what the reference's forward knows
            raise ValueError(f"Unknown CRPS crps_type {crps_type}")
        if crps_type not in ("skillspread", "naive skillspread") and alpha < 1.0:
            raise NotImplementedError("The alpha parameter (almost fair CRPS factor) is only supported for the skillspread kernels.")
        if not absolute and crps_type != "skillspread":              # crps_loss.py:540-545
            raise ValueError(f"the non-absolute path only works with the naive 'skillspread' CRPS kernel, but got crps_type {crps_type}")
        self.crps_type, self.alpha, self.eps, self.absolute = crps_type, alpha, eps, absolute
        self.register_buffer("ensemble_weights", None if ensemble_weights is None else ensemble_weights.float().reshape(-1).contiguous(),
                             persistent=False)

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, forecasts: torch.Tensor, observations: torch.Tensor, spectral_weights: Optional[torch.Tensor] = None,
                **kwargs) -> torch.Tensor:
        from . import distributed as thd
        if forecasts.dim() != 5:
            raise ValueError(f"Error, forecasts tensor expected to have 5 dimensions but found {forecasts.dim()}.")
        if spectral_weights is not None and spectral_weights.dim() != observations.dim():
            raise ValueError("the weights have to have the same number of dimensions as observations")
        dtype = forecasts.dtype
        with torch.autocast(device_type=forecasts.device.type, enabled=False):
            f = self.sht(forecasts.float()) / math.sqrt(4.0 * math.pi)
            o = self.sht(observations.float()) / math.sqrt(4.0 * math.pi)
        if self.absolute:
            f, o = torch.abs(f).to(dtype), torch.abs(o).to(dtype)
        B, E, Cc, L, M = f.shape
        if self.ensemble_distributed:            # crps_loss.py:566-581: members <-> (l, m) points over the ensemble group
            if not self.absolute:
                raise NotImplementedError("the ensemble-parallel spectral CRPS is built for absolute=True")
            w = (spectral_weights.expand(B, Cc, L, M).reshape(B, Cc, L * M) if spectral_weights is not None else None)
            fe, oe, qe, we, group = _ensemble_split(f.reshape(B, E, Cc, L * M), o.reshape(B, Cc, L * M),
                                                    self.lm_weights.reshape(-1).contiguous(), w)
            crps = CrpsFn.apply(fe.unsqueeze(-1), oe.unsqueeze(-1), qe, we.unsqueeze(-1) if we is not None else None,
                                _CRPS_TYPES[self.crps_type], self.alpha, self.eps, _ens_w(self.ensemble_weights, fe.shape[1]))
            crps = _ReduceFromGroupFn.apply(crps, group)
            return thd.reduce_from_spatial_region(crps) if self.spatial_distributed else crps
        if not self.absolute and E > 1:        # the naive kernel on the complex coefficients themselves (crps_loss.py:605-608)
            w = spectral_weights.expand(B, Cc, L, M) if spectral_weights is not None else None
            crps = CrpsComplexFn.apply(f, o, self.lm_weights.reshape(-1).contiguous(), w, self.alpha)
            return thd.reduce_from_spatial_region(crps) if self.spatial_distributed else crps
        if E == 1:
            w = self.lm_weights if spectral_weights is None else spectral_weights * self.lm_weights
            crps = (torch.abs(o - f.squeeze(1)).float() * w).reshape(B, Cc, L * M).sum(dim=-1)
        else:
            w = spectral_weights.expand(B, Cc, L, M) if spectral_weights is not None else None
            crps = CrpsFn.apply(f.contiguous(), o.contiguous(), self.lm_weights.reshape(-1).contiguous(), w, _CRPS_TYPES[self.crps_type],
                                self.alpha, self.eps, _ens_w(self.ensemble_weights, E))
        if self.spatial_distributed:
            crps = thd.reduce_from_spatial_region(crps)
        return crps
