"""CPU restatement of the reference's SFNO hot path in plain torch ops.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Each class names the
reference code it follows; parameter names / shapes are the reference's, so a
reference ``state_dict`` loads unchanged.  Pinned by ``tests/golden/*.npz``
(produced by the reference's own modules, ``oracle/make_golden.py``).

Runs in fp32 or fp64 on the CPU; bf16 autocast behaviour is emulated only where
a test asks for it (``amp_dtype``), mirroring the dtype flow documented in
SURVEY.md §3.2: 1x1 convs + GELU in bf16, instance norm keeps its input dtype,
SHT / contraction / iSHT in fp32 with the result cast back.
"""

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sht import RealSHT, InverseRealSHT


def contract_lwise(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``makani/models/common/contractions.py:23-24`` (dhconv)."""
    return torch.einsum("bgixy,giox->bgoxy", x, w)


def contract_lmwise(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``makani/models/common/contractions.py:19-20`` (diagonal)."""
    return torch.einsum("bgixy,gioxy->bgoxy", x, w)


def contract_sep_lmwise(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``makani/models/common/contractions.py:26-27`` (separable diagonal)."""
    return torch.einsum("bgixy,gixy->bgixy", x, w)


def contract_sep_lwise(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """``makani/models/common/contractions.py:30-31`` (separable dhconv)."""
    return torch.einsum("bgixy,gix->bgixy", x, w)


class SpectralConv(nn.Module):
    """``makani/models/common/spectral_convolution.py:116-264``."""

    def __init__(self, forward_transform, inverse_transform, in_channels, out_channels, num_groups=1,
                 operator_type="dhconv", separable=False, bias=False, gain=1.0):
        super().__init__()
        if in_channels % num_groups != 0:
            raise ValueError("in_channels must be divisible by num_groups")
        if out_channels % num_groups != 0:
            raise ValueError("out_channels must be divisible by num_groups")
        self.separable = separable
        self.forward_transform = forward_transform
        self.inverse_transform = inverse_transform
        self.in_channels, self.out_channels, self.num_groups = in_channels, out_channels, num_groups
        self.modes_lat = inverse_transform.lmax
        self.modes_lon = inverse_transform.mmax
        self.scale_residual = (forward_transform.nlat != inverse_transform.nlat) or (
            forward_transform.nlon != inverse_transform.nlon
        ) or (forward_transform.grid != inverse_transform.grid)
        self.operator_type = operator_type
        shape = [num_groups, in_channels // num_groups]
        if not separable:
            shape += [out_channels // num_groups]
        if operator_type == "diagonal":
            shape += [self.modes_lat, self.modes_lon]
        elif operator_type == "dhconv":
            shape += [self.modes_lat]
        else:
            raise ValueError(f"Unsupported operator type f{operator_type}")
        scale = math.sqrt(gain / (in_channels // num_groups)) * torch.ones(self.modes_lat, dtype=torch.complex64)
        scale[0] *= math.sqrt(2.0)
        # NB: broadcasts against the LAST weight dim exactly as the reference does
        # (spectral_convolution.py:189-193); for "diagonal" that needs lmax == mmax.
        self.weight = nn.Parameter(scale * torch.randn(*shape, dtype=torch.complex64))
        if bias:
            self.bias = nn.Parameter(torch.zeros(1, out_channels, 1, 1))

    def forward(self, x):
        dtype = x.dtype
        residual = x
        cdt = torch.float64 if dtype == torch.float64 else torch.float32
        x = x.to(cdt)
        x = self.forward_transform(x).contiguous()
        if self.scale_residual:
            residual = self.inverse_transform(x).to(dtype)
        B, C, H, W = x.shape
        x = x.reshape(B, self.num_groups, C // self.num_groups, H, W)
        w = self.weight.to(x.dtype)
        if self.separable:      # dispatch of contractions.py:35-54
            xp = contract_sep_lwise(x, w) if self.operator_type == "dhconv" else contract_sep_lmwise(x, w)
        else:
            xp = contract_lwise(x, w) if self.operator_type == "dhconv" else contract_lmwise(x, w)
        x = xp.reshape(B, self.out_channels, H, W).contiguous()
        x = self.inverse_transform(x).to(dtype)
        if hasattr(self, "bias"):
            x = x + self.bias.to(dtype)
        return x, residual


class _Filter(nn.Module):
    """``SpectralFilterLayer`` with ``filter_type="linear"`` (``sfnonet.py:52-166``)."""

    def __init__(self, fwd, inv, embed_dim, operator_type, bias, gain):
        super().__init__()
        self.filter = SpectralConv(fwd, inv, embed_dim, embed_dim, operator_type=operator_type, bias=bias, gain=gain)

    def forward(self, x):
        return self.filter(x)


class _Seq(nn.Module):
    """holder producing the reference's ``<name>.fwd.<idx>.weight`` keys."""

    def __init__(self, *mods):
        super().__init__()
        self.fwd = nn.Sequential(*mods)

    def forward(self, x):
        return self.fwd(x)


def _mlp(in_features, hidden, act, gain):
    """``makani/models/common/layers.py:768-823`` (nchw, drop_rate 0)."""
    fc1 = nn.Conv2d(in_features, hidden, 1, bias=True)
    fc2 = nn.Conv2d(hidden, in_features, 1, bias=True)
    nn.init.normal_(fc1.weight, std=math.sqrt(2.0 / in_features))
    nn.init.constant_(fc1.bias, 0.0)
    nn.init.normal_(fc2.weight, std=math.sqrt(gain / hidden))
    nn.init.constant_(fc2.bias, 0.0)
    return _Seq(fc1, act(), nn.Identity(), fc2, nn.Identity())


def _encdec(num_layers, inp, out, hidden, act, gain=1.0):
    """``makani/models/common/layers.py:603-643``."""
    mods, cur = [], inp
    for _ in range(num_layers):
        c = nn.Conv2d(cur, hidden, 1, bias=True)
        nn.init.normal_(c.weight, std=math.sqrt(2.0 / cur))
        nn.init.constant_(c.bias, 0.0)
        mods += [c, act()]
        cur = hidden
    c = nn.Conv2d(cur, out, 1, bias=False)
    nn.init.normal_(c.weight, std=math.sqrt(gain / cur))
    mods.append(c)
    return _Seq(*mods)


class GeometricInstanceNormS2(nn.Module):
    """``makani/models/common/layer_norm.py:30-160`` (serial): instance norm with normalised quadrature weights."""

    def __init__(self, img_shape, grid_type, num_features, eps=1e-6):
        super().__init__()
        from . import losses as _ol
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("quad_weight", _ol.quadrature_weights(_ol.GRID_TO_RULE[grid_type], tuple(img_shape), normalize=True),
                             persistent=False)

    def forward(self, x):
        from . import losses as _ol
        return _ol.geometric_instance_norm_s2(x, self.quad_weight, self.weight, self.bias, self.eps)


def _instance_norm(embed_dim):
    return nn.InstanceNorm2d(embed_dim, eps=1e-6, affine=True, track_running_stats=False)


class ChannelLayerNorm(nn.Module):
    """``DistributedLayerNorm`` (``makani/mpu/layer_norm.py:256-290``): nn.LayerNorm over the channels of an NCHW tensor"""

    def __init__(self, embed_dim, eps=1e-6):
        super().__init__()
        self.norm = nn.LayerNorm(embed_dim, eps=eps, elementwise_affine=True)

    def forward(self, x):
        return torch.transpose(self.norm(torch.transpose(x, 1, 3)), 1, 3).contiguous()


class NeuralOperatorBlock(nn.Module):
    """``makani/models/networks/sfnonet.py:169-408`` with the SFNO settings
    ``inner_skip="none"``, ``outer_skip="linear"``, ``use_mlp=True``."""

    def __init__(self, fwd, inv, embed_dim, operator_type, mlp_ratio, act, bias, norm_layer=None, use_mlp=True):
        super().__init__()
        norm_layer = norm_layer or (lambda: _instance_norm(embed_dim), lambda: _instance_norm(embed_dim))
        self.norm0 = norm_layer[0]()
        gain = 2.0 if act != nn.Identity else 1.0
        self.filter = _Filter(fwd, inv, embed_dim, operator_type, bias, gain)
        self.act_layer0 = act()
        self.norm1 = norm_layer[1]()
        gain = 1.0
        self.outer_skip = nn.Conv2d(embed_dim, embed_dim, 1, 1, bias=False)
        gain /= 2.0
        nn.init.normal_(self.outer_skip.weight, std=math.sqrt(gain / embed_dim))
        if use_mlp:
            self.mlp = _mlp(embed_dim, int(embed_dim * mlp_ratio), act, gain)

    def forward(self, x):
        x, residual = self.filter(x)
        x = self.norm0(x)
        x = self.act_layer0(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        x = self.norm1(x)
        x = x + self.outer_skip(residual)
        return x


class SphericalFourierNeuralOperatorNet(nn.Module):
    """``makani/models/networks/sfnonet.py:411-934`` restricted to the
    BASELINE configuration family: ``spectral_transform="sht"``,
    ``filter_type="linear"``, ``normalization_layer`` "instance_norm" / "instance_norm_s2" / "none",
    ``pos_embed="none"``, drop rates 0, serial (no model parallelism)."""

    def __init__(self, model_grid_type="equiangular", sht_grid_type="legendre-gauss", operator_type="dhconv",
                 inp_shape=(721, 1440), out_shape=(721, 1440), scale_factor=8, inp_chans=2, out_chans=2,
                 embed_dim=32, num_layers=4, mlp_ratio=2.0, encoder_ratio=1, decoder_ratio=1,
                 activation_function="gelu", encoder_layers=1, hard_thresholding_fraction=1.0, max_modes=None,
                 big_skip=True, bias=False, normalization_layer="instance_norm", pos_embed="none", use_mlp=True, **kwargs):
        super().__init__()
        self.inp_shape, self.out_shape = tuple(inp_shape), tuple(out_shape)
        self.inp_chans, self.out_chans, self.embed_dim, self.big_skip = inp_chans, out_chans, embed_dim, big_skip
        self.h = int(inp_shape[0] // scale_factor)
        self.w = int(inp_shape[1] // scale_factor)
        if max_modes is not None:
            ml, mm = max_modes
        else:
            ml = int(self.h * hard_thresholding_fraction)
            mm = int((self.w // 2 + 1) * hard_thresholding_fraction)
        self.trans_down = RealSHT(*inp_shape, lmax=ml, mmax=mm, grid=model_grid_type)
        self.itrans_up = InverseRealSHT(*out_shape, lmax=ml, mmax=mm, grid=model_grid_type)
        self.trans = RealSHT(self.h, self.w, lmax=ml, mmax=mm, grid=sht_grid_type)
        self.itrans = InverseRealSHT(self.h, self.w, lmax=ml, mmax=mm, grid=sht_grid_type)
        act = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU}[activation_function]
        self.encoder = _encdec(encoder_layers, inp_chans, embed_dim, int(encoder_ratio * embed_dim), act)
        # norm layers per block position, sfnonet.py:609-673
        if normalization_layer == "instance_norm":
            n_inp = n_mid = n_out = lambda: _instance_norm(embed_dim)
        elif normalization_layer == "instance_norm_s2":
            n_inp = n_mid = lambda: GeometricInstanceNormS2((self.h, self.w), model_grid_type, embed_dim)
            n_out = lambda: GeometricInstanceNormS2(tuple(out_shape), model_grid_type, embed_dim)
        elif normalization_layer == "layer_norm":
            n_inp = n_mid = n_out = lambda: ChannelLayerNorm(embed_dim)
        elif normalization_layer == "none":
            n_inp = n_mid = n_out = nn.Identity
        else:
            raise NotImplementedError(f"Error, normalization {normalization_layer} not implemented.")
        self.blocks = nn.ModuleList()
        for i in range(num_layers):
            fwd = self.trans_down if i == 0 else self.trans
            inv = self.itrans_up if i == num_layers - 1 else self.itrans
            norms = (n_inp, n_mid) if i == 0 else ((n_out, n_out) if i == num_layers - 1 else (n_mid, n_mid))
            self.blocks.append(NeuralOperatorBlock(fwd, inv, embed_dim, operator_type, mlp_ratio, act, bias, norm_layer=norms,
                                                   use_mlp=use_mlp))
        self.decoder = _encdec(encoder_layers, embed_dim, out_chans, int(decoder_ratio * embed_dim), act,
                               gain=0.5 if big_skip else 1.0)
        if big_skip:
            self.residual_transform = nn.Conv2d(inp_chans, out_chans, 1, bias=False)
            nn.init.normal_(self.residual_transform.weight, std=math.sqrt(0.5 / inp_chans))
        # learned position embedding, sfnonet.py:732-764
        if pos_embed == "direct":
            self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *self.inp_shape))
            self.pos_embed.type = "direct"
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
        elif pos_embed == "frequency":
            rc = nn.Parameter(torch.tril(torch.randn(1, embed_dim, ml, mm), diagonal=0))
            cc = nn.Parameter(torch.tril(torch.randn(1, embed_dim, ml, mm - 1), diagonal=-1))
            nn.init.trunc_normal_(rc, std=0.02)
            nn.init.trunc_normal_(cc, std=0.02)
            self.pos_embed = nn.ParameterList([rc, cc])
            self.pos_embed.type = "frequency"
        elif pos_embed not in ("none", "None", None):
            raise ValueError("Unknown position embedding type")

    def _pos_embed(self, x):
        """sfnonet.py:898-911"""
        if self.pos_embed.type == "frequency":
            pe = torch.stack([self.pos_embed[0], F.pad(self.pos_embed[1], (1, 0), "constant", 0)], dim=-1)
            pe = self.itrans_up(torch.view_as_complex(pe))
        else:
            pe = self.pos_embed
        return x + pe.to(dtype=x.dtype)

    def forward(self, x):
        if self.big_skip:
            if self.out_shape != self.inp_shape:
                residual = self.itrans_up(self.trans_down(x.float()).contiguous()).to(x.dtype)
            else:
                residual = x
        x = self.encoder(x)
        if hasattr(self, "pos_embed"):
            x = self._pos_embed(x)
        for blk in self.blocks:
            x = blk(x)
        x = self.decoder(x)
        if self.big_skip:
            x = x + self.residual_transform(residual)
        return x


def l2_loss(pred: torch.Tensor, tar: torch.Tensor, quad_w: torch.Tensor) -> torch.Tensor:
    """Quadrature-weighted squared L2 averaged over batch and channels: the
    shape of the reference's ``"l2"`` training loss
    (``makani/utils/losses/lp_loss.py:61-75`` with ``squared=True``;
    ``quad_w`` from ``GridQuadrature``, ``makani/utils/grids.py:102-191``,
    normalised to sum to 1)."""
    d = (pred.float() - tar.float()) ** 2
    return torch.mean(torch.sum(d * quad_w, dim=(-2, -1)))
