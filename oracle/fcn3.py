"""CPU restatement of the reference's FourCastNet3 network (``makani/models/networks/fourcastnet3.py``) in plain torch ops on
the oracle's own operators: DISCO convolution / ``ResampleS2`` (``oracle/disco.py``), SHT pair (``oracle/sht.py``), spectral
convolution, pointwise MLPs and norms (``oracle/sfno.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Serial, drop rates 0, no checkpointing.  Parameter names and shapes are the
reference's, so the state dicts of ``tests/golden/fcn3_*.npz`` — written by the reference's OWN module running on the same restated
operators (``oracle/make_golden.py::fcn3_fixtures``) — load strictly; ``tests/test_oracle_fcn3.py`` pins forward output, input
gradient and every parameter gradient of this file against them.  (The operators underneath are the "parity unpinned" part, see
``oracle/disco.py``; what is pinned here is the network code: channel grouping, encoders / decoders, global and local blocks,
layer scale, big skip, water clamp.)  Used by ``bench.py``'s ``cpu_baseline`` leg for the FourCastNet3 line.
"""
import math
import re
from collections import OrderedDict

import torch
import torch.nn as nn

from . import disco as od
from .sfno import ChannelLayerNorm, SpectralConv, _encdec, _instance_norm, _Seq
from .sht import InverseRealSHT, RealSHT


def get_channel_groups(channel_names, aux_channel_names=()):
    """``makani/utils/features.py:97-140``: atmospheric variables are ``<letters><pressure level>`` (except "d2"), grouped by
    level in order of first appearance; everything else is a surface variable; auxiliary channels follow the state channels,
    static ones ("xoro", "xlsml", "xlsms") after the dynamic ones"""
    groups, surf = OrderedDict(), []
    for idx, chn in enumerate(channel_names):
        if re.search("[a-z]{1,3}[0-9]{1,4}$", chn) is not None and chn != "d2":
            groups.setdefault(int(re.search("[0-9]{1,4}$", chn).group()), []).append(idx)
        else:
            surf.append(idx)
    sizes = {len(v) for v in groups.values()}
    if len(sizes) > 1:
        raise ValueError(f"expected all atmospheric pressure level groups to have the same number of channels, got {sorted(sizes)}")
    atmo = [i for v in groups.values() for i in v]
    dyn, stat = [], []
    for idx, chn in enumerate(aux_channel_names):
        (stat if chn in ("xoro", "xlsml", "xlsms") else dyn).append(idx + len(channel_names))
    return atmo, surf, dyn, stat, list(groups.keys())


def get_water_channels(channel_names):
    """``makani/utils/features.py:70-80``: names that start with "q" or "r", and "tcwv" """
    return [i for i, c in enumerate(channel_names) if c[0] in ("q", "r") or c == "tcwv"]


def compute_cutoff_radius(nlat, kernel_shape, basis_type):
    """``fourcastnet3.py:46-50``"""
    factor = {"piecewise linear": 0.5, "morlet": 0.5, "harmonic": 0.5, "zernike": math.sqrt(2.0)}
    return (kernel_shape[0] + 1) * factor[basis_type] * math.pi / float(nlat - 1)


def soft_clamp(x, offset=0.0):
    """``fourcastnet3.py:55-59``: 0 below 0, x^2 up to 1/2, x - 1/4 above"""
    x = x + offset
    y = torch.where(x > 0.0, x ** 2, 0.0)
    return torch.where(x >= 0.5, x - 0.25, y)


def _mlp(in_features, out_features, hidden, act, gain):
    """``makani/models/common/layers.py:727-823`` (nchw, drop rate 0): fc1 - act - drop - fc2 - drop"""
    fc1 = nn.Conv2d(in_features, hidden, 1, bias=True)
    fc2 = nn.Conv2d(hidden, out_features, 1, bias=True)
    nn.init.normal_(fc1.weight, std=math.sqrt(2.0 / in_features))
    nn.init.constant_(fc1.bias, 0.0)
    nn.init.normal_(fc2.weight, std=math.sqrt(gain / hidden))
    nn.init.constant_(fc2.bias, 0.0)
    return _Seq(fc1, act(), nn.Identity(), fc2, nn.Identity())


class LayerScale(nn.Module):
    """``makani/models/common/layers.py:154-196``: one learned factor per channel (a depthwise 1x1 convolution), initially 0.1"""

    def __init__(self, num_chans, init_value=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.full((num_chans, 1, 1, 1), init_value))

    def forward(self, x):
        return x * self.weight.reshape(1, -1, 1, 1).to(x.dtype)


def _norm_handle(embed_dim, normalization_layer):
    """``fourcastnet3.py:62-113``.  ("instance_norm_s2" cannot be constructed by the reference itself: it passes ``pole_mask=`` to
    a GeometricInstanceNormS2 that has no such argument)"""
    if normalization_layer == "layer_norm":
        return lambda: ChannelLayerNorm(embed_dim)
    if normalization_layer == "instance_norm":
        return lambda: _instance_norm(embed_dim)
    if normalization_layer == "none":
        return nn.Identity
    raise NotImplementedError(f"Error, normalization {normalization_layer} not implemented.")


class DiscreteContinuousEncoder(nn.Module):
    """``fourcastnet3.py:116-244``: DISCO convolution from the data grid to the model grid (+ activation and a pointwise MLP)"""

    def __init__(self, inp_shape, out_shape, grid_in, grid_out, inp_chans, out_chans, kernel_shape, basis_type, basis_norm_mode,
                 use_mlp=False, mlp_ratio=2.0, activation_function=nn.GELU, groups=1, bias=False):
        super().__init__()
        self.conv = od.DiscreteContinuousConvS2(inp_chans, out_chans, in_shape=inp_shape, out_shape=out_shape, kernel_shape=kernel_shape,
                                                basis_type=basis_type, basis_norm_mode=basis_norm_mode, grid_in=grid_in, grid_out=grid_out,
                                                groups=groups, bias=bias,
                                                theta_cutoff=compute_cutoff_radius(inp_shape[0], kernel_shape, basis_type))
        if use_mlp:
            with torch.no_grad():
                self.conv.weight *= math.sqrt(2.0)
            self.act = activation_function()
            self.mlp = _encdec(1, out_chans, out_chans, int(mlp_ratio * out_chans), activation_function)

    def forward(self, x):
        x = self.conv(x)
        if hasattr(self, "act"):
            x = self.act(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        return x


class DiscreteContinuousDecoder(nn.Module):
    """``fourcastnet3.py:247-416``: (activation, MLP,) up-sampling to the data grid in fp32 — bilinear ``ResampleS2`` or an SHT
    round trip — and a DISCO convolution on the data grid; the convolution never has a bias (``:372``)"""

    def __init__(self, inp_shape, out_shape, grid_in, grid_out, inp_chans, out_chans, kernel_shape, basis_type, basis_norm_mode,
                 use_mlp=False, mlp_ratio=2.0, activation_function=nn.GELU, groups=1, bias=False, upsample_sht=False):
        super().__init__()
        if use_mlp:
            self.mlp = _encdec(1, inp_chans, inp_chans, int(mlp_ratio * inp_chans), activation_function, gain=2.0)
            self.act = activation_function()
        if upsample_sht:
            self.sht = RealSHT(*inp_shape, grid=grid_in)
            self.isht = InverseRealSHT(*out_shape, lmax=self.sht.lmax, mmax=self.sht.mmax, grid=grid_out)
            self.upsample = nn.Sequential(self.sht, self.isht)
        else:
            self.upsample = od.ResampleS2(*inp_shape, *out_shape, grid_in=grid_in, grid_out=grid_out, mode="bilinear")
        self.conv = od.DiscreteContinuousConvS2(inp_chans, out_chans, in_shape=out_shape, out_shape=out_shape, kernel_shape=kernel_shape,
                                                basis_type=basis_type, basis_norm_mode=basis_norm_mode, grid_in=grid_out, grid_out=grid_out,
                                                groups=groups, bias=False,
                                                theta_cutoff=compute_cutoff_radius(out_shape[0], kernel_shape, basis_type))

    def forward(self, x):
        dtype = x.dtype
        if hasattr(self, "act"):
            x = self.act(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        x = self.conv(self.upsample(x.float()))
        return x.to(dtype)


class NeuralOperatorBlock(nn.Module):
    """``fourcastnet3.py:419-640``: norm1 - local (DISCO, cutoff doubled) or global (spectral, "dhconv") convolution - norm2 - MLP -
    layer scale, added to the first ``out_chans`` channels of the block input (the auxiliary embedding rides behind them)"""

    def __init__(self, forward_transform, inverse_transform, inp_chans, out_chans, conv_type="local", mlp_ratio=2.0, act_layer=nn.GELU,
                 normalization_layer="none", num_groups=1, skip="identity", layer_scale=True, use_mlp=False, kernel_shape=(3, 3),
                 basis_type="morlet", basis_norm_mode="mean", bias=False):
        super().__init__()
        self.inp_shape = (forward_transform.nlat, forward_transform.nlon)
        self.out_shape = (inverse_transform.nlat, inverse_transform.nlon)
        self.out_chans = out_chans
        if conv_type == "local":
            self.local_conv = od.DiscreteContinuousConvS2(
                inp_chans, inp_chans, in_shape=self.inp_shape, out_shape=self.out_shape, kernel_shape=kernel_shape, basis_type=basis_type,
                basis_norm_mode=basis_norm_mode, groups=num_groups, grid_in=forward_transform.grid, grid_out=inverse_transform.grid,
                bias=False, theta_cutoff=2 * compute_cutoff_radius(self.inp_shape[0], kernel_shape, basis_type))
        elif conv_type == "global":
            self.global_conv = SpectralConv(forward_transform, inverse_transform, inp_chans, inp_chans, operator_type="dhconv",
                                            num_groups=num_groups, bias=bias, gain=1.0)
        else:
            raise ValueError(f"Unknown convolution type {conv_type}")
        handle = _norm_handle(inp_chans, normalization_layer)
        self.norm1 = handle()
        self.norm2 = handle()
        if use_mlp:
            self.mlp = _mlp(inp_chans, out_chans, int(inp_chans * mlp_ratio), act_layer, 1.0)
        self.layer_scale = LayerScale(out_chans) if layer_scale else nn.Identity()
        if skip == "linear":
            self.skip = nn.Conv2d(inp_chans, out_chans, 1, 1, bias=False)
            nn.init.normal_(self.skip.weight, std=math.sqrt(1.0 / inp_chans))
        elif skip == "identity":
            self.skip = nn.Identity()
        elif skip != "none":
            raise ValueError(f"Unknown skip connection type {skip}")

    def forward(self, x):
        x = self.norm1(x)
        if hasattr(self, "global_conv"):
            dx, _ = self.global_conv(x)
        else:
            dx = self.local_conv(x)
        dx = self.norm2(dx)
        if hasattr(self, "mlp"):
            dx = self.mlp(dx)
        if hasattr(self, "skip"):
            return self.skip(x[..., : self.out_chans, :, :]) + self.layer_scale(dx)
        return dx


class AtmoSphericNeuralOperatorNet(nn.Module):
    """``fourcastnet3.py:643-1165``"""

    def __init__(self, model_grid_type="equiangular", sht_grid_type="legendre-gauss", inp_shape=(721, 1440), out_shape=(721, 1440),
                 kernel_shape=(3, 3), filter_basis_type="morlet", filter_basis_norm_mode="mean", scale_factor=8, encoder_mlp=False,
                 upsample_sht=False, channel_names=("u500", "v500"), aux_channel_names=(), n_history=0, atmo_embed_dim=8,
                 surf_embed_dim=8, aux_embed_dim=8, num_layers=4, num_groups=1, use_mlp=True, mlp_ratio=2.0, activation_function="gelu",
                 layer_scale=True, normalization_layer="none", max_modes=None, hard_thresholding_fraction=1.0, sfno_block_frequency=2,
                 big_skip=False, clamp_water=False, bias=False, **kwargs):
        super().__init__()
        if n_history != 0:
            raise ValueError(f"this model currently does not support history, expected n_history == 0 but got {n_history}")
        self.inp_shape, self.out_shape = tuple(inp_shape), tuple(out_shape)
        self.atmo_embed_dim, self.surf_embed_dim, self.aux_embed_dim, self.big_skip = atmo_embed_dim, surf_embed_dim, aux_embed_dim, big_skip
        self.h, self.w = int(inp_shape[0] // scale_factor), int(inp_shape[1] // scale_factor)
        if max_modes is not None:
            modes_lat, modes_lon = max_modes
        else:
            modes_lat = int(self.h * hard_thresholding_fraction)
            modes_lon = int((self.w // 2 + 1) * hard_thresholding_fraction)
        self.sht = RealSHT(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type)
        self.isht = InverseRealSHT(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type)
        atmo, surf, dyn, stat, levels = get_channel_groups(channel_names, aux_channel_names)
        self.n_atmo_groups = len(levels)
        if len(atmo) % self.n_atmo_groups:
            raise ValueError("number of atmospheric variables not divisible by the number of pressure levels")
        self.n_atmo_chans = len(atmo) // self.n_atmo_groups
        self.register_buffer("atmo_channels", torch.tensor(atmo, dtype=torch.long), persistent=False)
        self.register_buffer("surf_channels", torch.tensor(surf, dtype=torch.long), persistent=False)
        self.register_buffer("aux_channels", torch.tensor(dyn + stat, dtype=torch.long), persistent=False)
        self.n_surf_chans, self.n_aux_chans = len(surf), len(dyn) + len(stat)
        self.n_out_chans = self.n_atmo_groups * self.n_atmo_chans + self.n_surf_chans
        self.total_embed_dim = self.n_atmo_groups * atmo_embed_dim + surf_embed_dim
        kernel_shape = tuple(kernel_shape)
        if activation_function not in ("relu", "gelu", "silu"):
            raise ValueError(f"Unknown activation function {activation_function}")
        act = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU}[activation_function]
        common = dict(kernel_shape=kernel_shape, basis_type=filter_basis_type, basis_norm_mode=filter_basis_norm_mode,
                      activation_function=act, bias=bias, use_mlp=encoder_mlp)
        enc = dict(inp_shape=self.inp_shape, out_shape=(self.h, self.w), grid_in=model_grid_type, grid_out=sht_grid_type, **common)
        dec = dict(inp_shape=(self.h, self.w), out_shape=self.out_shape, grid_in=sht_grid_type, grid_out=model_grid_type,
                   upsample_sht=upsample_sht, **common)
        # construction order = the reference's (its init RNG stream): encoders atmo, surf; decoders atmo, surf; aux encoder; blocks
        self.atmo_encoder = DiscreteContinuousEncoder(inp_chans=self.n_atmo_chans, out_chans=atmo_embed_dim,
                                                      groups=math.gcd(self.n_atmo_chans, atmo_embed_dim), **enc)
        if self.n_surf_chans > 0:
            self.surf_encoder = DiscreteContinuousEncoder(inp_chans=self.n_surf_chans, out_chans=surf_embed_dim,
                                                          groups=math.gcd(self.n_surf_chans, surf_embed_dim), **enc)
        self.atmo_decoder = DiscreteContinuousDecoder(inp_chans=atmo_embed_dim, out_chans=self.n_atmo_chans,
                                                      groups=math.gcd(self.n_atmo_chans, atmo_embed_dim), **dec)
        if self.n_surf_chans > 0:
            self.surf_decoder = DiscreteContinuousDecoder(inp_chans=surf_embed_dim, out_chans=self.n_surf_chans,
                                                          groups=math.gcd(self.n_surf_chans, surf_embed_dim), **dec)
        if self.n_aux_chans > 0:
            self.aux_encoder = DiscreteContinuousEncoder(inp_chans=self.n_aux_chans, out_chans=aux_embed_dim,
                                                         groups=math.gcd(self.n_aux_chans, aux_embed_dim), **enc)
        self.blocks = nn.ModuleList()
        for i in range(num_layers):
            self.blocks.append(NeuralOperatorBlock(
                self.sht, self.isht, self.total_embed_dim + (self.n_aux_chans > 0) * aux_embed_dim, self.total_embed_dim,
                conv_type="global" if i % sfno_block_frequency == 0 else "local", mlp_ratio=mlp_ratio, act_layer=act,
                normalization_layer=normalization_layer, skip="identity", layer_scale=layer_scale, use_mlp=use_mlp,
                kernel_shape=kernel_shape, basis_type=filter_basis_type, basis_norm_mode=filter_basis_norm_mode, bias=bias))
        if big_skip:
            self.residual_transform = nn.Conv2d(self.n_out_chans, self.n_out_chans, 1, bias=False)
            nn.init.normal_(self.residual_transform.weight, mean=0.0, std=math.sqrt(0.5 / self.n_out_chans))
        if clamp_water:
            water = get_water_channels(channel_names)
            if water:
                self.register_buffer("water_channels", torch.tensor(water, dtype=torch.long), persistent=False)
                mask = torch.zeros(self.n_out_chans, dtype=torch.bool)
                mask[water] = True
                self.register_buffer("water_channel_mask", mask.view(1, -1, 1, 1), persistent=False)

    def encode(self, x):
        """``fourcastnet3.py:1001-1024``: every pressure level goes through the SAME atmospheric encoder (levels ride in the batch)"""
        batchdims = x.shape[:-3]
        xa = x[..., self.atmo_channels, :, :].contiguous().reshape(-1, self.n_atmo_chans, *x.shape[-2:])
        out = self.atmo_encoder(xa)
        out = out.reshape(*batchdims, self.n_atmo_groups * self.atmo_embed_dim, *out.shape[-2:])
        if hasattr(self, "surf_encoder"):
            out = torch.cat((out, self.surf_encoder(x[..., self.surf_channels, :, :].contiguous())), dim=-3)
        return out.reshape(*batchdims, self.total_embed_dim, *out.shape[-2:])

    def encode_auxiliary_channels(self, x):
        if not hasattr(self, "aux_encoder"):
            return None
        return self.aux_encoder(x[..., self.aux_channels, :, :])

    def decode(self, x):
        """``fourcastnet3.py:1041-1064``"""
        batchdims = x.shape[:-3]
        xa = x[..., : (self.n_atmo_groups * self.atmo_embed_dim), :, :].reshape(-1, self.atmo_embed_dim, *x.shape[-2:])
        xa = self.atmo_decoder(xa)
        out = torch.zeros(*batchdims, self.n_out_chans, *xa.shape[-2:], dtype=x.dtype, device=x.device)
        out[..., self.atmo_channels, :, :] = xa.reshape(*batchdims, -1, *xa.shape[-2:])
        if hasattr(self, "surf_decoder"):
            xs = self.surf_decoder(x[..., -self.surf_embed_dim:, :, :])
            out[..., self.surf_channels, :, :] = xs.reshape(*batchdims, -1, *xs.shape[-2:])
        return out

    def process(self, x, x_aux=None):
        for blk in self.blocks:
            if x_aux is not None:
                x = torch.cat([x, x_aux], dim=-3)
            x = blk(x)
        return x

    def clamp_water_channels(self, x):
        """``fourcastnet3.py:1107-1124`` (without normalisation statistics attached to the module)"""
        if hasattr(self, "water_channels"):
            w = soft_clamp(x[..., self.water_channels, :, :])
            w_full = torch.zeros_like(x)
            w_full.index_copy_(-3, self.water_channels, w.to(x.dtype))
            x = torch.where(self.water_channel_mask, w_full, x)
        return x

    def forward(self, x):
        if self.big_skip:
            residual = x[..., : self.n_out_chans, :, :].contiguous()
        x_aux = self.encode_auxiliary_channels(x)
        x = self.decode(self.process(self.encode(x), x_aux))
        if self.big_skip:
            x = x + self.residual_transform(residual)
        return self.clamp_water_channels(x)
