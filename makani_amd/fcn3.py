"""FourCastNet3 (``AtmoSphericNeuralOperatorNet``, ``makani/models/networks/fourcastnet3.py``) on the MI355X HIP path:
BASELINE config 4 (``fcn3_sc2_edim45_layers10``).

Boundary B3 of SURVEY.md §8b for this network: same constructor keywords (unknown ones are accepted and ignored,
``fourcastnet3.py:653-692``), same ``state_dict`` keys and shapes, ``forward((B, C + aux, H, W)) -> (B, C, H, W)``,
``encode`` / ``encode_auxiliary_channels`` / ``process`` / ``encode_process`` / ``decode`` / ``clamp_water_channels``.
What runs where:
  * encoders / decoders / "local" blocks: DISCO convolutions and bilinear resampling of ``makani_amd.disco``
    (``csrc/disco.hip`` + the channel GEMM kernels);
  * "global" blocks: ``makani_amd.SpectralConv`` (HIP FFT + split-bf16 Legendre / dhconv GEMMs);
  * norms, MLPs, skip convolutions: the pointwise HIP kernels of the SFNO path.
Spatial (h x w) model parallelism as in the reference: when ``comm.get_size("spatial") > 1`` the network builds the
distributed DISCO convolution / resampling (``makani_amd.disco``), the distributed transforms and the distributed instance
norms (``makani_amd.distributed``); every rank then works on its latitude / longitude shard.
"""
import math
import re
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from . import comm
from ._lib import device_guard
from . import distributed as thd
from .disco import (DiscreteContinuousConvS2, DistributedDiscreteContinuousConvS2, DistributedResampleS2, ResampleS2)
from .layers import MLP, ChannelLayerNorm, DropPath, EncoderDecoder, GeometricInstanceNormS2, InstanceNorm2d, PointwiseConv
from .sht import InverseRealSHT, RealSHT
from .spectral_conv import SpectralConv


# --------------------------------------------------------------------------- #
# helpers (fourcastnet3.py:46-59, makani/utils/features.py:70-140, makani/models/common/layers.py:154-197)
# --------------------------------------------------------------------------- #
def _compute_cutoff_radius(nlat, kernel_shape, basis_type):
    factor = {"piecewise linear": 0.5, "morlet": 0.5, "harmonic": 0.5, "zernike": math.sqrt(2.0)}
    return (kernel_shape[0] + 1) * factor[basis_type] * math.pi / float(nlat - 1)


def _soft_clamp(x, offset=0.0):
    x = x + offset
    y = torch.where(x > 0.0, x ** 2, 0.0)
    return torch.where(x >= 0.5, x - 0.25, y)


def get_water_channels(channel_names):
    return [c for c, ch in enumerate(channel_names) if ch[0] in {"q", "r"} or ch == "tcwv"]


def get_channel_groups(channel_names, aux_channel_names=()):
    """indices of the pressure-level variables (grouped by level), the surface variables and the auxiliary channels"""
    groups, surf = OrderedDict(), []
    for idx, chn in enumerate(channel_names):
        if re.search("[a-z]{1,3}[0-9]{1,4}$", chn) is not None and chn != "d2":
            groups.setdefault(int(re.search("[0-9]{1,4}$", chn).group()), []).append(idx)
        else:
            surf.append(idx)
    atmo, n = [], None
    for idx in groups.values():
        if n is not None and n != len(idx):
            raise ValueError(f"expected all atmospheric pressure level groups to have the same number of channels ({n}), "
                             f"but got {len(idx)}")
        n = len(idx)
        atmo += idx
    stat = [i + len(channel_names) for i, c in enumerate(aux_channel_names) if c in ("xoro", "xlsml", "xlsms")]
    dyn = [i + len(channel_names) for i, c in enumerate(aux_channel_names) if c not in ("xoro", "xlsml", "xlsms")]
    return atmo, surf, dyn, stat, list(groups.keys())


class LayerScale(nn.Module):
    """per-channel learned scale of a residual branch; the reference's depthwise 1x1 convolution is this multiply"""

    def __init__(self, num_chans=3, init_value=0.1):
        super().__init__()
        self.num_chans = num_chans
        self.weight = nn.Parameter(torch.full((num_chans, 1, 1, 1), float(init_value)))

    def forward(self, x):
        return x * self.weight.view(1, -1, 1, 1).to(x.dtype)


def _spatial_parallel():
    """as the reference decides it (``comm.get_size("spatial") > 1``; the transform layer is initialised from the tree's
    h / w groups on first use, ``fourcastnet3.py:340-343,922-927``)"""
    return thd.ensure_initialized()


def _conv_handle():
    return DistributedDiscreteContinuousConvS2 if _spatial_parallel() else DiscreteContinuousConvS2


def _norm_handle(h, w, embed_dim, normalization_layer="none", sht_grid_type="legendre-gauss"):
    if normalization_layer == "layer_norm":
        return partial(ChannelLayerNorm, normalized_shape=embed_dim, elementwise_affine=True, eps=1e-6)
    if normalization_layer == "instance_norm":
        if _spatial_parallel():
            return partial(thd.DistributedInstanceNorm2d, num_features=embed_dim, eps=1e-6, affine=True)
        return partial(InstanceNorm2d, num_features=embed_dim, eps=1e-6, affine=True, track_running_stats=False)
    if normalization_layer == "instance_norm_s2":
        handle = thd.DistributedGeometricInstanceNormS2 if _spatial_parallel() else GeometricInstanceNormS2
        return partial(handle, img_shape=(h, w), crop_shape=(h, w), crop_offset=(0, 0),
                       grid_type=sht_grid_type, num_features=embed_dim, eps=1e-6, affine=True)
    if normalization_layer == "none":
        return nn.Identity
    raise NotImplementedError(f"Error, normalization {normalization_layer} not implemented.")


def _annotate_spatial(conv):
    conv.weight.is_shared_mp = ["spatial"]
    conv.weight.sharded_dims_mp = [None, None, None]
    if conv.bias is not None:
        conv.bias.is_shared_mp = ["spatial"]
        conv.bias.sharded_dims_mp = [None]


# --------------------------------------------------------------------------- #
# encoder / decoder / block (fourcastnet3.py:117-253, 255-418, 421-638)
# --------------------------------------------------------------------------- #
class DiscreteContinuousEncoder(nn.Module):
    def __init__(self, inp_shape=(721, 1440), out_shape=(480, 960), grid_in="equiangular", grid_out="equiangular", inp_chans=2,
                 out_chans=2, kernel_shape=(3, 3), basis_type="harmonic", basis_norm_mode="mean", use_mlp=False, mlp_ratio=2.0,
                 activation_function=nn.GELU, groups=1, bias=False):
        super().__init__()
        cutoff = _compute_cutoff_radius(nlat=inp_shape[0], kernel_shape=kernel_shape, basis_type=basis_type)
        self.conv = _conv_handle()(inp_chans, out_chans, in_shape=inp_shape, out_shape=out_shape,
                                   kernel_shape=kernel_shape, basis_type=basis_type, basis_norm_mode=basis_norm_mode,
                                   grid_in=grid_in, grid_out=grid_out, groups=groups, bias=bias, theta_cutoff=cutoff)
        _annotate_spatial(self.conv)
        if use_mlp:
            with torch.no_grad():
                self.conv.weight *= math.sqrt(2.0)
            self.act = activation_function()
            self.mlp = EncoderDecoder(num_layers=1, input_dim=out_chans, output_dim=out_chans,
                                      hidden_dim=int(mlp_ratio * out_chans), act_layer=activation_function, input_format="nchw")

    def forward(self, x):
        x = self.conv(x)
        if hasattr(self, "act"):
            x = self.act(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        return x


class DiscreteContinuousDecoder(nn.Module):
    def __init__(self, inp_shape=(480, 960), out_shape=(721, 1440), grid_in="equiangular", grid_out="equiangular", inp_chans=2,
                 out_chans=2, kernel_shape=(3, 3), basis_type="harmonic", basis_norm_mode="mean", use_mlp=False, mlp_ratio=2.0,
                 activation_function=nn.GELU, groups=1, bias=False, upsample_sht=False):
        super().__init__()
        if use_mlp:
            self.mlp = EncoderDecoder(num_layers=1, input_dim=inp_chans, output_dim=inp_chans,
                                      hidden_dim=int(mlp_ratio * inp_chans), act_layer=activation_function, input_format="nchw",
                                      gain=2.0)
            self.act = activation_function()
        par = _spatial_parallel()
        if upsample_sht:
            sht, isht = (thd.DistributedRealSHT, thd.DistributedInverseRealSHT) if par else (RealSHT, InverseRealSHT)
            self.sht = sht(*inp_shape, grid=grid_in).float()
            self.isht = isht(*out_shape, lmax=self.sht.lmax, mmax=self.sht.mmax, grid=grid_out).float()
            self.upsample = nn.Sequential(self.sht, self.isht)
        else:
            self.upsample = (DistributedResampleS2 if par else ResampleS2)(*inp_shape, *out_shape, grid_in=grid_in,
                                                                            grid_out=grid_out, mode="bilinear")
        cutoff = _compute_cutoff_radius(nlat=out_shape[0], kernel_shape=kernel_shape, basis_type=basis_type)
        self.conv = _conv_handle()(inp_chans, out_chans, in_shape=out_shape, out_shape=out_shape,
                                   kernel_shape=kernel_shape, basis_type=basis_type, basis_norm_mode=basis_norm_mode,
                                   grid_in=grid_out, grid_out=grid_out, groups=groups, bias=False, theta_cutoff=cutoff)
        _annotate_spatial(self.conv)

    def forward(self, x):
        dtype = x.dtype
        if hasattr(self, "act"):
            x = self.act(x)
        if hasattr(self, "mlp"):
            x = self.mlp(x)
        with torch.autocast(device_type=x.device.type, enabled=False):      # the reference decodes in fp32
            x = x.to(torch.float32)
            x = self.upsample(x)
            x = self.conv(x)
        return x.to(dtype=dtype)


class NeuralOperatorBlock(nn.Module):
    """norm1 -> global (spectral) or local (DISCO) convolution -> norm2 -> MLP -> skip(x[:out_chans]) + layer_scale(dx)"""

    def __init__(self, forward_transform, inverse_transform, inp_chans, out_chans, conv_type="local", mlp_ratio=2.0,
                 mlp_drop_rate=0.0, path_drop_rate=0.0, act_layer=nn.GELU, normalization_layer="none", num_groups=1,
                 skip="identity", layer_scale=True, use_mlp=False, kernel_shape=(3, 3), basis_type="harmonic",
                 basis_norm_mode="mean", checkpointing_level=0, bias=False):
        super().__init__()
        self.inp_shape = (forward_transform.nlat, forward_transform.nlon)
        self.out_shape = (inverse_transform.nlat, inverse_transform.nlon)
        self.out_chans = out_chans
        gain_factor = 1.0
        if conv_type == "local":
            cutoff = 2 * _compute_cutoff_radius(nlat=self.inp_shape[0], kernel_shape=kernel_shape, basis_type=basis_type)
            self.local_conv = _conv_handle()(inp_chans, inp_chans, in_shape=self.inp_shape, out_shape=self.out_shape,
                                             kernel_shape=kernel_shape, basis_type=basis_type, basis_norm_mode=basis_norm_mode,
                                             groups=num_groups, grid_in=forward_transform.grid,
                                             grid_out=inverse_transform.grid, bias=False, theta_cutoff=cutoff)
            _annotate_spatial(self.local_conv)
            with torch.no_grad():
                self.local_conv.weight *= gain_factor
        elif conv_type == "global":
            self.global_conv = SpectralConv(forward_transform, inverse_transform, inp_chans, inp_chans, operator_type="dhconv",
                                            num_groups=num_groups, bias=bias, gain=gain_factor)
        else:
            raise ValueError(f"Unknown convolution type {conv_type}")
        handle = _norm_handle(self.inp_shape[0], self.inp_shape[1], inp_chans, normalization_layer=normalization_layer,
                              sht_grid_type=forward_transform.grid)
        self.norm1 = handle()
        self.norm2 = handle()
        if use_mlp:
            self.mlp = MLP(in_features=inp_chans, out_features=out_chans, hidden_features=int(inp_chans * mlp_ratio),
                           act_layer=act_layer, drop_rate=mlp_drop_rate, drop_type="features",
                           checkpointing=(checkpointing_level >= 2), gain=gain_factor)
        self.drop_path = DropPath(path_drop_rate) if path_drop_rate > 0.0 else nn.Identity()
        if layer_scale:
            self.layer_scale = LayerScale(out_chans)
            self.layer_scale.weight.is_shared_mp = ["spatial"]
            self.layer_scale.weight.sharded_dims_mp = [None, None, None, None]
        else:
            self.layer_scale = nn.Identity()
        if skip == "linear":
            self.skip = PointwiseConv(inp_chans, out_chans, bias=False)
            nn.init.normal_(self.skip.weight, std=math.sqrt(1.0 / inp_chans))
            self.skip.weight.is_shared_mp = ["spatial"]
            self.skip.weight.sharded_dims_mp = [None, None, None, None]
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
        dx = self.drop_path(dx)
        if hasattr(self, "skip"):
            res = self.skip(x[..., : self.out_chans, :, :])
            if isinstance(self.layer_scale, LayerScale):                          # skip + w[c] * dx in one pass
                return torch.addcmul(res, dx, self.layer_scale.weight.view(1, -1, 1, 1).to(dx.dtype))
            return res + self.layer_scale(dx)
        return dx


# --------------------------------------------------------------------------- #
# the network (fourcastnet3.py:641-1138)
# --------------------------------------------------------------------------- #
class AtmoSphericNeuralOperatorNet(nn.Module):
    def __init__(self, model_grid_type="equiangular", sht_grid_type="legendre-gauss", inp_shape=(721, 1440), out_shape=(721, 1440),
                 kernel_shape=(3, 3), filter_basis_type="harmonic", filter_basis_norm_mode="mean", scale_factor=8,
                 encoder_mlp=False, upsample_sht=False, channel_names=("u500", "v500"), aux_channel_names=(), n_history=0,
                 atmo_embed_dim=8, surf_embed_dim=8, aux_embed_dim=8, num_layers=4, num_groups=1, use_mlp=True, mlp_ratio=2.0,
                 activation_function="gelu", layer_scale=True, pos_drop_rate=0.0, path_drop_rate=0.0, mlp_drop_rate=0.0,
                 normalization_layer="none", max_modes=None, hard_thresholding_fraction=1.0, sfno_block_frequency=2,
                 big_skip=False, clamp_water=False, bias=False, checkpointing_level=0, freeze_encoder=False,
                 freeze_processor=False, **kwargs):
        super().__init__()
        self.inp_shape, self.out_shape = tuple(inp_shape), tuple(out_shape)
        self.atmo_embed_dim, self.surf_embed_dim, self.aux_embed_dim = atmo_embed_dim, surf_embed_dim, aux_embed_dim
        self.big_skip = big_skip
        self.checkpointing_level = checkpointing_level
        if n_history != 0:
            raise ValueError(f"this model currently does not support history, expected n_history == 0 but got {n_history}")
        self.h = int(self.inp_shape[0] // scale_factor)
        self.w = int(self.inp_shape[1] // scale_factor)
        self._init_spectral_transforms(model_grid_type, sht_grid_type, hard_thresholding_fraction, max_modes)
        self._precompute_channel_groups(list(channel_names), list(aux_channel_names))
        self.n_out_chans = self.n_atmo_groups * self.n_atmo_chans + self.n_surf_chans
        self.total_embed_dim = self.n_atmo_groups * self.atmo_embed_dim + self.surf_embed_dim
        kernel_shape = tuple(kernel_shape)
        try:
            act = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU}[activation_function]
        except KeyError:
            raise ValueError(f"Unknown activation function {activation_function}")

        enc = partial(DiscreteContinuousEncoder, inp_shape=self.inp_shape, out_shape=(self.h, self.w), grid_in=model_grid_type,
                      grid_out=sht_grid_type, kernel_shape=kernel_shape, basis_type=filter_basis_type,
                      basis_norm_mode=filter_basis_norm_mode, activation_function=act, bias=bias, use_mlp=encoder_mlp)
        dec = partial(DiscreteContinuousDecoder, inp_shape=(self.h, self.w), out_shape=self.out_shape, grid_in=sht_grid_type,
                      grid_out=model_grid_type, kernel_shape=kernel_shape, basis_type=filter_basis_type,
                      basis_norm_mode=filter_basis_norm_mode, activation_function=act, bias=bias, use_mlp=encoder_mlp,
                      upsample_sht=upsample_sht)
        # construction order = the reference's (it is the RNG stream of the initialisation)
        self.atmo_encoder = enc(inp_chans=self.n_atmo_chans, out_chans=atmo_embed_dim,
                                groups=math.gcd(self.n_atmo_chans, atmo_embed_dim))
        if self.n_surf_chans > 0:
            self.surf_encoder = enc(inp_chans=self.n_surf_chans, out_chans=surf_embed_dim,
                                    groups=math.gcd(self.n_surf_chans, surf_embed_dim))
        self.atmo_decoder = dec(inp_chans=atmo_embed_dim, out_chans=self.n_atmo_chans,
                                groups=math.gcd(self.n_atmo_chans, atmo_embed_dim))
        if self.n_surf_chans > 0:
            self.surf_decoder = dec(inp_chans=surf_embed_dim, out_chans=self.n_surf_chans,
                                    groups=math.gcd(self.n_surf_chans, surf_embed_dim))
        if self.n_aux_chans > 0:
            self.aux_encoder = enc(inp_chans=self.n_aux_chans, out_chans=aux_embed_dim,
                                   groups=math.gcd(self.n_aux_chans, aux_embed_dim))
        self.pos_drop = nn.Dropout(p=pos_drop_rate) if pos_drop_rate > 0.0 else nn.Identity()
        dpr = [v.item() for v in torch.linspace(0, path_drop_rate, num_layers)]
        self.blocks = nn.ModuleList([])
        for i in range(num_layers):
            self.blocks.append(NeuralOperatorBlock(
                self.sht, self.isht, self.total_embed_dim + (self.n_aux_chans > 0) * aux_embed_dim, self.total_embed_dim,
                conv_type="global" if i % sfno_block_frequency == 0 else "local", mlp_ratio=mlp_ratio,
                mlp_drop_rate=mlp_drop_rate, path_drop_rate=dpr[i], act_layer=act, normalization_layer=normalization_layer,
                skip="identity", layer_scale=layer_scale, use_mlp=use_mlp, kernel_shape=kernel_shape,
                basis_type=filter_basis_type, basis_norm_mode=filter_basis_norm_mode, bias=bias,
                checkpointing_level=checkpointing_level))
        if self.big_skip:
            self.residual_transform = PointwiseConv(self.n_out_chans, self.n_out_chans, bias=False)
            self.residual_transform.weight.is_shared_mp = ["spatial"]
            self.residual_transform.weight.sharded_dims_mp = [None, None, None, None]
            nn.init.normal_(self.residual_transform.weight, mean=0.0, std=math.sqrt(0.5 / self.n_out_chans))
        if clamp_water:
            water = get_water_channels(list(channel_names))
            if len(water) > 0:
                self.register_buffer("water_channels", torch.tensor(water, dtype=torch.long), persistent=False)
                mask = torch.zeros(self.n_out_chans, dtype=torch.bool)
                mask[water] = True
                self.register_buffer("water_channel_mask", mask.view(1, -1, 1, 1), persistent=False)
        if freeze_encoder:
            frozen = list(self.atmo_encoder.parameters()) + list(self.atmo_decoder.parameters())
            if hasattr(self, "surf_encoder"):
                frozen += list(self.surf_encoder.parameters()) + list(self.surf_decoder.parameters())
            if hasattr(self, "aux_encoder"):
                frozen += list(self.aux_encoder.parameters())
            if self.big_skip:
                frozen += list(self.residual_transform.parameters())
            for p in frozen:
                p.requires_grad = False
        if freeze_processor:
            for p in self.blocks.parameters():
                p.requires_grad = False

    def _init_spectral_transforms(self, model_grid_type="equiangular", sht_grid_type="legendre-gauss",
                                  hard_thresholding_fraction=1.0, max_modes=None):
        if max_modes is not None:
            modes_lat, modes_lon = max_modes
        else:
            modes_lat = int(self.h * hard_thresholding_fraction)
            modes_lon = int((self.w // 2 + 1) * hard_thresholding_fraction)
        sht, isht = (thd.DistributedRealSHT, thd.DistributedInverseRealSHT) if _spatial_parallel() else (RealSHT, InverseRealSHT)
        self.sht = sht(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type).float()
        self.isht = isht(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type).float()

    def _precompute_channel_groups(self, channel_names, aux_channel_names):
        atmo, surf, dyn, stat, levels = get_channel_groups(channel_names, aux_channel_names)
        aux = dyn + stat
        self.n_atmo_groups = len(levels)
        self.n_atmo_chans = len(atmo) // self.n_atmo_groups
        if len(atmo) % self.n_atmo_groups:
            raise ValueError("Expected number of atmospheric variables to be divisible by number of atmospheric groups but got "
                             f"{len(atmo)} and {self.n_atmo_groups}")
        self.register_buffer("atmo_channels", torch.LongTensor(atmo), persistent=False)
        self.register_buffer("surf_channels", torch.LongTensor(surf), persistent=False)
        self.register_buffer("aux_channels", torch.LongTensor(aux), persistent=False)
        self.n_surf_chans = self.surf_channels.shape[0]
        self.n_aux_chans = self.aux_channels.shape[0]

    def encode(self, x):
        batchdims = x.shape[:-3]
        x_atmo = x[..., self.atmo_channels, :, :].contiguous().reshape(-1, self.n_atmo_chans, *x.shape[-2:])
        x_out = self.atmo_encoder(x_atmo)
        x_out = x_out.reshape(*batchdims, self.n_atmo_groups * self.atmo_embed_dim, *x_out.shape[-2:])
        if hasattr(self, "surf_encoder"):
            x_surf = self.surf_encoder(x[..., self.surf_channels, :, :].contiguous())
            x_out = torch.cat((x_out, x_surf), dim=-3)
        return x_out.reshape(*batchdims, self.total_embed_dim, *x_out.shape[-2:])

    def encode_auxiliary_channels(self, x):
        if not hasattr(self, "aux_encoder"):
            return None
        batchdims = x.shape[:-3]
        x_aux = self.aux_encoder(x[..., self.aux_channels, :, :].contiguous())
        return x_aux.reshape(*batchdims, self.aux_embed_dim, *x_aux.shape[-2:])

    def decode(self, x):
        batchdims = x.shape[:-3]
        x_atmo = x[..., : (self.n_atmo_groups * self.atmo_embed_dim), :, :].reshape(-1, self.atmo_embed_dim, *x.shape[-2:])
        x_atmo = self.atmo_decoder(x_atmo)
        x_out = torch.zeros(*batchdims, self.n_out_chans, *x_atmo.shape[-2:], dtype=x.dtype, device=x.device)
        x_out[..., self.atmo_channels, :, :] = x_atmo.reshape(*batchdims, -1, *x_atmo.shape[-2:])
        if hasattr(self, "surf_decoder"):
            x_surf = self.surf_decoder(x[..., -self.surf_embed_dim:, :, :].contiguous())
            x_out[..., self.surf_channels, :, :] = x_surf.reshape(*batchdims, -1, *x_surf.shape[-2:])
        return x_out

    def process(self, x, x_aux=None):
        x = self.pos_drop(x)
        for blk in self.blocks:
            if x_aux is not None:
                x = torch.cat([x, x_aux], dim=-3)
            x = checkpoint(blk, x, use_reentrant=False) if self.checkpointing_level >= 3 else blk(x)
        return x

    def processor_blocks(self, x, x_aux=None):
        return self.process(x, x_aux)

    def encode_process(self, x):
        x_aux = self.encode_auxiliary_channels(x)
        x = checkpoint(self.encode, x, use_reentrant=False) if self.checkpointing_level >= 1 else self.encode(x)
        return self.process(x, x_aux)

    def clamp_water_channels(self, x):
        if hasattr(self, "water_channels"):
            if hasattr(self, "normalization_means") and hasattr(self, "normalization_stds"):
                means = self.normalization_means[self.water_channels].view(1, -1, 1, 1)
                stds = self.normalization_stds[self.water_channels].view(1, -1, 1, 1)
                offset = (means / stds).to(x.dtype)
                w = _soft_clamp(x[..., self.water_channels, :, :], offset=offset) - offset
            else:
                w = _soft_clamp(x[..., self.water_channels, :, :])
            w_full = torch.zeros_like(x)
            w_full.index_copy_(-3, self.water_channels, w.to(x.dtype))
            x = torch.where(self.water_channel_mask, w_full, x)
        return x

    @device_guard
    def forward(self, x):
        if self.big_skip:
            residual = x[..., : self.n_out_chans, :, :].contiguous()
        x = self.encode_process(x)
        x = checkpoint(self.decode, x, use_reentrant=False) if self.checkpointing_level >= 1 else self.decode(x)
        if self.big_skip:
            x = x + self.residual_transform(residual).to(x.dtype)
        return self.clamp_water_channels(x)
