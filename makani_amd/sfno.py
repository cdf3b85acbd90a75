"""Drop-in SFNO network (boundary B3 of SURVEY.md §8b).

``SphericalFourierNeuralOperatorNet`` takes the constructor arguments of
``makani/models/networks/sfnonet.py:518-553`` (unknown keyword arguments are
accepted and ignored, as the reference does at ``:552``), produces the same
``state_dict`` keys and shapes, and maps ``(B, inp_chans, H, W) -> (B, out_chans, H, W)``.
It can be named in a makani yaml as ``nettype: "makani_amd/sfno.py:SphericalFourierNeuralOperatorNet"``
(``makani/models/model_registry.py:189-192``).

Scope: the serial (one GPU per model instance) configuration family of
BASELINE.json — ``spectral_transform="sht"``, ``filter_type="linear"``,
every ``operator_type`` / ``separable`` combination of ``SpectralConv``, ``normalization_layer in {"instance_norm",
"instance_norm_s2", "layer_norm", "none"}``, ``pos_embed in {"none", "direct", "frequency"}``, dropout / stochastic depth
(torch's generator; the fused MLP pair is only taken with drop rate 0).  Anything else raises ``NotImplementedError``
(``ValueError`` where the reference raises one).
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ._lib import device_guard
from torch.utils.checkpoint import checkpoint

from . import distributed as thd
from .layers import MLP, ChannelLayerNorm, DropPath, EncoderDecoder, GeometricInstanceNormS2, InstanceNorm2d, PointwiseConv
from .sht import InverseRealSHT, RealSHT
from .spectral_conv import SpectralConv


class SpectralFilterLayer(nn.Module):
    """``sfnonet.py:52-166`` for ``filter_type="linear"``."""

    def __init__(self, forward_transform, inverse_transform, embed_dim, filter_type="linear", operator_type="diagonal",
                 hidden_size_factor=1, rank=1.0, separable=False, complex_activation="real", spectral_layers=1,
                 bias=False, drop_rate=0.0, gain=1.0):
        super().__init__()
        if filter_type != "linear":
            raise NotImplementedError("only filter_type='linear' (SpectralConv) is implemented")
        self.filter = SpectralConv(forward_transform, inverse_transform, embed_dim, embed_dim,
                                   operator_type=operator_type, separable=separable, bias=bias, gain=gain)

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        return self.filter(x)


class NeuralOperatorBlock(nn.Module):
    """``sfnonet.py:169-408``: filter -> norm0 -> act -> MLP -> norm1 -> + outer_skip(residual)."""

    def __init__(self, forward_transform, inverse_transform, embed_dim, filter_type="linear", operator_type="diagonal",
                 mlp_ratio=2.0, mlp_drop_rate=0.0, path_drop_rate=0.0, act_layer=nn.GELU,
                 norm_layer=(nn.Identity, nn.Identity), rank=1.0, separable=False, inner_skip="linear",
                 outer_skip=None, use_mlp=False, comm_feature_name="matmul", complex_activation="real",
                 spectral_layers=1, bias=False, final_activation=False, checkpointing_level=0):
        super().__init__()
        if hasattr(forward_transform, "lat_shapes"):
            self.input_shape_loc = (forward_transform.lat_shapes[forward_transform.comm_rank_polar],
                                    forward_transform.lon_shapes[forward_transform.comm_rank_azimuth])
            self.output_shape_loc = (inverse_transform.lat_shapes[inverse_transform.comm_rank_polar],
                                     inverse_transform.lon_shapes[inverse_transform.comm_rank_azimuth])
        else:
            self.input_shape_loc = (forward_transform.nlat, forward_transform.nlon)
            self.output_shape_loc = (inverse_transform.nlat, inverse_transform.nlon)

        # The SFNO builds every block with inner_skip="none" and outer_skip="linear" (sfnonet.py:680-700); those are the
        # two wirings of this boundary class.  Submodules are created in the reference's order (norm0, filter, norm1,
        # outer_skip, mlp) because that order IS the RNG stream of the initialisation.
        if inner_skip not in ("none", None):
            raise NotImplementedError(f"inner_skip={inner_skip!r}: the SFNO path uses inner_skip='none'")
        if outer_skip not in ("linear", "none", None):
            raise NotImplementedError(f"outer_skip={outer_skip!r}: the SFNO path uses outer_skip='linear' (or none)")
        has_act = act_layer is not nn.Identity
        self.norm0 = norm_layer[0]()
        self.filter = SpectralFilterLayer(forward_transform, inverse_transform, embed_dim, filter_type, operator_type,
                                          hidden_size_factor=mlp_ratio, rank=rank, separable=separable,
                                          complex_activation=complex_activation, spectral_layers=spectral_layers,
                                          bias=bias, drop_rate=path_drop_rate, gain=2.0 if has_act else 1.0)
        self.act_is_gelu = act_layer is nn.GELU
        self.act_layer0 = act_layer()
        self.norm1 = norm_layer[1]()
        gain = 2.0 if (final_activation and has_act) else 1.0
        if outer_skip == "linear":
            gain /= 2.0                                  # the block's two branches share the output variance
            self.outer_skip = PointwiseConv(embed_dim, embed_dim, bias=False)
            nn.init.normal_(self.outer_skip.weight, std=math.sqrt(gain / embed_dim))
        if use_mlp:
            self.mlp = MLP(in_features=embed_dim, hidden_features=int(embed_dim * mlp_ratio), act_layer=act_layer,
                           drop_rate=mlp_drop_rate, drop_type="features", checkpointing=(checkpointing_level >= 2), gain=gain)
        self.drop_path = DropPath(path_drop_rate) if path_drop_rate > 0.0 else nn.Identity()      # sfnonet.py:361-362
        if final_activation:
            self.act_layer1 = act_layer()

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        x, residual = self.filter(x)
        norms = (InstanceNorm2d, GeometricInstanceNormS2, thd.DistributedInstanceNorm2d)        # (the distributed S2 norm derives from the last)
        if self.act_is_gelu and isinstance(self.norm0, norms):
            x = self.norm0(x, fuse_gelu=True)                 # norm + exact GELU in one pass over the plane
        else:
            x = self.act_layer0(self.norm0(x))

        if hasattr(self, "mlp") and type(self.norm1) is InstanceNorm2d and self.mlp.can_defer_output_bias(x):
            x, pb = self.mlp.forward_deferred_bias(x)         # fc2's bias rides in norm1 (same rounding as y + b)
            x = self.norm1(x, pre_bias=pb)
        else:
            if hasattr(self, "mlp"):
                x = self.mlp(x)
            x = self.norm1(x)
        x = self.drop_path(x)

        if hasattr(self, "outer_skip"):
            # the skip GEMM accumulates into x (beta = 1) when x is a tensor nobody saved for backward: the output of
            # a norm kernel or of the MLP's last GEMM (+ bias); an activation output (ReLU saves it) is not
            fresh = (hasattr(self, "mlp") or isinstance(self.norm1, norms)) and isinstance(self.drop_path, nn.Identity)
            x = self.outer_skip(residual, add_to=x, add_to_is_fresh=fresh)
        if hasattr(self, "act_layer1"):
            x = self.act_layer1(x)
        return x


class SphericalFourierNeuralOperatorNet(nn.Module):
    """``makani/models/networks/sfnonet.py:411-934`` on the MI355X HIP path."""

    def __init__(self, spectral_transform="sht", model_grid_type="equiangular", sht_grid_type="legendre-gauss",
                 filter_type="linear", operator_type="dhconv", inp_shape=(721, 1440), out_shape=(721, 1440),
                 scale_factor=8, inp_chans=2, out_chans=2, embed_dim=32, num_layers=4, use_mlp=True, mlp_ratio=2.0,
                 encoder_ratio=1, decoder_ratio=1, activation_function="gelu", encoder_layers=1, pos_embed="none",
                 pos_drop_rate=0.0, path_drop_rate=0.0, mlp_drop_rate=0.0, normalization_layer="instance_norm",
                 max_modes=None, hard_thresholding_fraction=1.0, big_skip=True, rank=1.0, separable=False,
                 complex_activation="real", spectral_layers=3, bias=False, checkpointing_level=0, **kwargs):
        super().__init__()
        if spectral_transform != "sht":
            raise NotImplementedError("only spectral_transform='sht' is implemented")
        if pos_embed not in ("none", "None", None, "direct", "frequency"):
            raise ValueError("Unknown position embedding type")

        self.inp_shape, self.out_shape = tuple(inp_shape), tuple(out_shape)
        self.inp_chans, self.out_chans, self.embed_dim = inp_chans, out_chans, embed_dim
        self.big_skip = big_skip
        self.checkpointing_level = checkpointing_level
        self.h = int(self.inp_shape[0] // scale_factor)
        self.w = int(self.inp_shape[1] // scale_factor)
        self._init_spectral_transforms(model_grid_type, sht_grid_type, hard_thresholding_fraction, max_modes)

        try:
            act = {"relu": nn.ReLU, "gelu": nn.GELU, "silu": nn.SiLU}[activation_function]
        except KeyError:
            raise ValueError(f"Unknown activation function {activation_function}")

        self.encoder = EncoderDecoder(num_layers=encoder_layers, input_dim=inp_chans, output_dim=embed_dim,
                                      hidden_dim=int(encoder_ratio * embed_dim), act_layer=act, input_format="nchw")
        self.pos_drop = nn.Dropout(p=pos_drop_rate) if pos_drop_rate > 0.0 else nn.Identity()      # sfnonet.py:605-606
        dpr = [v.item() for v in torch.linspace(0, path_drop_rate, num_layers)]

        if normalization_layer == "layer_norm":              # sfnonet.py:609-613
            norm = partial(ChannelLayerNorm, normalized_shape=embed_dim, elementwise_affine=True, eps=1e-6)
        elif normalization_layer == "instance_norm":
            if self.spatial_parallel:     # sfnonet.py:614-617
                norm = partial(thd.DistributedInstanceNorm2d, num_features=embed_dim, eps=1e-6, affine=True)
            else:
                norm = partial(InstanceNorm2d, num_features=embed_dim, eps=1e-6, affine=True, track_running_stats=False)
        elif normalization_layer == "instance_norm_s2":        # sfnonet.py:620-645
            handle = thd.DistributedGeometricInstanceNormS2 if self.spatial_parallel else GeometricInstanceNormS2
            norm = partial(handle, img_shape=(self.h, self.w), crop_shape=(self.h, self.w), crop_offset=(0, 0),
                           grid_type=model_grid_type, num_features=embed_dim, eps=1e-6, affine=True)
            norm_out = partial(handle, img_shape=self.out_shape, crop_shape=self.out_shape, crop_offset=(0, 0),
                               grid_type=model_grid_type, num_features=embed_dim, eps=1e-6, affine=True)
        elif normalization_layer == "none":
            norm = nn.Identity
        else:
            raise NotImplementedError(f"Error, normalization {normalization_layer} not implemented.")
        if normalization_layer != "instance_norm_s2":
            norm_out = norm

        self.blocks = nn.ModuleList([])
        for i in range(num_layers):
            first, last = i == 0, i == num_layers - 1
            self.blocks.append(NeuralOperatorBlock(
                self.trans_down if first else self.trans,
                self.itrans_up if last else self.itrans,
                embed_dim, filter_type=filter_type, operator_type=operator_type, mlp_ratio=mlp_ratio,
                mlp_drop_rate=mlp_drop_rate, path_drop_rate=dpr[i], act_layer=act,
                norm_layer=(norm, norm) if (first or not last) else (norm_out, norm_out),       # sfnonet.py:668-673
                inner_skip="none", outer_skip="linear", use_mlp=use_mlp, rank=rank, separable=separable,
                complex_activation=complex_activation, spectral_layers=spectral_layers, bias=bias,
                checkpointing_level=checkpointing_level))

        self.decoder = EncoderDecoder(num_layers=encoder_layers, input_dim=embed_dim, output_dim=out_chans,
                                      hidden_dim=int(decoder_ratio * embed_dim), act_layer=act,
                                      gain=0.5 if big_skip else 1.0, input_format="nchw")
        if big_skip:
            self.residual_transform = PointwiseConv(inp_chans, out_chans, bias=False)
            self.residual_transform.weight.is_shared_mp = ["spatial"]
            self.residual_transform.weight.sharded_dims_mp = [None, None, None, None]
            nn.init.normal_(self.residual_transform.weight, mean=0.0, std=math.sqrt(0.5 / inp_chans))

        # learned position embedding (sfnonet.py:732-764)
        if pos_embed == "direct":
            self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, self.inp_shape_loc[0], self.inp_shape_loc[1]))
            self.pos_embed.is_shared_mp = []
            self.pos_embed.sharded_dims_mp = [None, None, "h", "w"]
            self.pos_embed.type = "direct"
            with torch.no_grad():
                nn.init.trunc_normal_(self.pos_embed, std=0.02)
        elif pos_embed == "frequency":
            if self.spatial_parallel:
                lmax_loc = self.itrans_up.l_shapes[self.itrans_up.comm_rank_polar]
                mmax_loc = self.itrans_up.m_shapes[self.itrans_up.comm_rank_azimuth]
            else:
                lmax_loc, mmax_loc = self.itrans_up.lmax, self.itrans_up.mmax
            rcoeffs = nn.Parameter(torch.tril(torch.randn(1, embed_dim, lmax_loc, mmax_loc), diagonal=0))
            ccoeffs = nn.Parameter(torch.tril(torch.randn(1, embed_dim, lmax_loc, mmax_loc - 1), diagonal=-1))
            with torch.no_grad():
                nn.init.trunc_normal_(rcoeffs, std=0.02)
                nn.init.trunc_normal_(ccoeffs, std=0.02)
            self.pos_embed = nn.ParameterList([rcoeffs, ccoeffs])
            self.pos_embed.type = "frequency"
            self.pos_embed.is_shared_mp = []
            self.pos_embed.sharded_dims_mp = [None, None, "h", "w"]

    def _add_pos_embed(self, x):
        """sfnonet.py:898-911: the embedding is either the parameter itself or synthesised from learned coefficients by
        the HIP inverse SHT (fp32), then added in the activation dtype"""
        if self.pos_embed.type == "frequency":
            pe = torch.stack([self.pos_embed[0], nn.functional.pad(self.pos_embed[1], (1, 0), "constant", 0)], dim=-1)
            pe = self.itrans_up(torch.view_as_complex(pe))
        else:
            pe = self.pos_embed
        return x + pe.to(dtype=x.dtype)

    def _init_spectral_transforms(self, model_grid_type, sht_grid_type, hard_thresholding_fraction, max_modes):
        if max_modes is not None:
            modes_lat, modes_lon = max_modes
        else:
            modes_lat = int(self.h * hard_thresholding_fraction)
            modes_lon = int((self.w // 2 + 1) * hard_thresholding_fraction)
        # spatial (h x w) model parallelism, decided as the reference does (sfnonet.py:786-805): the process-group tree
        # says so (makani_amd.comm, filled by comm.init(h, w) or adopted from makani.utils.comm when makani's driver
        # constructs this plug-in) and the transform layer is initialised from its "h" / "w" groups on first use.
        # An explicit makani_amd.distributed.init(...) by the caller is honoured as well.
        self.spatial_parallel = thd.ensure_initialized()
        sht, isht = (thd.DistributedRealSHT, thd.DistributedInverseRealSHT) if self.spatial_parallel else (RealSHT, InverseRealSHT)
        self.trans_down = sht(*self.inp_shape, lmax=modes_lat, mmax=modes_lon, grid=model_grid_type).float()
        self.itrans_up = isht(*self.out_shape, lmax=modes_lat, mmax=modes_lon, grid=model_grid_type).float()
        self.trans = sht(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type).float()
        self.itrans = isht(self.h, self.w, lmax=modes_lat, mmax=modes_lon, grid=sht_grid_type).float()
        if self.spatial_parallel:
            ih, iw = self.trans_down.comm_rank_polar, self.trans_down.comm_rank_azimuth
            self.inp_shape_loc = (self.trans_down.lat_shapes[ih], self.trans_down.lon_shapes[iw])
            self.out_shape_loc = (self.itrans_up.lat_shapes[ih], self.itrans_up.lon_shapes[iw])
            self.h_loc, self.w_loc = self.itrans.lat_shapes[ih], self.itrans.lon_shapes[iw]
        else:
            self.inp_shape_loc = (self.trans_down.nlat, self.trans_down.nlon)
            self.out_shape_loc = (self.itrans_up.nlat, self.itrans_up.nlon)
            self.h_loc, self.w_loc = self.itrans.nlat, self.itrans.nlon

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def _forward_features(self, x):
        for blk in self.blocks:
            if self.checkpointing_level >= 3 and torch.is_grad_enabled():
                x = checkpoint(blk, x, use_reentrant=False)
            else:
                x = blk(x)
        return x

    @torch.compiler.disable(recursive=True)
    @device_guard
    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda") and not x.requires_grad
                and self.big_skip and self.out_shape == self.inp_shape):
            # under autocast the fp32 input is consumed twice (encoder, big-skip convolution) and each consumer would cast it:
            # cast once (the same values; one 300 MB pass less per step at 721 x 1440)
            x = x.to(torch.get_autocast_dtype("cuda"))
        if self.big_skip:
            if self.out_shape != self.inp_shape:
                B, C = x.shape[:2]
                residual = self.itrans_up.synthesis(self.trans_down.analysis(x), B, C, out_dtype=x.dtype)
            else:
                residual = x
        if self.checkpointing_level >= 1 and torch.is_grad_enabled():
            x = checkpoint(self.encoder, x, use_reentrant=False)
        else:
            x = self.encoder(x)
        if hasattr(self, "pos_embed"):
            x = self._add_pos_embed(x)
        x = self.pos_drop(x)
        x = self._forward_features(x)
        if self.checkpointing_level >= 1 and torch.is_grad_enabled():
            x = checkpoint(self.decoder, x, use_reentrant=False)
        else:
            x = self.decoder(x)
        if self.big_skip:
            x = self.residual_transform(residual, add_to=x, add_to_is_fresh=True)      # x: output of the decoder's last GEMM
        return x
