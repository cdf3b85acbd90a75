"""Pointwise blocks of the SFNO (boundary rows a8-a11 of SURVEY.md §8a).

State-dict layout and initialisation follow
``makani/models/common/layers.py:537-661`` (EncoderDecoder) and ``:664-894`` (MLP):
``<name>.fwd.<idx>.weight`` of shape ``(Cout, Cin, 1, 1)``.

Everything here runs on HIP kernels under bf16 autocast: the channel GEMMs (1x1 convolutions on the NCHW
planes, forward / data gradient / weight gradient) on the LDS-DMA ring kernels of ``csrc/conv1x1.hip`` with the
bias, GELU, gelu' and skip-connection work in their epilogues, instance norm (+GELU) in ``csrc/pointwise.hip``.
Without autocast (fp32 parity runs) the GEMMs run on the fp32 GEMM engine of the Legendre transforms (``ops.chan_gemm_f32``):
no library GEMM in the product — DESIGN.md §5.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops


_DEFER_BIAS = True      # fc2's bias rides in the following instance norm (NeuralOperatorBlock); False: the plain path (A/B aid)


def hip_conv_eligible(x) -> bool:
    """bf16 compute (autocast or bf16 tensors), pixel count % 8 == 0: forward / data-gradient channel GEMMs run on the
    bf16 kernels of csrc/conv1x1.hip (LDS-DMA ring kernels, fused bias / GELU / gelu' / skip epilogues); everything else
    (fp32 parity runs, ragged toy grids) on the package's fp32 GEMM engine (``ops.ConvMmFn``).  No library GEMM either way."""
    if not x.is_cuda or x.dim() != 4 or (x.shape[-1] * x.shape[-2]) % 8 != 0:
        return False
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda") == torch.bfloat16
    return x.dtype == torch.bfloat16


class PointwiseConv(nn.Module):
    """1x1 convolution on NCHW: ``y[b] = W @ x[b]`` with ``x[b]`` viewed as (Cin, H*W).
    Parameter names/shapes of ``nn.Conv2d(cin, cout, 1)``."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        ops.want_bf16_shadow(self.weight)

    def matmul(self, x, add_to=None, add_to_is_fresh=False):
        """bias-free product; ``add_to`` (same shape as the result) is fused as the GEMM's C input.  The product is
        accumulated INTO ``add_to`` only when the caller states that no autograd node saved that tensor
        (``add_to_is_fresh``: it is the output of an instance norm or of a GEMM of this package)."""
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {C}")
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x = x.to(dt)
            add_to = add_to.to(dt) if add_to is not None else None
        with torch.autocast(device_type="cuda", enabled=False):
            return ops.ConvMmFn.apply(x.contiguous(), self.weight, add_to, add_to_is_fresh)

    @torch.compiler.disable(recursive=True)
    def forward(self, x, add_to=None, add_to_is_fresh=False):
        if hip_conv_eligible(x):
            xb = x.to(torch.bfloat16)
            r = add_to.to(torch.bfloat16) if add_to is not None else None
            return ops.Conv1x1Fn.apply(xb, self.weight, self.bias, r)
        y = self.matmul(x, add_to, add_to_is_fresh)
        if self.bias is not None:
            y = y + self.bias.to(y.dtype).view(1, -1, 1, 1)
        return y


class _Act(nn.Module):
    """placeholder keeping the reference's Sequential indices; GELU is fused into the previous op."""

    def __init__(self, act_layer):
        super().__init__()
        self.is_gelu = act_layer is nn.GELU
        self.act = None if self.is_gelu else act_layer()

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        return ops.BiasGeluFn.apply(x, None) if self.is_gelu else self.act(x)


def _conv_gelu_conv(c1: PointwiseConv, c2: PointwiseConv, x):
    """c2(gelu(c1(x))) as one autograd node on the bf16 HIP GEMMs (fused bias/GELU/gelu' epilogues)."""
    return ops.ConvGeluConvFn.apply(x.to(torch.bfloat16), c1.weight, c1.bias, c2.weight, c2.bias)


def _conv_act(conv: PointwiseConv, act: _Act, x):
    """act(conv(x) + bias) with bias+GELU fused in one HIP pass."""
    if act.is_gelu:
        return ops.BiasGeluFn.apply(conv.matmul(x), conv.bias)
    return act.act(conv(x))


class MLP(nn.Module):
    """``makani/models/common/layers.py:664-894`` for ``input_format="nchw"``, no dropout."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, output_bias=True,
                 input_format="nchw", drop_rate=0.0, drop_type="iid", checkpointing=False, gain=1.0, use_te=False,
                 **kwargs):
        super().__init__()
        if input_format != "nchw":
            raise NotImplementedError("the HIP MLP implements input_format='nchw'")
        if drop_rate > 0.0:                            # layers.py:798-806
            if drop_type == "iid":
                drop = nn.Dropout(drop_rate)
            elif drop_type == "features":
                drop = nn.Dropout2d(drop_rate)
            else:
                raise NotImplementedError(f"Error, drop_type {drop_type} not supported")
        else:
            drop = nn.Identity()
        self.has_dropout = drop_rate > 0.0
        self.checkpointing = checkpointing
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        fc1 = PointwiseConv(in_features, hidden_features, bias=True)
        fc2 = PointwiseConv(hidden_features, out_features, bias=output_bias)
        for p in (fc1.weight, fc1.bias, fc2.weight, fc2.bias):
            if p is not None:
                p.is_shared_mp = ["spatial"]
        nn.init.normal_(fc1.weight, mean=0.0, std=math.sqrt(2.0 / in_features))
        nn.init.constant_(fc1.bias, 0.0)
        nn.init.normal_(fc2.weight, mean=0.0, std=math.sqrt(gain / hidden_features))
        if fc2.bias is not None:
            nn.init.constant_(fc2.bias, 0.0)
        self.fwd = nn.Sequential(fc1, _Act(act_layer), drop, fc2, drop)

    def _run(self, x):
        if self.has_dropout:        # dropout sits between the two GEMMs and after the second: no fused pair (stochastic
            h = self.fwd[2](_conv_act(self.fwd[0], self.fwd[1], x))          # masks come from torch's generator)
            return self.fwd[4](self.fwd[3](h))
        if self.fwd[1].is_gelu and hip_conv_eligible(x):
            return _conv_gelu_conv(self.fwd[0], self.fwd[3], x)
        h = _conv_act(self.fwd[0], self.fwd[1], x)
        return self.fwd[3](h)

    def can_defer_output_bias(self, x) -> bool:
        """the output bias can ride in the instance norm that follows (no add pass, no reduction for its gradient)"""
        return self.fwd[3].bias is not None and not self.checkpointing and _DEFER_BIAS and not self.has_dropout

    @torch.compiler.disable(recursive=True)
    def forward_deferred_bias(self, x):
        """(fc2(act(fc1(x))) WITHOUT the output bias, that bias): for a caller that folds it into its next op"""
        if self.fwd[1].is_gelu and hip_conv_eligible(x):
            c1, c2 = self.fwd[0], self.fwd[3]
            return ops.ConvGeluConvFn.apply(x.to(torch.bfloat16), c1.weight, c1.bias, c2.weight, None), c2.bias
        h = _conv_act(self.fwd[0], self.fwd[1], x)
        return self.fwd[3].matmul(h), self.fwd[3].bias

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        if self.checkpointing and torch.is_grad_enabled():
            return torch.utils.checkpoint.checkpoint(self._run, x, use_reentrant=False)
        return self._run(x)


class EncoderDecoder(nn.Module):
    """``makani/models/common/layers.py:537-661`` for ``input_format="nchw"``, groups=1."""

    def __init__(self, num_layers, input_dim, output_dim, hidden_dim, act_layer, gain=1.0, input_format="nchw", groups=1):
        super().__init__()
        if input_format != "nchw" or groups != 1:
            raise NotImplementedError("the HIP EncoderDecoder implements input_format='nchw', groups=1")
        mods, cur = [], input_dim
        for _ in range(num_layers):
            c = PointwiseConv(cur, hidden_dim, bias=True)
            c.weight.is_shared_mp = ["spatial"]
            c.bias.is_shared_mp = ["spatial"]
            nn.init.normal_(c.weight, mean=0.0, std=math.sqrt(2.0 / cur))
            nn.init.constant_(c.bias, 0.0)
            mods += [c, _Act(act_layer)]
            cur = hidden_dim
        c = PointwiseConv(cur, output_dim, bias=False)
        c.weight.is_shared_mp = ["spatial"]
        nn.init.normal_(c.weight, mean=0.0, std=math.sqrt(gain / cur))
        mods.append(c)
        self.fwd = nn.Sequential(*mods)

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        mods = list(self.fwd)
        if len(mods) == 3 and mods[1].is_gelu and hip_conv_eligible(x):
            return _conv_gelu_conv(mods[0], mods[2], x)
        i = 0
        while i < len(mods) - 1:
            x = _conv_act(mods[i], mods[i + 1], x)
            i += 2
        return mods[-1](x)


class InstanceNorm2d(nn.Module):
    """``nn.InstanceNorm2d(num_features, eps, affine, track_running_stats=False)``
    (``makani/models/networks/sfnonet.py:618-620``); ``forward(x, fuse_gelu=True)`` applies an exact
    GELU in the same pass (the block's ``act_layer0``, ``sfnonet.py:387-393``)."""

    def __init__(self, num_features, eps=1e-5, affine=False, track_running_stats=False):
        super().__init__()
        if track_running_stats:
            raise NotImplementedError("running statistics are not used by the SFNO and are not implemented")
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
    def forward(self, x, fuse_gelu=False, pre_bias=None):
        if x.dim() != 4 or x.shape[1] != self.num_features:
            raise ValueError(f"expected (B, {self.num_features}, H, W), got {tuple(x.shape)}")
        return ops.InstanceNormFn.apply(x, self.weight, self.bias, self.eps, fuse_gelu, pre_bias)


class GeometricInstanceNormS2(nn.Module):
    """Instance normalisation with quadrature weights on the sphere
    (``makani/models/common/layer_norm.py:30-160``): mean = sum_ij q_ij x_ij, var = sum_ij q_ij (x_ij - mean)^2 with the
    normalised quadrature weights of the grid (``GridQuadrature(..., normalize=True)``), then the usual
    normalise (+ affine).  Statistics in fp32, output in the input dtype; two HIP passes forward, two backward."""

    def __init__(self, img_shape, crop_shape, crop_offset, grid_type, num_features, eps=1e-05, affine=False):
        super().__init__()
        from .losses import GridQuadrature, grid_to_quadrature_rule
        self.eps, self.affine, self.num_features = eps, affine, num_features
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.quadrature = GridQuadrature(grid_to_quadrature_rule(grid_type), img_shape=img_shape, crop_shape=crop_shape,
                                         crop_offset=crop_offset, normalize=True, distributed=False)
        self._qsum = float(self.quadrature.quad_weight.double().sum())      # 1 on the full grid, < 1 on a crop

    @torch.compiler.disable(recursive=True)
    def forward(self, x, fuse_gelu=False):
        q = self.quadrature.quad_weight
        if x.dim() != 4 or tuple(x.shape[-2:]) != tuple(q.shape[-2:]):
            raise ValueError(f"expected (B, C, {q.shape[-2]}, {q.shape[-1]}), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return ops.InstanceNormFn.apply(x, self.weight, self.bias, self.eps, fuse_gelu, None, q.reshape(-1), self._qsum)


class DropPath(nn.Module):
    """Stochastic depth per sample (``makani/models/common/layers.py:30-75``): in training the whole residual branch of a
    random subset of the batch is zeroed and the survivors are scaled by 1 / (1 - p); identity in eval mode."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


class ChannelLayerNorm(nn.Module):
    """``DistributedLayerNorm`` of ``makani/mpu/layer_norm.py:256-290`` (``normalization_layer="layer_norm"``): a layer
    norm over the CHANNELS of an NCHW tensor, one statistic per grid point — the same on a lat/lon shard as on the full
    grid, so the spatially parallel network uses this class unchanged.  State-dict keys ``norm.weight`` / ``norm.bias``
    as the reference (the ``nn.LayerNorm`` submodule only holds the parameters); the arithmetic is the HIP kernel of
    ``csrc/chan_layernorm.hip`` on the NCHW planes themselves — no transposes, no library layer norm.  Output dtype: float32
    under autocast (torch runs ``layer_norm`` in fp32 there, which is what the reference module returns), else the input's."""

    def __init__(self, normalized_shape, eps=1e-05, elementwise_affine=True, bias=True):
        super().__init__()
        self.norm = nn.LayerNorm(normalized_shape, eps=eps, elementwise_affine=elementwise_affine, bias=bias)
        if len(self.norm.normalized_shape) != 1:
            raise ValueError("ChannelLayerNorm normalises over the channel dimension: normalized_shape must be one integer")
        if elementwise_affine:
            self.norm.weight.is_shared_mp = ["model"]
            self.norm.weight.sharded_dims_mp = [None]
            if bias:
                self.norm.bias.is_shared_mp = ["model"]
                self.norm.bias.sharded_dims_mp = [None]

    @torch.compiler.disable(recursive=True)
    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.norm.normalized_shape[0]:
            raise ValueError(f"expected (B, {self.norm.normalized_shape[0]}, H, W), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("makani_amd ops need GPU tensors (the HIP path has no CPU fallback)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        out_dtype = torch.float32 if torch.is_autocast_enabled("cuda") else x.dtype
        return ops.ChannelLayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype)
