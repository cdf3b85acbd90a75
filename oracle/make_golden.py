"""Generate the golden fixtures under ``tests/golden/`` by running the
reference's OWN modules (imported unmodified from /root/reference through
``oracle/ref_shims.py``) on PyTorch-CPU.

TEST INFRASTRUCTURE.  Run in the build container only:

    TORCHDYNAMO_DISABLE=1 python -m oracle.make_golden

What is the reference's and what is restated
--------------------------------------------
The model code (``SphericalFourierNeuralOperatorNet``, ``SpectralConv``,
``_contract_lwise``, ``MLP``, ``EncoderDecoder``, ``nn.InstanceNorm2d``) is the
reference's, byte for byte.  The SHT underneath it is ``oracle/sht.py`` because
torch-harmonics is not installable here; that restatement is
pinned separately against scipy (``tests/test_oracle_sht.py``).

Each fixture stores: constructor kwargs, the reference ``state_dict``, a seeded
input, the forward output, and gradients of ``sum(y * g)`` (seeded ``g``) w.r.t.
the input and every parameter.
"""

import json
import os
import sys

import numpy as np
import torch

from . import ref_shims

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _run_model(cls, kwargs, batch, seed, name):
    torch.manual_seed(seed)
    model = cls(**kwargs)
    model.train()
    x = torch.rand(batch, kwargs["inp_chans"], *kwargs["inp_shape"], requires_grad=True)
    g = torch.randn(batch, kwargs["out_chans"], *kwargs["out_shape"])
    y = model(x)
    (y * g).sum().backward()
    rec = {"kwargs": np.array(json.dumps(kwargs)), "x": _np(x), "g": _np(g), "y": _np(y), "gx": _np(x.grad)}
    for k, v in model.state_dict().items():
        rec["param/" + k] = _np(v)
    for k, p in model.named_parameters():
        rec["grad/" + k] = _np(p.grad)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path)/1e6:.2f} MB  |y|={y.abs().mean().item():.4f}")


def sfno_fixtures():
    SFNO = ref_shims.import_reference_sfno()
    # BASELINE config 1: the reference's test fixture (tests/testutils.py:33-36,
    # tests/test_trainers.py:90-92): 64x128, 5 channels, ctor defaults, 2 layers
    _run_model(
        SFNO,
        dict(inp_shape=(64, 128), out_shape=(64, 128), inp_chans=5, out_chans=5, num_layers=2),
        batch=1, seed=333, name="sfno_tiny_64x128.npz",
    )
    # the grid of tests/test_models.py:66-103 (36x72), B=2, odd-ish sizes, scale 3 like config 2
    _run_model(
        SFNO,
        dict(inp_shape=(37, 72), out_shape=(37, 72), inp_chans=3, out_chans=3, num_layers=3, scale_factor=3,
             embed_dim=16, mlp_ratio=2),
        batch=2, seed=334, name="sfno_small_37x72.npz",
    )
    # output grid != input grid (the big skip is resampled through trans_down -> itrans_up, sfnonet.py:886-891) and
    # quadrature-weighted instance norms (normalization_layer "instance_norm_s2", sfnonet.py:620-645)
    _run_model(
        SFNO,
        dict(inp_shape=(33, 64), out_shape=(25, 48), inp_chans=4, out_chans=3, num_layers=3, scale_factor=2,
             embed_dim=12, mlp_ratio=2, normalization_layer="instance_norm_s2"),
        batch=2, seed=335, name="sfno_s2norm_resample_33x64.npz",
    )
    # non-default constructor options in one go: deeper encoder / decoder, SiLU, no big skip, truncated spectrum,
    # SpectralConv bias, no normalisation  /  ReLU, no MLP, explicit max_modes, wider encoder
    _run_model(
        SFNO,
        dict(inp_shape=(24, 48), out_shape=(24, 48), inp_chans=3, out_chans=2, num_layers=2, scale_factor=2, embed_dim=8,
             mlp_ratio=2, activation_function="silu", encoder_layers=2, big_skip=False, hard_thresholding_fraction=0.75,
             bias=True, normalization_layer="none"),
        batch=2, seed=340, name="sfno_options_a_24x48.npz",
    )
    _run_model(
        SFNO,
        dict(inp_shape=(24, 48), out_shape=(24, 48), inp_chans=3, out_chans=3, num_layers=3, scale_factor=3, embed_dim=8,
             activation_function="relu", use_mlp=False, max_modes=(6, 7), encoder_ratio=2, decoder_ratio=2,
             model_grid_type="legendre-gauss", sht_grid_type="equiangular"),
        batch=1, seed=341, name="sfno_options_b_24x48.npz",
    )
    # learned position embeddings (sfnonet.py:732-764,898-911)
    for i, pe in enumerate(("direct", "frequency")):
        _run_model(
            SFNO,
            dict(inp_shape=(19, 36), out_shape=(19, 36), inp_chans=2, out_chans=2, num_layers=2, scale_factor=2,
                 embed_dim=8, mlp_ratio=2, pos_embed=pe),
            batch=1, seed=336 + i, name=f"sfno_posembed_{pe}_19x36.npz",
        )


def sfno_layernorm_fixture():
    """normalization_layer="layer_norm" (sfnonet.py:609-613: DistributedLayerNorm over the channels, mpu/layer_norm.py:256-290)"""
    SFNO = ref_shims.import_reference_sfno()
    _run_model(
        SFNO,
        dict(inp_shape=(24, 48), out_shape=(24, 48), inp_chans=3, out_chans=3, num_layers=2, scale_factor=2, embed_dim=12,
             mlp_ratio=2, normalization_layer="layer_norm"),
        batch=2, seed=342, name="sfno_layernorm_24x48.npz",
    )


def crps_fixtures():
    """CRPSLoss (makani/utils/losses/crps_loss.py:277-452, the reference's own module): every built crps_type, with and
    without spatial weights, a NaN observation, ties between members, E = 1: values and forecast gradients."""
    CRPSLoss = ref_shims.import_reference_module("makani.utils.losses.crps_loss").CRPSLoss
    cases = [
        dict(img=(19, 36), grid="equiangular", E=4, crps_type="skillspread", alpha=1.0, wgt=False, nan=False, ties=False),
        dict(img=(19, 36), grid="equiangular", E=5, crps_type="skillspread", alpha=0.95, wgt=True, nan=True, ties=True),
        dict(img=(12, 24), grid="legendre-gauss", E=8, crps_type="naive skillspread", alpha=1.0, wgt=False, nan=False, ties=False),
        dict(img=(12, 24), grid="legendre-gauss", E=3, crps_type="probability weighted moment", alpha=1.0, wgt=True, nan=True, ties=False),
        dict(img=(17, 32), grid="equiangular", E=16, crps_type="gauss", alpha=1.0, wgt=False, nan=False, ties=False),
        # E = 1 (the reference's own E = 1 branch fails with spatial weights: crps_loss.py:375-437 leaves
        # spatial_weights_split unbound)
        dict(img=(17, 32), grid="equiangular", E=1, crps_type="skillspread", alpha=1.0, wgt=False, nan=False, ties=False),
        # round 3: the "cdf" form (crps_loss.py:55-122; FourCastNet3's first pre-training stage), plain and with per-member
        # ensemble weights + ties + spatial weights, and ensemble sizes between the instantiated register capacities
        dict(img=(12, 24), grid="legendre-gauss", E=6, crps_type="cdf", alpha=1.0, wgt=False, nan=False, ties=False),
        dict(img=(19, 36), grid="equiangular", E=9, crps_type="cdf", alpha=1.0, wgt=True, nan=False, ties=True, ens_w=True),
        dict(img=(12, 24), grid="legendre-gauss", E=2, crps_type="cdf", alpha=1.0, wgt=False, nan=False, ties=False),
        dict(img=(12, 24), grid="legendre-gauss", E=14, crps_type="skillspread", alpha=1.0, wgt=False, nan=False, ties=False),
        dict(img=(12, 24), grid="legendre-gauss", E=27, crps_type="probability weighted moment", alpha=1.0, wgt=False, nan=False, ties=False),
    ]
    rec = {"cases": json.dumps(cases)}
    for i, c in enumerate(cases):
        torch.manual_seed(500 + i)
        B, C = 2, 3
        ens_w = torch.rand(c["E"]) + 0.5 if c.get("ens_w") else None
        mod = CRPSLoss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0), channel_names=[str(k) for k in range(C)],
                       grid_type=c["grid"], crps_type=c["crps_type"], alpha=c["alpha"], ensemble_weights=ens_w)
        if ens_w is not None:
            rec[f"{i}_ens_w"] = _np(ens_w)
        f = torch.randn(B, c["E"], C, *c["img"])
        if c["ties"]:
            f[:, 1] = f[:, 0]                      # two equal members everywhere: ordinal ranks break the tie by position
            f[:, 3, :, ::2] = f[:, 2, :, ::2]
        f.requires_grad_(True)
        o = torch.randn(B, C, *c["img"])
        if c["nan"]:
            o[0, 1, 3, 5] = float("nan")
            o[1, 2, :2] = float("nan")
        wgt = torch.rand(B, C, *c["img"]) + 0.5 if c["wgt"] else None
        out = mod(f, o, wgt)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        rec[f"{i}_f"], rec[f"{i}_o"], rec[f"{i}_g"], rec[f"{i}_out"], rec[f"{i}_df"] = _np(f), _np(o), _np(g), _np(out), _np(f.grad)
        if wgt is not None:
            rec[f"{i}_wgt"] = _np(wgt)
    path = os.path.join(OUT, "crps_loss.npz")
    np.savez_compressed(path, **rec)
    print(f"crps_loss.npz: {os.path.getsize(path)/1e6:.2f} MB")


def crps_spectral_fixtures():
    """SpectralCRPSLoss (makani/utils/losses/crps_loss.py:454-637, the reference's own module on the restated SHT): the built
    score types on the absolute values of the coefficients, with spectral weights, a truncated lmax, E = 1."""
    SpectralCRPSLoss = ref_shims.import_reference_module("makani.utils.losses.crps_loss").SpectralCRPSLoss
    cases = [
        dict(img=(17, 32), grid="equiangular", E=4, crps_type="skillspread", alpha=1.0, wgt=False, lmax=None),
        dict(img=(12, 24), grid="legendre-gauss", E=5, crps_type="skillspread", alpha=0.95, wgt=True, lmax=None),
        dict(img=(17, 32), grid="equiangular", E=3, crps_type="probability weighted moment", alpha=1.0, wgt=False, lmax=9),
        dict(img=(12, 24), grid="legendre-gauss", E=8, crps_type="gauss", alpha=1.0, wgt=True, lmax=None),
        dict(img=(12, 24), grid="legendre-gauss", E=2, crps_type="skillspread", alpha=1.0, wgt=False, lmax=None),
        dict(img=(17, 32), grid="equiangular", E=1, crps_type="skillspread", alpha=1.0, wgt=False, lmax=None),
        dict(img=(12, 24), grid="legendre-gauss", E=5, crps_type="cdf", alpha=1.0, wgt=True, lmax=None),      # round 3
        # absolute=False: the naive skill / spread kernel on the complex coefficients themselves (crps_loss.py:536-545,605-608)
        dict(img=(17, 32), grid="equiangular", E=4, crps_type="skillspread", alpha=0.95, wgt=True, lmax=None, absolute=False),
        dict(img=(12, 24), grid="legendre-gauss", E=7, crps_type="skillspread", alpha=1.0, wgt=False, lmax=8, absolute=False),
    ]
    rec = {"cases": json.dumps(cases)}
    for i, c in enumerate(cases):
        torch.manual_seed(700 + i)
        B, C = 2, 3
        mod = SpectralCRPSLoss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0), channel_names=[str(k) for k in range(C)],
                               grid_type=c["grid"], lmax=c["lmax"], crps_type=c["crps_type"], alpha=c["alpha"],
                               absolute=c.get("absolute", True))
        f = torch.randn(B, c["E"], C, *c["img"], requires_grad=True)
        o = torch.randn(B, C, *c["img"])
        L, M = mod.lm_weights.shape
        # (the reference reshapes the product with lm_weights to (1, 1, L * M), crps_loss.py:578: weights per (l, m) only)
        wgt = torch.rand(1, 1, L, M) + 0.5 if c["wgt"] else None
        out = mod(f, o, wgt)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        rec[f"{i}_f"], rec[f"{i}_o"], rec[f"{i}_g"], rec[f"{i}_out"], rec[f"{i}_df"] = _np(f), _np(o), _np(g), _np(out), _np(f.grad)
        if wgt is not None:
            rec[f"{i}_wgt"] = _np(wgt)
    path = os.path.join(OUT, "crps_spectral.npz")
    np.savez_compressed(path, **rec)
    print(f"crps_spectral.npz: {os.path.getsize(path)/1e6:.2f} MB")


def fcn3_fixtures(which="recipe"):
    """FourCastNet3 (makani/models/networks/fourcastnet3.py, the reference's own module) on top of the restated
    torch-harmonics operators: SHT (oracle/sht.py) and DISCO convolution / ResampleS2 (oracle/disco.py).  The DISCO
    restatement is parity-unpinned against the package itself (see its header), so these fixtures pin the NETWORK code
    (channel grouping, encoders / decoders, global + local blocks, layer scale, big skip, water clamp), not psi."""
    FCN3 = ref_shims.import_reference_module("makani.models.networks.fourcastnet3").AtmoSphericNeuralOperatorNet
    chans = ["u500", "v500", "t500", "q500", "u850", "v850", "t850", "q850", "u10m", "v10m", "t2m", "tcwv"]
    cases = [
        # global + local blocks, aux channels, instance norm, big skip, water clamp, bilinear decoder
        ("fcn3_small_33x64.npz", 2, 350,
         dict(inp_shape=(33, 64), out_shape=(33, 64), scale_factor=2, filter_basis_type="morlet", channel_names=chans,
              aux_channel_names=["xzen", "xoro"], atmo_embed_dim=4, surf_embed_dim=8, aux_embed_dim=4, num_layers=4,
              normalization_layer="instance_norm", big_skip=True, clamp_water=True, sfno_block_frequency=2)),
        # encoder MLPs, SHT upsampling in the decoder, channel layer norm, bias, no layer scale, no aux, 3 layers
        # (normalization_layer="instance_norm_s2" cannot be built by the reference itself: fourcastnet3.py:98-108 passes
        # pole_mask= to a GeometricInstanceNormS2 that has no such argument)
        ("fcn3_options_24x48.npz", 1, 351,
         dict(inp_shape=(24, 48), out_shape=(24, 48), scale_factor=2, filter_basis_type="morlet", channel_names=chans[:8] + ["t2m"],
              aux_channel_names=[], atmo_embed_dim=6, surf_embed_dim=4, num_layers=3, encoder_mlp=True, upsample_sht=True,
              normalization_layer="layer_norm", layer_scale=False, bias=True, activation_function="silu",
              model_grid_type="equiangular", sht_grid_type="legendre-gauss", hard_thresholding_fraction=0.75)),
    ]
    # the other filter bases of torch-harmonics 0.7.4 - 0.8.0 through the reference's own network (kernel sizes 5 and 6 instead
    # of Morlet's 9; the Zernike cutoff heuristic, fourcastnet3.py:46-50, reaches sqrt(8) times farther)
    bases = [
        ("fcn3_piecewise_linear_24x48.npz", 1, 352,
         dict(inp_shape=(24, 48), out_shape=(24, 48), scale_factor=2, filter_basis_type="piecewise linear", kernel_shape=(3, 4),
              channel_names=chans[:8] + ["t2m", "tcwv"], aux_channel_names=["xzen"], atmo_embed_dim=4, surf_embed_dim=4, aux_embed_dim=2,
              num_layers=2, normalization_layer="instance_norm", sfno_block_frequency=2)),
        ("fcn3_zernike_24x48.npz", 1, 353,
         dict(inp_shape=(24, 48), out_shape=(24, 48), scale_factor=2, filter_basis_type="zernike", kernel_shape=(3, 3),
              channel_names=chans[:8] + ["t2m", "tcwv"], aux_channel_names=["xzen"], atmo_embed_dim=4, surf_embed_dim=4, aux_embed_dim=2,
              num_layers=2, normalization_layer="instance_norm", sfno_block_frequency=2, bias=True)),
    ]
    if which == "bases":
        cases = bases
    for name, batch, seed, kwargs in cases:
        torch.manual_seed(seed)
        model = FCN3(**kwargs)
        model.train()
        nin = len(kwargs["channel_names"]) + len(kwargs["aux_channel_names"])
        x = torch.rand(batch, nin, *kwargs["inp_shape"], requires_grad=True)
        g = torch.randn(batch, len(kwargs["channel_names"]), *kwargs["out_shape"])
        y = model(x)
        (y * g).sum().backward()
        rec = {"kwargs": np.array(json.dumps(kwargs)), "x": _np(x), "g": _np(g), "y": _np(y), "gx": _np(x.grad)}
        for k, v in model.state_dict().items():
            rec["param/" + k] = _np(v)
        for k, p_ in model.named_parameters():
            rec["grad/" + k] = _np(p_.grad)
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **rec)
        print(f"{name}: {os.path.getsize(path)/1e6:.2f} MB  |y|={y.abs().mean().item():.4f}")


SUBSAMPLE = 7


def seeded_field(seed, shape, kind):
    """x / g of the large-grid fixtures are not stored: both sides regenerate them from a seeded CPU generator (same torch
    build here and on the GPU box); the fixture holds their sums as a check"""
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=gen) if kind == "rand" else torch.randn(shape, generator=gen)


def fcn3_block_fixture():
    """ONE processor block of FourCastNet3 at BASELINE config 4's internal grid (360 x 720 Gauss, local DISCO convolution with
    the doubled cutoff, instance norms, MLP, layer scale, identity skip on the first out_chans channels) with reduced channel
    counts: the reference's own ``NeuralOperatorBlock`` (makani/models/networks/fourcastnet3.py:421-638) over the restated
    DISCO operator, evaluated with the entry-by-entry contraction (oracle.disco.disco_contraction_direct).  The fixture stores
    the parameters, every parameter gradient, the norms of the output / input gradient and both fields on a stride-7 lattice
    (the full fields would be 4 x 12 MB); x and g are regenerated from their seeds."""
    from . import disco as _disco
    from . import sht as _sht
    mod = ref_shims.import_reference_module("makani.models.networks.fourcastnet3")
    nlat, nlon, cin, cout = 360, 720, 12, 8
    fwd = _sht.RealSHT(nlat, nlon, grid="legendre-gauss")
    inv = _sht.InverseRealSHT(nlat, nlon, grid="legendre-gauss")
    kwargs = dict(inp_chans=cin, out_chans=cout, conv_type="local", mlp_ratio=2.0, normalization_layer="instance_norm",
                  skip="identity", layer_scale=True, use_mlp=True, kernel_shape=[3, 3], basis_type="morlet",
                  basis_norm_mode="mean", bias=False)
    old = _disco.DiscreteContinuousConvS2.contraction
    _disco.DiscreteContinuousConvS2.contraction = "direct"
    try:
        torch.manual_seed(352)
        blk = mod.NeuralOperatorBlock(fwd, inv, act_layer=torch.nn.GELU, **kwargs)
        blk.train()
        with torch.no_grad():                          # non-trivial affine parameters, biases and layer scale
            for n, p_ in blk.named_parameters():
                if n.endswith("bias"):
                    p_.normal_(0.0, 0.1)
                elif n.startswith("norm") or n.startswith("layer_scale"):
                    p_.mul_(1.0 + 0.3 * torch.randn_like(p_))
        x = seeded_field(3520, (1, cin, nlat, nlon), "rand").requires_grad_(True)
        g = seeded_field(3521, (1, cout, nlat, nlon), "randn")
        y = blk(x)
        (y * g).sum().backward()
    finally:
        _disco.DiscreteContinuousConvS2.contraction = old
    s = SUBSAMPLE
    rec = {"kwargs": np.array(json.dumps(dict(kwargs, nlat=nlat, nlon=nlon, grid="legendre-gauss", x_seed=3520, g_seed=3521,
                                             subsample=s))),
           "x_sum": np.float64(x.detach().double().sum()), "g_sum": np.float64(g.double().sum()),
           "y_sub": _np(y[..., ::s, ::s]), "gx_sub": _np(x.grad[..., ::s, ::s]),
           "y_norm": np.float64(y.detach().double().norm()), "gx_norm": np.float64(x.grad.double().norm())}
    for k, v in blk.state_dict().items():
        rec["param/" + k] = _np(v)
    for k, p_ in blk.named_parameters():
        rec["grad/" + k] = _np(p_.grad)
    path = os.path.join(OUT, "fcn3_local_block_360x720.npz")
    np.savez_compressed(path, **rec)
    print(f"fcn3_local_block_360x720.npz: {os.path.getsize(path)/1e6:.2f} MB  |y|={y.abs().mean().item():.4f}")


def fcn3_real_grid_fixture():
    """The WHOLE FourCastNet3 network at BASELINE config 4's grids (721 x 1440 equiangular in / out, 360 x 720 Gauss inside:
    DISCO encoders 721x1440 -> 360x720, one global (SHT) and two local (DISCO, doubled cutoff) processor blocks, bilinear
    ResampleS2 + DISCO decoders at 721 x 1440) with reduced channel counts (2 pressure levels x 3 variables + 2 surface
    variables, 2 auxiliary channels, embedding 4 / 4 / 4): the reference's own ``AtmoSphericNeuralOperatorNet``
    (makani/models/networks/fourcastnet3.py) over the restated operators, DISCO contractions evaluated entry by entry.
    Stored like the block fixture: parameters, every parameter gradient, norms and stride-7 lattices of y and gx; x and g are
    regenerated from their seeds.  Takes ~10 minutes of CPU."""
    from . import disco as _disco
    FCN3 = ref_shims.import_reference_module("makani.models.networks.fourcastnet3").AtmoSphericNeuralOperatorNet
    chans = ["u500", "v500", "t500", "u850", "v850", "t850", "u10m", "t2m"]
    kwargs = dict(inp_shape=(721, 1440), out_shape=(721, 1440), scale_factor=2, model_grid_type="equiangular", sht_grid_type="legendre-gauss",
                  filter_basis_type="morlet", kernel_shape=[3, 3], channel_names=chans, aux_channel_names=["xzen", "xoro"],
                  atmo_embed_dim=4, surf_embed_dim=4, aux_embed_dim=4, num_layers=3, sfno_block_frequency=3,
                  normalization_layer="none", use_mlp=True, mlp_ratio=2, activation_function="gelu", big_skip=False, bias=False,
                  encoder_mlp=False)
    old = _disco.DiscreteContinuousConvS2.contraction
    _disco.DiscreteContinuousConvS2.contraction = "direct"
    try:
        torch.manual_seed(353)
        model = FCN3(**kwargs)
        model.train()
        nin = len(chans) + 2
        x = seeded_field(3530, (1, nin, 721, 1440), "rand").requires_grad_(True)
        g = seeded_field(3531, (1, len(chans), 721, 1440), "randn")
        y = model(x)
        (y * g).sum().backward()
    finally:
        _disco.DiscreteContinuousConvS2.contraction = old
    st = SUBSAMPLE
    rec = {"kwargs": np.array(json.dumps(dict(kwargs, x_seed=3530, g_seed=3531, subsample=st))),
           "x_sum": np.float64(x.detach().double().sum()), "g_sum": np.float64(g.double().sum()),
           "y_sub": _np(y[..., ::st, ::st]), "gx_sub": _np(x.grad[..., ::st, ::st]),
           "y_norm": np.float64(y.detach().double().norm()), "gx_norm": np.float64(x.grad.double().norm())}
    for k, v in model.state_dict().items():
        rec["param/" + k] = _np(v)
    for k, p_ in model.named_parameters():
        rec["grad/" + k] = _np(p_.grad)
    path = os.path.join(OUT, "fcn3_config4_grids_721x1440.npz")
    np.savez_compressed(path, **rec)
    print(f"fcn3_config4_grids_721x1440.npz: {os.path.getsize(path)/1e6:.2f} MB  |y|={y.abs().mean().item():.4f}")


def spectral_conv_fixtures():
    sc = ref_shims.import_reference_module("makani.models.common.spectral_convolution")
    th = sys.modules["torch_harmonics"]
    rec = {}
    # (nlat_in, nlon_in, grid_in) -> (nlat_out, nlon_out, grid_out), lmax, mmax, Cin, Cout, B, operator
    cases = [
        (33, 64, "equiangular", 33, 64, "equiangular", 16, 17, 4, 6, 2, "dhconv"),
        (33, 64, "equiangular", 12, 24, "legendre-gauss", 12, 13, 8, 8, 1, "dhconv"),   # down-sampling (block 0)
        (12, 24, "legendre-gauss", 33, 64, "equiangular", 12, 13, 8, 8, 2, "dhconv"),   # up-sampling (last block)
        (12, 24, "legendre-gauss", 12, 24, "legendre-gauss", 12, 12, 6, 4, 2, "diagonal"),  # lmax == mmax: the reference init only broadcasts then
        # (..., num_groups, separable): the remaining contractions of contractions.py:17-54
        (12, 24, "legendre-gauss", 12, 24, "legendre-gauss", 12, 12, 6, 10, 2, "diagonal", 2, False),
        (12, 24, "legendre-gauss", 12, 24, "legendre-gauss", 12, 12, 7, 7, 2, "diagonal", 1, True),
        (33, 64, "equiangular", 33, 64, "equiangular", 16, 17, 6, 6, 2, "dhconv", 2, True),
        (33, 64, "equiangular", 12, 24, "legendre-gauss", 12, 13, 8, 16, 2, "dhconv", 2, False),
        (12, 24, "legendre-gauss", 33, 64, "equiangular", 12, 13, 12, 12, 1, "dhconv", 3, False),
    ]
    for idx, case in enumerate(cases):
        h0, w0, g0, h1, w1, g1, lmax, mmax, cin, cout, B, op = case[:12]
        groups, separable = case[12:] if len(case) > 12 else (1, False)
        torch.manual_seed(100 + idx)
        fwd = th.RealSHT(h0, w0, lmax=lmax, mmax=mmax, grid=g0).float()
        inv = th.InverseRealSHT(h1, w1, lmax=lmax, mmax=mmax, grid=g1).float()
        layer = sc.SpectralConv(fwd, inv, cin, cout, num_groups=groups, operator_type=op, separable=separable, bias=False, gain=2.0)
        x = torch.randn(B, cin, h0, w0, requires_grad=True)
        y, res = layer(x)
        gy = torch.randn_like(y)
        gr = torch.randn_like(res)
        ((y * gy).sum() + (res * gr).sum()).backward()
        p = f"case{idx}/"
        rec[p + "meta"] = np.array(json.dumps(dict(h0=h0, w0=w0, g0=g0, h1=h1, w1=w1, g1=g1, lmax=lmax, mmax=mmax,
                                                  cin=cin, cout=cout, B=B, op=op, groups=groups, separable=separable)))
        rec[p + "x"], rec[p + "w"], rec[p + "y"], rec[p + "res"] = _np(x), _np(layer.weight), _np(y), _np(res)
        rec[p + "gy"], rec[p + "gr"], rec[p + "gx"], rec[p + "gw"] = _np(gy), _np(gr), _np(x.grad), _np(layer.weight.grad)
    rec["ncases"] = np.array(len(cases))
    path = os.path.join(OUT, "spectral_conv.npz")
    np.savez_compressed(path, **rec)
    print(f"spectral_conv.npz: {os.path.getsize(path)/1e6:.2f} MB")


def contraction_fixtures():
    con = ref_shims.import_reference_module("makani.models.common.contractions")
    torch.manual_seed(7)
    x = torch.randn(2, 1, 6, 9, 10, dtype=torch.complex64)
    w = torch.randn(1, 6, 5, 9, dtype=torch.complex64)
    wd = torch.randn(1, 6, 5, 9, 10, dtype=torch.complex64)
    y = con._contract_dense_pytorch(x, w, separable=False, operator_type="dhconv")
    yd = con._contract_dense_pytorch(x, wd, separable=False, operator_type="diagonal")
    path = os.path.join(OUT, "contractions.npz")
    np.savez_compressed(path, x=_np(x), w=_np(w), wd=_np(wd), y=_np(y), yd=_np(yd))
    print(f"contractions.npz: {os.path.getsize(path)/1e6:.2f} MB")


def loss_fixtures():
    """GeometricLpLoss (makani/utils/losses/lp_loss.py:28-107) over every quadrature rule, with crop, p in
    {1, 1.5, 2}, relative / squared, optional weights: values and input gradients of the reference module."""
    GeometricLpLoss = ref_shims.import_reference_module("makani.utils.losses.lp_loss").GeometricLpLoss

    cases = [
        dict(img=(37, 72), crop=(37, 72), off=(0, 0), grid="equiangular", p=2.0, relative=False, squared=True, wgt=False),
        dict(img=(37, 72), crop=(37, 72), off=(0, 0), grid="equiangular", p=2.0, relative=False, squared=False, wgt=True),
        dict(img=(24, 48), crop=(24, 48), off=(0, 0), grid="legendre-gauss", p=1.0, relative=True, squared=False, wgt=False),
        dict(img=(33, 64), crop=(30, 60), off=(2, 3), grid="clenshaw-curtiss", p=1.5, relative=True, squared=True, wgt=False),
        dict(img=(33, 64), crop=(33, 61), off=(0, 1), grid="weatherbench2", p=3.0, relative=False, squared=False, wgt=False),
        dict(img=(19, 36), crop=(19, 36), off=(0, 0), grid="euclidean", p=2.0, relative=True, squared=False, wgt=True),
    ]
    rec = {"cases": json.dumps(cases)}
    for i, c in enumerate(cases):
        torch.manual_seed(100 + i)
        B, C = 2, 3
        mod = GeometricLpLoss(img_shape=c["img"], crop_shape=c["crop"], crop_offset=c["off"],
                              channel_names=[str(k) for k in range(C)], p=c["p"], relative=c["relative"],
                              squared=c["squared"], grid_type=c["grid"])
        prd = torch.randn(B, C, *c["crop"], requires_grad=True)
        tar = torch.randn(B, C, *c["crop"], requires_grad=True)
        wgt = torch.rand(B, C, *c["crop"]) + 0.5 if c["wgt"] else None
        out = mod(prd, tar, wgt)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        rec[f"{i}_q"] = _np(mod.quadrature.quad_weight.float())
        rec[f"{i}_prd"], rec[f"{i}_tar"], rec[f"{i}_g"] = _np(prd), _np(tar), _np(g)
        if wgt is not None:
            rec[f"{i}_wgt"] = _np(wgt)
        rec[f"{i}_out"], rec[f"{i}_dprd"], rec[f"{i}_dtar"] = _np(out), _np(prd.grad), _np(tar.grad)
    path = os.path.join(OUT, "geometric_lp_loss.npz")
    np.savez_compressed(path, **rec)
    print(f"geometric_lp_loss.npz: {os.path.getsize(path)/1e6:.2f} MB")

    # SpectralLpLoss (lp_loss.py:110-259) on top of the restated SHT
    SpectralLpLoss = ref_shims.import_reference_module("makani.utils.losses.lp_loss").SpectralLpLoss
    scases = [
        dict(img=(37, 72), grid="equiangular", p=2.0, relative=False, squared=True, wgt=False),
        dict(img=(37, 72), grid="equiangular", p=2.0, relative=True, squared=False, wgt=False),
        dict(img=(24, 48), grid="legendre-gauss", p=1.0, relative=False, squared=False, wgt=False),
        dict(img=(33, 64), grid="equiangular", p=3.0, relative=True, squared=True, wgt=True),
    ]
    rec = {"cases": json.dumps(scases)}
    for i, c in enumerate(scases):
        torch.manual_seed(200 + i)
        B, C = 2, 3
        mod = SpectralLpLoss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0),
                             channel_names=[str(k) for k in range(C)], grid_type=c["grid"], p=c["p"],
                             relative=c["relative"], squared=c["squared"])
        prd = torch.randn(B, C, *c["img"], requires_grad=True)
        tar = torch.randn(B, C, *c["img"], requires_grad=True)
        wgt = torch.rand(1, C, mod.sht.lmax, mod.sht.mmax) + 0.5 if c["wgt"] else None
        out = mod(prd, tar, wgt)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        rec[f"{i}_lm"] = _np(mod.lm_weights)
        rec[f"{i}_prd"], rec[f"{i}_tar"], rec[f"{i}_g"] = _np(prd), _np(tar), _np(g)
        if wgt is not None:
            rec[f"{i}_wgt"] = _np(wgt)
        rec[f"{i}_out"], rec[f"{i}_dprd"], rec[f"{i}_dtar"] = _np(out), _np(prd.grad), _np(tar.grad)
    path = os.path.join(OUT, "spectral_lp_loss.npz")
    np.savez_compressed(path, **rec)
    print(f"spectral_lp_loss.npz: {os.path.getsize(path)/1e6:.2f} MB")

    # SpectralH1Loss (utils/losses/h1_loss.py:30-180)
    SpectralH1Loss = ref_shims.import_reference_module("makani.utils.losses.h1_loss").SpectralH1Loss
    hcases = [
        dict(img=(37, 72), grid="equiangular", relative=False, squared=True, wgt=False),
        dict(img=(24, 48), grid="legendre-gauss", relative=True, squared=False, wgt=False),
        dict(img=(33, 64), grid="equiangular", relative=False, squared=False, wgt=True),
    ]
    rec = {"cases": json.dumps(hcases)}
    for i, c in enumerate(hcases):
        torch.manual_seed(400 + i)
        B, C = 2, 3
        mod = SpectralH1Loss(img_shape=c["img"], crop_shape=c["img"], crop_offset=(0, 0),
                             channel_names=[str(k) for k in range(C)], grid_type=c["grid"], relative=c["relative"],
                             squared=c["squared"])
        prd = torch.randn(B, C, *c["img"], requires_grad=True)
        tar = torch.randn(B, C, *c["img"], requires_grad=True)
        wgt = torch.rand(1, C, mod.sht.lmax, mod.sht.mmax) + 0.5 if c["wgt"] else None
        out = mod(prd, tar, wgt)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        rec[f"{i}_prd"], rec[f"{i}_tar"], rec[f"{i}_g"] = _np(prd), _np(tar), _np(g)
        if wgt is not None:
            rec[f"{i}_wgt"] = _np(wgt)
        rec[f"{i}_out"], rec[f"{i}_dprd"], rec[f"{i}_dtar"] = _np(out), _np(prd.grad), _np(tar.grad)
    path = os.path.join(OUT, "spectral_h1_loss.npz")
    np.savez_compressed(path, **rec)
    print(f"spectral_h1_loss.npz: {os.path.getsize(path)/1e6:.2f} MB")

    # GeometricInstanceNormS2 (models/common/layer_norm.py:30-160)
    S2Norm = ref_shims.import_reference_module("makani.models.common.layer_norm").GeometricInstanceNormS2
    ncases = [
        dict(img=(37, 72), crop=(37, 72), off=(0, 0), grid="equiangular", affine=True),
        dict(img=(24, 48), crop=(24, 48), off=(0, 0), grid="legendre-gauss", affine=False),
        dict(img=(33, 64), crop=(30, 60), off=(2, 3), grid="clenshaw-curtiss", affine=True),
    ]
    rec = {"cases": json.dumps(ncases)}
    for i, c in enumerate(ncases):
        torch.manual_seed(300 + i)
        B, C = 2, 5
        mod = S2Norm(img_shape=c["img"], crop_shape=c["crop"], crop_offset=c["off"], grid_type=c["grid"], num_features=C,
                     eps=1e-5, affine=c["affine"])
        if c["affine"]:
            with torch.no_grad():
                mod.weight.copy_(torch.randn(C) + 1.0)
                mod.bias.copy_(torch.randn(C))
        x = (torch.randn(B, C, *c["crop"]) * 2 + 1).requires_grad_(True)
        y = mod(x)
        g = torch.randn_like(y)
        (y * g).sum().backward()
        rec[f"{i}_x"], rec[f"{i}_y"], rec[f"{i}_g"] = _np(x), _np(y), _np(g)
        if x.grad is not None:                 # push-forward mode detaches the input of step 0 as well
            rec[f"{i}_dx"] = _np(x.grad)
        rec[f"{i}_q"] = _np(mod.quadrature.quad_weight.float())
        if c["affine"]:
            rec[f"{i}_w"], rec[f"{i}_b"] = _np(mod.weight), _np(mod.bias)
            rec[f"{i}_dw"], rec[f"{i}_db"] = _np(mod.weight.grad), _np(mod.bias.grad)
    path = os.path.join(OUT, "geometric_instance_norm_s2.npz")
    np.savez_compressed(path, **rec)
    print(f"geometric_instance_norm_s2.npz: {os.path.getsize(path)/1e6:.2f} MB")


def stepper_fixtures():
    """MultiStepWrapper rollouts (makani/models/stepper.py:176-345) of the reference around a small convolution:
    outputs in train / eval mode and the gradients of input and weights, for history windows, push-forward mode and
    rollout checkpointing.  The preprocessor runs with every optional stage off (history_normalization_mode "none")."""
    stepper = ref_shims.import_reference_module("makani.models.stepper")
    ParamsBase = ref_shims.import_reference_module("makani.utils.YParams").ParamsBase
    cases = [
        dict(n_history=0, n_future=0, push_forward=False, ckpt=False),
        dict(n_history=0, n_future=3, push_forward=False, ckpt=False),
        dict(n_history=0, n_future=2, push_forward=True, ckpt=False),
        dict(n_history=1, n_future=2, push_forward=False, ckpt=True),
        dict(n_history=2, n_future=3, push_forward=False, ckpt=False),
        dict(n_history=2, n_future=1, push_forward=True, ckpt=True),
    ]
    rec = {"cases": json.dumps(cases)}
    B, C, H, W = 2, 3, 8, 16
    for i, c in enumerate(cases):
        p = ParamsBase()
        for k, v in dict(img_shape_x=H, img_shape_y=W, img_shape_x_resampled=H, img_shape_y_resampled=W,
                         n_history=c["n_history"], history_normalization_mode="none", n_future=c["n_future"],
                         channel_names=[f"c{j}" for j in range(C)], batch_size=B, multistep_checkpoint=c["ckpt"],
                         multistep={"push_forward": c["push_forward"]}).items():
            p[k] = v
        torch.manual_seed(500 + i)
        cin = (c["n_history"] + 1) * C
        net = torch.nn.Sequential(torch.nn.Conv2d(cin, 5, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(5, C, 1)).double()
        wrap = stepper.MultiStepWrapper(p, lambda: net)
        x = torch.randn(B, cin, H, W, dtype=torch.float64, requires_grad=True)
        wrap.train()
        y = wrap(x)
        g = torch.randn_like(y)
        (y * g).sum().backward()
        rec[f"{i}_x"], rec[f"{i}_y"], rec[f"{i}_g"] = _np(x), _np(y), _np(g)
        if x.grad is not None:                 # push-forward mode detaches the input of step 0 as well
            rec[f"{i}_dx"] = _np(x.grad)
        for j, q in enumerate(net.parameters()):
            rec[f"{i}_p{j}"], rec[f"{i}_dp{j}"] = _np(q), _np(q.grad)
        wrap.eval()
        with torch.no_grad():
            rec[f"{i}_y_eval"] = _np(wrap(x))
    path = os.path.join(OUT, "multistep_rollout.npz")
    np.savez_compressed(path, **rec)
    print(f"multistep_rollout.npz: {os.path.getsize(path)/1e6:.2f} MB")


def main():
    if not ref_shims.reference_available():
        raise SystemExit("reference tree not found; golden fixtures can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["contractions", "spectral_conv", "sfno", "loss", "stepper", "fcn3", "crps", "crps_spectral"]
    if "contractions" in which:
        contraction_fixtures()
    if "spectral_conv" in which:
        spectral_conv_fixtures()
    if "sfno" in which:
        sfno_fixtures()
    if "sfno" in which or "sfno_layernorm" in which:
        sfno_layernorm_fixture()
    if "loss" in which:
        loss_fixtures()
    if "stepper" in which:
        stepper_fixtures()
    if "fcn3" in which:
        fcn3_fixtures()
    if "fcn3_bases" in which:
        fcn3_fixtures("bases")
    if "fcn3" in which or "fcn3_block" in which:
        fcn3_block_fixture()
    if "fcn3_real" in which:                     # (minutes of CPU: not part of the default set)
        fcn3_real_grid_fixture()
    if "crps" in which:
        crps_fixtures()
    if "crps_spectral" in which:
        crps_spectral_fixtures()


if __name__ == "__main__":
    main()
