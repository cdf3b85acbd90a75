"""FourCastNet3 plug-in (makani_amd/fcn3.py) against fixtures generated from the reference's own
``makani/models/networks/fourcastnet3.py`` (``python -m oracle.make_golden fcn3``): constructor / state-dict contract on
the CPU, forward + every gradient on the GPU (fp32 end-to-end tolerance 1e-4)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

FCN3_GOLDEN = ["fcn3_small_33x64.npz", "fcn3_options_24x48.npz", "fcn3_piecewise_linear_24x48.npz", "fcn3_zernike_24x48.npz"]


def _load(name):
    import makani_amd as ma
    g = load_golden(name)
    kwargs = json.loads(str(g["kwargs"]))
    model = ma.AtmoSphericNeuralOperatorNet(**kwargs, some_unknown_trainer_key=1)       # unknown kwargs are ignored
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    model.load_state_dict(sd, strict=True)
    return g, kwargs, model


@pytest.mark.parametrize("name", FCN3_GOLDEN)
def test_fcn3_state_dict_contract(name):
    g, kwargs, model = _load(name)
    ref_shapes = {k[len("param/"):]: g[k].shape for k in g.files if k.startswith("param/")}
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert own == {k: tuple(v) for k, v in ref_shapes.items()}
    assert set(dict(model.named_parameters())) == {k[len("grad/"):] for k in g.files if k.startswith("grad/")}
    for k, p in model.named_parameters():
        if ("coder.conv.weight" in k) or "local_conv" in k or "layer_scale" in k or k.startswith("residual_transform"):
            assert getattr(p, "is_shared_mp", None) == ["spatial"], k
    assert model.n_out_chans == len(kwargs["channel_names"])


def test_fcn3_channel_grouping_and_errors():
    from makani_amd import fcn3
    atmo, surf, dyn, stat, levels = fcn3.get_channel_groups(["u500", "v500", "t2m", "u850", "v850", "d2", "tcwv"], ["xzen", "xoro"])
    assert atmo == [0, 1, 3, 4] and surf == [2, 5, 6] and dyn == [7] and stat == [8] and levels == [500, 850]
    assert fcn3.get_water_channels(["q500", "t500", "r850", "tcwv", "u10m"]) == [0, 2, 3]
    with pytest.raises(ValueError):
        fcn3.get_channel_groups(["u500", "v500", "u850"])
    with pytest.raises(ValueError):
        fcn3.AtmoSphericNeuralOperatorNet(inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2, filter_basis_type="morlet", n_history=1)
    with pytest.raises(ValueError):
        fcn3.AtmoSphericNeuralOperatorNet(inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2, filter_basis_type="morlet",
                                          activation_function="tanh")
    x = torch.linspace(-1, 2, 13)
    y = fcn3._soft_clamp(x)
    assert (y[x <= 0] == 0).all() and torch.allclose(y[x >= 0.5], x[x >= 0.5] - 0.25)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FCN3_GOLDEN)
def test_fcn3_matches_reference_golden_fp32(name):
    g, kwargs, model = _load(name)
    model = model.to("cuda:0")
    x = torch.from_numpy(g["x"]).to("cuda:0").requires_grad_(True)
    y = model(x)
    (y * torch.from_numpy(g["g"]).to("cuda:0")).sum().backward()
    assert rel_l2(y, torch.from_numpy(g["y"])) < 1e-4
    assert rel_l2(x.grad, torch.from_numpy(g["gx"])) < 1e-4
    gmax = max(float(np.abs(g[k2]).max()) for k2 in g.files if k2.startswith("grad/"))
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        e = rel_l2(p.grad, ref)
        a = (p.grad.detach().cpu() - ref).abs().max().item()
        assert e < 2e-4 or a < 1e-4 * gmax, (k, e, a, gmax)        # zero-in-exact-arithmetic gradients: absolute scale


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fcn3_small_33x64.npz", "fcn3_piecewise_linear_24x48.npz", "fcn3_zernike_24x48.npz"])
def test_fcn3_bf16_autocast_runs_close_to_fp32(name):
    """bf16 autocast, the benchmark's precision — also with 5 and 6 basis functions per DISCO convolution (the channel mixes then
    run on group / channel counts the nine-function recipe never produces), forward and backward"""
    g, kwargs, model = _load(name)
    model = model.to("cuda:0")
    x = torch.from_numpy(g["x"]).to("cuda:0").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x)
    e_y = rel_l2(y.float(), torch.from_numpy(g["y"]))
    (y.float() * torch.from_numpy(g["g"]).to("cuda:0")).sum().backward()
    e_gx = rel_l2(x.grad, torch.from_numpy(g["gx"]))
    print(f"FourCastNet3 {name} bf16 autocast vs the fp32 fixture: output {e_y:.2e}, input gradient {e_gx:.2e}")
    assert e_y < 4e-2 and e_gx < 8e-2
    assert all(p.grad is not None and torch.isfinite(torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad).all()
               for p in model.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("amp,tol", [(False, 1e-4), (True, 2e-2)])
def test_fcn3_local_block_360x720_matches_reference_golden(amp, tol):
    """one FourCastNet3 processor block at BASELINE config 4's internal grid (360 x 720 Gauss, local DISCO convolution with the
    doubled cutoff, instance norms, MLP, layer scale; 12 -> 8 channels) against the reference's own NeuralOperatorBlock
    (fourcastnet3.py:421-638; fixture: oracle/make_golden.py fcn3_block_fixture).  The fixture holds the fields on a stride-7
    lattice plus their norms; weight gradients integrate over every pixel."""
    import types
    from makani_amd import fcn3
    from oracle.make_golden import seeded_field
    g = load_golden("fcn3_local_block_360x720.npz")
    kw = json.loads(str(g["kwargs"]))
    nlat, nlon, s = kw.pop("nlat"), kw.pop("nlon"), kw.pop("subsample")
    tr = types.SimpleNamespace(nlat=nlat, nlon=nlon, grid=kw.pop("grid"))
    xs, gs = kw.pop("x_seed"), kw.pop("g_seed")
    blk = fcn3.NeuralOperatorBlock(tr, tr, act_layer=torch.nn.GELU, **kw)
    blk.load_state_dict({k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}, strict=True)
    blk = blk.to("cuda:0")
    x = seeded_field(xs, (1, kw["inp_chans"], nlat, nlon), "rand")
    gy = seeded_field(gs, (1, kw["out_chans"], nlat, nlon), "randn")
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6 * x.numel() and abs(float(gy.double().sum()) - float(g["g_sum"])) < 1e-3
    xd = x.to("cuda:0").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = blk(xd)
    (y.float() * gy.to("cuda:0")).sum().backward()
    e_y = rel_l2(y.float()[..., ::s, ::s], torch.from_numpy(g["y_sub"]))
    e_gx = rel_l2(xd.grad[..., ::s, ::s], torch.from_numpy(g["gx_sub"]))
    n_y = abs(float(y.detach().double().norm()) - float(g["y_norm"])) / float(g["y_norm"])
    n_gx = abs(float(xd.grad.double().norm()) - float(g["gx_norm"])) / float(g["gx_norm"])
    print(f"FCN3 local block 360x720 amp={amp}: y {e_y:.2e} gx {e_gx:.2e} |y| {n_y:.2e} |gx| {n_gx:.2e}")
    assert e_y < tol and e_gx < tol and n_y < tol and n_gx < tol
    gmax = max(float(np.abs(g[k2]).max()) for k2 in g.files if k2.startswith("grad/"))
    for k, p in blk.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        e = rel_l2(p.grad, ref)
        a = (p.grad.detach().cpu().float() - ref).abs().max().item()
        assert e < 2 * tol or a < (1e-4 if not amp else 2e-2) * gmax, (k, e, a, gmax)


@pytest.mark.gpu
@pytest.mark.parametrize("amp,tol", [(False, 1e-4), (True, 4e-2)])
def test_fcn3_whole_network_at_config4_grids_matches_reference_golden(amp, tol):
    """the WHOLE FourCastNet3 network at BASELINE config 4's grids (721 x 1440 in / out, 360 x 720 inside: DISCO encoders, one
    global and two local processor blocks, ResampleS2 + DISCO decoders) with reduced channel counts against the reference's own
    ``AtmoSphericNeuralOperatorNet`` (fixture: oracle/make_golden.py fcn3_real_grid_fixture): output and input gradient on a
    stride-7 lattice + their norms, every parameter gradient; fp32 end to end 1e-4, bf16 autocast within the
    reference's own bf16 tolerance 4e-2 (tests/distributed/tests_distributed_layers.py:539)"""
    import os
    import makani_amd as ma
    from oracle.make_golden import seeded_field
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "fcn3_config4_grids_721x1440.npz")):
        pytest.skip("fixture not generated")
    g = load_golden("fcn3_config4_grids_721x1440.npz")
    kw = json.loads(str(g["kwargs"]))
    s, xs, gs = kw.pop("subsample"), kw.pop("x_seed"), kw.pop("g_seed")
    model = ma.AtmoSphericNeuralOperatorNet(**kw)
    model.load_state_dict({k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}, strict=True)
    model = model.to("cuda:0")
    nin = len(kw["channel_names"]) + len(kw["aux_channel_names"])
    x = seeded_field(xs, (1, nin, *kw["inp_shape"]), "rand")
    gy = seeded_field(gs, (1, len(kw["channel_names"]), *kw["out_shape"]), "randn")
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6 * x.numel() and abs(float(gy.double().sum()) - float(g["g_sum"])) < 1e-2
    xd = x.to("cuda:0").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = model(xd)
    (y.float() * gy.to("cuda:0")).sum().backward()
    e_y = rel_l2(y.detach().float()[..., ::s, ::s], torch.from_numpy(g["y_sub"]))
    e_gx = rel_l2(xd.grad[..., ::s, ::s], torch.from_numpy(g["gx_sub"]))
    n_y = abs(float(y.detach().double().norm()) - float(g["y_norm"])) / float(g["y_norm"])
    n_gx = abs(float(xd.grad.double().norm()) - float(g["gx_norm"])) / float(g["gx_norm"])
    print(f"FCN3 whole network at config 4's grids amp={amp}: y {e_y:.2e} gx {e_gx:.2e} |y| {n_y:.2e} |gx| {n_gx:.2e}")
    assert e_y < tol and e_gx < tol and n_y < tol and n_gx < tol
    gmax = max(float(np.abs(g[k2]).max()) for k2 in g.files if k2.startswith("grad/"))
    worst = 0.0
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        e = rel_l2(p.grad, ref)
        if ref.is_complex():
            a = (torch.view_as_real(p.grad.detach().cpu()) - torch.view_as_real(ref)).abs().max().item()
        else:
            a = (p.grad.detach().cpu().float() - ref).abs().max().item()
        assert e < 2 * tol or a < (1e-4 if not amp else 4e-2) * gmax, (k, e, a, gmax)
        worst = max(worst, e)
    print(f"    largest parameter-gradient rel-L2: {worst:.2e}")
