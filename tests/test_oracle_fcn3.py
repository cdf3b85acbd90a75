"""The FourCastNet3 restatement (oracle/fcn3.py) against fixtures written by the reference's OWN module
(``makani/models/networks/fourcastnet3.py`` imported through ``oracle/ref_shims.py``; ``python -m oracle.make_golden fcn3`` /
``fcn3_bases``): strict state-dict load, forward output, input gradient and every parameter gradient.  Both sides run the same
restated torch-harmonics operators, so the agreement is to fp32 round-off of a different evaluation order."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

FIXTURES = ["fcn3_small_33x64.npz", "fcn3_options_24x48.npz", "fcn3_piecewise_linear_24x48.npz", "fcn3_zernike_24x48.npz"]


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_fcn3_matches_the_reference_module(name):
    from oracle import fcn3 as of
    g = load_golden(name)
    kwargs = json.loads(str(g["kwargs"]))
    torch.manual_seed(0)
    model = of.AtmoSphericNeuralOperatorNet(**kwargs, some_unknown_trainer_key=1)
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    model.load_state_dict(sd, strict=True)
    assert set(dict(model.named_parameters())) == {k[len("grad/"):] for k in g.files if k.startswith("grad/")}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = model(x)
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_l2(y, torch.from_numpy(g["y"])) < 2e-6
    assert rel_l2(x.grad, torch.from_numpy(g["gx"])) < 2e-6
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        assert rel_l2(p.grad, ref) < 1e-5 or (p.grad - ref).abs().max().item() < 1e-6 * gmax, k


def test_oracle_fcn3_helpers():
    from oracle import fcn3 as of
    atmo, surf, dyn, stat, levels = of.get_channel_groups(["u500", "v500", "t2m", "u850", "v850", "d2", "tcwv"], ["xzen", "xoro"])
    assert atmo == [0, 1, 3, 4] and surf == [2, 5, 6] and dyn == [7] and stat == [8] and levels == [500, 850]
    assert of.get_water_channels(["q500", "t500", "r850", "tcwv", "u10m"]) == [0, 2, 3]
    with pytest.raises(ValueError):
        of.get_channel_groups(["u500", "v500", "u850"])
    x = torch.linspace(-1, 2, 13)
    y = of.soft_clamp(x)
    assert (y[x <= 0] == 0).all() and torch.allclose(y[x >= 0.5], x[x >= 0.5] - 0.25)
