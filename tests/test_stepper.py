"""Rollout wrapper (makani_amd/stepper.py) against golden vectors produced by the reference's own
MultiStepWrapper (makani/models/stepper.py:176-345; generator: oracle/make_golden.py::stepper_fixtures).
Host logic only — the wrapped network here is a small torch module on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

from makani_amd.stepper import MultiStepWrapper, SingleStepWrapper

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multistep_rollout.npz")


def _net(cin, C):
    return torch.nn.Sequential(torch.nn.Conv2d(cin, 5, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(5, C, 1)).double()


def _cases():
    z = np.load(GOLDEN)
    return z, json.loads(str(z["cases"]))


@pytest.mark.parametrize("i", range(6))
def test_rollout_matches_reference_wrapper(i):
    z, cases = _cases()
    c = cases[i]
    x = torch.from_numpy(z[f"{i}_x"]).requires_grad_(True)
    C = x.shape[1] // (c["n_history"] + 1)
    net = _net(x.shape[1], C)
    with torch.no_grad():
        for j, q in enumerate(net.parameters()):
            q.copy_(torch.from_numpy(z[f"{i}_p{j}"]))
    wrap = MultiStepWrapper(net, n_future=c["n_future"], n_history=c["n_history"], push_forward=c["push_forward"],
                            multistep_checkpoint=c["ckpt"])
    wrap.train()
    y = wrap(x)
    assert y.shape == (x.shape[0], (c["n_future"] + 1) * C, x.shape[2], x.shape[3])
    np.testing.assert_allclose(y.detach().numpy(), z[f"{i}_y"], rtol=0, atol=1e-12)
    (y * torch.from_numpy(z[f"{i}_g"])).sum().backward()
    if f"{i}_dx" in z.files:
        np.testing.assert_allclose(x.grad.numpy(), z[f"{i}_dx"], rtol=0, atol=1e-12)
    else:
        assert x.grad is None                 # push-forward: nothing flows back into the rollout's input
    for j, q in enumerate(net.parameters()):
        np.testing.assert_allclose(q.grad.numpy(), z[f"{i}_dp{j}"], rtol=0, atol=1e-12)
    wrap.eval()
    with torch.no_grad():
        np.testing.assert_allclose(wrap(x).numpy(), z[f"{i}_y_eval"], rtol=0, atol=1e-12)


class _Params(dict):
    __getattr__ = dict.__getitem__


def test_from_params_reads_makanis_keys_and_rejects_stages_out_of_scope():
    net = _net(6, 3)
    p = _Params(n_future=2, n_history=1, multistep={"push_forward": True}, multistep_checkpoint=True,
                history_normalization_mode="none")
    w = MultiStepWrapper.from_params(p, lambda: net)
    assert (w.n_future, w.n_history, w.push_forward_mode, w.multistep_checkpoint) == (2, 1, True, True)
    assert w.model is net
    with pytest.raises(NotImplementedError):
        MultiStepWrapper.from_params(_Params(p, history_normalization_mode="mean"), lambda: net)
    with pytest.raises(NotImplementedError):
        MultiStepWrapper.from_params(_Params(p, input_noise={"type": "white"}), lambda: net)
    with pytest.raises(NotImplementedError):
        MultiStepWrapper.from_params(_Params(p, add_zenith=True), lambda: net)
    with pytest.raises(ValueError):
        MultiStepWrapper(net, n_future=-1)


def test_checkpointing_refuses_private_generators_and_history_checks_shapes():
    class Seeded(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rng_cpu = torch.Generator()

        def forward(self, x):
            return x

    with pytest.raises(RuntimeError, match="private RNG"):
        MultiStepWrapper(Seeded(), n_future=1, multistep_checkpoint=True)
    MultiStepWrapper(Seeded(), n_future=1, multistep_checkpoint=False)
    w = MultiStepWrapper(torch.nn.Identity(), n_future=1, n_history=1)
    with pytest.raises(RuntimeError):
        w.append_history(torch.zeros(1, 5, 2, 2), torch.zeros(1, 2, 2, 2))
    with pytest.raises(RuntimeError):
        w.append_history(torch.zeros(1, 6, 2, 2), torch.zeros(1, 2, 2, 2))
    out = w.append_history(torch.arange(6.0).view(1, 6, 1, 1), torch.full((1, 3, 1, 1), 9.0))
    assert out.flatten().tolist() == [3.0, 4.0, 5.0, 9.0, 9.0, 9.0]


def test_single_step_wrapper_forwards_encode_process():
    class Net(torch.nn.Module):
        def forward(self, x):
            return x + 1

        def encode_process(self, x):
            return x * 2

    s = SingleStepWrapper(Net())
    x = torch.ones(2)
    assert torch.equal(s(x), x + 1) and torch.equal(s.encode_process(x), x * 2)
    with pytest.raises(NotImplementedError):
        SingleStepWrapper(torch.nn.Identity()).encode_process(x)
