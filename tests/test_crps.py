"""CRPSLoss on the HIP path (csrc/crps.hip) against fixtures generated from the reference's own
``makani/utils/losses/crps_loss.py`` (``python -m oracle.make_golden crps``): value and forecast gradient for every built
score type, with spatial weights, NaN observations, tied members and E = 1.  fp32 tolerance 1e-5."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2


def test_crps_constructor_contract():
    import makani_amd as ma
    kw = dict(img_shape=(9, 16), crop_shape=(9, 16), crop_offset=(0, 0), channel_names=["a", "b"], grid_type="equiangular")
    m = ma.CRPSLoss(**kw)
    assert m.crps_type == "skillspread" and m.n_channels == 2 and m.quad_weight_split.shape == (1, 1, 144)
    assert abs(float(m.quad_weight_split.sum()) - 1.0) < 1e-6
    with pytest.raises(ValueError):
        ma.CRPSLoss(crps_type="nonsense", **kw)
    assert ma.CRPSLoss(crps_type="cdf", **kw).crps_type == "cdf"
    assert ma.CRPSLoss(crps_type="cdf", ensemble_weights=torch.ones(4), **kw).ensemble_weights.shape == (4,)
    with pytest.raises(NotImplementedError):
        ma.CRPSLoss(crps_type="gauss", alpha=0.9, **kw)
    with pytest.raises(NotImplementedError):
        ma.CRPSLoss(crps_type="cdf", alpha=0.9, **kw)
    with pytest.raises(NotImplementedError):
        ma.CRPSLoss(ensemble_weights=torch.ones(4), **kw)            # skillspread: constant weights only (crps_loss.py:408-410)
    # the reference's "probability weighted moment" branch takes ensemble_weights as well (crps_loss.py:397-407; its kernel
    # ignores their values): a reference config with PWM + weights must construct
    assert ma.CRPSLoss(crps_type="probability weighted moment", ensemble_weights=torch.ones(4), **kw).ensemble_weights.shape == (4,)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 2, 9, 16), torch.zeros(2, 2, 9, 16))          # forecasts need the ensemble dimension


@pytest.mark.gpu
def test_crps_matches_reference_golden():
    import makani_amd as ma
    g = load_golden("crps_loss.npz")
    cases = json.loads(str(g["cases"]))
    for i, c in enumerate(cases):
        C = g[f"{i}_o"].shape[1]
        ens_w = torch.from_numpy(g[f"{i}_ens_w"]) if f"{i}_ens_w" in g.files else None
        mod = ma.CRPSLoss(img_shape=tuple(c["img"]), crop_shape=tuple(c["img"]), crop_offset=(0, 0),
                          channel_names=[str(k) for k in range(C)], grid_type=c["grid"], crps_type=c["crps_type"],
                          alpha=c["alpha"], ensemble_weights=ens_w).to("cuda:0")
        f = torch.from_numpy(g[f"{i}_f"]).to("cuda:0").requires_grad_(True)
        o = torch.from_numpy(g[f"{i}_o"]).to("cuda:0")
        w = torch.from_numpy(g[f"{i}_wgt"]).to("cuda:0") if f"{i}_wgt" in g.files else None
        out = mod(f, o, w)
        (out * torch.from_numpy(g[f"{i}_g"]).to("cuda:0")).sum().backward()
        assert out.shape == g[f"{i}_out"].shape
        assert torch.isfinite(out).all(), c
        assert rel_l2(out, torch.from_numpy(g[f"{i}_out"])) < 1e-5, (c, out, g[f"{i}_out"])
        assert rel_l2(f.grad, torch.from_numpy(g[f"{i}_df"])) < 1e-5, c


@pytest.mark.gpu
def test_crps_cdf_equals_fair_score_identity_and_large_ensemble_errors():
    """the piecewise-integrated "cdf" score is the classic ensemble CRPS  mean|f - o| - sum_ij |f_i - f_j| / (2 E^2)  (the
    "naive skillspread" form with alpha = 0 differs from it by the factor (E - 1) / E on the spread), for every ensemble size
    2..32; more than 32 members raise"""
    import makani_amd as ma
    torch.manual_seed(9)
    kw = dict(img_shape=(16, 32), crop_shape=(16, 32), crop_offset=(0, 0), channel_names=["a", "b"], grid_type="equiangular")
    cdf = ma.CRPSLoss(crps_type="cdf", **kw).to("cuda:0")
    q = cdf.quad_weight_split.reshape(16, 32)
    for E in (2, 3, 9, 11, 17, 31, 32):
        f = torch.randn(1, E, 2, 16, 32, device="cuda:0", dtype=torch.float64)
        o = torch.randn(1, 2, 16, 32, device="cuda:0", dtype=torch.float64)
        skill = (f - o.unsqueeze(1)).abs().mean(dim=1)
        spread = (f.unsqueeze(1) - f.unsqueeze(2)).abs().sum(dim=(1, 2)) / (2.0 * E * E)
        ref = ((skill - spread) * q).sum(dim=(-2, -1))
        got = cdf(f.float(), o.float())
        assert rel_l2(got, ref) < 1e-5, (E, got, ref)
    with pytest.raises(NotImplementedError):
        cdf(torch.zeros(1, 33, 2, 16, 32, device="cuda:0"), torch.zeros(1, 2, 16, 32, device="cuda:0"))


@pytest.mark.gpu
def test_crps_cdf_nan_observation_propagates_and_pwm_ignores_weights():
    """(i) the reference's "cdf" form does not mask NaN observations — its last term torch.clamp(obs - forecast,
    min=0) propagates the NaN into the point's score and hence into the (B, C) loss (crps_loss.py:55-122); the kernel's fmaxf
    would have dropped it.  (ii) "probability weighted moment" accepts ensemble_weights and ignores their values."""
    import makani_amd as ma
    kw = dict(img_shape=(9, 16), crop_shape=(9, 16), crop_offset=(0, 0), channel_names=["a", "b"], grid_type="equiangular")
    torch.manual_seed(1)
    f = torch.randn(1, 4, 2, 9, 16, device="cuda:0", requires_grad=True)
    o = torch.randn(1, 2, 9, 16, device="cuda:0")
    o_nan = o.clone()
    o_nan[0, 1, 3, 5] = float("nan")
    cdf = ma.CRPSLoss(crps_type="cdf", **kw).to("cuda:0")
    out = cdf(f, o_nan)
    assert torch.isfinite(out[0, 0]) and torch.isnan(out[0, 1])           # only the channel that holds the NaN
    assert torch.equal(out[0, 0], cdf(f, o)[0, 0])
    out[0, 0].backward()
    assert torch.isfinite(f.grad).all()
    pwm = ma.CRPSLoss(crps_type="probability weighted moment", **kw).to("cuda:0")
    pwm_w = ma.CRPSLoss(crps_type="probability weighted moment", ensemble_weights=torch.tensor([0.1, 0.2, 0.3, 0.4]), **kw).to("cuda:0")
    assert torch.equal(pwm(f, o), pwm_w(f, o))


@pytest.mark.gpu
def test_crps_bf16_forecasts_and_properties():
    import makani_amd as ma
    torch.manual_seed(1)
    mod = ma.CRPSLoss(img_shape=(32, 64), crop_shape=(32, 64), crop_offset=(0, 0), channel_names=["a"], grid_type="equiangular").to("cuda:0")
    f = torch.randn(2, 6, 1, 32, 64, device="cuda:0")
    o = torch.randn(2, 1, 32, 64, device="cuda:0")
    ref = mod(f, o)
    lo = mod(f.bfloat16(), o)
    assert rel_l2(lo, ref) < 1e-2
    # permuting the members leaves the score unchanged; a perfect, spread-free ensemble scores 0
    assert rel_l2(mod(f[:, [3, 0, 5, 1, 4, 2]], o), ref) < 1e-6
    z = mod(o.unsqueeze(1).expand(2, 6, 1, 32, 64).contiguous(), o)
    assert z.abs().max() < 1e-6
    # naive and rank forms of the fair score agree when no two members are equal
    naive = ma.CRPSLoss(img_shape=(32, 64), crop_shape=(32, 64), crop_offset=(0, 0), channel_names=["a"], grid_type="equiangular",
                        crps_type="naive skillspread").to("cuda:0")
    assert rel_l2(naive(f, o), ref) < 1e-5


def test_spectral_crps_constructor_contract():
    """SpectralCRPSLoss: SpectralBaseLoss' transform and Parseval weights (base_loss.py:345-404), the reference's option errors"""
    import makani_amd as ma
    kw = dict(img_shape=(12, 24), crop_shape=(12, 24), crop_offset=(0, 0), channel_names=["a", "b"], grid_type="legendre-gauss")
    m = ma.SpectralCRPSLoss(**kw)
    assert m.crps_type == "skillspread" and m.absolute and m.lm_weights.shape == (m.sht.lmax, m.sht.mmax) == (11, 11)     # the grid's bandlimit
    assert torch.allclose(m.lm_weights[:, 0], torch.full((11,), 1.0 / (4 * np.pi))) and torch.allclose(m.lm_weights[:, 1:], torch.full((11, 10), 2.0 / (4 * np.pi)))
    assert ma.SpectralCRPSLoss(lmax=7, **kw).lm_weights.shape == (7, 7)
    assert ma.SpectralCRPSLoss(crps_type="cdf", **kw).crps_type == "cdf" and not ma.SpectralCRPSLoss(absolute=False, **kw).absolute
    assert not ma.SpectralCRPSLoss(ensemble_distributed=True, **kw).ensemble_distributed      # no split "ensemble" group: serial, as the reference
    for bad, exc in ((dict(crps_type="naive skillspread"), ValueError),
                     (dict(absolute=False, crps_type="gauss"), ValueError), (dict(crps_type="gauss", alpha=0.9), NotImplementedError),
                     (dict(ensemble_weights=torch.ones(4)), NotImplementedError)):
        with pytest.raises(exc):
            ma.SpectralCRPSLoss(**bad, **kw)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 2, 12, 24), torch.zeros(2, 2, 12, 24))


@pytest.mark.gpu
def test_spectral_crps_matches_reference_golden():
    """value and forecast gradient against fixtures generated from the reference's SpectralCRPSLoss
    (``python -m oracle.make_golden crps_spectral``); fp32, 1e-5 on the value, 2e-5 on the gradient through the transform"""
    import makani_amd as ma
    g = load_golden("crps_spectral.npz")
    cases = json.loads(str(g["cases"]))
    for i, c in enumerate(cases):
        C = g[f"{i}_o"].shape[1]
        mod = ma.SpectralCRPSLoss(img_shape=tuple(c["img"]), crop_shape=tuple(c["img"]), crop_offset=(0, 0),
                                  channel_names=[str(k) for k in range(C)], grid_type=c["grid"], lmax=c["lmax"],
                                  crps_type=c["crps_type"], alpha=c["alpha"], absolute=c.get("absolute", True)).to("cuda:0")
        f = torch.from_numpy(g[f"{i}_f"]).to("cuda:0").requires_grad_(True)
        o = torch.from_numpy(g[f"{i}_o"]).to("cuda:0")
        w = torch.from_numpy(g[f"{i}_wgt"]).to("cuda:0") if f"{i}_wgt" in g.files else None
        out = mod(f, o, w)
        (out * torch.from_numpy(g[f"{i}_g"]).to("cuda:0")).sum().backward()
        assert out.shape == g[f"{i}_out"].shape and torch.isfinite(out).all(), c
        assert rel_l2(out, torch.from_numpy(g[f"{i}_out"])) < 1e-5, (c, out, g[f"{i}_out"])
        assert torch.isfinite(f.grad).all()
        if np.isfinite(g[f"{i}_df"]).all():
            assert rel_l2(f.grad, torch.from_numpy(g[f"{i}_df"])) < 2e-5, c
        else:
            # "gauss": the coefficients with l < m are exact zeros in every member, the ensemble spread there is 0 and the
            # reference's autograd returns NaN for the whole gradient (0 * inf in the derivative of the standard deviation);
            # the HIP kernel's gradient is finite (the score at zero spread has the subgradient 0), the VALUE is pinned above
            assert c["crps_type"] == "gauss"


def _worker_ensemble(rank, world, port):
    """ensemble_distributed=True (crps_loss.py:305-307,362-373,441-442,566-581): 2 batch entries x 2 ensemble ranks on one GPU,
    three members per rank: value and forecast gradient of CRPSLoss (every score type) and SpectralCRPSLoss against the
    serial modules on the gathered ensemble"""
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import makani_amd as ma
        import makani_amd.comm as mcomm
        dev = "cuda:0"
        mcomm.init(1, 1, ensemble=2)
        ie, ib = mcomm.get_rank("ensemble"), mcomm.get_rank("batch")
        assert (mcomm.get_size("ensemble"), mcomm.get_size("batch"), mcomm.get_size("data")) == (2, 2, 4) and rank == ib * 2 + ie
        torch.manual_seed(3)
        img, C, El = (19, 36), 3, 3
        f_all = torch.randn(2, 2 * El, C, *img)
        o_all = torch.randn(2, C, *img)
        g_all = torch.randn(2, C)
        w_all = torch.rand(2, C, *img) + 0.5
        kw = dict(img_shape=img, crop_shape=img, crop_offset=(0, 0), channel_names=[str(k) for k in range(C)], grid_type="equiangular")
        cases = [(ma.CRPSLoss, t, True) for t in ("skillspread", "cdf", "probability weighted moment", "gauss", "naive skillspread")]
        cases += [(ma.SpectralCRPSLoss, "skillspread", False), (ma.SpectralCRPSLoss, "cdf", False)]
        for cls, ctype, use_w in cases:
            ser = cls(crps_type=ctype, **kw).to(dev)
            par = cls(crps_type=ctype, ensemble_distributed=True, **kw).to(dev)
            assert par.ensemble_distributed and not ser.ensemble_distributed
            fs = f_all[ib:ib + 1].to(dev).requires_grad_(True)
            o, g = o_all[ib:ib + 1].to(dev), g_all[ib:ib + 1].to(dev)
            w = w_all[ib:ib + 1].to(dev) if use_w else None
            ref = ser(fs, o, w)
            (ref * g).sum().backward()
            fl = f_all[ib:ib + 1, ie * El:(ie + 1) * El].to(dev).requires_grad_(True)
            out = par(fl, o, w)
            (out * g).sum().backward()
            assert rel_l2(out, ref) < 1e-5, (cls.__name__, ctype, out, ref)
            assert rel_l2(fl.grad, fs.grad[:, ie * El:(ie + 1) * El]) < 2e-5, (cls.__name__, ctype)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ensemble_parallel_crps_matches_serial():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_ensemble, args=(4, port), nprocs=4, join=True)
