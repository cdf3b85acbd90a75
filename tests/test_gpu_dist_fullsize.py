"""BASELINE configs[2] and [4] WHOLE and at FULL SIZE before any multi-GPU box sees them:
``sfno_sc3_layers8_edim384`` at 721 x 1440 x 73 (384 channels, 8 layers, latitudes [181, 181, 181, 178], l-sharded 283 MB
spectral weights, the package's gradient-reduction hooks) under h4 w1 and h4 w2 spatial model parallelism — N ranks
share ONE GPU and exchange through gloo — forward + backward, every rank's output shard, input-gradient shard and REDUCED
parameter gradients against (1) the serial HIP model and (2) the CPU oracle (tests/_fullsize.py), and the same split
around ``MultiStepWrapper(n_future=3)`` (= multistep_count 4) with rollout checkpointing under bf16 autocast.

What these tests pin that the toy-grid tests cannot: the fused exchange schedule (makani_amd/dist_pipeline.py) with TWO
latitude chunks on the 721-row transforms, the weight-stationary channel GEMMs on ragged 130 k-pixel shards, the
distributed instance norm over 8 shards, the l-sharded dhconv kernels with their triangular shard offsets at L = 240.
Reference twins: tests/distributed/tests_distributed_model.py:218-330, makani/mpu/fft.py:148-182,214-249,
makani/mpu/mappings.py:321-525, makani/models/stepper.py:224-284.  Tolerances: fp32 <= 1e-4 end to end
(tests/distributed/tests_distributed_layers.py:71-76); bf16 relative to the reference's own CPU bf16 arithmetic on the
same shard (tests/test_gpu_headline.py explains why the flat 2e-2 cannot hold through eight bf16 layers)."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist

from _fullsize import CACHE_DIR, CONFIG2, config2_oracle, log_line, spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
BIG = ("filter.filter.weight", "fwd.0.weight", "fwd.2.weight", "fwd.3.weight", "outer_skip.weight")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _r(t):
    t = t.detach()
    if t.is_complex():
        t = torch.view_as_real(t.resolve_conj())
    return t


def _rel(a, b):
    a, b = _r(a).cpu().double(), _r(b).cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _absmax(a, b):
    return float((_r(a).cpu().double() - _r(b).cpu().double()).abs().max())


# --------------------------------------------------------------------------- #
# the serial HIP model in the pytest process: reference (1), kept in a file the ranks map
# --------------------------------------------------------------------------- #
@pytest.fixture(scope="module")
def oracle():
    return config2_oracle()


@pytest.fixture(scope="module")
def serial_hip(oracle):
    """forward + backward of the SERIAL HIP model (fp32) on the oracle's weights / input / cotangent"""
    import makani_amd as ma
    path = os.path.join(CACHE_DIR, f"config2_hip_serial_{os.getpid()}.pt")
    model = ma.SphericalFourierNeuralOperatorNet(**CONFIG2)
    model.load_state_dict(oracle["state"], strict=True)
    model = model.to("cuda:0").eval()
    xd = oracle["x"].to("cuda:0").requires_grad_(True)
    y = model(xd)
    (y * oracle["g"].to("cuda:0")).sum().backward()
    out = dict(y=y.detach().cpu(), gx=xd.grad.cpu(),
               grads={n: (torch.view_as_real(p.grad).cpu().contiguous() if p.grad.is_complex() else p.grad.cpu().contiguous())
                      for n, p in model.named_parameters()})
    e = _rel(out["y"], oracle["y"])
    assert e < TOL, e
    os.makedirs(CACHE_DIR, exist_ok=True)
    torch.save(out, path)
    del model, xd, y, out
    torch.cuda.empty_cache()
    yield path
    try:
        os.remove(path)
    except OSError:
        pass


def _setup_rank(rank, world, port, h, w):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    torch.set_num_threads(8)
    from _fullsize import share_gpu
    share_gpu(rank, world)                  # disjoint compute units per rank, set before the first GPU call (tests/_fullsize.py)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import makani_amd.comm as mcomm
    return mcomm.init(h, w)


def _contract(state, scale):
    """the oracle's weights made NON-CHAOTIC for rollouts: the big skip becomes the identity (73 -> 73 channels) and the decoder's
    output layer is scaled by ``scale``, so one step is x -> x + scale * f(x) and a perturbation of x (bf16 rounding) is carried
    into the next step by I + scale * J_f instead of by the full Jacobian of a random network: bf16 rounding stays bounded
    through the four steps and a distributed-vs-serial bf16 comparison can fail for a wrong shard instead of drowning in chaos
    (VERDICT r5 weak #1b).  Scaling skip / norm weights does not do it — the instance norms undo every amplitude change
    (measured: y 0.68 / 0.52 / 0.28 from fp32 at 1/2, 1/4, 1/8; gradients 1.4)."""
    if scale is None:
        return state
    out = dict(state)
    w = state["residual_transform.weight"]
    assert w.shape[0] == w.shape[1], w.shape
    out["residual_transform.weight"] = torch.eye(w.shape[0], dtype=w.dtype).reshape(w.shape)
    out["decoder.fwd.2.weight"] = state["decoder.fwd.2.weight"] * scale
    return out


def _sharded_model(oracle, ih, scale=None):
    """the distributed network of this rank with the oracle's weights (spectral weights: this polar rank's degrees)"""
    import makani_amd as ma
    model = ma.SphericalFourierNeuralOperatorNet(**CONFIG2)
    assert model.spatial_parallel
    td = model.trans_down
    l0, ll = sum(td.l_shapes[:ih]), td.l_shapes[ih]
    own = model.state_dict()
    state = _contract(oracle["state"], scale)
    with torch.no_grad():
        for k in own:
            src = state[k]
            if k.endswith("filter.filter.weight"):
                src = src[..., l0:l0 + ll]
            assert own[k].shape == src.shape, (k, own[k].shape, src.shape)
            own[k].copy_(src)
    return model.to("cuda:0"), (l0, ll)


def _assert_fused_two_chunks(model, h, w):
    """the schedule that ran: fused for every transform, two latitude chunks on the 721-row transforms"""
    from makani_amd import dist_pipeline as dp
    assert dp.FALLBACKS == [], dp.FALLBACKS
    for name in ("trans_down", "itrans_up", "trans", "itrans"):
        T = getattr(model, name)
        assert T.__dict__.get("_fused_ok") and all(T._fused_ok.values()), (name, T.__dict__.get("_fused_ok"))
        plans = T.__dict__.get("_plans", {})
        assert plans, f"{name}: no plan of the fused schedule was built"
        want = 2 if T.nlat == 721 else 1          # 240 latitudes over h = 4: 60 per rank, one chunk (dist_pipeline.py: two need >= 64)
        assert all(p.nc == want for p in plans.values()), (name, [p.nc for p in plans.values()])
    assert model.trans_down.lat_shapes == [181, 181, 181, 178] and model.trans_down.l_shapes == [60, 60, 60, 60]
    if w == 2:
        assert model.trans_down.lon_shapes == [720, 720] and model.trans_down.m_shapes == [121, 120]


def _worker_fwd_bwd(rank, world, port, h, w, amp, hip_path):
    _, ih, iw = _setup_rank(rank, world, port, h, w)
    try:
        import makani_amd.distributed as thd
        from _fullsize import config2_oracle, log_line
        dev = torch.device("cuda:0")
        oracle = config2_oracle()
        hip = torch.load(hip_path, mmap=True, weights_only=True)
        t0 = time.time()
        model, (l0, ll) = _sharded_model(oracle, ih)
        net = thd.init_gradient_reduction_hooks(model, dev)           # mappings.py:321-525: sums over h x w / w complete in backward()
        td = model.trans_down
        lat0, lon0 = sum(td.lat_shapes[:ih]), sum(td.lon_shapes[:iw])
        hl, wl = td.lat_shapes[ih], td.lon_shapes[iw]
        sl = (..., slice(lat0, lat0 + hl), slice(lon0, lon0 + wl))
        xl = oracle["x"][sl].to(dev).requires_grad_(True)
        gl = oracle["g"][sl].to(dev)
        torch.cuda.synchronize()
        t1 = time.time()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yl = net(xl)
        (yl.float() * gl).sum().backward()
        torch.cuda.synchronize()
        t2 = time.time()
        assert yl.shape == (1, 73, hl, wl)
        _assert_fused_two_chunks(model, h, w)

        errs = {}
        if not amp:
            errs["y"] = (_rel(yl, oracle["y"][sl]), _rel(yl, hip["y"][sl]))
            errs["gx"] = (_rel(xl.grad, oracle["gx"][sl]), _rel(xl.grad, hip["gx"][sl]))
        else:
            errs["y"] = (_rel(yl.float(), oracle["y"][sl]), _rel(oracle["y_bf16"][sl], oracle["y"][sl]))
            errs["gx"] = (_rel(xl.grad, oracle["gx"][sl]), _rel(oracle["bf16_gx"][sl], oracle["gx"][sl]))
        gmax = max(float(_r(v).abs().max()) for v in oracle["grads"].values())
        worst, bad = ("", 0.0), {}
        for n, p in model.named_parameters():
            spectral = n.endswith("filter.filter.weight")
            pick = (lambda t: t[..., l0:l0 + ll]) if spectral else (lambda t: t)
            ref = pick(oracle["grads"][n])
            if not amp:
                hg = hip["grads"][n]
                hg = hg[..., l0:l0 + ll, :] if spectral else hg                  # (the HIP file holds real views)
                e, e2, a = _rel(p.grad, ref), _rel(_r(p.grad), hg), _absmax(p.grad, ref)
                errs[n] = (e, e2)
                # gradients that are exactly zero by construction (a per-channel constant in front of an instance norm) are
                # accepted on the absolute scale of the largest gradient entry, as in tests/test_gpu_headline.py
                if not ((e < TOL and e2 < TOL) or a < 1e-5 * gmax):
                    bad[n] = (e, e2, a)
                if e > worst[1] and a >= 1e-5 * gmax:
                    worst = (n, e)
            elif n.endswith(BIG):
                e, e_ref = _rel(p.grad, ref), _rel(pick(oracle["bf16_grads"][n]), ref)
                errs[n] = (e, e_ref)
                if not (e < 0.15 and e <= 1.25 * e_ref):
                    bad[n] = (e, e_ref)
                if e / max(e_ref, 1e-30) > worst[1]:
                    worst = (n, e / max(e_ref, 1e-30))
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        mode = "bf16 autocast: HIP vs fp32 oracle / the oracle's own CPU bf16 on the same shard" if amp \
            else "fp32: vs oracle / vs serial HIP"
        log_line(f"config2 h{h}w{w} rank {rank} (ih {ih}, iw {iw}; {hl}x{wl} px, l {l0}..{l0 + ll}) {mode}:  "
                 f"y {errs['y'][0]:.2e} / {errs['y'][1]:.2e}  gx {errs['gx'][0]:.2e} / {errs['gx'][1]:.2e}  "
                 + (f"worst parameter gradient, as a RATIO to the oracle's own bf16 distance (gate 1.25): {worst[0]} x{worst[1]:.2f}  " if amp
                    else f"worst parameter gradient {worst[0]} {worst[1]:.2e}  ") +
                 f"[blocks.0 spectral {errs['blocks.0.filter.filter.weight'][0]:.2e}, blocks.7 spectral "
                 f"{errs['blocks.7.filter.filter.weight'][0]:.2e}, encoder {errs['encoder.fwd.0.weight'][0]:.2e}]  "
                 f"peak {peak:.1f} GiB, setup {t1 - t0:.0f} s, fwd+bwd over gloo {t2 - t1:.1f} s")
        if not amp:
            assert errs["y"][0] < TOL and errs["y"][1] < TOL and errs["gx"][0] < TOL and errs["gx"][1] < TOL, (rank, errs["y"], errs["gx"])
        else:
            # a shard's relative error scatters around the whole field's: 1.25 x the reference's own bf16 arithmetic on THIS shard
            assert errs["y"][0] < 6e-2 * 1.5 and errs["y"][0] <= 1.25 * errs["y"][1], (rank, errs["y"])
            assert errs["gx"][0] < 0.15 and errs["gx"][0] <= 1.25 * errs["gx"][1], (rank, errs["gx"])
        assert not bad, (rank, bad)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,amp", [(4, 1, False), (4, 2, False), (4, 2, True)])
def test_config2_fullsize_spatial_parallel_fwd_bwd(h, w, amp, oracle, serial_hip):
    """BASELINE configs[2] (h = 4) and the model instance of configs[4] (h4 w2): whole network, full size, forward + backward
    with the gradient hooks; fused schedule with two latitude chunks asserted; per-rank peak memory logged"""
    log_line(f"--- test_config2_fullsize_spatial_parallel_fwd_bwd h{h} w{w} {'bf16 autocast' if amp else 'fp32'} ---")
    spawn(_worker_fwd_bwd, (h * w, _free_port(), h, w, amp, serial_hip), h * w, timeout_s=900)


# --------------------------------------------------------------------------- #
# BASELINE configs[4]: multistep_count = 4 around the h4 w2 network, bf16 autocast, rollout checkpointing
# --------------------------------------------------------------------------- #
NF = 3


def _serial_rollout_pass(model, net, oracle, G, amp):
    model.zero_grad(set_to_none=True)
    xd = oracle["x"].to("cuda:0").requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = net(xd)
    (y.float() * G.to("cuda:0")).sum().backward()
    return dict(y=y.float().detach().cpu(), gx=xd.grad.cpu(),
                grads={n: _r(p.grad).cpu().contiguous() for n, p in model.named_parameters() if n.endswith(BIG)})


@pytest.fixture(scope="module")
def serial_rollout(oracle):
    """the 4-step rollout of the SERIAL HIP network with rollout checkpointing: (1) fp32 on the oracle's weights = the reference
    of the fp32 comparison; (2) on CONTRACTIVE weights (``_contract``; the larger of the scales 0.02, 0.01 whose serial bf16
    rollout stays within 1e-1 (output) / 1.5e-1 (input gradient) of its fp32 rollout) fp32 = the reference of the bf16
    comparison and bf16 autocast = its yardstick"""
    import makani_amd as ma
    from makani_amd.stepper import MultiStepWrapper
    path = os.path.join(CACHE_DIR, f"config2_hip_rollout_{os.getpid()}.pt")
    model = ma.SphericalFourierNeuralOperatorNet(**CONFIG2)
    model.load_state_dict(oracle["state"], strict=True)
    model = model.to("cuda:0")
    net = MultiStepWrapper(model, n_future=NF, multistep_checkpoint=True).train()
    G = torch.randn(1, 73 * (NF + 1), 721, 1440, generator=torch.Generator().manual_seed(5))
    plain = _serial_rollout_pass(model, net, oracle, G, False)
    chaos = _serial_rollout_pass(model, net, oracle, G, True)
    log_line(f"--- serial HIP 4-step rollout at 721x1440, the oracle's weights: bf16 autocast vs fp32: y {_rel(chaos['y'], plain['y']):.2e}  "
             f"gx {_rel(chaos['gx'], plain['gx']):.2e} (chaotic: not a gate) ---")
    del chaos
    scale, f32, yard = None, None, None
    for sc in (0.02, 0.01):          # (0.2 / 0.1 / 0.05 were measured: y 9.0e-2 / 3.7e-2 / 1.2e-2, gx 0.64 / 0.42 / 0.18: docs/LAB_NOTEBOOK.md 6.6)
        model.load_state_dict(_contract(oracle["state"], sc), strict=True)
        f32 = _serial_rollout_pass(model, net, oracle, G, False)
        b16 = _serial_rollout_pass(model, net, oracle, G, True)
        yard = dict(y=_rel(b16["y"], f32["y"]), gx=_rel(b16["gx"], f32["gx"]),
                    grads={n: _rel(b16["grads"][n], f32["grads"][n]) for n in f32["grads"]})
        log_line(f"--- serial HIP 4-step rollout, contractive weights (scale {sc}): bf16 autocast vs fp32: y {yard['y']:.2e}  gx {yard['gx']:.2e}  "
                 f"weight gradients {min(yard['grads'].values()):.2e}..{max(yard['grads'].values()):.2e}  "
                 f"peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB ---")
        del b16
        if yard["y"] <= 1e-1 and yard["gx"] <= 1.5e-1:
            scale = sc
            break
    assert scale is not None, ("no contractive setup found", yard)
    out = dict(G=G, y=plain["y"], gx=plain["gx"], grads=plain["grads"], scale=torch.tensor(scale),
               c_y=f32["y"], c_gx=f32["gx"], c_grads=f32["grads"],
               yard_y=torch.tensor(yard["y"]), yard_gx=torch.tensor(yard["gx"]),
               yard_grads={n: torch.tensor(v) for n, v in yard["grads"].items()})
    torch.save(out, path)
    del model, net, plain, f32, out
    torch.cuda.empty_cache()
    yield path
    try:
        os.remove(path)
    except OSError:
        pass


def _worker_rollout(rank, world, port, h, w, amp, ref_path):
    _, ih, iw = _setup_rank(rank, world, port, h, w)
    try:
        import makani_amd.distributed as thd
        from makani_amd.stepper import MultiStepWrapper
        from _fullsize import config2_oracle, log_line
        dev = torch.device("cuda:0")
        oracle = config2_oracle()
        ref = torch.load(ref_path, mmap=True, weights_only=True)
        scale = float(ref["scale"]) if amp else None          # bf16: the contractive weights (serial_rollout)
        model, (l0, ll) = _sharded_model(oracle, ih, scale)
        net = thd.init_gradient_reduction_hooks(model, dev)
        net = MultiStepWrapper(net, n_future=NF, multistep_checkpoint=True).train()
        ry, rgx, rgrads = (ref["c_y"], ref["c_gx"], ref["c_grads"]) if amp else (ref["y"], ref["gx"], ref["grads"])
        td = model.trans_down
        lat0, lon0 = sum(td.lat_shapes[:ih]), sum(td.lon_shapes[:iw])
        hl, wl = td.lat_shapes[ih], td.lon_shapes[iw]
        sl = (..., slice(lat0, lat0 + hl), slice(lon0, lon0 + wl))
        xl = oracle["x"][sl].to(dev).requires_grad_(True)
        torch.cuda.synchronize()
        t1 = time.time()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yl = net(xl)
        assert yl.shape == (1, 73 * (NF + 1), hl, wl)
        (yl.float() * ref["G"][sl].to(dev)).sum().backward()
        torch.cuda.synchronize()
        t2 = time.time()
        _assert_fused_two_chunks(model, h, w)
        e_y, e_gx = _rel(yl.float(), ry[sl]), _rel(xl.grad, rgx[sl])
        y_y, y_gx = float(ref["yard_y"]), float(ref["yard_gx"])
        worst, bad = ("", 0.0), {}
        for n, p in model.named_parameters():
            if not n.endswith(BIG):
                continue
            r = rgrads[n]
            r = r[..., l0:l0 + ll, :] if n.endswith("filter.filter.weight") else r
            e, yd = _rel(_r(p.grad), r), float(ref["yard_grads"][n])
            tol = 1.25 * yd if amp else TOL_ROLLOUT
            if e > tol:
                bad[n] = (e, tol)
            if e / tol > worst[1]:
                worst = (n, e / tol)
        if amp:
            what = (f"bf16 autocast on contractive weights (scale {scale}), vs the serial HIP fp32 rollout (distributed bf16 / serial "
                    f"bf16, gate 1.25 x):  y {e_y:.2e} / {y_y:.2e}  gx {e_gx:.2e} / {y_gx:.2e}")
        else:
            what = f"fp32, vs the serial HIP fp32 rollout:  y {e_y:.2e}  gx {e_gx:.2e}  (gate {TOL_ROLLOUT:.0e})"
        log_line(f"config2 multistep 4 (checkpointed) h{h}w{w} rank {rank}: {what}  worst weight gradient / its gate {worst[0]} "
                 f"{worst[1]:.2f}  peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, rollout fwd+bwd over gloo {t2 - t1:.1f} s")
        if amp:
            # contractive weights: bf16 rounding stays bounded through the four steps (the serial bf16 rollout sits y_y / y_gx from
            # its fp32 rollout, both <= 0.15), so a shard that is wrong by more than a quarter of that fails here; the composition
            # itself (rollout, checkpoint recomputation, hooks) is pinned at 1e-3 by the fp32 run on the oracle's own weights
            assert e_y <= 1.25 * y_y and e_gx <= 1.25 * y_gx, (rank, e_y, y_y, e_gx, y_gx)
        else:
            assert e_y < TOL_ROLLOUT and e_gx < TOL_ROLLOUT, (rank, e_y, e_gx)
        assert not bad, (rank, bad)
        dist.barrier()
    finally:
        dist.destroy_process_group()


TOL_ROLLOUT = 1e-3      # four chained steps: the fp32 differences of one step (3e-6 / 6e-6) grow with every re-entry


@pytest.mark.parametrize("amp", [False, True])
def test_config5_fullsize_multistep4_h4w2_checkpointed(amp, oracle, serial_rollout):
    """BASELINE configs[4]: SFNO 721 x 1440 x 73, multistep_count = 4, h = 4, w = 2 — 8 ranks on one GPU, all four outputs, the
    input gradient through the rollout and the reduced gradients of the spectral and channel-GEMM weights; fp32 (pins the
    composition: rollout, checkpoint recomputation through the distributed transforms, gradient hooks) and bf16 AMP (the
    configuration's precision) on CONTRACTIVE weights, where the serial bf16 rollout stays within 0.1 of its fp32 rollout and the
    distributed one must stay within 1.25 x that distance — a gate a wrong shard fails"""
    log_line(f"--- test_config5_fullsize_multistep4_h4w2_checkpointed {'bf16 autocast' if amp else 'fp32'} ---")
    spawn(_worker_rollout, (8, _free_port(), 4, 2, amp, serial_rollout), 8, timeout_s=1200)
