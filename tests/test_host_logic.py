"""CPU tests: the C-ABI library loads and exports every symbol the header declares, the host-side
precompute agrees with the oracle, the drop-in modules keep the reference's interface
(constructor, attributes, state_dict, error behaviour) and refuse to run without the HIP path."""
import ctypes
import json
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

SFNO_GOLDEN = ["sfno_tiny_64x128.npz", "sfno_small_37x72.npz", "sfno_s2norm_resample_33x64.npz",
               "sfno_posembed_direct_19x36.npz", "sfno_posembed_frequency_19x36.npz", "sfno_options_a_24x48.npz",
               "sfno_options_b_24x48.npz", "sfno_layernorm_24x48.npz"]

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from makani_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "makani_amd.h")).read()
    declared = set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/makani_amd.h but not exported"
    assert set(_lib.EXPORTS) == declared, (set(_lib.EXPORTS) ^ declared)
    assert _lib.lib().mk_version() >= 100


def test_gemm_descriptor_matches_header_layout():
    from makani_amd._lib import MkGemm
    header = open(os.path.join(ROOT, "include", "makani_amd.h")).read()
    body = header[header.index("typedef struct MkGemm {"):header.index("} MkGemm;")]
    names = re.findall(r"\b([a-zA-Z_]+)\s*(?=[,;])", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    names = [n for n in names if n not in ("float", "int", "long", "const")]
    assert names == [f for f, _ in MkGemm._fields_]


def test_argument_validation_without_gpu():
    """invalid descriptors are rejected on the host before any launch (no GPU needed)"""
    from makani_amd._lib import MkGemm, lib
    g = MkGemm()
    rc = lib().mk_sgemm_batched(ctypes.byref(g), None)
    assert rc < 0 and b"null" in lib().mk_last_error()
    r = (ctypes.c_int * 2)(4, 3)
    rc = lib().mk_rfft_rows(ctypes.c_void_p(16), 0, ctypes.c_void_p(16), ctypes.c_void_p(16), r, 2, 1, 1, 4, 8, 25, 5,
                            1.0, 1.0, 1.0, None)
    assert rc < 0 and b"even" in lib().mk_last_error()
    rc = lib().mk_rfft_rows(ctypes.c_void_p(16), 0, ctypes.c_void_p(16), ctypes.c_void_p(16), r, 2, 1, 1, 4, 8, 26, 5,
                            1.0, 1.0, 1.0, None)
    assert rc < 0 and b"radix product" in lib().mk_last_error()


@pytest.mark.parametrize("grid", ["legendre-gauss", "equiangular", "lobatto"])
@pytest.mark.parametrize("nlat", [2, 9, 64, 181] )
def test_quadrature_matches_oracle(grid, nlat):
    from makani_amd import legendre
    from oracle import sht as o
    if grid == "lobatto" and nlat < 3:
        pytest.skip("lobatto needs >= 3 nodes")
    a, b = legendre.colatitudes(nlat, grid), o.precompute_latitudes(nlat, grid)
    assert np.abs(a[0] - b[0]).max() < 1e-13 and np.abs(a[1] - b[1]).max() < 1e-13


@pytest.mark.parametrize("mmax,lmax,nlat,inverse,norm", [(17, 16, 33, False, "ortho"), (13, 12, 12, True, "ortho"),
                                                       (9, 20, 24, False, "four-pi"), (21, 20, 32, True, "schmidt")])
def test_legendre_matrix_matches_oracle(mmax, lmax, nlat, inverse, norm):
    from makani_amd import legendre
    from oracle import sht as o
    th, _ = legendre.colatitudes(nlat, "legendre-gauss")
    a = legendre.legendre_matrix(mmax, lmax, th, norm=norm, inverse=inverse)
    b = o.precompute_legpoly(mmax, lmax, th, norm=norm, inverse=inverse)
    assert np.abs(a - b).max() < 1e-13
    assert (a[np.triu_indices(min(mmax, lmax), 1)[1], np.triu_indices(min(mmax, lmax), 1)[0]] == 0).all()   # l < m


def test_fft_factorisation():
    from makani_amd import legendre
    for nlon in (4, 16, 24, 72, 128, 180, 256, 360, 480, 1440, 28, 62):
        r = legendre.factorize_half(nlon)
        assert int(np.prod(r)) == nlon // 2 and all(2 <= x <= 31 for x in r)
    with pytest.raises(NotImplementedError):
        legendre.factorize_half(25)
    with pytest.raises(NotImplementedError):
        legendre.factorize_half(2 * 37)
    tw = legendre.twiddle_table(24)
    assert tw.shape == (24, 2) and abs(tw[6, 1] + 1.0) < 1e-7 and abs(tw[6, 0]) < 1e-7


def test_transform_attributes_and_buffers():
    import makani_amd as ma
    S = ma.RealSHT(37, 72, lmax=12, mmax=13, grid="equiangular").float()
    I = ma.InverseRealSHT(12, 24, grid="legendre-gauss")
    assert (S.nlat, S.nlon, S.lmax, S.mmax, S.grid) == (37, 72, 12, 13, "equiangular")
    assert (I.lmax, I.mmax) == (12, 13)                       # torch-harmonics defaults
    assert S.weights.shape == (13, 12, 40) and S.weights.dtype == torch.float32
    assert (S.weights[:, :, 37:] == 0).all()
    assert S.weights_t.shape == (13, 37, 12) and torch.equal(S.weights_t, S.weights[:, :, :37].transpose(1, 2))
    assert I.pct.shape == (13, 12, 12) and I.pct_t.shape == (13, 12, 12)
    assert len(S.state_dict()) == 0 and len(I.state_dict()) == 0    # non-persistent buffers
    with pytest.raises(ValueError):
        ma.RealSHT(12, 24, mmax=14)


def test_spectral_conv_interface_and_errors():
    import makani_amd as ma
    f = ma.RealSHT(33, 64, lmax=12, mmax=13)
    i = ma.InverseRealSHT(12, 24, lmax=12, mmax=13, grid="legendre-gauss")
    c = ma.SpectralConv(f, i, 8, 6, operator_type="dhconv", gain=2.0)
    assert c.weight.shape == (1, 8, 6, 12) and c.weight.dtype == torch.complex64
    assert c.weight.is_shared_mp == ["matmul", "w"] and c.weight.sharded_dims_mp == [None, None, None, "h"]
    assert c.scale_residual and (c.modes_lat, c.modes_lon) == (12, 13)
    assert not hasattr(c, "bias")
    assert hasattr(ma.SpectralConv(f, i, 8, 6, bias=True), "bias")
    with pytest.raises(ValueError):
        ma.SpectralConv(f, i, 7, 6, num_groups=2)
    with pytest.raises(ValueError):
        ma.SpectralConv(f, i, 8, 6, operator_type="bogus")
    with pytest.raises(ValueError):
        ma.SpectralConv(ma.RealSHT(33, 64, lmax=10, mmax=13), i, 8, 6)
    # "diagonal": the reference initialises with a per-l scale broadcast against the last (m) axis
    # (spectral_convolution.py:184-193), which only works for lmax == mmax; mirrored, including the failure
    with pytest.raises(RuntimeError):
        ma.SpectralConv(f, i, 8, 6, operator_type="diagonal")
    f2 = ma.RealSHT(12, 24, lmax=12, mmax=12, grid="legendre-gauss")
    i2 = ma.InverseRealSHT(12, 24, lmax=12, mmax=12, grid="legendre-gauss")
    d = ma.SpectralConv(f2, i2, 8, 6, num_groups=2, operator_type="diagonal")
    assert d.weight.shape == (2, 4, 3, 12, 12)
    assert d.weight.is_shared_mp == ["matmul"] and d.weight.sharded_dims_mp == [None, None, None, "h", "w"]
    assert ma.SpectralConv(f2, i2, 8, 8, num_groups=2, operator_type="diagonal", separable=True).weight.shape == (2, 4, 12, 12)
    assert ma.SpectralConv(f2, i2, 8, 8, num_groups=2, operator_type="dhconv", separable=True).weight.shape == (2, 4, 12)
    assert ma.SpectralConv(f2, i2, 8, 16, num_groups=2).weight.shape == (2, 4, 8, 12)
    with pytest.raises(ValueError):
        ma.SpectralConv(f2, i2, 8, 6, separable=True)                    # separable operators keep the channel count
    with pytest.raises(NotImplementedError):
        ma.SpectralConv(f2, i2, 6, 6, num_groups=2)                      # grouped dhconv: group sizes % 4


@pytest.mark.parametrize("name", SFNO_GOLDEN)
def test_sfno_state_dict_is_reference_compatible(name):
    import makani_amd as ma
    g = load_golden(name)
    kwargs = json.loads(str(g["kwargs"]))
    model = ma.SphericalFourierNeuralOperatorNet(**kwargs, some_unknown_yaml_key=1)      # unknown kwargs are swallowed
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    own = model.state_dict()
    assert list(own.keys()) == list(sd.keys())
    for k in sd:
        assert own[k].shape == sd[k].shape and own[k].dtype == sd[k].dtype, k
    model.load_state_dict(sd, strict=True)
    for n, p in model.named_parameters():
        # annotations the reference sets (layers.py:613,633,780-784; mpu/layer_norm.py:121-122; sfnonet.py:727)
        tags = ("encoder", "decoder", "mlp", "norm", "residual_transform")
        if kwargs.get("normalization_layer") == "instance_norm_s2":          # models/common/layer_norm.py:30-160 sets none
            tags = tuple(t for t in tags if t != "norm")
        if kwargs.get("normalization_layer") == "layer_norm" and ".norm." in n:      # mpu/layer_norm.py:277-282: ["model"]
            assert p.is_shared_mp == ["model"], n
        elif any(t in n for t in tags):
            assert p.is_shared_mp == ["spatial"], n


def test_no_cpu_fallback():
    import makani_amd as ma
    model = ma.SphericalFourierNeuralOperatorNet(inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2, embed_dim=8,
                                                 num_layers=2, inp_chans=2, out_chans=2)
    with pytest.raises(RuntimeError, match="GPU"):
        model(torch.randn(1, 2, 16, 32))
    with pytest.raises(ValueError):
        ma.SphericalFourierNeuralOperatorNet(pos_embed="bogus", inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2)
    with pytest.raises(NotImplementedError):
        ma.SphericalFourierNeuralOperatorNet(normalization_layer="batch_norm", inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2)
    pe = ma.SphericalFourierNeuralOperatorNet(pos_embed="frequency", inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2, embed_dim=8)
    assert [tuple(p.shape) for p in pe.pos_embed] == [(1, 8, 8, 9), (1, 8, 8, 8)] and pe.pos_embed.type == "frequency"
    assert pe.no_weight_decay() == {"pos_embed", "cls_token"}
    with pytest.raises(NotImplementedError):
        ma.SphericalFourierNeuralOperatorNet(filter_type="non-linear", inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2)
    with pytest.raises(ValueError):
        ma.SphericalFourierNeuralOperatorNet(activation_function="tanh", inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2)


def test_product_never_imports_oracle():
    import glob
    for path in glob.glob(os.path.join(ROOT, "makani_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), path


def test_network_registers_with_the_reference_model_registry():
    """boundary B3 against the reference's own registry code (makani/models/model_registry.py:36-119): registration by
    class and by the "file.py:Class" string of the yaml `nettype` key; runs where the reference tree is present"""
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    mr = ref_shims.import_reference_module("makani.models.model_registry")
    import makani_amd as ma
    sys.path.insert(0, ROOT)
    import makani_plugin
    for n in ("SFNO_mi355x", "SFNO_mi355x_file"):
        mr._model_registry.pop(n, None)
    makani_plugin.register("SFNO_mi355x")
    assert "SFNO_mi355x" in mr.list_models() and mr._model_registry["SFNO_mi355x"] is ma.SphericalFourierNeuralOperatorNet
    with pytest.raises(ValueError):
        makani_plugin.register("SFNO_mi355x")                       # name already in use
    mr.register_model(os.path.join(ROOT, "makani_plugin.py") + ":SphericalFourierNeuralOperatorNet", "SFNO_mi355x_file")
    cls = mr._model_registry["SFNO_mi355x_file"]
    assert cls.__name__ == "SphericalFourierNeuralOperatorNet"
    # constructed the way get_model does (model_registry.py:201,228-235): shapes + channels + the yaml's keys, unknown ones ignored
    net = cls(inp_shape=(16, 32), out_shape=(16, 32), inp_chans=3, out_chans=3, scale_factor=2, embed_dim=8, num_layers=2,
              nettype="SFNO_mi355x_file", lr=1e-3, losses=[{"type": "l2"}])
    assert isinstance(net, ma.SphericalFourierNeuralOperatorNet) or type(net).__name__ == "SphericalFourierNeuralOperatorNet"
    for n in ("SFNO_mi355x", "SFNO_mi355x_file"):
        mr._model_registry.pop(n, None)


def test_fcn3_registers_with_the_reference_model_registry():
    """FourCastNet3 through the reference's registry: by class and by the yaml "file.py:Class" string, constructed the way
    get_model does (channel names arrive as yaml keys; inp_chans / out_chans and trainer keys are ignored)"""
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    mr = ref_shims.import_reference_module("makani.models.model_registry")
    import makani_amd as ma
    sys.path.insert(0, ROOT)
    import makani_plugin
    for n in ("FCN3_mi355x", "FCN3_mi355x_file"):
        mr._model_registry.pop(n, None)
    makani_plugin.register_fcn3("FCN3_mi355x")
    assert mr._model_registry["FCN3_mi355x"] is ma.AtmoSphericNeuralOperatorNet
    mr.register_model(os.path.join(ROOT, "makani_plugin.py") + ":AtmoSphericNeuralOperatorNet", "FCN3_mi355x_file")
    cls = mr._model_registry["FCN3_mi355x_file"]
    net = cls(inp_shape=(16, 32), out_shape=(16, 32), inp_chans=5, out_chans=4, scale_factor=2, filter_basis_type="morlet",
              channel_names=["u500", "v500", "u850", "v850"], aux_channel_names=["xzen"], atmo_embed_dim=4, surf_embed_dim=4,
              aux_embed_dim=2, num_layers=2, nettype="FCN3_mi355x_file", lr=1e-3, losses=[{"type": "l2"}])
    assert type(net).__name__ == "AtmoSphericNeuralOperatorNet" and net.n_out_chans == 4 and net.n_aux_chans == 1
    assert not hasattr(net, "surf_encoder")              # no surface variables in this channel list
    for n in ("FCN3_mi355x", "FCN3_mi355x_file"):
        mr._model_registry.pop(n, None)


def test_reference_get_model_builds_and_drives_the_network():
    """the reference's own get_model (model_registry.py:123-262) constructs the registered MI355X network from a
    yaml-shaped parameter set (every yaml key arrives as a constructor kwarg), wraps it in its MultiStepWrapper +
    Preprocessor2D, and the wrapped forward reaches the HIP path (which refuses CPU tensors: there is no fallback)"""
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    mr = ref_shims.import_reference_module("makani.models.model_registry")
    ParamsBase = ref_shims.import_reference_module("makani.utils.YParams").ParamsBase
    import makani_amd as ma
    sys.path.insert(0, ROOT)
    import makani_plugin
    mr._model_registry.pop("SFNO_mi355x", None)
    makani_plugin.register("SFNO_mi355x")
    p = ParamsBase()
    cfg = dict(nettype="SFNO_mi355x", model_grid_type="equiangular", sht_grid_type="legendre-gauss", filter_type="linear",
               scale_factor=2, embed_dim=8, num_layers=2, complex_activation="real", normalization_layer="instance_norm",
               hard_thresholding_fraction=1.0, use_mlp=True, mlp_mode="serial", mlp_ratio=2, separable=False,
               operator_type="dhconv", activation_function="gelu", pos_embed="none",
               losses=[{"type": "l2", "channel_weights": "auto"}], lr=1e-3, batch_size=2, weight_decay=0.0,
               img_shape_x=16, img_shape_y=32, img_shape_x_resampled=16, img_shape_y_resampled=32, N_in_channels=3,
               N_out_channels=3, n_history=0, n_future=2, history_normalization_mode="none", channel_names=["a", "b", "c"])
    for k, v in cfg.items():
        p[k] = v
    try:
        m = mr.get_model(p, multistep=True)
        assert type(m).__name__ == "MultiStepWrapper" and isinstance(m.model, ma.SphericalFourierNeuralOperatorNet)
        assert (m.model.inp_shape, m.model.out_shape, m.model.inp_chans, m.model.out_chans) == ((16, 32), (16, 32), 3, 3)
        assert m.model.embed_dim == 8 and len(m.model.blocks) == 2 and m.n_future == 2
        m.train()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(torch.rand(2, 3, 16, 32))
        s = mr.get_model(p, multistep=False)
        assert type(s).__name__ == "SingleStepWrapper" and isinstance(s.model, ma.SphericalFourierNeuralOperatorNet)
    finally:
        mr._model_registry.pop("SFNO_mi355x", None)


def test_reference_spectral_conv_accepts_the_hip_transforms():
    """boundary B1: the reference's own SpectralConv (spectral_convolution.py:116-211) built around makani_amd's
    transform objects reads the attributes it needs (nlat / nlon / lmax / mmax / grid) and ends up with the same
    configuration as makani_amd.SpectralConv around the same objects"""
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    sc = ref_shims.import_reference_module("makani.models.common.spectral_convolution")
    import makani_amd as ma
    f = ma.RealSHT(33, 64, lmax=12, mmax=13, grid="equiangular").float()
    i = ma.InverseRealSHT(12, 24, lmax=12, mmax=13, grid="legendre-gauss").float()
    torch.manual_seed(0)
    ref = sc.SpectralConv(f, i, 8, 6, operator_type="dhconv", bias=True, gain=2.0)
    torch.manual_seed(0)
    own = ma.SpectralConv(f, i, 8, 6, operator_type="dhconv", bias=True, gain=2.0)
    assert ref.weight.shape == own.weight.shape and torch.equal(ref.weight, own.weight)          # same init stream
    assert (ref.modes_lat, ref.modes_lon, ref.scale_residual) == (own.modes_lat, own.modes_lon, own.scale_residual) == (12, 13, True)
    assert (ref.modes_lat_local, ref.modes_lon_local, ref.nlat_local, ref.nlon_local) == (12, 13, 12, 24)
    assert ref.weight.is_shared_mp == own.weight.is_shared_mp and ref.weight.sharded_dims_mp == own.weight.sharded_dims_mp
    assert list(ref.state_dict().keys()) == [k for k in own.state_dict().keys() if not k.startswith(("forward_transform", "inverse_transform"))] \
        or list(ref.state_dict().keys()) == list(own.state_dict().keys())


def test_modules_opt_out_of_torch_compile():
    """makani compiles the model when jit_mode == "inductor" (utils/training/deterministic_trainer.py:316-319).  The HIP
    path is launched through a C ABI that dynamo cannot trace, so every public module forward is marked
    torch.compiler.disable (as the reference marks its own untraceable parts, sfnonet.py:765,840): a compiled wrapper
    simply runs the eager HIP path."""
    import makani_amd as ma
    net = ma.SphericalFourierNeuralOperatorNet(inp_shape=(16, 32), out_shape=(16, 32), scale_factor=2, embed_dim=8, num_layers=2,
                                               inp_chans=2, out_chans=2)
    mods = [net, net.blocks[0], net.blocks[0].filter.filter, net.trans, net.itrans, net.encoder, net.blocks[0].mlp, net.blocks[0].norm0,
            ma.GeometricLpLoss(img_shape=(16, 32), crop_shape=(16, 32), crop_offset=(0, 0), channel_names=["a", "b"], p=2.0)]
    for m in mods:
        assert getattr(m.forward, "_torchdynamo_disable", False), type(m).__name__
    compiled = torch.compile(net)
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # reaches the HIP entry point, not a dynamo error
        compiled(torch.randn(1, 2, 16, 32))


# --------------------------------------------------------------------------- #
# DISCO list builders (host side of csrc/disco_runs.hip)
# --------------------------------------------------------------------------- #
def _synthetic_entries(N):
    rng = np.random.default_rng(0)
    ent = []

    def add(t, k, row, lons):
        for l in lons:
            ent.append((t, k, row, l % N, rng.standard_normal()))
    add(0, 0, 0, range(-3, 4))                  # a run that crosses longitude 0
    add(0, 1, 0, range(-2, 5))
    add(0, 2, 0, range(-3, 3))
    add(0, 0, 1, range(N))                      # a whole latitude circle (polar row)
    add(0, 1, 1, range(0, N, 2))                # gaps: many one-tap runs
    add(1, 0, 2, [5, 6, 7])
    add(1, 2, 2, [6, 7, 8, 12])                 # two runs in one row
    add(2, 1, 0, [N - 1, 0])
    t, k, row, lon, val = map(np.array, zip(*ent))
    return t, k, row, lon, val.astype(np.float32)


def test_disco_run_lists_reproduce_the_convolution_tensor():
    """``_build_runs`` (per (segment, row): circular runs of consecutive longitudes, zero-padded to groups of R) and
    ``_build_fused`` (per (latitude, row): the union over the basis functions, aligned to 4 longitudes, K x 4 values per group)
    expand back to exactly the entries they were built from"""
    from makani_amd import disco
    N, K, R = 24, 3, 4
    t, k, row, lon, val = _synthetic_entries(N)
    dense = np.zeros((3, K, 3, N), np.float32)
    dense[t, k, row, lon] = val
    so, rn, vl = disco._build_runs(t * K + k, row, lon, val, 3 * K, N, R)
    assert vl.size % R == 0 and np.all(vl[-R:] == 0) and so[-1] == len(rn)
    rec = np.zeros_like(dense)
    for s_ in range(3 * K):
        for r in range(so[s_], so[s_ + 1]):
            rw, js, vo, ng = rn[r]
            assert vo % R == 0 and ng >= 1
            for g in range(ng * R):
                rec[s_ // K, s_ % K, rw, (js + g) % N] += vl[vo + g]
    assert np.array_equal(rec, dense)
    # the run that crosses longitude 0 is ONE run, a full circle is one run of N taps
    seg0 = rn[so[0]:so[1]]
    assert len(seg0) == 2 and sorted(int(x) for x in seg0[:, 3]) == [2, N // R]
    so, rn, vl = disco._build_fused(t, k, row, lon, val, 3, K, N)
    rec = np.zeros_like(dense)
    for tt in range(3):
        for r in range(so[tt], so[tt + 1]):
            rw, slot, vo, ng = rn[r]
            blk = vl[vo:vo + ng * K * 4].reshape(ng, K, 4)
            for g in range(ng):
                for kk in range(K):
                    for tau in range(4):
                        rec[tt, kk, rw, (slot * 4 + g * 4 + tau) % N] += blk[g, kk, tau]
    assert np.array_equal(rec, dense)
    assert np.all(vl[-K * 4:] == 0)


@pytest.mark.parametrize("basis,kshape", [("morlet", [3, 3]), ("morlet", [2, 4]), ("piecewise linear", [3, 4]), ("piecewise linear", [4, 3]),
                                          ("zernike", [3])])
def test_disco_lists_of_a_real_tensor_and_its_transpose(basis, kshape):
    """on a real convolution tensor: forward run lists, the adjoint's lists for latitude groups of 2 and 4 (image rows relative to
    the group's first touched output latitude) and the transposed tensor's lists all carry every entry exactly once — for every
    filter basis and for basis counts other than nine"""
    from makani_amd import disco
    shape = (12, 24)
    psi = disco.convolution_tensor(shape, shape, kshape, basis_type=basis, grid_in="equiangular", grid_out="equiangular",
                                   theta_cutoff=3 * math.pi / 11, basis_norm_mode="mean")
    assert psi["K"] == disco.basis_layout(basis, kshape)[1]
    L = disco._Lists(psi, shape, shape, "cpu")
    assert L.runs is not None and L.runs.R == 4 and disco.runs_radix(1440) == 4 and disco.runs_radix(1152) == 8 and disco.runs_radix(36 * 4 + 2) is None
    tot = float(np.abs(psi["v"]).sum())
    assert abs(float(L.runs.f_vals.abs().sum()) - tot) < 1e-4 * tot
    for LG in (2, 4):
        so, rn, vl, t_lo, t_n, mr = L.runs.bwd(LG)
        assert abs(float(vl.abs().sum()) - tot) < 1e-4 * tot and mr == int(t_n.max()) == L.runs.max_rows_b(LG)
        assert int(rn[:, 0].max()) < mr and so.numel() == shape[0] * psi["K"] + 1
    so, rn, vl, g_lo, g_n, mr = L.runs.fused(4)
    assert abs(float(vl.abs().sum()) - tot) < 1e-4 * tot and mr == L.runs.fused_rows(4)
    Lt = L.transposed()
    assert Lt.in_shape == shape and Lt.out_shape == shape and abs(float(Lt.runs.f_vals.abs().sum()) - tot) < 1e-4 * tot
    # transposing twice gives the forward lists back (same segments, same values)
    k, t, i, j, v = Lt.runs._e
    back = disco._RunLists(dict(k=k, t=i, i=t, j=(-j) % shape[1], v=v, K=psi["K"]), (shape, shape), "cpu")
    assert torch.equal(back.f_seg, L.runs.f_seg) and torch.equal(back.f_runs, L.runs.f_runs) and torch.equal(back.f_vals, L.runs.f_vals)


def _emulate_run_kernel(x, seg_off, runs, vals, seg, row0, nrows, R, lanes_per_wave=64):
    """numpy model of csrc/disco_runs.hip for ONE segment: de-interleaved row image (class = lon % R, slot = lon // R, one
    duplicate slot behind every class segment), a lane owns R consecutive output longitudes, every run streams blocks of R row
    elements through a sliding window (block j of the stream: class (bm + j) % R of slot base + (bm + j) // R)"""
    N = x.shape[1]
    n4 = N // R
    nw = (n4 + lanes_per_wave - 1) // lanes_per_wave
    segs = lanes_per_wave * nw + 1                                   # slots per class segment (compile-time in the kernel)
    img = np.zeros((nrows, R, segs))
    for r in range(nrows):
        for lon in range(N):
            img[r, lon % R, lon // R] = x[row0 + r, lon]
        img[r, :, n4] = img[r, :, 0]                                  # the duplicate slot
    out = np.zeros(N)
    for lane in range(n4):
        acc = np.zeros(R)
        for rr in range(seg_off[seg], seg_off[seg + 1]):
            row, js, voff, ng = (int(v) for v in runs[rr])
            bq, bm = js // R, js % R
            slot = (lane + bq) % n4

            def block(slot):
                return [img[row, (bm + j) % R, slot + (bm + j) // R] for j in range(R)]
            cur = block(slot)
            for g in range(ng):
                slot = (slot + 1) % n4
                nxt = block(slot)
                win = cur + nxt
                for tau in range(R):
                    for r in range(R):
                        acc[r] += vals[voff + g * R + tau] * win[tau + r]
                cur = nxt
        out[lane * R:(lane + 1) * R] = acc
    return out


def test_disco_run_kernel_addressing_model():
    """the addressing of the run-form kernel, restated in numpy on the device lists, reproduces the dense circular correlation
    y[t][p] = sum_{i, j} psi[k][t][i][j] x[i][(j + p) mod N] — forward lists and the adjoint's lists (latitude groups of 2)"""
    from makani_amd import disco
    shape = (10, 24)
    psi = disco.convolution_tensor(shape, shape, [3, 3], basis_type="morlet", grid_in="equiangular", grid_out="equiangular",
                                   theta_cutoff=3.5 * math.pi / 9, basis_norm_mode="mean")
    L = disco._Lists(psi, shape, shape, "cpu")
    RL, K, N = L.runs, psi["K"], shape[1]
    rng = np.random.default_rng(4)
    x = rng.standard_normal(shape)
    dense = np.zeros((K, shape[0], shape[0], N))
    dense[psi["k"], psi["t"], psi["i"], psi["j"]] = psi["v"]
    so, rn, vl = RL.f_seg.numpy(), RL.f_runs.numpy(), RL.f_vals.numpy()
    lat_lo, lat_n = RL.lat_lo.numpy(), RL.lat_n.numpy()
    for t, k in ((0, 0), (4, 3), (9, 8), (5, 1)):
        ref = np.array([sum(dense[k, t, i, j] * x[i, (j + p) % N] for i in range(shape[0]) for j in range(N) if dense[k, t, i, j] != 0)
                        for p in range(N)])
        got = _emulate_run_kernel(x, so, rn, vl, t * K + k, int(lat_lo[t]), int(lat_n[t]), RL.R)
        assert np.allclose(got, ref, atol=1e-6), (t, k)
    # adjoint: gx[i][q] = sum_{k, t, j} psi[k][t][i][j] gy[k][t][(q - j) mod N], latitude groups of 2 share the image of a k
    gy = rng.standard_normal((K, *shape))
    so, rn, vl, t_lo, t_n, _ = (a.numpy() if hasattr(a, "numpy") else a for a in RL.bwd(2))
    for i in (0, 3, 9):
        ref = np.array([sum(dense[k, t, i, j] * gy[k, t, (q - j) % N] for k in range(K) for t in range(shape[0]) for j in range(N)
                            if dense[k, t, i, j] != 0) for q in range(N)])
        got = sum(_emulate_run_kernel(gy[k], so, rn, vl, i * K + k, int(t_lo[(i // 2) * K + k]), int(t_n[(i // 2) * K + k]), RL.R)
                  for k in range(K))
        assert np.allclose(got, ref, atol=1e-6), i


def test_c_abi_host_side_planning_functions():
    """the planning entry points of the C ABI that need no GPU: which shapes the run-form DISCO kernels take (and with which
    radix / planes per workgroup / latitude group), the grouped-mix instantiations, the weight-gradient workspace"""
    import ctypes as C
    from makani_amd._lib import lib, MK_BF16, MK_F32
    L = lib()
    R, PB = C.c_int(0), C.c_int(0)
    # FourCastNet3's local block (720 longitudes, 10 image rows): R = 4, four planes of an fp32 image fit the LDS
    assert L.mk_disco_runs_shape(720, 10, 677, MK_BF16, 0, C.byref(R), C.byref(PB)) == 1 and (R.value, PB.value) == (4, 4)
    # 13 rows (four latitudes per workgroup) of four fp32 planes: 160 576 B, inside the 160 KB of a CU; 14 rows are not
    PB.value = 4
    assert L.mk_disco_runs_shape(720, 13, 677, MK_BF16, 0, C.byref(R), C.byref(PB)) == 1
    PB.value = 4
    assert L.mk_disco_runs_shape(720, 14, 677, MK_BF16, 0, C.byref(R), C.byref(PB)) == 0
    PB.value = 0
    assert L.mk_disco_runs_shape(720, 14, 677, MK_BF16, 0, C.byref(R), C.byref(PB)) == 1 and PB.value == 2
    PB.value = 4
    assert L.mk_disco_runs_shape(720, 14, 677, MK_BF16, 1, C.byref(R), C.byref(PB)) == 1          # bf16 image: half the bytes
    # the decoder grid: 1440 longitudes = 360 lanes of R = 4 in six waves; 1152 longitudes fall to R = 8; odd counts to the lists
    PB.value = 0
    assert L.mk_disco_runs_shape(1440, 6, 585, MK_F32, 0, C.byref(R), C.byref(PB)) == 1 and R.value == 4
    assert L.mk_disco_runs_shape(1152, 4, 8, MK_F32, 0, C.byref(R), C.byref(PB)) == 1 and R.value == 8
    assert L.mk_disco_runs_shape(722, 4, 8, MK_F32, 0, C.byref(R), C.byref(PB)) == 0
    assert L.mk_disco_runs_shape(720, 10, 1, MK_F32, 0, C.byref(R), C.byref(PB)) == 0                # a single plane: list kernels
    LG = C.c_int(0)
    assert L.mk_disco_fused_shape(720, 9, 13, 677, C.byref(LG), C.byref(PB)) == 1 and (LG.value, PB.value) == (4, 2)
    assert L.mk_disco_fused_shape(1440, 9, 6, 585, C.byref(LG), C.byref(PB)) == 1 and LG.value == 2
    assert L.mk_disco_fused_shape(720, 4, 13, 677, C.byref(LG), C.byref(PB)) == 0                    # K != 9: one stream per basis function
    assert [L.mk_group_mix_supported(c, r) for c, r in ((9, 9), (9, 8), (8, 9), (8, 8), (9, 10), (3, 9))] == [1, 1, 1, 1, 0, 0]
    assert L.mk_group_mix_blocks(1038240, MK_F32, 65) == 32 and L.mk_group_mix_blocks(64, MK_BF16, 4) == 1
    # weight gradient workspace = splits x (M x K partial products + M partial row sums for the fused bias gradient):
    # FourCastNet3's 677 x 6093 at 259 200 pixels is tiled 3 x 16 with 5 pixel splits
    L.mk_conv1x1_wgrad_workspace.restype = C.c_longlong
    assert L.mk_conv1x1_wgrad_workspace(677, 6093, 1, 259200) == 5 * 677 * (6093 + 1)
    assert L.mk_conv1x1_wgrad_workspace(384, 384, 1, 1038240) % (384 * (384 + 1)) == 0


def test_fused_schedule_plan_invariants():
    """makani_amd.dist_pipeline.Plan (host logic of the fused h x w exchange schedule): the plane blocks / sub-blocks partition
    the planes, every slab offset is a multiple of 4 floats (16-byte vectors), every rank of a configuration cuts the same
    number of latitude chunks, and a sender's slab for a destination has the size that destination expects from it"""
    import types
    from makani_amd import dist_pipeline as dp
    from makani_amd.distributed import compute_split_shapes as css

    def plans(h, w, nlat, nlon, L, M, P):
        out = {}
        for ih in range(h):
            for iw in range(w):
                T = types.SimpleNamespace(comm_size_polar=h, comm_size_azimuth=w, comm_rank_polar=ih, comm_rank_azimuth=iw, nlat=nlat,
                                          nlon=nlon, lmax=L, mmax=M, lat_shapes=css(nlat, h), lon_shapes=css(nlon, w),
                                          l_shapes=css(L, h), m_shapes=css(M, w))
                out[(ih, iw)] = dp.Plan(T, P)
        return out

    for (h, w, nlat, nlon, L, M, P) in [(4, 2, 721, 1440, 240, 241, 384), (4, 2, 721, 1440, 240, 241, 5), (2, 2, 240, 480, 240, 241, 73),
                                        (3, 1, 19, 48, 10, 11, 1), (1, 2, 33, 64, 16, 17, 6), (8, 1, 721, 1440, 240, 241, 96)]:
        ps = plans(h, w, nlat, nlon, L, M, P)
        p0 = ps[(0, 0)]
        assert sum(p0.pw) == P and len({p.nc for p in ps.values()}) == 1
        for j in range(w):
            assert sum(p0.valid[j]) == p0.pw[j] and all(s % 4 == 0 for s in p0.sub[j]) and sum(p0.sub[j]) >= p0.pw[j]
            assert all(0 <= p0.valid[j][i] <= p0.sub[j][i] for i in range(h))
        for (ih, iw), p in ps.items():
            assert all(p.base[j][i] % 4 == 0 for j in range(w) for i in range(h))
            assert p.f_total == sum(p.slab[j][i] for j in range(w) for i in range(h))
            c = p.chunks(p.hl)
            assert c[0] == 0 and c[-1] == p.hl and all(a <= b for a, b in zip(c[:-1], c[1:]))
            for i in range(h):
                for j in range(w):
                    # what (ih, iw) sends to (i, j) in step (3): its latitudes x M(j) x 2 x sub(iw, i); what (i, j) expects from
                    # (ih, iw): lat(ih) x M_loc(j) x 2 x sub(iw, i) rows of its block G[iw]
                    q = ps[(i, j)]
                    assert p.slab[j][i] == p.hl * q.Ml * 2 * q.sub[iw][i] == q.lat[ih] * q.Ml * 2 * p.sub[iw][i]


def test_share_gpu_sets_disjoint_compute_unit_ranges(monkeypatch):
    """ranks that share ONE GPU in a functional run get disjoint HSA_CU_MASK ranges (makani_amd/comm.py: share_gpu says why)"""
    import makani_amd.comm as mcomm
    monkeypatch.delenv("HSA_CU_MASK", raising=False)
    masks = [mcomm.share_gpu(r, 8) for r in range(8)]
    assert masks[0] == "0:0-31" and masks[7] == "0:224-255" and os.environ["HSA_CU_MASK"] == masks[7]
    spans = [tuple(int(v) for v in m.split(":")[1].split("-")) for m in masks]
    assert all(b - a == 31 for a, b in spans) and all(spans[i][1] < spans[i + 1][0] for i in range(7))
    assert mcomm.share_gpu(0, 1) == "0:0-255"
    monkeypatch.delenv("HSA_CU_MASK", raising=False)


def test_spectral_gemm_arithmetic_follows_torchs_tf32_switch(monkeypatch):
    """MAKANI_AMD_GEMM=auto (the default): torch.backends.cuda.matmul.allow_tf32 False (torch's default, the reference's test
    setting: tests/testutils.py disable_tf32) -> three-limb split "x6"; True (makani/train.py:87-88) -> two-limb split "x3";
    an explicit mode wins"""
    import torch
    from makani_amd import ops
    was = torch.backends.cuda.matmul.allow_tf32
    try:
        monkeypatch.setattr(ops, "GEMM_MODE", "auto")
        torch.backends.cuda.matmul.allow_tf32 = False
        assert ops.gemm_mode() == "x6"
        torch.backends.cuda.matmul.allow_tf32 = True
        assert ops.gemm_mode() == "x3"
        monkeypatch.setattr(ops, "GEMM_MODE", "fp32")
        assert ops.gemm_mode() == "fp32"
        monkeypatch.setattr(ops, "GEMM_MODE", "x6")
        assert ops.gemm_mode() == "x6"
    finally:
        torch.backends.cuda.matmul.allow_tf32 = was
