"""Import shims that let the reference's OWN, UNMODIFIED model code
(``/root/reference/makani/models/networks/sfnonet.py`` and friends) run on
PyTorch-CPU in the build container.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Only ``oracle/make_golden.py``
uses this, and only where ``/root/reference`` exists (never on the GPU box).
The shims provide the two un-vendored dependencies:

* ``torch_harmonics``  -> backed by the restatements in ``oracle/sht.py`` (transforms) and ``oracle/disco.py``
  (DISCO convolution, ResampleS2)
* ``physicsnemo``      -> inert ``ModelMetaData`` / ``Module.from_torch``
  (no arithmetic lives there)

plus inert stubs for optional I/O packages pulled in by ``makani/__init__.py``.
"""

import importlib
import importlib.machinery
import os
import sys
import types
from dataclasses import dataclass

import torch

from . import sht as _sht
from . import disco as _disco

REFERENCE_ROOT = os.environ.get("MAKANI_REFERENCE_ROOT", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    sys.modules[name] = m
    return m


class _Inert:
    """Object that swallows any attribute access / call (for wandb, h5py ...)."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()


def _inert_getattr(attr):
    if attr.startswith("__") and attr.endswith("__"):
        raise AttributeError(attr)
    return _Inert()


def _as_torch(fn):
    def wrapped(*a, **k):
        out = fn(*a, **k)
        return tuple(torch.from_numpy(o.copy()) for o in out)

    return wrapped


def install():
    """Install the shims (idempotent) and put the reference on sys.path."""
    if "torch_harmonics" in sys.modules and getattr(sys.modules["torch_harmonics"], "_is_oracle_shim", False):
        return

    # ---- torch_harmonics ---------------------------------------------------
    def _not_available(*a, **k):
        raise NotImplementedError("not provided by the oracle shim")

    state = {"init": False}

    def _thd_init(polar_group, azimuth_group):
        state["init"] = True

    quad = _mod(
        "torch_harmonics.quadrature",
        legendre_gauss_weights=_as_torch(_sht.legendre_gauss_weights),
        clenshaw_curtiss_weights=_as_torch(_sht.clenshaw_curtiss_weights),
        lobatto_weights=_as_torch(_sht.lobatto_weights),
        precompute_latitudes=_as_torch(_sht.precompute_latitudes),
        _precompute_latitudes=_as_torch(_sht.precompute_latitudes),
    )
    prim = _mod(
        "torch_harmonics.distributed.primitives",
        _gather=_not_available,
        _split=_not_available,
        _reduce=_not_available,
        _transpose=_not_available,
    )

    class _NoDist(torch.nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError("distributed transforms are not part of the serial oracle shim")

    thd = _mod(
        "torch_harmonics.distributed",
        init=_thd_init,
        is_initialized=lambda: state["init"],
        compute_split_shapes=_sht.compute_split_shapes,
        split_tensor_along_dim=_sht.split_tensor_along_dim,
        distributed_transpose_azimuth=_not_available,
        distributed_transpose_polar=_not_available,
        DistributedRealSHT=type("DistributedRealSHT", (_NoDist,), {}),
        DistributedInverseRealSHT=type("DistributedInverseRealSHT", (_NoDist,), {}),
        DistributedResampleS2=type("DistributedResampleS2", (_NoDist,), {}),
        DistributedDiscreteContinuousConvS2=type("DistributedDiscreteContinuousConvS2", (_NoDist,), {}),
        DistributedDiscreteContinuousConvTransposeS2=type("DistributedDiscreteContinuousConvTransposeS2", (_NoDist,), {}),
        primitives=prim,
    )
    th = _mod(
        "torch_harmonics",
        RealSHT=_sht.RealSHT,
        InverseRealSHT=_sht.InverseRealSHT,
        ResampleS2=_disco.ResampleS2,
        DiscreteContinuousConvS2=_disco.DiscreteContinuousConvS2,
        DiscreteContinuousConvTransposeS2=type("DiscreteContinuousConvTransposeS2", (_NoDist,), {}),
        quadrature=quad,
        distributed=thd,
        __version__="0.9.0+oracle",
        _is_oracle_shim=True,
    )
    th.__path__ = []
    thd.__path__ = []

    # ---- physicsnemo -------------------------------------------------------
    @dataclass
    class ModelMetaData:
        name: str = "unnamed"
        jit: bool = False
        cuda_graphs: bool = False
        amp_cpu: bool = False
        amp_gpu: bool = False

    class Module(torch.nn.Module):
        @classmethod
        def from_torch(cls, torch_model_class, meta=None, name=None, register=False):
            return torch_model_class

    class DistributedManager:
        _inst = None

        def __new__(cls):
            raise RuntimeError("DistributedManager shim: serial oracle only")

        @classmethod
        def initialize(cls):
            raise RuntimeError("DistributedManager shim: serial oracle only")

    class ProcessGroupNode:
        def __init__(self, name, size=None):
            self.name, self.size = name, size

    class ProcessGroupConfig:
        def __init__(self, root):
            self.root = root

        def add_node(self, node, parent=None):
            pass

        def set_leaf_group_sizes(self, sizes, update_parent_sizes=True):
            pass

    pn = _mod("physicsnemo", ModelMetaData=ModelMetaData, Module=Module, __version__="1.3.0+shim")
    pn.__path__ = []
    dist = _mod("physicsnemo.distributed")
    dist.__path__ = []
    _mod("physicsnemo.distributed.manager", DistributedManager=DistributedManager)
    _mod("physicsnemo.distributed.config", ProcessGroupNode=ProcessGroupNode, ProcessGroupConfig=ProcessGroupConfig)
    reg = _mod("physicsnemo.registry", ModelRegistry=_Inert)
    pn.registry = reg
    pn.distributed = dist

    # ---- inert optional packages ------------------------------------------
    for name in [
        "wandb",
        "h5py",
        "zarr",
        "more_itertools",
        "moviepy",
        "moviepy.video",
        "moviepy.video.io",
        "moviepy.video.io.ImageSequenceClip",
        "ruamel",
        "ruamel.yaml",
    ]:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            m = _mod(name)
            m.__path__ = []
            m.__getattr__ = _inert_getattr  # type: ignore[attr-defined]

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "makani"))


def import_reference_sfno():
    """Return the reference's ``SphericalFourierNeuralOperatorNet`` class."""
    install()
    mod = importlib.import_module("makani.models.networks.sfnonet")
    return mod.SphericalFourierNeuralOperatorNet


def import_reference_module(name: str):
    install()
    return importlib.import_module(name)
