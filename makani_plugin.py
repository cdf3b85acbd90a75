"""Single-file entry point for makani's model registry.

makani resolves ``nettype: "path/to/file.py:Class"`` by executing that FILE as a stand-alone module
(``makani/models/model_registry.py:69-94``), so the file must not rely on package-relative imports.  This one puts the
repository on ``sys.path`` and re-exports the MI355X network:

    # config/sfnonet.yaml
    nettype: "/path/to/repo/makani_plugin.py:SphericalFourierNeuralOperatorNet"

or, from Python, ``makani_plugin.register("SFNO_mi355x")`` (``model_registry.register_model``, ``:97-119``).
FourCastNet3 (``nettype: "FCN3"`` in config/fourcastnet3.yaml) binds the same way:

    nettype: "/path/to/repo/makani_plugin.py:AtmoSphericNeuralOperatorNet"       # or  makani_plugin.register_fcn3("FCN3_mi355x")
"""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from makani_amd import (AtmoSphericNeuralOperatorNet, MultiStepWrapper, SingleStepWrapper,  # noqa: E402,F401
                        SphericalFourierNeuralOperatorNet)


def register(name: str = "SFNO_mi355x") -> None:
    """register the network under ``name`` in makani's registry (requires makani to be importable)"""
    from makani.models import model_registry
    model_registry.register_model(SphericalFourierNeuralOperatorNet, name)


def register_fcn3(name: str = "FCN3_mi355x") -> None:
    """register FourCastNet3 (``AtmoSphericNeuralOperatorNet``) under ``name`` in makani's registry"""
    from makani.models import model_registry
    model_registry.register_model(AtmoSphericNeuralOperatorNet, name)
