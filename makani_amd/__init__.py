"""makani_amd — MI355X-native SFNO / spherical-harmonic-transform hot path behind
makani's nn.Module plug-in API.  HIP kernels live in ``csrc/`` behind the C ABI of
``include/makani_amd.h``; this package is the host-side mirror of the reference interface."""
from .sht import RealSHT, InverseRealSHT
from .spectral_conv import SpectralConv
from .layers import MLP, EncoderDecoder, InstanceNorm2d, PointwiseConv, GeometricInstanceNormS2
from .sfno import SphericalFourierNeuralOperatorNet, NeuralOperatorBlock, SpectralFilterLayer
from .losses import CRPSLoss, GeometricLpLoss, GridQuadrature, SpectralCRPSLoss, SpectralLpLoss, SpectralH1Loss
from .stepper import MultiStepWrapper, SingleStepWrapper
from .disco import DiscreteContinuousConvS2, ResampleS2
from .fcn3 import AtmoSphericNeuralOperatorNet

__all__ = ["RealSHT", "InverseRealSHT", "SpectralConv", "MLP", "EncoderDecoder", "InstanceNorm2d", "PointwiseConv",
           "SphericalFourierNeuralOperatorNet", "NeuralOperatorBlock", "SpectralFilterLayer", "GeometricLpLoss",
           "GridQuadrature", "SpectralLpLoss", "SpectralH1Loss", "CRPSLoss", "SpectralCRPSLoss", "GeometricInstanceNormS2", "MultiStepWrapper", "SingleStepWrapper",
           "DiscreteContinuousConvS2", "ResampleS2", "AtmoSphericNeuralOperatorNet"]
