"""Host-side mirror of the reference's src/inversion package for the hot path.

Same class names, constructor arguments and method meaning as
src/inversion/{model_params,model,inversion,data}.py of jamiebyer/mcmc; the
forward model, likelihood and Metropolis-Hastings step run in the HIP kernels
behind include/mcmc_amd.h.
"""
from .errors import DispersionError  # noqa: F401
from .model_params import ModelParams, DispersionCurveParams, EllipticityParams  # noqa: F401
from .model import Model  # noqa: F401
from .data import Data, SyntheticData  # noqa: F401
from .inversion import Inversion  # noqa: F401
