"""
ngmix_amd -- MI355X-native implementation of the ngmix pixel hot path
(gmix_nb render / loglike / fdiff, admom, em, fastexp) behind the reference's
GMix / Observation / Jacobian / Fitter API.

All pixel arithmetic runs in hand-written HIP kernels (ngmix_amd/csrc) reached
through the C ABI declared in include/ngmix_hip.h; there is no CPU fallback.
"""
from . import flags  # noqa: F401
from . import defaults  # noqa: F401
from . import gexceptions  # noqa: F401
from .gexceptions import (  # noqa: F401
    GMixRangeError, GMixFatalError, GMixMaxIterEM, PSFFluxFailure,
    BootPSFFailure, BootGalFailure, FFTRangeError,
)
from . import _lib  # noqa: F401
from . import gmix  # noqa: F401

__version__ = "0.1.0"
