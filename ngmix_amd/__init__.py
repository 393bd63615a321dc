"""
ngmix_amd -- MI355X-native implementation of the ngmix pixel hot path
(gmix_nb render / loglike / fdiff, admom, em, fastexp) behind the reference's
GMix / Observation / Jacobian / Fitter API.

All pixel arithmetic runs in hand-written HIP kernels (ngmix_amd/csrc) reached
through the C ABI declared in include/ngmix_hip.h; there is no CPU fallback.

    single objects:  GMix, GMixModel, Observation, Jacobian, Fitter,
                     run_admom, run_em  (the reference's API)
    batches:         ngmix_amd.batch.StampBatch / GMixBatch (N stamps resident
                     in HBM, one kernel launch per operation)
"""
from . import flags  # noqa: F401
from . import defaults  # noqa: F401
from . import gexceptions  # noqa: F401
from .gexceptions import (  # noqa: F401
    NGmixBaseException, GMixRangeError, GMixFatalError, GMixMaxIterEM, PSFFluxFailure,
    BootPSFFailure, BootGalFailure, FFTRangeError,
)
from . import _lib  # noqa: F401
from . import util  # noqa: F401
from .util import print_pars, srandu  # noqa: F401
from . import shape  # noqa: F401
from .shape import Shape  # noqa: F401
from . import moments  # noqa: F401
from . import jacobian  # noqa: F401
from .jacobian import Jacobian, DiagonalJacobian, UnitJacobian  # noqa: F401
from . import pixels  # noqa: F401
from . import gmix  # noqa: F401
from .gmix import (  # noqa: F401
    GMix, GMixModel, GMixCM, GMixCoellip, GMixList, MultiBandGMixList, make_gmix_model,
    gmix_concat,
    set_exact_kernels, get_exact_kernels,
)
from . import observation  # noqa: F401
from .observation import (  # noqa: F401
    Observation, ObsList, MultiBandObsList, get_mb_obs,
)
from . import admom  # noqa: F401
from . import em  # noqa: F401
from . import fitting  # noqa: F401
from . import gmix_ndim  # noqa: F401
from .gmix_ndim import GMixND  # noqa: F401
from . import priors  # noqa: F401
from . import joint_prior  # noqa: F401
from . import guessers  # noqa: F401
from . import runners  # noqa: F401
from . import bootstrap  # noqa: F401
from . import gaussmom  # noqa: F401
from .gaussmom import GaussMom, GaussMomBatch  # noqa: F401
from . import psfflux  # noqa: F401
from .psfflux import PSFFluxFitter, PSFFluxBatch  # noqa: F401
from . import batch  # noqa: F401
from . import prior_batch  # noqa: F401
from . import lm_batch  # noqa: F401
from .lm_batch import LMBatchFitter  # noqa: F401
from . import fastexp_nb  # noqa: F401
from . import gaussap  # noqa: F401
from . import prepsfmom  # noqa: F401
from . import ksigmamom  # noqa: F401
from . import simobs  # noqa: F401
from . import pipeline  # noqa: F401
from .pipeline import bootstrap_batch, bootstrap_many  # noqa: F401

__version__ = "0.1.0"
