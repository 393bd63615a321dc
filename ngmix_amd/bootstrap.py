"""
psf-then-object bootstrap (reference API: ngmix/bootstrap.py:67-154): fit the
psfs, drop observations whose psf fit failed, then fit the object.
"""
import logging

from .gexceptions import BootPSFFailure
from .observation import Observation, ObsList, MultiBandObsList

__all__ = ["Bootstrapper", "bootstrap", "remove_failed_psf_obs"]

BOOT_S2N_LOW = 2 ** 0
BOOT_R2_LOW = 2 ** 1
BOOT_R4_LOW = 2 ** 2
BOOT_TS2N_ROUND_FAIL = 2 ** 3
BOOT_ROUND_CONVOLVE_FAIL = 2 ** 4
BOOT_WEIGHTS_LOW = 2 ** 5

logger = logging.getLogger(__name__)


def _psf_ok(obs):
    return obs.psf.meta["result"]["flags"] == 0


def remove_failed_psf_obs(obs):
    """same container type with only the observations whose psf fit has
    flags == 0; BootPSFFailure when a band (or the single obs) has none left"""
    if isinstance(obs, MultiBandObsList):
        new = MultiBandObsList(meta=obs.meta)
        for obslist in obs:
            new.append(remove_failed_psf_obs(obslist))
        return new
    if isinstance(obs, ObsList):
        new = ObsList(meta=obs.meta)
        for o in obs:
            if _psf_ok(o):
                new.append(o)
        if len(new) == 0:
            raise BootPSFFailure("no good psf fits")
        return new
    if isinstance(obs, Observation):
        if not _psf_ok(obs):
            raise BootPSFFailure("no good psf fits")
        return obs
    raise ValueError('got obs input type: "%s", should be Observation, ObsList, '
                     "or MulitiBandObsList" % type(obs))


def bootstrap(obs, runner, psf_runner=None, ignore_failed_psf=True):
    if psf_runner is not None:
        psf_runner.go(obs=obs)
        if ignore_failed_psf:
            obs = remove_failed_psf_obs(obs=obs)
    return runner.go(obs=obs)


class Bootstrapper(object):
    def __init__(self, runner, psf_runner=None, ignore_failed_psf=True):
        self.runner = runner
        self.psf_runner = psf_runner
        self.ignore_failed_psf = ignore_failed_psf

    def go(self, obs):
        return bootstrap(obs=obs, runner=self.runner, psf_runner=self.psf_runner,
                         ignore_failed_psf=self.ignore_failed_psf)

    @property
    def fitter(self):
        return self.runner.fitter
