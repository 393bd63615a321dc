"""
A handful of fixed seeds of the fuzz campaign (tools/fuzz_vs_oracle.py, whose
logs are under profiles/) inside the suite: the noisy regimes the fixed-shape
oracle comparisons do not visit -- adaptive moments and EM at every
signal-to-noise (the flag paths, runs into maxiter, collapsing gaussians,
images with negative pixels), the per-object seam entry points on random
shapes, the weighted sums and deriv_images on ragged batches.  The seeds
include the ones on which the ORACLE's own result moves under a 1e-13 change
of its image (what the comparison holds there is in the tool's docstrings and
DESIGN section 4).
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.fixture(scope="module")
def fz():
    import fuzz_vs_oracle
    return fuzz_vs_oracle


@pytest.mark.parametrize("seed", [974759107, 104976270, 235073785, 777402747, 962125702,
                                  1058342680, 774472567, 5, 6, 7, 8])
def test_em_noisy_seed(fz, seed):
    nraise, nscatter = fz.em_noisy(seed)
    assert 0 <= nraise <= 8 and 0 <= nscatter <= 8


@pytest.mark.parametrize("seed", [463286626, 854052464, 11, 12, 13, 14])
def test_admom_noisy_seed(fz, seed):
    fz.admom_noisy(seed)


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_seam_forms_seed(fz, seed):
    fz.seam_forms(seed)


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_wsums_and_derivs_seed(fz, seed):
    fz.wsums_and_derivs(seed)


# ---- the forward-difference fits against MINPACK's lmdif
# (tools/fuzz_lm_fd_vs_minpack.py; profiles/r06_fuzz_lm_fd_vs_minpack*.log)
@pytest.fixture(scope="module")
def fzfd():
    import fuzz_lm_fd_vs_minpack
    return fuzz_lm_fd_vs_minpack


@pytest.mark.parametrize("cls,seed", [("coellip1", 41), ("coellip1", 42), ("coellip2", 43),
                                      ("coellip2", 44), ("bdf", 45), ("bdf", 46), ("bd", 47),
                                      ("bd", 48)])
def test_lmdif_seed_vs_minpack(fzfd, cls, seed):
    """co-elliptical fits with 1-2 gaussians, 'bdf', 'bd': flags and ier equal,
    nfev within one jacobian of MINPACK's (equal in 99.8 % of 11k fuzzed fits),
    the parameters of the fits both routes converge within 0.05 sigma"""
    import numpy as np
    _, n, res, ones = fzfd.one_case(seed, [cls])
    for o, one in enumerate(ones):
        assert (one["flags"] == 0) == (res["flags"][o] == 0), (o, one["flags"], res["flags"][o])
        assert one["ier"] == res["ier"][o]
        assert abs(int(one["nfev"]) - int(res["nfev"][o])) <= n + 1
        if one["flags"] == 0:
            assert np.all(np.abs(res["pars"][o] - one["pars"]) <= 0.05 * one["pars_err"])


@pytest.mark.parametrize("cls,seed", [("coellip3", 51), ("coellip3", 52), ("coellip4", 53),
                                      ("coellip4", 54), ("coellip5", 57), ("coellip5", 56)])
def test_lmdif_ill_conditioned_seed_vs_minpack(fzfd, cls, seed):
    """co-elliptical fits with 3-5 gaussians (cond(J) up to 1e8): the iteration
    ends for the same reason (ier), the covariance exists wherever MINPACK's does
    (LM_SINGULAR_MATRIX agrees: the double-double factor of lm_precise.hip) --
    what may differ is the sign of the smallest eigenvalue of a cond-1e16
    covariance (LM_NEG_COV_EIG, raised at the same rate by both routes) and, in
    the degenerate valleys, the number of evaluations"""
    from ngmix_amd import flags
    _, n, res, ones = fzfd.one_case(seed, [cls])
    for o, one in enumerate(ones):
        assert one["ier"] == res["ier"][o]
        sing = lambda f: (int(f) & flags.LM_SINGULAR_MATRIX) != 0  # noqa: E731
        assert sing(one["flags"]) == sing(res["flags"][o]), (o, one["flags"], res["flags"][o])


# ---- the batched bootstrap against the per-object Bootstrapper over MINPACK
# (tools/fuzz_boot.py; profiles/r06_fuzz_boot.log)
@pytest.mark.parametrize("seed", [71, 72, 73, 74])
def test_bootstrap_batch_seed_vs_per_object_bootstrapper(seed):
    """lmder object models, the lmder psf fitter: psf pass / fail, attempts and
    nfev of every psf fit, the epochs dropped, BootPSFFailure, flags, attempts,
    nfev and parameters (1e-4 sigma) of every object -- all equal"""
    import fuzz_boot
    stats = fuzz_boot.new_stats()
    case = fuzz_boot.one_case(seed, models=("exp", "gauss", "dev"), psf_kinds=("gauss",))
    fuzz_boot.compare(case, stats, seed)
    assert stats["objects"] >= 3 and not stats["odd"], stats["odd"]


# ---- fits with a joint prior (its rows from the prior kernel inside the device
# loop) against MINPACK with the host prior
# (tools/fuzz_lm_prior_vs_minpack.py; profiles/r06_fuzz_lm_prior_vs_minpack.log)
@pytest.mark.parametrize("cls,seed", [("gauss", 81), ("exp", 945978995), ("dev", 83),
                                      ("turb", 935445890), ("bdf", 778668515), ("bdf", 86),
                                      ("bd", 818168476), ("bd", 88)])
def test_prior_fit_seed_vs_minpack(cls, seed):
    """a random host joint prior (erf / normal / bounded normal / log-normal /
    truncated gaussian / flat terms) on a random object: flags and ier equal,
    nfev equal for the lmder models (within two jacobians for lmdif: equal in
    73,598 of 73,609 fuzzed fits), parameters within 1e-7 sigma for lmder and
    0.1 sigma for lmdif (the campaign's worst: 0.06), ln p to match"""
    import numpy as np
    import fuzz_lm_prior_vs_minpack as fz
    _, n, res, ones = fz.one_case(seed, [cls])
    lmder = cls in ("gauss", "exp", "dev")
    for o, one in enumerate(ones):
        assert (one["flags"] == 0) == (res["flags"][o] == 0)
        assert one["ier"] == res["ier"][o]
        assert abs(int(one["nfev"]) - int(res["nfev"][o])) <= (0 if lmder else 2 * (n + 1))
        if one["flags"] == 0:
            tol = 1e-7 if lmder else 0.1
            assert np.all(np.abs(res["pars"][o] - one["pars"]) <= tol * one["pars_err"])
            assert abs(res["lnprob"][o] - one["lnprob"]) <= (1e-8 if lmder else 0.5)
