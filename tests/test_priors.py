"""
ngmix_amd.priors / ngmix_amd.joint_prior against the reference:
tests/golden/priors.npz holds what the REFERENCE's priors returned for the
script of calls in tests/helpers/prior_cases.py (oracle/gen_golden_priors.py
ran it); the same script on ngmix_amd's classes must give the same numbers --
densities and residuals to the bit, the same exception classes, the same
seeded draws and the same generator state afterwards.
"""
import os

import numpy as np
import pytest

import ngmix_amd as ngmix
from helpers import prior_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "priors.npz")


@pytest.fixture(scope="module")
def both():
    want = dict(np.load(GOLD, allow_pickle=False))
    got = prior_cases.run(ngmix.priors, ngmix.joint_prior, guessers=ngmix.guessers)
    return want, got


def test_same_entries(both):
    want, got = both
    assert set(want) == set(got), sorted(set(want) ^ set(got))[:20]
    assert len(want) > 900


def test_every_entry_is_the_references(both):
    want, got = both
    bad = []
    for k in sorted(want):
        w, g = want[k], got[k]
        if w.dtype.kind == "U" or g.dtype.kind == "U":
            if str(w) != str(g):
                bad.append((k, str(w), str(g)))
        elif w.shape != g.shape or not np.array_equal(w, g, equal_nan=True):
            bad.append((k, w, g))
    assert not bad, (len(bad), bad[:8])


def test_fitmodel_rows_are_the_joint_priors():
    """the seam FitModel uses: fill_fdiff at the head of the residual vector,
    bounds for leastsqbound, GMixRangeError out of the shape prior"""
    rng = np.random.RandomState(5)
    p = ngmix.joint_prior.PriorSimpleSep(
        ngmix.priors.CenPrior(0.0, 0.0, 0.1, 0.1, rng=rng), ngmix.priors.GPriorBA(0.3, rng=rng),
        ngmix.priors.Normal(0.5, 0.2, rng=rng, bounds=(0.01, 5.0)),
        [ngmix.priors.FlatPrior(-10.0, 1e6, rng=rng)] * 2)
    assert p.nband == 2 and len(p.bounds) == 7 and p.bounds[4] == (0.01, 5.0)
    assert p.bounds[5] == (None, None)
    fdiff = np.zeros(20)
    pars = np.array([0.1, -0.1, 0.3, 0.0, 0.7, 10.0, 20.0])
    assert p.fill_fdiff(pars, fdiff) == 6
    np.testing.assert_allclose(fdiff[:2], [1.0, 1.0], rtol=1e-14)
    assert fdiff[3] == pytest.approx(1.0) and fdiff[4] == 0.0 and fdiff[5] == 0.0
    assert np.isclose(-0.5 * (fdiff[:6] ** 2).sum(), p.get_lnprob_scalar(pars))
    with pytest.raises(ngmix.GMixRangeError):
        p.fill_fdiff(np.array([0, 0, 0.8, 0.8, 0.7, 1.0, 1.0]), fdiff)
    s = p.sample(100)
    assert s.shape == (100, 7) and np.all(np.hypot(s[:, 2], s[:, 3]) < 1.0)
