import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The per-object Fitter of the tests is MINPACK on the host calling the seam
# kernels once per evaluation -- the INDEPENDENT check of the batched
# lock-step driver (and of the reference goldens' exact nfev).  Outside the
# tests Fitter.go defaults to the one-object batch of that driver; the tests of
# that route ask for it explicitly (batched=True).
# (setdefault: `NGMIX_FITTER_BATCHED=1 pytest -m gpu` runs the whole suite through
# the shipped default route instead -- green too: profiles/r06_gputest_batched_route.log)
os.environ.setdefault("NGMIX_FITTER_BATCHED", "0")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)"
    )


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
