"""No-op stand-in for numba, used ONLY in the build container to import the
pure-Python reference (/root/reference/ngmix) as an interpreted parity oracle
(SURVEY.md section 8c).  njit/jit return the function unchanged; vectorize maps
to numpy.vectorize.  Test infrastructure; never imported by the product."""
import numpy as _np


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(func):
        return func
    return wrap


njit = _identity_decorator
jit = _identity_decorator


def vectorize(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _np.vectorize(args[0])

    def wrap(func):
        return _np.vectorize(func)
    return wrap
