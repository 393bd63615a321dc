"""sentinel values and default LM parameters (reference: ngmix/defaults.py:10-17)"""
import numpy as np

PDEF = -9.999e9   # parameter default
CDEF = 9.999e9    # covariance / error default
LOWVAL = -np.inf
BIGVAL = 9999.0e47

DEFAULT_LM_PARS = {"maxfev": 4000, "ftol": 1.0e-5, "xtol": 1.0e-5}
