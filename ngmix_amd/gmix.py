"""
Gaussian-mixture model bookkeeping: model ids, parameter and gaussian counts
(reference: ngmix/gmix/gmix.py:1100-1193, 1245-1282).  The GMix host classes
live in ngmix_amd/gmix_classes.py.
"""

GMIX_FULL = 0
GMIX_GAUSS = 1
GMIX_TURB = 2
GMIX_EXP = 3
GMIX_DEV = 4
GMIX_BDC = 5
GMIX_BDF = 6
GMIX_COELLIP = 7
GMIX_CM = 9
GMIX_BD = 10

_NAMES = {
    GMIX_FULL: "full", GMIX_GAUSS: "gauss", GMIX_TURB: "turb", GMIX_EXP: "exp",
    GMIX_DEV: "dev", GMIX_BDC: "bdc", GMIX_BDF: "bdf", GMIX_COELLIP: "coellip",
    GMIX_CM: "cm", GMIX_BD: "bd",
}
_NUMS = {name: num for num, name in _NAMES.items()}

_NPARS = {
    GMIX_GAUSS: 6, GMIX_TURB: 6, GMIX_EXP: 6, GMIX_DEV: 6, GMIX_CM: 6,
    GMIX_BD: 8, GMIX_BDF: 7, GMIX_BDC: 8,
}

_NGAUSS = {
    GMIX_GAUSS: 1, "gauss": 1, GMIX_TURB: 3, "turb": 3, GMIX_EXP: 6, "exp": 6,
    GMIX_DEV: 10, "dev": 10, GMIX_CM: 16, GMIX_BD: 16, GMIX_BDF: 16,
    GMIX_BDC: 16,
    "em1": 1, "em2": 2, "em3": 3, "em4": 4, "em5": 5,
    "coellip1": 1, "coellip2": 2, "coellip3": 3, "coellip4": 4, "coellip5": 5,
}


def get_model_num(model):
    """numerical id for a model given by name or id"""
    if model in _NUMS:
        return _NUMS[model]
    if model in _NAMES:
        return model
    raise ValueError("unknown model: '%s'" % model)


def get_model_name(model):
    """string name for a model given by name or id"""
    if model in _NAMES:
        return _NAMES[model]
    if model in _NUMS:
        return model
    raise ValueError("unknown model: '%s'" % model)


def get_model_ngauss(model):
    """number of gaussians of a model"""
    if model not in _NGAUSS:
        raise ValueError("unknown model: '%s'" % model)
    return _NGAUSS[model]


def get_model_npars(model):
    """number of parameters of a model"""
    if model not in _NUMS and model not in _NAMES:
        raise ValueError("bad model: '%s'" % model)
    return _NPARS[get_model_num(model)]


def get_coellip_npars(ngauss):
    return 4 + 2 * ngauss


def get_coellip_ngauss(npars):
    return (npars - 4) // 2
