"""small numeric helpers used by result post-processing
(reference: ngmix/util.py:57-82)"""
import numpy as np


def get_ratio_var(a, b, var_a, var_b, cov_ab):
    """variance of a/b to first order"""
    if np.any(b == 0):
        raise ValueError("zero in denominator")
    ratio_sq = (a / b) ** 2
    return ratio_sq * (var_a / a ** 2 + var_b / b ** 2 - 2 * cov_ab / (a * b))


def get_ratio_error(a, b, var_a, var_b, cov_ab):
    """standard error of a/b; negative variances clip to zero"""
    var = np.clip(get_ratio_var(a, b, var_a, var_b, cov_ab), 0.0, np.inf)
    return np.sqrt(var)


def srandu(num=None, rng=None):
    """uniform deviates in [-1, 1]"""
    randu = np.random.uniform if rng is None else rng.uniform
    return randu(low=-1.0, high=1.0, size=num)


def format_pars(pars, fmt="%8.3g"):
    """the parameters on one line: each number in `fmt`, two blanks after
    each but the last, one after that (util.py:38-54)"""
    return "  ".join(fmt % p for p in pars) + " " if len(pars) else ""


def print_pars(pars, fmt="%8.3g", front=None, stream=None, logger=None):
    """print (or log) a parameter vector on one line"""
    import sys
    # (pars None prints as "None"; a logger gets a debug record -- util.py:5-35)
    line = "%s" % None if pars is None else format_pars(pars, fmt=fmt)
    if front is not None:
        line = "%s %s" % (front, line)
    if logger is not None:
        logger.debug(line)
    else:
        (stream if stream is not None else sys.stdout).write(line + "\n")
