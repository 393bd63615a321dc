"""
Retry / recursion orchestration around fitters (reference API:
ngmix/runners.py).  Pure host control flow, duck-typed on
fitter.go(obs=, guess=) returning a mapping with 'flags' (and optionally
get_gmix()), so it drives the HIP-backed fitters of this package unchanged.
"""
from .observation import Observation, ObsList, MultiBandObsList

__all__ = ["Runner", "PSFRunner", "run_fitter", "run_psf_fitter", "run_fitter_many"]


def run_fitter(obs, fitter, guesser=None, ntry=1):
    """call the fitter up to ntry times, with a fresh guess each time, until
    the result has flags == 0; the last result is returned either way"""
    for _ in range(ntry):
        if guesser is None:
            res = fitter.go(obs=obs)
        else:
            res = fitter.go(obs=obs, guess=guesser(obs=obs))
        if res["flags"] == 0:
            break
    return res


def run_fitter_many(obs, fitter, guesser, ntry=1):
    """
    run_fitter over a catalogue as batches (fitter.go_many): every object is
    fitted from guesser(obs=...) at once; the objects whose result has
    flags != 0 are fitted again, as a batch of their own, from a fresh guess, up
    to ntry attempts in all -- what the reference's loop over run_fitter does
    object by object (runners.py:116-150).

    obs: a sequence of Observation / ObsList / MultiBandObsList
    guesser: called as guesser(obs=o) per object (the reference's interface),
        or an object with guess_many(obs=sequence) -> (n, npars)

    Returns a list of per-object result dicts (each object's LAST attempt;
    'ntry' holds the attempts it took).
    """
    import numpy as np

    def guesses(objs):
        if hasattr(guesser, "guess_many"):
            return np.asarray(guesser.guess_many(obs=objs), dtype="f8")
        return np.array([guesser(obs=o) for o in objs], dtype="f8")

    obs = list(obs)
    out = [None] * len(obs)
    todo = list(range(len(obs)))
    for attempt in range(1, int(ntry) + 1):
        if not todo:
            break
        sub = [obs[i] for i in todo]
        res = fitter.go_many(sub, guesses(sub))
        flags = np.asarray(res.arrays["flags"])
        for k, i in enumerate(todo):
            r = res[k]
            r["ntry"] = attempt
            out[i] = r
        todo = [i for k, i in enumerate(todo) if flags[k] != 0]
    return out


def run_psf_fitter(obs, fitter, guesser=None, ntry=1, set_result=True):
    """
    Fit every observation's psf (or the observation itself when it has none),
    recursing through ObsList / MultiBandObsList and returning results in the
    same nesting.  With set_result the result goes to meta['result'] of the
    fitted observation and, on success, its gmix is set from res.get_gmix().
    """
    if isinstance(obs, (MultiBandObsList, ObsList)):
        return [run_psf_fitter(obs=sub, fitter=fitter, guesser=guesser, ntry=ntry,
                               set_result=set_result) for sub in obs]
    if not isinstance(obs, Observation):
        raise ValueError("obs must be an Observation, ObsList, or MultiBandObsList")
    target = obs.psf if obs.has_psf() else obs
    res = run_fitter(obs=target, fitter=fitter, guesser=guesser, ntry=ntry)
    if set_result:
        target.meta["result"] = res
        if res["flags"] == 0 and hasattr(res, "get_gmix"):
            target.gmix = res.get_gmix()
    return res


class RunnerBase(object):
    def __init__(self, fitter, guesser=None, ntry=1):
        self.fitter = fitter
        self.guesser = guesser
        self.ntry = ntry


class Runner(RunnerBase):
    """fitter + guesser + retries on the observation(s) themselves"""

    def go(self, obs):
        return run_fitter(obs=obs, fitter=self.fitter, guesser=self.guesser,
                          ntry=self.ntry)


class PSFRunner(RunnerBase):
    """fitter + guesser + retries on each observation's psf"""

    def __init__(self, fitter, guesser=None, ntry=1, set_result=True):
        super().__init__(fitter, guesser=guesser, ntry=ntry)
        self.set_result = set_result

    def go(self, obs):
        return run_psf_fitter(obs=obs, fitter=self.fitter, guesser=self.guesser,
                              ntry=self.ntry, set_result=self.set_result)
