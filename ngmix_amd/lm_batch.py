"""
Batched Levenberg-Marquardt fitting: N independent objects advance in lock
step, two kernel launches per LM step for all of them.

Reference (per object): Fitter.go -> run_leastsq -> scipy leastsq / MINPACK
lmder, calling back into FitModel.calc_fdiff / calc_jacobian once per
evaluation (ngmix/fitting/fitters.py:64-112, leastsqbound.py:33-155,
results.py:439-570).  Here the same iteration (csrc/lm_core.hpp: lmder's
decision logic, tested against MINPACK in tests/test_lm_core.py) runs on the
device for every object at once:

    ngmix_lm_eval_batch     residuals + analytic jacobian at each object's
                            trial point, reduced on chip to J^T J, J^T f, |f|^2
    ngmix_lm_advance_batch  one lmder step per object

    ngmix_lm_finalize_batch run_leastsq's packaging per object (cov_x from
                            R / ipvt, chi2/dof scaling, flags)

and the results are packaged exactly as run_leastsq / FitModel.set_fit_result
package one fit: flags, nfev, ier, pars, pars_err, pars_cov0, pars_cov, and for
flags == 0 lnprob, s2n_numer, s2n_denom, npix, chi2per, dof, s2n_w, s2n, g,
g_cov, g_err, T, T_err, flux, flux_err (flux_cov when nband > 1).

Scope: gauss / exp / dev with the analytic jacobian (lmder, as the reference's
Fitter: results.py SIMPLE_ANALYTIC_MODELS) and gauss / turb / exp / dev / bdf /
bd with MINPACK's forward differences evaluated inside the pixel pass (lmdif,
what the reference runs for the models without analytic derivatives).  A
batch prior (ngmix_amd/prior_batch.py) adds the reference's prior rows to every
object's residual vector and its bounds run leastsqbound's parameter transform
inside the iteration; prior=None is legal as in the reference (zero prior rows,
no bounds, results.py:354-357).
"""
import ctypes
import os

import numpy as np

from . import _lib
from .batch import GMixBatch, _dptr, _stream, _torch, _on_device
from .defaults import PDEF, CDEF, DEFAULT_LM_PARS
from .gmix import get_model_num
from .fitting import get_lm_n_prior_pars, STEP_PRIOR
from .prior_batch import as_batch_prior, bounds_arrays, prior_normal_sums

__all__ = ["LMBatchFitter"]


SIMPLE_ANALYTIC_MODELS = ("gauss", "exp", "dev")
# local (per band) parameter count of the models the device kernels fill
MODEL_NLOC = {"gauss": 6, "turb": 6, "exp": 6, "dev": 6, "bdf": 7, "bd": 8}


class LMBatchResult(dict):
    """the dict of per-object arrays LMBatchFitter.go returns (run_leastsq's
    and set_fit_result's keys); entries registered with set_lazy stay on the
    device and are downloaded when first read"""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self._lazy = {}

    def set_lazy(self, key, fetch):
        """fetch(result) -> value, called once when `key` is first read (it is
        handed the result instead of closing over it: a closure would make a
        reference cycle, and the pinned download buffers behind the arrays
        would wait for the cyclic collector instead of going back to the
        allocator when the result is dropped)"""
        self._lazy[key] = fetch

    def __missing__(self, key):
        if key in self._lazy:
            value = self._lazy.pop(key)(self)
            self[key] = value
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        if self._lazy.pop(key, None) is None:
            dict.__delitem__(self, key)

    # every whole-mapping view counts the lazy keys and reads through them, so
    # that items() / dict(res) / {**res} / copy / pickle see one ordinary dict
    def materialize(self):
        for key in list(self._lazy):
            self[key]
        return self

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def pop(self, key, *default):
        if key in self:
            value = self[key]
            dict.__delitem__(self, key)
            return value
        if default:
            return default[0]
        raise KeyError(key)

    def copy(self):
        return LMBatchResult(self.items())

    __copy__ = copy

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self.materialize())

    def __reduce__(self):
        # the fetchers are closures over device tensors: pickle the values
        return (LMBatchResult, (dict(self.items()),))


class _Job(object):
    """one batch on its way through LMBatchFitter (device arrays, events)"""


class _HipEvents(object):
    """n hipEvent_t of the library (ngmix_events_create), destroyed with the
    object"""

    def __init__(self, n):
        self.n = n
        self.handles = (ctypes.c_void_p * n)()
        _lib.check(_lib.lib().ngmix_events_create(n, self.handles), "ngmix_events_create")

    def record(self, i, stream=None):
        _lib.check(_lib.lib().ngmix_event_record(self.handles[i], stream or _stream()),
                   "ngmix_event_record")

    def elapsed_ms(self, i, j):
        ms = ctypes.c_float()
        _lib.check(_lib.lib().ngmix_event_elapsed_ms(self.handles[i], self.handles[j],
                                                      ctypes.byref(ms)),
                   "ngmix_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            _lib.lib().ngmix_events_destroy(self.n, self.handles)
        except Exception:
            pass


class _EventPool(object):
    """timing events handed out as ctypes arrays and taken back (bench.py's
    time_kernels mode records ~25 events per fit: no create / destroy per fit)"""

    def __init__(self):
        self.free = []
        self.owned = []

    def take(self, n):
        arr = (ctypes.c_void_p * n)()
        need = n - min(n, len(self.free))
        if need:
            fresh = _HipEvents(need)
            self.owned.append(fresh)
            self.free.extend(fresh.handles[i] for i in range(need))
        for i in range(n):
            arr[i] = self.free.pop()
        return arr

    def give(self, arr):
        if arr is not None:
            self.free.extend(arr[i] for i in range(len(arr)))


# A small batch (the per-object interface's one-object fits above all) is host
# bound: ~0.4 ms of Python per fit against 0.26 ms of kernels, and every
# torch.empty / copy_ / event is 2-8 us of it.  Its device buffers are therefore
# views of ONE allocation, its outputs (packed block | covariance triangle |
# finalize records) contiguous at the end of it so that they come back in ONE
# download, and the zeros of the statistics accumulator ride in the upload.
SMALL_BATCH = 64
# the most lock-step rounds queued blind from the last batch's count (a round
# that finds every fit finished costs two empty launches, ~20 us; co-elliptical
# psf fits run 60-70 rounds, their tails hundreds: a cap below the rounds a
# workload needs makes every call take the miss path)
ROUNDS_HINT_CAP = 256


def _carve(torch, dev, pieces):
    """one uint8 allocation cut into named views: pieces = [(name, shape,
    torch dtype)], each 16-byte aligned; returns (buffer, views, offsets)"""
    offs, at = {}, 0
    for name, shape, dt in pieces:
        nbytes = int(np.prod(shape)) * dt.itemsize
        offs[name] = (at, nbytes)
        at += (nbytes + 15) // 16 * 16
    buf = torch.empty(at, dtype=torch.uint8, device=dev)
    views = {}
    for name, shape, dt in pieces:
        a, nb = offs[name]
        views[name] = buf[a:a + nb].view(dt).reshape(shape)
    return buf, views, offs


class LMBatchFitter(object):
    """
    fitter = LMBatchFitter(model='exp')
    res = fitter.go(stamps, guess, psf=psf_gmixes)

    model: 'gauss' | 'exp' | 'dev' (analytic jacobian, MINPACK lmder, as the
        reference's Fitter) or 'turb' | 'bdf' | 'bd' (forward differences,
        MINPACK lmdif, as the reference runs the models without analytic
        derivatives), or 'coellip' with ngauss = 1..3 (CoellipFitter, lmdif)
    fit_pars: dict with maxfev / ftol / xtol as for Fitter (defaults
        DEFAULT_LM_PARS, ngmix/defaults.py:17)
    analytic_jacobian: False forces forward differences for gauss/exp/dev
        too (Fitter's switch of the same name)
    prior: a joint prior as the reference's callers build it
        (joint_prior.PriorSimpleSep ... of priors.py terms: put in its batch
        form by prior_batch.as_batch_prior), or a batch prior
        (prior_batch.PriorSimpleSepBatch, PriorBatchAdapter, or anything with
        their three members)
    """

    def __init__(self, model, fit_pars=None, analytic_jacobian=True, prior=None,
                 device_prior=True, ngauss=None):
        # host joint priors (joint_prior.py) are put in their batch form
        self.prior = as_batch_prior(prior)
        # evaluate a PriorSimpleSepBatch inside one kernel (False: torch ops)
        self.device_prior = device_prior
        self.ngauss = None
        if model == "coellip":
            # CoellipFitter (fitters.py:120-141): ngauss co-elliptical gaussians,
            # parameters cen1, cen2, g1, g2, T_1.., F_1.., one band, lmdif
            if ngauss is None or not 1 <= int(ngauss) <= (_lib.LM_NPMAX - 4) // 2:
                raise ValueError("coellip needs ngauss = 1..%d" % ((_lib.LM_NPMAX - 4) // 2))
            self.ngauss = int(ngauss)
            self.model = model
            self.nloc = 4 + 2 * self.ngauss
        elif model not in MODEL_NLOC:
            raise ValueError("LMBatchFitter supports %s and 'coellip'" % (tuple(MODEL_NLOC),))
        else:
            self.model = model
            self.nloc = MODEL_NLOC[model]
        self.fd = not (analytic_jacobian and model in SIMPLE_ANALYTIC_MODELS)
        self.fit_pars = dict(DEFAULT_LM_PARS if fit_pars is None else fit_pars)

    def go(self, stamps, guess, psf=None, stamp_obj=None, stamp_band=None,
           check_every=1):
        """
        one batch, start to finish

        stamps: StampBatch -- every observation (epoch / band) of every object
        guess: (nobj, nshape + nband) starting parameters: the model's shape
            parameters ([cen1, cen2, g1, g2, T] for gauss/turb/exp/dev, plus
            fracdev for bdf, plus logTratio and fracdev for bd) followed by
            one flux per band
        psf: GMixBatch with one mixture per stamp, or None
        stamp_obj: (nstamps,) object index of each stamp, non-decreasing;
            None: stamp i is object i
        stamp_band: (nstamps,) band of each stamp; None: band 0
        check_every: (host-driven loop only) read the count of running fits
            every so many rounds

        returns a dict of arrays indexed by object
        """
        job = self._enqueue(stamps, guess, psf, stamp_obj, stamp_band, check_every, False)
        return self._collect(job)

    # host time of the last batch, by what the host was doing (milliseconds):
    # "enqueue" (set-up and queueing, _enqueue), "wait" (blocked on the batch's
    # downloads: the GPU is the bottleneck while this is > 0), "package"
    # (_collect after the wait) -- a pipeline keeps up as long as enqueue +
    # package stays below the GPU's time per batch
    host_ms = None

    def go_stream(self, batches, check_every=1):
        """
        Fit a SEQUENCE of batches as a software pipeline: a generator of result
        dicts, one per batch, in order.  batches: an iterable of (stamps,
        guess, kwargs) with kwargs the keyword arguments of go() (psf=...,
        stamp_obj=..., stamp_band=...).

        Everything a batch needs on the device -- set-up, the lock-step rounds,
        finalize, pack, the downloads -- is queued without the host waiting for
        anything (_enqueue), and batch i + 1 is queued before the host turns to
        batch i's arrays: the GPU always has a whole batch of work in front of
        it, whatever the host's pace.  That matters twice: the host phases of a
        call disappear behind kernels, and the GPU stays in the clock state it
        only reaches under uninterrupted load (the same lm_eval launch takes
        1.30 ms then, 1.42-1.45 ms when the GPU idles a millisecond between
        launches: tools/lm_eval_warm.py).  Every result is what go() returns
        for that batch, bit for bit.
        """
        prev = None
        for stamps, guess, kw in batches:
            job = self._enqueue(stamps, guess, kw.get("psf"), kw.get("stamp_obj"),
                                kw.get("stamp_band"), check_every, True)
            if prev is not None:
                yield self._collect(prev)
            prev = job
        if prev is not None:
            yield self._collect(prev)

    # ------------------------------------------------------------------
    # the two halves of a fit: everything the device needs, queued; then the
    # host's half
    # ------------------------------------------------------------------

    def _mark(self, job, name):
        # fitter.time_phases = True: wall-clock per phase of this call, each
        # phase closed by a device synchronisation (a diagnostic: it removes
        # the overlap between phases); "nosync": host time only
        import time
        if job.phases is None:
            return
        if self.time_phases != "nosync":
            _torch().cuda.synchronize(job.dev)
        now = time.perf_counter()
        job.phases[name] = job.phases.get(name, 0.0) + (now - job.tmark) * 1e3
        job.tmark = now

    def _enqueue(self, stamps, guess, psf, stamp_obj, stamp_band, check_every, streaming):
        """set a batch up and queue its whole fit on the current stream; returns
        the job _collect() turns into the result dict"""
        import time
        torch = _torch()
        L = _lib.lib()
        dev = stamps.device
        job = _Job()
        job.dev = dev
        # (the device context and the stream handle are looked up once per fit:
        # each torch.cuda.device(...) / current_stream() costs ~10 us of host
        # time, a tenth of a one-object fit when done per launch)
        t0 = time.perf_counter()
        with _on_device(dev):
            job.stream = _stream()
            self._enqueue_on(job, stamps, guess, psf, stamp_obj, stamp_band,
                             check_every, streaming)
        job.host_enqueue_ms = (time.perf_counter() - t0) * 1e3
        return job

    def _enqueue_on(self, job, stamps, guess, psf, stamp_obj, stamp_band, check_every,
                    streaming):
        import time
        torch = _torch()
        L = _lib.lib()
        dev = job.dev
        job.phases = {} if getattr(self, "time_phases", False) else None
        job.tmark = time.perf_counter()
        self.phase_ms = job.phases
        guess = np.ascontiguousarray(np.atleast_2d(guess), dtype="f8")
        nobj, npars = guess.shape
        nshape = self.nloc - 1
        nband = npars - nshape
        if nband < 1 or npars > _lib.LM_NPMAX:
            raise ValueError("guess must have %d + nband (1..%d) columns"
                             % (nshape, _lib.LM_NPMAX - nshape))
        ns = stamps.n
        if stamp_obj is None:
            if ns != nobj:
                raise ValueError("stamp_obj is needed when objects have several stamps")
            sobj = np.arange(ns, dtype=np.int32)
        else:
            sobj = np.ascontiguousarray(stamp_obj, dtype=np.int32)
            if sobj.shape != (ns,):
                raise ValueError("stamp_obj must be (nstamps,) and non-decreasing")
            if nobj == 1:
                # (a one-object fit: the per-object interface's batches)
                if sobj.any():
                    raise ValueError("stamp_obj out of range")
            else:
                if np.any(np.diff(sobj) < 0):
                    raise ValueError("stamp_obj must be (nstamps,) and non-decreasing")
                if sobj.min() < 0 or sobj.max() >= nobj:
                    raise ValueError("stamp_obj out of range")
        if stamp_band is None:
            sband = np.zeros(ns, dtype=np.int32)
        else:
            sband = np.ascontiguousarray(stamp_band, dtype=np.int32)
            if sband.shape != (ns,) or sband.min() < 0 or sband.max() >= nband:
                raise ValueError("stamp_band out of range")
        trivial_map = stamp_obj is None and stamp_band is None   # stamp i = object i
        if trivial_map:
            obj_start = np.arange(nobj + 1, dtype=np.int64)
        elif nobj == 1:
            obj_start = np.array([0, ns], dtype=np.int64)
        else:
            obj_start = np.searchsorted(sobj, np.arange(nobj + 1)).astype(np.int64)
        if ns < nobj or (nobj > 1 and np.any(np.diff(obj_start) == 0)) or ns == 0:
            raise ValueError("every object needs at least one stamp")
        npsf = 0
        psf_host = None
        if psf is not None and isinstance(psf, np.ndarray):
            # host gauss2d records (nstamps, npsf): uploaded with the batch's
            # other small arrays (Fitter.go's one-object batches)
            psf_host = np.ascontiguousarray(psf, dtype=_lib.GAUSS2D_DTYPE).reshape(ns, -1)
            npsf = psf_host.shape[1]
        elif psf is not None:
            assert psf.n == ns, "one psf mixture per stamp"
            npsf = psf.ngauss

        fp = self.fit_pars
        lo = hi = None
        if self.prior is not None and getattr(self.prior, "bounds", None) is not None:
            lo, hi = bounds_arrays(self.prior.bounds, npars)
        # leastsq's convention: maxfev = 0 (or absent) means 100 (n + 1) function
        # calls with an analytic jacobian, 200 (n + 1) in forward-difference mode
        maxfev = int(fp.get("maxfev", 0)) or (200 if self.fd else 100) * (npars + 1)
        ev_init = self._timing_events(2)
        nsplit = self._nsplit_wanted()
        nsum = self.nloc * (self.nloc + 1) // 2 + self.nloc + 1
        # the loglike statistics of set_fit_result ride with the analytic
        # kernel's sums (lnprob = -fnorm^2 / 2 has no prior term to add)
        loop_stats = not self.fd and self.prior is None and \
            not getattr(self, "stats_pass", False) and \
            not os.environ.get("NGMIX_LM_JBASIS")
        # (see SMALL_BATCH: every device buffer of the fit out of one allocation)
        arena = None
        if nobj <= SMALL_BATCH and ns <= 16 * SMALL_BATCH and nsplit <= 1 and \
                not os.environ.get("NGMIX_LM_NO_ARENA"):
            ntri = npars * (npars + 1) // 2
            abuf, arena, arena_off = _carve(torch, dev, [
                ("states", (nobj, _lib.LM_STATE_DTYPE.itemsize), torch.uint8),
                ("sums", (ns, nsum), torch.float64),
                ("status", (ns,), torch.int32),
                ("sstats", (ns, 2), torch.float64),
                # the three outputs, contiguous: one download
                ("flat", (nobj * (2 * npars + _lib.LM_NCOLS),), torch.float64),
                ("tri", (nobj, ntri), torch.float64),
                ("rec", (nobj, 4 + 2 * npars + 2 * npars * npars), torch.float64)])
            arena["out"] = abuf[arena_off["flat"][0]:
                                arena_off["rec"][0] + arena_off["rec"][1]]
            arena["out_off"] = {k: arena_off[k][0] - arena_off["flat"][0]
                                for k in ("flat", "tri", "rec")}
        d_states = arena["states"] if arena is not None else torch.empty(
            (nobj, _lib.LM_STATE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        if trivial_map:
            npix_obj = stamps.npix_kept.astype(np.int64)
        elif nobj == 1:
            npix_obj = np.array([stamps.npix_kept.sum()], dtype=np.int64)
        else:
            npix_obj = np.add.reduceat(stamps.npix_kept.astype(np.int64), obj_start[:-1])
        # ONE upload for the guess, the stamp -> object / band maps, the object
        # offsets and the pixel counts (a host-to-device copy costs ~30 us
        # whatever its size; a one-object fit made five): 8-byte aligned pieces
        # of one buffer, viewed in their own types on the device
        need_maps = not trivial_map
        ns_pad = (ns + 1) // 2 * 2
        pieces = [("guess", guess.view(np.uint8).reshape(-1)),
                  ("npix", npix_obj.view(np.uint8).reshape(-1))]
        if arena is not None and loop_stats:
            pieces.append(("ostats", np.zeros(16 * nobj, dtype=np.uint8)))
        if psf_host is not None:
            pieces.append(("psf", psf_host.view(np.uint8).reshape(-1)))
        if need_maps:
            pad32 = lambda a: np.concatenate([a, np.zeros(ns_pad - ns, dtype=np.int32)])
            pieces += [("start", obj_start.view(np.uint8).reshape(-1)),
                       ("sobj", pad32(sobj).view(np.uint8).reshape(-1)),
                       ("sband", pad32(sband).view(np.uint8).reshape(-1))]
        # (through pinned memory, asynchronously: a copy from pageable memory
        # makes torch synchronise the stream -- the host would wait there for
        # everything queued before, i.e. for the previous batch of a pipeline,
        # and the GPU would idle while the rest of this batch is being queued)
        nbytes = sum(a.size for _, a in pieces)
        h_all = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        np.concatenate([a for _, a in pieces], out=h_all.numpy())
        d_all = h_all.to(dev, non_blocking=True)
        view, at = {}, 0
        for name, a in pieces:
            view[name] = d_all[at:at + a.size]
            at += a.size
        d_guess = view["guess"].view(torch.float64).reshape(nobj, npars)
        d_npix = view["npix"].view(torch.int64)
        if psf_host is not None:
            psf = GMixBatch(view["psf"].view(torch.float64).reshape(-1, 13), ns, npsf)
        d_sobj = d_sband = d_start = None
        if need_maps:
            d_start = view["start"].view(torch.int64)
            d_sobj = view["sobj"].view(torch.int32)[:ns]
            d_sband = view["sband"].view(torch.int32)[:ns]
        elif nsplit > 1:
            # stamp i is object i in band 0: the kernels take NULL for the three
            # maps (as pieces on several streams they need the absolute indices)
            d_sobj = torch.arange(ns, dtype=torch.int32, device=dev)
            d_sband = torch.zeros(ns, dtype=torch.int32, device=dev)
            d_start = torch.arange(nobj + 1, dtype=torch.int64, device=dev)
        if True:
            self._record(ev_init, 0, job.stream)
            _lib.check(L.ngmix_lm_init_batch(
                _dptr(d_states), nobj, npars, _dptr(d_guess),
                float(fp.get("ftol", 1.49012e-8)), float(fp.get("xtol", 1.49012e-8)),
                float(fp.get("gtol", 0.0)), maxfev, float(fp.get("factor", 100.0)),
                self._lm_mode(),
                _lib.ptr(lo) if lo is not None else None,
                _lib.ptr(hi) if hi is not None else None, job.stream),
                "ngmix_lm_init_batch")
            self._record(ev_init, 1, job.stream)
        d_sstats = d_ostats = None
        if arena is not None:
            d_sums, d_status = arena["sums"], arena["status"]
            if loop_stats:
                d_sstats = arena["sstats"]
                d_ostats = view["ostats"].view(torch.float64).reshape(nobj, 2)
        else:
            # (every row is written by the first round's launch)
            d_sums = torch.empty((ns, nsum), dtype=torch.float64, device=dev)
            d_status = torch.empty(ns, dtype=torch.int32, device=dev)
            if loop_stats:
                d_sstats = torch.empty((ns, 2), dtype=torch.float64, device=dev)
                d_ostats = torch.zeros((nobj, 2), dtype=torch.float64, device=dev)
        modnum = get_model_num(self.model)
        if self.ngauss is not None:
            modnum += 256 * self.ngauss   # the count rides with the model id
            if nband != 1:
                raise ValueError("coellip fits one band: guess needs %d columns" % self.nloc)
        d_osums = None
        prior_desc = None
        if self.prior is not None and self.device_prior and hasattr(self.prior, "descriptor"):
            prior_desc = self.prior.descriptor()
            if prior_desc is not None and nshape != 5 + int(prior_desc["nmid"][0]):
                raise ValueError("the prior is laid out for %d parameters before the fluxes, "
                                 "model '%s' has %d" % (5 + int(prior_desc["nmid"][0]),
                                                        self.model, nshape))
            if prior_desc is not None and int(prior_desc["nband"][0]) != nband:
                raise ValueError("the prior has %d flux terms, the guess %d bands"
                                 % (int(prior_desc["nband"][0]), nband))
        if prior_desc is not None:
            d_osums = torch.zeros((nobj, npars * (npars + 1) // 2 + npars + 1),
                                  dtype=torch.float64, device=dev)
        self.prior_path = ("kernel" if prior_desc is not None else
                           "torch" if self.prior is not None else None)
        # Forward-difference fits of ten and more local parameters (co-elliptical
        # psf fits with 3+ gaussians: cond(J) ~ 1e8 at the solution) take their
        # covariance from a double-double factorisation of the last jacobian's
        # normal equations (ngmix_lm_precise_cov_batch): the pixel passes leave
        # the point of every jacobian in d_jacpt.  fitter.precise_cov = False
        # (NGMIX_LM_NO_PRECISE_COV): the factor the iteration left, as before
        d_jacpt = None
        if self.fd and self.nloc >= _lib.LM_PRECISE_MIN_NLOC and \
                getattr(self, "precise_cov", True) and \
                not os.environ.get("NGMIX_LM_NO_PRECISE_COV") and self.prior is None:
            d_jacpt = torch.zeros((nobj, 3, _lib.LM_NPMAX), dtype=torch.float64, device=dev)
        if self.prior is not None and prior_desc is None:
            nsplit = 1
        nsplit = max(1, min(int(nsplit), nobj))

        job.__dict__.update(
            stamps=stamps, psf=psf, nobj=nobj, npars=npars, nband=nband, ns=ns, npsf=npsf,
            sobj=sobj, sband=sband, obj_start=obj_start, trivial_map=trivial_map,
            maxfev=maxfev, d_states=d_states, d_guess=d_guess, d_sobj=d_sobj,
            d_sband=d_sband, d_start=d_start, d_sums=d_sums, d_status=d_status,
            d_sstats=d_sstats, d_ostats=d_ostats, d_osums=d_osums, modnum=modnum,
            prior_desc=prior_desc, loop_stats=loop_stats, nsplit=nsplit, nsum=nsum,
            d_jacpt=d_jacpt,
            streaming=streaming, check_every=check_every, ev_init=ev_init,
            npix_obj=npix_obj, d_npix=d_npix, d_all=d_all, h_all=h_all, arena=arena,
            batch=stamps._batch(1), chunks=[], useful_rounds=None, ev_post=None,
            legacy_ev=None, loop_ms_host=0.0)
        # the host-free loop serves one piece with no prior or the kernel prior;
        # a torch-evaluated prior sits between the two launches of a round and
        # pieces on several streams are interleaved by the host: the host-driven
        # loop (NGMIX_LM_HOST_LOOP selects it for A/B)
        job.host_loop = (nsplit > 1 or (self.prior is not None and prior_desc is None)
                         or bool(os.environ.get("NGMIX_LM_HOST_LOOP"))
                         or bool(getattr(self, "host_loop", False)))
        self._mark(job, "setup")
        if job.host_loop:
            self._rounds_host_loop(job)
            self._queue_results(job)
            return job
        # ---- the whole fit, queued blind: R rounds sized by the last batch's
        # count, then finalize / pack / downloads.  _collect() finds out
        # whether R was enough (the rare miss: more rounds, results re-made).
        hint = getattr(self, "_rounds_hint", None)
        first = 4 if hint is None else max(1, min(int(hint), ROUNDS_HINT_CAP))
        self._queue_rounds(job, first)
        self._mark(job, "loop")
        self._queue_results(job)
        return job

    @staticmethod
    def _job_stream(job):
        """the stream a job was queued on (job.stream, a raw handle) as a torch
        stream, for the rare paths that queue torch work behind it again"""
        torch = _torch()
        handle = job.stream.value or 0
        if handle == 0:
            return torch.cuda.default_stream(job.dev)
        return torch.cuda.ExternalStream(handle, device=job.dev)

    def _lm_mode(self):
        """ngmix_lm_state.mode of this fitter's fits: lmdif, or lmder with the
        jacobian left out of the trials that are predicted to end the fit
        (fitter.lazy_jacobian = False or NGMIX_LM_EAGER_JAC: the jacobian with
        every evaluation; the iterates are the same to the bit either way)"""
        if self.fd:
            return _lib.LM_MODE_FD
        lazy = getattr(self, "lazy_jacobian", True) and \
            not os.environ.get("NGMIX_LM_EAGER_JAC") and \
            not os.environ.get("NGMIX_LM_JBASIS")
        return _lib.LM_MODE_ANALYTIC_LAZY if lazy else _lib.LM_MODE_ANALYTIC

    def _problem(self, job):
        """the ngmix_lm_problem record of a job (kept alive with it)"""
        P = _lib.LMProblem()
        P.batch = ctypes.pointer(job.batch)
        P.states = job.d_states.data_ptr()
        P.nobj = job.nobj
        opt = lambda t: t.data_ptr() if t is not None else None
        P.stamp_obj = opt(job.d_sobj)
        P.stamp_band = opt(job.d_sband)
        P.obj_start = opt(job.d_start)
        P.psf = job.psf.data.data_ptr() if job.psf is not None else None
        P.sums = job.d_sums.data_ptr()
        P.status = job.d_status.data_ptr()
        P.stamp_stats = opt(job.d_sstats)
        P.obj_stats = opt(job.d_ostats)
        if job.prior_desc is not None:
            P.prior = job.prior_desc.ctypes.data
            P.obj_sums = job.d_osums.data_ptr()
        P.prior_step = STEP_PRIOR
        P.model = job.modnum
        P.fd = int(self.fd)
        P.npsf = job.npsf
        P.nloc_npars = self._nloc_npars(job.npars)
        P.jac_point = opt(job.d_jacpt)
        return P

    def _nloc_npars(self, npars):
        """the nloc argument of ngmix_lm_advance_batch carrying the fits'
        parameter count (nloc + 256 npars), which selects the step's kernel: the
        register form for 6-8 parameters, the team form for 9-14;
        fitter.advance_hint = False asks for the generic one-thread form
        (NGMIX_LM_NPARS_GENERIC: what the tests compare the other two with,
        record by record)"""
        hint = npars if getattr(self, "advance_hint", True) else _lib.LM_NPARS_GENERIC
        return self.nloc + 256 * hint

    def _timing_events(self, n):
        """n hipEvent_t handles when fitter.time_kernels is set (bench.py), else
        None; released with the fitter"""
        if not getattr(self, "time_kernels", False):
            return None
        pool = self.__dict__.setdefault("_event_pool", _EventPool())
        return pool.take(n)

    def _record(self, events, i, stream=None):
        if events is not None:
            _lib.check(_lib.lib().ngmix_event_record(events[i], stream or _stream()),
                       "ngmix_event_record")

    def _queue_rounds(self, job, nrounds):
        """nrounds lock-step rounds by one call into the library
        (ngmix_lm_rounds_batch): no host between the launches"""
        torch = _torch()
        L = _lib.lib()
        if not hasattr(job, "problem"):
            job.problem = self._problem(job)
        d_counts = torch.empty(nrounds, dtype=torch.int32, device=job.dev)
        h_counts = torch.empty(nrounds, dtype=torch.int32, pin_memory=True)
        ev = self._timing_events(3 * nrounds)
        span = _HipEvents(2)
        # (the caller holds the device context; the copy of the counts is
        # queued by the library on the same stream)
        span.record(0, job.stream)
        _lib.check(L.ngmix_lm_rounds_batch(
            ctypes.byref(job.problem), nrounds, _dptr(d_counts),
            ctypes.c_void_p(h_counts.data_ptr()), ev, job.stream),
            "ngmix_lm_rounds_batch")
        span.record(1, job.stream)
        job.chunks.append((nrounds, d_counts, h_counts, ev, span))

    def _rounds_done(self, job):
        """after the last queued chunk has run: the count of fits still running"""
        return int(job.chunks[-1][2][-1])

    def _collect(self, job):
        """the host's half: wait for the downloads, make sure the rounds queued
        blind were enough (else run more and re-make the results), package"""
        import time
        torch = _torch()
        t0 = time.perf_counter()
        job.copied.synchronize()
        t1 = time.perf_counter()
        if not job.host_loop:
            grow = 2
            total = sum(c[0] for c in job.chunks)
            redo = False
            while self._rounds_done(job) != 0:
                if total > 2 * job.maxfev + 5:
                    raise RuntimeError("batched LM did not terminate")
                redo = True
                # (on the fit's OWN stream, whatever stream the consumer of
                # go_stream() iterates under, and waited for through the event the
                # chunk records there: the counts read next are this chunk's)
                with torch.cuda.device(job.dev), torch.cuda.stream(self._job_stream(job)):
                    self._queue_rounds(job, grow)
                _lib.check(_lib.lib().ngmix_event_synchronize(job.chunks[-1][4].handles[1]),
                           "ngmix_event_synchronize")
                total += grow
                grow = min(2 * grow, ROUNDS_HINT_CAP)
            if redo:
                with torch.cuda.device(job.dev), torch.cuda.stream(self._job_stream(job)):
                    self._queue_results(job)
                job.copied.synchronize()
            # counts after each round -> the rounds that had fits to advance
            counts = np.concatenate([c[2].numpy() for c in job.chunks])
            before = np.concatenate([[job.nobj], counts[:-1]])
            job.useful_rounds = int((before > 0).sum())
            self.rounds_launched = int(counts.size)
            self._rounds_hint = job.useful_rounds + 1
            # the device time of the rounds (events around every chunk)
            self.loop_seconds = sum(c[4].elapsed_ms(0, 1) for c in job.chunks) * 1e-3
            if job.chunks[0][3] is not None:
                self._kernel_times(job, before)
        else:
            self.loop_seconds = job.loop_ms_host * 1e-3
            self.rounds_launched = job.useful_rounds
            if job.legacy_ev is not None:
                self._kernel_times_host_loop(job)
        self.rounds = job.useful_rounds
        self.nsplit_used = job.nsplit
        self._d_states = job.d_states
        if getattr(self, "keep_job", False):
            self.last_job = job   # (tests: the device buffers of the batch)
        # (fitter.gmix belongs to the batch whose result is being returned)
        self._fit_ctx = job.fit_ctx
        self._gmix = None
        self._mark(job, "download")
        res = self._package(job)
        self._mark(job, "package")
        self.host_ms = {"enqueue": job.host_enqueue_ms, "wait": (t1 - t0) * 1e3,
                        "package": (time.perf_counter() - t1) * 1e3}
        return res

    def _kernel_times(self, job, before):
        """HIP-event times of every launch of the rounds (time_kernels)"""
        L = _lib.lib()
        ms = ctypes.c_float()

        def elapsed(a, b):
            _lib.check(L.ngmix_event_elapsed_ms(a, b, ctypes.byref(ms)), "elapsed")
            return float(ms.value)
        ev_ms, adv_ms = [], []
        for nr, _, _, ev, _ in job.chunks:
            for r in range(nr):
                ev_ms.append(elapsed(ev[3 * r], ev[3 * r + 1]))
                adv_ms.append(elapsed(ev[3 * r + 1], ev[3 * r + 2]))
        nstamp_obj = job.ns / float(job.nobj)
        w = before.astype("f8") * nstamp_obj
        # (the launches that found every fit finished are left out of the mean)
        live = before > 0
        self.eval_ms = float(np.mean(np.asarray(ev_ms)[live]))
        self.eval_ms_total = float(np.sum(ev_ms))
        self.eval_stamps_total = float(np.sum(w))
        self.eval_launches = [(float(t), float(x)) for t, x in zip(ev_ms, w)]
        self.eval_launch_count = int(live.sum())
        self.advance_ms_total = float(np.sum(adv_ms))
        self.kernel_ms = {
            "lm_eval": float(np.sum(ev_ms)), "lm_advance": float(np.sum(adv_ms)),
            "lm_init": elapsed(job.ev_init[0], job.ev_init[1]),
            "lm_finalize": elapsed(job.ev_post[0], job.ev_post[1]),
            "lm_pack": elapsed(job.ev_post[1], job.ev_post[2]),
        }
        pool = self.__dict__.get("_event_pool")
        if pool is not None:
            pool.give(job.ev_init)
            pool.give(job.ev_post)
            for c in job.chunks:
                pool.give(c[3])

    def _kernel_times_host_loop(self, job):
        ev = job.legacy_ev
        # per-launch mean, and the whole fit: stamps evaluated / time spent
        # (a launch whose count was never read -- check_every > 1 -- keeps
        # the last count known before it)
        last = 0.0
        for rec_ in ev:
            if rec_[2] is None:
                rec_[2] = last
            last = rec_[2]
        ms = [a.elapsed_time(b) for a, b, _ in ev]
        self.eval_ms = float(np.mean(ms))
        self.eval_ms_total = float(np.sum(ms))
        self.eval_stamps_total = float(np.sum([w for _, _, w in ev]))
        self.eval_launches = [(float(t), float(w)) for t, (_, _, w) in zip(ms, ev)]
        self.eval_launch_count = len(ms)

    def _rounds_host_loop(self, job):
        """the lock-step loop driven from the host, one round at a time (a prior
        evaluated by torch ops between the two launches of a round; pieces of
        the batch on several streams, fitter.nsplit).  One round is kept in
        flight ahead of the count being read."""
        import time
        torch = _torch()
        L = _lib.lib()
        dev = job.dev
        nobj, npars, ns, npsf = job.nobj, job.npars, job.ns, job.npsf
        psf, obj_start = job.psf, job.obj_start
        d_states, d_sums, d_status = job.d_states, job.d_sums, job.d_status
        d_sobj, d_sband, d_start = job.d_sobj, job.d_sband, job.d_start
        d_sstats, d_ostats, d_osums = job.d_sstats, job.d_ostats, job.d_osums
        prior_desc, loop_stats, nsum = job.prior_desc, job.loop_stats, job.nsum
        modnum, maxfev, check_every = job.modnum, job.maxfev, job.check_every
        b = job.batch
        # float64 view of the state records: the columns a prior needs
        fields = _lib.LM_STATE_DTYPE.fields
        sview = d_states.view(torch.float64)

        def col(name):
            a = fields[name][1] // 8
            return sview[:, a:a + npars]
        # Pieces of the batch on separate streams (fitter.nsplit = k, or the
        # NGMIX_LM_NSPLIT environment knob): one piece's lm_advance -- one thread
        # per fit, latency bound -- runs under another piece's pixel pass.
        # Measured on 100k fits: 6.65 against 6.70 ms for the loop once lm_advance
        # runs from registers, so the default is one piece.
        nsplit = job.nsplit
        isz = _lib.LM_STATE_DTYPE.itemsize
        wosum = npars * (npars + 1) // 2 + npars + 1

        def off(t, nbytes):
            return ctypes.c_void_p(t.data_ptr() + int(nbytes)) if t is not None else None
        subs = []
        for k in range(nsplit):
            o_lo, o_hi = nobj * k // nsplit, nobj * (k + 1) // nsplit
            s_lo, s_hi = int(obj_start[o_lo]), int(obj_start[o_hi])
            bk = _lib.Batch()
            ctypes.memmove(ctypes.byref(bk), ctypes.byref(b), ctypes.sizeof(bk))
            bk.nstamps = s_hi - s_lo
            bk.stamps = b.stamps + s_lo * _lib.STAMP_DTYPE.itemsize
            bk.jac = b.jac + s_lo * _lib.JACOBIAN_DTYPE.itemsize
            subs.append({
                "o": (o_lo, o_hi), "s": (s_lo, s_hi), "batch": bk,
                "stream": self._side_stream(dev, 1 + k) if nsplit > 1 else None,
                "nact": torch.zeros(1, dtype=torch.int32, device=dev),
                "live": True, "active": o_hi - o_lo,
                "ring": [torch.zeros(1, dtype=torch.int32, pin_memory=True)
                         for _ in range(3)],
                "pend": [], "launched": 0, "ev_idx": [],
            })
        rounds = 0
        if not job.streaming:
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        # time_kernels: HIP events around every pixel-pass launch (bench.py),
        # with the number of stamps each launch still had to evaluate
        ev = [] if getattr(self, "time_kernels", False) else None
        nstamp_obj = ns / float(nobj)

        def enqueue(sub):
            (o_lo, o_hi), (s_lo, s_hi) = sub["o"], sub["s"]
            if ev is not None:
                # (the stamps this launch still has to evaluate are known when
                # the previous round's counter is read: filled in then)
                ev.append([torch.cuda.Event(enable_timing=True),
                           torch.cuda.Event(enable_timing=True),
                           sub["active"] * nstamp_obj if not sub["launched"] else None])
                sub["ev_idx"].append(len(ev) - 1)
                ev[-1][0].record()
            _lib.check(L.ngmix_lm_eval_batch(
                ctypes.byref(sub["batch"]), modnum, int(self.fd), _dptr(d_states),
                off(d_sobj, 4 * s_lo), off(d_sband, 4 * s_lo),
                off(psf.data, 104 * npsf * s_lo) if psf is not None else None,
                npsf, off(d_sums, 8 * nsum * s_lo), off(d_status, 4 * s_lo),
                off(d_sstats, 16 * s_lo), _stream()),
                "ngmix_lm_eval_batch")
            if ev is not None:
                ev[-1][1].record()
            if prior_desc is not None:
                # the prior rows at the same trial points (results.py:454)
                _lib.check(L.ngmix_lm_prior_sums_batch(
                    off(d_states, isz * o_lo), o_hi - o_lo, _lib.ptr(prior_desc),
                    STEP_PRIOR, off(d_osums, 8 * wosum * o_lo), _stream()),
                    "ngmix_lm_prior_sums_batch")
            elif self.prior is not None:
                osums_box[0] = prior_normal_sums(
                    self.prior, col("xt"), col("xstep"), col("hstep"))[0] if self.fd \
                    else prior_normal_sums(self.prior, col("xt"))[0]
            osums = d_osums if prior_desc is not None else osums_box[0]
            # obj_start holds absolute stamp indices: sums / stamp_band stay whole
            _lib.check(L.ngmix_lm_advance_batch(
                off(d_states, isz * o_lo), o_hi - o_lo, off(d_start, 8 * o_lo),
                _dptr(d_sband), _dptr(d_sums), self._nloc_npars(npars),
                off(osums, 8 * wosum * o_lo) if osums is not None else None,
                _dptr(sub["nact"]), _dptr(d_sstats) if loop_stats else None,
                off(d_ostats, 16 * o_lo), _stream()),
                "ngmix_lm_advance_batch")
            # the count of fits still running, on its way to the host behind
            # the round that made it
            hook = getattr(self, "round_hook", None)
            if hook is not None:
                # (tests: the state records after every round of the host loop)
                hook(job, sub["launched"])
            h = sub["ring"][sub["launched"] % len(sub["ring"])]
            h.copy_(sub["nact"], non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            sub["pend"].append((done, h))
            sub["launched"] += 1

        osums_box = [None]
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            if nsplit > 1:
                start = torch.cuda.Event()
                start.record(main)
                for sub in subs:
                    sub["stream"].wait_event(start)
            # One round is kept in flight AHEAD of the counter being read: the
            # host reads round r's count of running fits (a 4-byte pinned copy
            # behind that round) while round r + 1 already runs, so the GPU
            # never waits for the host between rounds.  When the count is zero
            # the round in flight finds every fit finished and both its kernels
            # return at once (~30 us, once per call).
            depth = 1 if check_every == 1 else 0
            while any(sub["live"] for sub in subs):
                for sub in subs:
                    if not sub["live"]:
                        continue
                    with torch.cuda.stream(sub["stream"] or main):
                        if len(sub["pend"]) > depth and (
                                sub["launched"] % check_every == 0 or
                                sub["launched"] > 2 * maxfev):
                            while len(sub["pend"]) > depth:
                                done, h = sub["pend"].pop(0)
                            done.synchronize()
                            sub["active"] = int(h[0])
                            # (that count is what the launch after it evaluates)
                            k = sub["launched"] - len(sub["pend"])
                            if ev is not None and k < len(sub["ev_idx"]):
                                ev[sub["ev_idx"][k]][2] = sub["active"] * nstamp_obj
                            if sub["active"] == 0:
                                sub["live"] = False
                                continue
                        enqueue(sub)
                rounds += 1
                if rounds > 2 * maxfev + 5:
                    raise RuntimeError("batched LM did not terminate")
            # rounds that had fits to advance (not the one in flight at the end)
            rounds = max(sub["launched"] - len(sub["pend"]) for sub in subs)
            if nsplit > 1:
                for sub in subs:
                    main.wait_stream(sub["stream"])
        if not job.streaming:
            torch.cuda.synchronize(dev)
        # seconds in the lock-step loop (kernels + one 4-byte readback per round)
        job.loop_ms_host = (time.perf_counter() - t0) * 1e3
        job.useful_rounds = rounds
        job.legacy_ev = ev
        self._mark(job, "loop")

    def _queue_results(self, job):
        """run_leastsq's packaging, one thread per fit (ngmix_lm_finalize_batch),
        the statistics / packing kernel and the downloads, all queued"""
        torch = _torch()
        L = _lib.lib()
        dev = job.dev
        stamps, psf = job.stamps, job.psf
        nobj, n = job.nobj, job.npars
        obj_start, sband, sobj = job.obj_start, job.sband, job.sobj
        d_states = job.d_states
        fields = _lib.LM_STATE_DTYPE.fields
        sview = d_states.view(torch.float64)

        def col(name):
            a = fields[name][1] // 8
            return sview[:, a:a + n]
        d_npix = job.d_npix
        width = 4 + 2 * n + 2 * n * n
        arena = job.arena
        d_rec = arena["rec"] if arena is not None else \
            torch.empty((nobj, width), dtype=torch.float64, device=dev)
        d_ffx = None
        if self.prior is not None:
            # chi2/dof is over fdiff[n_prior_pars:] (leastsqbound.py:97): the
            # prior rows are left out -- and, the reference reserving more
            # slots than PriorSimpleSep fills (results.py:1050-1078 against
            # joint_prior.py:86-120) while the pixel rows start right after the
            # filled ones (results.py:454-461), so are the first pixels
            job.d_prior_lnp = None
            if job.prior_desc is not None:
                # one launch at the fits' points (the record the prior kernel
                # evaluated during the rounds)
                d_ffx = torch.empty(nobj, dtype=torch.float64, device=dev)
                job.d_prior_lnp = torch.empty(nobj, dtype=torch.float64, device=dev)
                _lib.check(L.ngmix_lm_prior_finish_batch(
                    _dptr(d_states), nobj, _lib.ptr(job.prior_desc), _dptr(d_ffx),
                    _dptr(job.d_prior_lnp), job.stream), "ngmix_lm_prior_finish_batch")
                nrows = 4 + int(job.prior_desc["nmid"][0]) + int(job.prior_desc["nband"][0])
            else:
                rows, bad = self.prior.fill_fdiff_batch(col("x"))
                d_ffx = torch.where(bad, torch.zeros_like(rows[:, 0]),
                                    (rows * rows).sum(dim=1))
                d_ffx = torch.where(torch.isfinite(d_ffx), d_ffx, torch.zeros_like(d_ffx))
                nrows = rows.shape[1]
            nskip = get_lm_n_prior_pars(self.model, job.nband) - nrows
            if nskip > 0:
                d_ffx = d_ffx + self._first_pixels_fdiff2(
                    stamps, psf, col("x"), obj_start, sband, nskip)
            d_ffx = d_ffx.contiguous()
        ev_post = self._timing_events(3)
        if job.d_jacpt is not None and not job.host_loop:
            # R / ipvt of the fits that ended, re-made from double-double normal
            # equations at the point of their last jacobian
            if not hasattr(job, "problem"):
                job.problem = self._problem(job)
            job.d_psums = torch.empty((job.ns, 2, job.nsum), dtype=torch.float64, device=dev)
            _lib.check(L.ngmix_lm_precise_cov_batch(
                ctypes.byref(job.problem), _dptr(job.d_psums), job.stream),
                "ngmix_lm_precise_cov_batch")
        if True:
            self._record(ev_post, 0, job.stream)
            _lib.check(L.ngmix_lm_finalize_batch(
                _dptr(d_states), nobj, _dptr(d_npix),
                _dptr(d_ffx) if d_ffx is not None else None, float(PDEF), float(CDEF),
                _dptr(d_rec), job.stream), "ngmix_lm_finalize_batch")
            self._record(ev_post, 1, job.stream)
        self._mark(job, "finalize")
        # Two downloads through pinned memory on a side stream (PyTorch's
        # caching host allocator: no hipHostMalloc after the first call):
        #   * the head -- pars, pars_err, flags / nfev / ier / dof / njev and
        #     the seven statistics columns, 2 n + 12 doubles per fit -- which
        #     go() waits for;
        #   * the upper triangle of pars_cov (n (n + 1) / 2 per fit: the matrix
        #     is symmetric to the bit), which keeps flowing after go() has
        #     returned and is waited for -- and mirrored -- when pars_cov is
        #     first read (LMBatchResult lazy key; so are the blocks cut from it).
        # pars_cov0 stays on the device until it is asked for.
        c0, c1 = 4 + 2 * n, 4 + 2 * n + n * n
        job.d_cov0 = d_rec[:, c0:c1]
        ntri = n * (n + 1) // 2
        d_tri = arena["tri"] if arena is not None else \
            torch.empty((nobj, ntri), dtype=torch.float64, device=dev)
        side = self._side_stream(dev)
        d_ok = d_rec[:, 0] == 0.0
        job.fit_ctx = self._fit_ctx = (stamps, psf, sobj, sband, d_rec, d_ok, n)
        self._gmix = None
        tot = None
        if not job.loop_stats:
            tot = self._loglike_at_solutions(
                stamps, psf, sobj, sband, obj_start,
                prior_lnp=getattr(job, "d_prior_lnp", None)).contiguous()
        # pars | pars_err rows, then twelve contiguous columns (integers and
        # statistics): one kernel, one download, contiguous host views
        ncols = _lib.LM_NCOLS
        d_flat = arena["flat"] if arena is not None else \
            torch.empty(nobj * (2 * n + ncols), dtype=torch.float64, device=dev)
        if True:
            _lib.check(L.ngmix_lm_pack_batch(
                _dptr(d_states), nobj, n, _dptr(d_rec),
                _dptr(job.d_ostats) if job.loop_stats else None,
                _dptr(tot), _dptr(d_npix), _dptr(d_flat),
                ctypes.c_void_p(d_flat.data_ptr() + 8 * nobj * 2 * n), _dptr(d_tri),
                job.stream),
                "ngmix_lm_pack_batch")
            self._record(ev_post, 2, job.stream)
        self._mark(job, "pack")
        if arena is not None:
            # one download of the three outputs, on the fit's own stream (a
            # side stream buys nothing when the host waits for all of it)
            d_out = arena["out"]
            h_out = torch.empty(d_out.shape, dtype=torch.uint8, pin_memory=True)
            h_out.copy_(d_out, non_blocking=True)
            copied = cov_copied = torch.cuda.Event()
            copied.record()
            oo = arena["out_off"]

            def hview(name, like):
                nb = like.numel() * 8
                return h_out[oo[name]:oo[name] + nb].view(torch.float64).reshape(like.shape)
            h_flat, h_tri = hview("flat", d_flat), hview("tri", d_tri)
            h_cov0 = hview("rec", d_rec)[:, c0:c1]
        else:
            h_flat = torch.empty(d_flat.shape, dtype=torch.float64, pin_memory=True)
            h_tri = torch.empty((nobj, ntri), dtype=torch.float64, pin_memory=True)
            self._mark(job, "pinned_alloc")
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                h_flat.copy_(d_flat, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record()
                h_tri.copy_(d_tri, non_blocking=True)
                h_cov0 = None
                if nobj <= 4096:
                    # (a small batch: pars_cov0 rides along instead of costing its
                    # first reader a synchronous download of its own)
                    h_cov0 = torch.empty((nobj, n * n), dtype=torch.float64,
                                         pin_memory=True)
                    h_cov0.copy_(job.d_cov0, non_blocking=True)
                cov_copied = torch.cuda.Event()
                cov_copied.record()
            d_flat.record_stream(side)
            d_tri.record_stream(side)
        pool = self.__dict__.get("_event_pool")
        if pool is not None and job.ev_post is not None:
            pool.give(job.ev_post)   # (results re-made after a miss)
        job.ev_post = ev_post
        job.h_flat, job.h_tri, job.h_cov0 = h_flat, h_tri, h_cov0
        job.copied, job.cov_copied = copied, cov_copied
        self._mark(job, "enqueue_copy")

    def _package(self, job):
        """the result dict over the downloaded block (views, no copies)"""
        nobj, n = job.nobj, job.npars
        ncols = _lib.LM_NCOLS
        flat = job.h_flat.numpy()
        rec = flat[:nobj * 2 * n].reshape(nobj, 2 * n)
        cols = flat[nobj * 2 * n:].reshape(ncols, nobj)
        res = LMBatchResult({
            "model": self.model,
            "flags": cols[0].astype(np.int64),
            "nfev": cols[1].astype(np.int64),
            "njev": cols[4].astype(np.int64),
            "ier": cols[2].astype(np.int64),
            # views of the downloaded block (no copies)
            "pars": rec[:, 0:n],
            "pars_err": rec[:, n:2 * n],
            "npix": job.npix_obj,
            "dof": cols[3].astype(np.int64),
        })
        h_tri, cov_copied, d_cov0 = job.h_tri, job.cov_copied, job.d_cov0

        def fetch_cov(_):
            cov_copied.synchronize()
            iu = np.triu_indices(n)
            a = np.empty((nobj, n, n))
            tri = h_tri.numpy()
            a[:, iu[0], iu[1]] = tri
            a[:, iu[1], iu[0]] = tri
            a.flags.writeable = False
            return a
        res.set_lazy("pars_cov", fetch_cov)
        h_cov0 = job.h_cov0

        def fetch_cov0(_):
            if h_cov0 is None:
                return d_cov0.cpu().numpy().reshape(nobj, n, n)
            cov_copied.synchronize()
            return h_cov0.numpy().reshape(nobj, n, n)
        res.set_lazy("pars_cov0", fetch_cov0)
        self._add_stats(res, cols[5:], job.nband)
        return res

    @property
    def gmix(self):
        """the fitted (pre-psf) mixtures of the last go(), one per stamp"""
        if self._gmix is None:
            self._fitted_mixtures()
        return self._gmix

    def _nsplit_wanted(self):
        nsplit = getattr(self, "nsplit", None)
        if nsplit is None and os.environ.get("NGMIX_LM_NSPLIT"):
            nsplit = int(os.environ["NGMIX_LM_NSPLIT"])   # A/B knob
        return 1 if nsplit is None else int(nsplit)

    def _side_stream(self, dev, which=0):
        torch = _torch()
        cache = self.__dict__.setdefault("_side_streams", {})
        key = (dev.type, dev.index, which)
        if key not in cache:
            cache[key] = torch.cuda.Stream(device=dev)
        return cache[key]

    def _first_pixels_fdiff2(self, stamps, psf, x, obj_start, sband, nskip):
        """sum of fdiff^2 over the first nskip listed pixels of each object's
        first stamp at the parameters x (device tensor (nobj, npars)): the rows
        the reference's chi2/dof leaves out next to the prior's (see go())"""
        torch = _torch()
        dev = stamps.device
        nobj = x.shape[0]
        nshape = self.nloc - 1
        s0 = obj_start[:-1]
        band_pars = torch.empty((nobj, self.nloc), dtype=torch.float64, device=dev)
        band_pars[:, :nshape] = x[:, :nshape]
        idx = torch.from_numpy(nshape + sband[s0].astype(np.int64)).to(dev)
        band_pars[:, nshape] = x.gather(1, idx[:, None])[:, 0]
        gm, _ = GMixBatch.from_pars(band_pars, self.model, device=dev, ngauss=self.ngauss)
        d_s0 = torch.from_numpy(s0).to(dev)
        if psf is not None:
            pdata = psf.data.reshape(stamps.n, psf.ngauss, 13)[d_s0]
            gm, _ = gm.convolve(GMixBatch(pdata.reshape(-1, 13).contiguous(), nobj,
                                          psf.ngauss))
        gm.set_norms()
        # one launch: the first nskip listed pixels of each object's first
        # stamp through fill_fdiff's own arithmetic
        out = torch.empty(nobj, dtype=torch.float64, device=dev)
        b = stamps._batch(1)
        with _on_device(dev):
            st = _lib.lib().ngmix_first_pixels_fdiff2_batch(
                ctypes.byref(b), _dptr(d_s0), _dptr(gm.data), gm.ngauss, nobj, int(nskip),
                _dptr(out), _stream())
        _lib.check(st, "ngmix_first_pixels_fdiff2_batch")
        return out

    def states(self):
        """the raw ngmix_lm_state records of the last go() (debugging)"""
        return self._d_states.cpu().numpy().reshape(-1).view(_lib.LM_STATE_DTYPE)

    def _fitted_mixtures(self):
        """the mixtures at the solutions of the last go(): (pre-psf, convolved),
        one per stamp; a harmless model stands in for failed fits (their
        statistics are not reported)"""
        torch = _torch()
        stamps, psf, sobj, sband, d_rec, d_ok, n = self._fit_ctx
        dev = stamps.device
        nobj = d_rec.shape[0]
        nshape = self.nloc - 1
        default = np.zeros(n)
        default[4] = 1.0
        if self.model == "bdf":
            default[5] = 0.5
        if self.model == "bd":
            default[6] = 0.5
        default[nshape:] = 1.0
        if self.model == "coellip":
            default[4:] = 1.0
        usable = torch.where(d_ok[:, None], d_rec[:, 4:4 + n],
                             torch.from_numpy(default).to(dev)[None, :])
        if stamps.n == nobj and np.all(sband == 0):
            band_pars = usable[:, :self.nloc].contiguous()
        else:
            d_sobj = torch.from_numpy(sobj.astype(np.int64)).to(dev)
            d_sband = torch.from_numpy(sband.astype(np.int64)).to(dev)
            band_pars = torch.empty((stamps.n, self.nloc), dtype=torch.float64,
                                    device=dev)
            per_stamp = usable[d_sobj]
            band_pars[:, :nshape] = per_stamp[:, :nshape]
            band_pars[:, nshape] = per_stamp.gather(1, (nshape + d_sband)[:, None])[:, 0]
        gm0, st0 = GMixBatch.from_pars(band_pars, self.model, device=dev,
                                       ngauss=self.ngauss)
        gm = gm0
        if psf is not None:
            gm, _ = gm0.convolve(psf)
        self._gmix = gm0
        return gm, usable

    def _loglike_at_solutions(self, stamps, psf, sobj, sband, obj_start, prior_lnp=None):
        """the device half of FitModel.set_fit_result (results.py:45-72,
        398-408) when the lock-step loop did not carry the statistics
        (forward-difference fits, fits with a prior): one batched get_loglike
        at the solutions, folded per object on the device; (nobj, 4) lnprob,
        s2n_numer, s2n_denom, npix"""
        torch = _torch()
        dev = stamps.device
        gm, usable = self._fitted_mixtures()
        nobj = usable.shape[0]
        out, st1 = stamps.loglike(gm)
        if stamps.n == nobj:
            tot = out
        else:
            # (stamps of an object are contiguous: a fixed-order segmented sum)
            lengths = torch.from_numpy(np.diff(obj_start)).to(dev)
            tot = torch.segment_reduce(out, "sum", lengths=lengths, axis=0)
        if self.prior is not None:
            # calc_lnprob adds the joint prior (results.py:410-437)
            tot = tot.clone()
            if prior_lnp is not None:
                tot[:, 0] += prior_lnp
            else:
                tot[:, 0] += self.prior.get_lnprob_batch(usable.contiguous())
        return tot

    def _add_stats(self, res, out, nband):
        """the host half: the keys FitModel.set_fit_result adds for fits with
        flags == 0 (NaN elsewhere); out: the seven statistics columns of
        ngmix_lm_pack_batch (contiguous rows); g / T / flux blocks are VIEWS
        of the downloaded arrays"""
        pars = res["pars"]
        nshape = self.nloc - 1
        res["lnprob"] = out[0]
        res["s2n_numer"] = out[1]
        res["s2n_denom"] = out[2]
        res["npix"] = out[3].astype(np.int64)
        res["dof"] = out[4].astype(np.int64)
        res["chi2per"] = out[5]
        res["s2n_w"] = out[6]
        res["s2n"] = res["s2n_w"]
        perr = res["pars_err"]
        res["g"] = pars[:, 2:4]
        res["g_err"] = perr[:, 2:4]
        # (blocks of pars_cov: read when asked for, like pars_cov itself)
        res.set_lazy("g_cov", lambda r: r["pars_cov"][:, 2:4, 2:4])
        # pars_err is sqrt(diag(pars_cov)) (fitters.py:333-339), made by the
        # finalize kernel with the same IEEE square root
        res["T"] = pars[:, 4]
        res["T_err"] = perr[:, 4]
        if self.model == "coellip":
            # CoellipFitModel._set_flux is a no-op (results.py:648-652); _set_T
            # is not overridden: T is pars[4], the first component's
            return
        if nband == 1:
            res["flux"] = pars[:, nshape]
            res["flux_err"] = perr[:, nshape]
        else:
            res["flux"] = pars[:, nshape:]
            res.set_lazy("flux_cov", lambda r: r["pars_cov"][:, nshape:, nshape:])
            res["flux_err"] = perr[:, nshape:]
