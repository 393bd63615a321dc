"""
Linear template / psf flux by zero-lag cross-correlation (reference API:
ngmix/fitting/fitters.py:144-181, results.py:677-914).

The arithmetic lives in ONE place, PSFFluxBatch: an exact render of every
stamp's unit-flux mixture (the render HIP kernel) or an uploaded template
image, then the cross-correlation sums as segmented reductions on the device.
The per-object PSFFluxFitter / PSFFluxFitModel of the reference API is a
one-object view of that batch.
"""
import numpy as np

from .defaults import PDEF, CDEF
from .flags import DIV_ZERO, BAD_VAR
from .observation import Observation, ObsList

__all__ = ["PSFFluxFitter", "PSFFluxFitModel", "PSFFluxBatch"]

_RESULT_KEYS = ("flags", "chi2per", "dof", "flux", "flux_err")


class PSFFluxFitModel(dict):
    """
    Result dict of the template-flux fit of ONE object (an Observation or an
    ObsList of its epochs): keys model, flags, chi2per, dof, flux, flux_err.

    What is correlated with the image, per epoch (results.py:832-897):
      do_psf=True   the psf observation's mixture (unit flux when
                    normalize_psf, its own flux otherwise), or its `template`
                    image when it has no mixture (normalised likewise);
      do_psf=False  the observation's own mixture / template, unit flux.
    """

    def __init__(self, obs, do_psf=True, normalize_psf=True):
        self["model"] = "template"
        self.npars = 1
        self.do_psf = do_psf
        self.normalize_psf = normalize_psf
        if isinstance(obs, Observation):
            epochs = ObsList()
            epochs.append(obs)
        elif isinstance(obs, ObsList):
            epochs = obs
        else:
            raise ValueError("obs should be Observation or ObsList")
        self.obs = epochs
        holders = [o.psf if do_psf else o for o in epochs]
        first = holders[0]
        self.use_template = not first.has_gmix()
        if self.use_template and not hasattr(first, "template"):
            raise ValueError("neither gmix or template image are set")
        self._holders = holders
        self.totpix = sum(o.pixels.size for o in epochs)

    # what the batch needs from the epochs ---------------------------------
    def _mixtures(self):
        """(nepoch, ngmax) gauss2d records; shorter mixtures are padded with
        zero-flux round gaussians, which render exactly 0"""
        from . import _lib
        mixes = [h.get_gmix().get_data() for h in self._holders]
        ngmax = max(m.size for m in mixes)
        rec = np.zeros((len(mixes), ngmax), dtype=_lib.GAUSS2D_DTYPE)
        rec["irr"] = rec["icc"] = rec["det"] = 1.0
        for i, m in enumerate(mixes):
            for name in ("p", "row", "col", "irr", "irc", "icc", "det"):
                rec[name][i, :m.size] = m[name]
        return rec

    def _templates(self):
        """flattened template images with unit sum, and the sums"""
        flat = [np.asarray(h.template, dtype="f8").ravel() for h in self._holders]
        sums = np.array([t.sum() for t in flat])
        return np.concatenate([t * (1.0 / s) for t, s in zip(flat, sums)]), sums

    def go(self):
        from .batch import StampBatch, GMixBatch
        stamps = StampBatch.from_observations(list(self.obs))
        nep = len(self.obs)
        same_object = np.zeros(nep, dtype=np.int64)
        # only the psf keeps its own normalisation, and only when asked to
        unit = not (self.do_psf and not self.normalize_psf)
        batch = PSFFluxBatch(normalize_psf=unit)
        if self.use_template:
            tmpl, sums = self._templates()
            res = batch.go_templates(stamps, tmpl, None if unit else sums,
                                     stamp_obj=same_object)
        else:
            gm = GMixBatch.from_numpy(self._mixtures(), device=stamps.device)
            res = batch.go(stamps, gm, stamp_obj=same_object)
        for k in _RESULT_KEYS:
            v = res[k][0]
            # dof is npix - 1, an integer, unless floored at 1e-6
            # (results.py:806-814)
            self[k] = int(v) if k == "flags" or (k == "dof" and v >= 1.0) else float(v)

    def get_effective_npix(self):
        """pixels with positive weight, over all epochs"""
        return sum(int((o.weight > 0).sum()) for o in self.obs)

    def get_dof(self):
        dof = self.get_effective_npix() - self.npars
        return dof if dof > 0 else 1.0e-6


class PSFFluxFitter(object):
    """psf or template flux; the centre is fixed, so the fit is linear"""

    def __init__(self, do_psf=True, normalize_psf=True):
        self.do_psf = do_psf
        self.normalize_psf = normalize_psf

    def go(self, obs):
        fit_model = PSFFluxFitModel(obs=obs, do_psf=self.do_psf,
                                    normalize_psf=self.normalize_psf)
        fit_model.go()
        return fit_model

    def go_many(self, obs):
        """the psf (template) fluxes of a sequence of objects -- Observation or
        ObsList each, their psf mixtures set -- by one batch (PSFFluxBatch):
        a dict of per-object arrays flags, flux, flux_err, chi2per, dof, equal to
        go() per object (results.py:677-914)"""
        from .batch import flatten_observations, GMixBatch
        if not self.do_psf:
            raise ValueError("go_many fits psf mixtures (do_psf=True); templates: "
                             "PSFFluxBatch.go_templates")
        stamps, sobj, sband, nband, psf = flatten_observations(obs)
        if nband != 1:
            raise ValueError("one band per call (PSFFluxFitter takes an Observation or an ObsList)")
        if psf is None:
            raise ValueError("the observations need their psf mixtures")
        return PSFFluxBatch(normalize_psf=self.normalize_psf).go(
            stamps, GMixBatch.from_numpy(psf, device=stamps.device), stamp_obj=sobj,
            nobj=len(obs))


class PSFFluxBatch(object):
    """
    PSFFluxFitter over a device-resident batch: one exact render of every
    stamp's (flux-normalised) psf mixture, then the zero-lag cross-correlation
    sums of PSFFluxFitModel.go (results.py:700-770) by a fused kernel (one
    pass over the model, image and ierr planes per call: csrc/template.hip).
    Objects may own several stamps (epochs), as an ObsList does.

        res = PSFFluxBatch().go(stamps, psf_gmixes, stamp_obj=None)

    res: dict of (nobj,) arrays flags, flux, flux_err, chi2per, dof
    """

    def __init__(self, normalize_psf=True):
        self.normalize_psf = normalize_psf

    def go(self, stamps, gm, stamp_obj=None, nobj=None):
        """gm: GMixBatch, one mixture per stamp (any flux); stamp_obj: the
        object (any order) each stamp belongs to; nobj: the number of objects
        when the last ones may own no stamp (they come back DIV_ZERO)"""
        import torch
        ns = stamps.n
        assert gm.n == ns, "one mixture per stamp"
        # the unit-flux template and the norm the fitted flux multiplies
        g = gm.clone()
        data = g.data.reshape(ns, g.ngauss, 13)
        psum = data[:, :, 0].sum(dim=1)
        safe = torch.where(psum != 0, psum, torch.ones_like(psum))
        data[:, :, 0] /= safe[:, None]
        data[:, :, 7] = 0.0  # norms are stale (gmix.py set_flux)
        model, status = stamps.render(g, fast_exp=False)
        # without normalisation the mixture keeps its own flux
        norm = None if self.normalize_psf else psum
        res = self._solve(stamps, model, norm, stamp_obj, nobj)
        res["status"] = status.cpu().numpy()
        return res

    def go_templates(self, stamps, templates, norms=None, stamp_obj=None):
        """templates: the stamps' template images (unit sum), flattened back to
        back in stamp order, host or device; norms: per-stamp factor the
        template keeps (None: 1)"""
        import torch
        dev = stamps.device
        model = torch.as_tensor(np.asarray(templates, dtype="f8")
                                if not isinstance(templates, torch.Tensor) else templates,
                                dtype=torch.float64).to(dev).reshape(-1)
        assert model.numel() == stamps.total_pix
        if norms is not None and not isinstance(norms, torch.Tensor):
            norms = torch.from_numpy(np.asarray(norms, dtype="f8")).to(dev)
        return self._solve(stamps, model, norms, stamp_obj)

    @staticmethod
    def _solve(stamps, model, norm, stamp_obj, nobj=None):
        """flux = sum(m I w) / sum(m m w) per object, chi2 of the scaled
        template, and the flags / errors of PSFFluxFitModel.go
        (results.py:700-770) for every object at once.

        Weights: the stamp store keeps ierr = sqrt(max(w, 0)) (pixels_nb.py:49-52)
        and not the raw weight map, so w here is ierr * ierr: equal to the
        reference's obs.weight to one rounding (sqrt then square, <= 1 ulp per
        pixel; the goldens hold to 1e-11), and a NEGATIVE weight counts as zero
        where the reference's sums would subtract it -- the reference's own
        pixel arrays (ierr) make the same clamp for every other fitter, and
        Observation rejects nothing here either.  Whether the epochs carry
        mixtures or templates is decided by the first epoch, as the reference
        does (results.py:778-789)."""
        import torch
        dev = stamps.device
        ns = stamps.n
        if stamp_obj is None:
            sobj = np.arange(ns, dtype=np.int64)
        else:
            sobj = np.ascontiguousarray(stamp_obj, dtype=np.int64)
            if sobj.shape != (ns,) or (ns and sobj.min() < 0):
                raise ValueError("stamp_obj must be (nstamps,) object indices")
        nobj = max(int(sobj.max()) + 1 if ns else 0, int(nobj or 0))
        d_sobj = torch.from_numpy(sobj).to(dev)
        # Two launches of one fused kernel (csrc/template.hip: one pass over the
        # model, image and ierr planes each) instead of a dozen torch passes
        # over them: the first with the templates' norms gives sum(m I w) and
        # sum(m m w) per stamp -- the flux is their ratio over the object's
        # stamps --, the second with flux * norm the chi2 of the scaled template.
        import ctypes
        from . import _lib
        from .batch import _dptr, _stream, _on_device
        L = _lib.lib()
        b = stamps._batch(1)
        model = model.contiguous()

        def sums_with(mult):
            out = torch.empty((ns, 4), dtype=torch.float64, device=dev)
            with _on_device(dev):
                _lib.check(L.ngmix_template_sums_batch(
                    ctypes.byref(b), _dptr(model), _dptr(mult) if mult is not None else None,
                    _dptr(out), _stream()), "ngmix_template_sums_batch")
            return out

        def per_object(x):
            out = torch.zeros(nobj, dtype=torch.float64, device=dev)
            return out.index_add_(0, d_sobj, x)

        first = sums_with(norm)
        xcorr = per_object(first[:, 0])
        msq = per_object(first[:, 1])
        zero = msq == 0
        flux = torch.where(zero, torch.full_like(msq, PDEF),
                           xcorr / torch.where(zero, torch.ones_like(msq), msq))
        # second pass: chi2 of the scaled template (linear in the flux)
        fl = flux[d_sobj] if norm is None else flux[d_sobj] * norm
        chi2 = per_object(sums_with(fl.contiguous())[:, 2])
        chi2 = torch.where(zero, torch.zeros_like(chi2), chi2)
        positive = first[:, 3].cpu().numpy()

        xcorr, msq, flux, chi2 = (t.cpu().numpy() for t in (xcorr, msq, flux, chi2))
        kept = np.zeros(nobj, dtype=np.int64)
        np.add.at(kept, sobj, stamps.npix_kept.astype(np.int64))
        eff = np.zeros(nobj, dtype=np.int64)
        np.add.at(eff, sobj, positive.astype(np.int64))
        dof = (eff - 1).astype("f8")
        flags = np.zeros(nobj, dtype=np.int64)
        # get_dof floors a non-positive dof at 1e-6, so ZERO_DOF is never
        # raised and chi2per is always set (results.py:752-757,822-830)
        dof_used = np.where(dof <= 0, 1.0e-6, dof)
        chi2per = chi2 / dof_used
        flux_err = np.full(nobj, CDEF)
        bad = (msq == 0) | (kept == 1)
        flags[bad] |= DIV_ZERO
        with np.errstate(all="ignore"):
            arg = chi2 / np.where(msq == 0, 1.0, msq) / np.where(kept == 1, 1, kept - 1)
        neg = ~bad & ~(arg >= 0.0)
        flags[neg] |= BAD_VAR
        okv = ~bad & (arg >= 0.0)
        flux_err[okv] = np.sqrt(arg[okv])
        return {"flags": flags, "flux": flux, "flux_err": flux_err, "chi2per": chi2per,
                "dof": dof_used}
