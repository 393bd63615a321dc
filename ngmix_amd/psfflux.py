"""
Linear template / psf flux by zero-lag cross-correlation (reference:
ngmix/fitting/fitters.py:144-181, results.py:677-914).  The model images come
from GMix.make_image (the render HIP kernel); the sums are O(npix) numpy.
"""
import numpy as np

from .defaults import PDEF, CDEF
from .flags import ZERO_DOF, DIV_ZERO, BAD_VAR
from .observation import Observation, ObsList

__all__ = ["PSFFluxFitter", "PSFFluxFitModel", "PSFFluxBatch"]


class PSFFluxFitModel(dict):
    def __init__(self, obs, do_psf=True, normalize_psf=True):
        self.do_psf = do_psf
        self.normalize_psf = normalize_psf
        self["model"] = "template"
        self.npars = 1
        self._set_obs(obs)

    def go(self):
        flags = 0
        xcorr_sum = 0.0
        msq_sum = 0.0
        chi2 = 0.0
        flux = PDEF
        flux_err = CDEF
        for ipass in (1, 2):
            for iobs, obs in enumerate(self.obs):
                im = obs.image
                wt = obs.weight
                if ipass == 1:
                    model = self._get_model(iobs)
                    xcorr_sum += (model * im * wt).sum()
                    msq_sum += (model * model * wt).sum()
                else:
                    model = self._get_model(iobs, flux=flux)
                    chi2 += ((model - im) ** 2 * wt).sum()
            if ipass == 1:
                if msq_sum == 0:
                    break
                flux = xcorr_sum / msq_sum
        dof = self.get_dof()
        chi2per = 9999.0
        if dof > 0:
            chi2per = chi2 / dof
        else:
            flags |= ZERO_DOF
        if msq_sum == 0 or self.totpix == 1:
            flags |= DIV_ZERO
        else:
            arg = chi2 / msq_sum / (self.totpix - 1)
            if arg >= 0.0:
                flux_err = np.sqrt(arg)
            else:
                flags |= BAD_VAR
        self.update({"flags": flags, "chi2per": chi2per, "dof": dof, "flux": flux,
                     "flux_err": flux_err})

    def _get_model(self, iobs, flux=None):
        if self.use_template:
            if flux is not None:
                model = self.template_list[iobs].copy()
                model *= (self.norm_list[iobs] * flux) / model.sum()
            else:
                model = self.template_list[iobs]
            return model
        if flux is None:
            gm = self.gmix_list[iobs]
        else:
            gm = self.gmix_list[iobs].copy()
            gm.set_flux(flux * self.norm_list[iobs])
        obs = self.obs[iobs]
        return gm.make_image(obs.image.shape, jacobian=obs.jacobian)

    def get_dof(self):
        dof = self.get_effective_npix() - self.npars
        if dof <= 0:
            dof = 1.0e-6
        return dof

    def _set_obs(self, obs_in):
        if isinstance(obs_in, Observation):
            obs_list = ObsList()
            obs_list.append(obs_in)
        elif isinstance(obs_in, ObsList):
            obs_list = obs_in
        else:
            raise ValueError("obs should be Observation or ObsList")
        tobs = obs_list[0]
        if self.do_psf:
            tobs = tobs.psf
        if not tobs.has_gmix():
            if not hasattr(tobs, "template"):
                raise ValueError("neither gmix or template image are set")
        self.obs = obs_list
        if tobs.has_gmix():
            self._set_gmix_and_norms()
        else:
            self._set_templates_and_norms()
        self.totpix = sum(obs.pixels.size for obs in self.obs)

    def _set_gmix_and_norms(self):
        self.use_template = False
        self.gmix_list = []
        self.norm_list = []
        for obs in self.obs:
            if self.do_psf:
                gmix = obs.get_psf_gmix()
                if self.normalize_psf:
                    gmix.set_flux(1.0)
            else:
                gmix = obs.get_gmix()
                gmix.set_flux(1.0)
            self.gmix_list.append(gmix)
            self.norm_list.append(gmix.get_flux())

    def _set_templates_and_norms(self):
        self.use_template = True
        self.template_list = []
        self.norm_list = []
        for obs in self.obs:
            if self.do_psf:
                template = obs.psf.template.copy()
                norm = template.sum()
                if self.normalize_psf:
                    template *= 1.0 / norm
                    norm = 1.0
            else:
                template = obs.template.copy()
                template *= 1.0 / template.sum()
                norm = 1.0
            self.template_list.append(template)
            self.norm_list.append(norm)

    def get_effective_npix(self):
        if not hasattr(self, "eff_npix"):
            self.eff_npix = sum(int((obs.weight > 0).sum()) for obs in self.obs)
        return self.eff_npix


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


class PSFFluxBatch(object):
    """
    PSFFluxFitter over a device-resident batch: one exact render of every
    stamp's (flux-normalised) psf mixture, then the zero-lag cross-correlation
    sums of PSFFluxFitModel.go (results.py:700-770) as segmented reductions on
    the device.  Objects may own several stamps (epochs), as an ObsList does.

        res = PSFFluxBatch().go(stamps, psf_gmixes, stamp_obj=None)

    res: dict of (nobj,) arrays flags, flux, flux_err, chi2per, dof
    """

    def __init__(self, normalize_psf=True):
        self.normalize_psf = normalize_psf

    def go(self, stamps, gm, stamp_obj=None):
        import torch
        dev = stamps.device
        ns = stamps.n
        assert gm.n == ns, "one mixture per stamp"
        if stamp_obj is None:
            sobj = np.arange(ns, dtype=np.int64)
        else:
            sobj = np.ascontiguousarray(stamp_obj, dtype=np.int64)
            if sobj.shape != (ns,) or np.any(np.diff(sobj) < 0):
                raise ValueError("stamp_obj must be (nstamps,) and non-decreasing")
        nobj = int(sobj.max()) + 1 if ns else 0
        d_sobj = torch.from_numpy(sobj).to(dev)

        # the unit-flux template and the norm the fitted flux multiplies
        g = gm.clone()
        data = g.data.reshape(ns, g.ngauss, 13)
        psum = data[:, :, 0].sum(dim=1)
        safe = torch.where(psum != 0, psum, torch.ones_like(psum))
        data[:, :, 0] /= safe[:, None]
        data[:, :, 7] = 0.0  # norms are stale (gmix.py set_flux)
        model, status = stamps.render(g, fast_exp=False)
        lengths = torch.from_numpy(stamps.npix).to(dev)
        if not self.normalize_psf:
            # the mixture keeps its own flux: norm_list = gmix.get_flux()
            model = model * torch.repeat_interleave(psum, lengths)
        wt = stamps.ierr * stamps.ierr

        def per_object(x):
            s = torch.segment_reduce(x, "sum", lengths=lengths)
            out = torch.zeros(nobj, dtype=torch.float64, device=dev)
            return out.index_add_(0, d_sobj, s)

        xcorr = per_object(model * stamps.val * wt)
        msq = per_object(model * model * wt)
        zero = msq == 0
        flux = torch.where(zero, torch.full_like(msq, PDEF),
                           xcorr / torch.where(zero, torch.ones_like(msq), msq))
        # second pass: chi2 of the scaled template (linear in the flux)
        fl = torch.repeat_interleave(flux[d_sobj], lengths)
        chi2 = per_object((fl * model - stamps.val) ** 2 * wt)
        chi2 = torch.where(zero, torch.zeros_like(chi2), chi2)

        xcorr, msq, flux, chi2 = (t.cpu().numpy() for t in (xcorr, msq, flux, chi2))
        kept = np.zeros(nobj, dtype=np.int64)
        np.add.at(kept, sobj, stamps.npix_kept.astype(np.int64))
        eff = np.zeros(nobj, dtype=np.int64)
        positive = torch.segment_reduce((stamps.ierr > 0).to(torch.float64), "sum",
                                        lengths=lengths).cpu().numpy()
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
                "dof": dof_used, "status": status.cpu().numpy()}
