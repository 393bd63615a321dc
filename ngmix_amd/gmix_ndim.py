"""
N-dimensional gaussian mixture as a probability density (reference:
ngmix/gmix_ndim/gmix_ndim.py, gmix_ndim_nb.py): a prior over several fit
parameters at once (size and flux, say), fitted to samples with sklearn's
GaussianMixture or set from weights / means / covariances.  Host numpy: the
density of all points of an array in a few vector operations instead of the
reference's per-point njit call -- the same sums in the same order per point
(chi^2 accumulated over (i, j) in row-major order, the log-sum-exp about the
largest component).

    GMixND(weights, means, covars, rng=...)   or   GMixND(rng=...).fit(data, ngauss)
    get_lnprob_scalar / get_prob_scalar (pars, component=None)
    get_lnprob_array / get_prob_array   (pars (n, ndim) or (n,), component=None)
    sample(n=None)        sklearn's sampler seeded by rng, as the reference
"""
import numpy as np

__all__ = ["GMixND"]


class GMixND(object):

    def __init__(self, weights=None, means=None, covars=None, file=None, rng=None):
        if rng is None:
            rng = np.random.RandomState()
        self.rng = rng
        given = [a is not None for a in (weights, means, covars)]
        if file is not None:
            self.load_mixture(file)
        elif all(given):
            self.set_mixture(weights, means, covars)
        elif any(given):
            raise RuntimeError("send all or none of weights, means, covars")

    def set_mixture(self, weights, means, covars):
        """weights (ngauss,), means (ngauss, ndim) or (ngauss,), covars
        (ngauss, ndim, ndim) or (ngauss,); copied"""
        weights = np.array(weights, dtype="f8", copy=True)
        means = np.array(means, dtype="f8", copy=True)
        covars = np.array(covars, dtype="f8", copy=True)
        if means.ndim == 1:
            means = means.reshape((means.size, 1))
        if covars.ndim == 1:
            covars = covars.reshape((covars.size, 1, 1))
        self.weights, self.means, self.covars = weights, means, covars
        self.ngauss = weights.size
        self.ndim = means.shape[1]
        self._calc_icovars_and_norms()
        self.tmp_lnprob = np.zeros(self.ngauss)
        self.xdiff = np.zeros(self.ndim)

    def _calc_icovars_and_norms(self):
        twopi = 2.0 * np.pi
        norms = np.zeros(self.ngauss)
        icovars = np.zeros((self.ngauss, self.ndim, self.ndim))
        for i in range(self.ngauss):
            cov = self.covars[i, :, :]
            icovars[i, :, :] = np.linalg.inv(cov)
            norms[i] = 1.0 / np.sqrt(twopi ** self.ndim * np.linalg.det(cov))
        self.norms = norms
        self.pnorms = norms * self.weights
        self.log_pnorms = np.log(self.pnorms)
        self.icovars = icovars

    @property
    def converged(self):
        return self._gmm.converged_

    def fit(self, data, ngauss, n_iter=5000, min_covar=1.0e-6, doplot=False, **keys):
        """fit ngauss gaussians to data (n,) or (n, ndim) with sklearn"""
        from sklearn.mixture import GaussianMixture
        if data.ndim == 1:
            data = data[:, np.newaxis]
        print("ngauss:   ", ngauss)
        print("n_iter:   ", n_iter)
        print("min_covar:", min_covar)
        gmm = GaussianMixture(n_components=ngauss, max_iter=n_iter, reg_covar=min_covar,
                              covariance_type="full", random_state=self.rng)
        gmm.fit(data)
        if not gmm.converged_:
            print("DID NOT CONVERGE")
        self._gmm = gmm
        self.set_mixture(gmm.weights_, gmm.means_, gmm.covariances_)
        if doplot:
            return self.plot(data=data, **keys)

    def plot(self, *args, **kw):  # pragma: no cover
        raise NotImplementedError("plotting needs matplotlib and esutil, which this build "
                                  "does not depend on")

    def save_mixture(self, fname):
        import fitsio
        print("writing gaussian mixture to :", fname)
        with fitsio.FITS(fname, "rw", clobber=True) as fits:
            fits.write(self.weights, extname="weights")
            fits.write(self.means, extname="means")
            fits.write(self.covars, extname="covars")

    def load_mixture(self, fname):
        import fitsio
        print("loading gaussian mixture from:", fname)
        with fitsio.FITS(fname) as fits:
            weights = fits["weights"].read()
            means = fits["means"].read()
            covars = fits["covars"].read()
        self.set_mixture(weights, means, covars)

    # ------------------------------------------------------------------
    def _component_lnprob(self, pars):
        """(n, ngauss) ln of weight * density of every component at pars (n,
        ndim): chi^2 summed over (i, j) in row-major order as the reference's
        loops do"""
        diff = pars[:, None, :] - self.means[None, :, :]
        chi2 = np.zeros(diff.shape[:2])
        for a in range(self.ndim):
            for b in range(self.ndim):
                chi2 += diff[:, :, a] * diff[:, :, b] * self.icovars[None, :, a, b]
        return -0.5 * chi2 + self.log_pnorms[None, :]

    def _evaluate(self, pars, dolog, component):
        lnp = self._component_lnprob(pars)
        if component is not None:
            assert 0 <= component < self.ngauss
            one = lnp[:, component]
            return one if dolog else np.exp(one)
        # (the reference's running maximum starts at -9.99e9)
        top = np.maximum(lnp.max(axis=1), -9.99e9)
        p = np.zeros(lnp.shape[0])
        for i in range(self.ngauss):
            p += np.exp(lnp[:, i] - top)
        return np.log(p) + top if dolog else p * np.exp(top)

    def _scalar(self, pars_in, dolog, component):
        pars = np.array(pars_in, dtype="f8", ndmin=1, order="C")
        return self._evaluate(pars[None, :self.ndim], dolog, component)[0]

    def _array(self, pars, dolog, component):
        pars = np.array(pars, dtype="f8", ndmin=1, order="C")
        if pars.ndim == 1:
            pars = pars[:, np.newaxis]
        return self._evaluate(pars[:, :self.ndim], dolog, component)

    def get_lnprob_scalar(self, pars_in, component=None):
        return self._scalar(pars_in, 1, component)

    def get_prob_scalar(self, pars_in, component=None):
        return self._scalar(pars_in, 0, component)

    def get_lnprob_array(self, pars, component=None):
        return self._array(pars, 1, component)

    def get_prob_array(self, pars, component=None):
        return self._array(pars, 0, component)

    # ------------------------------------------------------------------
    def sample(self, n=None):
        if not hasattr(self, "_gmm"):
            self._set_gmm()
        one = n is None
        samples, _ = self._gmm.sample(1 if one else n)
        if self.ndim == 1:
            samples = samples[:, 0]
        return samples[0] if one else samples

    def _make_gmm(self, ngauss):
        from sklearn.mixture import GaussianMixture
        return GaussianMixture(n_components=ngauss, max_iter=10000, reg_covar=1.0e-12,
                               covariance_type="full", random_state=self.rng)

    def _set_gmm(self):
        """an sklearn mixture with this mixture's numbers set by hand"""
        from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky
        gmm = self._make_gmm(self.weights.size)
        gmm.means_ = self.means.copy()
        gmm.covariances_ = self.covars.copy()
        gmm.weights_ = self.weights.copy()
        gmm.precisions_cholesky_ = _compute_precision_cholesky(self.covars, "full")
        self._gmm = gmm
