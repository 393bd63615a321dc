"""pytest plugin (build container only): the reference's OWN host-side test files run against ngmix_amd.
`ngmix` is aliased to ngmix_amd; `ngmix.tests` resolves to the reference's test directory, read in place."""
import sys
import types
import importlib
import pkgutil

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/repo/oracle/shim")      # numba / galsim stand-ins for the test helpers' imports
sys.path.insert(0, "/root/repo")
import numpy as np
import ngmix_amd

sys.modules["ngmix"] = ngmix_amd
for m in pkgutil.iter_modules(ngmix_amd.__path__):
    try:
        sub = importlib.import_module("ngmix_amd." + m.name)
    except Exception as e:      # noqa: BLE001
        continue
    sys.modules["ngmix." + m.name] = sub
tests = types.ModuleType("ngmix.tests")
tests.__path__ = ["/root/reference/ngmix/tests"]
tests.__package__ = "ngmix.tests"
sys.modules["ngmix.tests"] = tests
ngmix_amd.tests = tests


# no GPU in this container: the pixel list of an Observation by plain numpy (the HIP fill is held to the
# reference's to the bit by tests/test_gpu_pixpass.py on the GPU box)
def _numpy_make_pixels(image, weight, jacob, ignore_zero_weight=True):
    from ngmix_amd import _lib
    j = jacob._data[0] if hasattr(jacob, "_data") else jacob
    nrow, ncol = image.shape
    rows, cols = np.mgrid[0:nrow, 0:ncol]
    keep = (weight > 0) if ignore_zero_weight else np.ones_like(weight, dtype=bool)
    if ignore_zero_weight and not keep.any():
        raise ngmix_amd.GMixFatalError("no weights > 0")
    r, c = rows[keep].astype("f8"), cols[keep].astype("f8")
    out = np.zeros(r.size, dtype=_lib.PIXEL_DTYPE)
    out["v"] = j["dvdrow"] * (r - j["row0"]) + j["dvdcol"] * (c - j["col0"])
    out["u"] = j["dudrow"] * (r - j["row0"]) + j["dudcol"] * (c - j["col0"])
    out["area"] = j["scale"] ** 2
    out["val"] = image[keep]
    w = weight[keep]
    out["ierr"] = np.sqrt(np.where(w > 0, w, 0.0))
    return out


import ngmix_amd.observation as _obs
import ngmix_amd.pixels as _pix
_obs.make_pixels = _numpy_make_pixels

# the reference's conftest imports two modules outside the hot path for fixtures these test files do not use
for name in ("ngmix.prepsfmom", "ngmix.metacal", "ngmix.metacal.metacal"):
    stub = types.ModuleType(name)
    for f in ("turn_on_fft_caching", "turn_off_fft_caching", "turn_on_kernel_caching",
              "turn_off_kernel_caching", "turn_on_galsim_caching", "turn_off_galsim_caching"):
        setattr(stub, f, lambda: None)
    sys.modules[name] = stub
sys.modules["ngmix.metacal"].metacal = sys.modules["ngmix.metacal.metacal"]
ngmix_amd.prepsfmom = sys.modules["ngmix.prepsfmom"]
ngmix_amd.metacal = sys.modules["ngmix.metacal"]
_pix.make_pixels = _numpy_make_pixels
