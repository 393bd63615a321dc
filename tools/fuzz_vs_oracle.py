"""A fuzz campaign of the GPU kernels against the CPU oracle: the bodies of the
suite's own oracle comparisons (tests/test_gpu_pixpass.py::
test_batch_random_vs_oracle -- loglike / fill_fdiff / render, exact and fused,
sheared jacobians, masks, both pixel-list modes --, tests/test_gpu_iter.py::
test_admom_kernel_variants_vs_oracle, ::test_em_kernel_variants_vs_oracle and
::test_em_many_gaussians_vs_oracle -- exact numiter / flags, every EM kind) over RANDOM stamp shapes and mixture sizes instead of
the suite's dozen fixed ones.  Every case derives its own data seed from its
shape, so a failure is reproduced by its printed (shape, ngauss).

usage: python tools/fuzz_vs_oracle.py [seconds] [seed]   (default 120 s)
A log of a run is kept under profiles/."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_pixpass as tp  # noqa: E402
import test_gpu_iter as ti  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 20261003)


def dim():
    """stamp sides: mostly what surveys cut (multiples of 8 / 16), some odd,
    some tiny, a few beyond one wave's register slots"""
    u = rng.uniform()
    if u < 0.45:
        return int(rng.choice([8, 16, 24, 32, 40, 48, 56, 64, 72, 96]))
    if u < 0.85:
        # (>= 3: the suite's comparison wants a listed pixel in every stamp, and
        # with 3 % of the weights zeroed a stamp of one or two pixels may have none)
        return int(rng.randint(3, 100))
    return int(rng.randint(100, 160))


t0 = time.time()
counts = {"pixpass": 0, "admom": 0, "em": 0, "em_many": 0}
failures = []
while time.time() - t0 < budget:
    u = rng.uniform()
    try:
        if u < 0.6:
            dims = (dim(), dim())
            ng = int(rng.choice([1, 2, 3, 4, 6, 8, 10, 16, 24, 33, 40, 48]))
            exact = bool(rng.randint(2))
            case = ("pixpass", dims, ng, exact)
            tp.test_batch_random_vs_oracle(dims, ng, exact)
            counts["pixpass"] += 1
        elif u < 0.8:
            # (the iterative cases draw objects that fit the stamp: sides >= 16)
            shape = (max(dim(), 16), max(dim(), 16))
            shape = (min(shape[0], 96), min(shape[1], 96))
            case = ("admom", shape)
            ti.test_admom_kernel_variants_vs_oracle(shape)
            counts["admom"] += 1
        elif u < 0.9:
            shape = (min(max(dim(), 20), 64), min(max(dim(), 20), 64))
            ng = int(rng.randint(1, 3))
            case = ("em", shape, ng)
            ti.test_em_kernel_variants_vs_oracle(shape, ng)
            counts["em"] += 1
        else:
            # every run kind, up to nine object gaussians, 1- and 3-gaussian psfs
            # (with the comparison's own assertion of which kernel served it)
            shape = (min(max(dim(), 20), 80), min(max(dim(), 20), 80))
            ng, npsf, kind = int(rng.randint(1, 10)), int(rng.choice([1, 3])), int(rng.randint(4))
            case = ("em_many", shape, ng, npsf, kind)
            ti.test_em_many_gaussians_vs_oracle(shape, ng, npsf, kind)
            counts["em_many"] += 1
    except Exception:
        failures.append((case, traceback.format_exc(limit=3)))
        print("FAIL", case)
        print(failures[-1][1])
        sys.stdout.flush()
        if len(failures) >= 10:
            break
print("fuzz_vs_oracle: %.0f s, cases run %s, failures %d" % (time.time() - t0, counts,
                                                             len(failures)))
for case, tb in failures:
    print("  ", case)
sys.exit(1 if failures else 0)
