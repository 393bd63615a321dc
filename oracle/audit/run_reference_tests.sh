#!/bin/bash
# The reference's OWN test files (read in place under /root/reference, never copied) run against ngmix_amd:
# `ngmix` is aliased to ngmix_amd by oracle/audit/alias_plugin.py.  Build container only (no GPU: only the test
# files whose bodies are host logic can run; an Observation's pixel list is made by plain numpy here).
#   bash oracle/audit/run_reference_tests.sh > profiles/r06_reference_tests_vs_ngmix_amd.log
HERE=$(cd "$(dirname "$0")" && pwd)
cd /tmp && mkdir -p /tmp/work/alias_run && cd /tmp/work/alias_run
FILES="test_priors_simple test_priors_base test_priors_random test_priors_shear test_priors2d test_joint_priors
test_kde test_gmix_ndim test_shape test_moments test_flags test_gexceptions test_util test_print_pars test_obslist
test_multibandobslist test_observation test_jacobian"
for f in $FILES; do
    echo "== $f"
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$HERE timeout 1500 python -m pytest -p alias_plugin -p no:cacheprovider \
        --rootdir=/tmp/work/alias_run -q --no-header /root/reference/ngmix/tests/$f.py 2>&1 |
        grep "^E  .*Error\|passed\|failed" | sed 's/^E *//' | sort | uniq -c | sort -rn | head -6
done
