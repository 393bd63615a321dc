"""
Side-by-side differential audit of ngmix_amd's HOST logic against the
reference: both packages are imported in one process (the reference under the
numba shim) and the same calls are made on both; any difference in a returned
value (compared to the bit), a raised exception type or a mutated record is
printed as DIFF.  Build container only (it imports /root/reference and has no
GPU: audit_observation.py / audit_em_prep.py substitute the reference's pixel
fill for the HIP one so that the host logic around it can run).  What the audits found in
round 6 is pinned as data in tests/golden/host6.json, result_keys.json,
api_surface.json and api_signatures.json; the last run's output is
profiles/r06_host_audit.log.  TEST INFRASTRUCTURE ONLY.

    cd /tmp/work && for f in /root/repo/oracle/audit/audit_*.py; do
        PYTHONDONTWRITEBYTECODE=1 python $f; done
"""
import sys
sys.dont_write_bytecode=True
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), 'shim'), '/root/reference',
                os.path.dirname(os.path.dirname(HERE))]
import numpy as np
import ngmix as ref
import ngmix_amd as ours
def cmp(a,b):
    if isinstance(a,(tuple,list)): return len(a)==len(b) and all(cmp(x,y) for x,y in zip(a,b))
    if isinstance(a,dict): return set(a)==set(b) and all(cmp(a[k],b[k]) for k in a)
    if isinstance(a,str) or a is None: return a==b
    if hasattr(a,'g1') and hasattr(b,'g1'): return a.g1==b.g1 and a.g2==b.g2
    if hasattr(a,'get_full_pars'): return np.array_equal(a.get_full_pars(),b.get_full_pars(),equal_nan=True)
    if hasattr(a,'_data') and hasattr(b,'_data'):
        return all(np.array_equal(a._data[n],b._data[n],equal_nan=True) for n in a._data.dtype.names)
    if isinstance(a,np.ndarray) and a.dtype.names:
        return all(np.array_equal(a[n],b[n],equal_nan=True) for n in a.dtype.names)
    try:
        return np.array_equal(np.asarray(a,dtype=float),np.asarray(b,dtype=float),equal_nan=True)
    except Exception as e:
        return False
def isexc(r): return isinstance(r,tuple) and len(r)>0 and isinstance(r[0],str) and r[0]=='EXC'
ndiff=[0]
def run(label, fr, fo, *a, **k):
    try: r=fr(*a,**k)
    except Exception as e: r=('EXC',type(e).__name__, str(e)[:80])
    try: o=fo(*a,**k)
    except Exception as e: o=('EXC',type(e).__name__, str(e)[:80])
    if isexc(r): ok = isexc(o) and o[1]==r[1]
    else: ok=cmp(r,o)
    if not ok:
        ndiff[0]+=1; print('DIFF',label,a,k,'\n   ref',r,'\n  ours',o)
    return r,o
