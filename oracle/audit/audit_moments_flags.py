from common import *
import inspect
print(inspect.signature(ref.moments.make_mom_result), inspect.signature(ours.moments.make_mom_result))
rng=np.random.RandomState(11)
def rcov(n):
    a=rng.normal(size=(n,n)); return a@a.T*0.01
for n in (6,17):
  for trial in range(40):
    sums=rng.normal(size=n); sums[5]=abs(sums[5])*10+1; sums[4]=abs(sums[4])*5+0.5
    cov=rcov(n)
    kind=trial%8
    if kind==1: sums[5]=-1.0
    if kind==2: sums[4]=-0.5
    if kind==3: cov[5,5]=-1.0
    if kind==4: cov[4,4]=0.0
    if kind==5: sums[5]=0.0
    if kind==6: cov[:]=np.nan
    if kind==7: sums[2]=50.0
    for sums_norm in (None, 2.5):
        run('make_mom_result n=%d kind=%d'%(n,kind), ref.moments.make_mom_result, ours.moments.make_mom_result, sums.copy(), cov.copy(), sums_norm=sums_norm)
for fl in [0,1,2,3,2**4,2**10+2**3,2**20,2**30,2**29+1, 2**25]:
    run('get_flags_str', ref.flags.get_flags_str, ours.flags.get_flags_str, fl)
    run('get_flags_str type', lambda f: type(ref.flags.get_flags_str(f)).__name__, lambda f: type(ours.flags.get_flags_str(f)).__name__, fl)
rn=sorted(n for n in dir(ref.flags) if n.isupper()); 
for n in rn:
    if not hasattr(ours.flags,n): print('flags missing',n)
    elif getattr(ref.flags,n)!=getattr(ours.flags,n): print('flag value differs',n)
for n in ['NAME_MAP'] :
    if hasattr(ref.flags,n): print('NAME_MAP equal', getattr(ref.flags,n)==getattr(ours.flags,n))
# regularize_mom_shapes
print(inspect.signature(ref.moments.regularize_mom_shapes))
for trial in range(10):
    sums=rng.normal(size=6); sums[5]=abs(sums[5])*10+1; sums[4]=abs(sums[4])*5+0.5
    res_r=ref.moments.make_mom_result(sums.copy(), rcov(6)); 
    import copy
    res_o=copy.deepcopy(res_r)
    run('regularize', ref.moments.regularize_mom_shapes, ours.moments.regularize_mom_shapes, res_r, 1.2)
# defaults
for n in sorted(x for x in dir(ref.defaults) if x.isupper()) if hasattr(ref,'defaults') else []:
    if not hasattr(ours.defaults,n): print('defaults missing',n)
    elif getattr(ref.defaults,n)!=getattr(ours.defaults,n): print('defaults differ',n,getattr(ref.defaults,n),getattr(ours.defaults,n))
print('ndiff',ndiff[0])
