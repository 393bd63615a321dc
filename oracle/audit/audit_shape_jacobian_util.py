from common import *
import inspect
for g in [(0.2,-0.3),(0.0,0.0),(0.7,0.6)]:
    sr=ref.Shape(*g); so=ours.Shape(*g)
    for meth,args in [('get_sheared',(0.02,-0.03)),('get_rotated',(0.3,)),('copy',()),('__neg__',()),('__repr__',())]:
        if hasattr(sr,meth): run('Shape.'+meth, getattr(sr,meth), getattr(so,meth), *args)
    run('Shape.get_sheared(shape)', lambda: sr.get_sheared(ref.Shape(0.1,0.05)), lambda: so.get_sheared(ours.Shape(0.1,0.05)))
    for meth,args in [('set_g1g2',(0.1,0.2)),('rotate',(0.2,))]:
        getattr(sr,meth)(*args); getattr(so,meth)(*args)
        if not (sr.g1==so.g1 and sr.g2==so.g2): print('DIFF Shape.'+meth)
run('Shape(1,1)', lambda: ref.Shape(1.0,1.0), lambda: ours.Shape(1.0,1.0))
kw=dict(row=10.3,col=11.1,dvdrow=0.25,dvdcol=0.01,dudrow=-0.02,dudcol=0.27)
jr=ref.Jacobian(**kw); jo=ours.Jacobian(**kw)
for meth,args in [('get_cen',()),('get_vu',(3.0,4.5)),('get_rowcol',(0.3,-0.2)),('get_det',()),('get_scale',()),('get_area',()) ,('get_row0',()),('get_col0',()),('get_dvdrow',()),('get_dudcol',()),('get_dudrow',()),('get_dvdcol',()),('copy',()),('__repr__',()),('get_vu',(np.arange(3.0),np.arange(3.0)+1)),('get_rowcol',(np.arange(3.0),np.arange(3.0)+1)),('__call__',(1.0,2.0)),('get_data',())]:
    if not hasattr(jr,meth): print('ref lacks',meth); continue
    run('Jacobian.'+meth, getattr(jr,meth), getattr(jo,meth), *args)
for p in ['row0','col0','dvdrow','dvdcol','dudrow','dudcol','det','scale','area','cen']:
    if hasattr(jr,p): run('Jacobian.'+p, lambda: getattr(jr,p), lambda: getattr(jo,p))
jr.set_cen(row=5.0,col=6.0); jo.set_cen(row=5.0,col=6.0)
run('jac data', lambda: jr._data, lambda: jo._data)
run('Jacobian(x,y)', lambda: ref.Jacobian(x=1.0,y=2.0,dudx=.2,dudy=0.01,dvdx=0.02,dvdy=.2)._data, lambda: ours.Jacobian(x=1.0,y=2.0,dudx=.2,dudy=0.01,dvdx=0.02,dvdy=.2)._data)
run('Jacobian missing', lambda: ref.Jacobian(row=1.0), lambda: ours.Jacobian(row=1.0))
run('Diag', lambda: ref.DiagonalJacobian(row=1.,col=2.,scale=.3)._data, lambda: ours.DiagonalJacobian(row=1.,col=2.,scale=.3)._data)
run('Diag xy', lambda: ref.DiagonalJacobian(x=1.,y=2.,scale=.3)._data, lambda: ours.DiagonalJacobian(x=1.,y=2.,scale=.3)._data)
run('Unit', lambda: ref.UnitJacobian(row=1.,col=2.)._data, lambda: ours.UnitJacobian(row=1.,col=2.)._data)
run('srandu', lambda: ref.srandu(5, rng=np.random.RandomState(3)), lambda: ours.srandu(5, rng=np.random.RandomState(3)))
run('srandu()', lambda: ref.srandu(rng=np.random.RandomState(3)), lambda: ours.srandu(rng=np.random.RandomState(3)))
for n in ['get_ratio_error','get_ratio_var']:
    run('util.'+n, getattr(ref.util,n), getattr(ours.util,n), 1.0,2.0,0.1,0.2,0.01)
    run('util.'+n, getattr(ref.util,n), getattr(ours.util,n), 1.0,0.0,0.1,0.2,0.01)
run('format_pars', ref.util.format_pars, ours.util.format_pars, np.array([1.0,2.5e-8,3e10]))
run('format_pars', ref.util.format_pars, ours.util.format_pars, np.array([1.0,2.5e-8,3e10]), fmt='%.3f')
run('get_sheared_g1g2T', ref.moments.get_sheared_g1g2T, ours.moments.get_sheared_g1g2T, 0.1,0.2,0.9,0.02,-0.03)
print(inspect.signature(ref.moments.regularize_mom_shapes))
print('ndiff',ndiff[0])
