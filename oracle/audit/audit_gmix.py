from common import *
import inspect
rng=np.random.RandomState(9)
def gm_pair(pars, model):
    return ref.GMixModel(pars, model), ours.GMixModel(pars, model)
for model,pars in [('gauss',[0.1,-0.2,0.1,0.05,0.7,3.0]),('exp',[0.1,-0.2,0.1,0.05,0.7,3.0]),('dev',[0.,0.,-0.2,0.3,1.7,30.0]),('turb',[0.,0.,-0.2,0.3,1.7,30.0]),
                   ('bdf',[0.1,-0.2,0.1,0.05,0.7,0.4,3.0]),('bd',[0.1,-0.2,0.1,0.05,0.7,0.3,0.4,3.0])]:
    gr,go=gm_pair(pars,model)
    for meth,args in [('get_cen',()),('get_T',()),('get_sigma',()),('get_e1e2T',()),('get_g1g2T',()),('get_e1e2sigma',()),('get_g1g2sigma',()),('get_flux',()),('get_psum',()),('get_full_pars',()),('get_data',()),('copy',()),('__len__',()),('__repr__',()),('get_sheared',(0.02,-0.01)),('make_round',()),('make_round',(True,)),('get_gaussap_flux',()),('get_pars',()),('get_model',())]:
        if not hasattr(gr,meth):
            continue
        kw={}
        if meth=='get_gaussap_flux': kw={'fwhm':1.5}
        run('%s.%s'%(model,meth), getattr(gr,meth), getattr(go,meth), *args, **kw)
    run(model+'.get_sheared(Shape)', lambda: gr.get_sheared(ref.Shape(0.02,0.03)), lambda: go.get_sheared(ours.Shape(0.02,0.03)))
    for meth,args in [('set_cen',(0.3,0.4)),('set_flux',(5.0,)),('set_psum',(7.0,)),('scale_T',(1.3,)),('reset',()),('fill',(np.array(pars)*1.01,))]:
        try: getattr(gr,meth)(*args); er=None
        except Exception as e: er=type(e).__name__
        try: getattr(go,meth)(*args); eo=None
        except Exception as e: eo=type(e).__name__
        if er!=eo: print('DIFF exc',model,meth,er,eo)
        run('%s after %s'%(model,meth), gr.get_data, go.get_data)
    pg=gm_pair([0.,0.,0.01,0.02,0.3,1.0],'gauss')
    run(model+'.convolve', lambda: gr.convolve(pg[0]), lambda: go.convolve(pg[1]))
# GMix full
fp=np.array([1.0,0.1,0.2,0.5,0.1,0.6, 2.0,-0.1,0.0,0.8,-0.05,0.7])
run('GMix(pars)', lambda: ref.GMix(pars=fp), lambda: ours.GMix(pars=fp))
run('GMix(ngauss)', lambda: ref.GMix(ngauss=2), lambda: ours.GMix(ngauss=2))
run('GMix()', lambda: ref.GMix(), lambda: ours.GMix())
run('GMix(bad pars)', lambda: ref.GMix(pars=fp[:5]), lambda: ours.GMix(pars=fp[:5]))
run('GMixModel bad', lambda: ref.GMixModel([0,0,0,0,1],'gauss'), lambda: ours.GMixModel([0,0,0,0,1],'gauss'))
run('GMixModel badmodel', lambda: ref.GMixModel([0,0,0,0,1,1],'blah'), lambda: ours.GMixModel([0,0,0,0,1,1],'blah'))
run('GMixModel g>1', lambda: ref.GMixModel([0,0,0.9,0.9,1,1],'gauss'), lambda: ours.GMixModel([0,0,0.9,0.9,1,1],'gauss'))
run('GMixCoellip', lambda: ref.GMixCoellip([0,0,0.1,0.1,0.5,0.7,1.0,2.0]), lambda: ours.GMixCoellip([0,0,0.1,0.1,0.5,0.7,1.0,2.0]))
run('GMixCoellip bad', lambda: ref.GMixCoellip([0,0,0.1,0.1,0.5,0.7,1.0]), lambda: ours.GMixCoellip([0,0,0.1,0.1,0.5,0.7,1.0]))
run('GMixCM', lambda: ref.gmix.GMixCM(0.3,1.2,[0,0,0.1,0.1,0.5,2.0]), lambda: ours.GMixCM(0.3,1.2,[0,0,0.1,0.1,0.5,2.0]))
run('GMixCM copy', lambda: ref.gmix.GMixCM(0.3,1.2,[0,0,0.1,0.1,0.5,2.0]).copy(), lambda: ours.GMixCM(0.3,1.2,[0,0,0.1,0.1,0.5,2.0]).copy())
gr,go=gm_pair([0.1,-0.2,0.1,0.05,0.7,3.0],'exp')
run('gmix_concat', lambda: ref.gmix.gmix_concat([gr,gr]), lambda: ours.gmix.gmix_concat([go,go]))
for f in ['get_model_num','get_model_name','get_model_ngauss','get_model_npars']:
    if hasattr(ref.gmix,f):
        for a in ['exp','gauss','dev','turb','bdf','bd','cm','coellip','full',1,3,'blah']:
            run('gmix.'+f, getattr(ref.gmix,f), getattr(ours.gmix,f), a)
run('get_coellip_npars', ref.gmix.get_coellip_npars, ours.gmix.get_coellip_npars, 3) if hasattr(ref.gmix,'get_coellip_npars') else None
run('get_coellip_ngauss', ref.gmix.get_coellip_ngauss, ours.gmix.get_coellip_ngauss, 10) if hasattr(ref.gmix,'get_coellip_ngauss') else None
print('ndiff',ndiff[0])
