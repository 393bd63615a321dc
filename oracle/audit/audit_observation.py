from common import *
import copy
ours.observation.make_pixels = lambda image, weight, jacob, ignore_zero_weight=True: ref.pixels.make_pixels(image, weight, jacob, ignore_zero_weight=ignore_zero_weight)  # AUDIT ONLY: no GPU here
rng=np.random.RandomState(2)
im=rng.normal(size=(7,9)); wt=np.abs(rng.normal(size=(7,9)))+0.1; wt[2,3]=0.0; wt[4,4]=-1.0
def mk(mod, **kw):
    return mod.Observation(im.copy(), **kw)
def both(label, f):
    return run(label, lambda: f(ref), lambda: f(ours))
def data_of(o):
    d={'image':o.image.copy(),'weight':o.weight.copy(),'pixels':o.pixels.copy() if o.pixels is not None else None,
       'jac':o.jacobian.get_data().copy(),'meta':dict(o.meta)}
    for n in ['bmask','ormask','noise','mfrac']:
        d[n]= getattr(o,n).copy() if getattr(o,'has_'+n)() else None
    return d
both('default', lambda m: data_of(mk(m)))
both('weight', lambda m: data_of(mk(m, weight=wt.copy())))
both('weight noignore', lambda m: data_of(mk(m, weight=np.abs(wt), ignore_zero_weight=False)))
both('jac', lambda m: data_of(mk(m, weight=wt.copy(), jacobian=m.Jacobian(row=3.1,col=4.2,dvdrow=0.2,dvdcol=0.01,dudrow=-0.02,dudcol=0.25))))
both('bmask etc', lambda m: data_of(mk(m, weight=wt.copy(), bmask=np.ones((7,9),dtype='i4'), ormask=np.zeros((7,9),dtype='i4'), noise=im*2, mfrac=np.abs(im)*0.1, meta={'a':1})))
both('bad weight shape', lambda m: mk(m, weight=wt[:5]))
both('bad bmask shape', lambda m: mk(m, bmask=np.ones((3,3),dtype='i4')))
both('bad noise shape', lambda m: mk(m, noise=np.ones((3,3))))
both('bad mfrac shape', lambda m: mk(m, mfrac=np.ones((3,3))))
both('1d image', lambda m: m.Observation(np.zeros(5)))
both('int image', lambda m: data_of(m.Observation(np.arange(12).reshape(3,4))))
both('all zero weight', lambda m: mk(m, weight=np.zeros((7,9))))
both('all zero weight noignore', lambda m: data_of(mk(m, weight=np.zeros((7,9)), ignore_zero_weight=False)))
both('bad jacobian', lambda m: mk(m, jacobian=3))
both('bad meta', lambda m: mk(m, meta=3))
both('bad psf', lambda m: mk(m, psf=3))
both('bad gmix', lambda m: mk(m, gmix=3))
both('store_pixels False', lambda m: mk(m, store_pixels=False).pixels)
def setters(m):
    o=mk(m, weight=wt.copy())
    out=[]
    o.set_image(im*2); out.append(data_of(o))
    o.set_weight(wt*3); out.append(data_of(o))
    o.set_jacobian(m.DiagonalJacobian(row=1.,col=2.,scale=0.3)); out.append(data_of(o))
    o.set_image(im*3, update_pixels=False); out.append(data_of(o))
    o.update_pixels(); out.append(data_of(o))
    o.set_meta({'b':2}); o.update_meta_data({'c':3}); out.append(data_of(o))
    o.image = im*4; out.append(data_of(o))
    o.weight = wt*5; out.append(data_of(o))
    o.jacobian = m.UnitJacobian(row=0.,col=0.); out.append(data_of(o))
    o.bmask = np.ones((7,9),dtype='i4'); o.ormask=np.ones((7,9),dtype='i4'); o.noise=im; o.mfrac=np.abs(im); out.append(data_of(o))
    o.bmask=None; o.ormask=None; o.noise=None; o.mfrac=None; out.append(data_of(o))
    o.meta={'z':1}; out.append(data_of(o))
    return out
both('setters', setters)
def writeable(m):
    o=mk(m, weight=wt.copy())
    out=[]
    try:
        o.image[0,0]=5.0; out.append('wrote')
    except Exception as e: out.append(type(e).__name__)
    try:
        o.pixels['val'][0]=3; out.append('wrote pixels')
    except Exception as e: out.append(type(e).__name__)
    with o.writeable():
        o.image[0,0]=5.0
        o.weight[0,1]=0.0
    out.append(data_of(o))
    try:
        o.weight[0,0]=1.0; out.append('wrote')
    except Exception as e: out.append(type(e).__name__)
    return out
both('writeable', writeable)
def copies(m):
    p=m.Observation(im.copy(), gmix=m.GMixModel([0,0,0,0,1,1],'gauss'))
    o=mk(m, weight=wt.copy(), psf=p, meta={'a':[1,2]})
    c=o.copy(); d=copy.deepcopy(o); e=copy.copy(o)
    return [data_of(c),data_of(d),data_of(e),c.psf.gmix.get_full_pars(), c.has_psf(), c.has_psf_gmix(), o.get_psf_gmix().get_full_pars(), c.meta is o.meta]
both('copies', copies)
def s2n(m):
    o=mk(m, weight=wt.copy())
    return [o.get_s2n(), o.get_s2n_sums()]
both('s2n', s2n)
def accessors(m):
    o=mk(m, weight=wt.copy())
    out=[o.has_gmix(), o.has_psf(), o.has_psf_gmix(), o.has_bmask()]
    for f in ['get_gmix','get_psf','get_psf_gmix']:
        try: getattr(o,f)(); out.append('ok')
        except Exception as e: out.append(type(e).__name__)
    for attr in ['gmix','psf']:
        try: getattr(o,attr); out.append('ok')
        except Exception as e: out.append(type(e).__name__)
    o.set_gmix(m.GMixModel([0,0,0,0,1,1],'gauss')); out.append(o.gmix.get_full_pars())
    o.set_gmix(None); out.append(o.has_gmix())
    o.set_psf(None); out.append(o.has_psf())
    return out
both('accessors', accessors)
def lists(m):
    out=[]
    ol=m.ObsList(meta={'q':1}); ol.append(mk(m)); ol.append(mk(m,weight=wt.copy()))
    out += [len(ol), dict(ol.meta), ol.get_s2n(), ol.get_s2n_sums()]
    try: ol.append(3); out.append('ok')
    except Exception as e: out.append(type(e).__name__)
    try: ol[0]=3; out.append('ok')
    except Exception as e: out.append(type(e).__name__)
    mb=m.MultiBandObsList(meta={'r':2}); mb.append(ol); mb.append(ol)
    out += [len(mb), dict(mb.meta), mb.get_s2n(), mb.get_s2n_sums()]
    try: mb.append(mk(m)); out.append('ok')
    except Exception as e: out.append(type(e).__name__)
    ol.set_meta({'x':1}); ol.update_meta_data({'y':2}); out.append(dict(ol.meta))
    try: ol.set_meta(3)
    except Exception as e: out.append(type(e).__name__)
    for x in [mk(m), ol, mb, 3]:
        try:
            r=m.observation.get_mb_obs(x); out.append([len(r), len(r[0])])
        except Exception as e: out.append(type(e).__name__)
    c=copy.deepcopy(mb); out.append([len(c),len(c[0]),dict(c.meta)])
    return out
both('lists', lists)
print('ndiff',ndiff[0])
