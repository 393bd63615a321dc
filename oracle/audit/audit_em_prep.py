from common import *
ours.observation.make_pixels = lambda image, weight, jacob, ignore_zero_weight=True: ref.pixels.make_pixels(image, weight, jacob, ignore_zero_weight=ignore_zero_weight)  # AUDIT ONLY
rng=np.random.RandomState(4)
for trial in range(6):
    im=rng.normal(size=(8,8))*0.1 + (0.5 if trial%2 else -0.2)
    if trial==4: im[:]=0.0
    if trial==5: im=np.abs(im)+1
    run('prep_image', ref.em.prep_image, ours.em.prep_image, im.copy())
    def f(m):
        o=m.Observation(im.copy(), weight=np.full(im.shape,2.0), jacobian=m.DiagonalJacobian(row=3.5,col=3.5,scale=0.3))
        r=m.em.prep_obs(o)
        return [r[0].image.copy(), r[0].weight.copy(), r[0].pixels.copy(), r[1], r[0].jacobian.get_data()]
    run('prep_obs', lambda: f(ref), lambda: f(ours))
print('ndiff',ndiff[0])
