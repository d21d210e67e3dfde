"""Scratch: image split debugging."""
import sys
sys.path[:0] = ['path-tracer_amd/python', 'oracle', 'tests']
import numpy as np
from pt_hip import api, synth
sd = synth.cornell_box(200, 136)
g = api.Scene(sd)
for spp_pass in (0, 4):
    whole, wst = g.render(0, 6, traversal=1, samples_per_pass=spp_pass)
    print('whole', spp_pass, wst['paths'], wst['closest_rays'], wst['any_rays'])
for mode in (0, 1):
    for world in (2,):
        total = np.zeros_like(whole)
        for rank in range(world):
            film, st = g.render(0, 6, traversal=mode, block=(rank, world), samples_per_pass=4)
            print(' mode', mode, 'rank', rank, st['paths'], st['closest_rays'], st['any_rays'])
            total += film
        d = np.abs(total - whole)
        print(' max diff', d.max(), 'n diff', (d > 1e-4).sum(), 'weight diff', np.abs(total[..., 3] - whole[..., 3]).max())
        ys, xs = np.nonzero(d.max(-1) > 1e-4)
        print(' where', list(zip(xs[:10], ys[:10])))
a0, st = g.render(0, 6, traversal=1, integrator=api.INTEGRATOR_ALBEDO)
t = np.zeros_like(a0)
for rank in range(3):
    f, st = g.render(0, 6, traversal=1, integrator=api.INTEGRATOR_ALBEDO, block=(rank, 3))
    t += f
print('albedo diff', np.abs(t - a0).max())
