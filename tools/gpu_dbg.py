"""Scratch: GPU render of veach-mis against the reference's result.png (linear block means), by brightness band."""
import sys
sys.path[:0] = ['path-tracer_amd/python', 'oracle', 'tests']
import numpy as np
import helpers
from pt_hip import api
from pt_hip.scene import SceneData
want = np.load("tests/golden/veach_mis_result.npz")["linear"].astype(np.float64)
g = api.Scene(SceneData.load('tests/golden/veach_mis_scene.npz'))
bands = [0, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.01]
conv = helpers.png_block_means(api.film_to_image(g.render(5000, 7048, traversal=1)[0]))   # selector: converged, other samples
lum = conv.mean(-1)
for spp in (16, 256, 2048):
    film, _ = g.render(0, spp, traversal=1)
    got = helpers.png_block_means(api.film_to_image(film))
    print('spp', spp, 'sum ratio', got.sum() / want.sum(), 'mean |d| / mean', np.abs(got - want).mean() / want.mean())
    for lo, hi in zip(bands, bands[1:]):
        m = (lum >= lo) & (lum < hi)
        if m.any():
            print('   band %.3f-%.3f frac %.3f ratio got/ref %.4f   ref/conv %.4f' % (lo, hi, m.mean(), got[m].mean() / want[m].mean(), want[m].mean() / conv[m].mean()))
