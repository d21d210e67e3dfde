"""Scratch: long renders of every workload in both modes, checking that the two modes take the same paths (equal ray
counts) and that nothing hangs or faults at sizes the tests do not reach."""
import sys, time
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
import bench
from pt_hip import api
jobs = [('library-synth', 2048, 64), ('veach-mis', 4096, 256), ('bathroom-synth', 2048, 64), ('library-synth@16', 512, 32),
        ('cornell-box-synth', 1024, 64), ('library-synth@8:3840x2160', 64, 8)]
for w, spp, spp_exact in jobs:
    sd, _ = bench.load_workload(w)
    g = api.Scene(sd)
    t = time.time()
    film, st = g.render(0, spp, traversal=1)
    print(w, 'fast', spp, 'spp', round(st['seconds'], 2), 's', round((st['closest_rays'] + st['any_rays']) / st['seconds'] / 1e6), 'Mrays/s',
          'finite', bool(np.isfinite(film).all()), 'retraces', st['exact_retraces'], flush=True)
    f1, s1 = g.render(7, 7 + spp_exact, traversal=1)
    f0, s0 = g.render(7, 7 + spp_exact, traversal=0)
    same = (s0['closest_rays'], s0['any_rays']) == (s1['closest_rays'], s1['any_rays'])
    print('   exact', spp_exact, 'spp', round(s0['seconds'], 2), 's  same ray counts', same, 'films close', bool(np.allclose(f0, f1, rtol=1e-4, atol=1e-5)), flush=True)
    parts = sum(g.render(7, 7 + spp_exact, traversal=1, block=(r, 3))[0] for r in range(3))
    print('   3 block shares add up', bool(np.allclose(parts, f1, rtol=1e-4, atol=1e-5)), flush=True)
    assert same
    g.close()
print('soak ok')
