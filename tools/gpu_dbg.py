"""Scratch: share of ordered-mode queries that are repeated in reference order."""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
for w in ('library-synth', 'veach-mis', 'bathroom-synth'):
    sd, _ = bench.load_workload(w)
    g = api.Scene(sd)
    _, st = g.render(0, 16, traversal=1)
    print(w, 'closest', st['closest_rays'], 'any', st['any_rays'], 'retraces', st['exact_retraces'], 'share of closest %.4f' % (st['exact_retraces'] / st['closest_rays']))
    g.close()
