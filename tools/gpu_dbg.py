"""Scratch: work counters of a variant build."""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
for w in ('library-synth', 'veach-mis'):
    sd, _ = bench.load_workload(w)
    g = api.Scene(sd)
    _, st = g.render(0, 8, traversal=1, flags=api.FLAG_COUNT_WORK)
    rays = {'generate': st['paths'], 'extend': st['closest_rays'] - st['paths'], 'shadow': st['any_rays']}
    for k in rays:
        print(w, k, 'nodes/ray %.2f' % (st['nodes_fetched'][k] / rays[k]), 'tris-or-dead/ray %.2f' % (st['tris_fetched'][k] / rays[k]))
    g.close()
