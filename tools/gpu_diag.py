"""Scratch: per-kernel work counters, retrace counts and kernel times of one workload.  Usage: gpu_diag.py <workload> <spp> [spp_pass]"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
name, spp = sys.argv[1], int(sys.argv[2])
spp_pass = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sd, label = bench.load_workload(name)
g = api.Scene(sd)
for mode in (1, 0):
    g.render(0, 2, traversal=mode)
    film, st = g.render(0, spp, traversal=mode, samples_per_pass=spp_pass)
    film, ct = g.render(0, min(spp, 8), traversal=mode, flags=api.FLAG_COUNT_WORK)
    rays = st['closest_rays'] + st['any_rays']
    units = {'generate': ct['paths'], 'extend': ct['closest_rays'] - ct['paths'], 'shadow': ct['any_rays']}
    print(name, 'mode', mode, 'Mrays/s', round(rays / st['seconds'] / 1e6, 1), 'retraces/ray', st['exact_retraces'] / max(st['closest_rays'], 1),
          {k: round(v, 1) for k, v in st['kernel_ms'].items() if v},
          {k: (round(ct['nodes_fetched'][k] / max(units[k], 1), 2), round(ct['tris_fetched'][k] / max(units[k], 1), 2)) for k in units}, flush=True)
