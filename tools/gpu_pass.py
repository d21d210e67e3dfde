"""Scratch: throughput against samples_per_pass."""
import sys, time
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
name = sys.argv[1]
sd, label = bench.load_workload(name)
g = api.Scene(sd)
total = int(sys.argv[2])
for spp_pass in [int(x) for x in sys.argv[3:]]:
    g.render(0, min(total, spp_pass or total), traversal=1, samples_per_pass=spp_pass)
    film, st = g.render(0, total, traversal=1, samples_per_pass=spp_pass)
    rays = st['closest_rays'] + st['any_rays']
    print(name, 'spp_pass', spp_pass, 'sec', round(st['seconds'], 4), 'Mrays/s', round(rays / st['seconds'] / 1e6, 1), {k: round(v, 1) for k, v in st['kernel_ms'].items() if v}, flush=True)
import os
if os.environ.get('PT_COUNT'):
    film, st = g.render(0, min(total, 64), traversal=1, samples_per_pass=64, flags=api.FLAG_COUNT_WORK)
    un = {'generate': st['paths'], 'extend': st['closest_rays'] - st['paths'], 'shadow': st['any_rays']}
    print({k: (round(st['nodes_fetched'][k] / un[k], 2), round(st['nodes_from_lds'][k] / un[k], 2), round(st['tris_fetched'][k] / un[k], 2)) for k in un}, 'retraced', st['exact_retraces'], 'of', st['closest_rays'] + st['any_rays'], flush=True)
