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
