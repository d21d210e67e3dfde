"""Scratch: does a GPU render faster when two or three handles (own streams, own work buffers) share it?
Usage: python tools/gpu_overlap.py <workload> <spp> <spp per pass> <handles>..."""
import sys, time
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
name, total, spp_pass = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sd, label = bench.load_workload(name)
first = api.Scene(sd)
scenes = [first]
for n in [int(x) for x in sys.argv[4:]]:
    while len(scenes) < n:
        scenes.append(api.Scene(sd, bvh=(first.host.nodes, first.host.prim_ids)))
    use = scenes[:n]
    for rep in range(2):
        t0 = time.time()
        if n == 1:
            film, st = use[0].render(0, total, traversal=1, samples_per_pass=spp_pass)
        else:
            film, st = api.render_multi(use, 0, total, traversal=1, samples_per_pass=spp_pass)
        dt = time.time() - t0
    rays = st['closest_rays'] + st['any_rays']
    print(name, 'handles', n, 'spp_pass', spp_pass, 'wall', round(dt, 4), 'stats sec', round(st['seconds'], 4), 'Mrays/s (wall)', round(rays / dt / 1e6, 1), flush=True)
