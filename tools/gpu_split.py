import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
sd, label = bench.load_workload('library-synth')
g = api.Scene(sd)
N = 8
for name, kw, s in [('spp-split share (128 spp, all pixels)', dict(), (0, 128)),
                    ('tile-split share (1024 spp, 1/8 of the blocks)', dict(block=(3, N)), (0, 1024)),
                    ('tile-split share, 256-spp passes', dict(block=(3, N), samples_per_pass=256), (0, 1024)),
                    ('tile-split share, 1024-spp passes', dict(block=(3, N), samples_per_pass=1024), (0, 1024))]:
    g.render(s[0], s[1], traversal=1, **kw)
    for rep in range(2):
        film, st = g.render(s[0], s[1], traversal=1, **kw)
        rays = st['closest_rays'] + st['any_rays']
        print(name, 'sec', round(st['seconds'], 4), 'Mrays/s', round(rays / st['seconds'] / 1e6, 1), 'rays', rays, {k: round(v, 1) for k, v in st['kernel_ms'].items() if v}, flush=True)
