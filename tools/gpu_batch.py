"""Scratch: throughput of the batched queries (pt_hip_intersect_device / pt_hip_occluded_device) on random rays inside a
scene, both traversal modes.  Usage: gpu_batch.py <workload> <million rays>"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.', 'tests']
import numpy as np
import bench, helpers
from pt_hip import api
sd, label = bench.load_workload(sys.argv[1])
n = int(float(sys.argv[2]) * 1e6)
g = api.Scene(sd)
rng = np.random.default_rng(3)
lo, hi = sd.verts.min(0), sd.verts.max(0)
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays[:, 3:6] = d
rays[:, 7] = np.inf
shadow = rays.copy(); shadow[:, 7] = rng.uniform(0.1, 1.0, n) * float(np.linalg.norm(hi - lo)) * 0.3
d_rays, d_sh = helpers.DeviceBuffer.from_array(rays), helpers.DeviceBuffer.from_array(shadow)
d_hits, d_occ = helpers.DeviceBuffer(16 * n), helpers.DeviceBuffer(n)
for mode, name in ((api.TRAVERSE_FAST, "fast"), (api.TRAVERSE_EXACT, "exact")):
    for rep in range(3):
        ms_c = g.intersect_device(d_rays.ptr, n, d_hits.ptr, mode)
        ms_a = g.occluded_device(d_sh.ptr, n, d_occ.ptr, mode)
    hit = (d_hits.to_array(np.int32, (n, 4))[:, 0] >= 0).mean()
    occ = d_occ.to_array(np.uint8, (n,)).mean()
    print(sys.argv[1], name, "closest %.1f Mrays/s (hit rate %.2f)  any %.1f Mrays/s (occluded %.2f)" % (n / ms_c / 1e3, hit, n / ms_a / 1e3, occ), flush=True)
