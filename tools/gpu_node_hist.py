"""Scratch: share of the navigation-node visits that the first T nodes take (library built with -DPT_NODE_HIST, see
tools/build_variant.sh).  Usage: PT_HIP_LIB_DIR=build/var_<name> python tools/gpu_node_hist.py <workload> <spp>"""
import ctypes, sys
import numpy as np
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
name, spp = sys.argv[1], int(sys.argv[2])
sd, label = bench.load_workload(name)
g = api.Scene(sd)
lib = api.hip_lib()
N = 1 << 20
buf = np.zeros(N, np.uint32)
ptr = buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
lib.pt_hip_debug_node_hist(ptr, 1)
film, st = g.render(0, spp, traversal=1)
assert lib.pt_hip_debug_node_hist(ptr, 1) == 0
h = buf.astype(np.float64)
tot = h.sum()
rays = st['closest_rays'] + st['any_rays']
print(name, 'rays', rays, 'node visits', int(tot), 'per ray', round(tot / rays, 2), 'nodes visited at all', int((h > 0).sum()))
c = np.cumsum(h) / tot
best = np.cumsum(np.sort(h)[::-1]) / tot
for T in (1, 5, 21, 32, 48, 64, 85, 128, 213, 256, 341, 512, 1024, 4096, 16384, 65536):
    print(f'  first {T:6d} nodes in builder order: {c[T - 1]:.3f}   best possible {T:6d}: {best[T - 1]:.3f}', flush=True)
