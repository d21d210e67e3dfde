"""Scratch: lane participation per phase of the wave-refill loops (library built with -DPT_PHASE_STATS, see
tools/build_variant.sh).  Usage: PT_HIP_LIB_DIR=build/var_<name> python tools/gpu_phase.py <workload> <spp>"""
import ctypes, os, sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import bench
from pt_hip import api
name, spp = sys.argv[1], int(sys.argv[2])
sd, label = bench.load_workload(name)
g = api.Scene(sd)
lib = api.hip_lib()
buf = (ctypes.c_ulonglong * 32)()
g.render(0, spp, traversal=1)
lib.pt_hip_debug_phase(buf, 1)
film, st = g.render(0, spp, traversal=1)
lib.pt_hip_debug_phase(buf, 1)
for cls, nm in enumerate(['generate', 'extend', 'shadow']):
    c = [buf[cls * 8 + i] for i in range(8)]
    if not c[0]:
        continue
    print(name, nm, 'ms', round(st['kernel_ms'][nm], 1),
          'box steps', c[0], 'lanes/step', round(c[1] / c[0], 1),
          '| leaf phases', c[2], 'lanes/phase', round(c[3] / max(c[2], 1), 1),
          '| tri rounds', c[4], 'lanes/round', round(c[5] / max(c[4], 1), 1), 'rounds/phase', round(c[4] / max(c[2], 1), 2),
          '| commit rounds', c[6], 'lanes/round', round(c[7] / max(c[6], 1), 1),
          '| box steps per leaf phase', round(c[0] / max(c[2], 1), 2), flush=True)
