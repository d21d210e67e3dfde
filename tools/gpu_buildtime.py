"""Scratch: tree build time, host builder against pt_hip_build_bvh, and equality of the two trees.  Usage: gpu_buildtime.py <detail>..."""
import sys, time
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
from pt_hip import api, synth
api.init(0)
api.build_bvh_gpu(*[getattr(synth.library_synth(detail=1), k) for k in ("verts", "vi")])     # warm-up: code objects, first allocation
for detail in [int(x) for x in sys.argv[1:]]:
    sd = synth.library_synth(detail=detail)
    v, i = np.ascontiguousarray(sd.verts, np.float32), np.ascontiguousarray(sd.vi, np.uint32)
    t = time.time(); gn, gi, gd = api.build_bvh_gpu(v, i); tg = time.time() - t
    t = time.time(); hn, hi, hd = api.build_bvh(v, i); th = time.time() - t
    same = np.array_equal(gn.view(np.uint32), hn.view(np.uint32)) and np.array_equal(gi, hi) and gd == hd
    print("detail", detail, "tris", sd.n_tris, "nodes", len(gn), "depth", gd, "gpu %.2f s" % tg, "host %.2f s" % th, "identical", same, flush=True)
