"""Scratch: time to first pixel of the library stand-in at a given tessellation: mesh, tree (GPU or host builder), scene upload, first 1-spp frame."""
import sys, time
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
from pt_hip import api, synth
detail, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
api.init(0)
api.build_bvh_gpu(*[getattr(synth.library_synth(detail=1), k) for k in ("verts", "vi")])
for builder in ("gpu", "host"):
    t0 = time.time(); sd = synth.library_synth(detail=detail, width=w, height=h); t1 = time.time()
    tree = (api.build_bvh_gpu if builder == "gpu" else api.build_bvh)(sd.verts, sd.vi)[:2]; t2 = time.time()
    g = api.Scene(sd, bvh=tree); t3 = time.time()
    film, st = g.render(0, 1, traversal=api.TRAVERSE_FAST); t4 = time.time()
    print(builder, "tris", sd.n_tris, "mesh %.2f tree %.2f scene %.2f first frame %.2f total %.2f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0), flush=True)
    g.close()
