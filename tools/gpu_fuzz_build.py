"""Scratch: the GPU tree builder against the host builder on random meshes, node for node: soups of every size from one
triangle up, duplicated triangles (equal centroids: every sort tie), flattened and collapsed geometry, tessellated interiors.
Usage: gpu_fuzz_build.py <meshes>"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
from pt_hip import api, synth
api.init(0)
bad = 0
for seed in range(int(sys.argv[1])):
    rng = np.random.default_rng(9000 + seed)
    kind = seed % 4
    if kind == 0:
        sd = synth.random_soup(int(rng.integers(1, 12)), seed=seed)
    elif kind == 1:
        sd = synth.random_soup(int(rng.integers(12, 60000)), seed=seed, extent=float(rng.uniform(1, 50)), size=float(rng.uniform(0.02, 3.0)))
    elif kind == 2:
        sd = synth.library_synth(detail=int(rng.integers(1, 5)), seed=seed)
    else:
        sd = synth.bathroom_synth(detail=int(rng.integers(1, 5)), width=32, height=32)
    v, i = np.ascontiguousarray(sd.verts, np.float32).copy(), np.ascontiguousarray(sd.vi, np.uint32).copy()
    twist = int(rng.integers(0, 6))
    if twist == 1:
        i = np.concatenate([i, i, i[: len(i) // 3]])
    elif twist == 2:
        v[:, int(rng.integers(0, 3))] = np.float32(0.5)
    elif twist == 3:
        v = np.round(v * 4) / 4                                   # coarse grid: many equal coordinates and flat triangles
    elif twist == 4:
        k = rng.integers(0, len(i), max(1, len(i) // 5))
        i[k, 2] = i[k, 1]
    elif twist == 5:
        v *= np.float32(1e-6)
    gn, gi, gd = api.build_bvh_gpu(v, i)
    hn, hi, hd = api.build_bvh(v, i)
    same = np.array_equal(gn.view(np.uint32), hn.view(np.uint32)) and np.array_equal(gi, hi) and gd == hd
    print("seed", seed, "kind", kind, "twist", twist, "tris", len(i), "nodes", len(hn), "depth", hd, "identical", same, flush=True)
    bad += not same
print("meshes with differences:", bad)
sys.exit(1 if bad else 0)
