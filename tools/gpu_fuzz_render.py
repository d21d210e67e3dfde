"""Scratch: the path tracer in both traversal modes on random scenes: the radiance of every sample must be the same bit for
bit (same rays, same hits, same order of additions) with the same ray counts; the films agree up to the order of the
atomic additions at the seams of the film kernel's regions.  Usage: gpu_fuzz_render.py <scenes> <spp> [seed...]"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
from pt_hip import api, synth

seeds, spp = int(sys.argv[1]), int(sys.argv[2])
only = [int(x) for x in sys.argv[3:]]
bad = 0
for seed in (only or range(seeds)):
    rng = np.random.default_rng(5000 + seed)
    kind = seed % 5
    w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
    if kind == 0:
        sd = synth.random_soup(int(rng.integers(2, 20000)), seed=seed, width=w, height=h, extent=float(rng.uniform(1, 20)), size=float(rng.uniform(0.05, 3.0)))
    elif kind == 1:
        sd = synth.library_synth(detail=int(rng.integers(1, 4)), seed=seed, width=w, height=h)
    elif kind == 2:
        sd = synth.bathroom_synth(detail=int(rng.integers(1, 5)), width=w, height=h)
    elif kind == 3:
        sd = synth.cornell_box(width=w, height=h)
    else:
        sd = synth.textured_room(width=w, height=h)
    g = api.Scene(sd)
    sp = int(rng.integers(0, 3)) and int(rng.integers(1, spp + 1))          # samples per pass: automatic or small
    fe, se = g.render(0, spp, traversal=api.TRAVERSE_EXACT, samples_per_pass=sp)
    ff, sf = g.render(0, spp, traversal=api.TRAVERSE_FAST, samples_per_pass=sp)
    film = np.allclose(fe, ff, rtol=1e-5, atol=1e-6)
    counts = (se["closest_rays"], se["any_rays"]) == (sf["closest_rays"], sf["any_rays"])
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.repeat(np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32), spp, 0)
    smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
    le, pe, _ = g.li(pix, smp, traversal=api.TRAVERSE_EXACT)
    lf, pf, _ = g.li(pix, smp, traversal=api.TRAVERSE_FAST)
    same = np.array_equal(le.view(np.uint32), lf.view(np.uint32)) and np.array_equal(pe.view(np.uint32), pf.view(np.uint32))
    print("seed", seed, "kind", kind, sd.name, "tris", sd.n_tris, "%dx%d" % (w, h), "spp/pass", sp, "rays", se["closest_rays"], se["any_rays"],
          "retraces", sf["exact_retraces"], "samples", same, "film", film, "counts", counts, flush=True)
    if not (same and counts and film):
        bad += 1
        d = np.nonzero((le.view(np.uint32) != lf.view(np.uint32)).any(1))[0]
        print("   differing samples:", len(d), pix[d[:4]].tolist(), smp[d[:4]].tolist(), le[d[:4]].tolist(), lf[d[:4]].tolist())
    g.close()
print("scenes with differences:", bad)
sys.exit(1 if bad else 0)
