"""Scratch: ordered against reference-order traversal on random scenes and rays, bit for bit (hits, t, u, v, occlusion).
Scenes: triangle soups of several sizes and triangle scales (leaves of one to eight triangles, fat leaves of every mix),
tessellated boxes, the interior stand-in.  Rays: between random points of the scene bounds (both unbounded and ending
near surfaces), axis-parallel ones, and rays starting on triangles.  Usage: gpu_fuzz.py <seeds> <rays per scene>"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.']
import numpy as np
from pt_hip import api, synth

import os
GRAZE = os.environ.get("FUZZ_GRAZE") == "1"
seeds, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
only = [int(x) for x in sys.argv[3:]]
for seed in (only or range(seeds)):
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 4
    if kind == 0:
        sd = synth.random_soup(int(rng.integers(2, 20000)), seed=seed, extent=float(rng.uniform(1, 50)), size=float(rng.uniform(0.02, 3.0)))
    elif kind == 1:
        sd = synth.random_soup(int(rng.integers(2, 300)), seed=seed, extent=4.0, size=float(rng.uniform(1.0, 8.0)))   # heavy overlap: big leaves
    elif kind == 2:
        sd = synth.library_synth(detail=int(rng.integers(1, 4)), seed=seed)
    else:
        sd = synth.bathroom_synth(detail=int(rng.integers(1, 4)), width=64, height=64)
    # geometry variations on top (light_infos are dropped: only batched queries are made, and degenerate emitters are refused)
    twist = int(rng.integers(0, 6))
    sd.light_infos = []
    sd.verts = np.ascontiguousarray(sd.verts, np.float32).copy()
    if twist == 1:                                               # every triangle twice: exact distance ties everywhere
        for k in ("vi", "ni", "ti", "mat_id"):
            setattr(sd, k, np.concatenate([getattr(sd, k), getattr(sd, k)]))
    elif twist == 2:                                             # a tiny scene
        sd.verts *= np.float32(1e-5)
    elif twist == 3:                                             # far from the origin: coordinates with few bits left for detail
        sd.verts += np.float32(3000.0)
    elif twist == 4:                                             # flattened onto a plane: every box is flat in z
        sd.verts[:, 2] = np.float32(0.25)
    elif twist == 5:                                             # some triangles collapsed to segments and points
        k = rng.integers(0, sd.n_tris, max(1, sd.n_tris // 10))
        sd.vi = sd.vi.copy()
        sd.vi[k, 2] = sd.vi[k, 1]
        sd.vi[k[::2], 1] = sd.vi[k[::2], 0]
    g = api.Scene(sd)
    v = sd.verts[sd.vi - 1]                                      # T x 3 x 3
    lo, hi = sd.verts.min(0), sd.verts.max(0)
    ext = np.maximum(hi - lo, 1e-3 * np.abs(hi - lo).max())
    a = lo - 0.2 * ext + rng.random((n, 3)) * 1.4 * ext
    b = lo - 0.2 * ext + rng.random((n, 3)) * 1.4 * ext
    on = rng.random(n) < 0.4                                     # start on a random triangle, offset along nothing: grazing starts
    t = rng.integers(0, len(v), n)
    w = rng.dirichlet((1, 1, 1), n)
    p = (v[t] * w[:, :, None]).sum(1)
    a[on] = p[on]
    to = rng.random(n) < 0.4                                     # aim at a random point of a random triangle
    t2 = rng.integers(0, len(v), n)
    w2 = rng.dirichlet((1, 1, 1), n)
    b[to] = (v[t2] * w2[:, :, None]).sum(1)[to]
    snap = rng.random(n) < 0.15                                  # coordinates on a coarse grid: origins and targets on box and triangle planes
    unit = float(2.0 ** np.round(np.log2(ext.max() / 32)))       # a power of two: grid coordinates are exact
    a[snap] = np.round(a[snap] / unit) * unit
    b[snap] = np.round(b[snap] / unit) * unit
    d = b - a
    graze = rng.random(n) < (0.6 if GRAZE else 0.0)   # towards a point of a triangle, nearly in its plane
    if graze.any():
        tg = rng.integers(0, len(v), n)
        wg = rng.dirichlet((1, 1, 1), n)
        wg[rng.random(n) < 0.3] = np.array([0.5, 0.5, 0.0])                       # on an edge (a quad's diagonal among them)
        pt = (v[tg] * wg[:, :, None]).sum(1)
        e1, e2 = v[tg, 1] - v[tg, 0], v[tg, 2] - v[tg, 0]
        nrm = np.cross(e1, e2)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
        ang = rng.uniform(0, 2 * np.pi, n)
        u = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-30)
        w3 = np.cross(nrm, u)
        inplane = u * np.cos(ang)[:, None] + w3 * np.sin(ang)[:, None]
        eps = 10.0 ** rng.uniform(-6, -1.5, n) * np.where(rng.random(n) < 0.5, 1, -1)
        dg = inplane + nrm * eps[:, None]
        L = rng.uniform(0.05, 1.0, n) * ext.max()
        a[graze] = (pt - dg * L[:, None])[graze]
        b[graze] = pt[graze]
        d = b - a
    ax = rng.random(n) < 0.05                                    # axis-parallel: zero components
    d[ax] *= (rng.random((ax.sum(), 3)) < 0.5)
    d[(np.abs(d) < 1e-30).all(1)] = (0.0, 0.0, 1.0)
    dist = np.linalg.norm(d, axis=1)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = a, d / dist[:, None]
    rays[:, 6] = np.where(rng.random(n) < 0.5, 0.0, 1e-4 * ext.max() / 4)
    rays[:, 7] = np.where(rng.random(n) < 0.5, np.inf, dist * np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0.3, 1.5, n)))
    e, f = g.intersect(rays, api.TRAVERSE_EXACT), g.intersect(rays, api.TRAVERSE_FAST)
    same = np.array_equal(e.view(np.uint32), f.view(np.uint32))
    oe, of = g.occluded(rays, api.TRAVERSE_EXACT)[0], g.occluded(rays, api.TRAVERSE_FAST)[0]
    same_o = np.array_equal(oe, of)
    print("seed", seed, "kind", kind, "twist", twist, "tris", sd.n_tris, "hit rate %.2f" % (e["prim"] >= 0).mean(), "occluded %.2f" % oe.mean(), "closest", same, "any", same_o, flush=True)
    if not (same and same_o):
        bad += 1
        diff = np.nonzero((e.view(np.uint32).reshape(n, -1) != f.view(np.uint32).reshape(n, -1)).any(1))[0]
        print("  differing closest:", len(diff), diff[:5], "any:", np.nonzero(oe != of)[0][:5])
        np.set_printoptions(precision=9, linewidth=220)
        corners = sd.verts[sd.vi - 1].reshape(-1, 9)
        for i in diff[:4]:
            hit, tuv = api.probe_triangle(corners, np.repeat(rays[i:i + 1], len(corners), 0))
            print("   ray", rays[i], "\n     reference order", e[i], "ordered", f[i], "ordered, alone", g.intersect(rays[i:i + 1].copy(), api.TRAVERSE_FAST)[0],
                  "\n     triangles accepted on their own:", np.nonzero(hit)[0], tuv[hit == 1][:, 0])
    g.close()
print("scenes with differences:", bad)
sys.exit(1 if bad else 0)
