"""Function-level vectors of the CPU oracle for the hot path: what is computed, from fixed seeds.

`compute()` returns a dict of arrays; `make_vectors.py` stores it as function_vectors.npz and tests/test_golden_vectors.py
recomputes it and compares (bit for bit where only +, -, *, /, sqrt and integer arithmetic are involved, to 1e-6 where
libm's sin / cos / pow / exp enter).  A change of any arithmetic convention in oracle/pt_oracle.cpp -- summation order of
the dot products, the order of the two BRDF sample draws, a comparison turned strict -- turns one of them red.

The vectors pin the ORACLE against itself over time; what pins the oracle against the reference are the reference's own
images (tests/test_oracle.py).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p in (os.path.join(ROOT, "path-tracer_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

EXACT = ("tri_", "box_", "topo_", "hit_", "occ_", "sobol_", "splat_weight", "li_dims", "li_nrays", "li_pixel_sample")   # compared bit for bit


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def _adversarial_tri_rays(sd, rng, n):
    """Rays aimed at random points of random triangles, at edges and vertices exactly, nearly parallel to the plane, with the
    far end exactly at / just before the hit, plus plain random rays."""
    tri = rng.integers(0, sd.n_tris, n)
    v = sd.verts[sd.vi[tri] - 1].astype(np.float64)                       # n x 3 x 3
    kind = rng.integers(0, 6, n)
    b = rng.dirichlet((1, 1, 1), n)
    b[kind == 1] = np.array([0.5, 0.5, 0.0])                              # on an edge
    b[kind == 2] = np.array([1.0, 0.0, 0.0])                              # on a vertex
    target = (v * b[:, :, None]).sum(1)
    org = target + rng.normal(size=(n, 3)) * 3.0
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    graze = kind == 3                                                     # nearly in the triangle's plane: |det| around 1e-5
    e = v[:, 1] - v[:, 0]
    org[graze] = target[graze] + e[graze] * 2.0 + nrm[graze] * rng.uniform(-3e-5, 3e-5, (graze.sum(), 1))
    d = target - org
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d
    rays[:, 6] = np.where(kind == 5, (dist * 1.5).astype(np.float32), 1e-4)   # near end behind the hit
    rays[:, 7] = np.inf
    return tri.astype(np.uint32), rays, kind


def compute():
    import orc
    from pt_hip import api, synth
    from pt_hip.scene import SceneData
    out = {}
    rng = np.random.default_rng(20260211)

    # (i) ray-triangle (src/shape.cpp:50-89)
    soup = synth.random_soup(200, seed=5)
    o = orc.Scene(soup)
    tri, rays, kind = _adversarial_tri_rays(soup, rng, 10000)
    res = [o.tri_intersect(int(t), r) for t, r in zip(tri, rays)]
    hit = np.array([r[0] for r in res], np.uint8)
    t = np.array([r[2] for r in res], np.float32)
    # far end exactly at the computed distance must still hit (t <= max_dis), one ulp before must not
    at = rays.copy()
    at[:, 7] = np.where(hit == 1, t, 1.0)
    before = at.copy()
    before[:, 7] = np.nextafter(at[:, 7], np.float32(0))
    out["tri_prim"], out["tri_rays"] = tri, rays
    out["tri_hit"] = hit
    out["tri_t"] = np.where(hit == 1, t, 0).astype(np.float32)
    out["tri_bary"] = np.where(hit[:, None] == 1, np.array([r[1] for r in res], np.float32), 0).astype(np.float32)
    out["tri_hit_far_end_at_t"] = np.array([o.tri_intersect(int(p), r)[0] for p, r in zip(tri, at)], np.uint8)
    out["tri_hit_far_end_before_t"] = np.array([o.tri_intersect(int(p), r)[0] for p, r in zip(tri, before)], np.uint8)

    # (ii) ray-box (src/aabb.cpp:6-35): random, axis-parallel (division by zero), origin on a slab plane (0/0), inverted ranges
    n = 10000
    lo = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 2, (n, 3)) * (rng.random((n, 3)) > 0.1)).astype(np.float32)       # one in ten extents is flat
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = rng.uniform(-3, 3, (n, 3))
    d = rng.normal(size=(n, 3))
    aimed = rng.random(n) < 0.6                                                                # towards a point of the box
    d[aimed] = (lo + (hi - lo) * rng.random((n, 3)))[aimed] - r[aimed, 0:3]
    zero = rng.random((n, 3)) < 0.15
    d[zero] = 0.0
    d[(np.abs(d) < 1e-30).all(1)] = (1.0, 0.0, 0.0)
    r[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    on = rng.random((n, 3)) < 0.1
    r[:, 0:3] = np.where(on, np.where(rng.random((n, 3)) < 0.5, lo, hi), r[:, 0:3])             # origin exactly on a plane
    r[:, 6] = np.where(rng.random(n) < 0.2, rng.uniform(0, 4, n), 0.0)
    r[:, 7] = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0, 4, n))
    boxes = np.concatenate([lo, hi], 1).astype(np.float32)
    out["box_boxes"], out["box_rays"] = boxes, r
    out["box_hit"] = np.array([orc.lib().orc_aabb_intersect(orc._p(b, orc.c_f32p), orc._p(q, orc.c_f32p)) for b, q in zip(boxes, r)], np.uint8)

    # (iii) tree topology (src/bvh.cpp:99-234): the reference's own mesh as a digest, a 100-triangle mesh in full
    veach = SceneData.load(os.path.join(ROOT, "tests", "golden", "veach_mis_scene.npz"))
    vo = orc.Scene(veach)
    bx, first, count, prim = vo.bvh()
    out["topo_veach_nodes"] = np.array([len(first)], np.uint32)
    out["topo_veach_digest"] = digest(bx, first, count, prim)
    small = orc.Scene(synth.random_soup(100, seed=9))
    bx, first, count, prim = small.bvh()
    out["topo_small_boxes"], out["topo_small_first"], out["topo_small_count"], out["topo_small_prim"] = bx, first.astype(np.uint32), count.astype(np.uint32), prim.astype(np.uint32)

    # (iv) closest / any hit on the reference's mesh: every primary ray of sample 0 as a digest and every 97th in full;
    #      bounce and shadow rays recorded from the integrator
    import helpers
    prim_rays = helpers.primary_rays(vo, sample=0, step=1)
    h = vo.intersect(prim_rays)
    out["hit_primary_digest"] = digest(h["prim"], h["t"], h["bary"])
    out["hit_primary_every97"] = np.concatenate([h["prim"][::97, None].astype(np.float32).view(np.uint32), h["t"][::97, None].view(np.uint32),
                                                 h["bary"][::97].view(np.uint32)], 1)
    closest, shadow = helpers.recorded_rays(vo, 1500, seed=3)
    closest, shadow = closest[:10000], shadow[:10000]
    hc = vo.intersect(closest)
    occ, firstp = vo.occluded(shadow)
    out["hit_rays"], out["hit_prim"], out["hit_t"], out["hit_bary"] = closest, hc["prim"], hc["t"], hc["bary"]
    out["occ_rays"], out["occ_flag"], out["occ_first"] = shadow, occ, firstp

    # (v) Sobol' (src/sampler.cpp:6-36, 96-123): index and dimensions 0..15 at m = 10, 11, 12
    rows = []
    for m in (10, 11, 12):
        for k in range(200):
            px, py, s = int(rng.integers(0, 1 << m)), int(rng.integers(0, 1 << m)), int(rng.integers(0, 4096))
            idx = orc.sobol_index(m, s, px, py)
            rows.append([m, px, py, s, idx & 0xffffffff, idx >> 32] + [int(np.float32(orc.sobol_sample(idx, dim)).view(np.uint32)) for dim in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 1023)])
    out["sobol_rows"] = np.array(rows, np.uint64).astype(np.uint32)

    # (vi) Phong (src/material.cpp:9-115) on a scene with diffuse, glossy and mirror materials; (vii) area lights (src/light.cpp)
    room = synth.bathroom_synth(detail=1, width=64, height=64)
    ro = orc.Scene(room)
    n = 2000
    prims = rng.integers(0, room.n_tris, n).astype(np.uint32)
    bary = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    wo = rng.normal(size=(n, 3))
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    wi = rng.normal(size=(n, 3))
    wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    uc, u2 = rng.random(n).astype(np.float32), rng.random((n, 2)).astype(np.float32)
    out["phong_prim"], out["phong_bary"], out["phong_wo"], out["phong_wi"], out["phong_uc"], out["phong_u2"] = prims, bary, wo, wi, uc, u2
    out["phong_f"] = np.array([ro.brdf_eval(int(p), b, a, c) for p, b, a, c in zip(prims, bary, wo, wi)], np.float32)
    out["phong_pdf"] = np.array([ro.brdf_pdf(int(p), b, a, c) for p, b, a, c in zip(prims, bary, wo, wi)], np.float32)
    smp = [ro.brdf_sample(int(p), b, a, float(c), d2) for p, b, a, c, d2 in zip(prims, bary, wo, uc, u2)]
    out["phong_sample"] = np.array([[float(s[0])] + list(s[1]) + [s[2]] + list(s[3]) for s in smp], np.float32)
    lights = rng.integers(0, ro.n_lights, n).astype(np.uint32)
    ls = [ro.light_sample(int(l), int(p), b, d2) for l, p, b, d2 in zip(lights, prims, bary, u2)]
    out["light_id"] = lights
    out["light_sample"] = np.array([np.concatenate([s["L"], s["wi"], s["p"], s["n"], [s["pdfArea"], s["pdfDir"]]]) for s in ls], np.float32)

    # (viii) film (src/block.cpp:14-35, 60-95): the weight table and a 64 x 64 splat of 500 samples
    out["film_table"] = orc.filter_table()
    film = np.zeros((68, 68, 4), np.float32)
    pos = rng.uniform(0, 64, (500, 2)).astype(np.float32)
    rgb = rng.random((500, 3)).astype(np.float32)
    for p, c in zip(pos, rgb):
        orc.film_put(film, p, c)
    out["film_pos"], out["film_rgb"], out["film_splat"] = pos, rgb, film

    # (ix) Integrator::Li (src/integrator.cpp:45-141) for 256 pixels x 16 samples of the reference's scene, with the number
    #      of Sobol' dimensions each path consumed and its ray count (they pin the order of the sampler draws)
    pix = np.stack([rng.integers(0, veach.width, 256), rng.integers(0, veach.height, 256)], 1).astype(np.int32)
    li_rgb, li_dims, li_nrays, li_ps = [], [], [], []
    for px, py in pix:
        for s in range(16):
            r = vo.li(int(px), int(py), s, max_log=1)
            li_rgb.append(r["rgb"])
            li_dims.append(r["dims"])
            li_nrays.append(r["n_rays"])
            li_ps.append(r["pixel_sample"])
    out["li_pixels"] = pix
    out["li_rgb"] = np.array(li_rgb, np.float32)
    out["li_dims"] = np.array(li_dims, np.int32)
    out["li_nrays"] = np.array(li_nrays, np.int32)
    out["li_pixel_sample"] = np.array(li_ps, np.float32)
    return out


def compare(got, want):
    """Names of the entries that differ (bit for bit for the EXACT families, 1e-6 relative otherwise)."""
    bad = []
    for k in sorted(want):
        a, b = np.asarray(got[k]), np.asarray(want[k])
        if a.shape != b.shape:
            bad.append(k + " (shape)")
        elif k.startswith(EXACT) or a.dtype.kind in "ui":
            if a.tobytes() != b.astype(a.dtype).tobytes():
                bad.append(k)
        elif not np.allclose(a, b, rtol=1e-6, atol=1e-7, equal_nan=True):
            bad.append(k)
    return bad
