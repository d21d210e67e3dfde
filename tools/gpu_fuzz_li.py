"""Scratch: per-sample radiance of the HIP path tracer against the CPU oracle on random scenes (materials, lights, textures,
camera and frame size drawn at random on top of the scene generators).  The tolerance is the test suite's: film positions
bit for bit, >= 99.5 % of the samples within 1e-3 relative, >= 95 % identical in every bit, means within 2 %.  Usage: gpu_fuzz_li.py <scenes> <samples>"""
import sys
sys.path[:0] = ['path-tracer_amd/python', '.', 'oracle']
import numpy as np
from pt_hip import api, synth
import orc

seeds, n = int(sys.argv[1]), int(sys.argv[2])
only = [int(x) for x in sys.argv[3:]]
bad = 0
for seed in (only or range(seeds)):
    rng = np.random.default_rng(7000 + seed)
    kind = seed % 5
    w, h = int(rng.integers(17, 300)), int(rng.integers(17, 200))
    if kind == 0:
        sd = synth.random_soup(int(rng.integers(2, 3000)), seed=seed, width=w, height=h, extent=float(rng.uniform(1, 10)), size=float(rng.uniform(0.1, 3.0)),
                               n_lights=int(rng.integers(1, 9)))
    elif kind == 1:
        sd = synth.library_synth(detail=1, seed=seed, width=w, height=h)
    elif kind == 2:
        sd = synth.bathroom_synth(detail=int(rng.integers(1, 3)), width=w, height=h)
    elif kind == 3:
        sd = synth.cornell_box(width=w, height=h)
    else:
        sd = synth.textured_room(width=w, height=h)
    # material variations: mirror-like Ks, pure diffuse, pure specular, high exponents
    m = len(sd.mat_ns)
    pick = rng.random(m)
    sd.mat_ks = np.where((pick < 0.15)[:, None], 1000.0, np.where((pick > 0.85)[:, None], 0.0, sd.mat_ks)).astype(np.float32)
    sd.mat_kd = np.where(((pick > 0.15) & (pick < 0.25))[:, None], 0.0, sd.mat_kd).astype(np.float32)
    sd.mat_ns = np.where(rng.random(m) < 0.2, rng.uniform(200, 2000, m), sd.mat_ns).astype(np.float32)
    o, g = orc.Scene(sd), api.Scene(sd)
    pix = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.int32)
    smp = rng.integers(0, 4096, n).astype(np.uint32)
    res = [o.li(int(p[0]), int(p[1]), int(s)) for p, s in zip(pix, smp)]
    want = np.array([r["rgb"] for r in res])
    want_ps = np.array([r["pixel_sample"] for r in res])
    got, ps, st = g.li(pix, smp, traversal=api.TRAVERSE_FAST)
    err = np.abs(got - want).max(1) / (np.abs(want).max(1) + 1e-3)
    ok_ps = np.array_equal(ps.view(np.uint32), want_ps.view(np.uint32))
    close, exact = (err < 1e-3).mean(), (err == 0).mean()
    mean_ok = abs(got.mean() - want.mean()) <= 0.02 * max(want.mean(), 1e-6)
    fine = ok_ps and close >= 0.995 and exact >= 0.95 and mean_ok and np.isfinite(got).all() and (got >= 0).all()
    print("seed", seed, "kind", kind, sd.name, "tris", sd.n_tris, "%dx%d" % (w, h), "within 1e-3: %.4f" % close, "identical: %.3f" % exact,
          "means %.5g %.5g" % (got.mean(), want.mean()), "positions", ok_ps, "ok" if fine else "DIFFERENT", flush=True)
    if not fine:
        bad += 1
        w_ = np.argsort(-err)[:3]
        print("   worst:", pix[w_].tolist(), smp[w_].tolist(), got[w_].tolist(), want[w_].tolist())
    g.close()
print("scenes with differences:", bad)
sys.exit(1 if bad else 0)
