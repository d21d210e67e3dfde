"""GPU parity: the HIP path, called through the C ABI (libpt_hip.so), against the CPU oracle on the same inputs.

Bars: bit-exact for hit triangle index, t, u, v, occlusion flags, first occluder and Sobol' values; radiance samples
identical in every bit for >= 98 % (measured 99.3 - 100 %) and within 1e-3 for >= 99.8 %: the device's sine, cosine and
power are correctly rounded, glibc's are in all but ~1 % of their results, and a path that meets such a result can
diverge from there on.
"""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
import orc
from pt_hip import api, synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ sampler

def test_sobol_bit_exact(veach_gpu, veach_sd):
    rng = np.random.default_rng(0)
    n = 30000
    pix = np.stack([rng.integers(0, veach_sd.width, n), rng.integers(0, veach_sd.height, n)], 1).astype(np.int32)
    smp = rng.integers(0, 4096, n).astype(np.uint32)
    smp[:100] = np.arange(100)
    dim = rng.integers(0, 1024, n).astype(np.uint32)
    dim[:4] = [0, 1, 2, 1023]
    got = veach_gpu.sobol(pix, smp, dim)
    want = np.array([orc.sobol_sample(orc.sobol_index(11, int(s), int(p[0]), int(p[1])), int(d)) for p, s, d in zip(pix, smp, dim)],
                    np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_short_division_is_correctly_rounded():
    """The 5-instruction division used by the box test returns the same bits as the full IEEE sequence."""
    assert api.selftest_division(1 << 30, seed=12345) == 0
    assert api.selftest_division(1 << 28, seed=999) == 0


# ------------------------------------------------------------------------------------------------ traversal

@pytest.fixture(scope="module")
def veach_rays(veach_oracle):
    prim = helpers.primary_rays(veach_oracle, sample=0, step=3)
    closest, shadow = helpers.recorded_rays(veach_oracle, 4000)
    return prim, closest, shadow


def test_closest_hit_exact_mode_bit_exact(veach_gpu, veach_oracle, veach_rays):
    for rays in veach_rays[:2]:
        want = veach_oracle.intersect(rays)
        got = veach_gpu.intersect(rays, api.TRAVERSE_EXACT)
        assert helpers.hits_equal(got, want).all()
        assert (want["prim"] >= 0).mean() > 0.5


def test_closest_hit_fast_mode(veach_gpu, veach_oracle, veach_rays):
    """The ordered traversal reports the same triangle, t, u, v as the reference order: it takes the same box decisions
    and repeats a query in reference order whenever two candidates are too close for the order not to matter."""
    for rays in veach_rays[:2]:
        want = veach_oracle.intersect(rays)
        got = veach_gpu.intersect(rays, api.TRAVERSE_FAST)
        assert helpers.hits_equal(got, want).all()


def test_any_hit_bit_exact(veach_gpu, veach_oracle, veach_rays):
    shadow = veach_rays[2]
    occ, first = veach_oracle.occluded(shadow)
    g_occ, g_first = veach_gpu.occluded(shadow, api.TRAVERSE_EXACT)
    assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
    assert 0.05 < occ.mean() < 0.95
    f_occ, _ = veach_gpu.occluded(shadow, api.TRAVERSE_FAST)
    assert np.array_equal(f_occ, occ)


def test_empty_and_degenerate_batches(veach_gpu, veach_oracle):
    assert len(veach_gpu.intersect(np.zeros((0, 8), np.float32))) == 0
    assert len(veach_gpu.occluded(np.zeros((0, 8), np.float32))[0]) == 0
    # axis-parallel directions (zero components -> +-inf and NaN slabs), zero-length range, reversed range
    rays = np.array([[28.0, 5.0, 0.0, -1, 0, 0, 0, np.inf], [0.0, 20.0, 0.0, 0, -1, 0, 0, np.inf], [0, 3, -20, 0, 0, 1, 0, np.inf],
                     [28.0, 5.0, 0.0, -1, 0, 0, 5, 5], [28.0, 5.0, 0.0, -1, 0, 0, 9, 1], [0, 0, 0, 0, 0, 0, 0, 1]], np.float32)
    want = veach_oracle.intersect(rays)
    for mode in (api.TRAVERSE_EXACT, api.TRAVERSE_FAST):
        assert helpers.hits_equal(veach_gpu.intersect(rays, mode), want).all()
    occ, first = veach_oracle.occluded(rays)
    g_occ, g_first = veach_gpu.occluded(rays, api.TRAVERSE_EXACT)
    assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)


def test_equal_distance_hits_follow_reference_order():
    """Rays aimed exactly at shared edges and vertices of a flat grid hit two or more triangles at the same distance;
    the reference keeps the last one in its visit order (src/bvh.cpp:39-46, src/shape.cpp:88).  With axis-parallel
    directions a ray origin that lies on a slab plane gives 0/0 = NaN in the reference's box test, which then skips or
    keeps boxes erratically (src/aabb.cpp:7-34): both traversal modes must reproduce that as well."""
    sd = synth.grid_plane(16)
    o, g = orc.Scene(sd), api.Scene(sd)
    xs = np.arange(0, 17, 0.5, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs)
    n = gx.size
    down = np.zeros((n, 8), np.float32)
    down[:, 0], down[:, 1], down[:, 2] = gx.ravel(), 3.0, gz.ravel()
    down[:, 4] = -1.0
    down[:, 7] = np.inf
    diag = down.copy()                      # direction (1,-1,1): reaches the plane at t = 3 exactly on the grid point
    diag[:, 0] -= 3.0
    diag[:, 2] -= 3.0
    diag[:, 3], diag[:, 5] = 1.0, 1.0
    slanted = down.copy()                   # direction (1,-1,0): one zero component
    slanted[:, 0] -= 3.0
    slanted[:, 3] = 1.0
    for batch, lo, hi in ((down, 0.3, 0.9), (slanted, 0.3, 1.01), (diag, 0.8, 1.01)):
        want = o.intersect(batch)
        assert lo < (want["prim"] >= 0).mean() < hi
        assert helpers.hits_equal(g.intersect(batch, api.TRAVERSE_EXACT), want).all()
        assert helpers.hits_equal(g.intersect(batch, api.TRAVERSE_FAST), want).all()
        occ, first = o.occluded(batch)
        g_occ, g_first = g.occluded(batch, api.TRAVERSE_EXACT)
        assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
        assert np.array_equal(g.occluded(batch, api.TRAVERSE_FAST)[0], occ)
    g.close()


@pytest.mark.parametrize("make", [lambda: synth.random_soup(1, seed=1), lambda: synth.random_soup(7, seed=2),
                                  lambda: synth.random_soup(3000, seed=3), synth.cornell_box, lambda: synth.library_synth(detail=1),
                                  lambda: synth.bathroom_synth(detail=1, width=96, height=96), synth.textured_room])
def test_traversal_parity_synthetic_scenes(make):
    sd = make()
    o, g = orc.Scene(sd), api.Scene(sd)
    prim = helpers.primary_rays(o, sample=1, step=max(1, sd.width // 64))
    closest, shadow = helpers.recorded_rays(o, 300, seed=4)
    rays = np.concatenate([prim, closest])
    want = o.intersect(rays)
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    if len(shadow):
        occ, first = o.occluded(shadow)
        g_occ, g_first = g.occluded(shadow, api.TRAVERSE_EXACT)
        assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
        assert np.array_equal(g.occluded(shadow, api.TRAVERSE_FAST)[0], occ)
    g.close()


def test_rays_that_start_on_a_surface_hit_it_at_distance_zero():
    """Origins exactly on triangles (and on the shared planes of coplanar neighbours) with a near end of zero: the
    reference reports the hit at t = 0.  The ordered mode's distance windows are relative and have no width there; such
    queries must fall back to the reference order instead of replaying it up to a bound of zero."""
    sd = synth.library_synth(detail=1, seed=2)
    o, g = orc.Scene(sd), api.Scene(sd)
    rng = np.random.default_rng(11)
    n = 60000
    v = sd.verts[sd.vi - 1]
    t = rng.integers(0, len(v), n)
    w = rng.dirichlet((1, 1, 1), n)
    w[: n // 4] = np.array([0.5, 0.5, 0.0])                                  # on an edge: two triangles at distance zero
    org = (v[t] * w[:, :, None]).sum(1)
    d = rng.normal(size=(n, 3))
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.0, 2.0, n))
    want = o.intersect(rays)
    assert ((want["prim"] >= 0) & (want["t"] == 0)).mean() > 0.2
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    occ = o.occluded(rays)[0]
    assert np.array_equal(g.occluded(rays, api.TRAVERSE_EXACT)[0], occ) and np.array_equal(g.occluded(rays, api.TRAVERSE_FAST)[0], occ)
    g.close()


def test_grazing_rays_that_end_on_a_flat_scene():
    """Every triangle in the plane z = 0.25 (every box is flat, overlapping triangles at equal distances) and rays a few
    millimetres above it that end exactly where they reach a triangle: the distance of the triangle test and the plane
    distance of the flat boxes around it differ by rounding, on either side of the ray's far end.  The reference compares
    only the x/y entry of a box with its bound; the ordered mode culls by all three slabs and needs its margin there."""
    sd = synth.random_soup(220, seed=25, extent=4.0, size=6.0)
    sd.light_infos = []
    sd.verts = np.ascontiguousarray(sd.verts, np.float32).copy()
    sd.verts[:, 2] = np.float32(0.25)
    o, g = orc.Scene(sd), api.Scene(sd)
    rng = np.random.default_rng(7)
    n = 150000
    v = sd.verts[sd.vi - 1].astype(np.float64)
    w = rng.dirichlet((1, 1, 1), n)
    target = (v[rng.integers(0, len(v), n)] * w[:, :, None]).sum(1)
    org = target + rng.uniform(-12, 12, (n, 3))
    org[:, 2] = 0.25 + rng.uniform(-0.012, 0.012, n)
    d = target - org
    dist = np.linalg.norm(d, axis=1)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d / dist[:, None]
    rays[:, 7] = np.where(rng.random(n) < 0.8, dist, np.inf)
    want = o.intersect(rays)
    assert 0.2 < (want["prim"] >= 0).mean() < 0.999
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    occ = o.occluded(rays)[0]
    assert np.array_equal(g.occluded(rays, api.TRAVERSE_EXACT)[0], occ) and np.array_equal(g.occluded(rays, api.TRAVERSE_FAST)[0], occ)
    g.close()


def test_needle_and_collapsed_triangles_switch_the_scene_to_reference_order():
    """Large triangles collapsed to segments: for rays that start next to one, the reference's triangle test "hits" it at
    distances that have nothing to do with where it lies (its determinant is rounding noise above the 1e-5 threshold).
    No distance-ordered walk can follow that; a scene with such triangles is traversed in reference order in both modes
    (the ordered mode then reports the reference's first occluder too)."""
    sd = synth.random_soup(160, seed=69, extent=4.0, size=7.0)
    sd.light_infos = []
    rng = np.random.default_rng(3)
    k = rng.integers(0, sd.n_tris, 16)
    sd.vi = sd.vi.copy()
    sd.vi[k, 2] = sd.vi[k, 1]
    o, g = orc.Scene(sd), api.Scene(sd)
    n = 100000
    v = sd.verts[sd.vi - 1].astype(np.float64)
    w = rng.dirichlet((1, 1, 1), n)
    org = (v[k[rng.integers(0, len(k), n)]] * w[:, :, None]).sum(1)                 # on the collapsed triangles
    d = rng.normal(size=(n, 3))
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 7] = np.inf
    want = o.intersect(rays)
    assert ((want["prim"] >= 0) & np.isin(want["prim"], k)).sum() > 20              # the noise hits are there
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    occ, first = o.occluded(rays)
    g_occ, g_first = g.occluded(rays, api.TRAVERSE_FAST)
    assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
    g.close()


def test_astronomically_large_scene_takes_the_plain_division_path():
    """Coordinates beyond 2^59: the products of the short division could overflow, so every ray of such a scene is routed
    through the IEEE division form of the box test (in the follow-up kernels of the ordered mode).  Same hits as the
    oracle, bit for bit, in both modes."""
    sd = synth.random_soup(400, seed=21, extent=2e18, size=4e17)
    o, g = orc.Scene(sd), api.Scene(sd)
    rng = np.random.default_rng(5)
    n = 20000
    org = rng.normal(size=(n, 3))
    org *= 6e18 / np.linalg.norm(org, axis=1, keepdims=True)
    d = rng.uniform(-2e18, 2e18, (n, 3)) - org
    dist = np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d / dist
    rays[:, 7] = np.where(rng.random(n) < 0.5, np.inf, (dist[:, 0] * rng.uniform(0.5, 1.5, n)))   # camera rays would end after 10^4 units
    want = o.intersect(rays)
    assert 0.1 < (want["prim"] >= 0).mean() < 0.99
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    occ, first = o.occluded(rays)
    g_occ, g_first = g.occluded(rays, api.TRAVERSE_EXACT)
    assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
    assert np.array_equal(g.occluded(rays, api.TRAVERSE_FAST)[0], occ)
    g.close()


def test_hand_made_deep_tree_runs_on_the_spilled_stack():
    """A 46-level comb-shaped tree (helpers.comb_scene_and_tree) handed to the backend in place of the builder's: rays
    along the spine keep up to 44 deferred subtrees on the traversal stack, well past its LDS rows.  The oracle walks the
    SAME tree (its test hook for a given node array), so hits, occlusion flags and the first occluder must be the
    oracle's bit for bit in both modes -- the spilled part of the stack included."""
    sd, nodes, ids = helpers.comb_scene_and_tree(44)
    g, o = api.Scene(sd, bvh=(nodes, ids)), orc.Scene(sd, bvh=(nodes, ids))
    rng = np.random.default_rng(3)
    n = 6000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0] = 50.0
    rays[:, 1:3] = rng.uniform(-0.7, 0.7, (n, 2))
    rays[:, 3] = -1.0
    rays[:, 4:6] = rng.uniform(-0.01, 0.01, (n, 2))
    rays[: n // 3, 0] = -6.0                         # the other way round: nothing is deferred
    rays[: n // 3, 3] = 1.0
    rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True)
    rays[:, 7] = np.inf
    rays[n // 2:, 7] = rng.uniform(5.0, 60.0, n - n // 2)
    exact, fast, want = g.intersect(rays, api.TRAVERSE_EXACT), g.intersect(rays, api.TRAVERSE_FAST), o.intersect(rays)
    assert 0.3 < (want["prim"] >= 0).mean() < 0.999
    assert helpers.hits_equal(exact, want).all()
    assert helpers.hits_equal(fast, want).all()
    (occ_e, first_e), occ_f, (occ_w, first_w) = g.occluded(rays, api.TRAVERSE_EXACT), g.occluded(rays, api.TRAVERSE_FAST)[0], o.occluded(rays)
    assert np.array_equal(occ_e, occ_w) and np.array_equal(first_e, first_w) and np.array_equal(occ_f, occ_w)
    film, st = g.render(0, 8, traversal=api.TRAVERSE_FAST)
    film_e, st_e = g.render(0, 8, traversal=api.TRAVERSE_EXACT)
    want_film, _ = o.render(0, 8)
    assert (st["closest_rays"], st["any_rays"]) == (st_e["closest_rays"], st_e["any_rays"])
    assert helpers.rel_mse(api.film_to_image(film), orc.film_to_image(want_film)) < 1e-5
    g.close()


def test_tree_deeper_than_the_traversal_stack_is_refused():
    sd, nodes, ids = helpers.comb_scene_and_tree(100)
    with pytest.raises(api.PtError, match="deeper than the traversal stack"):
        api.Scene(sd, bvh=(nodes, ids))
    nodes[5]["first"] = 0                            # a child index that closes a cycle
    with pytest.raises(api.PtError, match="node shared"):
        api.Scene(sd, bvh=(nodes, ids))


# ------------------------------------------------------------------------------------------------ film and AOVs

@pytest.mark.parametrize("name,integ", [("normal", api.INTEGRATOR_NORMAL), ("albedo", api.INTEGRATOR_ALBEDO)])
def test_aov_film_matches_oracle_and_reference_images(veach_gpu, veach_oracle, name, integ):
    """Camera ray, Sobol' jitter, primary hit, normal interpolation and film reconstruction: equal to the oracle up to
    float summation order, and equal to the image the reference binary wrote (tests/golden) at the same tolerance the
    oracle itself meets."""
    import os
    from conftest import GOLDEN
    film, st = veach_gpu.render(0, 32, integrator=integ, traversal=api.TRAVERSE_EXACT)
    assert st["paths"] == 1280 * 720 * 32 and st["closest_rays"] == st["paths"] and st["any_rays"] == 0
    want, _ = veach_oracle.render(0, 32, integ)
    assert np.allclose(film, want, rtol=1e-4, atol=2e-4)      # sums of ~500 weighted terms in a different order
    z = np.load(os.path.join(GOLDEN, "veach_mis_aov.npz"))
    stride = int(z["stride"])
    d = np.abs(api.film_to_image(film)[::stride, ::stride] - z[name]).max(-1)
    assert d.mean() < 5e-6 and (d > 1e-5).mean() < 2e-3 and d.max() < 0.1


def test_film_weight_channel_is_exact_filter_sum(veach_gpu, veach_oracle):
    """The weight channel only depends on the sample positions: check it at a sample range the oracle never rendered
    with the path integrator (cheap AOV pass on the oracle side)."""
    film, _ = veach_gpu.render(100, 104, integrator=api.INTEGRATOR_ALBEDO)
    want, _ = veach_oracle.render(100, 104, orc.INTEGRATOR_ALBEDO)
    assert np.allclose(film[..., 3], want[..., 3], rtol=1e-5, atol=1e-6)
    assert film[:2].sum() > 0 and film[2:-2, 2:-2, 3].min() > 0    # border rows receive splats, every pixel has weight


# ------------------------------------------------------------------------------------------------ integrator

def test_per_sample_radiance_matches_oracle(veach_gpu, veach_oracle, veach_sd):
    """Integrator::Li for individual (pixel, sample) pairs.  Tolerance: at least 99% of the samples agree to 1e-3
    relative (most agree exactly); the rest may take a different path after a BRDF sample because device and host
    sin/cos/pow differ in the last bits.  The mean over all samples agrees to 2%."""
    rng = np.random.default_rng(7)
    n = 6000
    pix = np.stack([rng.integers(0, veach_sd.width, n), rng.integers(0, veach_sd.height, n)], 1).astype(np.int32)
    smp = rng.integers(0, 1024, n).astype(np.uint32)
    want = np.array([veach_oracle.li(int(p[0]), int(p[1]), int(s))["rgb"] for p, s in zip(pix, smp)])
    want_ps = np.array([veach_oracle.pixel_sample(int(p[0]), int(p[1]), int(s)) for p, s in zip(pix[:500], smp[:500])])
    for mode in (api.TRAVERSE_EXACT, api.TRAVERSE_FAST):
        got, ps, st = veach_gpu.li(pix, smp, traversal=mode)
        assert np.array_equal(ps[:500].view(np.uint32), want_ps.view(np.uint32))     # film positions are bit-exact
        err = np.abs(got - want).max(1) / (np.abs(want).max(1) + 1e-3)
        assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %
        assert (err == 0).mean() >= 0.5
        assert abs(got.mean() - want.mean()) <= 0.02 * want.mean()
        assert np.isfinite(got).all() and (got >= 0).all()


@pytest.mark.parametrize("make", [lambda: synth.library_synth(detail=1, width=96, height=64), lambda: synth.bathroom_synth(detail=2, width=64, height=64),
                                  lambda: synth.random_soup(3000, seed=3, width=80, height=60), synth.textured_room])
def test_both_traversal_modes_give_the_same_samples_bit_for_bit(make):
    """The two modes differ only in how the tree is walked: with identical hits and occlusion flags every sample takes
    the same path and adds the same terms in the same order, so the radiance of every (pixel, sample) pair must agree in
    every bit, and so must the ray counts of a render.  (Films agree only up to the order of atomic additions.)"""
    sd = make()
    g = api.Scene(sd)
    spp = 8
    ys, xs = np.mgrid[0:sd.height, 0:sd.width]
    pix = np.repeat(np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32), spp, 0)
    smp = np.tile(np.arange(spp, dtype=np.uint32), sd.width * sd.height)
    le, pe, _ = g.li(pix, smp, traversal=api.TRAVERSE_EXACT)
    lf, pf, _ = g.li(pix, smp, traversal=api.TRAVERSE_FAST)
    assert np.array_equal(pe.view(np.uint32), pf.view(np.uint32))
    assert np.array_equal(le.view(np.uint32), lf.view(np.uint32))
    assert np.isfinite(le).all() and le.max() > 0
    fe, se = g.render(0, spp, traversal=api.TRAVERSE_EXACT)
    ff, sf = g.render(0, spp, traversal=api.TRAVERSE_FAST, samples_per_pass=3)
    assert (se["closest_rays"], se["any_rays"]) == (sf["closest_rays"], sf["any_rays"])
    assert np.allclose(fe, ff, rtol=1e-5, atol=1e-6)
    g.close()


def test_path_traced_image_matches_oracle(veach_gpu, veach_oracle):
    """Converged-image criterion at equal sample sets: relMSE between the GPU film and the oracle film at 16 spp
    (same Sobol' points, so the noise cancels) must stay below 1e-5; the ray counts agree to 0.01%."""
    film, st = veach_gpu.render(0, 16, traversal=api.TRAVERSE_EXACT)
    want, ost = veach_oracle.render(0, 16)
    img, ref = api.film_to_image(film), orc.film_to_image(want)
    assert helpers.rel_mse(img, ref) < 1e-5
    # a pass of this size gathers its surviving paths into a few dense shards on the way and ends in the tail launch
    assert st["kernel_launches"].get("compact", 0) >= 1 and st["kernel_launches"].get("finish", 0) >= 1
    assert abs(st["closest_rays"] - ost["closest_rays"]) <= 1e-4 * ost["closest_rays"]
    assert abs(st["any_rays"] - ost["any_rays"]) <= 1e-4 * ost["any_rays"]
    fast, _ = veach_gpu.render(0, 16, traversal=api.TRAVERSE_FAST)
    assert helpers.rel_mse(api.film_to_image(fast), ref) < 1e-5


def test_path_traced_image_matches_reference_png(veach_gpu):
    """The HIP path against the reference binary's own path-traced image of this scene (scenes/veach-mis/result.png, kept
    as linear 8x8 block means in tests/golden/veach_mis_result.npz; see the oracle's test of the same name).  At 512
    samples per pixel our own noise is negligible: the total agrees to 0.3 % (measured 0.08 %) and every brightness band
    to 2 % (measured: within 1.2 %, most within 0.2 %); blocks are put into bands by a second render of other sample
    numbers, so that no image under comparison chooses its own band.  Both traversal modes."""
    from conftest import GOLDEN
    want = np.load(os.path.join(GOLDEN, "veach_mis_result.npz"))["linear"].astype(np.float64)
    lum = helpers.png_block_means(api.film_to_image(veach_gpu.render(6000, 6256, traversal=api.TRAVERSE_FAST)[0])).mean(-1)
    for mode, spp in ((api.TRAVERSE_FAST, 512), (api.TRAVERSE_EXACT, 128)):
        film, _ = veach_gpu.render(0, spp, traversal=mode)
        got = helpers.png_block_means(api.film_to_image(film))
        assert abs(got.sum() / want.sum() - 1.0) < 0.003
        assert np.abs(got - want).mean() < 0.025 * want.mean()      # the reference image's own noise
        for lo, hi in ((0.002, 0.005), (0.005, 0.01), (0.01, 0.02), (0.02, 0.05), (0.05, 0.1), (0.1, 0.2), (0.2, 0.5), (0.5, 1.01)):
            m = (lum >= lo) & (lum < hi)
            assert m.mean() > 0.02
            assert abs(got[m].mean() / want[m].mean() - 1.0) < 0.02, (mode, lo, hi)


def test_sample_ranges_add_up_and_pass_size_is_irrelevant(veach_gpu):
    """Multi-GPU sharding relies on this: films of disjoint sample ranges sum to the film of the union, and the number
    of samples kept in flight per pass does not change the result (both up to float summation order)."""
    a, _ = veach_gpu.render(0, 6, samples_per_pass=3)
    b, _ = veach_gpu.render(6, 12, samples_per_pass=6)
    c, _ = veach_gpu.render(0, 12, samples_per_pass=5)
    assert np.allclose(a + b, c, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("world", [2, 3, 7])
def test_block_shares_add_up_to_the_whole_frame(world):
    """Image split by 32x32 pixel blocks (the reference's BlockGenerator unit): the films of all ranks sum to the
    unsplit film, every path is rendered by exactly one rank, and a rank's film is empty away from its own blocks."""
    from pt_hip.dist import block_owner
    sd = synth.cornell_box(200, 136)             # 7 x 5 blocks, the last column and row partial
    g = api.Scene(sd)
    whole, wst = g.render(0, 6, traversal=api.TRAVERSE_FAST)
    total = np.zeros_like(whole)
    paths = rays = 0
    for rank in range(world):
        film, st = g.render(0, 6, traversal=api.TRAVERSE_FAST, block=(rank, world), samples_per_pass=4)
        total += film
        paths += st["paths"]
        rays += st["closest_rays"] + st["any_rays"]
        for by in range(5):
            for bx in range(7):
                if block_owner(bx, by, world) != rank and all(block_owner(bx + dx, by + dy, world) != rank
                                                              for dx in (-1, 0, 1) for dy in (-1, 0, 1)):
                    assert not film[2 + 32 * by:2 + 32 * by + 32, 2 + 32 * bx:2 + 32 * bx + 32].any()
    assert paths == wst["paths"] == 200 * 136 * 6
    assert rays == wst["closest_rays"] + wst["any_rays"]
    assert np.allclose(total, whole, rtol=1e-5, atol=1e-5)
    with pytest.raises(api.PtError):
        g.render(0, 1, block=(2, 2))
    g.close()


@pytest.mark.parametrize("make", [synth.cornell_box, lambda: synth.grid_plane(16, 96, 80), lambda: synth.random_soup(500, seed=8, width=100, height=76),
                                  synth.textured_room, lambda: synth.bathroom_synth(detail=1, width=96, height=96)])
def test_rendering_parity_synthetic_scenes(make):
    sd = make()
    o, g = orc.Scene(sd), api.Scene(sd)
    spp = 4 if sd.width > 256 else 16
    film, st = g.render(0, spp, traversal=api.TRAVERSE_EXACT)
    want, ost = o.render(0, spp)
    assert helpers.rel_mse(api.film_to_image(film), orc.film_to_image(want)) < 1e-4
    assert np.allclose(film[..., 3], want[..., 3], rtol=1e-5, atol=1e-6)
    assert abs(st["closest_rays"] - ost["closest_rays"]) <= 2e-3 * ost["closest_rays"] + 2
    # the ordered mode returns the same hit for every ray, so every path takes the same course: identical ray counts, and a
    # film that differs only by the order of float additions (this also covers the deferred-ray hand-off between workgroups)
    fast, fst = g.render(0, spp, traversal=api.TRAVERSE_FAST)
    assert (fst["closest_rays"], fst["any_rays"]) == (st["closest_rays"], st["any_rays"])
    assert np.allclose(fast, film, rtol=1e-4, atol=1e-5)
    small, sst = g.render(0, spp, traversal=api.TRAVERSE_FAST, samples_per_pass=3)
    assert (sst["closest_rays"], sst["any_rays"]) == (st["closest_rays"], st["any_rays"])
    g.close()


def test_textured_materials_and_vertex_normals_match_oracle():
    """Bitmap::sample (src/bitmap.cpp:146-166: clamp, bilinear, no flip), default uvs for faces without texture indices and
    faces with / without per-vertex normals: base-colour and normal AOVs against the oracle, and per-sample radiance."""
    sd = synth.textured_room()
    o, g = orc.Scene(sd), api.Scene(sd)
    for integ_g, integ_o in ((api.INTEGRATOR_ALBEDO, orc.INTEGRATOR_ALBEDO), (api.INTEGRATOR_NORMAL, orc.INTEGRATOR_NORMAL)):
        film, _ = g.render(0, 8, integrator=integ_g, traversal=api.TRAVERSE_FAST)
        want, _ = o.render(0, 8, integ_o)
        assert np.allclose(film, want, rtol=1e-4, atol=2e-4)
    img = api.film_to_image(g.render(0, 8, integrator=api.INTEGRATOR_ALBEDO)[0])
    assert len(np.unique(np.round(img.reshape(-1, 3), 2), axis=0)) > 20      # the textures really show up
    rng = np.random.default_rng(5)
    xy = np.stack([rng.integers(0, sd.width, 600), rng.integers(0, sd.height, 600)], 1).astype(np.int32)
    smp = rng.integers(0, 64, 600).astype(np.uint32)
    got = g.li(xy, smp)[0]
    want = np.array([o.li(int(x), int(y), int(s))["rgb"] for (x, y), s in zip(xy, smp)], np.float32)
    err = np.abs(got - want).max(1) / (np.abs(want).max(1) + 1e-3)
    assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %
    g.close()


def test_shading_sine_cosine_and_power_are_correctly_rounded():
    """The reference calls the C library's sinf / cosf (include/pt/common.h:103, 111) and powf (include/pt/common.h:108,
    src/material.cpp:34, 79, 108).  The kernels evaluate them in double and round once: the correctly rounded float, which
    is what glibc returns in all but 1.3 % (sine, cosine) and 0.1 % (power) of the cases; the single-precision device
    library would agree with glibc in 79 - 91 % only.  Checked against float64 results rounded to float32 and against the
    C library of this machine."""
    libm = C.CDLL("libm.so.6")
    for name in ("sinf", "cosf"):
        getattr(libm, name).restype, getattr(libm, name).argtypes = C.c_float, [C.c_float]
    libm.powf.restype, libm.powf.argtypes = C.c_float, [C.c_float, C.c_float]
    rng = np.random.default_rng(1)
    n = 100000
    u = rng.random(n, dtype=np.float32)
    phi = (np.float32(2 * np.pi) * u).astype(np.float32)          # the only angles the shading code forms
    phi[:8] = [0, np.pi, np.pi / 2, 2 * np.pi, 1.5 * np.pi, 1e-30, 7.9, 6.2831855]
    ns = rng.choice(np.array([0, 1, 2, 4, 6, 30, 100, 1397.325, 2000, 10000], np.float32), n)
    x = np.where(rng.random(n) < 0.5, u, np.cos(u * np.float32(1.5))).astype(np.float32)
    y = np.where(rng.random(n) < 0.5, ns, np.float32(1) / (ns + np.float32(1))).astype(np.float32)
    special = [(0, 3), (0, 0), (1, 1000), (0.5, 149), (0.5, 140), (0.5, 151), (1.0000001, 1e6), (1e-38, 1), (1e-45, 1), (0.999, 1e5),
               (2, 127.5), (2, 200), (-0.0, 3), (-0.0, 2.5), (-2, 2), (-2, 0.5), (np.inf, 2), (np.nan, 1), (0.3, np.nan), (0.3, -2), (0, -1)]
    x[:len(special)], y[:len(special)] = np.array(special, np.float32).T
    got = api.probe_math(np.concatenate([phi, x]), np.concatenate([np.zeros(n, np.float32), y]))
    sn, cs, pw = got[:n, 0], got[:n, 1], got[n:, 2]
    with np.errstate(all="ignore"):
        want = (np.sin(phi.astype(np.float64)).astype(np.float32), np.cos(phi.astype(np.float64)).astype(np.float32),
                np.power(x.astype(np.float64), y.astype(np.float64)).astype(np.float32))
    libc = (np.array([libm.sinf(float(v)) for v in phi], np.float32), np.array([libm.cosf(float(v)) for v in phi], np.float32),
            np.array([libm.powf(float(a), float(b)) for a, b in zip(x, y)], np.float32))

    def same(a, b):
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))

    for dev, cr, lc, floor in ((sn, want[0], libc[0], 0.98), (cs, want[1], libc[1], 0.98), (pw, want[2], libc[2], 0.998)):
        assert same(dev, cr).mean() > 0.9999            # float64 results are themselves off by up to half an ulp of double
        fin = np.isfinite(cr) & np.isfinite(dev)
        assert (np.abs(dev[fin].astype(np.float64) - cr[fin]) <= np.spacing(np.abs(cr[fin]))).all()   # never more than one ulp
        assert same(dev, lc).mean() > floor
    k = len(special)
    assert same(pw[:k], libc[2][:k]).all()              # zeros, ones, denormal results, overflow, negative bases, NaN


def test_mirror_chains_of_thousands_of_reflections_are_followed_to_the_end():
    """A specular bounce does not count towards maxBounce (src/integrator.cpp:84-87).  A mirror with interpolated normals
    reflects some rays below its own surface; such a path creeps along a crease of the pyramid for thousands of
    reflections (the oracle logs over 20 000 rays for single samples here) before it leaves and picks up its radiance.  The
    device loop must run as long: cutting the chain returns black for those samples."""
    sd = synth.textured_room(width=79, height=102)
    k = sd.mat_names.index("pyr")
    sd.mat_ks[k] = 1000.0                                                     # the reference's mirror (Ks > 999)
    o, g = orc.Scene(sd), api.Scene(sd)
    rng = np.random.default_rng(11)
    n = 3000
    xy = np.stack([rng.integers(0, sd.width, n), rng.integers(0, sd.height, n)], 1).astype(np.int32)
    smp = rng.integers(0, 2048, n).astype(np.uint32)
    res = [o.li(int(x), int(y), int(s), max_log=1) for (x, y), s in zip(xy, smp)]
    want = np.array([r["rgb"] for r in res], np.float32)
    n_rays = np.array([r["n_rays"] for r in res])
    assert (n_rays > 2000).mean() > 0.03                                      # far beyond 8 * maxBounce iterations
    for mode in (api.TRAVERSE_FAST, api.TRAVERSE_EXACT):
        got, _, st = g.li(xy, smp, traversal=mode)
        assert st["kernel_launches"]["finish"] == 1 and st["closest_rays"] > 0.4 * n_rays.sum()   # shadow rays are the other half
        err = np.abs(got - want).max(1) / (np.abs(want).max(1) + 1e-3)
        # 5 % of these samples are long chains.  A few of them end differently: the first diffuse bounce before the chain
        # differs in the last bit (device cos / sin against glibc's), the creep follows the same crease all the same, and
        # leaves it at another angle
        assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %
        assert abs(got.mean() - want.mean()) < 5e-3 * want.mean()
    g.close()


def test_tail_of_a_pass_in_one_launch_and_the_paths_that_outlive_it():
    """A render hands the thinned-out tail of a pass to one launch (k_finish) for up to maxBounce iterations per path; what
    is still alive then -- chains of specular bounces, here on the mirror pyramid -- goes back to the queues, the regular
    kernels carry on with it, and the last few chains end in a second such launch.  The list-mode entry point runs the same
    samples through the regular kernels alone (plus the final launch): both must trace the same paths, ray for ray, and the
    film must be the oracle's up to the few chains that leave their crease at another angle."""
    sd = synth.textured_room(width=40, height=50)
    sd.mat_ks[sd.mat_names.index("pyr")] = 1000.0
    o, g = orc.Scene(sd), api.Scene(sd)
    spp = 16
    film, st = g.render(0, spp, traversal=api.TRAVERSE_FAST)
    assert st["kernel_launches"]["finish"] == 2 and st["kernel_launches"]["shade"] > 10   # the regular kernels resumed after the first
    ys, xs, ss = np.meshgrid(np.arange(sd.height), np.arange(sd.width), np.arange(spp), indexing="ij")
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    _, _, st_li = g.li(xy, ss.ravel().astype(np.uint32), traversal=api.TRAVERSE_FAST)
    assert st_li["kernel_launches"]["finish"] == 1
    assert (st["closest_rays"], st["any_rays"]) == (st_li["closest_rays"], st_li["any_rays"])
    film_exact, st_exact = g.render(0, spp, traversal=api.TRAVERSE_EXACT)
    assert (st_exact["closest_rays"], st_exact["any_rays"]) == (st["closest_rays"], st["any_rays"])
    assert np.allclose(film_exact, film, rtol=1e-5, atol=1e-6)
    want, ost = o.render(0, spp, orc.INTEGRATOR_PATH)
    got_img, want_img = api.film_to_image(film), api.film_to_image(want)
    assert abs(got_img.mean() - want_img.mean()) < 1e-3 * want_img.mean()                   # measured 3e-5
    assert (np.abs(got_img - want_img).max(2) < 1e-3 * (want_img.max(2) + 1e-2)).mean() > 0.97  # measured 0.99
    assert abs(st["closest_rays"] - ost["closest_rays"]) < 5e-3 * ost["closest_rays"]       # measured 4e-4
    g.close()


def test_ultra_hd_frame_uses_the_4096_wide_sobol_grid():
    """3840x2160 needs the 2^12 pixel grid of SobolSampler (scale = roundUpPow2(max(W, H)), src/sampler.cpp:89-100): Sobol'
    values and film positions bit-exact, per-sample radiance and a one-sample AOV film against the oracle."""
    sd = synth.cornell_box(3840, 2160)
    o, g = orc.Scene(sd), api.Scene(sd)
    rng = np.random.default_rng(12)
    n = 3000
    pix = np.stack([rng.integers(0, sd.width, n), rng.integers(0, sd.height, n)], 1).astype(np.int32)
    pix[:4] = [[0, 0], [3839, 2159], [3839, 0], [2048, 2047]]
    smp = rng.integers(0, 4096, n).astype(np.uint32)
    dim = rng.integers(0, 1024, n).astype(np.uint32)
    want = np.array([orc.sobol_sample(orc.sobol_index(12, int(s), int(p[0]), int(p[1])), int(d)) for p, s, d in zip(pix, smp, dim)], np.float32)
    assert np.array_equal(g.sobol(pix, smp, dim).view(np.uint32), want.view(np.uint32))
    got, ps, _ = g.li(pix, smp)
    want_ps = np.array([o.pixel_sample(int(p[0]), int(p[1]), int(s)) for p, s in zip(pix[:400], smp[:400])])
    assert np.array_equal(ps[:400].view(np.uint32), want_ps.view(np.uint32))
    assert (np.floor(ps) == pix).all()
    want_li = np.array([o.li(int(p[0]), int(p[1]), int(s))["rgb"] for p, s in zip(pix, smp)], np.float32)
    err = np.abs(got - want_li).max(1) / (np.abs(want_li).max(1) + 1e-3)
    assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %
    film, st = g.render(4000, 4001, integrator=api.INTEGRATOR_NORMAL, traversal=api.TRAVERSE_FAST)
    want_film, ost = o.render(4000, 4001, orc.INTEGRATOR_NORMAL)
    assert film.shape == (2164, 3844, 4) and st["closest_rays"] == ost["closest_rays"] == 3840 * 2160
    assert np.allclose(film, want_film, rtol=1e-4, atol=2e-4)
    g.close()


@pytest.mark.parametrize("rho", [0.5, 0.8])
def test_furnace_radiance_is_the_geometric_series(rho):
    """The analytic check of the whole estimator on the GPU path: a closed, uniformly emitting diffuse enclosure returns
    sum_{k<32} rho^k (see the oracle's test of the same name)."""
    g = api.Scene(synth.furnace(rho))
    film, _ = g.render(0, 64, traversal=api.TRAVERSE_FAST)
    img = api.film_to_image(film)
    want = (1.0 - rho ** 32) / (1.0 - rho)
    assert abs(img.mean() - want) < 4e-3 * want
    g.close()


def test_work_counters_match_oracle_in_exact_mode(veach_gpu, veach_oracle):
    """Node and triangle records fetched by the reference-order traversal are the oracle's counts for primary rays."""
    film, st = veach_gpu.render(0, 1, integrator=api.INTEGRATOR_ALBEDO, traversal=api.TRAVERSE_EXACT, flags=api.FLAG_COUNT_WORK)
    _, ost = veach_oracle.render(0, 1, orc.INTEGRATOR_ALBEDO)
    assert st["nodes_fetched"]["generate"] == ost["nodes_visited"] and st["tris_fetched"]["generate"] == ost["tris_tested"]
    _, fst = veach_gpu.render(0, 1, integrator=api.INTEGRATOR_ALBEDO, traversal=api.TRAVERSE_FAST, flags=api.FLAG_COUNT_WORK)
    assert 0 < fst["nodes_fetched"]["generate"] < st["nodes_fetched"]["generate"]


# ------------------------------------------------------------------------------------------------ errors

def test_error_reporting(veach_gpu):
    lib = api.hip_lib()
    p = veach_gpu.params(5, 5)
    film = np.zeros((724, 1284, 4), np.float32)
    rc = lib.pt_hip_render(veach_gpu.h, C.byref(p), film.ctypes.data_as(api.c_f32p), None)
    assert rc == -1 and b"empty sample range" in lib.pt_hip_last_error()
    with pytest.raises(api.PtError):
        veach_gpu.intersect(np.zeros((4, 8), np.float32), mode=7)
    with pytest.raises(api.PtError):
        veach_gpu.li(np.array([[5000, 1]]), np.array([0]))
    with pytest.raises(api.PtError, match="samples_per_pass too large"):  # 3000 x 921 600 slots: beyond 2^31
        veach_gpu.render(0, 3000, samples_per_pass=3000)
    with pytest.raises(api.PtError, match="max_bounce"):
        veach_gpu.render(0, 1, max_bounce=64)
    film, st = veach_gpu.render(0, 1)                                     # the handle is still usable after errors
    assert st["paths"] == 1280 * 720
    bad = api.HostScene(synth.random_soup(5, seed=1))
    bad.desc.n_nodes = 1          # a root that claims children beyond the node array
    h = C.c_void_p()
    assert lib.pt_hip_scene_create(C.byref(bad.desc), C.byref(h)) < 0


@pytest.mark.gpu
def test_command_line_front_end_writes_the_same_images(tmp_path):
    """bin/PathTracer (the reference's flags, src/main.cpp:100-154) on an exported scene: its EXR outputs must equal
    what the same backend returns through the Python binding (same kernels, same sample ranges; neighbouring film regions
    overlap by the filter radius and are merged with float atomics, so the last bit may differ from run to run)."""
    import subprocess
    import exr_lite
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "bin", "PathTracer")
    assert os.path.exists(cli)
    sd = synth.cornell_box(96, 96)
    sd.export_dir(str(tmp_path), "cornell-box")
    r = subprocess.run([cli, "--no-gui", "-s", "4", "--backend", "hip", "--scenes", str(tmp_path), "--out", str(tmp_path), "cornell-box"],
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Rendering albedo map .. done." in r.stdout and "Rendering normal map .. done." in r.stdout and "Rendering .. done." in r.stdout
    g = api.Scene(sd)
    for stem, integ, spp in (("albedo", api.INTEGRATOR_ALBEDO, 32), ("normal", api.INTEGRATOR_NORMAL, 32), ("result", api.INTEGRATOR_PATH, 4)):
        film, _ = g.render(0, spp, integrator=integ, traversal=api.TRAVERSE_FAST)
        want = api.film_to_image(film)
        got = exr_lite.read_rgb(os.path.join(str(tmp_path), stem + ".exr"))
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-5, atol=1e-7), stem
    # two worker processes (they share the one GPU of a test box), split by sample range and by 32x32 blocks
    assert api.device_count() >= 1
    for split in ("spp", "tile"):
        out = tmp_path / split
        out.mkdir()
        r = subprocess.run([cli, "--no-gui", "-s", "4", "--gpus", "2", "--split", split, "--scenes", str(tmp_path), "--out", str(out),
                            "cornell-box"], capture_output=True, text=True, timeout=100)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "on 2 GPUs" in r.stdout
        for stem in ("albedo", "normal", "result"):
            one = exr_lite.read_rgb(os.path.join(str(tmp_path), stem + ".exr"))
            two = exr_lite.read_rgb(os.path.join(str(out), stem + ".exr"))
            assert np.allclose(one, two, rtol=1e-4, atol=1e-6), (split, stem)
    g.close()


# ------------------------------------------------------------------------------------------------ full-size workload

def test_full_size_library_workload_properties():
    """The bench workload at its full size (1.64 M triangles, 1280x720).  The oracle checks what it can do in seconds
    (30 k rays, one sample per pixel); beyond that the size-independent properties: the two traversal orders agree bit
    for bit on a million random rays, and films of disjoint sample ranges add up."""
    sd = synth.library_synth(detail=8)
    o, g = orc.Scene(sd), api.Scene(sd)
    closest, shadow = helpers.recorded_rays(o, 400, seed=11)
    prim = helpers.primary_rays(o, sample=2, step=9)
    rays = np.concatenate([prim, closest])
    want = o.intersect(rays)
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_EXACT), want).all()
    assert helpers.hits_equal(g.intersect(rays, api.TRAVERSE_FAST), want).all()
    occ, first = o.occluded(shadow)
    g_occ, g_first = g.occluded(shadow, api.TRAVERSE_EXACT)
    assert np.array_equal(g_occ, occ) and np.array_equal(g_first, first)
    assert np.array_equal(g.occluded(shadow, api.TRAVERSE_FAST)[0], occ)

    rng = np.random.default_rng(3)
    n = 1 << 20
    org = rng.uniform([-1.9, 0.05, -1.4], [1.9, 2.55, 1.4], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::1000, 1] = 0.0                                    # some axis-parallel rays (plain-division path)
    big = np.concatenate([org, d, np.zeros((n, 1), np.float32), np.full((n, 1), np.inf, np.float32)], 1).astype(np.float32)
    he, hf = g.intersect(big, api.TRAVERSE_EXACT), g.intersect(big, api.TRAVERSE_FAST)
    assert np.array_equal(he.view(np.uint32), hf.view(np.uint32))
    assert (he["prim"] >= 0).mean() > 0.99                # a closed room
    sub = slice(0, n, 53)
    assert helpers.hits_equal(he[sub], o.intersect(big[sub])).all()
    big[:, 7] = rng.uniform(0.05, 3.0, n).astype(np.float32)
    oe, _ = g.occluded(big, api.TRAVERSE_EXACT)
    assert np.array_equal(oe, g.occluded(big, api.TRAVERSE_FAST)[0])
    assert np.array_equal(oe[sub], o.occluded(big[sub])[0])

    a, sa = g.render(0, 1, traversal=api.TRAVERSE_FAST)
    b, sb = g.render(1, 3, traversal=api.TRAVERSE_FAST)
    c, sc = g.render(0, 3, traversal=api.TRAVERSE_FAST)
    assert sa["closest_rays"] + sb["closest_rays"] == sc["closest_rays"] and sa["any_rays"] + sb["any_rays"] == sc["any_rays"]
    assert np.allclose(a + b, c, rtol=1e-5, atol=1e-5)
    e, se = g.render(0, 3, traversal=api.TRAVERSE_EXACT)
    assert (se["closest_rays"], se["any_rays"]) == (sc["closest_rays"], sc["any_rays"])      # same hits, same paths
    assert np.allclose(e, c, rtol=1e-4, atol=1e-5)
    want, ost = o.render(0, 1, nthreads=16)
    assert helpers.rel_mse(api.film_to_image(a), orc.film_to_image(want)) < 1e-4
    assert abs(sa["closest_rays"] - ost["closest_rays"]) <= 1e-4 * ost["closest_rays"]
    g.close()


def test_full_size_bathroom_workload_properties():
    """The textured stand-in at its full size (0.44 M triangles, 768x768, a mirror, per-vertex normals): both traversal
    orders agree bit for bit on a million random rays inside the room, a sample of them equals the oracle's hits, and a
    render takes the same paths in both modes."""
    sd = synth.bathroom_synth()
    o, g = orc.Scene(sd), api.Scene(sd)
    lo, hi = sd.verts.min(0), sd.verts.max(0)
    rng = np.random.default_rng(17)
    n = 1 << 20
    org = rng.uniform(lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo), (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::997, 0] = 0.0                                     # some axis-parallel rays (plain-division path)
    rays = np.concatenate([org, d, np.zeros((n, 1), np.float32), np.full((n, 1), np.inf, np.float32)], 1).astype(np.float32)
    he, hf = g.intersect(rays, api.TRAVERSE_EXACT), g.intersect(rays, api.TRAVERSE_FAST)
    assert np.array_equal(he.view(np.uint32), hf.view(np.uint32))
    assert (he["prim"] >= 0).mean() > 0.9
    sub = slice(0, n, 61)
    assert helpers.hits_equal(he[sub], o.intersect(rays[sub])).all()
    rays[:, 7] = rng.uniform(0.02, 1.0, n).astype(np.float32) * float(np.linalg.norm(hi - lo))
    oe, _ = g.occluded(rays, api.TRAVERSE_EXACT)
    assert np.array_equal(oe, g.occluded(rays, api.TRAVERSE_FAST)[0])
    assert np.array_equal(oe[sub], o.occluded(rays[sub])[0])
    assert 0.05 < oe.mean() < 0.95
    f, sf = g.render(0, 4, traversal=api.TRAVERSE_FAST)
    e, se = g.render(0, 4, traversal=api.TRAVERSE_EXACT)
    assert (se["closest_rays"], se["any_rays"]) == (sf["closest_rays"], sf["any_rays"])
    assert np.allclose(e, f, rtol=1e-4, atol=1e-5)
    g.close()


def test_fifty_million_triangle_ultra_hd_workload_properties():
    """The largest configuration in shape: the library stand-in tessellated to 49 M triangles (a 98 M-node tree, 34 levels)
    at 3840x2160 (the sampler's 4096-wide grid).  The tree is built by the GPU builder (api.Scene's default) and, once more,
    by the host builder: the two must be identical node for node at this size too.  The oracle traverses the same tree (its
    own single-threaded builder would take minutes here; that it produces the host builder's tree is checked on smaller
    scenes in test_host.py).  Checked: both traversal orders agree bit for bit on a million random rays and on
    their occlusion flags, every 97th of them equals the oracle's answer, per-sample radiance agrees with the oracle, films
    of disjoint sample ranges add up, and both modes trace the same paths."""
    sd = synth.library_synth(detail=44, width=3840, height=2160)
    assert sd.n_tris > 49_000_000
    g = api.Scene(sd)
    assert g.host.depth >= 30
    host_nodes, host_ids, host_depth = api.build_bvh(sd.verts, sd.vi)
    assert host_depth == g.host.depth and np.array_equal(host_ids, g.host.prim_ids)
    assert np.array_equal(host_nodes.view(np.uint32), g.host.nodes.view(np.uint32))
    del host_nodes, host_ids
    o = orc.Scene(sd, bvh=(g.host.nodes, g.host.prim_ids))
    rng = np.random.default_rng(44)
    n = 1 << 20
    org = rng.uniform([-1.9, 0.05, -1.4], [1.9, 2.55, 1.4], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::1000, 2] = 0.0                                    # some axis-parallel rays (plain-division path)
    rays = np.concatenate([org, d, np.zeros((n, 1), np.float32), np.full((n, 1), np.inf, np.float32)], 1).astype(np.float32)
    he, hf = g.intersect(rays, api.TRAVERSE_EXACT), g.intersect(rays, api.TRAVERSE_FAST)
    assert np.array_equal(he.view(np.uint32), hf.view(np.uint32))
    assert (he["prim"] >= 0).mean() > 0.99
    sub = slice(0, n, 97)
    assert helpers.hits_equal(he[sub], o.intersect(rays[sub])).all()
    rays[:, 7] = rng.uniform(0.05, 3.0, n).astype(np.float32)
    oe, first = g.occluded(rays, api.TRAVERSE_EXACT)
    assert np.array_equal(oe, g.occluded(rays, api.TRAVERSE_FAST)[0])
    occ_w, first_w = o.occluded(rays[sub])
    assert np.array_equal(oe[sub], occ_w) and np.array_equal(first[sub], first_w)
    assert 0.05 < oe.mean() < 0.95

    pix = np.stack([rng.integers(0, sd.width, 1500), rng.integers(0, sd.height, 1500)], 1).astype(np.int32)
    smp = rng.integers(0, 4096, 1500).astype(np.uint32)
    want = np.array([o.li(int(p[0]), int(p[1]), int(s))["rgb"] for p, s in zip(pix, smp)], np.float32)
    got = g.li(pix, smp, traversal=api.TRAVERSE_FAST)[0]
    err = np.abs(got - want).max(1) / (np.abs(want).max(1) + 1e-3)
    assert (err < 1e-3).mean() >= 0.998 and (err == 0).mean() >= 0.98   # identical in every bit: measured 99.3 - 100 %

    a, sa = g.render(0, 1, traversal=api.TRAVERSE_FAST)
    b, sb = g.render(1, 2, traversal=api.TRAVERSE_FAST)
    c, sc = g.render(0, 2, traversal=api.TRAVERSE_FAST)
    assert sa["paths"] == 3840 * 2160
    assert sa["closest_rays"] + sb["closest_rays"] == sc["closest_rays"] and sa["any_rays"] + sb["any_rays"] == sc["any_rays"]
    assert np.allclose(a + b, c, rtol=1e-5, atol=1e-5)
    e, se = g.render(0, 1, traversal=api.TRAVERSE_EXACT)
    assert (se["closest_rays"], se["any_rays"]) == (sa["closest_rays"], sa["any_rays"])        # same hits, same paths
    assert np.allclose(e, a, rtol=1e-4, atol=1e-5)
    g.close()


# ------------------------------------------------------------------------------------------------ tree build on the GPU

@pytest.mark.parametrize("make", [lambda: synth.random_soup(1, seed=1), lambda: synth.random_soup(2, seed=5), lambda: synth.random_soup(7, seed=2),
                                  lambda: synth.random_soup(3000, seed=3), lambda: synth.grid_plane(16), synth.cornell_box, synth.textured_room,
                                  lambda: synth.library_synth(detail=1), lambda: synth.library_synth(detail=4),
                                  lambda: synth.bathroom_synth(detail=2, width=64, height=64), "veach"])
def test_gpu_tree_build_equals_the_host_builder_node_for_node(make, veach_sd):
    """pt_hip_build_bvh (all ranges of a level split at once with segmented scans, numbering replayed) against
    pt_host_build_bvh (the reference's sequential loop, itself equal to the oracle's builder: tests/test_host.py): every
    node's box, child index / leaf range, the triangle order and the depth, bit for bit.  grid_plane and the tessellated
    rooms are full of equal centroid coordinates and equally cheap split candidates."""
    sd = veach_sd if make == "veach" else make()
    want_nodes, want_ids, want_depth = api.build_bvh(sd.verts, sd.vi)
    nodes, ids, depth = api.build_bvh_gpu(sd.verts, sd.vi)
    assert len(nodes) == len(want_nodes) and depth == want_depth
    assert np.array_equal(ids, want_ids)
    assert np.array_equal(nodes.view(np.uint32), want_nodes.view(np.uint32))


@pytest.mark.parametrize("seed", [1, 17, 59])
def test_gpu_tree_build_on_coordinates_full_of_ties_and_signed_zeros(seed):
    """Vertices snapped to a coarse grid: equal centroids (every sort tie), flat and collapsed triangles, equally cheap
    split candidates, and box coordinates that are -0 in one triangle and +0 in the next.  The reference accumulates a box
    with the running value on the left of std::min / std::max, in the order of the axis-0 index array, so which zero a box
    ends up with is part of the tree; the device builder has to reproduce it (it once took the children's boxes from the
    split axis' scans)."""
    rng = np.random.default_rng(seed)
    sd = synth.random_soup(int(rng.integers(2000, 30000)), seed=seed, extent=float(rng.uniform(2, 30)), size=float(rng.uniform(0.3, 3.0)))
    verts = (np.round(sd.verts * 4) / 4).astype(np.float32)
    verts[verts == 0] = np.where(rng.random((verts == 0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
    want_nodes, want_ids, want_depth = api.build_bvh(verts, sd.vi)
    nodes, ids, depth = api.build_bvh_gpu(verts, sd.vi)
    assert len(nodes) == len(want_nodes) and depth == want_depth
    assert np.array_equal(ids, want_ids)
    assert np.array_equal(nodes.view(np.uint32), want_nodes.view(np.uint32))
    assert np.signbit(want_nodes["hi"][want_nodes["hi"] == 0]).any() and not np.signbit(want_nodes["hi"][want_nodes["hi"] == 0]).all()


def test_scene_description_without_a_tree_is_built_on_the_device():
    """pt_hip_scene_create with nodes == NULL builds the reference's tree itself: same hits, same film as with the host
    builder's tree handed over."""
    sd = synth.library_synth(detail=1)
    a, b = api.Scene(sd, builder="host"), api.Scene(sd, builder="deferred")
    assert len(b.host.nodes) == 0
    o = orc.Scene(sd)
    rays = helpers.primary_rays(o, sample=0, step=8)
    for mode in (api.TRAVERSE_EXACT, api.TRAVERSE_FAST):
        ha, hb = a.intersect(rays, mode), b.intersect(rays, mode)
        assert np.array_equal(ha.view(np.uint32), hb.view(np.uint32))
    assert helpers.hits_equal(hb, o.intersect(rays)).all()
    fa, sa = a.render(0, 2, traversal=api.TRAVERSE_FAST)
    fb, sb = b.render(0, 2, traversal=api.TRAVERSE_FAST)
    assert (sa["closest_rays"], sa["any_rays"]) == (sb["closest_rays"], sb["any_rays"]) and np.allclose(fa, fb, rtol=1e-5, atol=1e-6)
    a.close()
    b.close()


# ------------------------------------------------------------------------------------------------ several handles, one process

def test_one_process_drives_several_handles():
    """pt_hip_scene_create_on + pt_hip_render_multi: three handles of one scene (on a one-GPU box they share the GPU; on a
    node each sits on its own), one host thread per handle inside the call.  The frame split by sample number (5 samples
    over 3 handles: shares of 2, 2, 1) and by 32x32 blocks equals the frame of a single handle; more handles than samples
    leaves the surplus films empty; batched queries on two handles from two threads at once give the single-thread answers."""
    import threading
    sd = synth.cornell_box(200, 136)
    n_dev = api.device_count()
    g = [api.Scene(sd, device=k % n_dev) for k in range(3)]
    whole, wst = g[0].render(0, 5, traversal=api.TRAVERSE_FAST)
    film, st = api.render_multi(g, 0, 5, traversal=api.TRAVERSE_FAST)
    assert st["paths"] == wst["paths"] == 200 * 136 * 5
    assert (st["closest_rays"], st["any_rays"]) == (wst["closest_rays"], wst["any_rays"])
    assert np.allclose(film, whole, rtol=1e-5, atol=1e-5)
    film, st = api.render_multi(g, 0, 5, traversal=api.TRAVERSE_FAST, block=(0, 3))
    assert st["paths"] == wst["paths"] and np.allclose(film, whole, rtol=1e-5, atol=1e-5)
    film, st = api.render_multi(g, 3, 5, traversal=api.TRAVERSE_FAST)          # 2 samples, 3 handles
    two, _ = g[1].render(3, 5, traversal=api.TRAVERSE_FAST)
    assert st["paths"] == 200 * 136 * 2 and np.allclose(film, two, rtol=1e-5, atol=1e-5)
    with pytest.raises(api.PtError, match="block_count must equal"):
        api.render_multi(g, 0, 5, block=(0, 2))
    o = orc.Scene(sd)
    rays = helpers.primary_rays(o, sample=0, step=2)
    want = g[0].intersect(rays, api.TRAVERSE_FAST)
    got = [None, None]

    def work(k):
        for _ in range(5):
            got[k] = g[1 + k].intersect(rays, api.TRAVERSE_FAST)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32)) and np.array_equal(got[1].view(np.uint32), want.view(np.uint32))
    for s in g:
        s.close()
