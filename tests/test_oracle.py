"""The oracle against the reference's own artefacts and against first principles.

Pins: scenes/veach-mis/normal.exr and albedo.exr were written by the real reference binary (32 spp Sobol', Gaussian
film); tests/golden/veach_mis_aov.npz holds every 4th row/column of them.  scenes/veach-mis/result.png is the reference's
path-traced image of the same scene; tests/golden/veach_mis_result.npz holds its 8x8 block means (linear values).  Everything else here
checks the oracle's restated functions against closed forms or brute force.
"""
import os

import numpy as np
import pytest

import orc
from conftest import GOLDEN
from pt_hip import synth

REF_SCENE = "/root/reference/scenes/veach-mis"


@pytest.fixture(scope="module")
def aov_renders(veach_oracle):
    out = {}
    for name, integ in (("normal", orc.INTEGRATOR_NORMAL), ("albedo", orc.INTEGRATOR_ALBEDO)):
        film, stats = veach_oracle.render(0, 32, integ)
        out[name] = orc.film_to_image(film)
        assert stats["paths"] == 1280 * 720 * 32
    return out


@pytest.mark.parametrize("name", ["normal", "albedo"])
def test_aov_matches_reference_binary_output(aov_renders, name):
    """Camera, Sobol' pixel jitter, closest hit, normal interpolation, base colour and film reconstruction agree with
    the images the reference binary wrote.  Tolerance: the reference was built by another compiler against Eigen, so
    single samples on silhouette edges may flip between hit and miss; everything else agrees to float rounding."""
    z = np.load(os.path.join(GOLDEN, "veach_mis_aov.npz"))
    stride = int(z["stride"])
    ref = z[name]
    img = aov_renders[name][::stride, ::stride]
    d = np.abs(img - ref).max(-1)
    assert d.mean() < 5e-6
    assert (d > 1e-5).mean() < 2e-3
    assert (d > 1e-2).mean() < 2e-4
    assert d.max() < 0.1


@pytest.mark.skipif(not os.path.exists(REF_SCENE), reason="reference tree not present")
@pytest.mark.parametrize("name", ["normal", "albedo"])
def test_aov_full_image_against_reference_exr(aov_renders, name):
    from exr_lite import read_rgb
    ref = read_rgb(os.path.join(REF_SCENE, name + ".exr"))
    d = np.abs(aov_renders[name] - ref).max(-1)
    assert d.mean() < 5e-6
    assert (d > 1e-5).mean() < 2e-3
    assert d.max() < 0.1


def test_path_traced_image_matches_reference_png(veach_oracle):
    """The whole MIS integrator against the reference binary's own output: scenes/veach-mis/result.png (8-bit sRGB, few
    samples per pixel by its noise; tests/golden/veach_mis_result.npz keeps the means of its 8x8 pixel blocks after decoding
    to linear values).  16 samples per pixel of the oracle go through the same encoding and decoding.  The light the two
    images hold in total agrees to 0.5 % (measured 0.11 %), and so does every brightness band to a few per cent -- the
    glossy plates, the floor, the wall and the light sources all cover thousands of blocks, so a wrong BRDF, pdf, MIS
    weight or light sampling density would show in one band or another.  (Blocks are put into bands by the mean of the
    two images, so that neither image's noise decides the band; the nearly black band is left out: truncation to 8 bits
    makes the noisier image brighter there.)"""
    import helpers
    want = np.load(os.path.join(GOLDEN, "veach_mis_result.npz"))["linear"].astype(np.float64)
    film, _ = veach_oracle.render(0, 16)
    got = helpers.png_block_means(orc.film_to_image(film))
    assert abs(got.sum() / want.sum() - 1.0) < 0.005
    assert np.abs(got - want).mean() < 0.04 * want.mean()          # block by block: Monte Carlo noise of both images
    lum = (0.5 * (got + want)).mean(-1)
    for lo, hi, tol in ((0.002, 0.005, 0.06), (0.005, 0.01, 0.08), (0.01, 0.02, 0.03), (0.02, 0.05, 0.03), (0.05, 0.1, 0.01),
                        (0.1, 0.2, 0.01), (0.2, 0.5, 0.01), (0.5, 1.01, 0.01)):
        m = (lum >= lo) & (lum < hi)
        assert m.mean() > 0.02
        assert abs(got[m].mean() / want[m].mean() - 1.0) < tol, (lo, hi)


@pytest.mark.skipif(not os.path.exists(REF_SCENE), reason="reference tree not present")
def test_scene_fixture_matches_reference_files(veach_sd):
    from pt_hip.scene import load_scene_dir
    sd = load_scene_dir(REF_SCENE)
    for k in ("verts", "normals", "uvs", "vi", "ni", "ti", "mat_id", "mat_kd", "mat_ks", "mat_ns"):
        assert np.array_equal(getattr(sd, k), getattr(veach_sd, k)), k
    assert sd.light_infos == veach_sd.light_infos and sd.mat_names == veach_sd.mat_names
    assert (sd.width, sd.height) == (1280, 720) and sd.n_tris == 3092 and len(sd.lights()[0]) == 3040


# ------------------------------------------------------------------------------------------------ geometry

def _one_triangle():
    sd = synth.random_soup(1, seed=0)
    sd.verts[:] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    return orc.Scene(sd)


def test_triangle_known_answers():
    s = _one_triangle()
    hit, b, t = s.tri_intersect(0, [0.25, 0.25, -1, 0, 0, 1, 0, 10])
    assert hit and t == 1.0 and np.allclose(b, [0.5, 0.25, 0.25])
    # inclusive at t == tmax and t == tmin (src/shape.cpp:88)
    assert s.tri_intersect(0, [0.25, 0.25, -1, 0, 0, 1, 0, 1.0])[0]
    assert s.tri_intersect(0, [0.25, 0.25, -1, 0, 0, 1, 1.0, 5])[0]
    assert not s.tri_intersect(0, [0.25, 0.25, -1, 0, 0, 1, 0, 0.999])[0]
    # inclusive on the edges u = 0, v = 0 and u + v = 1
    assert s.tri_intersect(0, [0.0, 0.5, -1, 0, 0, 1, 0, 10])[0]
    assert s.tri_intersect(0, [0.5, 0.0, -1, 0, 0, 1, 0, 10])[0]
    assert s.tri_intersect(0, [0.5, 0.5, -1, 0, 0, 1, 0, 10])[0]
    assert not s.tri_intersect(0, [0.6, 0.5, -1, 0, 0, 1, 0, 10])[0]
    # |det| < 1e-5 is rejected: ray in the plane, and a tiny triangle seen head-on
    assert not s.tri_intersect(0, [-1, 0.2, 0, 1, 0, 0, 0, 10])[0]
    # back faces are hit too
    assert s.tri_intersect(0, [0.25, 0.25, 1, 0, 0, -1, 0, 10])[0]


def test_aabb_known_answers():
    L = orc.lib()
    box = np.array([0, 0, 0, 1, 1, 1], np.float32)

    def hit(ray):
        r = np.array(ray, np.float32)
        return bool(L.orc_aabb_intersect(box.ctypes.data_as(orc.c_f32p), r.ctypes.data_as(orc.c_f32p)))

    assert hit([-1, 0.5, 0.5, 1, 0, 0, 0, 10])
    assert not hit([-1, 0.5, 0.5, 1, 0, 0, 0, 1.0])        # tmin < max_dis is strict (src/aabb.cpp:34)
    assert hit([-1, 0.5, 0.5, 1, 0, 0, 0, 1.0001])
    assert not hit([2, 0.5, 0.5, 1, 0, 0, 0, 10])          # behind: tmax > min_dis fails
    assert not hit([-1, 2.0, 0.5, 1, 0, 0, 0, 10])         # zero direction component outside the slab -> +-inf, rejected
    assert hit([0.5, 0.5, 0.5, 1, 0, 0, 0, 10])            # origin inside
    # the z slab only rejects on overlap, it never tightens the interval (src/aabb.cpp:31-34): along +z the x/y
    # interval is (-inf, +inf), so neither max_dis nor a box behind the origin is noticed
    assert hit([0.5, 0.5, -1, 0, 0, 1, 0, 0.5])
    assert hit([0.5, 0.5, 2, 0, 0, 1, 0, 10])
    assert hit([0.5, 0.5, -5, 0.01, 0.01, 1, 0, 4.9])


@pytest.mark.parametrize("make", [lambda: synth.random_soup(300, seed=3), lambda: synth.grid_plane(8), synth.cornell_box])
def test_bvh_invariants(make):
    sd = make()
    s = orc.Scene(sd)
    boxes, first, count, prim = s.bvh()
    n = sd.n_tris
    assert sorted(prim.tolist()) == list(range(n))
    leaves = count > 0
    assert count.max() <= 8 and count[leaves].sum() == n
    covered = np.zeros(n, int)
    for f, c in zip(first[leaves], count[leaves]):
        covered[int(f):int(f + c)] += 1
    assert (covered == 1).all()

    def area(b):
        d = b[3:] - b[:3]
        return (d[0] + d[1]) * d[2] + d[0] * d[1]

    v = sd.verts[sd.vi - 1]
    for i in np.nonzero(~leaves)[0]:
        c0, c1 = int(first[i]), int(first[i]) + 1
        assert c1 < len(boxes)
        assert area(boxes[c0]) >= area(boxes[c1])                      # larger child first (src/bvh.cpp:137-144)
        assert (np.minimum(boxes[c0][:3], boxes[c1][:3]) == boxes[i][:3]).all()
        assert (np.maximum(boxes[c0][3:], boxes[c1][3:]) == boxes[i][3:]).all()
    for i in np.nonzero(leaves)[0]:
        tri = v[prim[int(first[i]):int(first[i] + count[i])].astype(int)].reshape(-1, 3)
        assert (tri.min(0) == boxes[i][:3]).all() and (tri.max(0) == boxes[i][3:]).all()


def test_closest_hit_equals_brute_force():
    sd = synth.random_soup(200, seed=11)
    s = orc.Scene(sd)
    rng = np.random.default_rng(2)
    o = rng.uniform(-8, 8, (400, 3))
    d = rng.normal(size=(400, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.zeros((400, 1)), np.full((400, 1), np.inf)], 1).astype(np.float32)
    hits = s.intersect(rays)
    for r, h in zip(rays, hits):
        best = (np.inf, -1)
        for p in range(sd.n_tris):
            ok, _, t = s.tri_intersect(p, r)
            if ok and t <= best[0]:
                best = (t, p)
        assert h["prim"] == best[1]
        if best[1] >= 0:
            assert h["t"] == np.float32(best[0])
    occ, first = s.occluded(rays)
    assert ((hits["prim"] >= 0) == (occ == 1)).all()


def test_surface_completion(veach_oracle, veach_sd):
    ray = veach_oracle.camera_ray(640.5, 360.5)
    hit, surf = veach_oracle.intersect(ray[None], surfaces=True)
    assert hit["prim"][0] >= 0
    p = ray[:3] + hit["t"][0] * ray[3:6]
    assert np.allclose(surf["p"][0], p, atol=1e-4)
    assert abs(np.linalg.norm(surf["n"][0]) - 1) < 1e-6 and abs(np.linalg.norm(surf["ng"][0]) - 1) < 1e-6


# ------------------------------------------------------------------------------------------------ camera / film

def test_camera_centre_ray_points_at_lookat(veach_oracle, veach_sd):
    r = veach_oracle.camera_ray(veach_sd.width / 2, veach_sd.height / 2)
    want = np.array(veach_sd.lookat) - np.array(veach_sd.eye)
    want /= np.linalg.norm(want)
    assert np.allclose(r[3:6], want, atol=1e-6)
    assert np.allclose(r[:3], veach_sd.eye)
    # on the axis the range is [near, far] * proj_nume, and proj_nume = 1 / z(depth 0.5) = (far + near) / (2 far near) ~ 50
    # for near 0.01, far 10000 (src/camera.cpp:76-79, 86-88)
    pn = veach_oracle.camera()[2]
    assert np.isclose(pn, 50.0, rtol=1e-4)
    assert np.isclose(r[6], 0.01 * pn, rtol=1e-5) and np.isclose(r[7], 10000.0 * pn, rtol=1e-5)


def test_filter_table():
    f = orc.filter_table()
    assert f.shape == (33,) and f[32] == 0
    x = 2.0 * np.arange(32) / 32
    want = np.maximum(0, np.exp(-2 * x * x) - np.exp(-8.0))
    assert np.allclose(f[:32], want, rtol=1e-6, atol=1e-9)


def test_film_put_single_sample():
    film = np.zeros((12, 12, 4), np.float32)
    orc.film_put(film, [3.25, 4.5], [1.0, 2.0, 3.0])
    f = orc.filter_table()
    ys, xs = np.nonzero(film[..., 3])
    lx, ly = 3.25 + 2, 4.5 + 2
    assert xs.min() >= np.ceil(lx - 2) and xs.max() <= np.floor(lx + 2)
    for y, x in zip(ys, xs):
        w = f[int(abs(x - lx) * 16)] * f[int(abs(y - ly) * 16)]
        assert np.isclose(film[y, x, 3], w, rtol=1e-6)
        assert np.allclose(film[y, x, :3], np.array([1, 2, 3]) * w, rtol=1e-6)
    # samples outside the image are dropped (src/block.cpp:67-71)
    film2 = np.zeros((12, 12, 4), np.float32)
    orc.film_put(film2, [-0.1, 3.0], [1, 1, 1])
    orc.film_put(film2, [8.0, 3.0], [1, 1, 1])
    assert not film2.any()


def test_render_is_thread_count_independent():
    sd = synth.random_soup(60, seed=5, width=70, height=50)
    s = orc.Scene(sd)
    a, _ = s.render(0, 2, nthreads=1)
    b, _ = s.render(0, 2, nthreads=5)
    assert np.array_equal(a, b)
    # sample ranges add up: [0,2) + [2,4) == [0,4) up to summation order
    c, _ = s.render(2, 4, nthreads=3)
    d, _ = s.render(0, 4, nthreads=3)
    assert np.allclose(a + c, d, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ shading models

def test_brdf_sample_consistent_with_eval_and_pdf(veach_oracle, veach_sd):
    rng = np.random.default_rng(0)
    n_checked = 0
    for _ in range(400):
        prim = int(rng.integers(veach_sd.n_tris))
        b = rng.dirichlet([1, 1, 1]).astype(np.float32)
        wo = rng.normal(size=3)
        wo /= np.linalg.norm(wo)
        spec, wi, pdf, f = veach_oracle.brdf_sample(prim, b, wo, rng.random(), rng.random(2))
        assert not spec
        if pdf == 0:
            continue
        n_checked += 1
        assert abs(np.linalg.norm(wi) - 1) < 1e-5
        assert np.isclose(veach_oracle.brdf_pdf(prim, b, wo, wi), pdf, rtol=1e-5)
        assert np.allclose(veach_oracle.brdf_eval(prim, b, wo, wi), f, rtol=1e-4, atol=1e-7)
    assert n_checked > 100


def test_phong_pdf_integrates_to_one(veach_oracle, veach_sd):
    """Monte Carlo: E_uniform_sphere[pdf] * 4 pi == 1 when the lobe lies in the upper hemisphere (normal incidence)."""
    prim = int(np.nonzero(veach_sd.mat_id == 4)[0][0])  # Ns = 100 plate
    b = np.array([1 / 3, 1 / 3, 1 / 3], np.float32)
    _, surf = veach_oracle.intersect(np.zeros((1, 8), np.float32), surfaces=True)
    rng = np.random.default_rng(1)
    # find the shading normal by sampling the BRDF many times: mean direction ~ normal for wo = n; use geometry instead
    v = veach_sd.verts[veach_sd.vi[prim] - 1]
    n = np.cross(v[0] - v[2], v[1] - v[2])
    n /= np.linalg.norm(n)
    w = rng.normal(size=(40000, 3))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    w = w[(w @ n) > 0]
    est = np.mean([veach_oracle.brdf_pdf(prim, b, n, wi) for wi in w[:8000]]) * 2 * np.pi
    assert abs(est - 1) < 0.08


def test_light_sample_geometry(veach_oracle, veach_sd):
    rng = np.random.default_rng(3)
    floor = int(np.nonzero(veach_sd.mat_id == 0)[0][0])
    b = np.array([0.3, 0.3, 0.4], np.float32)
    got = 0
    for _ in range(200):
        li = int(rng.integers(veach_oracle.n_lights))
        s = veach_oracle.light_sample(li, floor, b, rng.random(2))
        if s["pdfDir"] == 0:
            assert not s["L"].any()
            continue
        got += 1
        assert abs(np.linalg.norm(s["wi"]) - 1) < 1e-5 and s["pdfDir"] > 0 and s["pdfArea"] > 0
    assert got > 20


def test_li_ray_log_and_dimensions(veach_oracle):
    """Each bounce draws 3 (light) + 3 (BRDF) dimensions, plus one for roulette from the third bounce on; the sampler
    starts at dimension 2 (src/sampler.cpp:96-100, src/integrator.cpp:72-103)."""
    r = veach_oracle.li(640, 500, 7, max_log=80)
    rays = r["rays"]
    assert r["n_rays"] == len(rays) and rays[0, 8] == 0
    n_closest = int((rays[:, 8] == 0).sum())
    assert 2 + 6 * (n_closest - 1) <= r["dims"] <= 2 + 7 * n_closest
    assert np.isfinite(r["rgb"]).all() and (r["rgb"] >= 0).all()


REF_TEXTURES = "/root/reference/scenes/bathroom/textures"


@pytest.mark.parametrize("which", ["procedural", "reference files"])
def test_texture_lookup_against_numpy_restatement(which):
    """Base-colour AOV samples on the textured floor / wall against an independent numpy version of Bitmap::sample
    (src/bitmap.cpp:146-166: clamp to [0,1], s = max(u*cols - 0.5, 0), bilinear with clamped upper texel, no flip) fed
    with the uv interpolation of Intersection::complete (src/shape.cpp:163-165).  Second case: the bathroom stand-in with
    the reference's own texture files (JPG / PNG, non-square), decoded as Bitmap::load does.  Parity of this row stays
    unpinned by the reference: it holds no reproducible image of a textured scene (the bathroom mesh is not in its tree)."""
    from pt_hip import synth
    if which == "procedural":
        sd = synth.textured_room()
    else:
        if not os.path.isdir(REF_TEXTURES):
            pytest.skip("reference tree not present")
        sd = synth.bathroom_synth(detail=1, width=96, height=96, texture_dir=REF_TEXTURES)
        assert sd.textures[1].shape == (1454, 2048, 3) and sd.textures[0].shape == (1024, 996, 3)
    o = orc.Scene(sd)

    def sample(tex, uv):
        h, w = tex.shape[:2]
        u, v = np.clip(uv, 0.0, 1.0).astype(np.float32)
        s, t = max(np.float32(u * w) - np.float32(0.5), np.float32(0)), max(np.float32(v * h) - np.float32(0.5), np.float32(0))
        x0, y0 = int(s), int(t)
        x1, y1 = min(x0 + 1, w - 1), min(y0 + 1, h - 1)
        a, b = float(s - x0), float(t - y0)
        mix = lambda p, q, f: p.astype(np.float64) * (1.0 - f) + (q * np.float32(f)).astype(np.float64)
        top, bot = mix(tex[y0, x0], tex[y0, x1], a).astype(np.float32), mix(tex[y1, x0], tex[y1, x1], a).astype(np.float32)
        return mix(top, bot, b).astype(np.float32)

    checked = 0
    for px in range(0, sd.width, 5):
        for py in range(0, sd.height, 5):
            sx, sy = o.pixel_sample(px, py, 3)
            ray = o.camera_ray(sx, sy)
            hit = o.intersect(ray[None])[0]
            prim = int(hit["prim"])
            if prim < 0 or sd.mat_tex[sd.mat_id[prim]] < 0:
                continue
            b = hit["bary"]
            if sd.ti[prim, 0]:
                c = sd.uvs[sd.ti[prim] - 1]
            else:
                c = np.array([(0, 0), (1, 0), (1, 1)], np.float32)
            uv = b[0] * c[0] + (b[1] * c[1] + b[2] * c[2])
            want = sample(sd.textures[sd.mat_tex[sd.mat_id[prim]]], uv)
            got = o.li(px, py, 3, orc.INTEGRATOR_ALBEDO)["rgb"]
            assert np.allclose(got, want, rtol=2e-5, atol=2e-6), (px, py, prim, got, want)
            checked += 1
    assert checked > 50


def test_reference_texture_files_are_decoded_like_bitmap_load(tmp_path):
    """Bitmap::load (src/bitmap.cpp:68-93): 8-bit samples / 255 then Color3f::toLinearRGB (include/pt/color.h:70-83), rows
    top to bottom.  The decoder of the Python loader against a scalar restatement on the file's raw samples, through
    load_scene_dir on a scene whose .mtl names the reference's own files with map_Kd, and through the PPM that
    tools/convert_textures.py writes for the C++ front end."""
    import math
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    from pt_hip import scene, synth
    if not os.path.isdir(REF_TEXTURES):
        pytest.skip("reference tree not present")
    sd = synth.cornell_box(32, 32)
    tmp_path = tmp_path / "room"                   # export_dir writes <root>/<name>/<name>.*
    sd.export_dir(str(tmp_path.parent), "room")
    with open(tmp_path / "room.mtl", "a") as f:
        f.write("\nnewmtl Tiled\nKd 0 0 0\nmap_Kd textures/floor_tiles.png\nnewmtl Wooden\nKd 0 0 0\nmap_Kd textures/wood.jpg\n"
                "newmtl Wooden2\nKd 0 0 0\nmap_Kd textures/wood.jpg\n")
    os.symlink(REF_TEXTURES, tmp_path / "textures")
    got = scene.load_scene_dir(str(tmp_path), "room")
    assert list(got.mat_tex[-3:]) == [0, 1, 1] and (got.mat_tex[:-3] == -1).all() and len(got.textures) == 2
    for tex, name in zip(got.textures, ("floor_tiles.png", "wood.jpg")):
        raw = np.asarray(Image.open(os.path.join(REF_TEXTURES, name)).convert("RGB"))
        assert tex.shape == raw.shape and tex.dtype == np.float32
        rng = np.random.default_rng(1)
        for y, x in zip(rng.integers(0, raw.shape[0], 300), rng.integers(0, raw.shape[1], 300)):
            for c in range(3):
                v = np.float32(raw[y, x, c]) / np.float32(255.0)
                want = v * np.float32(1.0 / 12.92) if v <= np.float32(0.04045) else np.float32(
                    math.pow(float((v + np.float32(0.055)) * np.float32(1.0 / 1.055)), 2.4))
                assert abs(float(tex[y, x, c]) - float(want)) <= 5e-7 * max(float(want), 1e-3)      # a few ulps: powf against pow
    # a map_Kd file that is not there fails like the reference does
    with open(tmp_path / "room.mtl", "a") as f:
        f.write("newmtl Rugged\nmap_Kd textures/rug.png\n")
    with pytest.raises(ValueError, match="Fail to load image file"):
        scene.load_scene_dir(str(tmp_path), "room")
    out = tmp_path / "ppm"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_textures.py"), str(tmp_path), "room.mtl", "--out", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "missing:" in r.stdout, r.stdout + r.stderr
    ppm = open(out / "textures" / "floor_tiles.ppm", "rb").read()
    assert ppm.startswith(b"P6\n996 1024\n255\n") and len(ppm) == len(b"P6\n996 1024\n255\n") + 996 * 1024 * 3


@pytest.mark.parametrize("rho", [0.5, 0.8])
def test_furnace_radiance_is_the_geometric_series(rho):
    """Independent of any reference output: in a closed, uniformly emitting diffuse enclosure the MIS path tracer
    (src/integrator.cpp:45-141: emission with power-heuristic weights, light sampling, cosine sampling, Russian roulette)
    must return sum_{k<32} rho^k for every pixel in expectation.  Catches any error in a pdf, a weight or a cosine."""
    from pt_hip import synth
    o = orc.Scene(synth.furnace(rho))
    film, _ = o.render(0, 64, nthreads=8)
    img = orc.film_to_image(film)
    want = (1.0 - rho ** 32) / (1.0 - rho)
    assert abs(img.mean() - want) < 4e-3 * want
    assert np.abs(img.mean(axis=(0, 1)) - want).max() < 6e-3 * want
