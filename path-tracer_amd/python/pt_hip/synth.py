"""Deterministic stand-in scenes.

Three of the reference's four scene meshes are not in its tree (.MISSING_LARGE_BLOBS), so the configurations that
name them run on generated geometry with the reference's own camera, light and material parameters
(scenes/<name>/<name>.xml and .mtl).  Every generator is a pure function of its arguments.
"""
import numpy as np

from .scene import SceneData


def _scene(name, verts, tris, mat_id, mats, light_infos, cam, normals=None, ni=None, uvs=None, ti=None, mat_tex=None, textures=()):
    verts = np.asarray(verts, np.float32).reshape(-1, 3)
    vi = np.asarray(tris, np.uint32).reshape(-1, 3) + 1
    n = len(vi)
    return SceneData(
        name=name, verts=verts, normals=np.zeros((0, 3), np.float32) if normals is None else np.asarray(normals, np.float32),
        uvs=np.zeros((0, 2), np.float32) if uvs is None else np.asarray(uvs, np.float32),
        vi=vi, ni=np.zeros((n, 3), np.uint32) if ni is None else np.asarray(ni, np.uint32),
        ti=np.zeros((n, 3), np.uint32) if ti is None else np.asarray(ti, np.uint32), mat_id=np.asarray(mat_id, np.uint32),
        mat_names=[m[0] for m in mats], mat_kd=np.asarray([m[1] for m in mats], np.float32).reshape(-1, 3),
        mat_ks=np.asarray([m[2] for m in mats], np.float32).reshape(-1, 3), mat_ns=np.asarray([m[3] for m in mats], np.float32),
        mat_tex=np.full(len(mats), -1, np.int32) if mat_tex is None else np.asarray(mat_tex, np.int32),
        textures=[np.asarray(t, np.float32) for t in textures], light_infos=light_infos, **cam)


class _Mesh:
    def __init__(self):
        self.v, self.t, self.m = [], [], []

    def quad(self, a, b, c, d, mat):
        i = len(self.v)
        self.v += [a, b, c, d]
        self.t += [(i, i + 1, i + 2), (i, i + 2, i + 3)]
        self.m += [mat, mat]

    def box(self, lo, hi, mat):
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        self.quad((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0), mat)
        self.quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), mat)
        self.quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), mat)
        self.quad((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1), mat)
        self.quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), mat)
        self.quad((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0), mat)


def cornell_box(width=768, height=768):
    """Classic Cornell box geometry (the published measurement data: 555-unit room, two blocks, ceiling light) with the
    material names and values of scenes/cornell-box/cornell-box.mtl and the camera/light of cornell-box.xml."""
    mats = [("DiffuseWhite", (0.79, 0.76, 0.73), (0, 0, 0), 1.0), ("LeftWall", (0.9, 0.14, 0.0), (0, 0, 0), 1.0),
            ("RightWall", (0.2, 0.86, 0.0), (0, 0, 0), 1.0), ("Light", (0, 0, 0), (0, 0, 0), 1.0)]
    W, L, R, E = 0, 1, 2, 3
    m = _Mesh()
    m.quad((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2), W)                      # floor
    m.quad((556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0), W)          # ceiling
    m.quad((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2), W)        # back wall
    m.quad((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2), R)                      # right wall
    m.quad((552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0), L)          # left wall
    m.quad((343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227), E)      # light, just under the ceiling
    short = [((130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)),
             ((290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)),
             ((130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)),
             ((82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)),
             ((240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225))]
    tall = [((423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)),
            ((423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)),
            ((472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)),
            ((314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)),
            ((265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247))]
    for q in short + tall:
        m.quad(*q, W)
    cam = dict(width=width, height=height, fovy=float(np.float32(39.3077)), eye=(278.0, 273.0, -800.0), lookat=(278.0, 273.0, -799.0),
               up=(0.0, 1.0, 0.0))
    return _scene("cornell-box-synth", m.v, m.t, m.m, mats, [("Light", (34.0, 24.0, 8.0))], cam)


def furnace(rho=0.5, width=48, height=48):
    """A closed cube whose walls all emit radiance 1 and reflect diffusely with albedo rho, seen from inside.  Every path
    vertex is an emitter, so the radiance along any ray is the geometric series 1 + rho + rho^2 + ... (cut at the
    integrator's 32 bounces): an analytic answer for the whole MIS estimator."""
    m = _Mesh()
    m.box((-1, -1, -1), (1, 1, 1), 0)
    tris = [(a, c, b) for a, b, c in m.t]           # wound so that the geometric normals face inwards
    mats = [("emit", (rho, rho, rho), (0, 0, 0), 1.0)]
    cam = dict(width=width, height=height, fovy=70.0, eye=(0.1, 0.2, -0.3), lookat=(0.3, 0.1, 1.0), up=(0.0, 1.0, 0.0))
    return _scene("furnace", m.v, tris, m.m, mats, [("emit", (1.0, 1.0, 1.0))], cam)


def random_soup(n_tris, seed=0, width=64, height=48, extent=4.0, size=0.6, n_lights=4, glossy=True):
    """Random triangle soup in a cube, a few emissive triangles, a camera looking at it.  For unit and parity tests."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n_tris, 1, 3))
    v = (c + rng.uniform(-size, size, (n_tris, 3, 3))).astype(np.float32).reshape(-1, 3)
    tris = np.arange(3 * n_tris).reshape(-1, 3)
    mats = [("grey", (0.5, 0.5, 0.5), (0, 0, 0), 1.0), ("glossy", (0.1, 0.2, 0.3), (0.3, 0.2, 0.1), 50.0 if glossy else 1.0),
            ("emit", (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]
    mat_id = rng.integers(0, 2, n_tris)
    mat_id[rng.choice(n_tris, min(n_lights, n_tris), replace=False)] = 2
    cam = dict(width=width, height=height, fovy=50.0, eye=(0.0, 0.5, -3.0 * extent), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0))
    return _scene("soup%d" % n_tris, v, tris, mat_id, mats, [("emit", (20.0, 18.0, 15.0))], cam)


def grid_plane(n=16, width=64, height=64):
    """A flat n x n grid of quads (many equal centroid coordinates and flat boxes: exercises sort ties, the median-split
    fallback and the equal-distance rule) lit by a small quad above it."""
    m = _Mesh()
    for j in range(n):
        for i in range(n):
            m.quad((i, 0, j), (i, 0, j + 1), (i + 1, 0, j + 1), (i + 1, 0, j), 0)
    h = n / 2
    m.quad((h - 1, 4, h - 1), (h + 1, 4, h - 1), (h + 1, 4, h + 1), (h - 1, 4, h + 1), 1)
    mats = [("floor", (0.6, 0.6, 0.6), (0.2, 0.2, 0.2), 20.0), ("lamp", (0, 0, 0), (0, 0, 0), 1.0)]
    cam = dict(width=width, height=height, fovy=45.0, eye=(h, 6.0, -n * 0.8), lookat=(h, 0.0, h), up=(0.0, 1.0, 0.0))
    return _scene("grid%d" % n, m.v, m.t, m.m, mats, [("lamp", (30.0, 30.0, 30.0))], cam)


def textured_room(width=96, height=72):
    """Floor and back wall with texture coordinates (some outside [0, 1]: Bitmap::sample clamps, src/bitmap.cpp:153-154;
    the second floor triangle has no texture indices: default uvs, src/shape.cpp:163-165), two textures of odd sizes, a
    pyramid with per-vertex normals on two of its four faces (include/pt/mesh.h:37-44), a side wall that is a perfect
    mirror (a Ks component above 999, src/material.cpp:24,46,94: the specular branch of the integrator,
    src/integrator.cpp:84-87) and a ceiling light."""
    v = [(-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2),            # floor 0..3
         (-2, 0, 2), (2, 0, 2), (2, 3, 2), (-2, 3, 2),               # back wall 4..7
         (-2, 0, -2), (-2, 0, 2), (-2, 3, 2), (-2, 3, -2),           # side wall 8..11
         (-0.5, 2.9, -0.5), (0.5, 2.9, -0.5), (0.5, 2.9, 0.5), (-0.5, 2.9, 0.5),   # light 12..15
         (0.2, 0, 0.2), (1.4, 0, 0.2), (1.4, 0, 1.4), (0.2, 0, 1.4), (0.8, 1.2, 0.8)]  # pyramid 16..20
    tris = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (8, 9, 10), (8, 10, 11), (12, 13, 14), (12, 14, 15),
            (16, 17, 20), (17, 18, 20), (18, 19, 20), (19, 16, 20)]
    mat_id = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 4, 4]
    uvs = [(0, 0), (1, 0), (1, 1), (0, 1), (-0.25, 0.1), (1.3, 0.2), (1.2, 1.4), (-0.1, 0.9)]
    ti = np.zeros((12, 3), np.uint32)
    ti[0] = (1, 3, 2)
    ti[2] = (5, 6, 7)
    ti[3] = (5, 7, 8)
    ti[8] = (1, 2, 3)       # the pyramid's material has no texture: indices present but unused
    nrm = [(-0.3, 0.6, -0.7), (0.3, 0.6, -0.7), (0.0, 1.0, 0.0), (0.7, 0.6, 0.3)]
    nrm = [tuple(np.asarray(n) / np.linalg.norm(n)) for n in nrm]
    ni = np.zeros((12, 3), np.uint32)
    ni[8] = (1, 2, 3)
    ni[9] = (2, 4, 3)
    yy, xx = np.mgrid[0:7, 0:11]
    checker = np.where(((xx + yy) % 2)[..., None] == 0, np.array([0.8, 0.7, 0.2]), np.array([0.1, 0.15, 0.5])).astype(np.float32)
    yy, xx = np.mgrid[0:5, 0:3]
    ramp = np.stack([xx / 2.0, yy / 4.0, 0.25 + 0.0 * xx], -1).astype(np.float32)
    mats = [("floor", (0.5, 0.5, 0.5), (0, 0, 0), 1.0), ("wall", (0.4, 0.4, 0.4), (0.1, 0.1, 0.1), 30.0),
            ("side", (0.0, 0.0, 0.0), (1000.0, 1000.0, 1000.0), 1.0), ("lamp", (0, 0, 0), (0, 0, 0), 1.0), ("pyr", (0.6, 0.3, 0.2), (0, 0, 0), 1.0)]
    cam = dict(width=width, height=height, fovy=55.0, eye=(1.6, 1.4, -3.2), lookat=(0.0, 1.0, 1.0), up=(0.0, 1.0, 0.0))
    return _scene("textured-room", v, tris, mat_id, mats, [("lamp", (25.0, 24.0, 22.0))], cam, normals=nrm, ni=ni, uvs=uvs, ti=ti,
                  mat_tex=[0, 1, -1, -1, -1], textures=[checker, ramp])


def _grid_box(lo, hi, n):
    """Axis-aligned box with every face tessellated into n x n quads: vertices (float64) and triangle indices."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    t = np.linspace(0.0, 1.0, n + 1)
    a, b = np.meshgrid(t, t, indexing="ij")
    a, b = a.ravel(), b.ravel()
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    i, j = i.ravel(), j.ravel()
    q00, q10, q11, q01 = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
    verts, tris = [], []
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, 1):
            p = np.zeros((len(a), 3))
            p[:, axis] = hi[axis] if side else lo[axis]
            p[:, u] = lo[u] + a * (hi[u] - lo[u])
            p[:, v] = lo[v] + b * (hi[v] - lo[v])
            base = sum(len(x) for x in verts)
            verts.append(p)
            if side:
                tris.append(np.stack([q00, q10, q11], 1) + base)
                tris.append(np.stack([q00, q11, q01], 1) + base)
            else:
                tris.append(np.stack([q00, q11, q10], 1) + base)
                tris.append(np.stack([q00, q01, q11], 1) + base)
    return np.concatenate(verts), np.concatenate(tris)


def _icosphere(center, radius, level):
    """Subdivided icosahedron, 20 * 4^level triangles."""
    g = (1 + 5 ** 0.5) / 2
    v = [(-1, g, 0), (1, g, 0), (-1, -g, 0), (1, -g, 0), (0, -1, g), (0, 1, g), (0, -1, -g), (0, 1, -g), (g, 0, -1), (g, 0, 1), (-g, 0, -1),
         (-g, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4),
         (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(i, j):
            k = (min(i, j), max(i, j))
            if k not in cache:
                m = v[i] + v[j]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(center, np.float64) + radius * np.array(v), np.array(f)


def library_synth(detail=4, seed=0x11B4A47, width=1280, height=720):
    """Stand-in for the reference's library scene, whose mesh is not in its tree: a 4 x 2.6 x 3 m reading room, three
    walls lined with seven-level bookcases full of books 2-4 cm thick, a table, two wall sconces (the emitters) and a
    few globes.  Every box face is tessellated into detail x detail quads, so the triangle count scales with detail^2
    (detail 4: about 0.45 M triangles, detail 8: about 1.8 M).  Camera and lights are those of
    scenes/library/library.xml; the materials take the parameter sets of library.mtl (Ns 100 / 5 / 360).
    Books stand 0.5 mm above their board and keep a gap of 8 % of their thickness, so there are no coincident faces."""
    rng = np.random.default_rng(seed)
    WALL, WOOD, BOOK_A, BOOK_B, LAMP1, LAMP2, GLOBE = range(7)
    mats = [("Walls", (0.6, 0.58, 0.55), (0.0, 0.0, 0.0), 5.0), ("Wood", (0.35, 0.2, 0.1), (0.1, 0.1, 0.1), 100.0),
            ("BookA", (0.5, 0.1, 0.1), (0.05, 0.05, 0.05), 5.0), ("BookB", (0.1, 0.2, 0.4), (0.2, 0.2, 0.2), 360.0),
            ("Sconce_Glass.001", (0.5, 0.5, 0.5), (0, 0, 0), 1.0), ("Sconce_Glass", (0.5, 0.5, 0.5), (0, 0, 0), 1.0),
            ("Globe", (0.3, 0.5, 0.3), (0.3, 0.3, 0.3), 100.0)]
    V, T, M = [], [], []
    nv = 0

    def add(verts, tris, mat):
        nonlocal nv
        V.append(verts)
        T.append(tris + nv)
        M.append(np.full(len(tris), mat))
        nv += len(verts)

    def box(lo, hi, mat, n=detail):
        add(*_grid_box(lo, hi, n), mat)

    X0, X1, Y0, Y1, Z0, Z1 = -2.0, 2.0, 0.0, 2.6, -1.5, 1.5
    t = 0.05
    big = max(2, 2 * detail)
    box((X0, Y0 - t, Z0), (X1, Y0, Z1), WALL, big)
    box((X0, Y1, Z0), (X1, Y1 + t, Z1), WALL, big)
    box((X0 - t, Y0, Z0), (X0, Y1, Z1), WALL, big)
    box((X1, Y0, Z0), (X1 + t, Y1, Z1), WALL, big)
    box((X0, Y0, Z0 - t), (X1, Y1, Z0), WALL, big)
    box((X0, Y0, Z1), (X1, Y1, Z1 + t), WALL, big)
    box((-1.68, 1.9, 0.6), (-1.6, 2.1, 0.8), LAMP1, 2)           # sconces on the -x side, in front of the bookcase
    box((-1.68, 1.9, -0.2), (-1.6, 2.1, 0.0), LAMP2, 2)
    box((0.2, 0.72, -0.5), (1.4, 0.76, 0.5), WOOD)                # table top and legs
    for lx, lz in ((0.25, -0.45), (1.3, -0.45), (0.25, 0.4), (1.3, 0.4)):
        box((lx, 0.0, lz), (lx + 0.05, 0.72, lz + 0.05), WOOD, 2)
    for c, r in (((0.5, 0.76 + 0.12, 0.1), 0.12), ((1.0, 0.76 + 0.08, -0.2), 0.08), ((-1.0, 0.15, 1.0), 0.15)):
        add(*_icosphere(c, r, 3 if detail <= 4 else 4), GLOBE)
    boards = []
    for lv in range(7):
        y = 0.1 + lv * 0.33
        boards.append(((X0 + 0.02, y, Z0 + 0.02), (X0 + 0.32, y + 0.02, Z1 - 0.02), 2))      # run along z
        boards.append(((X0 + 0.4, y, Z0 + 0.02), (X1 - 0.6, y + 0.02, Z0 + 0.32), 0))        # run along x
        boards.append(((X0 + 0.4, y, Z1 - 0.32), (X1 - 0.6, y + 0.02, Z1 - 0.02), 0))
    for lo, hi, axis in boards:
        box(lo, hi, WOOD)
        length = hi[axis] - lo[axis]
        pos = 0.0
        while True:
            thick = rng.uniform(0.02, 0.04)
            if pos + thick > length:
                break
            hgt, dep = rng.uniform(0.16, 0.29), rng.uniform(0.15, 0.27)
            blo, bhi = list(lo), list(hi)
            blo[1], bhi[1] = hi[1] + 0.0005, hi[1] + 0.0005 + hgt
            blo[axis], bhi[axis] = lo[axis] + pos, lo[axis] + pos + thick * 0.92
            o = 2 if axis == 0 else 0
            bhi[o] = blo[o] + dep
            box(tuple(blo), tuple(bhi), BOOK_A if rng.random() < 0.5 else BOOK_B)
            pos += thick
    cam = dict(width=width, height=height, fovy=60.0, eye=(float(np.float32(1.6146)), 1.0, float(np.float32(1.0596))),
               lookat=(float(np.float32(0.2)), float(np.float32(1.2)), float(np.float32(0.7))), up=(0.0, 1.0, 0.0))
    lights = [("Sconce_Glass.001", (26.0, 26.0, 26.0)), ("Sconce_Glass", (16.0, 14.0, 8.0))]
    return _scene("library-synth", np.concatenate(V), np.concatenate(T), np.concatenate(M), mats, lights, cam)


def _grid_quad(origin, eu, ev, n, uv_lo=(0.0, 0.0), uv_hi=(1.0, 1.0)):
    """Parallelogram origin + a*eu + b*ev tessellated into n x n quads, with texture coordinates running from uv_lo to uv_hi."""
    t = np.linspace(0.0, 1.0, n + 1)
    a, b = np.meshgrid(t, t, indexing="ij")
    a, b = a.ravel(), b.ravel()
    verts = np.asarray(origin, np.float64) + a[:, None] * np.asarray(eu, np.float64) + b[:, None] * np.asarray(ev, np.float64)
    uvs = np.stack([uv_lo[0] + a * (uv_hi[0] - uv_lo[0]), uv_lo[1] + b * (uv_hi[1] - uv_lo[1])], 1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    i, j = i.ravel(), j.ravel()
    q00, q10, q11, q01 = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
    tris = np.concatenate([np.stack([q00, q10, q11], 1), np.stack([q00, q11, q01], 1)])
    return verts, uvs, tris


def _procedural_textures():
    """Stand-ins for the bathroom's texture files (tiles, marble, wood, rug, picture), linear RGB float."""
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    tiles = np.where((((xx // 32) + (yy // 32)) % 2)[..., None] == 0, np.float32([0.75, 0.75, 0.7]), np.float32([0.25, 0.3, 0.35]))
    tiles = tiles * np.where(((xx % 32 < 2) | (yy % 32 < 2))[..., None], np.float32(0.3), np.float32(1.0))
    marble = 0.6 + 0.25 * np.sin(xx / 17.0 + 3.0 * np.sin(yy / 29.0))[..., None] * np.float32([1.0, 0.95, 0.9])
    wood = np.float32([0.45, 0.28, 0.15]) * (0.7 + 0.3 * np.sin(yy / 3.0 + 2.0 * np.sin(xx / 40.0)))[..., None]
    rug = np.float32([0.5, 0.1, 0.1]) * (0.6 + 0.4 * (((xx // 8) + (yy // 8)) % 2))[..., None]
    pic = np.stack([xx / 255.0, yy / 255.0, 0.5 + 0.5 * np.sin((xx + yy) / 20.0)], -1)
    return [np.clip(t, 0.0, 1.0).astype(np.float32) for t in (tiles, marble, wood, rug, pic)]


def bathroom_textures(texture_dir):
    """The bathroom's own texture images from a copy of the reference's scenes/bathroom/textures (floor_tiles.png,
    marble.jpg, wood.jpg, picture1.jpg; rug.png is not in the reference's tree, the procedural rug stands in), decoded the
    way Bitmap::load does (scene.load_image_linear).  Order: tiles, marble, wood, rug, picture."""
    import os
    from .scene import load_image_linear
    proc = _procedural_textures()
    files = ("floor_tiles.png", "marble.jpg", "wood.jpg", None, "picture1.jpg")
    return [proc[i] if f is None else load_image_linear(os.path.join(texture_dir, f)) for i, f in enumerate(files)]


def bathroom_synth(detail=9, width=768, height=768, texture_dir=None):
    """Stand-in for the reference's bathroom scene, whose mesh is not in its tree: a tiled room with a tub, a vanity, a
    glossy wall panel, spheres and cylinders-as-prisms of the metal materials, a rug and a picture, lit by one ceiling panel.
    Camera and light are those of scenes/bathroom/bathroom.xml; material values are taken from bathroom.mtl (Floor, Marble,
    Wood, Rug, Painting1 textured; Mirror, StainlessSmooth, StainlessRough, Bathtube, GreyWall, Light untextured); the
    texture images are procedural unless texture_dir names a copy of the reference's scenes/bathroom/textures
    (bathroom_textures).  Triangle count grows with detail^2 (detail 9: about 0.47 M)."""
    TILES, MARBLE, WOODT, RUGT, PIC = range(5)
    mats = [("Floor", (0, 0, 0), (0, 0, 0), 4.0, TILES), ("Marble", (0.5, 0.5, 0.5), (0, 0, 0), 1.0, MARBLE),
            ("Wood", (0, 0, 0), (0, 0, 0), 6.0, WOODT), ("Rug", (0.5, 0.5, 0.5), (0, 0, 0), 1.0, RUGT),
            ("Painting1", (0.5, 0.5, 0.5), (0, 0, 0), 1.0, PIC), ("Mirror", (0, 0, 0), (1, 1, 1), 10000.0, -1),
            ("StainlessSmooth", (0.2, 0.3, 0.4), (0.2, 0.3, 0.4), 6.0, -1), ("StainlessRough", (0.8, 0.7, 0.8), (0.8, 0.7, 0.8), 9.0, -1),
            ("Bathtube", (0.612066, 0.612066, 0.612066), (0.612066, 0.612066, 0.612066), 1000.0, -1),
            ("GreyWall", (0.893289, 0.893289, 0.893289), (0, 0, 0), 1.0, -1), ("Light", (0, 0, 0), (0, 0, 0), 1.0, -1)]
    FLOOR, MARB, WOOD, RUG, PAINT, MIRROR, SSMOOTH, SROUGH, TUB, WALL, LIGHT = range(11)
    V, U, T, TI, M = [], [], [], [], []
    nv = nu = 0

    def add(verts, tris, mat, uvs=None):
        nonlocal nv, nu
        V.append(verts)
        T.append(tris + nv)
        M.append(np.full(len(tris), mat))
        if uvs is None:
            TI.append(np.zeros((len(tris), 3), np.int64))
        else:
            U.append(uvs)
            TI.append(tris + nu + 1)
            nu += len(uvs)
        nv += len(verts)

    def quad(vut, mat, textured=True):
        verts, uvs, tris = vut
        add(verts, tris, mat, uvs if textured else None)

    X0, X1, Y0, Y1, Z0, Z1 = -3.6, 0.9, 0.0, 2.5, -3.5, 0.9
    n_big = max(2, 24 * detail)
    quad(_grid_quad((X0, Y0, Z0), (0, 0, Z1 - Z0), (X1 - X0, 0, 0), n_big), FLOOR)                 # floor, normal +y
    quad(_grid_quad((X0, Y1, Z0), (X1 - X0, 0, 0), (0, 0, Z1 - Z0), n_big // 2), WALL, False)     # ceiling
    quad(_grid_quad((X0, Y0, Z0), (X1 - X0, 0, 0), (0, Y1, 0), n_big), MARB)                      # wall z = Z0
    quad(_grid_quad((X0, Y0, Z0), (0, Y1, 0), (0, 0, Z1 - Z0), n_big), MARB)                      # wall x = X0
    quad(_grid_quad((X1, Y0, Z0), (0, 0, Z1 - Z0), (0, Y1, 0), n_big // 2), WALL, False)          # wall x = X1
    quad(_grid_quad((X0, Y0, Z1), (0, Y1, 0), (X1 - X0, 0, 0), n_big // 2), WALL, False)          # wall z = Z1
    quad(_grid_quad((-1.9, 2.48, -2.0), (0.8, 0, 0), (0, 0, 0.8), 2), LIGHT, False)               # ceiling panel, facing down
    quad(_grid_quad((-2.6, 0.005, -1.6), (0, 0, 1.4), (1.8, 0, 0), 4 * detail), RUG)
    quad(_grid_quad((X0 + 0.01, 1.1, -2.2), (0, 0.9, 0), (0, 0, 1.2), 2 * detail), PAINT)         # picture on wall x = X0
    quad(_grid_quad((-2.4, 0.9, Z0 + 0.01), (1.6, 0, 0), (0, 1.2, 0), 2 * detail), MIRROR)        # glossy panel on wall z = Z0

    def box(lo, hi, mat, n):
        add(*_grid_box(lo, hi, n), mat)

    box((-3.5, 0.0, -3.4), (-1.7, 0.55, -2.6), TUB, 6 * detail)                                            # tub
    box((-0.9, 0.0, -3.4), (0.6, 0.8, -2.9), WOOD, 4 * detail)                                             # vanity (untextured faces: no uvs)
    box((-0.9, 0.8, -3.4), (0.6, 0.84, -2.9), MARB, 2 * detail)
    lvl = 4 if detail >= 4 else 2
    for c, r, m in (((-0.5, 0.84 + 0.09, -3.15), 0.09, SSMOOTH), ((0.1, 0.84 + 0.07, -3.1), 0.07, SROUGH), ((-2.6, 0.55 + 0.12, -3.0), 0.12, SSMOOTH),
                    ((-1.2, 0.15, -1.0), 0.15, SROUGH), ((-3.2, 0.2, 0.2), 0.2, TUB)):
        add(*_icosphere(c, r, lvl), m)
    box((-3.55, 0.0, -0.6), (-3.45, 1.8, -0.5), SSMOOTH, 2 * detail)                                        # towel rail posts
    box((-3.55, 0.0, 0.3), (-3.45, 1.8, 0.4), SSMOOTH, 2 * detail)
    cam = dict(width=width, height=height, fovy=55.0,
               eye=tuple(float(np.float32(x)) for x in (0.0072405338287353516, 0.9124049544334412, -0.2275838851928711)),
               lookat=tuple(float(np.float32(x)) for x in (-2.787562608718872, 0.9699121117591858, -2.6775901317596436)), up=(0.0, 1.0, 0.0))
    return _scene("bathroom-synth", np.concatenate(V), np.concatenate(T), np.concatenate(M), [m[:4] for m in mats], [("Light", (30.0, 30.0, 30.0))],
                  cam, uvs=np.concatenate(U), ti=np.concatenate(TI), mat_tex=[m[4] for m in mats],
                  textures=_procedural_textures() if texture_dir is None else bathroom_textures(texture_dir))

