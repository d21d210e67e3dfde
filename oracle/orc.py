"""ctypes wrapper of the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
from the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INTEGRATOR_PATH, INTEGRATOR_NORMAL, INTEGRATOR_ALBEDO = 0, 1, 2

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)
c_u8p = C.POINTER(C.c_uint8)


class SceneDesc(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_verts", "n_normals", "n_uvs", "n_tris", "n_mats", "n_lights", "n_textures")] + [
        ("verts", c_f32p), ("normals", c_f32p), ("uvs", c_f32p), ("vi", c_u32p), ("ni", c_u32p), ("ti", c_u32p),
        ("mat_id", c_u32p), ("mat_kd", c_f32p), ("mat_ks", c_f32p), ("mat_ns", c_f32p), ("mat_tex", c_i32p),
        ("light_tri", c_u32p), ("light_lemit", c_f32p), ("tex_w", c_u32p), ("tex_h", c_u32p), ("tex_rgb", C.POINTER(c_f32p)),
        ("width", C.c_uint32), ("height", C.c_uint32), ("fovy", C.c_float),
        ("eye", C.c_float * 3), ("lookat", C.c_float * 3), ("up", C.c_float * 3)]


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("bary", C.c_float * 3)]


class Surface(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("n", C.c_float * 3), ("ng", C.c_float * 3), ("uv", C.c_float * 2)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("paths", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("seconds", C.c_double)]


HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("bary", np.float32, 3)])
SURF_DTYPE = np.dtype([("p", np.float32, 3), ("n", np.float32, 3), ("ng", np.float32, 3), ("uv", np.float32, 2)])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpt_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_create_with_tree.restype = C.c_void_p
        L.orc_scene_create_with_tree.argtypes = [C.POINTER(SceneDesc), C.c_size_t, c_f32p, c_u32p, c_u32p, c_u32p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_bvh_num_nodes.restype = C.c_size_t
        L.orc_bvh_num_nodes.argtypes = [C.c_void_p]
        L.orc_bvh_get.argtypes = [C.c_void_p, c_f32p, c_u64p, c_u64p, c_u64p]
        L.orc_tri_intersect.restype = C.c_int
        L.orc_tri_intersect.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p]
        L.orc_aabb_intersect.restype = C.c_int
        L.orc_aabb_intersect.argtypes = [c_f32p, c_f32p]
        L.orc_intersect.argtypes = [C.c_void_p, c_f32p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_occluded.argtypes = [C.c_void_p, c_f32p, C.c_size_t, c_u8p, c_i32p]
        L.orc_traversal_counters.argtypes = [C.c_void_p, c_u64p, c_u64p]
        L.orc_sobol_index.restype = C.c_uint64
        L.orc_sobol_index.argtypes = [C.c_uint32, C.c_uint64, C.c_int32, C.c_int32]
        L.orc_sobol_sample.restype = C.c_float
        L.orc_sobol_sample.argtypes = [C.c_int64, C.c_int]
        L.orc_pixel_sample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, c_f32p]
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, c_f32p]
        L.orc_camera_get.argtypes = [C.c_void_p, c_f32p, c_f32p, c_f32p]
        L.orc_brdf_eval.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_brdf_pdf.restype = C.c_float
        L.orc_brdf_pdf.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p]
        L.orc_brdf_sample.restype = C.c_int
        L.orc_brdf_sample.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_light_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, c_f32p, c_f32p, c_f32p]
        L.orc_light_pdf.restype = C.c_float
        L.orc_light_pdf.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, c_f32p, c_f32p]
        L.orc_li.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_uint32, c_f32p, c_f32p, c_f32p, C.c_size_t,
                             C.POINTER(C.c_size_t), c_i32p]
        L.orc_filter_table.argtypes = [c_f32p]
        L.orc_film_put.argtypes = [c_f32p, C.c_uint32, C.c_uint32, c_f32p, c_f32p]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, c_f32p, C.POINTER(Stats)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Scene:
    """An oracle scene built from a pt_hip.scene.SceneData."""

    def __init__(self, sd, bvh=None):
        """bvh = (nodes with fields lo, hi, first, count; prim_ids): traverse this tree instead of running the oracle's own
        builder (for scenes of tens of millions of triangles)."""
        L = lib()
        self.sd = sd
        tri, lemit = sd.lights()
        k = self._keep = dict(
            verts=_f32(sd.verts), normals=_f32(sd.normals), uvs=_f32(sd.uvs),
            vi=np.ascontiguousarray(sd.vi, np.uint32), ni=np.ascontiguousarray(sd.ni, np.uint32),
            ti=np.ascontiguousarray(sd.ti, np.uint32), mat_id=np.ascontiguousarray(sd.mat_id, np.uint32),
            kd=_f32(sd.mat_kd), ks=_f32(sd.mat_ks), ns=_f32(sd.mat_ns), tex=np.ascontiguousarray(sd.mat_tex, np.int32),
            ltri=np.ascontiguousarray(tri, np.uint32), lemit=_f32(lemit),
            tw=np.array([t.shape[1] for t in sd.textures], np.uint32), th=np.array([t.shape[0] for t in sd.textures], np.uint32),
            trgb=[_f32(t) for t in sd.textures])
        d = SceneDesc()
        d.n_verts, d.n_normals, d.n_uvs = len(k["verts"]), len(k["normals"]), len(k["uvs"])
        d.n_tris, d.n_mats, d.n_lights, d.n_textures = sd.n_tris, len(k["kd"]), len(k["ltri"]), len(k["trgb"])
        d.verts, d.normals, d.uvs = _p(k["verts"], c_f32p), _p(k["normals"], c_f32p), _p(k["uvs"], c_f32p)
        d.vi, d.ni, d.ti, d.mat_id = _p(k["vi"], c_u32p), _p(k["ni"], c_u32p), _p(k["ti"], c_u32p), _p(k["mat_id"], c_u32p)
        d.mat_kd, d.mat_ks, d.mat_ns, d.mat_tex = _p(k["kd"], c_f32p), _p(k["ks"], c_f32p), _p(k["ns"], c_f32p), _p(k["tex"], c_i32p)
        d.light_tri, d.light_lemit = _p(k["ltri"], c_u32p), _p(k["lemit"], c_f32p)
        d.tex_w, d.tex_h = _p(k["tw"], c_u32p), _p(k["th"], c_u32p)
        arr = (c_f32p * max(1, len(k["trgb"])))(*[_p(t, c_f32p) for t in k["trgb"]])
        k["tex_ptrs"] = arr
        d.tex_rgb = C.cast(arr, C.POINTER(c_f32p))
        d.width, d.height, d.fovy = sd.width, sd.height, sd.fovy
        d.eye, d.lookat, d.up = (C.c_float * 3)(*sd.eye), (C.c_float * 3)(*sd.lookat), (C.c_float * 3)(*sd.up)
        if bvh is None:
            self.h = L.orc_scene_create(C.byref(d))
        else:
            nodes, ids = bvh
            boxes = np.ascontiguousarray(np.concatenate([nodes["lo"], nodes["hi"]], 1), np.float32)
            first, count = np.ascontiguousarray(nodes["first"], np.uint32), np.ascontiguousarray(nodes["count"], np.uint32)
            ids = np.ascontiguousarray(ids, np.uint32)
            self.h = L.orc_scene_create_with_tree(C.byref(d), len(nodes), _p(boxes, c_f32p), _p(first, c_u32p), _p(count, c_u32p), _p(ids, c_u32p))
        self.n_lights = d.n_lights

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- BVH
    def bvh(self):
        L = lib()
        n = L.orc_bvh_num_nodes(self.h)
        boxes = np.zeros((n, 6), np.float32)
        first = np.zeros(n, np.uint64)
        count = np.zeros(n, np.uint64)
        prim = np.zeros(self.sd.n_tris, np.uint64)
        L.orc_bvh_get(self.h, _p(boxes, c_f32p), _p(first, c_u64p), _p(count, c_u64p), _p(prim, c_u64p))
        return boxes, first, count, prim

    # -- geometry
    def tri_intersect(self, prim, ray8):
        r = _f32(ray8)
        b = np.zeros(3, np.float32)
        t = np.zeros(1, np.float32)
        hit = lib().orc_tri_intersect(self.h, int(prim), _p(r, c_f32p), _p(b, c_f32p), _p(t, c_f32p))
        return bool(hit), b, float(t[0])

    def intersect(self, rays, surfaces=False):
        rays = _f32(rays).reshape(-1, 8)
        out = np.zeros(len(rays), HIT_DTYPE)
        surf = np.zeros(len(rays), SURF_DTYPE) if surfaces else None
        lib().orc_intersect(self.h, _p(rays, c_f32p), len(rays), out.ctypes.data, surf.ctypes.data if surfaces else None)
        return (out, surf) if surfaces else out

    def occluded(self, rays):
        rays = _f32(rays).reshape(-1, 8)
        out = np.zeros(len(rays), np.uint8)
        first = np.zeros(len(rays), np.int32)
        lib().orc_occluded(self.h, _p(rays, c_f32p), len(rays), _p(out, c_u8p), _p(first, c_i32p))
        return out, first

    def traversal_counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_traversal_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # -- sampler / camera
    def pixel_sample(self, px, py, s):
        o = np.zeros(2, np.float32)
        lib().orc_pixel_sample(self.h, px, py, s, _p(o, c_f32p))
        return o

    def camera_ray(self, sx, sy):
        o = np.zeros(8, np.float32)
        lib().orc_camera_ray(self.h, float(sx), float(sy), _p(o, c_f32p))
        return o

    def camera(self):
        a, b, c = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(1, np.float32)
        lib().orc_camera_get(self.h, _p(a, c_f32p), _p(b, c_f32p), _p(c, c_f32p))
        return a.reshape(4, 4), b.reshape(4, 4), float(c[0])

    # -- shading
    def brdf_eval(self, prim, bary, wo, wi):
        f = np.zeros(3, np.float32)
        lib().orc_brdf_eval(self.h, prim, _p(_f32(bary), c_f32p), _p(_f32(wo), c_f32p), _p(_f32(wi), c_f32p), _p(f, c_f32p))
        return f

    def brdf_pdf(self, prim, bary, wo, wi):
        return lib().orc_brdf_pdf(self.h, prim, _p(_f32(bary), c_f32p), _p(_f32(wo), c_f32p), _p(_f32(wi), c_f32p))

    def brdf_sample(self, prim, bary, wo, uc, u):
        wi, pdf, f = np.zeros(3, np.float32), np.zeros(1, np.float32), np.zeros(3, np.float32)
        spec = lib().orc_brdf_sample(self.h, prim, _p(_f32(bary), c_f32p), _p(_f32(wo), c_f32p), float(uc), _p(_f32(u), c_f32p),
                                     _p(wi, c_f32p), _p(pdf, c_f32p), _p(f, c_f32p))
        return bool(spec), wi, float(pdf[0]), f

    def light_sample(self, light, prim, bary, u):
        o = np.zeros(14, np.float32)
        lib().orc_light_sample(self.h, light, prim, _p(_f32(bary), c_f32p), _p(_f32(u), c_f32p), _p(o, c_f32p))
        return dict(L=o[0:3], wi=o[3:6], p=o[6:9], n=o[9:12], pdfArea=float(o[12]), pdfDir=float(o[13]))

    def light_pdf(self, light, prim, bary, ray8):
        return lib().orc_light_pdf(self.h, light, prim, _p(_f32(bary), c_f32p), _p(_f32(ray8), c_f32p))

    # -- integrator
    def li(self, px, py, sample, integrator=INTEGRATOR_PATH, max_log=0):
        rgb, ps = np.zeros(3, np.float32), np.zeros(2, np.float32)
        log = np.zeros((max(1, max_log), 9), np.float32)
        n_log, dims = C.c_size_t(0), C.c_int32(0)
        lib().orc_li(self.h, integrator, px, py, sample, _p(rgb, c_f32p), _p(ps, c_f32p), _p(log, c_f32p) if max_log else None,
                     max_log, C.byref(n_log), C.byref(dims))
        return dict(rgb=rgb, pixel_sample=ps, rays=log[: min(n_log.value, max_log)], n_rays=n_log.value, dims=dims.value)

    def render(self, spp_begin, spp_end, integrator=INTEGRATOR_PATH, nthreads=0, max_blocks=0):
        W, H = self.sd.width, self.sd.height
        film = np.zeros((H + 4, W + 4, 4), np.float32)
        st = Stats()
        lib().orc_render(self.h, integrator, spp_begin, spp_end, nthreads, max_blocks, _p(film, c_f32p), C.byref(st))
        stats = {n: getattr(st, n) for n, _ in Stats._fields_}
        return film, stats


def sobol_index(m, frame, px, py):
    return lib().orc_sobol_index(m, frame, px, py)


def sobol_sample(index, dim):
    return lib().orc_sobol_sample(index, dim)


def filter_table():
    o = np.zeros(33, np.float32)
    lib().orc_filter_table(_p(o, c_f32p))
    return o


def film_put(film, pos, rgb):
    H, W = film.shape[0] - 4, film.shape[1] - 4
    lib().orc_film_put(_p(film, c_f32p), W, H, _p(_f32(pos), c_f32p), _p(_f32(rgb), c_f32p))


def film_to_image(film):
    """rgb / weight on the interior (color.h:135-140, main.cpp:36-43)."""
    inner = film[2:-2, 2:-2]
    w = inner[..., 3:4]
    return np.where(w != 0, inner[..., :3] / np.where(w != 0, w, 1), 0).astype(np.float32)
