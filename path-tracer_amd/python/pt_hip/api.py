"""ctypes bindings of libpt_host.so and libpt_hip.so (include/pt_host.h, include/pt_hip.h).

There is no CPU fallback: every call goes to the shared libraries, and a missing library raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
LIB_DIR = os.environ.get("PT_HIP_LIB_DIR") or os.path.join(_PKG, "lib")   # where libpt_hip.so and libpt_host.so are

TRAVERSE_EXACT, TRAVERSE_FAST = 0, 1
FLAG_NO_KERNEL_TIMING, FLAG_COUNT_WORK = 1, 2
INTEGRATOR_PATH, INTEGRATOR_NORMAL, INTEGRATOR_ALBEDO = 0, 1, 2
NOALLOC_MAX_RAYS = 67108864      # PT_NOALLOC_MAX_RAYS
KERNEL_NAMES = ["generate", "extend", "shade", "shadow", "film", "retrace", "finish", "compact"]

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class BvhNode(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("first", C.c_uint32), ("count", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("ns", C.c_float), ("tex", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("tri", C.c_uint32), ("lemit", C.c_float * 3), ("area", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("rgb", c_f32p)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("eye", C.c_float * 3), ("sample2camera", C.c_float * 16),
                ("camera2world", C.c_float * 16), ("proj_nume", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_verts", "n_normals", "n_uvs", "n_tris", "n_mats", "n_lights", "n_textures", "n_nodes")] + [
        ("verts", c_f32p), ("normals", c_f32p), ("uvs", c_f32p), ("vi", c_u32p), ("ni", c_u32p), ("ti", c_u32p),
        ("mat_id", c_u32p), ("light_id", c_i32p), ("mats", C.POINTER(Material)), ("lights", C.POINTER(Light)),
        ("textures", C.POINTER(Texture)), ("camera", Camera), ("nodes", C.POINTER(BvhNode)), ("prim_ids", c_u32p)]


class RenderParams(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_end", C.c_uint32), ("max_bounce", C.c_uint32), ("integrator", C.c_int32),
                ("traversal", C.c_int32), ("samples_per_pass", C.c_uint32), ("flags", C.c_uint32), ("block_rank", C.c_uint16),
                ("block_count", C.c_uint16)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("exact_retraces", C.c_uint64),
                ("nodes_fetched", C.c_uint64 * 3), ("tris_fetched", C.c_uint64 * 3), ("seconds", C.c_double), ("kernel_ms", C.c_double * 8), ("kernel_launches", C.c_uint64 * 8),
                ("nodes_from_lds", C.c_uint64 * 3), ("samples_per_pass", C.c_uint32), ("passes", C.c_uint32),
                ("shaded_hits", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("paths", "closest_rays", "any_rays", "exact_retraces", "seconds")}
        d["nodes_fetched"] = {n: self.nodes_fetched[i] for i, n in enumerate(("generate", "extend", "shadow"))}
        d["tris_fetched"] = {n: self.tris_fetched[i] for i, n in enumerate(("generate", "extend", "shadow"))}
        d["nodes_from_lds"] = {n: self.nodes_from_lds[i] for i, n in enumerate(("generate", "extend", "shadow"))}
        d["samples_per_pass"], d["passes"], d["shaded_hits"] = self.samples_per_pass, self.passes, self.shaded_hits
        d["kernel_ms"] = {KERNEL_NAMES[i]: self.kernel_ms[i] for i in range(8) if self.kernel_launches[i]}
        d["kernel_launches"] = {KERNEL_NAMES[i]: self.kernel_launches[i] for i in range(8) if self.kernel_launches[i]}
        return d


NODE_DTYPE = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("first", np.uint32), ("count", np.uint32)])
MAT_DTYPE = np.dtype([("kd", np.float32, 3), ("ks", np.float32, 3), ("ns", np.float32), ("tex", np.int32)])
LIGHT_DTYPE = np.dtype([("tri", np.uint32), ("lemit", np.float32, 3), ("area", np.float32)])
RAY_DTYPE = np.dtype([("org", np.float32, 3), ("dir", np.float32, 3), ("tmin", np.float32), ("tmax", np.float32)])
HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])

_host = None
_hip = None


class PtError(RuntimeError):
    pass


def host_lib():
    global _host
    if _host is None:
        L = C.CDLL(os.path.join(LIB_DIR, "libpt_host.so"))
        L.pt_host_last_error.restype = C.c_char_p
        L.pt_host_build_bvh.argtypes = [c_f32p, C.c_uint32, c_u32p, C.c_uint32, C.POINTER(C.POINTER(BvhNode)), c_u32p,
                                        C.POINTER(c_u32p), c_u32p]
        L.pt_host_free.argtypes = [C.c_void_p]
        L.pt_host_camera.argtypes = [C.c_uint32, C.c_uint32, C.c_float, c_f32p, c_f32p, c_f32p, C.POINTER(Camera)]
        L.pt_host_make_lights.argtypes = [c_f32p, c_u32p, C.c_uint32, c_u32p, c_f32p, C.c_uint32, C.POINTER(Light), c_i32p]
        L.pt_host_filter_table.argtypes = [c_f32p]
        L.pt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.pt_host_scene_desc.restype = C.POINTER(SceneDesc)
        L.pt_host_scene_desc.argtypes = [C.c_void_p]
        L.pt_host_scene_free.argtypes = [C.c_void_p]
        L.pt_host_save_png.argtypes = [C.c_char_p, c_f32p, C.c_uint32, C.c_uint32, C.c_int]
        L.pt_host_load_image.argtypes = [C.c_char_p, C.POINTER(c_f32p), c_u32p, c_u32p]
        _host = L
    return _host


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own copy of the HIP runtime (same soname as the system's).  A process that initialises both
    copies loses the GPU in the second one ("No HIP GPUs are available"), so whichever of torch and this backend is loaded
    first must decide for both: when torch is installed but not imported yet, its copy is loaded here, globally, and
    libpt_hip.so's dependency then resolves to it -- as it does anyway when torch was imported first.  PT_HIP_SYSTEM_RUNTIME=1
    keeps the system's runtime (for processes that never import torch)."""
    if "torch" in sys.modules or os.environ.get("PT_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(lib):
            C.CDLL(lib, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def hip_lib():
    global _hip
    if _hip is None:
        _share_torch_hip_runtime()
        L = C.CDLL(os.path.join(LIB_DIR, "libpt_hip.so"))
        L.pt_hip_last_error.restype = C.c_char_p
        L.pt_hip_init.argtypes = [C.c_int]
        L.pt_hip_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.pt_hip_scene_create_on.argtypes = [C.c_int, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.pt_hip_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RenderParams), c_f32p, C.POINTER(Stats)]
        L.pt_hip_scene_destroy.argtypes = [C.c_void_p]
        L.pt_hip_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.pt_hip_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, c_u8p, c_i32p, C.c_int]
        L.pt_hip_intersect_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, c_f32p]
        L.pt_hip_occluded_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, c_f32p]
        L.pt_hip_intersect_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.pt_hip_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), c_f32p, C.POINTER(Stats)]
        L.pt_hip_render_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.pt_hip_li.argtypes = [C.c_void_p, C.POINTER(RenderParams), c_i32p, c_u32p, C.c_size_t, c_f32p, c_f32p, C.POINTER(Stats)]
        L.pt_hip_sobol.argtypes = [C.c_void_p, c_i32p, c_u32p, c_u32p, C.c_size_t, c_f32p]
        L.pt_hip_selftest_division.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.pt_hip_probe_box.argtypes = [c_f32p, C.c_void_p, C.c_size_t, c_u8p]
        L.pt_hip_probe_triangle.argtypes = [c_f32p, C.c_void_p, C.c_size_t, c_u8p, c_f32p]
        L.pt_hip_probe_math.argtypes = [c_f32p, c_f32p, C.c_size_t, c_f32p]
        L.pt_hip_build_bvh.argtypes = [c_f32p, C.c_uint32, c_u32p, C.c_uint32, C.POINTER(C.POINTER(BvhNode)), c_u32p, C.POINTER(c_u32p), c_u32p]
        L.pt_hip_free.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def _chk_host(rc):
    if rc != 0:
        raise PtError("pt_host error %d: %s" % (rc, host_lib().pt_host_last_error().decode()))


def _chk(rc):
    if rc != 0:
        raise PtError("pt_hip error %d: %s" % (rc, hip_lib().pt_hip_last_error().decode()))


def _p(a, t):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------------------------------------- host substrate

def build_bvh(verts, vi):
    verts = np.ascontiguousarray(verts, np.float32)
    vi = np.ascontiguousarray(vi, np.uint32)
    L = host_lib()
    nodes_p, ids_p = C.POINTER(BvhNode)(), c_u32p()
    n_nodes, depth = C.c_uint32(), C.c_uint32()
    _chk_host(L.pt_host_build_bvh(_p(verts, c_f32p), len(verts), _p(vi, c_u32p), len(vi), C.byref(nodes_p), C.byref(n_nodes),
                                  C.byref(ids_p), C.byref(depth)))
    # through uint32 views: a byte-string copy is limited to 2 GiB, 50 M triangles make 3 GB of nodes
    words = np.ctypeslib.as_array(C.cast(nodes_p, c_u32p), shape=(n_nodes.value * (C.sizeof(BvhNode) // 4),))
    nodes = words.copy().view(NODE_DTYPE)
    ids = np.ctypeslib.as_array(ids_p, shape=(len(vi),)).copy()
    L.pt_host_free(nodes_p)
    L.pt_host_free(ids_p)
    return nodes, ids, depth.value


def build_bvh_gpu(verts, vi, device=0):
    """The same tree from pt_hip_build_bvh (segmented-scan sweep-SAH on the GPU)."""
    _default_gpu(device)
    verts = np.ascontiguousarray(verts, np.float32)
    vi = np.ascontiguousarray(vi, np.uint32)
    L = hip_lib()
    nodes_p, ids_p = C.POINTER(BvhNode)(), c_u32p()
    n_nodes, depth = C.c_uint32(), C.c_uint32()
    _chk(L.pt_hip_build_bvh(_p(verts, c_f32p), len(verts), _p(vi, c_u32p), len(vi), C.byref(nodes_p), C.byref(n_nodes), C.byref(ids_p),
                            C.byref(depth)))
    words = np.ctypeslib.as_array(C.cast(nodes_p, c_u32p), shape=(n_nodes.value * (C.sizeof(BvhNode) // 4),))
    nodes = words.copy().view(NODE_DTYPE)
    ids = np.ctypeslib.as_array(ids_p, shape=(len(vi),)).copy()
    L.pt_hip_free(nodes_p)
    L.pt_hip_free(ids_p)
    return nodes, ids, depth.value


def save_png(path, rgb, to_srgb=True):
    """Bitmap::savePNG through pt_host_save_png: rgb is height x width x 3 float32 (linear when to_srgb)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    _chk_host(host_lib().pt_host_save_png(str(path).encode(), _p(rgb, c_f32p), rgb.shape[1], rgb.shape[0], 1 if to_srgb else 0))


def load_image(path):
    """Bitmap::load through pt_host_load_image (PNG, PPM, PFM): height x width x 3 float32, linear."""
    L = host_lib()
    ptr, w, h = c_f32p(), C.c_uint32(), C.c_uint32()
    _chk_host(L.pt_host_load_image(str(path).encode(), C.byref(ptr), C.byref(w), C.byref(h)))
    out = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    L.pt_host_free(ptr)
    return out


def make_camera(sd):
    cam = Camera()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    _chk_host(host_lib().pt_host_camera(sd.width, sd.height, sd.fovy, f3(sd.eye), f3(sd.lookat), f3(sd.up), C.byref(cam)))
    return cam


def make_lights(sd):
    tri, lemit = sd.lights()
    verts = np.ascontiguousarray(sd.verts, np.float32)
    vi = np.ascontiguousarray(sd.vi, np.uint32)
    lights = np.zeros(len(tri), LIGHT_DTYPE)
    light_id = np.zeros(sd.n_tris, np.int32)
    lemit = np.ascontiguousarray(lemit, np.float32)
    _chk_host(host_lib().pt_host_make_lights(_p(verts, c_f32p), _p(vi, c_u32p), sd.n_tris, _p(tri, c_u32p), _p(lemit, c_f32p),
                                             len(tri), lights.ctypes.data_as(C.POINTER(Light)), _p(light_id, c_i32p)))
    return lights, light_id


def filter_table():
    o = np.zeros(33, np.float32)
    host_lib().pt_host_filter_table(_p(o, c_f32p))
    return o


class HostScene:
    """A pt_scene_desc assembled from a SceneData through the host substrate (BVH, camera, light table)."""

    def __init__(self, sd, bvh=None, builder="host"):
        """bvh = (nodes as NODE_DTYPE, prim_ids): use this tree instead of building the reference's (any binary tree in
        the reference's node layout is a valid input of the backend; tests use it for shapes the builder never makes).
        builder: "host" = pt_host_build_bvh, "gpu" = pt_hip_build_bvh (the same tree, built on the device),
        "deferred" = leave the tree out of the description: pt_hip_scene_create builds it."""
        self.sd = sd
        k = self.keep = {}
        k["verts"] = np.ascontiguousarray(sd.verts, np.float32)
        k["normals"] = np.ascontiguousarray(sd.normals, np.float32)
        k["uvs"] = np.ascontiguousarray(sd.uvs, np.float32)
        for n in ("vi", "ni", "ti", "mat_id"):
            k[n] = np.ascontiguousarray(getattr(sd, n), np.uint32)
        mats = np.zeros(len(sd.mat_names), MAT_DTYPE)
        mats["kd"], mats["ks"], mats["ns"], mats["tex"] = sd.mat_kd, sd.mat_ks, sd.mat_ns, sd.mat_tex
        k["mats"] = mats
        k["lights"], k["light_id"] = make_lights(sd)
        if bvh is None and builder == "deferred":
            self.nodes, self.prim_ids, self.depth = np.zeros(0, NODE_DTYPE), np.zeros(0, np.uint32), None
        elif bvh is None:
            self.nodes, self.prim_ids, self.depth = (build_bvh_gpu if builder == "gpu" else build_bvh)(k["verts"], k["vi"])
        else:
            self.nodes, self.prim_ids, self.depth = np.ascontiguousarray(bvh[0], NODE_DTYPE), np.ascontiguousarray(bvh[1], np.uint32), None
        k["tex"] = [np.ascontiguousarray(t, np.float32) for t in sd.textures]
        texs = (Texture * max(1, len(k["tex"])))()
        for i, t in enumerate(k["tex"]):
            texs[i].w, texs[i].h, texs[i].rgb = t.shape[1], t.shape[0], _p(t, c_f32p)
        k["texs"] = texs
        d = self.desc = SceneDesc()
        d.n_verts, d.n_normals, d.n_uvs = len(k["verts"]), len(k["normals"]), len(k["uvs"])
        d.n_tris, d.n_mats, d.n_lights, d.n_textures, d.n_nodes = sd.n_tris, len(mats), len(k["lights"]), len(k["tex"]), len(self.nodes)
        d.verts, d.normals, d.uvs = _p(k["verts"], c_f32p), _p(k["normals"], c_f32p), _p(k["uvs"], c_f32p)
        d.vi, d.ni, d.ti, d.mat_id = (_p(k[n], c_u32p) for n in ("vi", "ni", "ti", "mat_id"))
        d.light_id = _p(k["light_id"], c_i32p)
        d.mats = mats.ctypes.data_as(C.POINTER(Material))
        d.lights = k["lights"].ctypes.data_as(C.POINTER(Light))
        d.textures = C.cast(texs, C.POINTER(Texture))
        d.camera = make_camera(sd)
        if len(self.nodes):
            d.nodes = self.nodes.ctypes.data_as(C.POINTER(BvhNode))
            d.prim_ids = _p(self.prim_ids, c_u32p)


# ------------------------------------------------------------------------------------------------- device backend

_inited = None


def init(device=0):
    """pt_hip_init once per process: `device` becomes the default GPU (of pt_hip_scene_create and the calls without a
    handle); later calls with another ordinal leave it alone -- handles are bound to their own GPU (Scene(device=...))."""
    global _inited
    if _inited is None:
        _chk(hip_lib().pt_hip_init(device))
        _inited = device


def _default_gpu(device):
    """For the calls that take no handle (tree builder, probes, self test): they run on the process's default GPU, so an
    explicit ordinal other than that one is refused instead of being silently ignored."""
    init(device)
    if device != _inited:
        raise PtError("this call runs on the default GPU of the process (%d, set by the first pt_hip_init); %d was asked for: "
                      "use a Scene(device=...) handle for work on another GPU" % (_inited, device))


class Scene:
    """Device scene (pt_hip_scene) built from a SceneData."""

    def __init__(self, sd, device=0, bvh=None, builder="gpu"):
        """builder: who builds the reference-topology tree when bvh is None (see HostScene); the GPU builder by default."""
        init(device)
        self.host = HostScene(sd, bvh, builder)
        self.sd = sd
        self.device = device
        h = C.c_void_p()
        _chk(hip_lib().pt_hip_scene_create_on(device, C.byref(self.host.desc), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            hip_lib().pt_hip_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def intersect(self, rays, mode=TRAVERSE_EXACT):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _chk(hip_lib().pt_hip_intersect(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, mode))
        return hits

    def intersect_device(self, d_rays_ptr, n, d_hits_ptr, mode=TRAVERSE_FAST, noalloc=False):
        """Closest hits of n pt_ray records at device address d_rays_ptr into n pt_hit records at d_hits_ptr.  noalloc=True
        takes the entry point that allocates nothing (at most NOALLOC_MAX_RAYS rays); returns the kernel time in ms or None."""
        if noalloc:
            _chk(hip_lib().pt_hip_intersect_device_noalloc(self.h, C.c_void_p(d_rays_ptr), n, C.c_void_p(d_hits_ptr), mode))
            return None
        ms = C.c_float(0.0)
        _chk(hip_lib().pt_hip_intersect_device(self.h, C.c_void_p(d_rays_ptr), n, C.c_void_p(d_hits_ptr), mode, C.byref(ms)))
        return ms.value

    def occluded_device(self, d_rays_ptr, n, d_occluded_ptr, mode=TRAVERSE_FAST):
        """Occlusion flags (one byte each) of n pt_ray records at device address d_rays_ptr; returns the kernel time in ms."""
        ms = C.c_float(0.0)
        _chk(hip_lib().pt_hip_occluded_device(self.h, C.c_void_p(d_rays_ptr), n, C.c_void_p(d_occluded_ptr), mode, C.byref(ms)))
        return ms.value

    def occluded(self, rays, mode=TRAVERSE_EXACT):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        first = np.zeros(len(rays), np.int32)
        _chk(hip_lib().pt_hip_occluded(self.h, rays.ctypes.data, len(rays), _p(occ, c_u8p), _p(first, c_i32p), mode))
        return occ, first

    def params(self, spp_begin, spp_end, integrator=INTEGRATOR_PATH, traversal=TRAVERSE_EXACT, samples_per_pass=0, max_bounce=32, flags=0,
               block=(0, 0)):
        """block = (rank, count): render only this rank's share of the 32x32 pixel blocks (see pt_render_params)."""
        p = RenderParams()
        p.spp_begin, p.spp_end, p.max_bounce = spp_begin, spp_end, max_bounce
        p.integrator, p.traversal, p.samples_per_pass, p.flags = integrator, traversal, samples_per_pass, flags
        p.block_rank, p.block_count = block
        return p

    def render(self, spp_begin, spp_end, **kw):
        W, H = self.sd.width, self.sd.height
        film = np.zeros((H + 4, W + 4, 4), np.float32)
        st = Stats()
        p = self.params(spp_begin, spp_end, **kw)
        _chk(hip_lib().pt_hip_render(self.h, C.byref(p), _p(film, c_f32p), C.byref(st)))
        return film, st.as_dict()

    def render_device(self, d_film_ptr, spp_begin, spp_end, **kw):
        st = Stats()
        p = self.params(spp_begin, spp_end, **kw)
        _chk(hip_lib().pt_hip_render_device(self.h, C.byref(p), C.c_void_p(d_film_ptr), C.byref(st)))
        return st.as_dict()

    def li(self, pixel_xy, sample_index, **kw):
        pix = np.ascontiguousarray(pixel_xy, np.int32).reshape(-1, 2)
        smp = np.ascontiguousarray(sample_index, np.uint32)
        rgb = np.zeros((len(pix), 3), np.float32)
        ps = np.zeros((len(pix), 2), np.float32)
        st = Stats()
        p = self.params(0, 1, **kw)
        _chk(hip_lib().pt_hip_li(self.h, C.byref(p), _p(pix, c_i32p), _p(smp, c_u32p), len(pix), _p(rgb, c_f32p), _p(ps, c_f32p),
                                 C.byref(st)))
        return rgb, ps, st.as_dict()

    def sobol(self, pixel_xy, sample_index, dim):
        pix = np.ascontiguousarray(pixel_xy, np.int32).reshape(-1, 2)
        smp = np.ascontiguousarray(sample_index, np.uint32)
        dim = np.ascontiguousarray(dim, np.uint32)
        out = np.zeros(len(pix), np.float32)
        _chk(hip_lib().pt_hip_sobol(self.h, _p(pix, c_i32p), _p(smp, c_u32p), _p(dim, c_u32p), len(pix), _p(out, c_f32p)))
        return out


def render_multi(scenes, spp_begin, spp_end, **kw):
    """One frame on several handles of the same scene (one per GPU) from this process: pt_hip_render_multi."""
    sd = scenes[0].sd
    film = np.zeros((sd.height + 4, sd.width + 4, 4), np.float32)
    st = Stats()
    p = scenes[0].params(spp_begin, spp_end, **kw)
    handles = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    _chk(hip_lib().pt_hip_render_multi(handles, len(scenes), C.byref(p), _p(film, c_f32p), C.byref(st)))
    return film, st.as_dict()


def device_count():
    n = hip_lib().pt_hip_device_count()
    if n < 0:
        raise PtError("pt_hip error %d: %s" % (n, hip_lib().pt_hip_last_error().decode()))
    return n


def selftest_division(pairs, seed=1, device=0):
    _default_gpu(device)
    bad = C.c_uint64(0)
    _chk(hip_lib().pt_hip_selftest_division(pairs, seed, C.byref(bad)))
    return bad.value


def probe_box(boxes6, rays, device=0):
    """AABB::intersect for explicit (box, ray) pairs as the kernels evaluate it: n x 6 boxes, n x 8 rays -> uint8[n]."""
    _default_gpu(device)
    boxes6, rays = np.ascontiguousarray(boxes6, np.float32), np.ascontiguousarray(rays, np.float32)
    out = np.zeros(len(rays), np.uint8)
    _chk(hip_lib().pt_hip_probe_box(_p(boxes6, c_f32p), rays.ctypes.data_as(C.c_void_p), len(rays), _p(out, c_u8p)))
    return out


def probe_triangle(tri9, rays, device=0):
    """Triangle::intersect for explicit (triangle, ray) pairs: n x 9 corners, n x 8 rays -> (uint8[n], float32[n, 3] t u v)."""
    _default_gpu(device)
    tri9, rays = np.ascontiguousarray(tri9, np.float32), np.ascontiguousarray(rays, np.float32)
    hit, tuv = np.zeros(len(rays), np.uint8), np.zeros((len(rays), 3), np.float32)
    _chk(hip_lib().pt_hip_probe_triangle(_p(tri9, c_f32p), rays.ctypes.data_as(C.c_void_p), len(rays), _p(hit, c_u8p), _p(tuv, c_f32p)))
    return hit, tuv


def probe_math(x, y, device=0):
    """The shading code's sin(x), cos(x) and x ** y for explicit float32 operands -> float32[n, 3]."""
    _default_gpu(device)
    x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
    out = np.zeros((len(x), 3), np.float32)
    _chk(hip_lib().pt_hip_probe_math(_p(x, c_f32p), _p(y, c_f32p), len(x), _p(out, c_f32p)))
    return out


def film_to_image(film):
    inner = film[2:-2, 2:-2]
    w = inner[..., 3:4]
    return np.where(w != 0, inner[..., :3] / np.where(w != 0, w, 1), 0).astype(np.float32)
