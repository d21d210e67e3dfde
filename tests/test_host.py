"""Host substrate (libpt_host.so) against the oracle, the scene readers, and the exported C ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc
from pt_hip import api, synth
from pt_hip.scene import load_scene_dir

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _same_bvh(sd):
    nodes, ids, depth = api.build_bvh(sd.verts, sd.vi)
    boxes, first, count, prim = orc.Scene(sd).bvh()
    assert len(nodes) == len(boxes)
    assert np.array_equal(nodes["lo"].view(np.uint32), boxes[:, :3].view(np.uint32))
    assert np.array_equal(nodes["hi"].view(np.uint32), boxes[:, 3:].view(np.uint32))
    assert np.array_equal(nodes["first"], first) and np.array_equal(nodes["count"], count)
    assert np.array_equal(ids, prim)
    return nodes, depth


def test_bvh_identical_to_oracle_veach(veach_sd):
    nodes, depth = _same_bvh(veach_sd)
    assert len(nodes) == 6183 and depth == 17


@pytest.mark.parametrize("make", [
    lambda: synth.random_soup(1, seed=1), lambda: synth.random_soup(2, seed=2), lambda: synth.random_soup(9, seed=3),
    lambda: synth.random_soup(1000, seed=4), lambda: synth.grid_plane(16), synth.cornell_box,
    lambda: synth.library_synth(detail=1)])
def test_bvh_identical_to_oracle_synthetic(make):
    _same_bvh(make())


def test_parallel_build_numbers_nodes_like_the_sequential_reference_order():
    """Above 65536 triangles the host builder splits large ranges on the calling thread, builds the subtrees in tasks and
    replays the reference's node numbering afterwards: the result must still be the oracle's (single-threaded) tree."""
    sd = synth.library_synth(detail=4)
    assert sd.n_tris > 4 * 65536
    _same_bvh(sd)
    _same_bvh(synth.random_soup(150000, seed=12))


def test_bvh_many_identical_triangles_uses_median_fallback():
    """20 triangles collapsed to one point: every box has zero area, so no split costs less than a leaf and more than
    8 primitives force the median split on axis 0 (src/bvh.cpp:176-184)."""
    sd = synth.random_soup(20, seed=9)
    sd.verts[:] = sd.verts[0]
    nodes, _ = _same_bvh(sd)
    assert nodes["count"].max() <= 8 and (nodes["count"] > 1).any()


def test_camera_and_lights_and_filter_equal_oracle(veach_sd):
    hs = api.HostScene(veach_sd)
    o = orc.Scene(veach_sd)
    s2c, c2w, pn = o.camera()
    cam = hs.desc.camera
    assert (np.array(cam.sample2camera[:], np.float32).reshape(4, 4) == s2c).all()   # == : signed zeros may differ
    assert (np.array(cam.camera2world[:], np.float32).reshape(4, 4) == c2w).all()
    assert cam.proj_nume == pn and (cam.width, cam.height) == (1280, 720)
    assert np.array_equal(api.filter_table().view(np.uint32), orc.filter_table().view(np.uint32))
    lights = hs.keep["lights"]
    assert len(lights) == 3040 and (hs.keep["light_id"] >= 0).sum() == 3040
    tri = veach_sd.verts[veach_sd.vi[lights["tri"][5]] - 1].astype(np.float64)
    assert np.isclose(lights["area"][5], 0.5 * np.linalg.norm(np.cross(tri[1] - tri[0], tri[2] - tri[0])), rtol=1e-4)


SMALL_OBJ = """mtllib s.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.2 0.2 1
v 0.8 0.2 1
v 0.5 0.8 1
vn 0 0 1
vt 0 0
vt 1 0
vt 1 1
usemtl wall
f 1/1/1 2/2/1 3/3/1 4/1/1
usemtl lamp
f -3//1 -2//1 -1//1
"""
SMALL_MTL = """newmtl wall
Kd 0.5 0.25 0.125
Ks 0.1 0.1 0.1
Ns 30
newmtl lamp
Kd 0 0 0
Ks 0 0 0
Ns 1
"""
SMALL_XML = """<?xml version="1.0" encoding="utf-8"?>
<camera type="perspective" width="40" height="30" fovy="35.5">
	<eye x="0.5" y="0.5" z="-3"/>
	<lookat x="0.5" y="0.5" z="0"/>
	<up x="0" y="1" z="0"/>
</camera>
<backend type="hip" gpus="8"/>
<!-- <light mtlname="wall" radiance="1,1,1"/> -->
<light mtlname="lamp" radiance="5, 4.5,3"/>
"""


def _load_with_host(folder, name):
    L = api.host_lib()
    h = C.c_void_p()
    rc = L.pt_host_scene_load(os.path.join(folder, name + ".obj").encode(), os.path.join(folder, name + ".xml").encode(), C.byref(h))
    assert rc == 0, L.pt_host_last_error()
    return h, L.pt_host_scene_desc(h).contents


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy() if n else np.zeros(0, dtype)


def _check_against_python_loader(desc, sd):
    assert (desc.n_tris, desc.n_verts, desc.n_normals, desc.n_uvs) == (sd.n_tris, len(sd.verts), len(sd.normals), len(sd.uvs))
    assert np.array_equal(_arr(desc.verts, 3 * desc.n_verts, np.float32), sd.verts.ravel())
    assert np.array_equal(_arr(desc.vi, 3 * desc.n_tris, np.uint32), sd.vi.ravel())
    assert np.array_equal(_arr(desc.ni, 3 * desc.n_tris, np.uint32), sd.ni.ravel())
    assert np.array_equal(_arr(desc.ti, 3 * desc.n_tris, np.uint32), sd.ti.ravel())
    assert np.array_equal(_arr(desc.mat_id, desc.n_tris, np.uint32), sd.mat_id)
    tri, lemit = sd.lights()
    assert desc.n_lights == len(tri)
    for i in range(desc.n_lights):
        assert desc.lights[i].tri == tri[i] and list(desc.lights[i].lemit) == list(lemit[i])
    hs = api.HostScene(sd)
    assert desc.n_nodes == len(hs.nodes)
    assert bytes(C.string_at(desc.nodes, desc.n_nodes * 32)) == hs.nodes.tobytes()
    assert np.array_equal(_arr(desc.prim_ids, desc.n_tris, np.uint32), hs.prim_ids)
    assert bytes(desc.camera) == bytes(hs.desc.camera)


def test_scene_loader_small_scene(tmp_path):
    for n, txt in (("s.obj", SMALL_OBJ), ("s.mtl", SMALL_MTL), ("s.xml", SMALL_XML)):
        (tmp_path / n).write_text(txt)
    h, desc = _load_with_host(str(tmp_path), "s")
    sd = load_scene_dir(str(tmp_path), "s")
    assert sd.n_tris == 3 and sd.light_infos == [("lamp", (5.0, 4.5, 3.0))]   # quad fan-triangulated, comment skipped
    _check_against_python_loader(desc, sd)
    assert desc.mats[0].ns == 30 and list(desc.mats[0].kd) == [0.5, 0.25, 0.125]
    api.host_lib().pt_host_scene_free(h)


@pytest.mark.skipif(not os.path.exists("/root/reference/scenes/veach-mis/veach-mis.obj"), reason="reference tree not present")
def test_scene_loader_veach(veach_sd):
    h, desc = _load_with_host("/root/reference/scenes/veach-mis", "veach-mis")
    _check_against_python_loader(desc, veach_sd)
    api.host_lib().pt_host_scene_free(h)


def test_scene_loader_errors(tmp_path):
    L = api.host_lib()
    h = C.c_void_p()
    assert L.pt_host_scene_load(b"/nonexistent/a.obj", b"/nonexistent/a.xml", C.byref(h)) < 0
    assert b"Cannot open" in L.pt_host_last_error()
    (tmp_path / "s.obj").write_text(SMALL_OBJ)
    (tmp_path / "s.mtl").write_text(SMALL_MTL)
    (tmp_path / "s.xml").write_text(SMALL_XML.replace("perspective", "ortho"))
    assert L.pt_host_scene_load(str(tmp_path / "s.obj").encode(), str(tmp_path / "s.xml").encode(), C.byref(h)) < 0
    assert b"Only support perspective camera!" in L.pt_host_last_error()   # same message as src/scene.cpp:130


def test_build_bvh_rejects_bad_indices():
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(api.PtError):
        api.build_bvh(v, np.array([[1, 2, 4]], np.uint32))


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_(?:hip|host)_[a-z_0-9]+)\s*\(", txt)))


@pytest.mark.parametrize("header,libname", [("pt_hip.h", "libpt_hip.so"), ("pt_host.h", "libpt_host.so")])
def test_library_exports_every_declared_symbol(header, libname):
    """The C-ABI libraries load without a GPU and export exactly what include/*.h declares (no compute call is made)."""
    path = os.path.join(api.LIB_DIR, libname)
    assert os.path.exists(path), "build the libraries first: make -C path-tracer_amd"
    lib = C.CDLL(path)
    names = _declared_functions(header)
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), n


def test_headers_are_plain_c_and_the_binding_mirrors_their_layout(tmp_path):
    """include/*.h compile as C99 on their own, and every struct has the size and field offsets the ctypes binding
    assumes (the binding is hand-written: this is what catches it drifting from the header)."""
    import subprocess
    inc = os.path.join(os.path.dirname(api.LIB_DIR), "..", "include")
    mirror = {"pt_bvh_node": api.BvhNode, "pt_material": api.Material, "pt_light": api.Light, "pt_texture": api.Texture,
              "pt_camera": api.Camera, "pt_scene_desc": api.SceneDesc, "pt_render_params": api.RenderParams, "pt_stats": api.Stats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pt_hip.h"', '#include "pt_host.h"', 'int main(void) {']
    for cname, cls in mirror.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('printf("\\n");')
    lines += ['printf("pt_ray %zu\\n", sizeof(pt_ray)); printf("pt_hit %zu\\n", sizeof(pt_hit));', 'return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, str(src), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = dict((l.split()[0], [int(v) for v in l.split()[1:]]) for l in subprocess.run([exe], capture_output=True, text=True).stdout.splitlines())
    for cname, cls in mirror.items():
        assert out[cname][0] == C.sizeof(cls), cname
        assert out[cname][1:] == [getattr(cls, f).offset for f, _ in cls._fields_], cname
    assert out["pt_ray"] == [api.RAY_DTYPE.itemsize] and out["pt_hit"] == [api.HIT_DTYPE.itemsize]


def test_hip_library_reports_errors_without_a_gpu_or_init():
    lib = api.hip_lib()
    assert lib.pt_hip_abi_version() == 2
    h = C.c_void_p()
    rc = lib.pt_hip_scene_create(None, C.byref(h))
    assert rc < 0 and lib.pt_hip_last_error()


def test_export_round_trip_through_both_loaders(tmp_path):
    """SceneData.export_dir writes float32-exact OBJ/MTL/XML: the C++ loader must reproduce the arrays, BVH and camera."""
    from pt_hip import synth
    for sd in (synth.cornell_box(96, 96), synth.library_synth(detail=1, width=64, height=48)):
        folder = sd.export_dir(str(tmp_path), "cornell-box")
        h, desc = _load_with_host(folder, "cornell-box")
        _check_against_python_loader(desc, sd)
        for i in range(desc.n_mats):
            assert list(desc.mats[i].kd) == list(sd.mat_kd[i]) and list(desc.mats[i].ks) == list(sd.mat_ks[i]) and desc.mats[i].ns == sd.mat_ns[i]
        api.host_lib().pt_host_scene_free(h)


CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd", "bin", "PathTracer")


def test_cli_argument_errors_match_the_reference(tmp_path):
    """src/main.cpp:110-154: unknown tokens and bad counts print a message and return -1; load errors return 1."""
    import subprocess
    assert os.path.exists(CLI), "build the front end first (make -C path-tracer_amd cli)"
    r = subprocess.run([CLI, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 255 and "Unknown argument: --bogus" in r.stderr
    r = subprocess.run([CLI, "--spp", "0"], capture_output=True, text=True)
    assert r.returncode == 255 and '"--spp" argument expects a positive integer' in r.stderr
    r = subprocess.run([CLI, "-t"], capture_output=True, text=True)
    assert r.returncode == 255 and '"--threads" argument expects a positive integer' in r.stderr
    r = subprocess.run([CLI, "--bdpt"], capture_output=True, text=True)
    assert r.returncode == 255
    r = subprocess.run([CLI, "--split", "rows"], capture_output=True, text=True)
    assert r.returncode == 255 and '"--split" expects spp or tile' in r.stderr
    r = subprocess.run([CLI, "--no-gui", "--scenes", str(tmp_path), "library"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Error: ")
