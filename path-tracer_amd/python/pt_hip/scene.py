"""Flat scene container shared by the test/bench glue.

A SceneData holds exactly what crosses the C boundary: indexed triangle mesh (1-based indices, 0 = absent, as in the
reference's include/pt/mesh.h:31-52), Phong materials, the light list in reference order (src/scene.cpp:251-260) and
the camera parameters of the scene XML (src/scene.cpp:127-159).  Loaders for the reference's OBJ/MTL/XML subset live
here too so that fixtures can be (re)generated from a scene directory; the C++ host has its own loaders.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field

import numpy as np


@dataclass
class SceneData:
    name: str
    verts: np.ndarray          # (nv,3) f32
    normals: np.ndarray        # (nn,3) f32
    uvs: np.ndarray            # (nt,2) f32
    vi: np.ndarray             # (ntri,3) u32, 1-based
    ni: np.ndarray             # (ntri,3) u32, 0 = absent
    ti: np.ndarray             # (ntri,3) u32, 0 = absent
    mat_id: np.ndarray         # (ntri,) u32
    mat_names: list
    mat_kd: np.ndarray         # (nm,3) f32
    mat_ks: np.ndarray         # (nm,3) f32
    mat_ns: np.ndarray         # (nm,) f32
    mat_tex: np.ndarray        # (nm,) i32, -1 = none
    textures: list = field(default_factory=list)   # list of (h,w,3) f32 linear RGB
    light_infos: list = field(default_factory=list)  # [(mtlname, (r,g,b))] in XML order
    width: int = 0
    height: int = 0
    fovy: float = 0.0
    eye: tuple = (0.0, 0.0, 0.0)
    lookat: tuple = (0.0, 0.0, 1.0)
    up: tuple = (0.0, 1.0, 0.0)

    @property
    def n_tris(self):
        return int(self.mat_id.shape[0])

    def lights(self):
        """Light list in reference order: XML order outer, triangle order inner (src/scene.cpp:251-260)."""
        tri, lemit = [], []
        names = np.array(self.mat_names, dtype=object)
        for mtlname, rad in self.light_infos:
            mats = np.nonzero(names == mtlname)[0]
            if mats.size == 0:
                continue
            sel = np.nonzero(np.isin(self.mat_id, mats))[0]
            tri.append(sel.astype(np.uint32))
            lemit.append(np.tile(np.asarray(rad, np.float32), (sel.size, 1)))
        if not tri:
            return np.zeros(0, np.uint32), np.zeros((0, 3), np.float32)
        return np.concatenate(tri), np.concatenate(lemit).astype(np.float32)

    def export_dir(self, root, name=None):
        """Write <root>/<name>/<name>.{obj,mtl,xml} in the subset of the formats the reference reads (src/scene.cpp:23-177),
        with float32-exact number formatting.  Textures are not exported."""
        name = name or self.name
        folder = os.path.join(root, name)
        os.makedirs(folder, exist_ok=True)
        f9 = lambda v: " ".join(repr(float(np.float32(x))) for x in v)
        with open(os.path.join(folder, name + ".mtl"), "w") as f:
            for i, n in enumerate(self.mat_names):
                f.write("newmtl %s\nKd %s\nKs %s\nNs %s\n" % (n, f9(self.mat_kd[i]), f9(self.mat_ks[i]), f9([self.mat_ns[i]])))
        with open(os.path.join(folder, name + ".obj"), "w") as f:
            f.write("mtllib %s.mtl\n" % name)
            for v in self.verts:
                f.write("v %s\n" % f9(v))
            for v in self.normals:
                f.write("vn %s\n" % f9(v))
            for v in self.uvs:
                f.write("vt %s\n" % f9(v))
            cur = -1
            for t in range(self.n_tris):
                if int(self.mat_id[t]) != cur:
                    cur = int(self.mat_id[t])
                    f.write("usemtl %s\n" % self.mat_names[cur])
                c = []
                for k in range(3):
                    v, n, u = int(self.vi[t, k]), int(self.ni[t, k]), int(self.ti[t, k])
                    c.append("%d" % v if not (n or u) else "%d/%s/%s" % (v, u or "", n or "") if n else "%d/%d" % (v, u))
                f.write("f %s\n" % " ".join(c))
        with open(os.path.join(folder, name + ".xml"), "w") as f:
            f.write('<?xml version="1.0" encoding="utf-8"?>\n<camera type="perspective" width="%d" height="%d" fovy="%s">\n'
                    % (self.width, self.height, f9([self.fovy])))
            for tag, v in (("eye", self.eye), ("lookat", self.lookat), ("up", self.up)):
                f.write('\t<%s x="%s" y="%s" z="%s"/>\n' % ((tag,) + tuple(f9([c]) for c in v)))
            f.write("</camera>\n")
            for n, rad in self.light_infos:
                f.write('<light mtlname="%s" radiance="%s"/>\n' % (n, ",".join(f9([c]) for c in rad)))
        return folder

    def save(self, path):
        if self.textures:
            raise ValueError("textured scenes are not stored as npz fixtures")
        np.savez_compressed(
            path, name=self.name, verts=self.verts, normals=self.normals, uvs=self.uvs, vi=self.vi, ni=self.ni, ti=self.ti,
            mat_id=self.mat_id, mat_names=np.array(self.mat_names), mat_kd=self.mat_kd, mat_ks=self.mat_ks, mat_ns=self.mat_ns,
            mat_tex=self.mat_tex, light_names=np.array([n for n, _ in self.light_infos]),
            light_rad=np.array([r for _, r in self.light_infos], np.float32).reshape(-1, 3),
            camera=np.array([self.width, self.height, self.fovy, *self.eye, *self.lookat, *self.up], np.float64))

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        cam = z["camera"]
        # camera values went through float32 in the XML reader of the reference (pugixml as_float)
        return SceneData(
            name=str(z["name"]), verts=z["verts"], normals=z["normals"], uvs=z["uvs"], vi=z["vi"], ni=z["ni"], ti=z["ti"],
            mat_id=z["mat_id"], mat_names=[str(s) for s in z["mat_names"]], mat_kd=z["mat_kd"], mat_ks=z["mat_ks"],
            mat_ns=z["mat_ns"], mat_tex=z["mat_tex"],
            light_infos=[(str(n), tuple(float(x) for x in r)) for n, r in zip(z["light_names"], z["light_rad"])],
            width=int(cam[0]), height=int(cam[1]), fovy=float(cam[2]), eye=tuple(cam[3:6]), lookat=tuple(cam[6:9]),
            up=tuple(cam[9:12]))


# ----------------------------------------------------------------------------- loaders (reference file formats)

def srgb_to_linear(v):
    """Color3f::toLinearRGB (include/pt/color.h:70-83) in float32."""
    v = np.asarray(v, np.float32)
    hi = np.power((v + np.float32(0.055)) * np.float32(1.0 / 1.055), np.float32(2.4), dtype=np.float32)
    return np.where(v <= np.float32(0.04045), v * np.float32(1.0 / 12.92), hi).astype(np.float32)


def load_image_linear(path):
    """Bitmap::load (src/bitmap.cpp:68-93): the file's 8-bit samples / 255, sRGB -> linear, the first three channels of every
    pixel; rows top to bottom, no flip.  The reference decodes with stb_image keeping the file's channel count; PIL gives
    the same samples for 8-bit RGB / RGBA files (JPEG decoders may differ in the last bit of a sample).  Images with fewer
    than three channels, which the reference would read out of step (it takes three consecutive bytes per pixel whatever
    the channel count), are expanded to grey RGB instead."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA"):
            im = im.convert("RGBA" if "A" in im.mode or "transparency" in im.info else "RGB")
        a = np.asarray(im, np.uint8)[..., :3]
    return srgb_to_linear(a.astype(np.float32) / np.float32(255.0))


def load_mtl(path):
    """Subset of MTL used by the reference (src/scene.cpp:87-110): Kd, Ks, Ns, map_Kd."""
    mats = []
    cur = None
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "newmtl":
                cur = {"name": " ".join(tok[1:]), "Kd": (0.0, 0.0, 0.0), "Ks": (0.0, 0.0, 0.0), "Ns": 1.0, "map_Kd": None}
                mats.append(cur)
            elif cur is not None and tok[0] in ("Kd", "Ks"):
                cur[tok[0]] = tuple(float(x) for x in tok[1:4])
            elif cur is not None and tok[0] == "Ns":
                cur["Ns"] = float(tok[1])
            elif cur is not None and tok[0] == "map_Kd":
                cur["map_Kd"] = tok[-1]
    return mats


def load_obj(path):
    """Triangulated OBJ reader with v/vt/vn, usemtl, mtllib; polygons are fan-triangulated like tinyobjloader does."""
    verts, normals, uvs = [], [], []
    vi, ni, ti, mat_id = [], [], [], []
    mats, mat_index, cur_mat = [], {}, -1
    base = os.path.dirname(path)
    with open(path) as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            tok = line.split()
            if not tok:
                continue
            k = tok[0]
            if k == "v":
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "vn":
                normals.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif k == "vt":
                uvs.append((float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0))
            elif k == "mtllib":
                mats = load_mtl(os.path.join(base, " ".join(tok[1:])))
                mat_index = {m["name"]: i for i, m in enumerate(mats)}
            elif k == "usemtl":
                cur_mat = mat_index.get(" ".join(tok[1:]), -1)
            elif k == "f":
                corners = []
                for c in tok[1:]:
                    p = c.split("/")
                    a = int(p[0])
                    b = int(p[1]) if len(p) > 1 and p[1] else 0
                    d = int(p[2]) if len(p) > 2 and p[2] else 0
                    a = a if a > 0 else len(verts) + 1 + a
                    b = b if b >= 0 else len(uvs) + 1 + b
                    d = d if d >= 0 else len(normals) + 1 + d
                    corners.append((a, b, d))
                for j in range(1, len(corners) - 1):
                    tri = (corners[0], corners[j], corners[j + 1])
                    vi.append([c[0] for c in tri])
                    ti.append([c[1] for c in tri])
                    ni.append([c[2] for c in tri])
                    mat_id.append(cur_mat)
    return (np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(normals, np.float32).reshape(-1, 3),
            np.asarray(uvs, np.float32).reshape(-1, 2), np.asarray(vi, np.uint32).reshape(-1, 3),
            np.asarray(ni, np.uint32).reshape(-1, 3), np.asarray(ti, np.uint32).reshape(-1, 3),
            np.asarray(mat_id, np.int64), mats)


def load_xml(path):
    """The two element kinds the reference reads (src/scene.cpp:127-173): <camera> and <light>."""
    txt = open(path).read()
    txt = re.sub(r"<!--.*?-->", "", txt, flags=re.S)

    def attrs(s):
        return dict(re.findall(r'(\w+)\s*=\s*"([^"]*)"', s))

    cam = re.search(r"<camera\b([^>]*)>(.*?)</camera>", txt, flags=re.S)
    ca = attrs(cam.group(1))
    if ca.get("type") != "perspective":
        raise ValueError("Only support perspective camera!")

    def vec(tag):
        a = attrs(re.search(r"<%s\b([^>]*)/?>" % tag, cam.group(2)).group(1))
        return tuple(float(np.float32(a[k])) for k in ("x", "y", "z"))

    lights = []
    for m in re.finditer(r"<light\b([^>]*)/?>", txt):
        a = attrs(m.group(1))
        rad = tuple(float(np.float32(x)) for x in a["radiance"].split(","))
        lights.append((a["mtlname"], rad))
    return dict(width=int(ca["width"]), height=int(ca["height"]), fovy=float(np.float32(ca["fovy"])), eye=vec("eye"),
                lookat=vec("lookat"), up=vec("up"), light_infos=lights)


def load_scene_dir(folder, name=None):
    """A scene directory in the reference's layout (<name>.obj + .mtl + .xml, textures named by map_Kd relative to the .mtl,
    src/scene.cpp:23-177).  A map_Kd file that cannot be read raises, as Scene::loadOBJ does."""
    name = name or os.path.basename(os.path.normpath(folder))
    verts, normals, uvs, vi, ni, ti, mat_id, mats = load_obj(os.path.join(folder, name + ".obj"))
    cam = load_xml(os.path.join(folder, name + ".xml"))
    if (mat_id < 0).any():
        raise ValueError("face without material")
    textures, mat_tex, seen = [], np.full(len(mats), -1, np.int32), {}
    for i, m in enumerate(mats):
        if m["map_Kd"]:
            tex_path = os.path.join(folder, m["map_Kd"].replace("\\", "/"))
            if tex_path not in seen:
                try:
                    textures.append(load_image_linear(tex_path))
                except OSError as e:
                    raise ValueError('Fail to load image file: "%s"! (%s)' % (tex_path, e))
                seen[tex_path] = len(textures) - 1
            mat_tex[i] = seen[tex_path]
    return SceneData(
        name=name, verts=verts, normals=normals, uvs=uvs, vi=vi, ni=ni, ti=ti, mat_id=mat_id.astype(np.uint32),
        mat_names=[m["name"] for m in mats], mat_kd=np.asarray([m["Kd"] for m in mats], np.float32).reshape(-1, 3),
        mat_ks=np.asarray([m["Ks"] for m in mats], np.float32).reshape(-1, 3),
        mat_ns=np.asarray([m["Ns"] for m in mats], np.float32), mat_tex=mat_tex, textures=textures, **cam)
