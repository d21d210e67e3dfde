// Scene files -> flat pt_scene_desc.
//
// Covers what the reference's Scene does between main() and render(): loadOBJ (src/scene.cpp:23-114), loadXML
// (src/scene.cpp:116-177) and preprocess (src/scene.cpp:222-263), without its third-party readers.  The OBJ subset is
// v / vn / vt / f / usemtl / mtllib (polygons are fan-triangulated, negative indices are relative); the MTL subset is
// newmtl / Kd / Ks / Ns / map_Kd; the XML subset is the <camera> element with its eye/lookat/up children and any
// number of <light mtlname radiance> elements.  Unknown elements (for example <backend .../>) are ignored, exactly
// as the reference's reader ignores them.  Textures are accepted as binary PPM (P6) or PFM (PF) only.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "pt_host_internal.h"

struct pt_host_scene {
    std::vector<float> verts, normals, uvs;
    std::vector<uint32_t> vi, ni, ti, mat_id;
    std::vector<int32_t> light_id;
    std::vector<std::string> mat_names;
    std::vector<pt_material> mats;
    std::vector<pt_light> lights;
    std::vector<std::vector<float>> tex_data;
    std::vector<pt_texture> textures;
    std::vector<pt_bvh_node> nodes;
    std::vector<uint32_t> prim_ids;
    pt_scene_desc desc{};
};

namespace pt_host {
namespace {

std::string folder_of(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

bool load_texture(const std::string& path, pt_host_scene& s, int32_t& id) {
    std::vector<float> rgb;
    uint32_t w = 0, h = 0;
    if (!pt_host::load_image_linear(path, rgb, w, h)) return false;  // PNG, JPEG, PPM, PFM (src/bitmap.cpp:68-93)
    id = int32_t(s.textures.size());
    s.tex_data.push_back(std::move(rgb));
    s.textures.push_back(pt_texture{w, h, nullptr});
    return true;
}

void load_mtl(const std::string& path, pt_host_scene& s) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Cannot open material file \"" + path + "\"");
    std::string line;
    int cur = -1;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::string key;
        if (!(is >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            std::string name;
            std::getline(is >> std::ws, name);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ')) name.pop_back();
            s.mat_names.push_back(name);
            s.mats.push_back(pt_material{{0, 0, 0}, {0, 0, 0}, 1.0f, -1});
            cur = int(s.mats.size()) - 1;
        } else if (cur >= 0 && key == "Kd") {
            is >> s.mats[cur].kd[0] >> s.mats[cur].kd[1] >> s.mats[cur].kd[2];
        } else if (cur >= 0 && key == "Ks") {
            is >> s.mats[cur].ks[0] >> s.mats[cur].ks[1] >> s.mats[cur].ks[2];
        } else if (cur >= 0 && key == "Ns") {
            is >> s.mats[cur].ns;
        } else if (cur >= 0 && key == "map_Kd") {
            std::string tex, tok;
            while (is >> tok) tex = tok;
            int32_t id = -1;
            // PNG and JPEG files are decoded here; a converted <stem>.ppm / .png / .pfm next to an unreadable file is taken instead
            const std::string full = folder_of(path) + tex;
            const size_t dot = full.find_last_of('.');
            const std::string stem = dot == std::string::npos ? full : full.substr(0, dot);
            if (!load_texture(full, s, id) && !load_texture(stem + ".ppm", s, id) && !load_texture(stem + ".png", s, id) && !load_texture(stem + ".pfm", s, id))
                throw std::runtime_error("Fail to load image file: \"" + full + "\"! (PNG, JPEG, PPM and PFM are decoded here)");
            s.mats[cur].tex = id;
        }
    }
}

void load_obj(const std::string& path, pt_host_scene& s) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Cannot open file \"" + path + "\"");
    std::map<std::string, int> by_name;
    int cur_mat = -1;
    std::string line;
    struct Corner {
        long v, t, n;
    };
    std::vector<Corner> poly;
    while (std::getline(f, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e;
            p += 1;
            for (int k = 0; k < 3; ++k) {
                s.verts.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'v' && p[1] == 'n') {
            char* e;
            p += 2;
            for (int k = 0; k < 3; ++k) {
                s.normals.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'v' && p[1] == 't') {
            char* e;
            p += 2;
            for (int k = 0; k < 2; ++k) {
                s.uvs.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            poly.clear();
            ++p;
            for (;;) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                Corner c{0, 0, 0};
                char* e;
                c.v = std::strtol(p, &e, 10);
                p = e;
                if (*p == '/') {
                    ++p;
                    if (*p != '/') {
                        c.t = std::strtol(p, &e, 10);
                        p = e;
                    }
                    if (*p == '/') {
                        ++p;
                        c.n = std::strtol(p, &e, 10);
                        p = e;
                    }
                }
                if (c.v < 0) c.v += long(s.verts.size() / 3) + 1;
                if (c.t < 0) c.t += long(s.uvs.size() / 2) + 1;
                if (c.n < 0) c.n += long(s.normals.size() / 3) + 1;
                poly.push_back(c);
            }
            if (poly.size() < 3) continue;
            if (cur_mat < 0) throw std::runtime_error("face without material in \"" + path + "\"");
            for (size_t j = 1; j + 1 < poly.size(); ++j) {
                const Corner tri[3] = {poly[0], poly[j], poly[j + 1]};
                for (const Corner& c : tri) {  // stored 1-based, 0 = absent (src/scene.cpp:69-74)
                    s.vi.push_back(uint32_t(c.v));
                    s.ti.push_back(uint32_t(c.t));
                    s.ni.push_back(uint32_t(c.n));
                }
                s.mat_id.push_back(uint32_t(cur_mat));
            }
        } else if (!std::strncmp(p, "usemtl", 6)) {
            std::string name(p + 6);
            const size_t b = name.find_first_not_of(" \t");
            name = b == std::string::npos ? "" : name.substr(b);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ')) name.pop_back();
            auto it = by_name.find(name);
            cur_mat = it == by_name.end() ? -1 : it->second;
        } else if (!std::strncmp(p, "mtllib", 6)) {
            std::string name(p + 6);
            const size_t b = name.find_first_not_of(" \t");
            name = b == std::string::npos ? "" : name.substr(b);
            while (!name.empty() && (name.back() == '\r' || name.back() == ' ')) name.pop_back();
            load_mtl(folder_of(path) + name, s);
            for (size_t i = 0; i < s.mat_names.size(); ++i) by_name.emplace(s.mat_names[i], int(i));
        }
    }
    if (s.mat_id.empty()) throw std::runtime_error("no faces in \"" + path + "\"");
}

// --- tiny XML subset ------------------------------------------------------------------------------------------------

std::string attr(const std::string& tag, const std::string& name) {
    size_t p = 0;
    while ((p = tag.find(name, p)) != std::string::npos) {
        const bool start_ok = p == 0 || tag[p - 1] == ' ' || tag[p - 1] == '\t' || tag[p - 1] == '\n';
        size_t q = p + name.size();
        while (q < tag.size() && (tag[q] == ' ' || tag[q] == '\t')) ++q;
        if (start_ok && q < tag.size() && tag[q] == '=') {
            const size_t a = tag.find('"', q);
            const size_t b = a == std::string::npos ? a : tag.find('"', a + 1);
            if (b != std::string::npos) return tag.substr(a + 1, b - a - 1);
        }
        p += name.size();
    }
    return std::string();
}

// body of the first <name ...> element at or after pos; returns the tag text (without <>), sets end past it
bool next_tag(const std::string& xml, const std::string& name, size_t& pos, std::string& tag) {
    for (;;) {
        const size_t p = xml.find("<" + name, pos);
        if (p == std::string::npos) return false;
        const char c = xml[p + 1 + name.size()];
        const size_t e = xml.find('>', p);
        if (e == std::string::npos) return false;
        pos = e + 1;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '>' || c == '/') {
            tag = xml.substr(p + 1, e - p - 1);
            return true;
        }
    }
}

struct XmlScene {
    uint32_t width = 0, height = 0;
    float fovy = 0;
    float eye[3] = {0, 0, 0}, lookat[3] = {0, 0, 0}, up[3] = {0, 0, 0};
    std::vector<std::pair<std::string, std::array<float, 3>>> lights;
};

XmlScene load_xml(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Cannot open file \"" + path + "\"");
    std::stringstream ss;
    ss << f.rdbuf();
    std::string xml = ss.str();
    for (size_t a; (a = xml.find("<!--")) != std::string::npos;) {  // strip comments
        const size_t b = xml.find("-->", a);
        xml.erase(a, b == std::string::npos ? std::string::npos : b + 3 - a);
    }
    XmlScene out;
    size_t pos = 0;
    std::string tag;
    if (!next_tag(xml, "camera", pos, tag)) throw std::runtime_error("Only support perspective camera!");
    if (attr(tag, "type") != "perspective") throw std::runtime_error("Only support perspective camera!");  // src/scene.cpp:129-130
    out.width = uint32_t(std::strtoul(attr(tag, "width").c_str(), nullptr, 10));
    out.height = uint32_t(std::strtoul(attr(tag, "height").c_str(), nullptr, 10));
    out.fovy = float(std::strtod(attr(tag, "fovy").c_str(), nullptr));
    const size_t cam_end = xml.find("</camera>", pos);
    const std::string body = xml.substr(pos, cam_end == std::string::npos ? std::string::npos : cam_end - pos);
    auto vec = [&](const char* name, float* dst) {
        size_t p = 0;
        std::string t;
        if (!next_tag(body, name, p, t)) return;
        dst[0] = float(std::strtod(attr(t, "x").c_str(), nullptr));
        dst[1] = float(std::strtod(attr(t, "y").c_str(), nullptr));
        dst[2] = float(std::strtod(attr(t, "z").c_str(), nullptr));
    };
    vec("eye", out.eye);
    vec("lookat", out.lookat);
    vec("up", out.up);
    pos = 0;
    while (next_tag(xml, "light", pos, tag)) {  // src/scene.cpp:163-173
        std::array<float, 3> rad{0, 0, 0};
        std::string r = attr(tag, "radiance");
        size_t p = 0;
        for (int k = 0; k < 3 && p <= r.size(); ++k) {
            const size_t c = r.find(',', p);
            rad[k] = float(std::strtod(r.substr(p, c == std::string::npos ? c : c - p).c_str(), nullptr));
            if (c == std::string::npos) break;
            p = c + 1;
        }
        out.lights.emplace_back(attr(tag, "mtlname"), rad);
    }
    return out;
}

}  // namespace
}  // namespace pt_host

using namespace pt_host;

extern "C" {

int pt_host_scene_load(const char* obj_path, const char* xml_path, pt_host_scene** out) {
    if (!obj_path || !xml_path || !out) {
        set_error("pt_host_scene_load: null argument");
        return PT_ERR_INVALID;
    }
    pt_host_scene* s = new pt_host_scene();
    try {
        load_obj(obj_path, *s);
        const XmlScene x = load_xml(xml_path);
        const uint32_t n_tris = uint32_t(s->mat_id.size());
        const uint32_t n_verts = uint32_t(s->verts.size() / 3);
        for (uint32_t v : s->vi)
            if (v == 0 || v > n_verts) throw std::runtime_error("vertex index out of range");
        for (uint32_t v : s->ni)
            if (v > s->normals.size() / 3) throw std::runtime_error("normal index out of range");
        for (uint32_t v : s->ti)
            if (v > s->uvs.size() / 2) throw std::runtime_error("texture coordinate index out of range");

        // Scene::createAreaLights (src/scene.cpp:251-263): XML order outer, triangle order inner, match by material name
        std::vector<uint32_t> ltri;
        std::vector<float> lemit;
        for (const auto& li : x.lights)
            for (uint32_t t = 0; t < n_tris; ++t)
                if (s->mat_names[s->mat_id[t]] == li.first) {
                    ltri.push_back(t);
                    lemit.insert(lemit.end(), li.second.begin(), li.second.end());
                }
        s->lights.resize(ltri.size());
        s->light_id.resize(n_tris);
        if (pt_host_make_lights(s->verts.data(), s->vi.data(), n_tris, ltri.data(), lemit.data(), uint32_t(ltri.size()),
                                s->lights.data(), s->light_id.data()) != PT_OK)
            throw std::runtime_error(pt_host_last_error());

        uint32_t depth = 0;
        build_bvh(s->verts.data(), s->vi.data(), n_tris, s->nodes, s->prim_ids, depth);
        for (size_t i = 0; i < s->textures.size(); ++i) s->textures[i].rgb = s->tex_data[i].data();

        pt_scene_desc& d = s->desc;
        d.n_verts = n_verts;
        d.n_normals = uint32_t(s->normals.size() / 3);
        d.n_uvs = uint32_t(s->uvs.size() / 2);
        d.n_tris = n_tris;
        d.n_mats = uint32_t(s->mats.size());
        d.n_lights = uint32_t(s->lights.size());
        d.n_textures = uint32_t(s->textures.size());
        d.n_nodes = uint32_t(s->nodes.size());
        d.verts = s->verts.data();
        d.normals = s->normals.data();
        d.uvs = s->uvs.data();
        d.vi = s->vi.data();
        d.ni = s->ni.data();
        d.ti = s->ti.data();
        d.mat_id = s->mat_id.data();
        d.light_id = s->light_id.data();
        d.mats = s->mats.data();
        d.lights = s->lights.data();
        d.textures = s->textures.data();
        d.nodes = s->nodes.data();
        d.prim_ids = s->prim_ids.data();
        make_camera(x.width, x.height, x.fovy, x.eye, x.lookat, x.up, d.camera);
    } catch (const std::exception& e) {
        set_error(e.what());
        delete s;
        return PT_ERR_INVALID;
    }
    *out = s;
    return PT_OK;
}

const pt_scene_desc* pt_host_scene_desc(const pt_host_scene* s) { return s ? &s->desc : nullptr; }

void pt_host_scene_free(pt_host_scene* s) { delete s; }

}  // extern "C"
