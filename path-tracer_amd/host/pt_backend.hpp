// C++ mirror of the reference's operator interface for the accelerated path, on top of the C ABI (pt_hip.h, pt_host.h).
//
//   reference                                       here
//   class Accel (include/pt/accel.h:10-24)          pt::hip::Accel         build(), rayIntersect(ray, its), rayIntersect(ray),
//                                                                          plus the batched forms a GPU wants
//   class Integrator (include/pt/integrator.h:9-19) pt::hip::Integrator    Path / Geometry / BaseColor by enum, toString()
//   render(scene, sampler, integrator, result)      pt::hip::render()      fills an ImageBlock-shaped film
//   (src/main.cpp:76-98)
//   PathTracerException (include/pt/common.h:75-80) pt::hip::Exception     thrown for every non-zero return code
//
// The types are header-only so that a reference maintainer can drop this file next to main.cpp and link libpt_hip.so
// and libpt_host.so; nothing here depends on Eigen, TBB or any other reference dependency.
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pt_hip.h"
#include "pt_host.h"

namespace pt {
namespace hip {

class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& what) : std::runtime_error(what) {}
};

inline void check(int rc) {
    if (rc != 0) throw Exception(pt_hip_last_error());
}
inline void check_host(int rc) {
    if (rc != 0) throw Exception(pt_host_last_error());
}

// include/pt/ray.h:7-27
struct Ray {
    float org[3], dir[3], min_dis = 0.0f, max_dis = 3.402823466e38f;
};

// What Intersection carries before complete() (src/shape.cpp:140-166): the triangle and the barycentrics.
struct Intersection {
    int32_t prim = -1;
    float t = 0.0f, bary[3] = {0, 0, 0};
};

// Scene as loaded by Scene::loadOBJ / loadXML / preprocess (src/scene.cpp:23-207), host side.
class Scene {
public:
    Scene(const std::string& obj_path, const std::string& xml_path) { check_host(pt_host_scene_load(obj_path.c_str(), xml_path.c_str(), &m_scene)); }
    ~Scene() { pt_host_scene_free(m_scene); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    const pt_scene_desc* desc() const { return pt_host_scene_desc(m_scene); }
    uint32_t width() const { return desc()->camera.width; }
    uint32_t height() const { return desc()->camera.height; }
    std::string toString() const {
        const pt_scene_desc* d = desc();
        return "Scene[\n  triangles = " + std::to_string(d->n_tris) + ",\n  lights = " + std::to_string(d->n_lights) +
               ",\n  materials = " + std::to_string(d->n_mats) + ",\n  bvh nodes = " + std::to_string(d->n_nodes) + ",\n  screen = " +
               std::to_string(width()) + "x" + std::to_string(height()) + "\n]";
    }

private:
    pt_host_scene* m_scene = nullptr;
};

// Accel on the GPU.  build() uploads the scene with the host-built BVH; the single-ray forms exist for interface parity
// (one launch per call: use the batched forms for anything but debugging).
class Accel {
public:
    // device < 0: the GPU pt_hip_init selected; otherwise the handle is bound to that GPU (several per process are fine)
    Accel(const Scene* scene, int traversal = PT_TRAVERSE_FAST, int device = -1) : m_host(scene), m_traversal(traversal), m_device(device) {}
    ~Accel() { pt_hip_scene_destroy(m_dev); }
    Accel(const Accel&) = delete;
    Accel& operator=(const Accel&) = delete;

    void build() {
        if (!m_dev) check(m_device < 0 ? pt_hip_scene_create(m_host->desc(), &m_dev) : pt_hip_scene_create_on(m_device, m_host->desc(), &m_dev));
    }
    bool rayIntersect(const Ray& ray, Intersection& its) {
        pt_hit h;
        const pt_ray r = toAbi(ray);
        check(pt_hip_intersect(m_dev, &r, 1, &h, m_traversal));
        its.prim = h.prim;
        its.t = h.t;
        its.bary[0] = 1.0f - h.u - h.v;
        its.bary[1] = h.u;
        its.bary[2] = h.v;
        return h.prim >= 0;
    }
    bool rayIntersect(const Ray& ray) {
        uint8_t occ = 0;
        const pt_ray r = toAbi(ray);
        check(pt_hip_occluded(m_dev, &r, 1, &occ, nullptr, m_traversal));
        return occ != 0;
    }
    void rayIntersect(const std::vector<pt_ray>& rays, std::vector<pt_hit>& hits) {
        hits.resize(rays.size());
        check(pt_hip_intersect(m_dev, rays.data(), rays.size(), hits.data(), m_traversal));
    }
    void rayIntersect(const std::vector<pt_ray>& rays, std::vector<uint8_t>& occluded) {
        occluded.resize(rays.size());
        check(pt_hip_occluded(m_dev, rays.data(), rays.size(), occluded.data(), nullptr, m_traversal));
    }
    std::string toString() const { return std::string("HipBVH[traversal = ") + (m_traversal == PT_TRAVERSE_EXACT ? "exact" : "fast") + "]"; }
    pt_hip_scene* handle() const { return m_dev; }
    int traversal() const { return m_traversal; }

private:
    static pt_ray toAbi(const Ray& r) { return pt_ray{{r.org[0], r.org[1], r.org[2]}, {r.dir[0], r.dir[1], r.dir[2]}, r.min_dis, r.max_dis}; }
    const Scene* m_host;
    pt_hip_scene* m_dev = nullptr;
    int m_traversal;
    int m_device;
};

// The three integrators of src/integrator.cpp selected by value; Li() is not exposed per sample on purpose (the
// per-sample entry point for tests is pt_hip_li), rendering goes through render().
class Integrator {
public:
    explicit Integrator(int kind) : m_kind(kind) {}
    int kind() const { return m_kind; }
    std::string toString() const {
        return m_kind == PT_INTEGRATOR_PATH ? "PathIntegrator[]" : m_kind == PT_INTEGRATOR_NORMAL ? "GeometryIntegrator[]" : "BaseColorIntegrator[]";
    }

private:
    int m_kind;
};
struct PathIntegrator : Integrator {
    PathIntegrator() : Integrator(PT_INTEGRATOR_PATH) {}
};
struct GeometryIntegrator : Integrator {
    GeometryIntegrator() : Integrator(PT_INTEGRATOR_NORMAL) {}
};
struct BaseColorIntegrator : Integrator {
    BaseColorIntegrator() : Integrator(PT_INTEGRATOR_ALBEDO) {}
};

// ImageBlock-shaped accumulation buffer (include/pt/block.h:20-60): (H+4) x (W+4) RGBA, weight in A, 2-pixel border.
struct Film {
    uint32_t width = 0, height = 0;
    std::vector<float> rgba;
    Film(uint32_t w, uint32_t h) : width(w), height(h), rgba(size_t(4) * (w + 4) * (h + 4), 0.0f) {}
    void clear() { std::fill(rgba.begin(), rgba.end(), 0.0f); }
    void add(const Film& o) {
        for (size_t i = 0; i < rgba.size(); ++i) rgba[i] += o.rgba[i];
    }
    // writeBitmap (src/main.cpp:22-52): Color4f::divideByFilterWeight (include/pt/color.h:135-140) of the interior
    std::vector<float> toRGB() const {
        std::vector<float> out(size_t(3) * width * height);
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t x = 0; x < width; ++x) {
                const float* p = &rgba[(size_t(y + 2) * (width + 4) + (x + 2)) * 4];
                float* o = &out[(size_t(y) * width + x) * 3];
                if (p[3] != 0.0f) {
                    o[0] = p[0] / p[3];
                    o[1] = p[1] / p[3];
                    o[2] = p[2] / p[3];
                } else {
                    o[0] = o[1] = o[2] = 0.0f;
                }
            }
        return out;
    }
};

// render() of src/main.cpp:76-98 for sample numbers [spp_begin, spp_end): adds into `result`.  With block_count > 1
// only this rank's share of the 32x32 pixel blocks is rendered (pt_render_params::block_rank / block_count).
inline pt_stats render(Accel* accel, const Integrator* integrator, uint32_t spp_begin, uint32_t spp_end, Film* result,
                       uint32_t block_rank = 0, uint32_t block_count = 0) {
    pt_render_params p{};
    p.spp_begin = spp_begin;
    p.spp_end = spp_end;
    p.max_bounce = 32;
    p.block_rank = uint16_t(block_rank);
    p.block_count = uint16_t(block_count);
    p.integrator = integrator->kind();
    p.traversal = accel->traversal();
    Film part(result->width, result->height);
    pt_stats st{};
    check(pt_hip_render(accel->handle(), &p, part.rgba.data(), &st));
    result->add(part);
    return st;
}

// The same frame on several accelerators of one scene, one per GPU, from this process (pt_hip_render_multi): the sample
// range is cut into contiguous shares, or with by_blocks every accelerator renders its share of the 32x32 pixel blocks;
// the films are summed on the first one's GPU.  Adds into `result`.
inline pt_stats render(const std::vector<Accel*>& accels, const Integrator* integrator, uint32_t spp_begin, uint32_t spp_end, Film* result,
                       bool by_blocks = false) {
    if (accels.size() == 1) return render(accels[0], integrator, spp_begin, spp_end, result);
    pt_render_params p{};
    p.spp_begin = spp_begin;
    p.spp_end = spp_end;
    p.max_bounce = 32;
    p.block_count = by_blocks ? uint16_t(accels.size()) : uint16_t(0);
    p.integrator = integrator->kind();
    p.traversal = accels[0]->traversal();
    std::vector<pt_hip_scene*> handles;
    for (Accel* a : accels) handles.push_back(a->handle());
    Film part(result->width, result->height);
    pt_stats st{};
    check(pt_hip_render_multi(handles.data(), int(handles.size()), &p, part.rgba.data(), &st));
    result->add(part);
    return st;
}

}  // namespace hip
}  // namespace pt
