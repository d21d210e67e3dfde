// extern "C" entry points of libpt_host.so (declared in include/pt_host.h).
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "pt_host_internal.h"

namespace pt_host {
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
}  // namespace pt_host

using namespace pt_host;

extern "C" {

const char* pt_host_last_error(void) { return g_error.c_str(); }

void pt_host_free(void* p) { std::free(p); }

int pt_host_build_bvh(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes,
                      uint32_t* n_nodes, uint32_t** prim_ids, uint32_t* max_depth) {
    if (!verts || !vi || !nodes || !n_nodes || !prim_ids || n_tris == 0) {
        set_error("pt_host_build_bvh: null argument or empty mesh");
        return PT_ERR_INVALID;
    }
    for (size_t i = 0; i < 3 * size_t(n_tris); ++i)
        if (vi[i] == 0 || vi[i] > n_verts) {
            set_error("pt_host_build_bvh: vertex index out of range at corner " + std::to_string(i));
            return PT_ERR_INVALID;
        }
    try {
        std::vector<pt_bvh_node> nd;
        std::vector<uint32_t> ids;
        uint32_t depth = 0;
        build_bvh(verts, vi, n_tris, nd, ids, depth);
        *nodes = static_cast<pt_bvh_node*>(std::malloc(nd.size() * sizeof(pt_bvh_node)));
        *prim_ids = static_cast<uint32_t*>(std::malloc(ids.size() * sizeof(uint32_t)));
        if (!*nodes || !*prim_ids) {
            std::free(*nodes);
            std::free(*prim_ids);
            set_error("pt_host_build_bvh: out of memory");
            return PT_ERR_LIMIT;
        }
        std::memcpy(*nodes, nd.data(), nd.size() * sizeof(pt_bvh_node));
        std::memcpy(*prim_ids, ids.data(), ids.size() * sizeof(uint32_t));
        *n_nodes = uint32_t(nd.size());
        if (max_depth) *max_depth = depth;
    } catch (const std::exception& e) {
        set_error(std::string("pt_host_build_bvh: ") + e.what());
        return PT_ERR_INVALID;
    }
    return PT_OK;
}

int pt_host_camera(uint32_t width, uint32_t height, float fovy, const float eye[3], const float lookat[3], const float up[3],
                   pt_camera* out) {
    if (!out || !eye || !lookat || !up || width == 0 || height == 0) {
        set_error("pt_host_camera: bad argument");
        return PT_ERR_INVALID;
    }
    make_camera(width, height, fovy, eye, lookat, up, *out);
    return PT_OK;
}

int pt_host_make_lights(const float* verts, const uint32_t* vi, uint32_t n_tris, const uint32_t* light_tri,
                        const float* light_lemit, uint32_t n_lights, pt_light* lights, int32_t* light_id) {
    if (!verts || !vi || !light_id || (n_lights && (!light_tri || !light_lemit || !lights))) {
        set_error("pt_host_make_lights: null argument");
        return PT_ERR_INVALID;
    }
    for (uint32_t t = 0; t < n_tris; ++t) light_id[t] = -1;
    for (uint32_t i = 0; i < n_lights; ++i) {
        if (light_tri[i] >= n_tris) {
            set_error("pt_host_make_lights: light triangle out of range");
            return PT_ERR_INVALID;
        }
        lights[i].tri = light_tri[i];
        std::memcpy(lights[i].lemit, light_lemit + 3 * size_t(i), sizeof(float) * 3);
        lights[i].area = triangle_area(verts, vi, light_tri[i]);
        light_id[light_tri[i]] = int32_t(i);
    }
    return PT_OK;
}

void pt_host_filter_table(float* out33) { filter_table(out33); }

int pt_host_save_png(const char* path, const float* rgb, uint32_t width, uint32_t height, int to_srgb) {
    if (!path || !rgb || width == 0 || height == 0) {
        set_error("pt_host_save_png: bad argument");
        return PT_ERR_INVALID;
    }
    try {
        save_png(path, rgb, width, height, to_srgb != 0);
    } catch (const std::exception& e) {
        set_error(e.what());
        return PT_ERR_INVALID;
    }
    return PT_OK;
}

int pt_host_load_image(const char* path, float** rgb, uint32_t* width, uint32_t* height) {
    if (!path || !rgb || !width || !height) {
        set_error("pt_host_load_image: null argument");
        return PT_ERR_INVALID;
    }
    try {
        std::vector<float> px;
        if (!load_image_linear(path, px, *width, *height)) {
            set_error(std::string("Fail to load image file: \"") + path + "\"! (PNG, JPEG, binary PPM and PFM are decoded here)");
            return PT_ERR_INVALID;
        }
        *rgb = static_cast<float*>(std::malloc(px.size() * sizeof(float)));
        if (!*rgb) {
            set_error("pt_host_load_image: out of memory");
            return PT_ERR_DEVICE;
        }
        std::memcpy(*rgb, px.data(), px.size() * sizeof(float));
    } catch (const std::exception& e) {
        set_error(e.what());
        return PT_ERR_INVALID;
    }
    return PT_OK;
}

}  // extern "C"
