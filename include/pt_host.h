/*
 * pt_host.h -- C ABI of the host-side substrate (libpt_host.so, plain C++17, no GPU dependency).
 *
 * These are the pieces of the reference's host that decide what the GPU sees and therefore have to be restated
 * exactly: the sweep-SAH BVH builder (its topology fixes the traversal order and with it which triangle wins a tie),
 * the camera matrices, the light table, and the scene file readers.  The output of these calls is what
 * pt_hip_scene_create (pt_hip.h) consumes.  File:line citations are relative to the reference repository root.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include "pt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* BVHTree::build + BVHTreeBuilder::build (src/bvh.cpp:13-24, 99-234; include/pt/bvh.h:63-116): top-down full-sweep
 * SAH over three centroid-sorted index arrays, leaves of 1..8 triangles, larger-area child stored first.
 * verts: 3 floats per vertex; vi: 3 one-based vertex indices per triangle.
 * On success *nodes (n_nodes entries) and *prim_ids (n_tris entries) are malloc'd; release with pt_host_free.
 * max_depth (nullable) receives the height of the tree (root = 1). */
int pt_host_build_bvh(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes,
                      uint32_t* n_nodes, uint32_t** prim_ids, uint32_t* max_depth);
void pt_host_free(void* p);

/* Camera::Camera (src/camera.cpp:20-80): camera-to-world, projection, raster transform, their inverse, proj_nume. */
int pt_host_camera(uint32_t width, uint32_t height, float fovy, const float eye[3], const float lookat[3], const float up[3],
                   pt_camera* out);

/* AreaLight::AreaLight (src/light.cpp:10-12) for a list of emissive triangles already in reference order; also fills
 * light_id (n_tris entries, -1 = none; a triangle listed twice keeps the later light, as Triangle::setLight does). */
int pt_host_make_lights(const float* verts, const uint32_t* vi, uint32_t n_tris, const uint32_t* light_tri,
                        const float* light_lemit, uint32_t n_lights, pt_light* lights, int32_t* light_id);

/* GaussianFilter table of ImageBlock (src/block.cpp:14-25, include/pt/filter.h:26-35): 33 floats. */
void pt_host_filter_table(float* out33);

/* ---- image files (no third-party codec) ------------------------------------------------------------------------------ */

/* Bitmap::savePNG (src/bitmap.cpp:122-144): width x height x 3 floats, row 0 first; with to_srgb != 0 every value goes
 * through Color3f::toSRGB (include/pt/color.h:57-69) first, then * 255, clamped, truncated to 8 bits. */
int pt_host_save_png(const char* path, const float* rgb, uint32_t width, uint32_t height, int to_srgb);
/* Bitmap::load (src/bitmap.cpp:68-93) for PNG (non-interlaced; grey, palette, RGB, with or without alpha, 1-16 bits),
 * JPEG (8-bit baseline and progressive Huffman-coded files, grey or YCbCr), binary PPM and PFM: 8-bit samples / 255
 * through Color3f::toLinearRGB (include/pt/color.h:70-83).  *rgb receives width x height x 3 floats (malloc'd; release with
 * pt_host_free).  JPEG decoders differ in how they round: values agree with stb_image's (which the reference uses) and
 * libjpeg's to within a level or two of the 8-bit scale, not bit for bit. */
int pt_host_load_image(const char* path, float** rgb, uint32_t* width, uint32_t* height);

/* ---- scene files --------------------------------------------------------------------------------------------------- */

typedef struct pt_host_scene pt_host_scene;

/* Scene::loadOBJ + Scene::loadXML + Scene::preprocess (src/scene.cpp:23-177, 222-263) for ./<dir>/<name>.obj|.xml:
 * reads mesh, materials (textures: PNG, JPEG, PPM and PFM as pt_host_load_image decodes them; anything else is reported
 * as an error), camera and lights, builds
 * the BVH and the light table.  The returned description stays valid until pt_host_scene_free. */
int pt_host_scene_load(const char* obj_path, const char* xml_path, pt_host_scene** out);
const pt_scene_desc* pt_host_scene_desc(const pt_host_scene* s);
void pt_host_scene_free(pt_host_scene* s);
const char* pt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_HOST_H */
