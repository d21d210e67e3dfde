/*
 * pt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a scalar CPU restatement of the
 * reference renderer's hot path (LinzhouLi/Path-Tracer).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * HIP product never links, includes or calls anything in this directory.
 *
 * Pinning status: the reference cannot be built in this image (all of its
 * ext/ dependencies are absent), and it ships no unit tests.  The oracle is
 * pinned by the two float images the real reference binary wrote for the one
 * scene whose mesh is present (scenes/veach-mis/normal.exr, albedo.exr:
 * camera, Sobol' pixel jitter, closest hit, normal interpolation, film).
 * Everything past the primary hit (Phong, light sampling, MIS, roulette) is
 * restated from the source text only: "parity unpinned" at function level.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Flat scene input.  Index arrays are 1-based, 0 = absent (mesh.h:31-52). */
typedef struct orc_scene_desc {
    uint32_t n_verts, n_normals, n_uvs, n_tris, n_mats, n_lights, n_textures;
    const float*    verts;    /* 3 per vertex */
    const float*    normals;  /* 3 per normal */
    const float*    uvs;      /* 2 per uv */
    const uint32_t* vi;       /* 3 per triangle */
    const uint32_t* ni;       /* 3 per triangle */
    const uint32_t* ti;       /* 3 per triangle */
    const uint32_t* mat_id;   /* 1 per triangle */
    const float*    mat_kd;   /* 3 per material */
    const float*    mat_ks;   /* 3 per material */
    const float*    mat_ns;   /* 1 per material */
    const int32_t*  mat_tex;  /* 1 per material, -1 = none */
    const uint32_t* light_tri;   /* n_lights, in reference order (scene.cpp:251-260) */
    const float*    light_lemit; /* 3 per light */
    const uint32_t* tex_w;    /* n_textures */
    const uint32_t* tex_h;
    const float* const* tex_rgb; /* n_textures pointers, linear RGB, row-major h x w x 3 */
    /* camera (camera.cpp:46-80) */
    uint32_t width, height;
    float fovy;
    float eye[3], lookat[3], up[3];
} orc_scene_desc;

/* ray = org[3], dir[3], tmin, tmax (8 floats) */
typedef struct orc_hit {
    int32_t prim;   /* triangle index, -1 = miss */
    float   t;
    float   bary[3];
} orc_hit;

typedef struct orc_surface {
    float p[3], n[3], ng[3], uv[2];
} orc_surface;

typedef struct orc_stats {
    uint64_t closest_rays, any_rays, paths;
    uint64_t nodes_visited, tris_tested; /* traversal work counters */
    double   seconds;
} orc_stats;

enum { ORC_INTEGRATOR_PATH = 0, ORC_INTEGRATOR_NORMAL = 1, ORC_INTEGRATOR_ALBEDO = 2 };

orc_scene* orc_scene_create(const orc_scene_desc* d);
/* Test hook: the same scene over a tree built elsewhere (n_nodes nodes: 6 floats min,max each, first, count; prim_ids
 * n_tris) instead of the restated builder's -- for scenes where the single-threaded build would take minutes. */
orc_scene* orc_scene_create_with_tree(const orc_scene_desc* d, size_t n_nodes, const float* boxes, const uint32_t* first,
                                      const uint32_t* count, const uint32_t* prim_ids);
void       orc_scene_destroy(orc_scene* s);

/* BVH as built by the restated sweep-SAH builder (bvh.cpp:99-234). */
size_t orc_bvh_num_nodes(const orc_scene* s);
void   orc_bvh_get(const orc_scene* s, float* boxes /*6 per node: min,max*/,
                   uint64_t* first, uint64_t* count, uint64_t* prim_ids /*n_tris*/);

/* Geometry unit functions. */
int  orc_tri_intersect(const orc_scene* s, uint32_t prim, const float* ray8, float* bary3, float* t);
int  orc_aabb_intersect(const float* box6, const float* ray8);
void orc_intersect(const orc_scene* s, const float* rays8, size_t n, orc_hit* out, orc_surface* surf /*nullable*/);
void orc_occluded(const orc_scene* s, const float* rays8, size_t n, uint8_t* out, int32_t* first_prim /*nullable*/);
void orc_traversal_counters(const orc_scene* s, uint64_t* nodes, uint64_t* tris); /* since creation, single-thread calls only */

/* Sampler. */
uint64_t orc_sobol_index(uint32_t m, uint64_t frame, int32_t px, int32_t py);
float    orc_sobol_sample(int64_t index, int dim);
void     orc_pixel_sample(const orc_scene* s, int32_t px, int32_t py, uint32_t sample, float* pixel_sample2);

/* Camera. */
void orc_camera_ray(const orc_scene* s, float sx, float sy, float* ray8);
void orc_camera_get(const orc_scene* s, float* sample2camera16, float* camera2world16, float* proj_nume);

/* Shading models, evaluated at a surface point given by (prim, bary). */
void  orc_brdf_eval(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, const float* wi3, float* f3);
float orc_brdf_pdf(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, const float* wi3);
/* returns specular flag; out: wi[3], pdf, f[3] */
int   orc_brdf_sample(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, float uc, const float* u2,
                      float* wi3, float* pdf, float* f3);
/* light sampling from a surface point: out L[3], wi[3], p[3], n[3], pdfArea, pdfDir */
void  orc_light_sample(const orc_scene* s, uint32_t light, uint32_t surf_prim, const float* surf_bary3, const float* u2, float* out14);
float orc_light_pdf(const orc_scene* s, uint32_t light, uint32_t hit_prim, const float* hit_bary3, const float* ray8);

/* One radiance sample.  ray_log (nullable): 9 floats per ray (ray8 + kind 0=closest,1=any), at most max_log rays. */
void orc_li(const orc_scene* s, int integrator, int32_t px, int32_t py, uint32_t sample, float* rgb3, float* pixel_sample2,
            float* ray_log, size_t max_log, size_t* n_log, int32_t* dims_used);

/* Film. */
void orc_filter_table(float* out33);
void orc_film_put(float* film_rgba, uint32_t width, uint32_t height, const float* pos2, const float* rgb3);

/* Whole frame: samples [spp_begin, spp_end) of every pixel into film (H+4)x(W+4)x4, zeroed by the callee.
 * Blocks of 32x32 pixels in the reference's spiral order; nthreads workers; block results are merged in
 * spiral order so the output does not depend on nthreads.
 * max_blocks > 0 renders only the first max_blocks blocks (bounded timing samples). */
void orc_render(const orc_scene* s, int integrator, uint32_t spp_begin, uint32_t spp_end, int nthreads, uint32_t max_blocks,
                float* film_rgba, orc_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
