/*
 * pt_hip.h -- C ABI of the MI355X rendering backend (libpt_hip.so).
 *
 * This is the drop-in boundary for the hot path of LinzhouLi/Path-Tracer: BVH closest-hit and any-hit traversal,
 * the ray-triangle test and the MIS path-tracing integrator loop.  Plain C types only; no exceptions cross it.
 * Every function returns 0 on success and a negative pt_status on failure; pt_hip_last_error() gives the message
 * (thread-local), which a C++ host rethrows as the reference's PathTracerException (include/pt/common.h:75-80).
 *
 * Multi-GPU rendering splits one frame by sample number (or by 32x32 pixel blocks) and sums the films with one
 * reduction.  Two hosts for that: one process per GPU (pt_hip_init selects the GPU, the films are reduced by the caller,
 * e.g. with RCCL), or one process driving several GPUs: a scene handle is bound to the GPU it was created on
 * (pt_hip_scene_create_on) and owns a stream there, and pt_hip_render_multi renders one frame on a set of handles.
 *
 * A handle is not thread-safe: one host thread per handle at a time, as the reference's render thread
 * (src/main.cpp:185); different handles may be driven from different threads concurrently.
 *
 * File:line citations are relative to the reference repository root.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_HIP_ABI_VERSION 2

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument / scene description fails validation */
    PT_ERR_DEVICE = -2,    /* HIP runtime error (message carries the HIP error string) */
    PT_ERR_NOT_INIT = -3,  /* pt_hip_init has not been called */
    PT_ERR_LIMIT = -4      /* scene exceeds a documented limit (BVH depth, triangle count) */
} pt_status;

/* ---- flat scene description (host memory, read-only during pt_hip_scene_create, not retained) -------------------- */

/* Binary BVH node exactly as the reference builds it (include/pt/bvh.h:13-45), with 32-bit indices.
 * Interior: count == 0, children are nodes first and first+1.  Leaf: count in [1,8], first indexes prim_ids. */
typedef struct pt_bvh_node {
    float    lo[3];
    float    hi[3];
    uint32_t first;
    uint32_t count;
} pt_bvh_node;

/* Phong material (include/pt/material.h:17-62): Kd, Ks, Ns, optional diffuse texture (-1 = none). */
typedef struct pt_material {
    float   kd[3];
    float   ks[3];
    float   ns;
    int32_t tex;
} pt_material;

/* One area light per emissive triangle, in the reference's order: XML <light> order outer, triangle order inner
 * (src/scene.cpp:251-260).  area = Triangle::surfaceArea() (src/shape.cpp:32-36). */
typedef struct pt_light {
    uint32_t tri;
    float    lemit[3];
    float    area;
} pt_light;

/* Linear-RGB float texture, row-major h x w x 3 (src/bitmap.cpp:68-93). */
typedef struct pt_texture {
    uint32_t     w, h;
    const float* rgb;
} pt_texture;

/* Camera after the constructor of src/camera.cpp:46-80 has run.  Matrices are row-major 4x4. */
typedef struct pt_camera {
    uint32_t width, height;
    float    eye[3];
    float    sample2camera[16];
    float    camera2world[16];
    float    proj_nume;
} pt_camera;

typedef struct pt_scene_desc {
    uint32_t n_verts, n_normals, n_uvs, n_tris, n_mats, n_lights, n_textures, n_nodes;
    const float*       verts;    /* 3 per vertex */
    const float*       normals;  /* 3 per normal (may be NULL when n_normals == 0) */
    const float*       uvs;      /* 2 per uv     (may be NULL when n_uvs == 0) */
    const uint32_t*    vi;       /* 3 per triangle, 1-based (include/pt/mesh.h:31-35) */
    const uint32_t*    ni;       /* 3 per triangle, 1-based, 0 = absent (mesh.h:37-44) */
    const uint32_t*    ti;       /* 3 per triangle, 1-based, 0 = absent (mesh.h:46-53) */
    const uint32_t*    mat_id;   /* 1 per triangle */
    const int32_t*     light_id; /* 1 per triangle, -1 = not an emitter */
    const pt_material* mats;
    const pt_light*    lights;
    const pt_texture*  textures;
    pt_camera          camera;
    const pt_bvh_node* nodes;    /* n_nodes, root = 0; NULL with n_nodes == 0 and prim_ids == NULL: pt_hip_scene_create builds
                                    the reference's tree itself (pt_hip_build_bvh) */
    const uint32_t*    prim_ids; /* n_tris: leaf order -> triangle index (BVHTree::m_prim_ids) */
} pt_scene_desc;

/* ---- rays and hits ------------------------------------------------------------------------------------------------ */

/* include/pt/ray.h:7-27 without the unused reciprocal direction. */
typedef struct pt_ray {
    float org[3];
    float dir[3];
    float tmin;
    float tmax;
} pt_ray;

/* Closest hit.  prim = triangle index or -1; barycentrics are (1-u-v, u, v) as in src/shape.cpp:86. */
typedef struct pt_hit {
    int32_t prim;
    float   t;
    float   u;
    float   v;
} pt_hit;

/* Traversal modes.
 * EXACT replays the reference's visit order on the binary tree (src/bvh.cpp:26-86): pop, cull interior nodes with the
 * current max distance, never box-test leaves, accept t <= max, last accepted wins.
 * FAST is an ordered (nearest slot first, distance-culled) traversal of a 4-wide node layout collapsed from the same tree,
 * with slot boxes quantised outwards to 16 bits.  Interior nodes are only navigated, with a conservative slab test; the
 * reference's own box decision is taken once per query, on the parent box of the leaf that holds the result, which decides
 * whether the reference reaches that leaf at all (its slab test, src/aabb.cpp:6-35, is monotone under box nesting: a ray
 * that it accepts for a box it accepts for every enclosing box).  A query is repeated in the reference's order whenever two
 * candidates (or a candidate and the ray's far end) lie within a 2^-13 relative distance of each other, and rays with a
 * zero, tiny, huge or non-finite direction component are traced in the reference's order from the start, so that the
 * reported triangle, t, u and v are the reference's in both modes; occlusion queries are order-independent by
 * construction.  For occluded rays FAST reports first_prim = -2 (the "first accepted triangle" is a property of the
 * reference's visit order).  The ordered traversal culls by distance and therefore relies on a triangle's computed distance
 * lying where the triangle lies (within the 2^-13 window of its boxes' entry distances).  Needle-shaped and collapsed
 * triangles (|e1 x e2| < 2^-10 |e1| |e2|, large enough to pass the reference's |det| >= 1e-5 test) break that -- the
 * reference "hits" them at distances unrelated to their position -- so a scene that holds any is traversed in reference
 * order whatever mode is asked for (first_prim is then the reference's).  The same noise, one ray at a time, is what FAST
 * does not reproduce: a ray within about 1e-4 rad of the plane of a triangle several units long, with |det| just above the
 * reference's threshold, for which the reference computes a distance far from where the ray meets the triangle (measured:
 * under one in a million rays aimed that way on purpose, none in ordinary rays); EXACT follows the reference there too. */
enum { PT_TRAVERSE_EXACT = 0, PT_TRAVERSE_FAST = 1 };

/* Integrators (src/integrator.cpp): MIS path tracer :45-141, shading normal :20-31, base colour :33-43. */
enum { PT_INTEGRATOR_PATH = 0, PT_INTEGRATOR_NORMAL = 1, PT_INTEGRATOR_ALBEDO = 2 };

typedef struct pt_render_params {
    uint32_t spp_begin;        /* first Sobol' sample index (inclusive) */
    uint32_t spp_end;          /* last sample index (exclusive) */
    uint32_t max_bounce;       /* 32 in the reference (src/integrator.cpp:50); 0 selects 32.  Specular bounces are not counted
                                * (src/integrator.cpp:84-87): a chain of mirror reflections is followed however long it is
                                * (the loop gives up after 2^20 iterations of one pass, where the reference would not return) */
    int32_t  integrator;       /* PT_INTEGRATOR_* */
    int32_t  traversal;        /* PT_TRAVERSE_* */
    uint32_t samples_per_pass; /* samples of every pixel in flight at once; 0 = automatic (up to 537 M paths, 120 GB of queues) */
    uint32_t flags;            /* PT_FLAG_* */
    /* Image split across processes: with block_count > 1 the call renders only the 32x32 pixel blocks (the unit of the
     * reference's BlockGenerator, include/pt/block.h:17, src/block.cpp:116-158) whose column + row index is congruent
     * to block_rank modulo block_count; their splats still reach the 2-pixel rim of neighbouring blocks, so the films
     * of all ranks add up to the whole frame exactly as the reference's block images do (src/block.cpp:102-110).
     * 0 or 1 = the whole image. */
    uint16_t block_rank;
    uint16_t block_count;
} pt_render_params;

#define PT_FLAG_NO_KERNEL_TIMING 1u /* do not bracket kernels with HIP events (pt_stats.kernel_ms stays 0) */
#define PT_FLAG_COUNT_WORK 2u       /* run the instrumented kernels: pt_stats.nodes_fetched / tris_fetched are filled */

#define PT_MAX_KERNEL_STATS 8
typedef struct pt_stats {
    uint64_t paths;           /* camera paths started */
    uint64_t closest_rays;    /* closest-hit queries (Scene::rayIntersect(ray, its)) */
    uint64_t any_rays;        /* any-hit queries (Scene::unocculded) */
    uint64_t exact_retraces;  /* FAST-mode rays re-traced in EXACT mode */
    /* PT_FLAG_COUNT_WORK only: BVH node records and triangle records read by the traversals launched from the
     * generate, extend and shadow kernels (index 0, 1, 2) */
    uint64_t nodes_fetched[3];
    uint64_t tris_fetched[3];
    double   seconds;         /* wall time of the call, device work included */
    /* per-kernel device time (HIP events on the launch stream) and launch counts */
    double   kernel_ms[PT_MAX_KERNEL_STATS];
    uint64_t kernel_launches[PT_MAX_KERNEL_STATS];
    /* PT_FLAG_COUNT_WORK only: those of nodes_fetched that the ordered traversal read from its copy of the top of the
     * navigation tree in LDS instead of from memory */
    uint64_t nodes_from_lds[3];
    uint32_t samples_per_pass; /* samples of every pixel kept in flight per pass (the automatic choice when the caller gave 0) */
    uint32_t passes;           /* passes the call took */
    uint64_t shaded_hits;      /* hit-list entries handed to the shade step (closest-hit queries that found something); the
                                * further iterations that a tail launch runs on its own are not listed */
} pt_stats;
/* indices into kernel_ms / kernel_launches; PT_K_FINISH = the launches that run few paths for many iterations in one go
 * (the thinned-out tail of a pass; the last chains of specular bounces beyond max_bounce iterations);
 * PT_K_COMPACT = the gathering of the surviving paths into a few dense shards */
enum { PT_K_GENERATE = 0, PT_K_EXTEND = 1, PT_K_SHADE = 2, PT_K_SHADOW = 3, PT_K_FILM = 4, PT_K_RETRACE = 5, PT_K_FINISH = 6, PT_K_COMPACT = 7 };

typedef struct pt_hip_scene pt_hip_scene;

/* ---- life cycle --------------------------------------------------------------------------------------------------- */

/* Selects the default GPU of this process (the one pt_hip_scene_create and the entry points without a handle use) and
 * creates its stream.  No counterpart in the reference (CPU only). */
int pt_hip_init(int device_ordinal);
/* Number of GPUs visible to this process (>= 0), or a negative PT_ERR_*.  May be called before pt_hip_init. */
int pt_hip_device_count(void);
int pt_hip_shutdown(void);
const char* pt_hip_last_error(void);
int pt_hip_abi_version(void);

/* Replaces Scene::preprocess's accelerator hand-off, `m_accel = new BVHTree(&m_shapes); m_accel->build()`
 * (src/scene.cpp:231-232): validates the description, flattens the host-built BVH into the device layouts and
 * uploads mesh, materials, lights, camera and the Sobol' tables. */
int pt_hip_scene_create(const pt_scene_desc* desc, pt_hip_scene** out);
/* The same on an explicit GPU (0 <= device_ordinal < pt_hip_device_count()): the handle, its device copies and its stream
 * live there, whichever GPU pt_hip_init selected. */
int pt_hip_scene_create_on(int device_ordinal, const pt_scene_desc* desc, pt_hip_scene** out);
/* Replaces `delete m_accel` in Scene::~Scene (include/pt/scene.h:18-26): frees device copies only. */
int pt_hip_scene_destroy(pt_hip_scene* scene);

/* BVHTree::build + BVHTreeBuilder::build (src/bvh.cpp:13-24, 99-234; include/pt/bvh.h:63-116) on the GPU: the same tree,
 * node for node, as the reference's sequential builder (and as pt_host_build_bvh) -- all ranges of a tree level are split
 * at once with segmented scans, the node numbers of the reference's stack order are replayed afterwards; only the three
 * initial std::sort calls stay on the host (their order among equal keys is a property of that algorithm).
 * verts: 3 floats per vertex; vi: 3 one-based vertex indices per triangle.  On success *nodes (n_nodes entries) and
 * *prim_ids (n_tris entries) are malloc'd: release with pt_hip_free.  max_depth (nullable): height of the tree, root = 1.
 * Needs pt_hip_init. */
int pt_hip_build_bvh(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes, uint32_t* n_nodes,
                     uint32_t** prim_ids, uint32_t* max_depth);
void pt_hip_free(void* p);

/* ---- batched ray queries (the Accel interface, include/pt/accel.h:10-24) ------------------------------------------ */

/* Accel::rayIntersect(const Ray&, Intersection&) -> BVHTree::rayIntersect (src/bvh.cpp:26-59), n rays at once.
 * Host pointers; copies in, traces, copies out. */
int pt_hip_intersect(pt_hip_scene* scene, const pt_ray* rays, size_t n, pt_hit* hits, int mode);
/* Accel::rayIntersect(const Ray&) -> BVHTree::rayIntersect (src/bvh.cpp:61-86).  occluded[i] = 1 when any triangle is
 * hit.  first_prim (nullable) receives the first triangle found in the reference's order, -1 if none; it is only
 * defined in EXACT mode (FAST mode writes -2 for occluded rays). */
int pt_hip_occluded(pt_hip_scene* scene, const pt_ray* rays, size_t n, uint8_t* occluded, int32_t* first_prim, int mode);
/* Same two queries on device-resident buffers (no copies; used for throughput measurement).  Synchronous. */
int pt_hip_intersect_device(pt_hip_scene* scene, const pt_ray* d_rays, size_t n, pt_hit* d_hits, int mode, float* kernel_ms);
int pt_hip_occluded_device(pt_hip_scene* scene, const pt_ray* d_rays, size_t n, uint8_t* d_occluded, int mode, float* kernel_ms);

/* No counterpart in the reference.  pt_hip_intersect_device for callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc): same kernels and results, but the per-shard fetch counters of the ordered
 * traversal live in a fixed array inside the library image (one copy per GPU and process, held until the process ends)
 * instead of a buffer the handle grows on demand, no events are created (hence no kernel_ms), and the call makes no
 * hipMalloc / hipFree.  At most PT_NOALLOC_MAX_RAYS rays per call (PT_ERR_LIMIT beyond that).  All calls on one GPU share
 * the array; the library serialises them (threads that call it for handles on the same GPU wait for each other).
 * Synchronous: the work runs on the handle's own stream and the call returns when it is done, so it is NOT for code under
 * HIP graph capture (a capturing stream cannot be synchronised, and this work is not on it). */
#define PT_NOALLOC_MAX_RAYS 67108864u /* 16384 shards of 4096 rays: 64 KB of counters */
int pt_hip_intersect_device_noalloc(pt_hip_scene* scene, const pt_ray* d_rays, size_t n, pt_hit* d_hits, int mode);

/* ---- rendering (the Integrator interface + frame loop) ------------------------------------------------------------- */

/* Replaces render(scene, sampler, integrator, result) (src/main.cpp:76-98) for samples [spp_begin, spp_end) of every
 * pixel with a SobolSampler (src/sampler.cpp:89-123).  film_rgba is (height+4) x (width+4) x 4 floats, row-major, the
 * layout of the reference's ImageBlock with its 2-pixel border (include/pt/block.h:35, src/block.cpp:14-35); it is
 * overwritten (not accumulated into).  Host pointer. */
int pt_hip_render(pt_hip_scene* scene, const pt_render_params* params, float* film_rgba, pt_stats* stats);
/* Same with the film in device memory (for a device-side reduce across processes). */
int pt_hip_render_device(pt_hip_scene* scene, const pt_render_params* params, float* d_film_rgba, pt_stats* stats);

/* render(...) (src/main.cpp:76-98) of one frame on n handles of the same scene, one per GPU, from one process: the sample
 * range [spp_begin, spp_end) is cut into n contiguous shares (or, with block_count == n, handle k renders the 32x32
 * blocks of rank k), one host thread drives each handle, and the films are summed on the first handle's GPU (peer copies
 * over xGMI + one add per film: the frame's only reduction, src/block.cpp:102-110) and copied to film_rgba (host).
 * stats: counts and per-kernel times summed over the GPUs, seconds = wall time of the call. */
int pt_hip_render_multi(pt_hip_scene* const* scenes, int n, const pt_render_params* params, float* film_rgba, pt_stats* stats);

/* Integrator::Li (include/pt/integrator.h:9-19) for an explicit list of (pixel, sample index) pairs: writes the
 * radiance of each sample (3 floats) and its film position (2 floats, nullable) without touching a film.
 * pixel_xy holds n (x, y) pairs. */
int pt_hip_li(pt_hip_scene* scene, const pt_render_params* params, const int32_t* pixel_xy, const uint32_t* sample_index,
              size_t n, float* rgb, float* pixel_sample, pt_stats* stats);

/* SobolSampler::sampleDimension after startPixelSample (src/sampler.cpp:96-100, include/pt/sampler.h:122):
 * value of dimension dim[i] for pixel (x,y)[i] and sample index sample_index[i]. */
int pt_hip_sobol(pt_hip_scene* scene, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim, size_t n,
                 float* out);

/* No counterpart in the reference.  The box test reproduces AABB::intersect's six divisions (src/aabb.cpp:7-27) with a
 * shorter correctly rounded sequence; this runs `pairs` pseudo-random and adversarial operand pairs through both and
 * reports how many quotients differ (must be 0). */
int pt_hip_selftest_division(uint64_t pairs, uint64_t seed, uint64_t* mismatches);

/* No counterpart in the reference (test hooks).  The two bare functions of the traversal for explicit operand pairs, exactly
 * as the kernels evaluate them: AABB::intersect (src/aabb.cpp:6-35; boxes6 = lo.xyz, hi.xyz per pair; accepted[i] = 1 when
 * the reference would accept box i for ray i with the ray's own near and far end) and Triangle::intersect
 * (src/shape.cpp:50-89; tri9 = the three corners per pair; tuv = t, u, v of accepted hits, zeros otherwise).  Host
 * pointers.  Need pt_hip_init. */
int pt_hip_probe_box(const float* boxes6, const pt_ray* rays, size_t n, uint8_t* accepted);
int pt_hip_probe_triangle(const float* tri9, const pt_ray* rays, size_t n, uint8_t* hit, float* tuv);

/* No counterpart in the reference (test hook).  The shading code's sine, cosine and power for explicit operands: the
 * reference calls the C library's sinf / cosf (include/pt/common.h:103, 111) and powf (include/pt/common.h:108, src/material.cpp:34, 79, 108), the
 * kernels evaluate them in double and round once.  sin_cos_pow = sin(x[i]), cos(x[i]), x[i]^y[i] per element.  Host
 * pointers.  Needs pt_hip_init. */
int pt_hip_probe_math(const float* x, const float* y, size_t n, float* sin_cos_pow);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
