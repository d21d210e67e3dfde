// libpt_hip.so: C ABI (include/pt_hip.h) over the HIP kernels in kernels.h.
//
// Host-side responsibilities: validate the flat scene description, flatten the binary BVH (handed in, or built here by
// pt_hip_build_bvh) into the two device layouts (reference-order nodes and the 4-wide navigation nodes of the ordered
// traversal), build the Sobol' look-up tables, own the per-pass work buffers, and drive the wavefront loop
// (generate -> [shade -> shadow + extend]* -> film).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "host_common.h"
#include "kernels.h"
#include "pt_hip.h"
#include "sobol_tables.inc"

using namespace ptd;

namespace {

thread_local std::string g_error;
// The device and stream the calling thread works on.  A scene handle is bound to the device it was created on and owns a
// stream there; every entry point that takes a handle makes that pair the thread's current one (bind), so several
// handles -- one per GPU -- can be driven from several host threads of one process.  Entry points without a handle use
// the pair pt_hip_init set up.
int g_default_device = -1;
hipStream_t g_default_stream = nullptr;
thread_local int g_device = -1;
thread_local hipStream_t g_stream = nullptr;

constexpr uint32_t kMaxGridY = 65535;
constexpr uint32_t kMaxGridZ = 8;  // shards of a pass: up to kMaxGridZ * kMaxGridY (2^31 slots would be 524288 shards)
// launch geometry (workgroups per shard, shards); n_shards is a multiple of its grid.z by construction (ws_reserve)
inline dim3 shard_grid(uint32_t per_shard, uint32_t n_shards) {
    const uint32_t gz = (n_shards + kMaxGridY - 1) / kMaxGridY;
    return dim3(per_shard, n_shards / gz, gz);
}

// Fetch counters of pt_hip_intersect_device_noalloc: one word per shard of kShard rays, in the library's own image so
// that the entry point needs no allocation.  Every GPU holds its own copy at its own address; host code must use the
// per-device address that pt_hip_init obtains with hipGetSymbolAddress (g_noalloc_fetch), never this name.
constexpr uint32_t kNoAllocShards = PT_NOALLOC_MAX_RAYS / kShard;
static_assert(kNoAllocShards * kShard == PT_NOALLOC_MAX_RAYS && kNoAllocShards <= kMaxGridY, "whole shards, one launch");
static __device__ __align__(256) uint32_t scratch[kNoAllocShards];
constexpr int kMaxDevices = 64;
std::atomic<uint32_t*> g_noalloc_fetch[kMaxDevices] = {};
std::mutex g_noalloc_setup;              // guards the first hipGetSymbolAddress of a device
std::mutex g_noalloc_call[kMaxDevices];  // one call at a time per GPU: every call on a GPU uses that GPU's one array

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                           std::to_string(__LINE__) + ")");                                    \
    } while (0)

// ---- Sobol' helpers (host) ----------------------------------------------------------------------------------------

inline uint32_t sobol_word(uint64_t index, int dim) {  // the XOR part of sobolSample (src/sampler.cpp:6-14)
    uint32_t v = 0;
    for (int i = dim * 52; index != 0; index >>= 1, ++i)
        if (index & 1) v ^= kSobolMatrices32[i];
    return v;
}
// The same for all 1024 dimensions of one index: row `bit` of the transposed table holds that bit's generator word of
// every dimension, so the whole row of a sample number is a handful of 1024-word XORs.
inline void sobol_words_all_dims(uint64_t index, uint32_t* out) {
    static const std::vector<uint32_t> by_bit = [] {
        std::vector<uint32_t> t(size_t(52) * 1024);
        for (int dim = 0; dim < 1024; ++dim)
            for (int bit = 0; bit < 52; ++bit) t[size_t(bit) * 1024 + dim] = kSobolMatrices32[dim * 52 + bit];
        return t;
    }();
    std::memset(out, 0, 1024 * sizeof(uint32_t));
    for (int bit = 0; index != 0; index >>= 1, ++bit) {
        if (!(index & 1)) continue;
        const uint32_t* row = by_bit.data() + size_t(bit) * 1024;
        for (int dim = 0; dim < 1024; ++dim) out[dim] ^= row[dim];
    }
}
inline int32_t round_up_pow2(int32_t v) {  // src/sampler.cpp:68-76
    v--;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v + 1;
}
// index bits contributed by the sample number alone (sobolIntervalToIndex with pixel (0,0), src/sampler.cpp:16-36)
inline uint64_t frame_index(uint32_t m, uint64_t frame) {
    if (m == 0) return frame;
    uint64_t index = frame << (2 * m);
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= kVdCSobolMatrices[m - 1][c];
    for (int c = 0; delta; delta >>= 1, ++c)
        if (delta & 1) index ^= kVdCSobolMatricesInv[m - 1][c];
    return index;
}

constexpr uint64_t kAutoSlots = uint64_t(2) * 65535 * 4096;  // default number of path slots in flight per pass (120 GB of queues)
constexpr uint64_t kBytesPerSlot = 224;             // two path queues, hit list, shadow queue, deferred lists, radiance slot
constexpr int kRing = 4;  // counter snapshots in flight between the stream and the host

// Once every path has had max_bounce iterations (only chains of specular bounces are left) and at most this many paths
// are alive, one launch of k_finish runs them to their end, a wave per path: as many as the chip holds at once (256 CUs
// x 8 waves at that kernel's register count).  More paths than that are better off in the regular loop, whose cost per
// iteration does not depend on their number.
constexpr uint32_t kFinishPaths = 2048;
// The tail of a pass (ordinary paths, at most max_bounce iterations each) goes to k_finish once at most this many paths
// are alive, but not before the first kTailStart bounces have gone through the regular kernels.
// Measured on 128- and 512-spp passes of the 1.6 M triangle interior / veach-mis: 8192 paths +1.3 / +3.2 %, 16384
// +1.6 / +3.3 %, 32768 with four paths per wave +1.9 / +3.8 %, 131072 (four or eight per wave) +1.2 / +3.7 %, 524288
// -4 %: k_finish lasts as long as its longest path, and beyond what the chip holds at once it queues.
#ifndef PT_TAIL_PATHS
#define PT_TAIL_PATHS 32768
#endif
constexpr uint32_t kTailPaths = PT_TAIL_PATHS;
constexpr uint32_t kTailStart = 6;
// The paths of a pass are gathered into dense shards whenever they fill at most 1 / kCompactShare of the shards in play
// (measured on 512-spp passes of the 1.6 M triangle interior, against never: once at 1/32 +4.0 %, once at 1/8 +5.4 %,
// every time at 1/8 +5.8 %, every time at 1/16 +5.1 %).
#ifndef PT_COMPACT_SHARE
#define PT_COMPACT_SHARE 8
#endif
constexpr uint64_t kCompactShare = PT_COMPACT_SHARE;
constexpr uint32_t kTailLanes = 4;

struct Workspace {
    uint32_t cap_slots = 0;   // padded to whole shards
    uint32_t n_shards = 0;
    PathQ q[2]{};
    HitQ hq{};                // count points at hit_count[bounce & 1]
    ShadowQ sh{};
    DeferQ dqe{}, dqs{};      // rays handed back by the wave-refill extend / shadow kernels
    float4* lbuf = nullptr;
    // device counters, n_shards words each:
    //   [q0 | q1 | hit0 | hit1 | shadow | fetch extend | fetch shadow | deferred extend | deferred shadow | done extend |
    //    done shadow], then kMiscWords
    uint32_t* counters = nullptr;
    uint32_t* hit_count[2] = {nullptr, nullptr};
    uint32_t* fetch_e = nullptr;
    uint32_t* fetch_s = nullptr;
    uint32_t* done_e = nullptr;
    uint32_t* done_s = nullptr;
    uint32_t* misc = nullptr;
    uint32_t* h_counters = nullptr;  // pinned host mirror: kRing snapshots of the first 5 arrays, then kMiscWords
    hipEvent_t ring_ev[kRing] = {};
    uint32_t* frame_tab = nullptr;
    size_t frame_cap = 0;  // in frames
    uint2* finish_list = nullptr;  // (shard, hit-list entry) pairs handed to k_finish (max(kFinishPaths, kTailPaths) of them)
    std::vector<void*> owned;
    size_t misc_offset() const { return (size_t(11) * n_shards + 1) & ~size_t(1); }  // 64-bit atomics live in misc: keep it 8-byte aligned
    size_t counter_words() const { return misc_offset() + kMiscWords; }
    size_t snapshot_words() const { return size_t(5) * n_shards; }
    uint32_t* h_misc() const { return h_counters + size_t(kRing) * snapshot_words(); }
};

struct EventPool {
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;  // kernel id of pair i (events 2i, 2i+1)
    size_t used = 0;
};

}  // namespace

struct pt_hip_scene {
    DevScene dev{};
    std::vector<void*> owned;
    uint32_t sob_m = 0;
    uint32_t depth = 0;
    uint32_t width = 0, height = 0;
    // The mesh holds needle-shaped or collapsed triangles for which Triangle::intersect returns distances that have little
    // to do with where the triangle lies (see unreliable_triangle): the ordered traversal's distance culling cannot follow
    // the reference there, so every query on this scene runs in reference order, whatever mode the caller asks for.
    bool reference_order_only = false;
    Workspace ws;
    EventPool events;
    int device = -1;                  // the GPU this handle lives on
    hipStream_t stream = nullptr;     // its own stream there
    uint32_t* batch_fetch = nullptr;  // per-shard fetch counters of the batched ordered queries
    size_t batch_fetch_cap = 0;
    // grow-only device staging of the host-pointer entry points (rays, hits, flags, sample lists, film), so that a
    // caller that renders or queries batch after batch does not pay an allocation per call
    enum { kStageSlots = 6 };
    void* stage[kStageSlots] = {};
    size_t stage_cap[kStageSlots] = {};
    hipEvent_t query_ev[2] = {};  // start / stop of a batched query
    // image split (pt_render_params::block_count > 1): tile tables of the most recent (rank, count)
    uint32_t* tile_list = nullptr;
    int32_t* tile_pos = nullptr;
    uint32_t split_key = 0, split_tiles = 0;
    uint64_t split_pixels = 0;
    // handles that render at the same time on this handle's GPU (set by pt_hip_render_multi for the length of its call):
    // the automatic pass size gives each an equal share of the free device memory
    uint32_t sharing = 1;
};

namespace {

template <typename T>
int upload(pt_hip_scene* s, const T* src, size_t count, const T** dst) {
    void* d = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&d, bytes));
    s->owned.push_back(d);
    if (count) HIP_TRY(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(d);
    return PT_OK;
}

int validate(const pt_scene_desc* d, uint32_t& depth_out) {
    if (!d) return fail(PT_ERR_INVALID, "scene description is null");
    if (d->n_tris == 0 || d->n_nodes == 0) return fail(PT_ERR_INVALID, "scene has no triangles or no BVH nodes");
    if (!d->verts || !d->vi || !d->mat_id || !d->light_id || !d->mats || !d->nodes || !d->prim_ids)
        return fail(PT_ERR_INVALID, "scene description has a null required array");
    if ((d->n_lights && !d->lights) || (d->n_textures && !d->textures) || (d->n_normals && !d->normals) || (d->n_uvs && !d->uvs))
        return fail(PT_ERR_INVALID, "scene description has a null array with a non-zero count (lights, textures, normals or uvs)");
    if (d->n_tris > kLeafFirstMask) return fail(PT_ERR_LIMIT, "more than 2^27-1 triangles");
    if (d->camera.width == 0 || d->camera.height == 0 || d->camera.width > 32768 || d->camera.height > 32768)
        return fail(PT_ERR_INVALID, "camera resolution out of range (1..32768)");
    for (size_t i = 0; i < 3 * size_t(d->n_tris); ++i) {
        if (d->vi[i] == 0 || d->vi[i] > d->n_verts) return fail(PT_ERR_INVALID, "vertex index out of range");
        if (d->n_normals && d->ni && d->ni[i] > d->n_normals) return fail(PT_ERR_INVALID, "normal index out of range");
        if (d->n_uvs && d->ti && d->ti[i] > d->n_uvs) return fail(PT_ERR_INVALID, "uv index out of range");
    }
    if ((d->n_normals && !d->ni) || (d->n_uvs && !d->ti)) return fail(PT_ERR_INVALID, "normals/uvs given without index arrays");
    for (uint32_t t = 0; t < d->n_tris; ++t) {
        if (d->mat_id[t] >= d->n_mats) return fail(PT_ERR_INVALID, "material id out of range");
        if (d->light_id[t] >= int32_t(d->n_lights)) return fail(PT_ERR_INVALID, "light id out of range");
        if (d->prim_ids[t] >= d->n_tris) return fail(PT_ERR_INVALID, "prim_ids entry out of range");
    }
    for (uint32_t i = 0; i < d->n_mats; ++i)
        if (d->mats[i].tex >= int32_t(d->n_textures)) return fail(PT_ERR_INVALID, "texture id out of range");
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        if (d->lights[i].tri >= d->n_tris) return fail(PT_ERR_INVALID, "light triangle out of range");
        // zero, negative or NaN: 1/area would put inf or NaN into every radiance value.  +inf is what Triangle::surfaceArea
        // (src/shape.cpp:32-36) itself returns for astronomically large triangles (its norm overflows); 1/area is then 0.
        if (!(d->lights[i].area > 0.0f)) return fail(PT_ERR_INVALID, "light with a zero, negative or NaN area");
    }
    // walk the tree: every node reachable once, ranges in bounds, depth within the traversal stack
    std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 1u}};
    std::vector<uint8_t> seen(d->n_nodes, 0);
    uint32_t depth = 0;
    size_t leaf_prims = 0;
    while (!stack.empty()) {
        auto [ni, dep] = stack.back();
        stack.pop_back();
        if (ni >= d->n_nodes || seen[ni]) return fail(PT_ERR_INVALID, "BVH child index out of range or node shared");
        seen[ni] = 1;
        depth = std::max(depth, dep);
        const pt_bvh_node& n = d->nodes[ni];
        if (n.count) {
            if (n.count > 8 || size_t(n.first) + n.count > d->n_tris) return fail(PT_ERR_INVALID, "BVH leaf range out of bounds");
            leaf_prims += n.count;
        } else {
            stack.push_back({n.first, dep + 1});
            stack.push_back({n.first + 1, dep + 1});
        }
    }
    if (leaf_prims != d->n_tris) return fail(PT_ERR_INVALID, "BVH leaves do not cover every triangle exactly once");
    // reference order: one deferred sibling per level; ordered mode: up to three deferred slots per two levels
    // (an 8-wide navigation node: up to seven deferred slots per three levels)
    const uint32_t ordered_need = kWide == 8 ? 7 * ((depth + 2) / 3) + 8 : 3 * ((depth + 1) / 2) + 4;
    if (depth + 2 > uint32_t(kMaxStack) || ordered_need > uint32_t(kMaxStack))
        return fail(PT_ERR_LIMIT, "BVH deeper than the traversal stack (" + std::to_string(kWide == 8 ? 3 * ((kMaxStack - 8) / 7) : 2 * ((kMaxStack - 4) / 3)) + " levels)");
    depth_out = depth;
    return PT_OK;
}

// f(begin, end) over [0, n) on up to 16 threads
template <typename F>
void parallel_for(size_t n, F f) {
    const size_t hw = std::max<size_t>(1, std::min<size_t>(16, std::thread::hardware_concurrency()));
    const size_t parts = std::min(hw, (n + 4095) / 4096);
    if (parts <= 1) {
        f(size_t(0), n);
        return;
    }
    std::vector<std::thread> th;
    for (size_t k = 1; k < parts; ++k) th.emplace_back([=] { f(n * k / parts, n * (k + 1) / parts); });
    f(size_t(0), n / parts);
    for (std::thread& t : th) t.join();
}

// Builds a device array out of `n_units` units of variable size: unit u occupies items [offset(u), offset(u + 1)) and is
// written by fill(u, out).  The array is produced in chunks of a few million items: worker threads fill one pinned staging
// buffer while the previous one is on its way to the device, so neither the fill nor the copy of a multi-gigabyte array
// runs on one thread or through pageable memory.
template <typename T, typename Offset, typename Fill>
int fill_upload(pt_hip_scene* s, size_t n_units, Offset offset, const T** dst, Fill fill) {
    const size_t total = offset(n_units);
    void* dv = nullptr;
    HIP_TRY(hipMalloc(&dv, std::max<size_t>(total, 1) * sizeof(T)));
    s->owned.push_back(dv);
    *dst = static_cast<const T*>(dv);
    if (total == 0) return PT_OK;
    const size_t chunk_items = std::max<size_t>(1, (size_t(32) << 20) / sizeof(T));
    size_t largest = 0;  // the staging buffers hold a whole number of units
    {
        size_t u = 0;
        while (u < n_units) {
            size_t v = u + 1;
            while (v < n_units && offset(v + 1) - offset(u) <= chunk_items) ++v;
            largest = std::max(largest, offset(v) - offset(u));
            u = v;
        }
    }
    T* stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    auto cleanup = [&] {
        for (int k = 0; k < 2; ++k) {
            if (stage[k]) (void)hipHostFree(stage[k]);
            if (done[k]) (void)hipEventDestroy(done[k]);
        }
    };
    for (int k = 0; k < 2; ++k) {
        if (hipHostMalloc(reinterpret_cast<void**>(&stage[k]), largest * sizeof(T), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return fail(PT_ERR_DEVICE, "no pinned staging memory for the scene upload");
        }
    }
    int rc = PT_OK;
    size_t u = 0;
    for (int k = 0; u < n_units && rc == PT_OK; k ^= 1) {
        size_t v = u + 1;
        while (v < n_units && offset(v + 1) - offset(u) <= chunk_items) ++v;
        const size_t first = offset(u), items = offset(v) - first;
        if (hipEventSynchronize(done[k]) != hipSuccess) rc = fail(PT_ERR_DEVICE, "scene upload failed");  // the copy that last used this buffer
        T* out = stage[k];
        parallel_for(v - u, [&](size_t b, size_t e) {
            for (size_t x = u + b; x < u + e; ++x) fill(x, out + (offset(x) - first));
        });
        if (rc == PT_OK && (hipMemcpyAsync(static_cast<T*>(dv) + first, out, items * sizeof(T), hipMemcpyHostToDevice, g_stream) != hipSuccess ||
                            hipEventRecord(done[k], g_stream) != hipSuccess))
            rc = fail(PT_ERR_DEVICE, "scene upload failed");
        u = v;
    }
    if (hipStreamSynchronize(g_stream) != hipSuccess && rc == PT_OK) rc = fail(PT_ERR_DEVICE, "scene upload failed");
    cleanup();
    return rc;
}

// phase times of pt_hip_scene_create on stderr when PT_HIP_SCENE_TIMING is set
struct PhaseClock {
    bool on = std::getenv("PT_HIP_SCENE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pt_hip_scene_create: %-28s %.2f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

// Moeller-Trumbore (src/shape.cpp:50-89) divides by det = e1 . (d x e2), of magnitude |e1 x e2| cos(angle to the normal).
// For a needle or collapsed triangle, |e1 x e2| below 2^-10 |e1| |e2|, the rounding noise of that expression (about
// 2^-20 |e1| |e2|) is a sizeable part of it or all of it: the distance it returns is off by far more than the 2^-13 window
// the ordered traversal works with, up to "hits" at distance zero for rays that pass the triangle's box anywhere.  Such a
// triangle matters only if det can pass the reference's |det| >= 1e-5 test at all.
bool unreliable_triangle(const float* e1, const float* e2) {
    const double cx = double(e1[1]) * e2[2] - double(e1[2]) * e2[1], cy = double(e1[2]) * e2[0] - double(e1[0]) * e2[2],
                 cz = double(e1[0]) * e2[1] - double(e1[1]) * e2[0];
    const double area2 = std::sqrt(cx * cx + cy * cy + cz * cz);
    const double scale = std::sqrt((double(e1[0]) * e1[0] + double(e1[1]) * e1[1] + double(e1[2]) * e1[2]) *
                                   (double(e2[0]) * e2[0] + double(e2[1]) * e2[1] + double(e2[2]) * e2[2]));
    return area2 < 0x1p-10 * scale && area2 + 0x1p-19 * scale >= 1e-5;
}

// mesh has normals and this face's first normal index is non-zero (include/pt/mesh.h:37-39)
inline bool tri_has_normals(const pt_scene_desc* d, uint32_t prim) { return d->n_normals != 0 && d->ni && d->ni[3 * size_t(prim)] != 0; }

int build_scene(pt_hip_scene* s, const pt_scene_desc* d) {
    DevScene& dev = s->dev;
    PhaseClock clk;
    // Leaf array: every leaf is a header record (its parent's box, see leaf_parent_accepts) followed by its triangles
    // {p0, prim} {e1} {e2}.  An interior node whose two children are both leaves with at most kFatLeafTris triangles
    // between them is a FAT LEAF of the ordered traversal: its children lie next to each other in node order and have the
    // same parent box, so the second one goes without a header and the pair reads as one leaf (traverse.h, fat leaves).
    // slot_of[i] = first record of leaf node i (its header, or its first triangle when it has none); one entry more than
    // there are nodes: the records of nodes [0, i) -- interior nodes take none.
    std::vector<uint32_t> slot_of(size_t(d->n_nodes) + 1, 0), parent(d->n_nodes, 0);
    std::vector<uint8_t> fat(d->n_nodes, 0);
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const pt_bvh_node& b = d->nodes[i];
        if (b.count) continue;
        parent[b.first] = parent[b.first + 1] = i;
        const uint32_t ca = d->nodes[b.first].count, cb = d->nodes[b.first + 1].count;
        fat[i] = (ca != 0u && cb != 0u && ca + cb <= kFatLeafTris) ? 1 : 0;
    }
    // the second child of a fat leaf: no header of its own
    auto headless = [&](uint32_t i) { return i != 0u && fat[parent[i]] && i == d->nodes[parent[i]].first + 1u; };

    // ---- navigation tree of the ordered traversal: which binary nodes become navigation nodes, and their numbers.
    // The navigation node of binary interior node N holds up to kWide slots: N's two children, some of them opened into
    // their own children in turn; slots stay in the binary tree's left-to-right order, and the interior slots are
    // navigation nodes themselves.
    auto leaf_like = [&](uint32_t c) { return d->nodes[c].count != 0u || fat[c]; };
    auto surface = [&](uint32_t c) {
        const pt_bvh_node& b = d->nodes[c];
        const double x = double(b.hi[0]) - double(b.lo[0]), y = double(b.hi[1]) - double(b.lo[1]), z = double(b.hi[2]) - double(b.lo[2]);
        return x * y + y * z + z * x;
    };
    // Which slots a navigation node gets.  Default: the choice that minimises the expected number of navigation nodes a
    // random ray visits, sum of surface(N) over the navigation nodes N (a ray that meets the scene meets a box with a
    // probability proportional to its surface).  F(n, i) = that sum for the subtree of binary node n when n may take up
    // to i slots of the node above it: 0 for a leaf or fat leaf; otherwise either n is one slot and a navigation node
    // itself, surface(n) + G(n, kWide), or it is opened and its children share the i slots, G(n, i) = the best split
    // F(left, a) + F(right, i - a).  Filled bottom-up; `share[n][i]` keeps the left child's part (0: n stays one slot).
    // PT_COLLAPSE=greedy in the environment selects the earlier rule instead: open the interior slot with the largest
    // surface until the node is full.
    const char* collapse_env = std::getenv("PT_COLLAPSE");
    const bool greedy = collapse_env && std::strcmp(collapse_env, "greedy") == 0;
    bool forward = true;  // every child index exceeds its parent's, as in every tree a top-down builder emits
    for (uint32_t i = 0; i < d->n_nodes && forward; ++i)
        if (d->nodes[i].count == 0u && d->nodes[i].first <= i) forward = false;
    std::vector<float> cost;     // F(n, 1..kWide)
    std::vector<uint8_t> share;  // share[n][i], i = 2..kWide, and share[n][1] = the left child's part when n is a navigation node
    if (!greedy) {
        cost.assign(size_t(d->n_nodes) * kWide, 0.0f);
        share.assign(size_t(d->n_nodes) * kWide, 0);
        std::vector<uint32_t> bottom_up;
        if (!forward) {  // children before parents
            std::vector<uint32_t> stack{0u};
            while (!stack.empty()) {
                const uint32_t n = stack.back();
                stack.pop_back();
                bottom_up.push_back(n);
                if (d->nodes[n].count == 0u) {
                    stack.push_back(d->nodes[n].first);
                    stack.push_back(d->nodes[n].first + 1u);
                }
            }
            std::reverse(bottom_up.begin(), bottom_up.end());
        }
        for (uint32_t k = 0; k < d->n_nodes; ++k) {
            const uint32_t n = forward ? d->n_nodes - 1u - k : bottom_up[k];
            if (leaf_like(n)) continue;  // all zero
            const float* fl = &cost[size_t(d->nodes[n].first) * kWide];
            const float* fr = fl + kWide;
            float g[kWide + 1];
            uint8_t ga[kWide + 1];
            for (int j = 2; j <= kWide; ++j) {
                g[j] = std::numeric_limits<float>::infinity();
                ga[j] = 1;
                for (int a = 1; a < j; ++a) {
                    const float c = fl[a - 1] + fr[j - a - 1];
                    if (c < g[j]) {
                        g[j] = c;
                        ga[j] = uint8_t(a);
                    }
                }
            }
            float* f = &cost[size_t(n) * kWide];
            uint8_t* sh = &share[size_t(n) * kWide];
            f[0] = float(surface(n)) + g[kWide];
            sh[0] = ga[kWide];
            for (int i = 2; i <= kWide; ++i) {
                const bool opened = g[i] < f[0];
                f[i - 1] = opened ? g[i] : f[0];
                sh[i - 1] = opened ? ga[i] : 0;
            }
        }
        clk.lap("collapse costs");
    }
    auto collapse = [&](uint32_t n, uint32_t* slots) -> int {
        int ns = 0;
        if (!greedy) {
            // (binary node, slots it may take), left to right
            std::pair<uint32_t, int> todo[2 * kWide];
            int nt = 0;
            const int left = share[size_t(n) * kWide];
            todo[nt++] = {d->nodes[n].first + 1u, kWide - left};
            todo[nt++] = {d->nodes[n].first, left};
            while (nt > 0) {
                const auto [m, j] = todo[--nt];
                const int a = (j >= 2 && !leaf_like(m)) ? share[size_t(m) * kWide + size_t(j - 1)] : 0;
                if (a == 0) {
                    slots[ns++] = m;
                } else {
                    todo[nt++] = {d->nodes[m].first + 1u, j - a};
                    todo[nt++] = {d->nodes[m].first, a};
                }
            }
            return ns;
        }
        slots[ns++] = d->nodes[n].first;
        slots[ns++] = d->nodes[n].first + 1u;
        while (ns < kWide) {
            int open = -1;
            double best = -1.0;
            for (int k = 0; k < ns; ++k) {
                if (leaf_like(slots[k])) continue;
                const double a = surface(slots[k]);
                if (a > best) {
                    best = a;
                    open = k;
                }
            }
            if (open < 0) break;
            const uint32_t c = slots[open];
            for (int k = ns; k > open + 1; --k) slots[k] = slots[k - 1];
            slots[open] = d->nodes[c].first;
            slots[open + 1] = d->nodes[c].first + 1u;
            ++ns;
        }
        return ns;
    };
    // triangles of a leaf-like slot (a leaf, or a fat leaf: both children together)
    auto unit_tris = [&](uint32_t c) -> uint32_t {
        const pt_bvh_node& b = d->nodes[c];
        return b.count ? b.count : d->nodes[b.first].count + d->nodes[b.first + 1].count;
    };
    // `order`: binary node of navigation node 0, 1, ...; `units`: the leaves and fat leaves in the order of their records in
    // the leaf array (a unit = one header record followed by its triangles)
    std::vector<uint32_t> wide_id(d->n_nodes, kEmptyRef), order, units;
    std::vector<uint32_t> child_base, leaf_base;  // 8-wide only, per navigation node: number of its first interior slot, record of its first leaf slot
    if (leaf_like(0u)) units.push_back(0u);
    if (kWide == 4) {
        // Numbered in the order of their binary node index (the reference's numbering keeps a subtree contiguous) when every
        // child index exceeds its parent's, as in every tree a top-down builder emits -- one streaming pass; otherwise
        // depth-first.  Leaf records in the order of the binary nodes.
        if (forward) {
            std::vector<uint8_t> is_nav(d->n_nodes, 0);
            if (!leaf_like(0u)) is_nav[0] = 1;
            for (uint32_t i = 0; i < d->n_nodes; ++i) {
                if (!is_nav[i]) continue;
                wide_id[i] = uint32_t(order.size());
                order.push_back(i);
                uint32_t slots[kWide];
                const int ns = collapse(i, slots);
                for (int k = 0; k < ns; ++k)
                    if (!leaf_like(slots[k])) is_nav[slots[k]] = 1;
            }
        } else {
            std::vector<uint32_t> stack;
            if (!leaf_like(0u)) stack.push_back(0u);
            while (!stack.empty()) {
                const uint32_t n = stack.back();
                stack.pop_back();
                wide_id[n] = uint32_t(order.size());
                order.push_back(n);
                uint32_t slots[kWide];
                const int ns = collapse(n, slots);
                for (int k = ns - 1; k >= 0; --k)
                    if (!leaf_like(slots[k])) stack.push_back(slots[k]);
            }
        }
        // The navigation nodes a ray is most likely to visit come first: starting at the root, the node with the largest
        // surface among the children of those already taken (a ray that meets a box meets a box inside it with a
        // probability of about the ratio of their surfaces), kTreeletMax of them.  The traversal kernels keep a prefix of
        // this list in LDS (treelet_fill in traverse.h); the other nodes keep their order behind it.
        if (!order.empty()) {
            std::priority_queue<std::pair<double, uint32_t>> frontier;
            frontier.push({surface(order[0]), order[0]});
            std::vector<uint32_t> top;
            std::vector<uint8_t> in_top(d->n_nodes, 0);
            while (!frontier.empty() && top.size() < kTreeletMax) {
                const uint32_t n = frontier.top().second;
                frontier.pop();
                top.push_back(n);
                in_top[n] = 1;
                uint32_t slots[kWide];
                const int ns = collapse(n, slots);
                for (int k = 0; k < ns; ++k)
                    if (!leaf_like(slots[k])) frontier.push({surface(slots[k]), slots[k]});
            }
            for (uint32_t n : order)
                if (!in_top[n]) top.push_back(n);
            order.swap(top);
            for (size_t i = 0; i < order.size(); ++i) wide_id[order[i]] = uint32_t(i);
        }
        for (uint32_t i = 1; i < d->n_nodes; ++i)
            if (leaf_like(i) && !(d->nodes[i].count && fat[parent[i]])) units.push_back(i);  // a fat leaf's children belong to its unit
    } else if (!leaf_like(0u)) {
        // 8-wide: references are implicit (traverse.h, Nav8), so the interior slots of a node are numbered together, and
        // its leaf slots follow one another in the leaf array, at the moment the node is EXPANDED.  Nodes are expanded by
        // surface, largest first, until kTreeletMax are numbered (the most visited ones come first, for the LDS treelet);
        // from there on depth-first from each of the nodes still waiting, so that a subtree's records stay together.
        order.push_back(0u);
        wide_id[0] = 0u;
        uint32_t next_record = 0;
        auto expand = [&](uint32_t n, std::vector<uint32_t>& fresh) {
            const uint32_t id = wide_id[n];
            if (child_base.size() <= id) {
                child_base.resize(size_t(id) + 1, 0u);
                leaf_base.resize(size_t(id) + 1, 0u);
            }
            child_base[id] = uint32_t(order.size());
            leaf_base[id] = next_record;
            uint32_t slots[kWide];
            const int ns = collapse(n, slots);
            for (int k = 0; k < ns; ++k) {
                const uint32_t c = slots[k];
                if (leaf_like(c)) {
                    units.push_back(c);
                    next_record += 1u + unit_tris(c);
                } else {
                    wide_id[c] = uint32_t(order.size());
                    order.push_back(c);
                    fresh.push_back(c);
                }
            }
        };
        std::priority_queue<std::pair<double, uint32_t>> frontier;
        frontier.push({surface(0u), 0u});
        std::vector<uint32_t> fresh;
        while (!frontier.empty() && order.size() < kTreeletMax) {
            const uint32_t n = frontier.top().second;
            frontier.pop();
            fresh.clear();
            expand(n, fresh);
            for (uint32_t c : fresh) frontier.push({surface(c), c});
        }
        std::vector<uint32_t> waiting;
        while (!frontier.empty()) {
            waiting.push_back(frontier.top().second);
            frontier.pop();
        }
        std::sort(waiting.begin(), waiting.end(), [&](uint32_t a, uint32_t b) { return wide_id[a] < wide_id[b]; });
        std::vector<uint32_t> stack;
        for (uint32_t root_of : waiting) {
            stack.assign(1, root_of);
            while (!stack.empty()) {
                const uint32_t n = stack.back();
                stack.pop_back();
                fresh.clear();
                expand(n, fresh);
                for (size_t k = fresh.size(); k-- > 0;) stack.push_back(fresh[k]);
            }
        }
        child_base.resize(order.size(), 0u);
        leaf_base.resize(order.size(), 0u);
    }
    if (order.size() >= (size_t(1) << 26)) return fail(PT_ERR_LIMIT, "more than 2^26 navigation nodes");  // byte offsets of nav_fetch are 32-bit
    // records of the leaf array: unit u starts at unit_record[u]; slot_of[i] = first record of leaf node i (its header, or
    // its first triangle when it is the second child of a fat leaf)
    std::vector<uint32_t> unit_record(units.size() + 1, 0);
    {
        uint64_t n_slots64 = 0;
        for (size_t u = 0; u < units.size(); ++u) {
            unit_record[u] = uint32_t(n_slots64);
            const uint32_t c = units[u];
            const pt_bvh_node& b = d->nodes[c];
            if (b.count) {
                slot_of[c] = uint32_t(n_slots64);
            } else {
                slot_of[b.first] = uint32_t(n_slots64);
                slot_of[b.first + 1] = uint32_t(n_slots64) + 1u + d->nodes[b.first].count;
            }
            n_slots64 += 1u + unit_tris(c);
            if (n_slots64 > kLeafFirstMask) return fail(PT_ERR_LIMIT, "triangles + leaves exceed 2^27-1 leaf-array records");
        }
        unit_record[units.size()] = uint32_t(n_slots64);
    }
    clk.lap("navigation tree + leaf slots");
    // reference-order nodes
    if (int rc = fill_upload<float4>(
            s, d->n_nodes, [](size_t u) { return 2 * u; }, &dev.nodes, [&](size_t i, float4* o) {
                const pt_bvh_node& b = d->nodes[i];
                uint32_t first = b.count ? slot_of[i] + (headless(uint32_t(i)) ? 0u : 1u) : b.first, count = b.count;
                float ff, fc;
                std::memcpy(&ff, &first, 4);
                std::memcpy(&fc, &count, 4);
                o[0] = make_float4(b.lo[0], b.lo[1], b.lo[2], b.hi[0]);
                o[1] = make_float4(b.hi[1], b.hi[2], ff, fc);
            }))
        return rc;
    // leaf array: unit = a leaf or a fat leaf: header {parent box} + its triangles
    std::atomic<uint32_t> unreliable{0};
    if (int rc = fill_upload<float4>(
            s, units.size(), [&](size_t u) { return 3 * size_t(unit_record[u]); }, &dev.tris, [&](size_t u, float4* h) {
                const uint32_t c = units[u];
                const pt_bvh_node& cb = d->nodes[c];
                // the box the reference tests before it reaches these triangles: the leaf's parent (for a fat leaf, the fat
                // node itself: the parent of both of its leaves)
                const bool is_root_leaf = c == 0u && cb.count != 0u;
                const pt_bvh_node& pb = cb.count ? d->nodes[parent[c]] : cb;
                const uint32_t flags = is_root_leaf ? kLeafUntested : 0u;
                float ffl;
                std::memcpy(&ffl, &flags, 4);
                h[0] = make_float4(pb.lo[0], pb.lo[1], pb.lo[2], pb.hi[0]);
                h[1] = make_float4(pb.hi[1], pb.hi[2], ffl, 0.0f);
                h[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float4* o = h + 3;
                const uint32_t leaves[2] = {cb.count ? c : cb.first, cb.count ? c : cb.first + 1u};
                for (int l = 0; l < (cb.count ? 1 : 2); ++l) {
                    const pt_bvh_node& b = d->nodes[leaves[l]];
                    for (uint32_t k = 0; k < b.count; ++k, o += 3) {
                        const uint32_t prim = d->prim_ids[b.first + k];
                        const uint32_t prim_word = prim | (tri_has_normals(d, prim) ? kPrimHasNormals : 0u) | (d->light_id[prim] >= 0 ? kPrimEmitter : 0u);
                        const float* p0 = d->verts + 3 * size_t(d->vi[3 * size_t(prim) + 0] - 1);
                        const float* p1 = d->verts + 3 * size_t(d->vi[3 * size_t(prim) + 1] - 1);
                        const float* p2 = d->verts + 3 * size_t(d->vi[3 * size_t(prim) + 2] - 1);
                        float fp;
                        std::memcpy(&fp, &prim_word, 4);
                        o[0] = make_float4(p0[0], p0[1], p0[2], fp);
                        o[1] = make_float4(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], 0.0f);
                        o[2] = make_float4(p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2], 0.0f);
                        if (unreliable_triangle(&o[1].x, &o[2].x)) unreliable.fetch_add(1u, std::memory_order_relaxed);
                    }
                }
            }))
        return rc;
    s->reference_order_only = unreliable.load() != 0u;
    if (clk.on && s->reference_order_only)
        std::fprintf(stderr, "pt_hip_scene_create: %u needle-shaped or collapsed triangles: every query runs in reference order\n", unreliable.load());
    clk.lap("binary nodes + leaf array");
    // navigation nodes of the ordered traversal
    {
        const pt_bvh_node& root = d->nodes[0];
        double org[3], step[3];
        for (int a = 0; a < 3; ++a) {
            dev.q_org[a] = root.lo[a];
            // spacing rounded up so that grid point 65535 lies at or beyond the far face; never zero (flat scenes)
            const float st = float(std::max((double(root.hi[a]) - double(root.lo[a])) / 65535.0 * (1.0 + 1e-6), 1e-30));
            dev.q_step[a] = st;
            org[a] = dev.q_org[a];
            step[a] = dev.q_step[a];
        }
        // The grid planes that enclose a box most tightly: the largest grid point at or below its lower face, the smallest
        // at or above its upper face (checked in extended precision, where org + g * step is exact for these magnitudes).
        // The device never forms a plane's position: its distance estimates carry their own padding (box_cheap).
        // PT_GRID_SAFETY=1 in the environment adds one grid step on either side, as earlier versions did.
        const double safety = (std::getenv("PT_GRID_SAFETY") && std::atoi(std::getenv("PT_GRID_SAFETY")) != 0) ? 1.0 : 0.0;
        // (clamped to the grid before the check, so that a face that is not finite ends at the grid's end instead of in a loop)
        auto plane = [&](double g, int a) { return (long double)(org[a]) + (long double)(g) * (long double)(step[a]); };
        auto q_lo = [&](float v, int a) -> uint32_t {
            double g = std::floor((double(v) - org[a]) / step[a]);
            if (!(g >= 0.0)) g = 0.0;
            if (g > 65535.0) g = 65535.0;
            for (int k = 0; k < 2 && g > 0.0 && plane(g, a) > (long double)(v); ++k) g -= 1.0;
            return uint32_t(std::max(0.0, g - safety));
        };
        auto q_hi = [&](float v, int a) -> uint32_t {
            double g = std::ceil((double(v) - org[a]) / step[a]);
            if (!(g <= 65535.0)) g = 65535.0;
            if (g < 0.0) g = 0.0;
            for (int k = 0; k < 2 && g < 65535.0 && plane(g, a) < (long double)(v); ++k) g += 1.0;
            return uint32_t(std::min(65535.0, g + safety));
        };
        auto leaf_ref = [&](uint32_t ni) -> uint32_t {  // of a leaf or fat leaf
            const pt_bvh_node& b = d->nodes[ni];
            return kLeafBit | (unit_tris(ni) << kLeafCountShift) | slot_of[b.count ? ni : b.first];
        };
        auto ref_of = [&](uint32_t ni) -> uint32_t { return leaf_like(ni) ? leaf_ref(ni) : wide_id[ni]; };
        dev.n_treelet = uint32_t(std::min<size_t>(kTreeletMax, order.size()));
        std::atomic<int> bad{0};
        if (int rc = fill_upload<uint4>(
                s, std::max<size_t>(order.size(), 1), [](size_t u) { return size_t(kNavVec4) * u; }, &dev.qnodes, [&](size_t wi, uint4* w) {
                    // (a tree with navigation nodes has two leaves at least: four records or more)
                    for (int k = 0; k < kNavVec4; ++k) w[k] = make_uint4(0u, 0u, 0u, kWide == 4 ? empty_slot_ref(unit_record[units.size()] - 2u) : 0u);
                    if (wi >= order.size()) return;  // a scene whose root is a leaf has no navigation node
                    const uint32_t n = order[wi];
                    uint32_t slots[kWide];
                    const int n_slots = collapse(n, slots);
                    if (kWide == 4) {
                        // 64 bytes: per slot the box on a 16-bit grid over the root box, rounded outwards, and its reference
                        for (int k = 0; k < n_slots; ++k) {
                            const pt_bvh_node& c = d->nodes[slots[k]];
                            uint4& o = w[k];
                            o.x = q_lo(c.lo[0], 0) | (q_hi(c.hi[0], 0) << 16);
                            o.y = q_lo(c.lo[1], 1) | (q_hi(c.hi[1], 1) << 16);
                            o.z = q_lo(c.lo[2], 2) | (q_hi(c.hi[2], 2) << 16);
                            o.w = ref_of(slots[k]);
                        }
                        return;
                    }
                    // 80 bytes (traverse.h, Nav8): an 8-bit grid over this node's own box, origin = its lower corner
                    const pt_bvh_node& nb = d->nodes[n];
                    uint32_t ebits[3];
                    double sp[3];
                    for (int a = 0; a < 3; ++a) {
                        const double extent = double(nb.hi[a]) - double(nb.lo[a]);
                        int k = -126;
                        if (extent > 0.0) k = std::max(-126, int(std::ceil(std::log2(extent / 252.0))));
                        while (std::ldexp(253.0, k) < extent) ++k;   // (log2 rounding)
                        if (k > 127) bad.store(1);
                        ebits[a] = uint32_t(k + 127);
                        sp[a] = std::ldexp(1.0, k);
                    }
                    float px = nb.lo[0], py = nb.lo[1], pz = nb.lo[2];
                    std::memcpy(&w[0].x, &px, 4);
                    std::memcpy(&w[0].y, &py, 4);
                    std::memcpy(&w[0].z, &pz, 4);
                    w[0].w = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16);
                    w[1].x = child_base[wi];
                    w[1].y = leaf_base[wi];
                    uint32_t expect_child = child_base[wi], expect_leaf = leaf_base[wi];
                    for (int k = 0; k < n_slots; ++k) {
                        const uint32_t c = slots[k];
                        const pt_bvh_node& cb = d->nodes[c];
                        uint32_t meta;
                        if (leaf_like(c)) {
                            meta = unit_tris(c);
                            if (slot_of[cb.count ? c : cb.first] != expect_leaf || meta == 0u || meta > 15u) bad.store(2);
                            expect_leaf += 1u + meta;
                        } else {
                            meta = kNav8Interior;
                            if (wide_id[c] != expect_child) bad.store(3);
                            ++expect_child;
                        }
                        uint32_t& mw = k < 4 ? w[1].z : w[1].w;
                        mw |= meta << (8 * (k & 3));
                        for (int a = 0; a < 3; ++a) {
                            const double lo = (double(cb.lo[a]) - double(nb.lo[a])) / sp[a], hi = (double(cb.hi[a]) - double(nb.lo[a])) / sp[a];
                            const uint32_t ql = uint32_t(std::min(255.0, std::max(0.0, std::floor(lo) - 1.0)));
                            const uint32_t qh = uint32_t(std::min(255.0, std::max(0.0, std::ceil(hi) + 1.0)));
                            // the stored planes enclose the slot's box (exactly representable in double)
                            if (double(nb.lo[a]) + ql * sp[a] > double(cb.lo[a]) || double(nb.lo[a]) + qh * sp[a] < double(cb.hi[a])) bad.store(4);
                            uint4& v = w[2 + a];
                            uint32_t& wl = k < 4 ? v.x : v.y;
                            uint32_t& wh = k < 4 ? v.z : v.w;
                            wl |= ql << (8 * (k & 3));
                            wh |= qh << (8 * (k & 3));
                        }
                    }
                }))
            return rc;
        if (bad.load()) return fail(PT_ERR_INVALID, "navigation node packing failed its own check (" + std::to_string(bad.load()) + "): boxes outside the tree's nesting or not finite");
        dev.wide_root = ref_of(0);
    }
    clk.lap("navigation nodes");
    // shading records: one 128-byte record per triangle, one 96-byte record per light
    {
        auto vert = [&](uint32_t prim, int k) { return d->verts + 3 * size_t(d->vi[3 * size_t(prim) + k] - 1); };
        auto has_normals = [&](uint32_t prim) { return tri_has_normals(d, prim); };
        auto normal = [&](uint32_t prim, int k) { return d->normals + 3 * size_t(d->ni[3 * size_t(prim) + k] - 1); };
        for (uint32_t t = 0; t < d->n_tris; ++t)
            if (has_normals(t))
                for (int k = 0; k < 3; ++k)
                    if (d->ni[3 * size_t(t) + k] == 0) return fail(PT_ERR_INVALID, "face with a partial set of normal indices");
        auto bits = [](uint32_t u) {
            float f;
            std::memcpy(&f, &u, 4);
            return f;
        };
        auto pack_geometry = [&](uint32_t prim, float4* o) {  // v0 v1 v2 n0 n1 n2 in 4.5 float4 (light records)
            const float *a = vert(prim, 0), *b = vert(prim, 1), *c = vert(prim, 2);
            float n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (has_normals(prim))
                for (int k = 0; k < 3; ++k) std::memcpy(n + 3 * k, normal(prim, k), 12);
            o[0] = make_float4(a[0], a[1], a[2], b[0]);
            o[1] = make_float4(b[1], b[2], c[0], c[1]);
            o[2] = make_float4(c[2], n[0], n[1], n[2]);
            o[3] = make_float4(n[3], n[4], n[5], n[6]);
            o[4] = make_float4(n[7], n[8], 0.0f, 0.0f);
        };
        if (d->n_tris > kPrimMask) return fail(PT_ERR_LIMIT, "more than 2^29-1 triangles");
        bool any_normals = false;
        for (uint32_t t = 0; t < d->n_tris && !any_normals; ++t) any_normals = has_normals(t);
        if (int rc = fill_upload<float4>(
                s, d->n_tris, [](size_t u) { return size_t(kTriRecVec4) * u; }, &dev.trec, [&](size_t ti_, float4* o) {
                    const uint32_t t = uint32_t(ti_);
                    const float *a = vert(t, 0), *b = vert(t, 1), *c = vert(t, 2);
                    uint32_t flags = has_normals(t) ? kTriHasNormals : 0u;
                    if (d->n_uvs != 0 && d->ti && d->ti[3 * size_t(t)] != 0) flags |= kTriHasUvs;
                    const int32_t lid = d->light_id[t];
                    o[0] = make_float4(a[0], a[1], a[2], b[0]);
                    o[1] = make_float4(b[1], b[2], c[0], c[1]);
                    o[2] = make_float4(c[2], bits(flags), bits(uint32_t(lid)), bits(d->mat_id[t]));
                    o[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (lid >= 0) {
                        const pt_light& l = d->lights[lid];
                        o[3] = make_float4(float(1.0 / double(l.area)), l.lemit[0], l.lemit[1], l.lemit[2]);
                    }
                }))
            return rc;
        if (int rc = fill_upload<float4>(
                s, any_normals ? d->n_tris : 1u, [](size_t u) { return size_t(kTriNormalVec4) * u; }, &dev.tnrm, [&](size_t ti_, float4* o) {
                    const uint32_t t = uint32_t(ti_);
                    float n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    if (any_normals && has_normals(t))
                        for (int k = 0; k < 3; ++k) std::memcpy(n + 3 * k, normal(t, k), 12);
                    o[0] = make_float4(n[0], n[1], n[2], n[3]);
                    o[1] = make_float4(n[4], n[5], n[6], n[7]);
                    o[2] = make_float4(n[8], 0.0f, 0.0f, 0.0f);
                }))
            return rc;
        {
            std::vector<float4> mats(size_t(kMaterialVec4) * std::max<uint32_t>(d->n_mats, 1u));
            for (uint32_t i = 0; i < d->n_mats; ++i) {
                const pt_material& m = d->mats[i];
                mats[size_t(kMaterialVec4) * i] = make_float4(m.kd[0], m.kd[1], m.kd[2], m.ns);
                mats[size_t(kMaterialVec4) * i + 1] = make_float4(m.ks[0], m.ks[1], m.ks[2], bits(uint32_t(m.tex)));
            }
            if (int rc = upload(s, mats.data(), mats.size(), &dev.mats)) return rc;
        }
        std::vector<float4> lrec(size_t(kLightRecVec4) * d->n_lights);
        for (uint32_t i = 0; i < d->n_lights; ++i) {
            const pt_light& l = d->lights[i];
            float4* o = &lrec[size_t(kLightRecVec4) * i];
            pack_geometry(l.tri, o);
            o[4].z = bits(has_normals(l.tri) ? kTriHasNormals : 0u);
            o[4].w = float(1.0 / double(l.area));
            o[5] = make_float4(l.lemit[0], l.lemit[1], l.lemit[2], 0.0f);
        }
        if (int rc = upload(s, lrec.data(), lrec.size(), &dev.lrec)) return rc;
        if (int rc = upload(s, d->uvs, 2 * size_t(d->n_uvs), &dev.uvs)) return rc;
        if (int rc = upload(s, d->ti, d->ti ? 3 * size_t(d->n_tris) : 0, &dev.ti)) return rc;
        std::vector<DevTexture> t(d->n_textures);
        std::vector<float> texels;
        for (uint32_t i = 0; i < d->n_textures; ++i) {
            const pt_texture& src = d->textures[i];
            if (!src.rgb || src.w == 0 || src.h == 0) return fail(PT_ERR_INVALID, "empty texture");
            t[i] = DevTexture{src.w, src.h, uint64_t(texels.size())};
            texels.insert(texels.end(), src.rgb, src.rgb + size_t(3) * src.w * src.h);
        }
        if (int rc = upload(s, t.data(), t.size(), &dev.textures)) return rc;
        if (int rc = upload(s, texels.data(), texels.size(), &dev.tex_data)) return rc;
    }
    clk.lap("shading records + textures");
    dev.n_tris = d->n_tris;
    dev.n_lights = d->n_lights;
    {
        const pt_bvh_node& root = d->nodes[0];
        float m = 0.0f;
        for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(root.lo[a]), std::fabs(root.hi[a])));
        dev.all_special = (m < 0x1p59f) ? 0u : 1u;  // also 1 when a bound is not a number
    }
    // camera
    std::memcpy(dev.cam.s2c, d->camera.sample2camera, 64);
    std::memcpy(dev.cam.c2w, d->camera.camera2world, 64);
    std::memcpy(dev.cam.eye, d->camera.eye, 12);
    dev.cam.proj_nume = d->camera.proj_nume;
    dev.cam.width = d->camera.width;
    dev.cam.height = d->camera.height;
    s->width = d->camera.width;
    s->height = d->camera.height;
    // Sobol' tables.  scale and m as in SobolSampler (src/sampler.cpp:89-100); for a power of two log2int is exact.
    // X[px][dim] / Y[py][dim]: the Sobol' word of dimension dim for the index bits that pixel column px / row py
    // contributes through sobolIntervalToIndex (src/sampler.cpp:29-33).
    {
        const uint32_t scale = uint32_t(round_up_pow2(int32_t(std::max(d->camera.width, d->camera.height))));
        uint32_t m = 0;
        while ((1u << m) < scale) ++m;
        s->sob_m = m;
        dev.sob_scale = scale;
        std::vector<uint32_t> X(size_t(1024) * scale, 0u), Y(size_t(1024) * scale, 0u);
        for (int dim = 0; dim < 1024; ++dim) {
            uint32_t col[64];
            for (uint32_t c = 0; c < 2 * m; ++c) col[c] = sobol_word(kVdCSobolMatricesInv[m - 1][c], dim);
            for (uint32_t p = 1; p < scale; ++p) {
                const int low = __builtin_ctz(p);
                const uint32_t prev = p & (p - 1);
                Y[size_t(p) * 1024 + dim] = Y[size_t(prev) * 1024 + dim] ^ col[low];      // pixel y: bits 0..m-1 of the interval number
                X[size_t(p) * 1024 + dim] = X[size_t(prev) * 1024 + dim] ^ col[m + low];  // pixel x: bits m..2m-1
            }
        }
        if (int rc = upload(s, X.data(), X.size(), &dev.sob_x)) return rc;
        if (int rc = upload(s, Y.data(), Y.size(), &dev.sob_y)) return rc;
    }
    // tabulated Gaussian filter, radius 2, stddev 0.5 (src/block.cpp:14-25, include/pt/filter.h:26-35)
    {
        const float radius = 2.0f, stddev = 0.5f;
        const float alpha = -1.0f / (2.0f * stddev * stddev);
        for (int i = 0; i < 32; ++i) {
            const float pos = (radius * i) / 32;
            dev.filter[i] = std::max(0.0f, std::exp(alpha * pos * pos) - std::exp(alpha * radius * radius));
        }
        dev.filter[32] = 0.0f;
    }
    return PT_OK;
}

// ---- work buffers ---------------------------------------------------------------------------------------------------

template <typename T>
int ws_alloc(Workspace& ws, T** p, size_t count) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
    ws.owned.push_back(d);
    *p = static_cast<T*>(d);
    return PT_OK;
}

void ws_release(Workspace& ws) {
    for (void* p : ws.owned) (void)hipFree(p);
    ws.owned.clear();
    if (ws.h_counters) (void)hipHostFree(ws.h_counters);
    for (hipEvent_t e : ws.ring_ev)
        if (e) (void)hipEventDestroy(e);
    ws = Workspace();
}

// the eleven per-shard counter arrays for K shards in play, back to back at the start of ws.counters (ws.misc stays where
// ws_reserve put it: behind the arrays of the pass's full shard count)
void ws_layout(Workspace& ws, uint32_t n_shards) {
    const size_t K = n_shards;
    ws.q[0].count = ws.counters;
    ws.q[1].count = ws.counters + K;
    ws.hit_count[0] = ws.counters + 2 * K;
    ws.hit_count[1] = ws.counters + 3 * K;
    ws.hq.count = ws.hit_count[0];
    ws.sh.count = ws.counters + 4 * K;
    ws.fetch_e = ws.counters + 5 * K;
    ws.fetch_s = ws.counters + 6 * K;
    ws.dqe.count = ws.counters + 7 * K;
    ws.dqs.count = ws.counters + 8 * K;
    ws.done_e = ws.counters + 9 * K;
    ws.done_s = ws.counters + 10 * K;
}

int ws_reserve(Workspace& ws, uint32_t n_slots, size_t n_frames) {
    uint32_t n_shards = (n_slots + kShard - 1) / kShard;
    const uint32_t gz = (n_shards + kMaxGridY - 1) / kMaxGridY;
    if (gz > kMaxGridZ) return fail(PT_ERR_LIMIT, "too many samples in flight for one pass: lower samples_per_pass");
    n_shards = (n_shards + gz - 1) / gz * gz;  // whole grid.z layers; the extra shards stay empty
    const uint32_t padded = n_shards * kShard;
    if (padded > ws.cap_slots || n_frames > ws.frame_cap) {
        ws_release(ws);
        ws.n_shards = n_shards;
        for (int k = 0; k < 2; ++k) {
            if (int rc = ws_alloc(ws, &ws.q[k].a, padded)) return rc;
            if (int rc = ws_alloc(ws, &ws.q[k].b, padded)) return rc;
            if (int rc = ws_alloc(ws, &ws.q[k].c, padded)) return rc;
            if (int rc = ws_alloc(ws, &ws.q[k].slot, padded)) return rc;
        }
        if (int rc = ws_alloc(ws, &ws.hq.h0, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.hq.h1, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.sh.a, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.sh.b, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.sh.c, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.dqe.e, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.dqs.e, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.lbuf, padded)) return rc;
        if (int rc = ws_alloc(ws, &ws.counters, ws.counter_words())) return rc;
        if (int rc = ws_alloc(ws, &ws.frame_tab, n_frames * 1024)) return rc;
        if (int rc = ws_alloc(ws, &ws.finish_list, std::max(kFinishPaths, kTailPaths))) return rc;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ws.h_counters), (size_t(kRing) * ws.snapshot_words() + kMiscWords) * sizeof(uint32_t),
                              hipHostMallocDefault));
        for (hipEvent_t& e : ws.ring_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ws.cap_slots = padded;
        ws.frame_cap = n_frames;
    }
    // a smaller request reuses the buffers with fewer shards in play; counters are laid out for this request
    ws.n_shards = n_shards;
    ws.misc = ws.counters + ws.misc_offset();
    ws_layout(ws, n_shards);
    return PT_OK;
}

// ---- per-kernel timing with HIP events on the launch stream ---------------------------------------------------------

struct Timer {
    pt_hip_scene* s;
    bool on;
    int begin(int kind) {
        if (!on) return PT_OK;
        EventPool& p = s->events;
        if (p.used * 2 + 2 > p.ev.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                p.ev.push_back(e);
            }
            p.kind.push_back(kind);
        }
        p.kind[p.used] = kind;
        HIP_TRY(hipEventRecord(p.ev[2 * p.used], g_stream));
        return PT_OK;
    }
    int end() {
        if (!on) return PT_OK;
        EventPool& p = s->events;
        HIP_TRY(hipEventRecord(p.ev[2 * p.used + 1], g_stream));
        ++p.used;
        return PT_OK;
    }
    // call after a stream synchronisation
    int collect(pt_stats* st) {
        EventPool& p = s->events;
        for (size_t i = 0; i < p.used; ++i) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
            if (st) {
                st->kernel_ms[p.kind[i]] += ms;
                st->kernel_launches[p.kind[i]] += 1;
            }
        }
        p.used = 0;
        return PT_OK;
    }
};

inline uint32_t blocks(uint32_t n, uint32_t b) { return uint32_t((uint64_t(n) + b - 1) / b); }

// Workgroups per shard of a wave-refill kernel.  A wave that keeps refilling its lanes only runs at poor occupancy once,
// while it drains at the end, so the fewer waves share a queue the better (measured with 32 k shards: 8 per shard 2.77,
// 4: 2.84, 2: 2.87, 1: 2.75 Grays/s) -- as long as the launch still has enough workgroups to fill the chip.
inline uint32_t wave_blocks(uint32_t n_shards, uint32_t largest_fill) {
    const uint32_t want = std::min(kMaxWaveBlocks, std::max(kMinWaveBlocks, blocks(16384, std::max(n_shards, 1u))));
    return std::min(want, std::max(1u, blocks(largest_fill, kBlock)));
}
// The same for k_shade (256 threads): one workgroup per shard is best once there are enough shards (fewer waves contend
// for the shard's queue counters; measured with 32 k shards: 8 per shard 33.1, 4: 29.8, 2: 27.5, 1: 25.9 ms).
inline uint32_t shade_blocks(uint32_t n_shards, uint32_t largest_fill) {
    const uint32_t want = std::min(kShadeBlocks, std::max(1u, blocks(8192, std::max(n_shards, 1u))));
    return std::min(want, std::max(1u, blocks(largest_fill, 256)));
}

struct Fill {
    uint32_t max = 0;
    uint64_t sum = 0;
};
inline Fill fill_of(const uint32_t* counts, uint32_t n) {
    Fill f;
    for (uint32_t i = 0; i < n; ++i) {
        f.max = std::max(f.max, counts[i]);
        f.sum += counts[i];
    }
    return f;
}

int launch_generate(pt_hip_scene* s, const PassArgs& pa, Timer& tm, bool count) {
    Workspace& ws = s->ws;
    HIP_TRY(hipMemsetAsync(ws.counters, 0, ws.counter_words() * sizeof(uint32_t), g_stream));
    if (int rc = tm.begin(PT_K_GENERATE)) return rc;
    ws.hq.count = ws.hit_count[0];
    if (pa.traversal == PT_TRAVERSE_FAST) {
        // the extend kernel's fetch / deferred / done counters are free until the first extend (zeroed above, and again by
        // the first shade)
        const GenerateIo io{pa, ws.q[0], ws.hq, ws.dqe, ws.lbuf};
        const dim3 grid = shard_grid(wave_blocks(ws.n_shards, pa.n_slots), ws.n_shards);
        if (count)
            hipLaunchKernelGGL((k_generate_wave<Count, ClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, ws.misc);
        else
            hipLaunchKernelGGL((k_generate_wave<NoCount, ClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, ws.misc);
    } else if (count) {
        // reference order: camera rays of a tile stay together all the way, one lane per sample slot without refilling
        // is the faster form here (measured: 91 against 132 ms per 256 spp on the 1.6 M triangle interior)
        hipLaunchKernelGGL(k_generate<Count>, dim3(blocks(pa.n_slots, kBlock)), dim3(kBlock), 0, g_stream, s->dev, pa, ws.q[0], ws.hq, ws.lbuf,
                           ws.misc);
    } else {
        hipLaunchKernelGGL(k_generate<NoCount>, dim3(blocks(pa.n_slots, kBlock)), dim3(kBlock), 0, g_stream, s->dev, pa, ws.q[0], ws.hq,
                           ws.lbuf, ws.misc);
    }
    if (int rc = tm.end()) return rc;
    // every path that k_generate kept is also a hit-list entry at the same position
    HIP_TRY(hipMemcpyAsync(ws.hit_count[0], ws.q[0].count, ws.n_shards * sizeof(uint32_t), hipMemcpyDeviceToDevice, g_stream));
    return PT_OK;
}

constexpr uint32_t kMaxIterations = 1u << 20;  // of the bounce loop, per pass

// The bounce loop shared by pt_hip_render_device and pt_hip_li; k_generate has filled path queue 0 and hit list 0.
//
// The host never waits for a kernel it needs the result of right away.  After the shade of bounce b it queues a copy of
// the queue counters into a pinned snapshot and launches shadow(b) and extend(b) without looking at it; the snapshot is
// read before bounce b+1 is launched, by which time it has long arrived (shadow and extend are the long kernels).  That
// is enough because every kernel walks its shard's queue with a block-stride loop bounded by the device-side counter, so
// grids do not depend on the fills, and a shard's fill can only shrink from bounce to bounce (a path stays in its shard
// and produces at most one continuation and one shadow ray), so the fills of bounce b bound those of b+1 -- until the
// paths are gathered into fewer, full shards (below), which starts the bound afresh.
// Measured: letting the host run further ahead gains nothing (the gaps were never the syncs), and grids sized by stale
// fills cost a lot -- a workgroup that finds nothing to do still takes ~0.2 us of dispatch (scratch + 12 KB LDS).
int run_bounces(pt_hip_scene* s, const PassArgs& pa, pt_stats* st, Timer& tm, bool count) {
    Workspace& ws = s->ws;
    uint32_t K = ws.n_shards;          // shards in play: all of the pass's until the paths are gathered into a few (below)
    size_t snap = ws.snapshot_words();
    const bool fast = pa.traversal == PT_TRAVERSE_FAST;
    uint32_t bound = std::min<uint32_t>(kShard, pa.n_slots);  // upper bound of every shard fill from here on
    bool finished = false;
    bool tail_done = false;
    bool regrouped = false;  // the snapshot about to be read lists paths a second time (in their new shards after k_compact): not new rays
    uint64_t finish_closest = 0, finish_any = 0;  // k_finish's query totals read so far in this pass
    uint64_t live = ~0ull;                 // paths alive after the latest snapshot read
    const uint32_t* live_counts = nullptr;  // ... per shard (in the pinned ring: valid until kRing more snapshots are queued)
    uint32_t launched = 0, consumed = 0;  // snapshots queued / read; snapshot i describes bounce i (taken after its shade)
    auto consume = [&](bool wait) -> int {
        while (consumed < launched) {
            hipEvent_t ev = ws.ring_ev[consumed % kRing];
            if (wait) {
                HIP_TRY(hipEventSynchronize(ev));
            } else {
                const hipError_t q = hipEventQuery(ev);
                if (q == hipErrorNotReady) return PT_OK;
                HIP_TRY(q);
            }
            const uint32_t* c = ws.h_counters + size_t(consumed % kRing) * snap;
            const int nxt = int(consumed & 1u) ^ 1;
            const Fill next = fill_of(c + size_t(nxt) * K, K), shadow = fill_of(c + 4 * size_t(K), K);
            if (st && !regrouped) {
                st->any_rays += shadow.sum;
                st->closest_rays += next.sum;
                // the hit list this bounce's shade worked on (still in place: the extend that refills it runs later)
                st->shaded_hits += fill_of(c + size_t(2 + int(consumed & 1u)) * K, K).sum;
            }
            regrouped = false;
            bound = std::min(bound, std::max(next.max, 1u));
            if (next.max == 0) finished = true;
            live = next.sum;
            live_counts = c + size_t(nxt) * K;
            ++consumed;
        }
        return PT_OK;
    };
    // Every loop iteration is one pass of the reference's loop body.  A specular bounce does not advance the bounce counter
    // (src/integrator.cpp:84-87), so max_bounce does not bound the iterations: a mirror with interpolated normals can
    // reflect a ray below its own surface, and the path then creeps along a crease for thousands of reflections before
    // it escapes.  The reference follows it to the end and so does this loop (a few stragglers cost three small launches
    // per reflection); kMaxIterations only stops a path that never ends, on which the reference would not return.
    for (uint32_t iter = 0; iter < kMaxIterations; ++iter) {
        if (int rc = consume(true)) return rc;  // the previous bounce's snapshot
        if (finished) break;
        // Russian roulette has thinned the paths out, a fraction of every shard is in use: gather them into the first shards
        // (k_compact) and go on with those -- again whenever the shards in play have emptied that far.  The counter arrays are laid out anew for the smaller shard count at the start of the
        // counter block; the old hit counts, which k_compact reads, lie behind that (11 Kc <= 2 K).  The gathered hit list
        // goes into the shadow queue's buffers (free between two bounces), which then swap roles with the hit list's.
        {
            uint32_t Kc = 0;
            if (live - 1 < uint64_t(pa.n_slots)) {  // a snapshot has been read and something is alive
                Kc = uint32_t((live + kShard - 1) / kShard);
                const uint32_t gz = (Kc + kMaxGridY - 1) / kMaxGridY;
                Kc = (Kc + gz - 1) / gz * gz;  // whole grid.z layers, as shard_grid expects
            }
            if (Kc != 0 && pa.list_xy == nullptr && iter >= 1 && live > kTailPaths && live * kCompactShare <= std::min<uint64_t>(pa.n_slots, uint64_t(K) * kShard) &&
                size_t(11) * Kc <= size_t(2) * K) {
                const int cur = int(iter & 1u), nxt = cur ^ 1;
                HitQ hq_old = ws.hq;
                hq_old.count = ws.hit_count[cur];
                const PathQ q_old = ws.q[cur];
                HIP_TRY(hipMemsetAsync(ws.counters, 0, size_t(11) * Kc * sizeof(uint32_t), g_stream));
                ws_layout(ws, Kc);
                if (int rc = tm.begin(PT_K_COMPACT)) return rc;
                uint32_t* offsets = reinterpret_cast<uint32_t*>(ws.dqe.e);  // K + 1 words of the deferred-ray queue, idle between two bounces
                hipLaunchKernelGGL(k_shard_offsets, dim3(1), dim3(kOffsetsBlock), 0, g_stream, hq_old.count, K, offsets);
                hipLaunchKernelGGL(k_compact, shard_grid(1, K), dim3(256), 0, g_stream, hq_old, q_old, ws.q[nxt], ws.sh.a, ws.sh.b, offsets);
                hipLaunchKernelGGL(k_dense_counts, dim3(blocks(Kc, 256)), dim3(256), 0, g_stream, offsets + K, ws.q[nxt].count, ws.hit_count[nxt], Kc);
                if (int rc = tm.end()) return rc;
                std::swap(ws.hq.h0, ws.sh.a);
                std::swap(ws.hq.h1, ws.sh.b);
                K = Kc;
                snap = size_t(5) * Kc;
                bound = std::min<uint32_t>(kShard, uint32_t(live));
                // this iteration's snapshot: the same paths, in their new shards (they are not counted as rays again)
                HIP_TRY(hipMemcpyAsync(ws.h_counters + size_t(launched % kRing) * snap, ws.counters, snap * sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
                HIP_TRY(hipEventRecord(ws.ring_ev[launched % kRing], g_stream));
                ++launched;
                regrouped = true;
                continue;
            }
        }
        // k_finish takes over (see there): for good once only chains of specular bounces are left and few of them, or once
        // per pass for up to max_bounce iterations when Russian roulette has thinned the paths out -- what is alive after
        // that (specular chains again) comes back to the queues and the loop goes on with it.  List-mode calls (pt_hip_li)
        // keep to the regular kernels for the tail: they are how single samples of those kernels are checked.
        const bool chains = iter >= pa.max_bounce && live <= kFinishPaths;
        const bool tail = !tail_done && pa.list_xy == nullptr && iter >= kTailStart && live <= kTailPaths;
        if (chains || tail) {
            // `launched == consumed` here: live_counts describes the queue that the hit list of this iteration came from
            // (a shard's hit list is at most as long as the path queue it came from)
            std::vector<uint2> list;
            for (uint32_t k = 0; k < K; ++k)
                for (uint32_t j = 0; j < live_counts[k]; ++j) list.push_back(make_uint2(k, j));
            const int cur = int(iter & 1u), nxt = cur ^ 1;
            ws.hq.count = ws.hit_count[cur];
            HitQ hq_next = ws.hq;
            hq_next.count = ws.hit_count[nxt];
            HIP_TRY(hipMemcpyAsync(ws.finish_list, list.data(), list.size() * sizeof(uint2), hipMemcpyHostToDevice, g_stream));
            // the counters as the shade and extend kernels of this iteration would have left them, bar the paths that
            // come back: those are appended by k_finish
            for (uint32_t* p : {ws.q[cur].count, hq_next.count, ws.fetch_e, ws.dqe.count, ws.done_e}) HIP_TRY(hipMemsetAsync(p, 0, K * sizeof(uint32_t), g_stream));
            if (int rc = tm.begin(PT_K_FINISH)) return rc;
            const uint32_t budget = chains ? kMaxIterations - iter : pa.max_bounce;
            const uint32_t lanes = chains ? 1u : kTailLanes;  // paths per wave
            if (count)
                hipLaunchKernelGGL(k_finish<Count>, dim3(blocks(uint32_t(list.size()), lanes)), dim3(kFinishBlock), 0, g_stream, s->dev, pa, ws.hq, ws.q[cur], ws.q[nxt], hq_next,
                                   ws.finish_list, uint32_t(list.size()), lanes, ws.lbuf, budget, ws.misc);
            else
                hipLaunchKernelGGL(k_finish<NoCount>, dim3(blocks(uint32_t(list.size()), lanes)), dim3(kFinishBlock), 0, g_stream, s->dev, pa, ws.hq, ws.q[cur], ws.q[nxt], hq_next,
                                   ws.finish_list, uint32_t(list.size()), lanes, ws.lbuf, budget, ws.misc);
            if (int rc = tm.end()) return rc;
            uint32_t* h = ws.h_misc() + 16;
            HIP_TRY(hipMemcpyAsync(h, ws.misc + 16, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
            // this iteration's snapshot: the paths that came back
            HIP_TRY(hipMemcpyAsync(ws.h_counters + size_t(launched % kRing) * snap, ws.counters, snap * sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
            HIP_TRY(hipEventRecord(ws.ring_ev[launched % kRing], g_stream));
            ++launched;
            HIP_TRY(hipStreamSynchronize(g_stream));  // `list` must outlive its copy, and the totals are read now
            const uint64_t fc = uint64_t(h[0]) | (uint64_t(h[1]) << 32), fa = uint64_t(h[2]) | (uint64_t(h[3]) << 32);
            if (st) {
                st->closest_rays += fc - finish_closest;  // the totals run over the pass
                st->any_rays += fa - finish_any;
            }
            finish_closest = fc;
            finish_any = fa;
            if (chains) break;
            tail_done = true;
            continue;
        }
        if (iter != 0 && iter % 256u == 0) {  // long chains: fold the timing events in now instead of keeping three pairs per iteration
            HIP_TRY(hipStreamSynchronize(g_stream));
            if (int rc = tm.collect(st)) return rc;
        }
        const int cur = int(iter & 1u), nxt = cur ^ 1;
        ws.hq.count = ws.hit_count[cur];
        HitQ hq_next = ws.hq;
        hq_next.count = ws.hit_count[nxt];
        if (int rc = tm.begin(PT_K_SHADE)) return rc;
        hipLaunchKernelGGL(k_shade, shard_grid(shade_blocks(K, bound), K), dim3(256), 0, g_stream, s->dev, pa, ws.hq, ws.q[cur], ws.q[nxt], ws.sh, ws.lbuf,
                           ZeroList{{hq_next.count, ws.fetch_e, ws.dqe.count, ws.done_e, nullptr}});
        if (int rc = tm.end()) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h_counters + size_t(launched % kRing) * snap, ws.counters, snap * sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
        HIP_TRY(hipEventRecord(ws.ring_ev[launched % kRing], g_stream));
        ++launched;
        {
            if (int rc = tm.begin(PT_K_SHADOW)) return rc;
            const dim3 grid = shard_grid(wave_blocks(K, bound), K);
            const ShadowIo io{ws.sh, ws.dqs, ws.lbuf};
            if (fast && count)
                hipLaunchKernelGGL((k_shadow_wave<Count, AnyLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_s, ws.done_s, ws.misc);
            else if (fast)
                hipLaunchKernelGGL((k_shadow_wave<NoCount, AnyLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_s, ws.done_s, ws.misc);
            else if (count)
                hipLaunchKernelGGL((k_shadow_wave<Count, ExactAnyLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_s, ws.done_s, ws.misc);
            else
                hipLaunchKernelGGL((k_shadow_wave<NoCount, ExactAnyLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_s, ws.done_s, ws.misc);
            if (int rc = tm.end()) return rc;
        }
        {
            // clears what the NEXT bounce's shade and shadow accumulate into (this bounce's shadow kernel is done with them)
            const ZeroList zl{{ws.q[cur].count, ws.sh.count, ws.fetch_s, ws.dqs.count, ws.done_s}};
            if (int rc = tm.begin(PT_K_EXTEND)) return rc;
            const dim3 grid = shard_grid(wave_blocks(K, bound), K);
            const ExtendIo io{ws.q[nxt], hq_next, ws.dqe};
            if (fast && count)
                hipLaunchKernelGGL((k_extend_wave<Count, ClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, zl, ws.misc);
            else if (fast)
                hipLaunchKernelGGL((k_extend_wave<NoCount, ClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, zl, ws.misc);
            else if (count)
                hipLaunchKernelGGL((k_extend_wave<Count, ExactClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, zl, ws.misc);
            else
                hipLaunchKernelGGL((k_extend_wave<NoCount, ExactClosestLane>), grid, dim3(kBlock), 0, g_stream, s->dev, io, ws.fetch_e, ws.done_e, zl, ws.misc);
            if (int rc = tm.end()) return rc;
        }
    }
    if (int rc = consume(true)) return rc;
    HIP_TRY(hipGetLastError());
    ws_layout(ws, ws.n_shards);  // the next pass starts with all shards again
    return PT_OK;
}

// device staging buffer `slot` of at least `bytes` bytes (contents are not preserved when it grows)
template <typename T>
int stage_reserve(pt_hip_scene* s, int slot, size_t bytes, T** out) {
    if (bytes > s->stage_cap[slot]) {
        if (s->stage[slot]) {
            HIP_TRY(hipStreamSynchronize(g_stream));
            (void)hipFree(s->stage[slot]);
        }
        s->stage[slot] = nullptr;
        s->stage_cap[slot] = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
        HIP_TRY(hipMalloc(&s->stage[slot], want));
        s->stage_cap[slot] = want;
    }
    *out = static_cast<T*>(s->stage[slot]);
    return PT_OK;
}

// makes pt_hip_init's device and stream the calling thread's current pair
int bind_default() {
    if (g_default_device < 0) return fail(PT_ERR_NOT_INIT, "pt_hip_init has not been called");
    HIP_TRY(hipSetDevice(g_default_device));
    g_device = g_default_device;
    g_stream = g_default_stream;
    return PT_OK;
}

// this device's copy of the static fetch counters (pt_hip_intersect_device_noalloc); the device must be current
int noalloc_address(int device) {
    if (device >= kMaxDevices) return PT_OK;
    std::lock_guard<std::mutex> lock(g_noalloc_setup);  // handles are created from several threads
    if (!g_noalloc_fetch[device].load(std::memory_order_acquire)) {
        uint32_t* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
        g_noalloc_fetch[device].store(p, std::memory_order_release);
    }
    return PT_OK;
}

// every entry point that takes a handle starts here: the handle's device and stream become the thread's current pair
int check_ready(const pt_hip_scene* s) {
    if (g_default_device < 0) return fail(PT_ERR_NOT_INIT, "pt_hip_init has not been called");
    if (!s) return fail(PT_ERR_INVALID, "scene handle is null");
    HIP_TRY(hipSetDevice(s->device));
    g_device = s->device;
    g_stream = s->stream;
    return PT_OK;
}

// zeroed fetch counters for a batch of n rays cut into shards of kShard
int batch_fetch_reserve(pt_hip_scene* s, uint32_t n_shards) {
    if (n_shards > s->batch_fetch_cap) {
        if (s->batch_fetch) (void)hipFree(s->batch_fetch);
        s->batch_fetch = nullptr;
        s->batch_fetch_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->batch_fetch), size_t(n_shards) * sizeof(uint32_t)));
        s->batch_fetch_cap = n_shards;
    }
    HIP_TRY(hipMemsetAsync(s->batch_fetch, 0, size_t(n_shards) * sizeof(uint32_t), g_stream));
    return PT_OK;
}

// `fetch`: zeroed per-shard fetch counters for the ordered mode (one per kShard rays); null = the handle's own, grown on demand
int launch_intersect(pt_hip_scene* s, const pt_ray* d_rays, size_t n, pt_hit* d_hits, int mode, uint32_t* fetch = nullptr) {
    const float4* rays = reinterpret_cast<const float4*>(d_rays);
    float4* hits = reinterpret_cast<float4*>(d_hits);
    if (s->reference_order_only) mode = PT_TRAVERSE_EXACT;
    if (mode == PT_TRAVERSE_FAST) {
        const uint32_t n_shards = blocks(uint32_t(n), kShard);
        if (!fetch) {
            if (int rc = batch_fetch_reserve(s, n_shards)) return rc;
            fetch = s->batch_fetch;
        }
        for (uint32_t s0 = 0; s0 < n_shards; s0 += kMaxGridY) {  // grid.y is limited to 65535
            const size_t off = size_t(s0) * kShard;
            const BatchClosestIo io{rays + 2 * off, uint32_t(std::min<size_t>(n - off, size_t(kMaxGridY) * kShard)), hits + off};
            hipLaunchKernelGGL(k_intersect_wave<NoCount>, dim3(wave_blocks(blocks(io.n, kShard), io.n), blocks(io.n, kShard)), dim3(kBlock),
                               0, g_stream, s->dev, io, fetch + s0, (uint32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_intersect_fix<NoCount>, dim3(blocks(uint32_t(n), kBlock)), dim3(kBlock), 0, g_stream, s->dev, rays, uint32_t(n),
                           hits, (uint32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_intersect<NoCount>, dim3(blocks(uint32_t(n), kBlock)), dim3(kBlock), 0, g_stream, s->dev, rays, uint32_t(n), hits,
                           (uint32_t*)nullptr);
    }
    return PT_OK;
}

int launch_occluded(pt_hip_scene* s, const pt_ray* d_rays, size_t n, uint8_t* d_occ, int32_t* d_first, int mode) {
    const float4* rays = reinterpret_cast<const float4*>(d_rays);
    if (s->reference_order_only) mode = PT_TRAVERSE_EXACT;
    if (mode == PT_TRAVERSE_FAST) {
        const uint32_t n_shards = blocks(uint32_t(n), kShard);
        if (int rc = batch_fetch_reserve(s, n_shards)) return rc;
        for (uint32_t s0 = 0; s0 < n_shards; s0 += kMaxGridY) {
            const size_t off = size_t(s0) * kShard;
            const BatchAnyIo io{rays + 2 * off, uint32_t(std::min<size_t>(n - off, size_t(kMaxGridY) * kShard)), d_occ + off};
            hipLaunchKernelGGL(k_occluded_wave<NoCount>, dim3(wave_blocks(blocks(io.n, kShard), io.n), blocks(io.n, kShard)), dim3(kBlock),
                               0, g_stream, s->dev, io, s->batch_fetch + s0, (uint32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_occluded_fix<NoCount>, dim3(blocks(uint32_t(n), kBlock)), dim3(kBlock), 0, g_stream, s->dev, rays, uint32_t(n),
                           d_occ, d_first, (uint32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_occluded<NoCount>, dim3(blocks(uint32_t(n), kBlock)), dim3(kBlock), 0, g_stream, s->dev, rays, uint32_t(n), d_occ,
                           d_first, (uint32_t*)nullptr);
    }
    return PT_OK;
}

int normalise(const pt_render_params* in, pt_render_params& p) {
    if (!in) return fail(PT_ERR_INVALID, "render parameters are null");
    p = *in;
    if (p.max_bounce == 0) p.max_bounce = 32;
    if (p.max_bounce > 63) return fail(PT_ERR_INVALID, "max_bounce must be <= 63");
    if (p.spp_end <= p.spp_begin) return fail(PT_ERR_INVALID, "empty sample range");
    if (p.integrator < 0 || p.integrator > 2) return fail(PT_ERR_INVALID, "unknown integrator");
    if (p.traversal < 0 || p.traversal > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (p.block_count > 1 && p.block_rank >= p.block_count) return fail(PT_ERR_INVALID, "block_rank must be below block_count");
    return PT_OK;
}

// Tile tables of an image split: the 8x8 tiles of the 32x32 blocks with (column + row) % count == rank, block after
// block, and the inverse table.  Kept until a call asks for another (rank, count).
int prepare_split(pt_hip_scene* s, uint32_t rank, uint32_t count, uint32_t tiles_x, uint32_t tiles_y) {
    const uint32_t key = (count << 16) | rank;
    if (s->split_key == key) return PT_OK;
    const size_t n_tiles = size_t(tiles_x) * tiles_y;
    if (!s->tile_list) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, n_tiles * sizeof(uint32_t)));
        s->owned.push_back(d);
        s->tile_list = static_cast<uint32_t*>(d);
        HIP_TRY(hipMalloc(&d, n_tiles * sizeof(int32_t)));
        s->owned.push_back(d);
        s->tile_pos = static_cast<int32_t*>(d);
    }
    std::vector<uint32_t> list;
    std::vector<int32_t> pos(n_tiles, -1);
    uint64_t pixels = 0;
    const uint32_t blocks_x = (tiles_x + 3) / 4, blocks_y = (tiles_y + 3) / 4;
    for (uint32_t by = 0; by < blocks_y; ++by)
        for (uint32_t bx = 0; bx < blocks_x; ++bx) {
            if ((bx + by) % count != rank) continue;
            for (uint32_t ty = 4 * by; ty < std::min(4 * by + 4, tiles_y); ++ty)
                for (uint32_t tx = 4 * bx; tx < std::min(4 * bx + 4, tiles_x); ++tx) {
                    pos[size_t(ty) * tiles_x + tx] = int32_t(list.size());
                    list.push_back(ty * tiles_x + tx);
                    pixels += uint64_t(std::min(8u, s->width - 8 * tx)) * std::min(8u, s->height - 8 * ty);
                }
        }
    HIP_TRY(hipStreamSynchronize(g_stream));  // an earlier call's kernels may still read the tables
    if (!list.empty()) HIP_TRY(hipMemcpy(s->tile_list, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->tile_pos, pos.data(), n_tiles * sizeof(int32_t), hipMemcpyHostToDevice));
    s->split_key = key;
    s->split_tiles = uint32_t(list.size());
    s->split_pixels = pixels;
    return PT_OK;
}

// device copies of the host arrays of a probe call, released when the call ends
struct ProbeBuffers {
    std::vector<void*> mem;
    ~ProbeBuffers() {
        for (void* p : mem) (void)hipFree(p);
    }
    template <typename T>
    int in(const T* host, size_t count, T** dev) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(dev), std::max<size_t>(count * sizeof(T), 16)));
        mem.push_back(*dev);
        if (host) HIP_TRY(hipMemcpy(*dev, host, count * sizeof(T), hipMemcpyHostToDevice));
        return PT_OK;
    }
};


}  // namespace

namespace pth {
int fail(int code, const std::string& msg) { return ::fail(code, msg); }
int bind_default() { return ::bind_default(); }
hipStream_t stream() { return g_stream; }
int device() { return g_device; }
}  // namespace pth

// ======================================================================================================== C ABI

extern "C" {

const char* pt_hip_last_error(void) { return g_error.c_str(); }
int pt_hip_abi_version(void) { return PT_HIP_ABI_VERSION; }

int pt_hip_device_count(void) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    return n;
}

int pt_hip_init(int device_ordinal) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(PT_ERR_INVALID, "device ordinal " + std::to_string(device_ordinal) + " out of range (" + std::to_string(n) + " devices)");
    HIP_TRY(hipSetDevice(device_ordinal));
    if (g_default_stream && g_default_device != device_ordinal) {
        (void)hipStreamDestroy(g_default_stream);
        g_default_stream = nullptr;
    }
    if (!g_default_stream) HIP_TRY(hipStreamCreateWithFlags(&g_default_stream, hipStreamNonBlocking));
    if (int rc = noalloc_address(device_ordinal)) return rc;
    g_default_device = device_ordinal;
    return bind_default();
}

int pt_hip_shutdown(void) {
    if (g_default_stream) {
        (void)hipSetDevice(g_default_device);
        (void)hipStreamSynchronize(g_default_stream);
        (void)hipStreamDestroy(g_default_stream);
        g_default_stream = nullptr;
    }
    g_default_device = -1;
    g_device = -1;
    g_stream = nullptr;
    return PT_OK;
}

int pt_hip_scene_create(const pt_scene_desc* desc, pt_hip_scene** out) {
    if (g_default_device < 0) return fail(PT_ERR_NOT_INIT, "pt_hip_init has not been called");
    return pt_hip_scene_create_on(g_default_device, desc, out);
}

int pt_hip_scene_create_on(int device_ordinal, const pt_scene_desc* desc, pt_hip_scene** out) {
    if (g_default_device < 0) return fail(PT_ERR_NOT_INIT, "pt_hip_init has not been called");
    if (!out) return fail(PT_ERR_INVALID, "output handle pointer is null");
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device_ordinal < 0 || device_ordinal >= n_dev)
        return fail(PT_ERR_INVALID, "device ordinal " + std::to_string(device_ordinal) + " out of range (" + std::to_string(n_dev) + " devices)");
    // the handle's stream first: everything below (tree build, uploads) already runs on it
    HIP_TRY(hipSetDevice(device_ordinal));
    hipStream_t stream = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard {  // on every failure below: no stream left behind, and the thread's current pair does not dangle
        hipStream_t st;
        ~StreamGuard() {
            if (!st) return;
            (void)hipStreamDestroy(st);
            g_stream = nullptr;
            g_device = -1;
        }
    } guard{stream};
    if (int rc = noalloc_address(device_ordinal)) return rc;
    g_device = device_ordinal;
    g_stream = stream;
    uint32_t depth = 0;
    PhaseClock clk;
    // a description without a tree: build the reference's tree here, on the GPU
    pt_scene_desc with_tree;
    struct Built {
        pt_bvh_node* nodes = nullptr;
        uint32_t* ids = nullptr;
        ~Built() {
            pt_hip_free(nodes);
            pt_hip_free(ids);
        }
    } built;
    if (desc && desc->n_nodes == 0 && !desc->nodes && !desc->prim_ids && desc->verts && desc->vi && desc->n_tris) {
        with_tree = *desc;
        if (int rc = pth::build_bvh_on_current(desc->verts, desc->n_verts, desc->vi, desc->n_tris, &built.nodes, &with_tree.n_nodes, &built.ids, nullptr))
            return rc;
        with_tree.nodes = built.nodes;
        with_tree.prim_ids = built.ids;
        desc = &with_tree;
        clk.lap("tree (pt_hip_build_bvh)");
    }
    if (int rc = validate(desc, depth)) return rc;
    clk.lap("validate");
    pt_hip_scene* s = new pt_hip_scene();
    s->depth = depth;
    s->device = device_ordinal;
    s->stream = stream;
    guard.st = nullptr;  // the handle owns it now
    if (int rc = build_scene(s, desc)) {
        pt_hip_scene_destroy(s);
        return rc;
    }
    *out = s;
    return PT_OK;
}

int pt_hip_scene_destroy(pt_hip_scene* s) {
    if (!s) return PT_OK;
    if (s->device >= 0) (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    ws_release(s->ws);
    for (hipEvent_t e : s->events.ev) (void)hipEventDestroy(e);
    if (s->batch_fetch) (void)hipFree(s->batch_fetch);
    for (void* p : s->stage)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : s->query_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (g_stream == s->stream) {  // the thread's current pair must not dangle
        g_stream = nullptr;
        g_device = -1;
    }
    delete s;
    return PT_OK;
}

// start / stop events of a batched query: created once per handle
static int query_events(pt_hip_scene* s) {
    for (hipEvent_t& e : s->query_ev)
        if (!e) HIP_TRY(hipEventCreate(&e));
    return PT_OK;
}

int pt_hip_intersect_device(pt_hip_scene* s, const pt_ray* d_rays, size_t n, pt_hit* d_hits, int mode, float* kernel_ms) {
    if (int rc = check_ready(s)) return rc;
    if (mode < 0 || mode > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return PT_OK;
    if (!d_rays || !d_hits || n > 0xffffffffull) return fail(PT_ERR_INVALID, "bad ray buffer");
    if (int rc = query_events(s)) return rc;
    HIP_TRY(hipEventRecord(s->query_ev[0], g_stream));
    if (int rc = launch_intersect(s, d_rays, n, d_hits, mode)) return rc;
    HIP_TRY(hipEventRecord(s->query_ev[1], g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->query_ev[0], s->query_ev[1]));
    if (kernel_ms) *kernel_ms = ms;
    return PT_OK;
}

int pt_hip_occluded_device(pt_hip_scene* s, const pt_ray* d_rays, size_t n, uint8_t* d_occ, int mode, float* kernel_ms) {
    if (int rc = check_ready(s)) return rc;
    if (mode < 0 || mode > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return PT_OK;
    if (!d_rays || !d_occ || n > 0xffffffffull) return fail(PT_ERR_INVALID, "bad ray buffer");
    if (int rc = query_events(s)) return rc;
    HIP_TRY(hipEventRecord(s->query_ev[0], g_stream));
    if (int rc = launch_occluded(s, d_rays, n, d_occ, nullptr, mode)) return rc;
    HIP_TRY(hipEventRecord(s->query_ev[1], g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->query_ev[0], s->query_ev[1]));
    if (kernel_ms) *kernel_ms = ms;
    return PT_OK;
}

int pt_hip_intersect_device_noalloc(pt_hip_scene* s, const pt_ray* d_rays, size_t n, pt_hit* d_hits, int mode) {
    if (int rc = check_ready(s)) return rc;
    if (mode < 0 || mode > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return PT_OK;
    if (!d_rays || !d_hits) return fail(PT_ERR_INVALID, "bad ray buffer");
    if (n > PT_NOALLOC_MAX_RAYS)
        return fail(PT_ERR_LIMIT, "pt_hip_intersect_device_noalloc takes at most " + std::to_string(PT_NOALLOC_MAX_RAYS) + " rays per call (" +
                                      std::to_string(n) + " given): split the batch or use pt_hip_intersect_device");
    uint32_t* fetch = g_device < kMaxDevices ? g_noalloc_fetch[g_device].load(std::memory_order_acquire) : nullptr;
    if (!fetch) return fail(PT_ERR_NOT_INIT, "no static work buffer on this device (device ordinal beyond " + std::to_string(kMaxDevices - 1) + ")");
    // Several handles may live on one GPU, each driven by its own host thread, and they all share this GPU's array: the
    // call is synchronous anyway, so it holds the GPU's lock from the clearing of the counters to the end of the kernels.
    std::lock_guard<std::mutex> lock(g_noalloc_call[g_device]);
    // the buffer keeps the previous call's counters: clear the part this call uses
    if (mode == PT_TRAVERSE_FAST) HIP_TRY(hipMemsetAsync(fetch, 0, size_t(blocks(uint32_t(n), kShard)) * sizeof(uint32_t), g_stream));
    if (int rc = launch_intersect(s, d_rays, n, d_hits, mode, fetch)) return rc;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_hip_intersect(pt_hip_scene* s, const pt_ray* rays, size_t n, pt_hit* hits, int mode) {
    if (int rc = check_ready(s)) return rc;
    if (mode < 0 || mode > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return PT_OK;
    if (!rays || !hits || n > 0xffffffffull) return fail(PT_ERR_INVALID, "bad ray buffer");
    pt_ray* d_rays = nullptr;
    pt_hit* d_hits = nullptr;
    if (int rc = stage_reserve(s, 0, n * sizeof(pt_ray), &d_rays)) return rc;
    if (int rc = stage_reserve(s, 1, n * sizeof(pt_hit), &d_hits)) return rc;
    HIP_TRY(hipMemcpyAsync(d_rays, rays, n * sizeof(pt_ray), hipMemcpyHostToDevice, g_stream));
    if (int rc = launch_intersect(s, d_rays, n, d_hits, mode)) return rc;
    HIP_TRY(hipMemcpyAsync(hits, d_hits, n * sizeof(pt_hit), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_hip_occluded(pt_hip_scene* s, const pt_ray* rays, size_t n, uint8_t* occluded, int32_t* first_prim, int mode) {
    if (int rc = check_ready(s)) return rc;
    if (mode < 0 || mode > 1) return fail(PT_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return PT_OK;
    if (!rays || !occluded || n > 0xffffffffull) return fail(PT_ERR_INVALID, "bad ray buffer");
    pt_ray* d_rays = nullptr;
    uint8_t* d_occ = nullptr;
    int32_t* d_first = nullptr;
    if (int rc = stage_reserve(s, 0, n * sizeof(pt_ray), &d_rays)) return rc;
    if (int rc = stage_reserve(s, 1, n * sizeof(int32_t), &d_first)) return rc;
    if (int rc = stage_reserve(s, 2, n, &d_occ)) return rc;
    HIP_TRY(hipMemcpyAsync(d_rays, rays, n * sizeof(pt_ray), hipMemcpyHostToDevice, g_stream));
    if (int rc = launch_occluded(s, d_rays, n, d_occ, d_first, mode)) return rc;
    HIP_TRY(hipMemcpyAsync(occluded, d_occ, n, hipMemcpyDeviceToHost, g_stream));
    if (first_prim) HIP_TRY(hipMemcpyAsync(first_prim, d_first, n * sizeof(int32_t), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_hip_render_device(pt_hip_scene* s, const pt_render_params* params, float* d_film, pt_stats* st) {
    if (int rc = check_ready(s)) return rc;
    pt_render_params p;
    if (int rc = normalise(params, p)) return rc;
    if (!d_film) return fail(PT_ERR_INVALID, "film pointer is null");
    const auto t0 = std::chrono::steady_clock::now();
    if (st) std::memset(st, 0, sizeof(*st));
    const uint32_t W = s->width, H = s->height;
    const uint32_t tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    const bool split = p.block_count > 1;
    if (split)
        if (int rc = prepare_split(s, p.block_rank, p.block_count, tiles_x, tiles_y)) return rc;
    const uint64_t slots_per_sample = uint64_t(split ? s->split_tiles : tiles_x * tiles_y) * 64;
    const uint64_t pixels = split ? s->split_pixels : uint64_t(W) * H;
    const uint32_t total_spp = p.spp_end - p.spp_begin;
    if (slots_per_sample == 0) {  // more ranks than blocks: nothing of this image is ours
        HIP_TRY(hipMemsetAsync(d_film, 0, sizeof(float) * 4 * size_t(W + 4) * size_t(H + 4), g_stream));
        HIP_TRY(hipStreamSynchronize(g_stream));
        return PT_OK;
    }
    uint32_t spp_pass = p.samples_per_pass;
    if (spp_pass == 0) {
        // Automatic: as many samples per pixel in flight as fit kAutoSlots path slots (224 B of queue space each).  Every
        // pass ends in a tail of a dozen nearly empty bounces whose cost does not depend on the pass size, so larger
        // passes are faster (measured at 1280x720: 8 M slots 1.5, 64 M 2.5, 128 M 2.7, 256 M 2.8 Grays/s on a 1.6 M
        // triangle interior; with the later kernels 128 M 3.5, 236 M 3.7, 472 M 3.8); 537 M slots are 120 GB of the 288 GB.
        // Less when the device is short of memory.
        const uint32_t sharing = std::max(1u, s->sharing);
        uint64_t slots = kAutoSlots / sharing;
        if (s->ws.cap_slots < slots) {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const uint64_t held = uint64_t(s->ws.cap_slots) * kBytesPerSlot;
            // handles that share the GPU all read the same free figure before any of them has allocated
            slots = std::min<uint64_t>(slots, (uint64_t(free_b) / sharing + held) * 6 / 10 / kBytesPerSlot);
        }
        spp_pass = uint32_t(std::min<uint64_t>(0xffffu, std::max<uint64_t>(1, slots / slots_per_sample)));  // 16-bit frame field of the path state
    }
    spp_pass = std::min(spp_pass, total_spp);
    if (p.samples_per_pass == 0) {  // automatic: passes of equal size instead of full ones and a small remainder
        const uint32_t n_pass = (total_spp + spp_pass - 1) / spp_pass;
        spp_pass = (total_spp + n_pass - 1) / n_pass;
    }
    if (slots_per_sample * spp_pass > 0x7fffffffull) return fail(PT_ERR_LIMIT, "samples_per_pass too large for this resolution");
    if (spp_pass > 0xffffu) return fail(PT_ERR_LIMIT, "samples_per_pass must be < 65536");
    for (;;) {
        const int rc = ws_reserve(s->ws, uint32_t(slots_per_sample * spp_pass), spp_pass);
        if (rc == PT_OK) break;
        // the automatic choice did not fit after all (other users of the GPU): smaller passes instead of giving up
        const bool out_of_memory = hipGetLastError() == hipErrorOutOfMemory || g_error.find(hipGetErrorString(hipErrorOutOfMemory)) != std::string::npos;
        if (p.samples_per_pass != 0 || spp_pass <= 1 || !out_of_memory) return rc;
        ws_release(s->ws);
        spp_pass = (spp_pass + 1) / 2;
    }
    Workspace& ws = s->ws;
    HIP_TRY(hipMemsetAsync(d_film, 0, sizeof(float) * 4 * size_t(W + 4) * size_t(H + 4), g_stream));
    s->events.used = 0;  // an earlier call that failed half-way may have left recorded pairs behind
    Timer tm{s, (p.flags & PT_FLAG_NO_KERNEL_TIMING) == 0};
    const bool count = (p.flags & PT_FLAG_COUNT_WORK) != 0;
    std::vector<uint32_t> frame_words;
    for (uint32_t f0 = p.spp_begin; f0 < p.spp_end; f0 += spp_pass) {
        const uint32_t nf = std::min(spp_pass, p.spp_end - f0);
        frame_words.resize(size_t(nf) * 1024);
        for (uint32_t k = 0; k < nf; ++k) {
            const uint64_t idx = frame_index(s->sob_m, uint64_t(f0) + k);
            sobol_words_all_dims(idx, frame_words.data() + size_t(k) * 1024);
        }
        HIP_TRY(hipMemcpyAsync(ws.frame_tab, frame_words.data(), frame_words.size() * 4, hipMemcpyHostToDevice, g_stream));
        PassArgs pa{};
        pa.frame_tab = ws.frame_tab;
        pa.spp_pass = nf;
        pa.tiles_x = tiles_x;
        pa.tiles_y = tiles_y;
        pa.n_slots = uint32_t(slots_per_sample * nf);
        pa.tile_list = split ? s->tile_list : nullptr;
        pa.tile_pos = split ? s->tile_pos : nullptr;
        pa.integrator = p.integrator;
        pa.traversal = s->reference_order_only ? int32_t(PT_TRAVERSE_EXACT) : p.traversal;
        pa.max_bounce = p.max_bounce;
        if (int rc = launch_generate(s, pa, tm, count)) return rc;
        if (st) {
            st->paths += pixels * nf;
            st->closest_rays += pixels * nf;
        }
        if (int rc = run_bounces(s, pa, st, tm, count)) return rc;
        if (int rc = tm.begin(PT_K_FILM)) return rc;
        hipLaunchKernelGGL(k_film, dim3(((tiles_x + 1) / 2) * ((tiles_y + 1) / 2)), dim3(256), 0, g_stream, s->dev, pa, ws.lbuf, d_film);
        if (int rc = tm.end()) return rc;
        HIP_TRY(hipMemcpyAsync(ws.h_misc(), ws.misc, kMiscWords * sizeof(uint32_t), hipMemcpyDeviceToHost, g_stream));
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipGetLastError());
        if (st) {
            st->exact_retraces += ws.h_misc()[0];
            for (int k = 0; k < 3; ++k) {
                st->nodes_fetched[k] += uint64_t(ws.h_misc()[4 + 4 * k]) | (uint64_t(ws.h_misc()[5 + 4 * k]) << 32);
                st->tris_fetched[k] += uint64_t(ws.h_misc()[6 + 4 * k]) | (uint64_t(ws.h_misc()[7 + 4 * k]) << 32);
                st->nodes_from_lds[k] += uint64_t(ws.h_misc()[20 + 2 * k]) | (uint64_t(ws.h_misc()[21 + 2 * k]) << 32);
            }
            st->passes += 1;
        }
        if (int rc = tm.collect(st)) return rc;
    }
    if (st) st->samples_per_pass = spp_pass;
    if (st) st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

int pt_hip_render(pt_hip_scene* s, const pt_render_params* params, float* film, pt_stats* st) {
    if (int rc = check_ready(s)) return rc;
    if (!film) return fail(PT_ERR_INVALID, "film pointer is null");
    const size_t bytes = sizeof(float) * 4 * size_t(s->width + 4) * size_t(s->height + 4);
    float* d_film = nullptr;
    if (int rc = stage_reserve(s, 3, bytes, &d_film)) return rc;
    if (int rc = pt_hip_render_device(s, params, d_film, st)) return rc;
    HIP_TRY(hipMemcpy(film, d_film, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

// adds b to a (n floats)
__global__ void k_add_film(float* a, const float* b, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

int pt_hip_render_multi(pt_hip_scene* const* scenes, int n, const pt_render_params* params, float* film, pt_stats* st) {
    if (!scenes || n <= 0) return fail(PT_ERR_INVALID, "no scene handles");
    if (!film) return fail(PT_ERR_INVALID, "film pointer is null");
    pt_render_params p;
    if (int rc = normalise(params, p)) return rc;
    for (int k = 0; k < n; ++k) {
        if (!scenes[k]) return fail(PT_ERR_INVALID, "scene handle is null");
        if (scenes[k]->width != scenes[0]->width || scenes[k]->height != scenes[0]->height) return fail(PT_ERR_INVALID, "handles of different image sizes");
        for (int j = 0; j < k; ++j)
            if (scenes[j] == scenes[k]) return fail(PT_ERR_INVALID, "the same handle twice");
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t floats = 4 * size_t(scenes[0]->width + 4) * size_t(scenes[0]->height + 4), bytes = floats * sizeof(float);
    // shares: contiguous ranges of sample numbers, or the 32x32 blocks of one rank each when the caller asked for blocks
    const bool by_blocks = p.block_count > 1;
    if (by_blocks && int(p.block_count) != n) return fail(PT_ERR_INVALID, "block_count must equal the number of handles");
    const uint32_t total = p.spp_end - p.spp_begin;
    std::vector<pt_render_params> share(n, p);
    std::vector<float*> d_film(n, nullptr);
    std::vector<pt_stats> stats(n);
    std::vector<int> rcs(n, PT_OK);
    std::vector<std::string> errs(n);
    for (int k = 0; k < n; ++k) {
        if (by_blocks) {
            share[k].block_rank = uint16_t(k);
        } else {
            const uint32_t base = total / uint32_t(n), rem = total % uint32_t(n);
            share[k].spp_begin = p.spp_begin + uint32_t(k) * base + std::min(uint32_t(k), rem);
            share[k].spp_end = share[k].spp_begin + base + (uint32_t(k) < rem ? 1u : 0u);
        }
        if (int rc = check_ready(scenes[k])) return rc;
        if (int rc = stage_reserve(scenes[k], 3, bytes, &d_film[k])) return rc;
    }
    // handles on the same GPU share its free memory when they choose their pass size
    for (int k = 0; k < n; ++k) {
        uint32_t same = 0;
        for (int j = 0; j < n; ++j) same += scenes[j]->device == scenes[k]->device ? 1u : 0u;
        scenes[k]->sharing = same;
    }
    // one host thread per handle, as the single-GPU entry point is driven
    std::vector<std::thread> th;
    for (int k = 0; k < n; ++k)
        th.emplace_back([&, k] {
            if (share[k].spp_end <= share[k].spp_begin) {  // more GPUs than samples: an empty film
                rcs[k] = check_ready(scenes[k]);
                if (!rcs[k] && (hipMemsetAsync(d_film[k], 0, bytes, g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess))
                    rcs[k] = PT_ERR_DEVICE;
                std::memset(&stats[k], 0, sizeof(pt_stats));
            } else {
                rcs[k] = pt_hip_render_device(scenes[k], &share[k], d_film[k], &stats[k]);
            }
            if (rcs[k]) errs[k] = g_error;
        });
    for (std::thread& t : th) t.join();
    for (int k = 0; k < n; ++k) scenes[k]->sharing = 1;
    for (int k = 0; k < n; ++k)
        if (rcs[k]) return fail(rcs[k], "GPU " + std::to_string(scenes[k]->device) + ": " + errs[k]);
    // the one reduction of the frame (src/block.cpp:102-110): films are summed on the first handle's GPU
    if (int rc = check_ready(scenes[0])) return rc;
    float* d_tmp = nullptr;
    if (n > 1)
        if (int rc = stage_reserve(scenes[0], 4, bytes, &d_tmp)) return rc;
    for (int k = 1; k < n; ++k) {
        HIP_TRY(hipMemcpyPeerAsync(d_tmp, scenes[0]->device, d_film[k], scenes[k]->device, bytes, g_stream));
        hipLaunchKernelGGL(k_add_film, dim3(unsigned((floats + 255) / 256)), dim3(256), 0, g_stream, d_film[0], d_tmp, floats);
    }
    HIP_TRY(hipMemcpyAsync(film, d_film[0], bytes, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    if (st) {
        std::memset(st, 0, sizeof(*st));
        for (int k = 0; k < n; ++k) {
            st->paths += stats[k].paths;
            st->closest_rays += stats[k].closest_rays;
            st->any_rays += stats[k].any_rays;
            st->exact_retraces += stats[k].exact_retraces;
            for (int j = 0; j < 3; ++j) {
                st->nodes_fetched[j] += stats[k].nodes_fetched[j];
                st->tris_fetched[j] += stats[k].tris_fetched[j];
                st->nodes_from_lds[j] += stats[k].nodes_from_lds[j];
                if (j == 0) st->shaded_hits += stats[k].shaded_hits;
            }
            for (int j = 0; j < PT_MAX_KERNEL_STATS; ++j) {
                st->kernel_ms[j] += stats[k].kernel_ms[j];
                st->kernel_launches[j] += stats[k].kernel_launches[j];
            }
        }
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return PT_OK;
}

int pt_hip_li(pt_hip_scene* s, const pt_render_params* params, const int32_t* pixel_xy, const uint32_t* sample_index, size_t n, float* rgb,
              float* pixel_sample, pt_stats* st) {
    if (int rc = check_ready(s)) return rc;
    pt_render_params p;
    pt_render_params tmp = params ? *params : pt_render_params{};
    tmp.spp_begin = 0;
    tmp.spp_end = 1;
    if (int rc = normalise(&tmp, p)) return rc;
    if (n == 0) return PT_OK;
    if (!pixel_xy || !sample_index || !rgb || n > 0x7fffffffull) return fail(PT_ERR_INVALID, "bad sample list");
    if (st) std::memset(st, 0, sizeof(*st));
    const auto t0 = std::chrono::steady_clock::now();
    std::map<uint32_t, uint32_t> local;
    std::vector<uint32_t> frame_local(n);
    for (size_t i = 0; i < n; ++i) {
        if (pixel_xy[2 * i] < 0 || pixel_xy[2 * i + 1] < 0 || uint32_t(pixel_xy[2 * i]) >= s->width || uint32_t(pixel_xy[2 * i + 1]) >= s->height)
            return fail(PT_ERR_INVALID, "pixel outside the image");
        auto it = local.emplace(sample_index[i], uint32_t(local.size())).first;
        frame_local[i] = it->second;
    }
    if (local.size() > 0xffffu) return fail(PT_ERR_LIMIT, "more than 65535 distinct sample indices in one call");
    std::vector<uint32_t> frame_words(local.size() * 1024);
    for (const auto& kv : local) {
        const uint64_t idx = frame_index(s->sob_m, kv.first);
        sobol_words_all_dims(idx, frame_words.data() + size_t(kv.second) * 1024);
    }
    if (int rc = ws_reserve(s->ws, uint32_t(n), local.size())) return rc;
    Workspace& ws = s->ws;
    int32_t* d_xy = nullptr;
    uint32_t* d_fl = nullptr;
    float* d_ps = nullptr;
    if (int r = stage_reserve(s, 0, n * 8, &d_xy)) return r;
    if (int r = stage_reserve(s, 1, n * 4, &d_fl)) return r;
    if (int r = stage_reserve(s, 2, n * 8, &d_ps)) return r;
    int rc = PT_OK;
    auto run = [&]() -> int {
        HIP_TRY(hipMemcpy(d_xy, pixel_xy, n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_fl, frame_local.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ws.frame_tab, frame_words.data(), frame_words.size() * 4, hipMemcpyHostToDevice));
        PassArgs pa{};
        pa.frame_tab = ws.frame_tab;
        pa.spp_pass = 1;
        pa.n_slots = uint32_t(n);
        pa.integrator = p.integrator;
        pa.traversal = s->reference_order_only ? int32_t(PT_TRAVERSE_EXACT) : p.traversal;
        pa.max_bounce = p.max_bounce;
        pa.list_xy = d_xy;
        pa.list_frame = d_fl;
        pa.list_pixel_sample = d_ps;
        s->events.used = 0;
        Timer tm{s, (p.flags & PT_FLAG_NO_KERNEL_TIMING) == 0};
        const bool count = (p.flags & PT_FLAG_COUNT_WORK) != 0;
        if (int r = launch_generate(s, pa, tm, count)) return r;
        if (st) {
            st->paths += n;
            st->closest_rays += n;
        }
        if (int r = run_bounces(s, pa, st, tm, count)) return r;
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipGetLastError());
        if (int r = tm.collect(st)) return r;
        std::vector<float4> L(n);
        HIP_TRY(hipMemcpy(L.data(), ws.lbuf, n * sizeof(float4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            rgb[3 * i] = L[i].x;
            rgb[3 * i + 1] = L[i].y;
            rgb[3 * i + 2] = L[i].z;
        }
        if (pixel_sample) HIP_TRY(hipMemcpy(pixel_sample, d_ps, n * 8, hipMemcpyDeviceToHost));
        return PT_OK;
    };
    rc = run();
    if (st) st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int pt_hip_selftest_division(uint64_t pairs, uint64_t seed, uint64_t* mismatches) {
    if (int rc = bind_default()) return rc;
    if (!mismatches) return fail(PT_ERR_INVALID, "null output");
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
    auto run = [&]() -> int {
        HIP_TRY(hipMemsetAsync(d, 0, sizeof(unsigned long long), g_stream));
        const uint32_t threads = 256 * 1024, per_thread = uint32_t(std::max<uint64_t>(1, (pairs + threads - 1) / threads));
        hipLaunchKernelGGL(k_div_selftest, dim3(threads / 256), dim3(256), 0, g_stream, seed, per_thread, d);
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipGetLastError());
        unsigned long long h = 0;
        HIP_TRY(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        *mismatches = h;
        return PT_OK;
    };
    const int rc = run();
    (void)hipFree(d);
    return rc;
}

int pt_hip_probe_box(const float* boxes6, const pt_ray* rays, size_t n, uint8_t* accepted) {
    if (int rc = bind_default()) return rc;
    if (n == 0) return PT_OK;
    if (!boxes6 || !rays || !accepted || n > 0x7fffffffull) return fail(PT_ERR_INVALID, "bad argument");
    ProbeBuffers pb;
    float* d_box = nullptr;
    pt_ray* d_ray = nullptr;
    uint8_t* d_out = nullptr;
    if (int rc = pb.in(boxes6, 6 * n, &d_box)) return rc;
    if (int rc = pb.in(rays, n, &d_ray)) return rc;
    if (int rc = pb.in<uint8_t>(nullptr, n, &d_out)) return rc;
    DevScene none{};  // only the "every ray takes the plain division" switch of make_pre is read: off
    hipLaunchKernelGGL(k_probe_box, dim3(blocks(uint32_t(n), 256)), dim3(256), 0, g_stream, none, d_box, reinterpret_cast<const float4*>(d_ray), uint32_t(n),
                       d_out);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(accepted, d_out, n, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_hip_probe_triangle(const float* tri9, const pt_ray* rays, size_t n, uint8_t* hit, float* tuv) {
    if (int rc = bind_default()) return rc;
    if (n == 0) return PT_OK;
    if (!tri9 || !rays || !hit || !tuv || n > 0x7fffffffull) return fail(PT_ERR_INVALID, "bad argument");
    ProbeBuffers pb;
    float *d_tri = nullptr, *d_tuv = nullptr;
    pt_ray* d_ray = nullptr;
    uint8_t* d_hit = nullptr;
    if (int rc = pb.in(tri9, 9 * n, &d_tri)) return rc;
    if (int rc = pb.in(rays, n, &d_ray)) return rc;
    if (int rc = pb.in<uint8_t>(nullptr, n, &d_hit)) return rc;
    if (int rc = pb.in<float>(nullptr, 3 * n, &d_tuv)) return rc;
    hipLaunchKernelGGL(k_probe_triangle, dim3(blocks(uint32_t(n), 256)), dim3(256), 0, g_stream, d_tri, reinterpret_cast<const float4*>(d_ray), uint32_t(n),
                       d_hit, d_tuv);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hit, d_hit, n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tuv, d_tuv, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_hip_probe_math(const float* x, const float* y, size_t n, float* sin_cos_pow) {
    if (int rc = bind_default()) return rc;
    if (n == 0) return PT_OK;
    if (!x || !y || !sin_cos_pow || n > 0x7fffffffull) return fail(PT_ERR_INVALID, "bad argument");
    ProbeBuffers pb;
    float *d_x = nullptr, *d_y = nullptr, *d_out = nullptr;
    if (int rc = pb.in(x, n, &d_x)) return rc;
    if (int rc = pb.in(y, n, &d_y)) return rc;
    if (int rc = pb.in<float>(nullptr, 3 * n, &d_out)) return rc;
    hipLaunchKernelGGL(k_probe_math, dim3(blocks(uint32_t(n), 256)), dim3(256), 0, g_stream, d_x, d_y, uint32_t(n), d_out);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(sin_cos_pow, d_out, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_hip_sobol(pt_hip_scene* s, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim, size_t n, float* out) {
    if (int rc = check_ready(s)) return rc;
    if (n == 0) return PT_OK;
    if (!pixel_xy || !sample_index || !dim || !out || n > 0x7fffffffull) return fail(PT_ERR_INVALID, "bad argument");
    std::map<uint32_t, uint32_t> local;
    std::vector<uint32_t> frame_local(n);
    for (size_t i = 0; i < n; ++i) {
        if (pixel_xy[2 * i] < 0 || pixel_xy[2 * i + 1] < 0 || uint32_t(pixel_xy[2 * i]) >= s->dev.sob_scale ||
            uint32_t(pixel_xy[2 * i + 1]) >= s->dev.sob_scale || dim[i] >= 1024)
            return fail(PT_ERR_INVALID, "pixel or dimension out of range");
        frame_local[i] = local.emplace(sample_index[i], uint32_t(local.size())).first->second;
    }
    std::vector<uint32_t> frame_words(local.size() * 1024);
    for (const auto& kv : local) {
        const uint64_t idx = frame_index(s->sob_m, kv.first);
        sobol_words_all_dims(idx, frame_words.data() + size_t(kv.second) * 1024);
    }
    void *d_tab = nullptr, *d_xy = nullptr, *d_fl = nullptr, *d_dim = nullptr, *d_out = nullptr;
    if (int r = stage_reserve(s, 0, n * 8, &d_xy)) return r;
    if (int r = stage_reserve(s, 1, n * 4, &d_fl)) return r;
    if (int r = stage_reserve(s, 2, n * 4, &d_dim)) return r;
    if (int r = stage_reserve(s, 4, n * 4, &d_out)) return r;
    if (int r = stage_reserve(s, 5, frame_words.size() * 4, &d_tab)) return r;
    auto run = [&]() -> int {
        HIP_TRY(hipMemcpy(d_tab, frame_words.data(), frame_words.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_xy, pixel_xy, n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_fl, frame_local.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_dim, dim, n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sobol, dim3(blocks(uint32_t(n), 256)), dim3(256), 0, g_stream, s->dev, static_cast<const uint32_t*>(d_tab),
                           static_cast<const int32_t*>(d_xy), static_cast<const uint32_t*>(d_fl), static_cast<const uint32_t*>(d_dim),
                           uint32_t(n), static_cast<float*>(d_out));
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
        return PT_OK;
    };
    return run();
}

}  // extern "C"

#ifdef PT_PHASE_STATS
// diagnostic builds only (not declared in the public header): reads, and optionally clears, the phase counters
extern "C" int pt_hip_debug_phase(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_phase), sizeof(ptd::g_phase)) != hipSuccess) return -1;
    if (reset) {
        static const unsigned long long zero[4][8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_phase), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef PT_NODE_HIST
// diagnostic builds only: reads, and optionally clears, the per-node visit counters
extern "C" int pt_hip_debug_node_hist(uint32_t* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_node_hist), sizeof(uint32_t) * ptd::kNodeHist) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ptd::g_node_hist)) != hipSuccess) return -1;
        if (hipMemset(p, 0, sizeof(uint32_t) * ptd::kNodeHist) != hipSuccess) return -1;
    }
    return 0;
}
#endif
