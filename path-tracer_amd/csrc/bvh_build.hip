// Sweep-SAH BVH build on the GPU: pt_hip_build_bvh (include/pt_hip.h).
//
// Produces the reference's tree, node for node (BVHTreeBuilder, src/bvh.cpp:99-234, include/pt/bvh.h:63-116): same split
// positions, same leaf contents, same child order, same node numbering.  The reference builds top-down with an explicit
// stack, one range at a time: three sweeps over the range (one per axis, over index arrays sorted by centroid), the
// cheapest of all candidate positions wins, the two other index arrays are stably partitioned.  Nothing in that depends on
// the order in which the RANGES are processed -- only the node numbers do, and they can be replayed afterwards -- so here
// all ranges of one tree level are split at once, with segmented scans over the whole index arrays:
//
//   per level, per axis:   gather the triangle boxes in that axis' order
//                          segmented inclusive scan forwards  -> box of [begin, i]   (the sweep's left side)
//                          segmented inclusive scan backwards -> box of [i, end)     (its right side)
//                          cost of every candidate position, segmented arg-min (lowest position among equals)
//   per level:             per range: cheapest of the three axes (earlier axis among equals), leaf / split / median
//                          split decision, children's boxes, larger half-area first (src/bvh.cpp:137-144)
//                          marks, segmented stable partition of the three index arrays
//
// Equality with the sequential code: box unions are min / max and therefore exact in any order; a candidate's cost is the
// same expression on the same operands, halfArea * (count - 1.0f) per side (include/pt/bvh.h:98-100, compiled without
// contraction like the host builder); the reference's early exits (src/bvh.cpp:207-218, 226-229) only skip candidates
// whose one-sided cost already exceeds the best total so far -- costs grow monotonically with the box and the count --
// so they never skip a candidate that would have won, and its strict `<` keeps the first of equally cheap candidates in
// (axis, position) order, which is what the arg-min here returns.
//
// Node numbers: the reference numbers a node's children when it pops the node, and it finishes the smaller child's whole
// subtree first (src/bvh.cpp:151-156).  With cnt(X) = interior nodes in X's subtree: the children of X sit at alloc(X) and
// alloc(X) + 1, alloc(first-processed child) = alloc(X) + 2, alloc(the other) = alloc(X) + 2 + 2 cnt(first), alloc(root) = 1.
// Two sweeps over the levels (bottom-up for cnt, top-down for alloc) replay it.
//
// The three initial sorts stay on the host: the reference sorts with std::sort, whose order among equal centroid
// coordinates is a property of that algorithm (tessellated meshes are full of equal coordinates); sorting (key, index)
// pairs with the same comparator performs the same sequence of comparisons and moves, three axes on three threads.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "host_common.h"
#include "pt_hip.h"

namespace {

using pth::fail;

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                                           std::to_string(__LINE__) + ")");                                    \
    } while (0)

struct Box {
    float lo[3], hi[3];
};

// element of the segmented box scans: `head` starts a new segment
struct SegBox {
    Box b;
    uint32_t head;
    uint32_t pad;
};
struct SegBoxOp {
    __host__ __device__ SegBox operator()(const SegBox& a, const SegBox& c) const {
        if (c.head) return c;
        // AABB::operator+= (include/pt/aabb.h:40-44) with the running box on the left: std::min / std::max semantics, the
        // earlier operand stays on a tie -- which is what decides between -0 and +0
        SegBox r;
        for (int k = 0; k < 3; ++k) {
            r.b.lo[k] = (c.b.lo[k] < a.b.lo[k]) ? c.b.lo[k] : a.b.lo[k];
            r.b.hi[k] = (a.b.hi[k] < c.b.hi[k]) ? c.b.hi[k] : a.b.hi[k];
        }
        r.head = a.head;
        r.pad = 0;
        return r;
    }
};

// element of the segmented arg-min scan: the lowest position among equal costs
struct SegMin {
    float cost;
    uint32_t pos;
    uint32_t head;
};
struct SegMinOp {
    __host__ __device__ SegMin operator()(const SegMin& a, const SegMin& c) const {
        if (c.head) return c;
        SegMin r = (c.cost < a.cost) ? c : a;  // a comes first: it keeps ties
        r.head = a.head;
        return r;
    }
};

// element of the segmented count scan of the stable partition
struct SegCount {
    uint32_t n;
    uint32_t head;
};
struct SegCountOp {
    __host__ __device__ SegCount operator()(const SegCount& a, const SegCount& c) const {
        if (c.head) return c;
        return SegCount{a.n + c.n, a.head};
    }
};

// include/pt/aabb.h:83-88
__host__ __device__ inline float half_area(const Box& b) {
    if (b.hi[0] < b.lo[0]) return 0.0f;
    const float w = b.hi[0] - b.lo[0], h = b.hi[1] - b.lo[1], d = b.hi[2] - b.lo[2];
    return (w + h) * d + w * h;
}

// Nodes in build order (children of one node adjacent: kid, kid + 1 = larger half-area first).
struct BuildNodes {
    Box* box;
    uint32_t* begin;
    uint32_t* end;
    uint32_t* kid;     // build id of the first child, 0 = leaf or not split yet
    uint32_t* cut;     // split position of a node that is being split in this level
    uint32_t* axis;    // its axis
    uint32_t* swap;    // 1: the range [cut, end) became the first child
    uint32_t* cnt;     // interior nodes in the subtree
    uint32_t* alloc;   // final index of the node's first child
    uint32_t* index;   // final index of the node
};

constexpr int kThreads = 256;
inline unsigned grid_for(size_t n) { return unsigned((n + kThreads - 1) / kThreads); }

// boxes of axis `ord` gathered forwards (heads at range begins) and backwards (heads at range ends)
__global__ void k_gather(uint32_t n, const Box* prim, const uint32_t* ord, const uint32_t* owner, BuildNodes nd, SegBox* fwd, SegBox* bwd) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = owner[i];
    SegBox e;
    e.b = prim[ord[i]];
    e.pad = 0;
    e.head = (i == nd.begin[o]) ? 1u : 0u;
    fwd[i] = e;
    e.head = (i + 1u == nd.end[o]) ? 1u : 0u;
    bwd[n - 1u - i] = e;
}

// The box of every node of the current level as the reference computes it (computeAABB, include/pt/bvh.h:87-92): the union
// of its triangles in the order of the axis-0 index array, the running box on the left of every min / max -- i.e. the
// axis-0 forward scan at the range's last position.  (k_decide creates children with boxes taken from the split axis'
// scans; those agree in value but not necessarily in the sign of a zero coordinate.)
__global__ void k_boxes(uint32_t level_begin, uint32_t level_end, BuildNodes nd, const SegBox* pre0) {
    const uint32_t x = level_begin + blockIdx.x * kThreads + threadIdx.x;
    if (x >= level_end) return;
    if (nd.end[x] > nd.begin[x]) nd.box[x] = pre0[nd.end[x] - 1u].b;
}

// cost of splitting after position i (left = [begin, i], right = [i + 1, end)): src/bvh.cpp:219-229, bvh.h:98-100
__global__ void k_cost(uint32_t n, const uint32_t* owner, BuildNodes nd, uint32_t level_begin, const SegBox* pre, const SegBox* suf_rev, SegMin* out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = owner[i];
    const uint32_t b = nd.begin[o], e = nd.end[o];
    SegMin m;
    m.cost = __builtin_huge_valf();
    m.pos = i + 1u;
    m.head = (i == b) ? 1u : 0u;
    if (o >= level_begin && e - b > 1u && i + 1u < e) {
        const float lc = half_area(pre[i].b) * (float(i + 1u - b) - 1.0f);
        const float rc = half_area(suf_rev[n - 2u - i].b) * (float(e - (i + 1u)) - 1.0f);  // suffix box of [i + 1, end)
        m.cost = lc + rc;
    }
    out[i] = m;
}

// Per node of the current level: leaf, split at the cheapest candidate, or median split (src/bvh.cpp:168-199); children
// are created with their boxes, larger half-area first (src/bvh.cpp:137-144).
__global__ void k_decide(uint32_t level_begin, uint32_t level_end, BuildNodes nd, const SegMin* m0, const SegMin* m1, const SegMin* m2,
                         const SegBox* p0, const SegBox* p1, const SegBox* p2, const SegBox* s0, const SegBox* s1, const SegBox* s2, uint32_t n,
                         uint32_t* n_nodes) {
    const uint32_t x = level_begin + blockIdx.x * kThreads + threadIdx.x;
    if (x >= level_end) return;
    const uint32_t b = nd.begin[x], e = nd.end[x];
    nd.kid[x] = 0u;
    nd.cut[x] = 0u;
    if (e - b <= 1u) return;  // src/bvh.cpp:125
    const float whole = half_area(nd.box[x]) * float(e - b);  // include/pt/bvh.h:94-96
    uint32_t pos = (b + e + 1u) / 2u, axis = 0u;
    float cost = whole;
    const SegMin* ms[3] = {m0, m1, m2};
    for (uint32_t a = 0; a < 3u; ++a) {
        const SegMin m = ms[a][e - 1u];  // inclusive scan: the range's result sits at its last position
        if (m.cost < cost) {
            cost = m.cost;
            pos = m.pos;
            axis = a;
        }
    }
    if (cost >= whole) {
        if (e - b <= 8u) return;  // MaxLeafSize
        pos = (b + e + 1u) / 2u;
        axis = 0u;  // AABB::getMaxAxis() yields 0 whatever the extents are (include/pt/aabb.h:123-128)
    }
    const SegBox* pre = axis == 0u ? p0 : (axis == 1u ? p1 : p2);
    const SegBox* suf = axis == 0u ? s0 : (axis == 1u ? s1 : s2);
    Box left = pre[pos - 1u].b, right = suf[n - 1u - pos].b;
    const bool sw = half_area(left) < half_area(right);
    const uint32_t kid = atomicAdd(n_nodes, 2u);
    nd.kid[x] = kid;
    nd.cut[x] = pos;
    nd.axis[x] = axis;
    nd.swap[x] = sw ? 1u : 0u;
    nd.box[kid] = sw ? right : left;
    nd.box[kid + 1u] = sw ? left : right;
    nd.begin[kid] = sw ? pos : b;
    nd.end[kid] = sw ? e : pos;
    nd.begin[kid + 1u] = sw ? b : pos;
    nd.end[kid + 1u] = sw ? pos : e;
    nd.kid[kid] = nd.kid[kid + 1u] = 0u;
    nd.cut[kid] = nd.cut[kid + 1u] = 0u;
}

// include/pt/bvh.h:102-105: triangles left of the cut in the split axis' order are marked
__global__ void k_mark(uint32_t n, const uint32_t* owner, BuildNodes nd, const uint32_t* o0, const uint32_t* o1, const uint32_t* o2, uint8_t* mark) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = owner[i], cut = nd.cut[x];
    if (cut == 0u) return;
    const uint32_t a = nd.axis[x];
    const uint32_t prim = (a == 0u ? o0 : (a == 1u ? o1 : o2))[i];
    mark[prim] = (i < cut) ? 1 : 0;
}

// stable partition, step 1: 1 for elements that stay in front (every element of a range that is not being split)
__global__ void k_flags(uint32_t n, const uint32_t* owner, BuildNodes nd, const uint32_t* ord, const uint8_t* mark, SegCount* out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = owner[i];
    const bool front = nd.cut[x] == 0u || mark[ord[i]] != 0;
    out[i] = SegCount{front ? 1u : 0u, (i == nd.begin[x]) ? 1u : 0u};
}

// step 2: scatter (inclusive counts of the front elements)
__global__ void k_scatter(uint32_t n, const uint32_t* owner, BuildNodes nd, const uint32_t* ord, const uint8_t* mark, const SegCount* inc,
                          uint32_t* out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = owner[i], cut = nd.cut[x];
    uint32_t dst = i;
    if (cut != 0u) {
        const uint32_t b = nd.begin[x];
        const bool front = mark[ord[i]] != 0;
        const uint32_t before = inc[i].n - (front ? 1u : 0u);  // front elements before i in this range
        dst = front ? b + before : cut + (i - b - before);
    }
    out[dst] = ord[i];
}

__global__ void k_owner(uint32_t n, uint32_t* owner, BuildNodes nd) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = owner[i], cut = nd.cut[x];
    if (cut == 0u) return;
    const bool left = i < cut;
    owner[i] = nd.kid[x] + ((left != (nd.swap[x] != 0u)) ? 0u : 1u);  // swapped: the left range is the second child
}

// bottom-up: interior nodes per subtree
__global__ void k_count(uint32_t level_begin, uint32_t level_end, BuildNodes nd) {
    const uint32_t x = level_begin + blockIdx.x * kThreads + threadIdx.x;
    if (x >= level_end) return;
    const uint32_t k = nd.kid[x];
    nd.cnt[x] = k ? 1u + nd.cnt[k] + nd.cnt[k + 1u] : 0u;
}

// top-down: final numbers (the smaller range is finished first; on equal sizes the second child, src/bvh.cpp:151-156)
__global__ void k_number(uint32_t level_begin, uint32_t level_end, BuildNodes nd) {
    const uint32_t x = level_begin + blockIdx.x * kThreads + threadIdx.x;
    if (x >= level_end) return;
    const uint32_t k = nd.kid[x];
    if (!k) return;
    const uint32_t a = nd.alloc[x];
    nd.index[k] = a;
    nd.index[k + 1u] = a + 1u;
    const uint32_t s0 = nd.end[k] - nd.begin[k], s1 = nd.end[k + 1u] - nd.begin[k + 1u];
    const uint32_t first = (s0 < s1) ? k : k + 1u, second = (s0 < s1) ? k + 1u : k;
    nd.alloc[first] = a + 2u;
    nd.alloc[second] = a + 2u + 2u * nd.cnt[first];
}

__global__ void k_emit(uint32_t n_nodes, BuildNodes nd, pt_bvh_node* out) {
    const uint32_t x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= n_nodes) return;
    pt_bvh_node o;
    for (int k = 0; k < 3; ++k) {
        o.lo[k] = nd.box[x].lo[k];
        o.hi[k] = nd.box[x].hi[k];
    }
    const bool interior = nd.kid[x] != 0u;
    o.first = interior ? nd.alloc[x] : nd.begin[x];
    o.count = interior ? 0u : nd.end[x] - nd.begin[x];
    out[nd.index[x]] = o;
}

struct DeviceArena {
    std::vector<void*> owned;
    ~DeviceArena() {
        for (void* p : owned) (void)hipFree(p);
    }
    template <typename T>
    int get(T** p, size_t count) {
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
        owned.push_back(d);
        *p = static_cast<T*>(d);
        return PT_OK;
    }
};

struct KeyIdx {
    float key;
    uint32_t idx;
};

}  // namespace

int pth::build_bvh_on_current(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes_out, uint32_t* n_nodes_out,
                              uint32_t** prim_ids_out, uint32_t* max_depth_out) {
    if (pth::device() < 0) return fail(PT_ERR_NOT_INIT, "pt_hip_init has not been called");
    if (!verts || !vi || !nodes_out || !n_nodes_out || !prim_ids_out || n_tris == 0) return fail(PT_ERR_INVALID, "null argument or empty mesh");
    if (n_tris > 0x3fffffffu) return fail(PT_ERR_LIMIT, "too many triangles");
    for (size_t i = 0; i < 3 * size_t(n_tris); ++i)
        if (vi[i] == 0 || vi[i] > n_verts) return fail(PT_ERR_INVALID, "vertex index out of range");
    const bool timing = std::getenv("PT_HIP_BUILD_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t stream = pth::stream();
    const uint32_t n = n_tris;

    // triangle boxes and centroids (include/pt/aabb.h:27-30, src/shape.cpp:44-48), the three sorts (src/bvh.cpp:105-111)
    std::vector<Box> prim(n);
    std::vector<KeyIdx> keys[3];
    for (auto& k : keys) k.resize(n);
    for (uint32_t t = 0; t < n; ++t) {
        const float* a = verts + 3 * size_t(vi[3 * size_t(t) + 0] - 1);
        const float* b = verts + 3 * size_t(vi[3 * size_t(t) + 1] - 1);
        const float* c = verts + 3 * size_t(vi[3 * size_t(t) + 2] - 1);
        for (int k = 0; k < 3; ++k) {
            prim[t].lo[k] = std::min(std::min(a[k], b[k]), c[k]);
            prim[t].hi[k] = std::max(std::max(a[k], b[k]), c[k]);
            keys[k][t] = KeyIdx{((a[k] + b[k]) + c[k]) / 3.0f, t};
        }
    }
    auto sort_axis = [&keys](int axis) { std::sort(keys[axis].begin(), keys[axis].end(), [](const KeyIdx& p, const KeyIdx& q) { return p.key < q.key; }); };
    {
        std::thread t1(sort_axis, 1), t2(sort_axis, 2);
        sort_axis(0);
        t1.join();
        t2.join();
    }
    const auto t1 = std::chrono::steady_clock::now();

    DeviceArena mem;
    Box* d_prim = nullptr;
    uint32_t *d_ord[3] = {}, *d_tmp = nullptr, *d_owner = nullptr, *d_n_nodes = nullptr;
    uint8_t* d_mark = nullptr;
    SegBox *d_fwd = nullptr, *d_bwd = nullptr, *d_pre[3] = {}, *d_suf[3] = {};
    SegMin *d_cost = nullptr, *d_min[3] = {};
    SegCount *d_flag = nullptr, *d_inc = nullptr;
    pt_bvh_node* d_out = nullptr;
    BuildNodes nd{};
    const size_t cap = 2 * size_t(n);
    if (int rc = mem.get(&d_prim, n)) return rc;
    for (int a = 0; a < 3; ++a) {
        if (int rc = mem.get(&d_ord[a], n)) return rc;
        if (int rc = mem.get(&d_pre[a], n)) return rc;
        if (int rc = mem.get(&d_suf[a], n)) return rc;
        if (int rc = mem.get(&d_min[a], n)) return rc;
    }
    if (int rc = mem.get(&d_tmp, n)) return rc;
    if (int rc = mem.get(&d_owner, n)) return rc;
    if (int rc = mem.get(&d_n_nodes, 1)) return rc;
    if (int rc = mem.get(&d_mark, n)) return rc;
    if (int rc = mem.get(&d_fwd, n)) return rc;
    if (int rc = mem.get(&d_bwd, n)) return rc;
    if (int rc = mem.get(&d_cost, n)) return rc;
    if (int rc = mem.get(&d_flag, n)) return rc;
    if (int rc = mem.get(&d_inc, n)) return rc;
    if (int rc = mem.get(&d_out, cap)) return rc;
    if (int rc = mem.get(&nd.box, cap)) return rc;
    for (uint32_t** p : {&nd.begin, &nd.end, &nd.kid, &nd.cut, &nd.axis, &nd.swap, &nd.cnt, &nd.alloc, &nd.index})
        if (int rc = mem.get(p, cap)) return rc;

    // scan scratch: the largest of the three element types decides
    size_t tmp_bytes = 0, need = 0;
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, need, d_fwd, d_pre[0], SegBoxOp(), int(n), stream);
    tmp_bytes = std::max(tmp_bytes, need);
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, need, d_cost, d_min[0], SegMinOp(), int(n), stream);
    tmp_bytes = std::max(tmp_bytes, need);
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, need, d_flag, d_inc, SegCountOp(), int(n), stream);
    tmp_bytes = std::max(tmp_bytes, need);
    uint8_t* d_scan_tmp = nullptr;
    if (int rc = mem.get(&d_scan_tmp, tmp_bytes)) return rc;

    HIP_TRY(hipMemcpyAsync(d_prim, prim.data(), size_t(n) * sizeof(Box), hipMemcpyHostToDevice, stream));
    {
        std::vector<uint32_t> ord(n);
        for (int a = 0; a < 3; ++a) {
            for (uint32_t i = 0; i < n; ++i) ord[i] = keys[a][i].idx;
            HIP_TRY(hipMemcpyAsync(d_ord[a], ord.data(), size_t(n) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
    }
    // root: build id 0 over [0, n), its box = union of all triangles (include/pt/bvh.h:87-92)
    {
        Box root = prim[0];
        for (uint32_t t = 1; t < n; ++t)
            for (int k = 0; k < 3; ++k) {
                root.lo[k] = std::min(root.lo[k], prim[t].lo[k]);
                root.hi[k] = std::max(root.hi[k], prim[t].hi[k]);
            }
        const uint32_t zero = 0, one = 1;
        HIP_TRY(hipMemcpyAsync(nd.box, &root, sizeof(Box), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(nd.begin, &zero, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(nd.end, &n, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(nd.kid, &zero, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(nd.cut, &zero, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_n_nodes, &one, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_owner, 0, size_t(n) * 4, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }

    std::vector<uint32_t> level_start{0u};  // build ids of level L: [level_start[L], level_start[L + 1])
    uint32_t level_begin = 0, level_end = 1;
    while (level_end > level_begin) {
        for (int a = 0; a < 3; ++a) {
            hipLaunchKernelGGL(k_gather, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_prim, d_ord[a], d_owner, nd, d_fwd, d_bwd);
            HIP_TRY(hipcub::DeviceScan::InclusiveScan(d_scan_tmp, tmp_bytes, d_fwd, d_pre[a], SegBoxOp(), int(n), stream));
            HIP_TRY(hipcub::DeviceScan::InclusiveScan(d_scan_tmp, tmp_bytes, d_bwd, d_suf[a], SegBoxOp(), int(n), stream));
            if (a == 0) hipLaunchKernelGGL(k_boxes, dim3(grid_for(level_end - level_begin)), dim3(kThreads), 0, stream, level_begin, level_end, nd, d_pre[0]);
            hipLaunchKernelGGL(k_cost, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_owner, nd, level_begin, d_pre[a], d_suf[a], d_cost);
            HIP_TRY(hipcub::DeviceScan::InclusiveScan(d_scan_tmp, tmp_bytes, d_cost, d_min[a], SegMinOp(), int(n), stream));
        }
        hipLaunchKernelGGL(k_decide, dim3(grid_for(level_end - level_begin)), dim3(kThreads), 0, stream, level_begin, level_end, nd, d_min[0], d_min[1],
                           d_min[2], d_pre[0], d_pre[1], d_pre[2], d_suf[0], d_suf[1], d_suf[2], n, d_n_nodes);
        hipLaunchKernelGGL(k_mark, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_owner, nd, d_ord[0], d_ord[1], d_ord[2], d_mark);
        for (int a = 0; a < 3; ++a) {
            hipLaunchKernelGGL(k_flags, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_owner, nd, d_ord[a], d_mark, d_flag);
            HIP_TRY(hipcub::DeviceScan::InclusiveScan(d_scan_tmp, tmp_bytes, d_flag, d_inc, SegCountOp(), int(n), stream));
            hipLaunchKernelGGL(k_scatter, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_owner, nd, d_ord[a], d_mark, d_inc, d_tmp);
            std::swap(d_ord[a], d_tmp);
        }
        hipLaunchKernelGGL(k_owner, dim3(grid_for(n)), dim3(kThreads), 0, stream, n, d_owner, nd);
        uint32_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, d_n_nodes, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        level_begin = level_end;
        level_end = total;
        level_start.push_back(level_begin);
        if (level_start.size() > 4096) return fail(PT_ERR_LIMIT, "tree deeper than 4096 levels");
    }
    const uint32_t n_nodes = level_end;
    const uint32_t levels = uint32_t(level_start.size()) - 1;  // the last entry opens an empty level
    // numbering: counts bottom-up, numbers top-down
    for (uint32_t l = levels; l-- > 0;) {
        const uint32_t b = level_start[l], e = level_start[l + 1];
        hipLaunchKernelGGL(k_count, dim3(grid_for(e - b)), dim3(kThreads), 0, stream, b, e, nd);
    }
    {
        const uint32_t zero = 0, one = 1;
        HIP_TRY(hipMemcpyAsync(nd.index, &zero, 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(nd.alloc, &one, 4, hipMemcpyHostToDevice, stream));
    }
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t b = level_start[l], e = level_start[l + 1];
        hipLaunchKernelGGL(k_number, dim3(grid_for(e - b)), dim3(kThreads), 0, stream, b, e, nd);
    }
    hipLaunchKernelGGL(k_emit, dim3(grid_for(n_nodes)), dim3(kThreads), 0, stream, n_nodes, nd, d_out);
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    const auto t2 = std::chrono::steady_clock::now();

    pt_bvh_node* nodes = static_cast<pt_bvh_node*>(std::malloc(size_t(n_nodes) * sizeof(pt_bvh_node)));
    uint32_t* ids = static_cast<uint32_t*>(std::malloc(size_t(n) * sizeof(uint32_t)));
    if (!nodes || !ids) {
        std::free(nodes);
        std::free(ids);
        return fail(PT_ERR_DEVICE, "out of host memory for the tree");
    }
    if (hipMemcpy(nodes, d_out, size_t(n_nodes) * sizeof(pt_bvh_node), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(ids, d_ord[0], size_t(n) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
        std::free(nodes);
        std::free(ids);
        return fail(PT_ERR_DEVICE, "tree download failed");
    }
    *nodes_out = nodes;
    *n_nodes_out = n_nodes;
    *prim_ids_out = ids;
    if (max_depth_out) *max_depth_out = levels;
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        auto s = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        std::fprintf(stderr, "pt_hip_build_bvh: %u triangles, %u nodes, %u levels: boxes + sorts %.2f s, device build %.2f s, download %.2f s\n", n, n_nodes,
                     levels, s(t0, t1), s(t1, t2), s(t2, t3));
    }
    return PT_OK;
}

extern "C" {

void pt_hip_free(void* p) { std::free(p); }

int pt_hip_build_bvh(const float* verts, uint32_t n_verts, const uint32_t* vi, uint32_t n_tris, pt_bvh_node** nodes, uint32_t* n_nodes, uint32_t** prim_ids,
                     uint32_t* max_depth) {
    if (int rc = pth::bind_default()) return rc;
    return pth::build_bvh_on_current(verts, n_verts, vi, n_tris, nodes, n_nodes, prim_ids, max_depth);
}

}  // extern "C"
