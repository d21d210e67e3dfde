// Sweep-SAH BVH builder: the host-side topology source for the GPU traversal kernels.
//
// Restates BVHTreeBuilder of the reference (src/bvh.cpp:99-234, include/pt/bvh.h:63-116) with 32-bit indices.  The
// topology (node numbering, leaf contents, child order) must be identical to the reference's because the reference
// traversal visits children in a fixed order and lets the last equally-near triangle win (src/bvh.cpp:39-53,
// src/shape.cpp:88).  Sorting uses std::sort / std::stable_partition like the reference so that ties between equal
// centroid coordinates resolve the way the same standard library resolves them there.
//
// The reference builds on one thread.  Here the three initial sorts and the subtrees run on several: a subtree only
// touches its own range of the three order arrays, and the reference's stack discipline finishes a whole subtree before
// it starts the sibling, so a subtree's nodes form one contiguous run of indices whose internal numbering does not
// depend on where the run starts.  Large ranges are split on the calling thread and their halves handed to tasks;
// ranges below a cut-off are built by the sequential loop into a locally numbered array; a final sequential pass
// replays the numbering over the small tree of large ranges and appends the local arrays with one offset each.  The
// result is identical, node for node, to the single-threaded build (tested against the oracle's builder).
#include "pt_host_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <future>
#include <limits>
#include <memory>
#include <numeric>
#include <thread>

namespace pt_host {

namespace {

struct Bounds {
    float lo[3], hi[3];
    Bounds() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();  // include/pt/aabb.h:10-13
            hi[k] = std::numeric_limits<float>::lowest();
        }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Bounds& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    // include/pt/aabb.h:83-88
    float halfArea() const {
        if (hi[0] < lo[0]) return 0.0f;
        const float w = hi[0] - lo[0], h = hi[1] - lo[1], d = hi[2] - lo[2];
        return (w + h) * d + w * h;
    }
};

class SweepBuilder {
public:
    SweepBuilder(const float* verts, const uint32_t* vi, uint32_t n) : n_(n), prim_(n), mark_(n), suffix_(n) {
        std::vector<float> centroid[3];
        for (auto& c : centroid) c.resize(n);
        for (uint32_t t = 0; t < n; ++t) {
            const float* a = verts + 3 * size_t(vi[3 * t + 0] - 1);
            const float* b = verts + 3 * size_t(vi[3 * t + 1] - 1);
            const float* c = verts + 3 * size_t(vi[3 * t + 2] - 1);
            Bounds bb;  // AABB(v0, v1, v2), include/pt/aabb.h:27-30
            for (int k = 0; k < 3; ++k) bb.lo[k] = bb.hi[k] = a[k];
            bb.grow(b);
            bb.grow(c);
            prim_[t] = bb;
            for (int k = 0; k < 3; ++k) centroid[k][t] = ((a[k] + b[k]) + c[k]) / 3.0f;  // src/shape.cpp:44-48
        }
        auto sort_axis = [&](int axis) {
            order_[axis].resize(n);
            std::iota(order_[axis].begin(), order_[axis].end(), uint32_t(0));
            const std::vector<float>& key = centroid[axis];
            std::sort(order_[axis].begin(), order_[axis].end(), [&key](uint32_t i, uint32_t j) { return key[i] < key[j]; });
        };
        if (n > kParallelCutoff) {
            std::thread t1(sort_axis, 1), t2(sort_axis, 2);
            sort_axis(0);
            t1.join();
            t2.join();
        } else {
            for (int axis = 0; axis < 3; ++axis) sort_axis(axis);
        }
    }

    void build(std::vector<pt_bvh_node>& nodes, std::vector<uint32_t>& prim_ids, uint32_t& max_depth) {
        max_depth = 0;
        nodes.clear();
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (n_ <= kParallelCutoff || hw == 1) {
            build_range(0, n_, 1, unite(0, n_), nodes, max_depth);
        } else {
            budget_.store(int(hw) - 1);
            cutoff_ = std::max<uint32_t>(kParallelCutoff / 4, n_ / (8 * hw));
            std::unique_ptr<Top> root = build_top(0, n_, 1, unite(0, n_));
            nodes.reserve(2 * size_t(n_));
            nodes.push_back(pt_bvh_node{});
            store(nodes[0], root->bounds);
            emit(*root, 0, nodes, max_depth);
        }
        prim_ids = order_[0];
        nodes.shrink_to_fit();
    }

private:
    static constexpr uint32_t kParallelCutoff = 1u << 16;  // ranges up to this many triangles are not worth a task

    // The reference's loop (src/bvh.cpp:99-166) on the range [begin, end): nodes[0] is the root of the range, children
    // are numbered in the order the reference's stack creates them.
    void build_range(uint32_t begin, uint32_t end, uint32_t depth, const Bounds& bounds, std::vector<pt_bvh_node>& nodes, uint32_t& max_depth) {
        struct Work {
            uint32_t node, begin, end, depth;
        };
        nodes.clear();
        nodes.reserve(2 * size_t(end - begin));
        nodes.push_back(pt_bvh_node{});
        store(nodes[0], bounds);
        std::vector<Work> todo{Work{0, begin, end, depth}};
        while (!todo.empty()) {
            const Work w = todo.back();
            todo.pop_back();
            max_depth = std::max(max_depth, w.depth);
            uint32_t cut = 0;
            if (w.end - w.begin > 1 && split(load(nodes[w.node]), w.begin, w.end, cut)) {
                const uint32_t child = uint32_t(nodes.size());
                nodes[w.node].first = child;
                nodes[w.node].count = 0;
                nodes.resize(size_t(child) + 2);
                Bounds left = unite(w.begin, cut), right = unite(cut, w.end);
                Work a{child, w.begin, cut, w.depth + 1}, b{child + 1, cut, w.end, w.depth + 1};
                if (left.halfArea() < right.halfArea()) {  // larger area first: src/bvh.cpp:137-144
                    std::swap(left, right);
                    std::swap(a.begin, b.begin);
                    std::swap(a.end, b.end);
                }
                store(nodes[child], left);
                store(nodes[child + 1], right);
                if (a.end - a.begin < b.end - b.begin) std::swap(a, b);  // bigger range is processed last: src/bvh.cpp:151-156
                todo.push_back(a);
                todo.push_back(b);
            } else {
                nodes[w.node].first = w.begin;
                nodes[w.node].count = w.end - w.begin;
            }
        }
    }

    // A large range: either split here (kid[0], kid[1] = the children in node-index order) or built sequentially (local).
    struct Top {
        uint32_t begin = 0, end = 0, depth = 0, local_depth = 0;
        Bounds bounds;
        bool is_split = false;
        std::unique_ptr<Top> kid[2];
        std::vector<pt_bvh_node> local;
    };

    std::unique_ptr<Top> build_top(uint32_t begin, uint32_t end, uint32_t depth, const Bounds& bounds) {
        std::unique_ptr<Top> t(new Top);
        t->begin = begin;
        t->end = end;
        t->depth = depth;
        t->bounds = bounds;
        uint32_t cut = 0;
        if (end - begin <= cutoff_ || !split(bounds, begin, end, cut)) {
            // small, or a leaf of more than cutoff_ triangles that the cost model refuses to split (cannot happen with
            // MaxLeafSize 8, but the sequential loop handles it either way)
            build_range(begin, end, depth, bounds, t->local, t->local_depth);
            return t;
        }
        t->is_split = true;
        Bounds b0 = unite(begin, cut), b1 = unite(cut, end);
        uint32_t r0[2] = {begin, cut}, r1[2] = {cut, end};
        if (b0.halfArea() < b1.halfArea()) {
            std::swap(b0, b1);
            std::swap(r0[0], r1[0]);
            std::swap(r0[1], r1[1]);
        }
        std::future<std::unique_ptr<Top>> other;
        const bool spawn = budget_.fetch_sub(1) > 0;
        if (spawn) {
            other = std::async(std::launch::async, [this, r1, depth, b1] {
                std::unique_ptr<Top> k = build_top(r1[0], r1[1], depth + 1, b1);
                budget_.fetch_add(1);
                return k;
            });
        } else {
            budget_.fetch_add(1);
        }
        t->kid[0] = build_top(r0[0], r0[1], depth + 1, b0);
        t->kid[1] = spawn ? other.get() : build_top(r1[0], r1[1], depth + 1, b1);
        return t;
    }

    // Replays the reference's numbering over the tree of large ranges; nodes[index] already holds this range's bounds.
    void emit(const Top& t, uint32_t index, std::vector<pt_bvh_node>& nodes, uint32_t& max_depth) const {
        max_depth = std::max(max_depth, t.depth);
        if (!t.is_split) {
            max_depth = std::max(max_depth, t.local_depth);
            // local[0] is this node; local[i >= 1] lands at base + i - 1
            const uint32_t base = uint32_t(nodes.size());
            auto fix = [base](pt_bvh_node n) {
                if (n.count == 0) n.first = base + n.first - 1;
                return n;
            };
            nodes[index] = fix(t.local[0]);
            nodes.resize(size_t(base) + t.local.size() - 1);
            for (size_t i = 1; i < t.local.size(); ++i) nodes[base + i - 1] = fix(t.local[i]);
            return;
        }
        const uint32_t child = uint32_t(nodes.size());
        nodes[index].first = child;
        nodes[index].count = 0;
        nodes.resize(size_t(child) + 2);
        store(nodes[child], t.kid[0]->bounds);
        store(nodes[child + 1], t.kid[1]->bounds);
        // the reference pushes the bigger range first, so the smaller one (kid[1] on a tie) is numbered first
        const uint32_t s0 = t.kid[0]->end - t.kid[0]->begin, s1 = t.kid[1]->end - t.kid[1]->begin;
        const int first = (s0 < s1) ? 0 : 1;
        emit(*t.kid[first], child + uint32_t(first), nodes, max_depth);
        emit(*t.kid[first ^ 1], child + uint32_t(first ^ 1), nodes, max_depth);
    }

    static void store(pt_bvh_node& n, const Bounds& b) {
        for (int k = 0; k < 3; ++k) {
            n.lo[k] = b.lo[k];
            n.hi[k] = b.hi[k];
        }
    }
    static Bounds load(const pt_bvh_node& n) {
        Bounds b;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = n.lo[k];
            b.hi[k] = n.hi[k];
        }
        return b;
    }
    Bounds unite(uint32_t begin, uint32_t end) const {  // include/pt/bvh.h:87-92 (always in axis-0 order)
        Bounds b;
        for (uint32_t i = begin; i < end; ++i) b.grow(prim_[order_[0][i]]);
        return b;
    }
    // include/pt/bvh.h:98-100: area * (count - 1.0f)
    static float partCost(uint32_t begin, uint32_t end, const Bounds& b) { return b.halfArea() * (float(end - begin) - 1.0f); }

    // src/bvh.cpp:201-234
    void sweep(int axis, uint32_t begin, uint32_t end, uint32_t& best_pos, float& best_cost, int& best_axis) {
        const std::vector<uint32_t>& ord = order_[axis];
        uint32_t stop = begin;
        Bounds acc;
        uint32_t i = end - 1;
        while (i > begin) {  // right-to-left in chunks of 32 with an early exit between chunks
            const uint32_t chunk_end = i - std::min(i - begin, uint32_t(32));
            float c = 0.0f;
            while (i > chunk_end) {
                acc.grow(prim_[ord[i]]);
                c = partCost(i, end, acc);
                suffix_[i] = c;
                --i;
            }
            if (c > best_cost) {
                stop = i;
                break;
            }
        }
        Bounds pre;
        for (uint32_t k = begin; k < stop; ++k) pre.grow(prim_[ord[k]]);
        for (uint32_t k = stop; k + 1 < end; ++k) {
            pre.grow(prim_[ord[k]]);
            const float lc = partCost(begin, k + 1, pre);
            const float total = lc + suffix_[k + 1];
            if (total < best_cost) {
                best_pos = k + 1;
                best_cost = total;
                best_axis = axis;
            } else if (lc > best_cost) {
                break;
            }
        }
    }

    // src/bvh.cpp:168-199
    bool split(const Bounds& bounds, uint32_t begin, uint32_t end, uint32_t& cut) {
        const float whole = bounds.halfArea() * float(end - begin);  // include/pt/bvh.h:94-96
        uint32_t pos = (begin + end + 1) / 2;
        float cost = whole;
        int axis = 0;
        for (int a = 0; a < 3; ++a) sweep(a, begin, end, pos, cost, axis);
        if (cost >= whole) {
            if (end - begin <= 8) return false;  // MaxLeafSize
            pos = (begin + end + 1) / 2;
            axis = 0;  // AABB::getMaxAxis() yields 0 whatever the extents are (include/pt/aabb.h:123-128)
        }
        for (uint32_t i = begin; i < pos; ++i) mark_[order_[axis][i]] = 1;
        for (uint32_t i = pos; i < end; ++i) mark_[order_[axis][i]] = 0;
        for (int a = 0; a < 3; ++a)
            if (a != axis)
                std::stable_partition(order_[a].begin() + begin, order_[a].begin() + end, [this](uint32_t p) { return mark_[p] != 0; });
        cut = pos;
        return true;
    }

    uint32_t n_;
    uint32_t cutoff_ = 0;
    std::atomic<int> budget_{0};
    std::vector<Bounds> prim_;
    std::vector<uint8_t> mark_;
    std::vector<float> suffix_;
    std::vector<uint32_t> order_[3];
};

}  // namespace

void build_bvh(const float* verts, const uint32_t* vi, uint32_t n_tris, std::vector<pt_bvh_node>& nodes,
               std::vector<uint32_t>& prim_ids, uint32_t& max_depth) {
    const auto t0 = std::chrono::steady_clock::now();
    SweepBuilder b(verts, vi, n_tris);
    const auto t1 = std::chrono::steady_clock::now();
    b.build(nodes, prim_ids, max_depth);
    const auto t2 = std::chrono::steady_clock::now();
    if (std::getenv("PT_HOST_BUILD_TIMING"))
        std::fprintf(stderr, "build_bvh: %u triangles, boxes + sorts %.2f s, tree %.2f s\n", n_tris, std::chrono::duration<double>(t1 - t0).count(),
                     std::chrono::duration<double>(t2 - t1).count());
}

}  // namespace pt_host
