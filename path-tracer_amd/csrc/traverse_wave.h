// Ordered traversal restructured for wave64 occupancy: a lane's query is advanced one node at a time by a loop that
// the whole wave shares, so that
//   - box steps and triangle steps run in separate phases (all lanes of a phase execute the same code), and
//   - lanes whose query has ended take the next ray from the queue as soon as too few lanes are left busy, instead of
//     idling until the slowest lane of the wave is done.
// The ordered lanes (ClosestLane, AnyLane) navigate with the conservative slab test box_cheap and take the reference's
// exact box decision once per query, on the parent box of the leaf that holds the result (why that is enough:
// traverse.h, above box_cheap; why it can wait until the walk has ended: ClosestLane::leaf);
// candidates are the triangles the reference would accept, and two candidates (or a candidate and the ray's far end)
// within the kNearTie window of each other are flagged, because then the reference's visit order decides which one it
// keeps.  Deferred subtrees that have fallen behind the best hit by the time they are popped are dropped without a fetch
// (DistStack).  Rays that need the plain-division slab test ("special") and queries whose result could depend on the
// visit order are not handled here; they are handed back to the caller (`defer`), which traces them in reference order
// (closest_exact / closest_exact_t with a distance bound / any_exact) in a follow-up step.
#pragma once

#include "traverse.h"

namespace ptd {

constexpr uint32_t kDone = kLeafBit;     // a leaf reference without triangles: nothing left to visit
// Measured on a 1.6 M triangle interior scene: flat optimum around these values (refill 20..48, box lanes 12..40: within 3 %).
#ifndef PT_REFILL_BELOW
#define PT_REFILL_BELOW 24
#endif
#ifndef PT_MIN_BOX_LANES
#define PT_MIN_BOX_LANES 20
#endif
constexpr uint32_t kRefillBelow = PT_REFILL_BELOW;   // fetch new rays once fewer lanes than this are busy
constexpr uint32_t kMinBoxLanes = PT_MIN_BOX_LANES;  // leave the box phase for the triangle phase once fewer lanes than this have a box step to do

// Deferred-ray marker "trace this ray with the plain reference-order routine" (otherwise a deferred ray carries a distance
// bound >= 0 up to which the reference order is replayed).
constexpr float kDeferExact = -1.0f;

PTD uint32_t lane_id() { return __lane_id(); }
// Ballots straight from the condition (the runtime's __ballot / __any take an int: a select and a compare per call).
PTD uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
PTD bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
PTD uint32_t wave_count(bool p) { return uint32_t(__builtin_popcountll(__builtin_amdgcn_ballot_w64(p))); }

// Wave-level queue allocation: one atomic per wave, ballot + prefix popcount for the lane offset.
PTD uint32_t wave_alloc(bool want, uint32_t* counter) {
    const uint64_t mask = wave_ballot(want);
    const uint32_t total = uint32_t(__popcll(mask));
    uint32_t base = 0;
    if (total != 0u) {
        const int leader = __ffsll((unsigned long long)mask) - 1;
        if (int(lane_id()) == leader) base = atomicAdd(counter, total);
        base = uint32_t(__shfl(int(base), leader));
    }
    return base + uint32_t(__popcll(mask & ((1ull << lane_id()) - 1ull)));
}

// Diagnostic builds (-DPT_PHASE_STATS) count, per kernel class, how many lanes take part in each phase of the shared
// loops: [0] box steps of a wave, [1] lanes in them, [2] leaf phases, [3] lanes in them, [4] triangle-loop rounds,
// [5] lane-triangles, [6] commit/fetch rounds, [7] lanes committed.  Compiles to nothing otherwise.
#ifdef PT_PHASE_STATS
__device__ unsigned long long g_phase[4][8];
struct PhaseStats {
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PTD void box(uint32_t lanes) {
        c[0] += 1;
        c[1] += lanes;
    }
    PTD void leaf(bool active, uint32_t cur) {
        const uint32_t count = active ? ((cur >> kLeafCountShift) & 15u) : 0u;
        c[2] += 1;
        c[3] += uint32_t(__popcll(__ballot(active)));
        for (uint32_t k = 0; k < 15u; ++k) {
            const uint32_t n = uint32_t(__popcll(__ballot(count > k)));
            c[4] += n != 0u;
            c[5] += n;
        }
    }
    PTD void round(bool fin) {
        c[6] += 1;
        c[7] += uint32_t(__popcll(__ballot(fin)));
    }
    PTD void flush(int cls) {
        if (lane_id() == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_phase[cls][i], c[i]);
    }
};
#else
struct PhaseStats {
    PTD void box(uint32_t) {}
    PTD void leaf(bool, uint32_t) {}
    PTD void round(bool) {}
    PTD void flush(int) {}
};
#endif

// Lane interface of the shared loops (closest_wave / any_wave):
//   StackT, kSpillWords            the lane's stack type and the private words it needs
//   idle()                         put the lane into the "nothing to do" state
//   begin(sc, ray, pre)            start a query
//   box_ready() / leaf_ready()     the next step is a box step / a triangle step
//   finished()                     the query has ended
//   kTreeletNodes                  navigation nodes the lane type wants in LDS (0: none)
//   inner(sc, tl, st, wc), leaf(sc, st, wc)
//   settle(sc, [bound,] wc)        once per query when the walk has ended: true = repeat it in reference order
// per-ray constants of the navigation test, by node width
template <int W>
struct NavRayOf {
    typedef CheapRay type;
};
template <>
struct NavRayOf<8> {
    typedef CheapRay8 type;
};
typedef NavRayOf<kWide>::type NavRay;

struct ClosestLane {
    typedef DistStack StackT;
    static constexpr int kSpillWords = 2 * DistStack::kSpill;
    static constexpr int kLdsRows = 2 * DistStack::kRows;
    static constexpr bool kReferenceOrder = false;
    static constexpr uint32_t kTreeletNodes = kTreelet;
    PTD void idle() {
        cur = kDone;
        hit = HitD{-1, 0.0f, 0.0f, 0.0f};
    }
    PTD bool box_ready() const { return !(cur >> 31); }
    PTD bool leaf_ready() const { return cur > kDone; }
    PTD bool finished() const { return cur == kDone; }
    PTD const V3& org() const { return o; }
    PTD const V3& dir() const { return d; }
    V3 o, d;
    NavRay cr;
    float tmin, tmax;
    float window;       // candidates are accepted up to here: the best hit's distance widened by kNearTie, or the ray's far end
    bool near_tie;
    uint32_t hit_leaf;  // header slot of the leaf that holds the current winner
    HitD hit;
    uint32_t cur;

    PTD void begin(const DevScene& sc, const RayD& ray, const RayPre& p) {
        o = p.o;
        d = p.d;
        tmin = p.tmin;
        cr.set(sc, p);
        tmax = ray.tmax;
        window = ray.tmax;
        near_tie = false;
        hit_leaf = 0u;
        hit = HitD{-1, 0.0f, 0.0f, 0.0f};
        cur = sc.wide_root;
    }
    // Boxes are culled against the acceptance bound widened by kNearTie once more: a triangle's computed distance may lie
    // before the entry distance of the (flat) box around it by rounding, and at the ray's far end there is no other
    // margin (a ray that ends on a surface it grazes: the plane distance of the box can come out a few ulps beyond the
    // far end while the triangle's distance is accepted).
    PTD float cull_bound() const { return window * (1.0f + kNearTie); }
    // the nearest deferred subtree that can still hold a candidate
    PTD void next(DistStack& st) {
        cur = kDone;
        const float far = cull_bound();
        while (!st.empty()) {
            uint32_t ref;
            float entry;
            st.pop(ref, entry);
            if (entry <= far) {
                cur = ref;
                break;
            }
        }
    }

    // one navigation node: its (up to) four slots through the conservative test; the nearest one that can hold a candidate
    // is visited next, the others are deferred in far-to-near order, each with its entry distance.  An unused slot is not
    // tested for: it holds a box that is a single point and the reference of a leaf without triangles (empty_slot_ref in
    // scene_dev.h), so the rare ray that passes through that point visits nothing.
    template <class C>
    PTD void inner(const DevScene& sc, const Treelet& tl, DistStack& st, C& wc) {
        wc.node();
        if (cur < tl.n) wc.top();
#if PT_NAV_WIDTH == 8
        inner8(sc, tl, st);
#else
        {
            uint4 c0, c1, c2, c3;
            nav_fetch(sc, tl, cur, c0, c1, c2, c3);
            const float inf = __builtin_huge_valf(), far = cull_bound();
            const PlaneSwap sw(cr);
            const BoxC b0 = box_cheap(c0.x, c0.y, c0.z, cr, sw), b1 = box_cheap(c1.x, c1.y, c1.z, cr, sw);
            const BoxC b2 = box_cheap(c2.x, c2.y, c2.z, cr, sw), b3 = box_cheap(c3.x, c3.y, c3.z, cr, sw);
            float k0 = cheap_go(b0, tmin, far) ? b0.tn : inf;
            float k1 = cheap_go(b1, tmin, far) ? b1.tn : inf;
            float k2 = cheap_go(b2, tmin, far) ? b2.tn : inf;
            float k3 = cheap_go(b3, tmin, far) ? b3.tn : inf;
            uint32_t r0 = c0.w, r1 = c1.w, r2 = c2.w, r3 = c3.w;
            auto order = [](float& ka, uint32_t& ra, float& kb, uint32_t& rb) {  // afterwards ka <= kb
                const bool sw = kb < ka;
                const float kt = sw ? kb : ka;
                const uint32_t rt = sw ? rb : ra;
                kb = sw ? ka : kb;
                rb = sw ? ra : rb;
                ka = kt;
                ra = rt;
            };
            order(k0, r0, k1, r1);
            order(k2, r2, k3, r3);
            order(k0, r0, k2, r2);
            order(k1, r1, k3, r3);
            order(k1, r1, k2, r2);
            if (k3 < inf) st.push(r3, k3);
            if (k2 < inf) st.push(r2, k2);
            if (k1 < inf) st.push(r1, k1);
            if (k0 < inf)
                cur = r0;
            else
                next(st);
        }
#endif
    }
#if PT_NAV_WIDTH == 8
    // The same for an 8-wide node: the slots are tried in turn; the nearest one so far is kept in registers and visited
    // next, every other one that can hold a candidate goes onto the stack with its entry distance (in slot order: what is
    // popped first is decided by the distances only in so far as entries beyond the best hit are dropped).
    PTD void inner8(const DevScene& sc, const Treelet& tl, DistStack& st) {
        Nav8 n;
        n.fetch(sc, tl, cur);
        const NodeFrame8 f(n, o, cr);
        const float inf = __builtin_huge_valf(), far = cull_bound();
        uint32_t next_child = n.v1.x, next_leaf = n.v1.y;
        float best = inf;
        uint32_t best_ref = kDone;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t meta = byte_of(k < 4 ? n.v1.z : n.v1.w, k & 3);
            const bool interior = (meta & kNav8Interior) != 0u;
            const uint32_t ref = interior ? next_child : (kLeafBit | (meta << kLeafCountShift) | next_leaf);
            next_child += interior ? 1u : 0u;
            next_leaf += (interior || meta == 0u) ? 0u : 1u + meta;
            const BoxC b = box_cheap8(n, f, k);
            if (bool(meta != 0u) & cheap_go(b, tmin, far)) {
                const bool nearer = b.tn < best;
                const float push_k = nearer ? best : b.tn;
                const uint32_t push_r = nearer ? best_ref : ref;
                if (push_k < inf) st.push(push_r, push_k);
                best = nearer ? b.tn : best;
                best_ref = nearer ? ref : best_ref;
            }
        }
        if (best < inf)
            cur = best_ref;
        else
            next(st);
    }
#endif

    // A leaf: its triangles are tried at once; whether the reference reaches this leaf at all (its decision on the leaf's
    // parent box, see box_cheap in traverse.h) is settled once per ray, for the leaf of the final winner (settle()).  A
    // candidate from a leaf the reference does not reach can only shrink the window to beyond the true winner's distance,
    // so it never hides the true winner; if it ends up as the winner itself, settle() sends the ray to the full query.
    template <class C>
    PTD void leaf(const DevScene& sc, DistStack& st, C& wc) {
        const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 15u;
        // the next triangle's record is requested before the current one is tested (the last one is requested twice)
        TriRecord nxt;
        nxt.fetch(sc, first + 1u);
        for (uint32_t k = 0; k < count; ++k) {
            TriRecord tr = nxt;
            nxt.fetch(sc, first + 1u + min(k + 1u, count - 1u));
            tr.pin();
            wc.tri();
            float t, u, v;
            if (tri_hit(tr.p0(), tr.e1(), tr.e2(), o, d, tmin, window, t, u, v)) {
                near_tie = !(hit.prim < 0 || t * (1.0f + kNearTie) < hit.t);
                if (hit.prim < 0 || t <= hit.t) {
                    hit.prim = tr.prim();
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit_leaf = first;
                    window = fminf(t * (1.0f + kNearTie), tmax);
                }
            }
        }
        next(st);
    }

    // Called once when the walk has ended.  Returns true when the query has to be repeated in reference order, `bound` =
    // how far (kDeferExact = all the way):
    //  - the reference does not accept the parent box of the winner's leaf, so it never sees the winner: full query;
    //  - it enters that box beyond the window behind the winner (in any of the three slabs: a grazing ray's triangle
    //    distance can lie well before the plane distance of the flat box around it): every bound the reference compares
    //    with an ancestor of the leaf is another candidate's distance or the ray's far end, all beyond the window (else
    //    near_tie below), and the ancestors' entry distances are at most the parent's -- so a parent entered within the
    //    window makes sure that none of them is culled by such a bound, nor by the bounded replay; otherwise only the
    //    full query is safe;
    //  - two candidates, or the winner and the ray's far end, within the window of each other: the visit order decides.
    template <class C>
    PTD bool settle(const DevScene& sc, float& bound, C& wc) {
        if (hit.prim < 0) return false;
        wc.tri();
        // the ray's reciprocals are only needed here, once per query, and are recomputed rather than kept
        const RayPre pre = make_pre(sc, RayD{o, d, tmin, tmax});
        float entry;
        if (!leaf_parent_accepts(sc, hit_leaf, pre, tmax, entry) || !(entry <= hit.t * (1.0f + kNearTie))) {
            bound = kDeferExact;
            return true;
        }
        bound = hit.t * (1.0f + 2.0f * kNearTie);
        const bool order_matters = near_tie || (tmax < __builtin_huge_valf() && !(hit.t * (1.0f + kNearTie) < tmax));
        // A winner at distance zero (a ray that starts on a surface with a near end of zero): the relative window has no
        // width there, and a replay bounded to 0 would cull the very boxes that hold the winner (the reference enters a
        // box whose entry distance is BELOW its bound, and 0 < 0 fails).  Replay such a query all the way.
        if (order_matters && !(bound > hit.t)) bound = kDeferExact;
        return order_matters;
    }
};

struct AnyLane {
    typedef AnyStack StackT;
    static constexpr int kSpillWords = AnyStack::kSpill;
    static constexpr int kLdsRows = AnyStack::kRows;
    static constexpr bool kReferenceOrder = false;
    static constexpr uint32_t kTreeletNodes = kTreelet;
    PTD void idle() {
        cur = kDone;
        occluded = false;
    }
    PTD bool box_ready() const { return !(cur >> 31); }
    PTD bool leaf_ready() const { return cur > kDone; }
    PTD bool finished() const { return cur == kDone; }
    V3 o, d;
    NavRay cr;
    float tmin, tmax;
    float far_end;  // tmax widened by the near-tie window: a box the ray enters beyond it is skipped
    bool occluded;
    uint32_t hit_leaf;  // header slot of the leaf whose triangle ended the walk
    uint32_t cur;

    PTD void begin(const DevScene& sc, const RayD& ray, const RayPre& p) {
        o = p.o;
        d = p.d;
        tmin = p.tmin;
        cr.set(sc, p);
        tmax = ray.tmax;
        far_end = ray.tmax * (1.0f + kNearTie);
        occluded = false;
        hit_leaf = 0u;
        cur = sc.wide_root;
    }
    PTD void next(AnyStack& st) { cur = st.empty() ? kDone : st.pop(); }

    // The reference's bound is constant for this query, so the leaves it reaches do not depend on the visit order: it
    // reaches a leaf iff it accepts the leaf's parent box.  Any walk that visits every such leaf that can hold an accepted
    // triangle, and tries a leaf's triangles only after the parent's exact test, returns what the reference returns.
    template <class C>
    PTD void inner(const DevScene& sc, const Treelet& tl, AnyStack& st, C& wc) {
        wc.node();
        if (cur < tl.n) wc.top();
#if PT_NAV_WIDTH == 8
        inner8(sc, tl, st);
#else
        {
            uint4 c0, c1, c2, c3;
            nav_fetch(sc, tl, cur, c0, c1, c2, c3);
            const PlaneSwap sw(cr);
            const bool g0 = cheap_go(box_cheap(c0.x, c0.y, c0.z, cr, sw), tmin, far_end);
            const bool g1 = cheap_go(box_cheap(c1.x, c1.y, c1.z, cr, sw), tmin, far_end);
            const bool g2 = cheap_go(box_cheap(c2.x, c2.y, c2.z, cr, sw), tmin, far_end);
            const bool g3 = cheap_go(box_cheap(c3.x, c3.y, c3.z, cr, sw), tmin, far_end);
            // the order does not change the answer; slots are listed larger subtree first, as the builder orders children for
            // exactly this query (src/bvh.cpp:137-144), so the first slot is visited first and the last one pushed first
            uint32_t nxt = kDone;
            if (g3) nxt = c3.w;
            if (g2) {
                if (nxt != kDone) st.push(nxt);
                nxt = c2.w;
            }
            if (g1) {
                if (nxt != kDone) st.push(nxt);
                nxt = c1.w;
            }
            if (g0) {
                if (nxt != kDone) st.push(nxt);
                nxt = c0.w;
            }
            if (nxt != kDone)
                cur = nxt;
            else
                next(st);
        }
#endif
    }
#if PT_NAV_WIDTH == 8
    // 8-wide node: every slot that can hold an occluder is visited, the first one next, the others from the stack
    PTD void inner8(const DevScene& sc, const Treelet& tl, AnyStack& st) {
        Nav8 n;
        n.fetch(sc, tl, cur);
        const NodeFrame8 f(n, o, cr);
        uint32_t next_child = n.v1.x, next_leaf = n.v1.y;
        uint32_t nxt = kDone;
        uint32_t refs[8];
        bool go[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t meta = byte_of(k < 4 ? n.v1.z : n.v1.w, k & 3);
            const bool interior = (meta & kNav8Interior) != 0u;
            refs[k] = interior ? next_child : (kLeafBit | (meta << kLeafCountShift) | next_leaf);
            next_child += interior ? 1u : 0u;
            next_leaf += (interior || meta == 0u) ? 0u : 1u + meta;
            go[k] = bool(meta != 0u) & cheap_go(box_cheap8(n, f, k), tmin, far_end);
        }
        // slots are listed larger subtree first, as the builder orders children for this query (src/bvh.cpp:137-144): the
        // first slot is visited first and the last one pushed first
#pragma unroll
        for (int k = 7; k >= 0; --k) {
            if (go[k]) {
                if (nxt != kDone) st.push(nxt);
                nxt = refs[k];
            }
        }
        if (nxt != kDone)
            cur = nxt;
        else
            next(st);
    }
#endif

    // the first accepted triangle ends the walk; whether the reference reaches its leaf is settled afterwards
    template <class C>
    PTD void leaf(const DevScene& sc, AnyStack& st, C& wc) {
        const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 15u;
        TriRecord nxt;
        nxt.fetch(sc, first + 1u);
        for (uint32_t k = 0; k < count; ++k) {
            TriRecord tr = nxt;
            nxt.fetch(sc, first + 1u + min(k + 1u, count - 1u));
            tr.pin();
            wc.tri();
            float t, u, v;
            if (tri_hit(tr.p0(), tr.e1(), tr.e2(), o, d, tmin, tmax, t, u, v)) {
                occluded = true;
                hit_leaf = first;
                cur = kDone;
                return;
            }
        }
        next(st);
    }

    // Called once when the walk has ended.  The reference reports "occluded" iff it reaches some leaf with an accepted
    // triangle, i.e. iff it accepts that leaf's parent box.  If it does not accept the parent of the leaf found here,
    // another leaf may still qualify: the ray is handed back and traced in reference order (true = defer).
    template <class C>
    PTD bool settle(const DevScene& sc, C& wc) {
        if (!occluded) return false;
        wc.tri();
        const RayPre pre = make_pre(sc, RayD{o, d, tmin, tmax});
        float entry;
        if (leaf_parent_accepts(sc, hit_leaf, pre, tmax, entry)) return false;
        occluded = false;
        return true;
    }
};

// ---- reference order in the same loops -----------------------------------------------------------------------------
//
// src/bvh.cpp:26-59 and :61-86 step by step on the binary tree: a popped node is fetched; a leaf's triangles are tried
// without a box test; an interior node is culled with the CURRENT bound and its children are pushed as (first,
// first + 1), so first + 1 is looked at first.  Nothing is reordered or left out: per ray this is closest_exact_t /
// any_exact_t, only the lanes of a wave take their box steps and their triangle steps together and finished lanes are
// refilled.
constexpr uint32_t kNoNode = 0xffffffffu;

struct ExactClosestLane {
    typedef Stack StackT;
    static constexpr int kSpillWords = kSpillDepth;
    static constexpr int kLdsRows = kLdsDepth;
    static constexpr bool kReferenceOrder = true;
    static constexpr uint32_t kTreeletNodes = 0;  // walks the binary nodes
    RayPre pre;
    PTD const V3& org() const { return pre.o; }
    PTD const V3& dir() const { return pre.d; }
    float max_dis;
    HitD hit;
    uint32_t cur;                      // binary node to fetch next, or kNoNode
    uint32_t leaf_first, leaf_count;   // the leaf the lane stands on (count != 0)

    PTD void idle() {
        cur = kNoNode;
        leaf_count = 0u;
        hit = HitD{-1, 0.0f, 0.0f, 0.0f};
    }
    PTD void begin(const DevScene&, const RayD& ray, const RayPre& p) {
        pre = p;
        max_dis = ray.tmax;
        hit = HitD{-1, 0.0f, 0.0f, 0.0f};
        cur = 0u;
        leaf_first = leaf_count = 0u;
    }
    PTD bool box_ready() const { return leaf_count == 0u && cur != kNoNode; }
    PTD bool leaf_ready() const { return leaf_count != 0u; }
    PTD bool finished() const { return leaf_count == 0u && cur == kNoNode; }
    PTD void next(Stack& st) { cur = st.empty() ? kNoNode : st.pop(); }

    template <class C>
    PTD void inner(const DevScene& sc, const Treelet&, Stack& st, C& wc) {
        wc.node();
        const float4 n0 = sc.nodes[2 * size_t(cur)];
        const float4 n1 = sc.nodes[2 * size_t(cur) + 1];
        const uint32_t first = as_u(n1.z), count = as_u(n1.w);
        if (count != 0u) {
            leaf_first = first;
            leaf_count = count;
            return;
        }
        const BoxT b = box_decide<false>(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, pre);
        if (b.overlap && b.tn < max_dis) {
            st.push(first);
            cur = first + 1u;
        } else {
            next(st);
        }
    }
    template <class C>
    PTD void leaf(const DevScene& sc, Stack& st, C& wc) {
        for (uint32_t k = 0; k < leaf_count; ++k) {
            V3 p0, e1, e2;
            int32_t prim;
            load_tri(sc, leaf_first + k, p0, e1, e2, prim);
            wc.tri();
            float t, u, v;
            if (tri_hit(p0, e1, e2, pre.o, pre.d, pre.tmin, max_dis, t, u, v)) {
                max_dis = t;
                hit = HitD{prim, t, u, v};
            }
        }
        leaf_count = 0u;
        next(st);
    }
    template <class C>
    PTD bool settle(const DevScene&, float&, C&) { return false; }
};

struct ExactAnyLane {
    typedef Stack StackT;
    static constexpr int kSpillWords = kSpillDepth;
    static constexpr int kLdsRows = kLdsDepth;
    static constexpr bool kReferenceOrder = true;
    static constexpr uint32_t kTreeletNodes = 0;  // walks the binary nodes
    RayPre pre;
    float tmax;
    bool occluded;
    uint32_t cur, leaf_first, leaf_count;

    PTD void idle() {
        cur = kNoNode;
        leaf_count = 0u;
        occluded = false;
    }
    PTD void begin(const DevScene&, const RayD& ray, const RayPre& p) {
        pre = p;
        tmax = ray.tmax;
        occluded = false;
        cur = 0u;
        leaf_first = leaf_count = 0u;
    }
    PTD bool box_ready() const { return leaf_count == 0u && cur != kNoNode; }
    PTD bool leaf_ready() const { return leaf_count != 0u; }
    PTD bool finished() const { return leaf_count == 0u && cur == kNoNode; }
    PTD void next(Stack& st) { cur = st.empty() ? kNoNode : st.pop(); }

    template <class C>
    PTD void inner(const DevScene& sc, const Treelet&, Stack& st, C& wc) {
        wc.node();
        const float4 n0 = sc.nodes[2 * size_t(cur)];
        const float4 n1 = sc.nodes[2 * size_t(cur) + 1];
        const uint32_t first = as_u(n1.z), count = as_u(n1.w);
        if (count != 0u) {
            leaf_first = first;
            leaf_count = count;
            return;
        }
        const BoxT b = box_decide<false>(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, pre);
        if (b.overlap && b.tn < tmax) {
            st.push(first);
            cur = first + 1u;
        } else {
            next(st);
        }
    }
    template <class C>
    PTD void leaf(const DevScene& sc, Stack& st, C& wc) {
        for (uint32_t k = 0; k < leaf_count; ++k) {
            V3 p0, e1, e2;
            int32_t prim;
            load_tri(sc, leaf_first + k, p0, e1, e2, prim);
            wc.tri();
            float t, u, v;
            if (tri_hit(p0, e1, e2, pre.o, pre.d, pre.tmin, tmax, t, u, v)) {
                occluded = true;
                leaf_count = 0u;
                cur = kNoNode;
                return;
            }
        }
        leaf_count = 0u;
        next(st);
    }
    template <class C>
    PTD bool settle(const DevScene&, C&) { return false; }
};

// lane states of the shared loop
constexpr int kLaneEmpty = 0, kLaneBusy = 1, kLaneDone = 2;

// Takes `want` lanes' next queue positions from the shard's fetch counter.  Returns the position (valid when < count) and
// clears `more` once the counter has passed the end of the queue.
PTD uint32_t wave_fetch(bool want, uint32_t* counter, uint32_t count, bool& more) {
    const uint64_t mask = wave_ballot(want);
    const uint32_t n = uint32_t(__popcll(mask));
    uint32_t base = 0;
    if (n != 0u) {
        const int leader = __ffsll((unsigned long long)mask) - 1;
        if (int(lane_id()) == leader) base = atomicAdd(counter, n);
        base = uint32_t(__shfl(int(base), leader));
        more = base + n < count;
    }
    return base + uint32_t(__popcll(mask & ((1ull << lane_id()) - 1ull)));
}

// Closest hit for the rays of one shard.  Io supplies
//   uint32_t count(shard)                                   rays in the shard's queue
//   bool load(sc, shard, j, RayD&, Carry&)                     ray j and whatever the caller wants back at commit time;
//                                                           false = position j holds no query (reported as a miss)
//   void commit(sc, shard, fin, j, hit, defer, bound, carry, o, d)   called by the whole wave; fin marks lanes with a result
// lds_stack: this lane's column of a [Lane::kLdsRows][kBlock] array of words.
//
// A query whose walk has ended stays in its lane until the next refill; settle() and commit() then run once for all
// lanes that have a result (run per loop round for the one or two lanes that just ended, settle() -- an exact box
// decision behind a dependent fetch -- cost as much as a box step of the whole wave).
template <class Lane, class Io, class C>
PTD void closest_wave(const DevScene& sc, Io& io, uint32_t shard, uint32_t* fetch, uint32_t* lds_stack, const Treelet& tl, C& wc) {
    const uint32_t count = io.count(shard);
    uint32_t spill[Lane::kSpillWords];
    typename Lane::StackT st;
    st.init(lds_stack, spill);
    Lane L;
    L.idle();
    typename Io::Carry carry{};
    uint32_t j = 0;
    int state = kLaneEmpty;
    bool defer = false, more = true;
    float bound = 0.0f;
    PhaseStats ps;
    for (;;) {
        const bool walked = state == kLaneBusy && L.finished();
        const bool fin = walked || state == kLaneDone;
        if (wave_any(fin)) {
            ps.round(fin);
            if (walked) defer = L.settle(sc, bound, wc);
            io.commit(sc, shard, fin, j, L.hit, defer, bound, carry, L.org(), L.dir());
            if (fin) state = kLaneEmpty;
        }
        if (more) {
            const bool want = state == kLaneEmpty;
            const uint32_t jj = wave_fetch(want, fetch + shard, count, more);
            if (want && jj < count) {
                j = jj;
                RayD ray;
                const bool has_ray = io.load(sc, shard, j, ray, carry);
                const RayPre pre = make_pre(sc, ray);
                if (!has_ray) {  // a queue position without a query (a pixel slot outside the image)
                    state = kLaneDone;
                    defer = false;
                    L.idle();
                } else if (pre.special) {
                    state = kLaneDone;
                    defer = true;
                    bound = kDeferExact;
                    L.idle();
                } else {
                    L.begin(sc, ray, pre);
                    st.reset();
                    state = kLaneBusy;
                    defer = false;
                }
            }
        }
        if (!wave_any(state == kLaneBusy)) {
            if (more || wave_any(state == kLaneDone)) continue;
            break;
        }
        const uint32_t low = more ? kRefillBelow : 1u;
        for (;;) {
            for (;;) {
                const bool inner = L.box_ready();  // only a busy lane has a node to visit: the others hold kDone
                const uint32_t n_inner = wave_count(inner);
                if (n_inner == 0u) break;
                ps.box(n_inner);
                if (inner) L.inner(sc, tl, st, wc);
                if (n_inner < kMinBoxLanes) break;
            }
            if constexpr (!Lane::kReferenceOrder) ps.leaf(L.leaf_ready(), L.cur);
            if (L.leaf_ready()) L.leaf(sc, st, wc);
            if (wave_count(!L.finished()) < low) break;
        }
    }
    ps.flush(Io::kPhaseClass);
}

// Any hit for the rays of one shard.  Io: count(shard), load(sc, shard, j, RayD&, Carry&),
// commit(sc, shard, fin, j, occluded, defer, carry) -- per lane, no wave-level operation required.
template <class Lane, class Io, class C>
PTD void any_wave(const DevScene& sc, Io& io, uint32_t shard, uint32_t* fetch, uint32_t* lds_stack, const Treelet& tl, C& wc) {
    const uint32_t count = io.count(shard);
    uint32_t spill[Lane::kSpillWords];
    typename Lane::StackT st;
    st.init(lds_stack, spill);
    Lane L;
    L.idle();
    typename Io::Carry carry{};
    uint32_t j = 0;
    int state = kLaneEmpty;
    bool defer = false, more = true;
    PhaseStats ps;
    for (;;) {
        const bool walked = state == kLaneBusy && L.finished();
        const bool fin = walked || state == kLaneDone;
        if (wave_any(fin)) {
            ps.round(fin);
            if (walked) defer = L.settle(sc, wc);
            io.commit(sc, shard, fin, j, L.occluded, defer, carry);
            if (fin) state = kLaneEmpty;
        }
        if (more) {
            const bool want = state == kLaneEmpty;
            const uint32_t jj = wave_fetch(want, fetch + shard, count, more);
            if (want && jj < count) {
                j = jj;
                RayD ray;
                io.load(sc, shard, j, ray, carry);
                const RayPre pre = make_pre(sc, ray);
                if (pre.special) {
                    state = kLaneDone;
                    defer = true;
                    L.idle();
                } else {
                    L.begin(sc, ray, pre);
                    st.reset();
                    state = kLaneBusy;
                    defer = false;
                }
            }
        }
        if (!wave_any(state == kLaneBusy)) {
            if (more || wave_any(state == kLaneDone)) continue;
            break;
        }
        const uint32_t low = more ? kRefillBelow : 1u;
        for (;;) {
            for (;;) {
                const bool inner = L.box_ready();  // only a busy lane has a node to visit: the others hold kDone
                const uint32_t n_inner = wave_count(inner);
                if (n_inner == 0u) break;
                ps.box(n_inner);
                if (inner) L.inner(sc, tl, st, wc);
                if (n_inner < kMinBoxLanes) break;
            }
            if constexpr (!Lane::kReferenceOrder) ps.leaf(L.leaf_ready(), L.cur);
            if (L.leaf_ready()) L.leaf(sc, st, wc);
            if (wave_count(!L.finished()) < low) break;
        }
    }
    ps.flush(Io::kPhaseClass);
}

// LDS words per workgroup of a kernel built on the wave loops for lane type Lane; the follow-up traversals of deferred
// rays use the same memory as a plain stack of kLdsDepth rows
// uint4 of LDS for the workgroup's copy of the top navigation nodes, for lane type Lane (at least one, so that the array
// can be declared)
template <class Lane>
constexpr int treelet_lds_vec4() {
    return Lane::kTreeletNodes != 0u ? int(Lane::kTreeletNodes) * kNavVec4 : 1;
}
template <class Lane>
PTD Treelet wave_treelet(const DevScene& sc, uint4* s_nodes) {
    if constexpr (Lane::kTreeletNodes != 0u)
        return treelet_fill(sc, s_nodes);
    else
        return Treelet{nullptr, 0u};
}

// LDS words per workgroup of a kernel built on the wave loops for lane type Lane; the follow-up traversals of deferred
// rays use the same memory as a plain stack of Lane::kLdsRows rows
template <class Lane>
constexpr int wave_lds_words() {
    return Lane::kLdsRows * kBlock;
}

}  // namespace ptd
