// BVH traversal for wave64: reference-order ("exact") and ordered 2-wide ("fast") closest-hit / any-hit.
//
// Reference behaviour being reproduced (paths relative to the reference root):
//   src/bvh.cpp:26-59   closest hit: explicit stack, pop a node, leaves are never box-tested, interior nodes are culled
//                       with the CURRENT max distance, children pushed as (first, first+1) so first+1 is visited first
//   src/bvh.cpp:61-86   any hit: same walk, returns at the first accepted triangle
//   src/aabb.cpp:6-35   slab test with six real divisions; the z slab rejects but does not tighten the interval
//   src/shape.cpp:50-89 Moeller-Trumbore with |det| < 1e-5 rejection and inclusive bounds on u, v, u+v and t
//
// Per-lane traversal stack: the first kLdsDepth entries live in LDS, laid out [entry][lane] so that a wave's 64 lanes
// hit 64 consecutive banks; deeper entries spill to a private (scratch) array.  The host rejects trees deeper than
// kLdsDepth + kSpillDepth.
#pragma once

#include "scene_dev.h"

namespace ptd {

constexpr int kBlock = 128;      // threads per workgroup of every traversal kernel
constexpr int kLdsDepth = 24;    // stack entries per lane in LDS
constexpr int kSpillDepth = 72;  // further entries per lane in scratch
constexpr int kMaxStack = kLdsDepth + kSpillDepth;

struct Stack {
    uint32_t* lds;  // &shared[0][threadIdx.x]
    uint32_t spill[kSpillDepth];
    int sp;
    PTD void init(uint32_t* base) {
        lds = base;
        sp = 0;
    }
    PTD void push(uint32_t v) {
        if (sp < kLdsDepth)
            lds[sp * kBlock] = v;
        else
            spill[sp - kLdsDepth] = v;
        ++sp;
    }
    PTD uint32_t pop() {
        --sp;
        return (sp < kLdsDepth) ? lds[sp * kBlock] : spill[sp - kLdsDepth];
    }
    PTD bool empty() const { return sp == 0; }
};

struct RayD {
    V3 o, d;
    float tmin, tmax;
};

// Work counters (node records fetched, triangle records fetched).  NoCount compiles to nothing; Count is used by the
// instrumented kernel instantiations that produce the per-ray averages for the bandwidth model.
struct NoCount {
    PTD void node() {}
    PTD void tri() {}
};
struct Count {
    uint32_t nodes = 0, tris = 0;
    PTD void node() { ++nodes; }
    PTD void tri() { ++tris; }
};

struct HitD {
    int32_t prim;
    float t, u, v;
};

// src/aabb.cpp:6-35, verbatim in behaviour (NaN compares included)
PTD bool box_hit_ref(float4 n0, float4 n1, const RayD& r, float max_dis) {
    const float lox = n0.x, loy = n0.y, loz = n0.z, hix = n0.w, hiy = n1.x, hiz = n1.y;
    float tmin = (lox - r.o.x) / r.d.x;
    float tmax = (hix - r.o.x) / r.d.x;
    if (tmin > tmax) {
        const float s = tmin;
        tmin = tmax;
        tmax = s;
    }
    float tymin = (loy - r.o.y) / r.d.y;
    float tymax = (hiy - r.o.y) / r.d.y;
    if (tymin > tymax) {
        const float s = tymin;
        tymin = tymax;
        tymax = s;
    }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (loz - r.o.z) / r.d.z;
    float tzmax = (hiz - r.o.z) / r.d.z;
    if (tzmin > tzmax) {
        const float s = tzmin;
        tzmin = tzmax;
        tzmax = s;
    }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return (tmin < max_dis) && (tmax > r.tmin);
}

// src/shape.cpp:50-89 on a pre-gathered record {p0, e1 = p1 - p0, e2 = p2 - p0} (the subtractions are exact
// restatements: the reference computes the same two differences per test)
PTD bool tri_hit(V3 p0, V3 e1, V3 e2, const RayD& r, float max_dis, float& t, float& u, float& v) {
    const V3 pvec = cross(r.d, e2);
    const float det = dot(e1, pvec);
    if (det > -1e-5f && det < 1e-5f) return false;
    const float inv = 1.0f / det;
    const V3 tvec = r.o - p0;
    u = dot(tvec, pvec) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const V3 qvec = cross(tvec, e1);
    v = dot(r.d, qvec) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, qvec) * inv;
    return t >= r.tmin && t <= max_dis;
}

PTD void load_tri(const DevScene& sc, uint32_t slot, V3& p0, V3& e1, V3& e2, int32_t& prim) {
    const float4 a = sc.tris[3 * size_t(slot) + 0];
    const float4 b = sc.tris[3 * size_t(slot) + 1];
    const float4 c = sc.tris[3 * size_t(slot) + 2];
    p0 = V3{a.x, a.y, a.z};
    e1 = V3{b.x, b.y, b.z};
    e2 = V3{c.x, c.y, c.z};
    prim = int32_t(as_u(a.w));
}

// ------------------------------------------------------------------------------------------------ reference order

// Closest hit in the reference's visit order.  Returns true when something was hit.
template <class C>
PTD bool closest_exact(const DevScene& sc, const RayD& ray, HitD& hit, uint32_t* lds_stack, C& wc) {
    Stack st;
    st.init(lds_stack);
    st.push(0u);
    float max_dis = ray.tmax;
    bool found = false;
    hit.prim = -1;
    hit.t = 0.0f;
    hit.u = hit.v = 0.0f;
    while (!st.empty()) {
        const uint32_t ni = st.pop();
        wc.node();
        const float4 n0 = sc.nodes[2 * size_t(ni)];
        const float4 n1 = sc.nodes[2 * size_t(ni) + 1];
        const uint32_t first = as_u(n1.z), count = as_u(n1.w);
        if (count != 0u) {
            for (uint32_t k = 0; k < count; ++k) {
                V3 p0, e1, e2;
                int32_t prim;
                load_tri(sc, first + k, p0, e1, e2, prim);
                wc.tri();
                float t, u, v;
                if (tri_hit(p0, e1, e2, ray, max_dis, t, u, v)) {
                    max_dis = t;
                    found = true;
                    hit.prim = prim;
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                }
            }
        } else if (box_hit_ref(n0, n1, ray, max_dis)) {
            st.push(first);
            st.push(first + 1u);
        }
    }
    return found;
}

// Any hit in the reference's visit order; first_prim receives the first accepted triangle or -1.
template <class C>
PTD bool any_exact(const DevScene& sc, const RayD& ray, int32_t& first_prim, uint32_t* lds_stack, C& wc) {
    Stack st;
    st.init(lds_stack);
    st.push(0u);
    first_prim = -1;
    while (!st.empty()) {
        const uint32_t ni = st.pop();
        wc.node();
        const float4 n0 = sc.nodes[2 * size_t(ni)];
        const float4 n1 = sc.nodes[2 * size_t(ni) + 1];
        const uint32_t first = as_u(n1.z), count = as_u(n1.w);
        if (count != 0u) {
            for (uint32_t k = 0; k < count; ++k) {
                V3 p0, e1, e2;
                int32_t prim;
                load_tri(sc, first + k, p0, e1, e2, prim);
                wc.tri();
                float t, u, v;
                if (tri_hit(p0, e1, e2, ray, ray.tmax, t, u, v)) {
                    first_prim = prim;
                    return true;
                }
            }
        } else if (box_hit_ref(n0, n1, ray, ray.tmax)) {
            st.push(first);
            st.push(first + 1u);
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ ordered, 2-wide

// Conservative slab test against both child boxes of a wide node: multiplies by a reciprocal direction, ignores NaN
// slabs (hardware min/max return the non-NaN operand) and widens the far distance by a few ulps, so that a box is
// only rejected when no accepted hit can lie inside it.  (The reference's test is looser still: its z slab does not
// tighten the interval.  Boxes it accepts beyond that cannot contain a nearer hit.)  near0/near1 are entry distances
// used only for ordering.
struct RayFast {
    V3 o, d, rd;
    float tmin;
};

PTD void box2(const float4 w0, const float4 w1, const float4 w2, const RayFast& r, float best, bool& h0, bool& h1, float& near0,
              float& near1) {
    {
        const float tx0 = (w0.x - r.o.x) * r.rd.x, tx1 = (w0.w - r.o.x) * r.rd.x;
        const float ty0 = (w0.y - r.o.y) * r.rd.y, ty1 = (w1.x - r.o.y) * r.rd.y;
        const float tz0 = (w0.z - r.o.z) * r.rd.z, tz1 = (w1.y - r.o.z) * r.rd.z;
        const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), r.tmin));
        const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), best)) * 1.000001f;
        near0 = tn;
        h0 = tn <= tf;
    }
    {
        const float tx0 = (w1.z - r.o.x) * r.rd.x, tx1 = (w2.y - r.o.x) * r.rd.x;
        const float ty0 = (w1.w - r.o.y) * r.rd.y, ty1 = (w2.z - r.o.y) * r.rd.y;
        const float tz0 = (w2.x - r.o.z) * r.rd.z, tz1 = (w2.w - r.o.z) * r.rd.z;
        const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), r.tmin));
        const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), best)) * 1.000001f;
        near1 = tn;
        h1 = tn <= tf;
    }
}

PTD RayFast make_fast(const RayD& ray) {
    RayFast r;
    r.o = ray.o;
    r.d = ray.d;
    // a zero direction component would give 0 * inf = NaN for an origin on a slab plane; a huge finite reciprocal gives
    // 0 there (the ray touches the plane) and +-huge elsewhere, which keeps the test conservative
    auto rcp = [](float d) { return fabsf(d) < 1e-30f ? copysignf(1e30f, d) : 1.0f / d; };
    r.rd = V3{rcp(ray.d.x), rcp(ray.d.y), rcp(ray.d.z)};
    r.tmin = ray.tmin;
    return r;
}

// Ordered closest hit.  `tie` is set when two different triangles were accepted at the same distance: the caller then
// re-traces the ray with closest_exact so that the winner is the one the reference's visit order picks.
template <class C>
PTD bool closest_fast(const DevScene& sc, const RayD& ray, HitD& hit, bool& tie, uint32_t* lds_stack, C& wc) {
    Stack st;
    st.init(lds_stack);
    const RayFast r = make_fast(ray);
    float best = ray.tmax;
    hit.prim = -1;
    hit.t = 0.0f;
    hit.u = hit.v = 0.0f;
    tie = false;
    uint32_t cur = sc.wide_root;
    for (;;) {
        if (!(cur & kLeafBit)) {
            const float4* w = sc.wide + 4 * size_t(cur);
            wc.node();
            const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            bool h0, h1;
            float n0, n1;
            box2(w0, w1, w2, r, best, h0, h1, n0, n1);
            const uint32_t r0 = as_u(w3.x), r1 = as_u(w3.y);
            if (h0 && h1) {
                const bool swap = n1 < n0;
                cur = swap ? r1 : r0;
                st.push(swap ? r0 : r1);
                continue;
            }
            if (h0 || h1) {
                cur = h0 ? r0 : r1;
                continue;
            }
        } else {
            const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 15u;
            for (uint32_t k = 0; k < count; ++k) {
                V3 p0, e1, e2;
                int32_t prim;
                load_tri(sc, first + k, p0, e1, e2, prim);
                wc.tri();
                float t, u, v;
                if (tri_hit(p0, e1, e2, ray, best, t, u, v)) {
                    if (t == best && hit.prim >= 0) tie = true;
                    best = t;
                    hit.prim = prim;
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                }
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    return hit.prim >= 0;
}

// Ordered any hit (no ordering needed for correctness; children are visited first-listed first, which is the
// larger-area child the builder stores first for shadow rays).
template <class C>
PTD bool any_fast(const DevScene& sc, const RayD& ray, uint32_t* lds_stack, C& wc) {
    Stack st;
    st.init(lds_stack);
    const RayFast r = make_fast(ray);
    uint32_t cur = sc.wide_root;
    for (;;) {
        if (!(cur & kLeafBit)) {
            const float4* w = sc.wide + 4 * size_t(cur);
            wc.node();
            const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            bool h0, h1;
            float n0, n1;
            box2(w0, w1, w2, r, ray.tmax, h0, h1, n0, n1);
            const uint32_t r0 = as_u(w3.x), r1 = as_u(w3.y);
            if (h0 && h1) {
                cur = r0;
                st.push(r1);
                continue;
            }
            if (h0 || h1) {
                cur = h0 ? r0 : r1;
                continue;
            }
        } else {
            const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 15u;
            for (uint32_t k = 0; k < count; ++k) {
                V3 p0, e1, e2;
                int32_t prim;
                load_tri(sc, first + k, p0, e1, e2, prim);
                wc.tri();
                float t, u, v;
                if (tri_hit(p0, e1, e2, ray, ray.tmax, t, u, v)) return true;
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
    return false;
}

}  // namespace ptd
