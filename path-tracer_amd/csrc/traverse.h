// BVH traversal for wave64: reference-order ("exact") and ordered 4-wide ("fast") closest-hit / any-hit.
//
// Reference behaviour being reproduced (paths relative to the reference root):
//   src/bvh.cpp:26-59   closest hit: explicit stack, pop a node, leaves are never box-tested, interior nodes are culled
//                       with the CURRENT max distance, children pushed as (first, first+1) so first+1 is visited first
//   src/bvh.cpp:61-86   any hit: same walk, returns at the first accepted triangle
//   src/aabb.cpp:6-35   slab test with six real divisions; the z slab rejects but does not tighten the interval
//   src/shape.cpp:50-89 Moeller-Trumbore with |det| < 1e-5 rejection and inclusive bounds on u, v, u+v and t
//
// The reference's slab test is not conservative: near box faces it can reject a box that contains a hit, and its
// outcome with NaN slabs (zero direction components) is erratic.  What it reports therefore depends on exactly which
// boxes it accepts, so BOTH modes evaluate the identical accept/reject decision, from bit-identical quotients:
//   - box_decide() computes the six quotients with a 5-instruction correctly rounded division (reciprocal computed
//     once per ray, one multiply, two fused residual corrections); rays with a zero, tiny, huge or non-finite direction
//     component are "special" and take the plain IEEE division so that the NaN / infinity cases come out as in the
//     reference, and so does any box for which one of the quotient estimates is zero or tiny;
//   - the ordered mode (traverse_wave.h) navigates with a conservative test and takes this exact decision once per
//     query, on the parent box of the leaf that holds the result (see box_cheap below for why that is enough); it differs
//     from the reference only in the ORDER in which subtrees are visited and in culling by distance, and falls back to
//     the reference order whenever order could matter.
//
// Per-lane traversal stack: the first kLdsDepth entries live in LDS, laid out [entry][lane] so that a wave's 64 lanes
// hit 64 consecutive banks; deeper entries spill to a private (scratch) array.  The host rejects trees deeper than
// kMaxStack entries.
#pragma once

#include "scene_dev.h"

namespace ptd {

constexpr int kBlock = 128;      // threads per workgroup of every traversal kernel
// 20 rows = 10 KB of LDS per workgroup: 14 workgroups (7 waves per SIMD, what the shadow kernel is built for) fit into
// the 160 KB of a CU; measured against 24 rows at 6 waves per SIMD: no difference.
#ifndef PT_LDS_DEPTH
#define PT_LDS_DEPTH 20
#endif
constexpr int kLdsDepth = PT_LDS_DEPTH;    // stack entries per lane in LDS
constexpr int kMaxStack = 96;              // deepest stack a traversal may need (the host rejects deeper trees)
constexpr int kSpillDepth = kMaxStack - kLdsDepth;  // further entries per lane in scratch

// Explicit address spaces: with generic pointers the compiler merges the two arms of pop() into one flat load.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(5))) uint32_t private_u32;

// The stack pointer is kept as the LDS byte address of the next free entry (a lane's entries are kBlock words apart), so
// that a push or pop in the LDS part costs a compare and the move of the pointer, no address arithmetic; the index is
// only formed for the scratch part.  The end of the LDS part is tested against a wave-uniform address: the columns of
// a wave's lanes lie within one row, so "below the wave's first word of row ROWS" is the same for all of them.
constexpr uint32_t kRowBytes = kBlock * 4u;
PTD uint32_t lds_addr(const uint32_t* p) { return uint32_t(uintptr_t((const lds_u32*)p)); }
PTD lds_u32* lds_ptr(uint32_t addr) { return (lds_u32*)uintptr_t(addr); }
template <int ROWS>
struct StackN {
    static constexpr int kRows = ROWS;            // entries per lane in LDS
    static constexpr int kSpill = kMaxStack - ROWS;  // further entries per lane in scratch
    uint32_t lo;         // address of &shared[0][threadIdx.x]
    uint32_t top;        // address of &shared[sp][threadIdx.x], also beyond the LDS rows (then only a number)
    uint32_t lim;        // wave-uniform: address of the wave's first word of row ROWS
    private_u32* spill;  // kSpill private words, declared by the function that owns the stack (kept out of this
                         // struct so that the compiler keeps everything but the array itself in registers)
    PTD void init(uint32_t* base, uint32_t* spill_words) {
        lo = lds_addr(base);
        top = lo;
        lim = uint32_t(__builtin_amdgcn_readfirstlane(int(lo - 4u * (threadIdx.x & 63u)))) + uint32_t(ROWS) * kRowBytes;
        spill = (private_u32*)spill_words;
    }
    PTD void reset() { top = lo; }
    PTD int over() const { return int((top - lo) / kRowBytes) - ROWS; }  // index into the scratch part
    PTD void push(uint32_t v) {
        if (top < lim)
            *lds_ptr(top) = v;
        else
            spill[over()] = v;
        top += kRowBytes;
    }
    PTD uint32_t pop() {
        top -= kRowBytes;
        return (top < lim) ? *lds_ptr(top) : spill[over()];
    }
    PTD bool empty() const { return top == lo; }
};
typedef StackN<kLdsDepth> Stack;

// Stack of the wave-level ordered closest-hit traversal: a deferred subtree is kept together with the distance at which
// the ray enters its box, so that a subtree that lies beyond the best hit by the time it is popped is dropped without
// its node being fetched.  ROWS entries per lane in LDS (references in rows [0, ROWS), distances in rows
// [ROWS, 2 ROWS) of a [row][lane] array), the rest in two private arrays.
template <int ROWS>
struct DistStackN {
    static constexpr int kRows = ROWS;
    static constexpr int kSpill = kMaxStack - ROWS;
    uint32_t lo, top, lim;  // as in StackN; top: the next free reference row of this lane
    private_u32* spill;     // kSpill references, then kSpill distances
    PTD void init(uint32_t* base, uint32_t* spill_words) {
        lo = lds_addr(base);
        top = lo;
        lim = uint32_t(__builtin_amdgcn_readfirstlane(int(lo - 4u * (threadIdx.x & 63u)))) + uint32_t(ROWS) * kRowBytes;
        spill = (private_u32*)spill_words;
    }
    PTD void reset() { top = lo; }
    PTD int over() const { return int((top - lo) / kRowBytes) - ROWS; }
    PTD void push(uint32_t v, float d) {
        if (top < lim) {
            lds_ptr(top)[0] = v;
            lds_ptr(top)[ROWS * kBlock] = __float_as_uint(d);
        } else {
            spill[over()] = v;
            spill[over() + kSpill] = __float_as_uint(d);
        }
        top += kRowBytes;
    }
    PTD void pop(uint32_t& v, float& d) {
        top -= kRowBytes;
        if (top < lim) {
            v = lds_ptr(top)[0];
            d = __uint_as_float(lds_ptr(top)[ROWS * kBlock]);
        } else {
            v = spill[over()];
            d = __uint_as_float(spill[over() + kSpill]);
        }
    }
    PTD bool empty() const { return top == lo; }
};

// LDS rows of the ordered traversals' stacks (closest hit: references and entry distances, kDistRows each): 16 rows = 8 KB
// per workgroup, which with the 2 KB of the LDS treelet (below) lets 14 workgroups -- 7 waves per SIMD, what the shadow
// kernel is built for -- share a CU's 160 KB (20 rows measured the same on their own).
#ifndef PT_DIST_ROWS
#define PT_DIST_ROWS 8
#endif
#ifndef PT_ANY_ROWS
#define PT_ANY_ROWS 16
#endif
constexpr int kDistRows = PT_DIST_ROWS;
constexpr int kAnyRows = PT_ANY_ROWS;
typedef DistStackN<kDistRows> DistStack;
typedef StackN<kAnyRows> AnyStack;

struct RayD {
    V3 o, d;
    float tmin, tmax;
};

struct HitD {
    int32_t prim;
    float t, u, v;
};

// Work counters (node records fetched, triangle records fetched).  NoCount compiles to nothing; Count is used by the
// instrumented kernel instantiations that produce the per-ray averages for the bandwidth model.
struct NoCount {
    PTD void node() {}
    PTD void top() {}
    PTD void tri() {}
};
struct Count {
    uint32_t nodes = 0, tops = 0, tris = 0;  // tops: those of `nodes` that were read from the LDS treelet
    PTD void node() { ++nodes; }
    PTD void top() { ++tops; }
    PTD void tri() { ++tris; }
};

// ------------------------------------------------------------------------------------------------ slab test

// Ray with its per-axis reciprocals.  `special` marks rays whose quotients must come from the plain division.
struct RayPre {
    V3 o, d, y;  // y = 1 / d, correctly rounded
    float tmin;
    bool special;
};

// Operand ranges for which the multiply-and-refine division (div_refine) is used.  Direction components within 2^-30 .. 2^30 in magnitude, origin and scene
// coordinates below 2^59: then no product overflows, and as long as the first estimate of a quotient is at least kTinyQ in
// magnitude every residual is a multiple of 2^-149, i.e. exact.  Everything else takes the division instruction.
constexpr float kDirLo = 0x1p-30f, kDirHi = 0x1p30f, kCoordHi = 0x1p59f, kTinyQ = 0x1p-60f;

PTD RayPre make_pre(const DevScene& sc, const RayD& r) {
    RayPre p;
    p.o = r.o;
    p.d = r.d;
    p.y = V3{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    p.tmin = r.tmin;
    auto odd = [](float d) { return !(fabsf(d) > kDirLo && fabsf(d) < kDirHi); };  // zero, tiny, huge, inf, NaN
    p.special = sc.all_special != 0u || odd(r.d.x) || odd(r.d.y) || odd(r.d.z) || !(fabsf(r.o.x) < kCoordHi) || !(fabsf(r.o.y) < kCoordHi) ||
                !(fabsf(r.o.z) < kCoordHi);
    return p;
}

// n / d, correctly rounded, from the estimate p = n * y with y = RN(1 / d): two residual corrections with fused
// multiply-adds (the residual n - q*d of a quotient that is already within an ulp is exact in an FMA).  Valid for the
// operand ranges above; callers check |p| >= kTinyQ.
PTD float div_refine(float p, float n, float d, float y) {
    float r = __fmaf_rn(-p, d, n);
    float q = __fmaf_rn(r, y, p);
    r = __fmaf_rn(-q, d, n);
    return __fmaf_rn(r, y, q);
}

// AABB::intersect (src/aabb.cpp:6-35) for finite quotients: the swaps are min / max, the two early-outs and the final
// test are the comparisons below.  `overlap` is the part that does not depend on the distance bound; `tn` is the
// entry distance of the x/y interval that the reference compares with max_dis (the z slab never tightens it),
// `order` = max(tn, z entry) is only used to visit the nearer child first.
struct BoxT {
    bool overlap;  // slabs overlap and the x/y interval ends after ray.tmin
    float tn;      // x/y entry distance
    float order;
    float leave;   // distance at which the ray leaves the box (all three slabs)
};

template <bool SPECIAL>
PTD BoxT box_decide(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayPre& r) {
    BoxT b;
    if (SPECIAL) {
        // verbatim control flow with IEEE divisions, so that NaN and infinity cases fall the way they do in the reference
        float tmin = (lox - r.o.x) / r.d.x, tmax = (hix - r.o.x) / r.d.x;
        if (tmin > tmax) {
            const float s = tmin;
            tmin = tmax;
            tmax = s;
        }
        float tymin = (loy - r.o.y) / r.d.y, tymax = (hiy - r.o.y) / r.d.y;
        if (tymin > tymax) {
            const float s = tymin;
            tymin = tymax;
            tymax = s;
        }
        b.overlap = !((tmin > tymax) || (tymin > tmax));
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = (loz - r.o.z) / r.d.z, tzmax = (hiz - r.o.z) / r.d.z;
        if (tzmin > tzmax) {
            const float s = tzmin;
            tzmin = tzmax;
            tzmax = s;
        }
        b.overlap = b.overlap && !((tmin > tzmax) || (tzmin > tmax)) && (tmax > r.tmin);
        b.tn = tmin;
        b.order = (tzmin > tmin) ? tzmin : tmin;
        b.leave = (tzmax < tmax) ? tzmax : tmax;
        return b;
    }
    const float nx0 = lox - r.o.x, nx1 = hix - r.o.x, ny0 = loy - r.o.y, ny1 = hiy - r.o.y, nz0 = loz - r.o.z, nz1 = hiz - r.o.z;
    const float px0 = nx0 * r.y.x, px1 = nx1 * r.y.x, py0 = ny0 * r.y.y, py1 = ny1 * r.y.y, pz0 = nz0 * r.y.z, pz1 = nz1 * r.y.z;
    const float smallest = fminf(fminf(fminf(fabsf(px0), fabsf(px1)), fminf(fabsf(py0), fabsf(py1))), fminf(fabsf(pz0), fabsf(pz1)));
    if (!(smallest >= kTinyQ)) return box_decide<true>(lox, loy, loz, hix, hiy, hiz, r);  // rare: a zero or tiny quotient
    const float x0 = div_refine(px0, nx0, r.d.x, r.y.x), x1 = div_refine(px1, nx1, r.d.x, r.y.x);
    const float y0 = div_refine(py0, ny0, r.d.y, r.y.y), y1 = div_refine(py1, ny1, r.d.y, r.y.y);
    const float z0 = div_refine(pz0, nz0, r.d.z, r.y.z), z1 = div_refine(pz1, nz1, r.d.z, r.y.z);
    const float tn = fmaxf(fminf(x0, x1), fminf(y0, y1)), tf = fminf(fmaxf(x0, x1), fmaxf(y0, y1));
    const float zn = fminf(z0, z1), zf = fmaxf(z0, z1);
    b.overlap = (tn <= tf) && (tn <= zf) && (zn <= tf) && (tf > r.tmin);
    b.tn = tn;
    b.order = fmaxf(tn, zn);
    b.leave = fminf(tf, zf);
    return b;
}

// src/shape.cpp:50-89 on a pre-gathered record {p0, e1 = p1 - p0, e2 = p2 - p0} (the subtractions are exact
// restatements: the reference computes the same two differences per test)
PTD bool tri_hit(V3 p0, V3 e1, V3 e2, const V3& o, const V3& d, float min_dis, float max_dis, float& t, float& u, float& v) {
    const V3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (det > -1e-5f && det < 1e-5f) return false;
    const float inv = 1.0f / det;
    const V3 tvec = o - p0;
    u = dot(tvec, pvec) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const V3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, qvec) * inv;
    return t >= min_dis && t <= max_dis;
}

// ------------------------------------------------------------------------------------------------ conservative slab test
//
// Why the ordered wave traversal may walk the tree with a cheap, padded slab test and still report what the reference
// reports: AABB::intersect (src/aabb.cpp:6-35) is MONOTONE under box nesting.  For a ray with finite quotients and boxes
// D inside A (the builder's boxes are exact unions, so lo_A <= lo_D <= hi_D <= hi_A holds exactly), every step that
// leads to a quotient -- the subtraction, the correctly rounded division -- is monotone in the plane coordinate, hence
// entry_A <= entry_D and exit_D <= exit_A per axis, in the reference's own arithmetic and without any error term.  The
// reference accepts a box iff  max3(entries) <= min3(exits)  and  max(x, y entries) < max_dis  and
// min(x, y exits) > min_dis  (its two early-outs and the final comparison are exactly that), so
//      accept(D, bound)  implies  accept(A, bound)  for every ancestor A of D.
// The reference reaches a leaf iff it accepts all interior ancestors, i.e. iff it accepts the leaf's PARENT box (leaves
// are never box-tested, src/bvh.cpp:38-48).  So only one exact decision per visited leaf is needed: the leaf array holds
// the parent's box in a header record in front of the leaf's triangles, and the lane evaluates box_decide() on it before
// it tries the triangles.  Everything above the leaf is navigation, for which any test that never skips a box the
// reference accepts (and that can hold an acceptable triangle) will do.
//
// Navigation nodes (DevScene::qnodes): a 4-wide tree collapsed from the reference's binary tree (a node's slots start as
// its two children; the interior slot with the largest surface is opened into its own children until there are four
// slots or only leaves are left), one 64-byte record per node: per slot the box as 16-bit
// coordinates on a grid over the scene bounds (origin q_org, spacing q_step per axis), rounded outwards by at least one
// grid step, and the reference of the slot's subtree:
//      4 x {c.x, c.y, c.z, ref},   c.x = lo.x | hi.x << 16  etc.,  for an unused slot a point box and the reference of an empty leaf (empty_slot_ref).
// With the exact decisions gone from the interior nodes the traversal kernels are bound by the latency of the dependent
// node fetches (a handful of waves per SIMD, each waiting for its next node), so the layout halves the number of
// fetches on a ray's way down: 16 bytes per box instead of 24 + reference, two tree levels per fetch.
//
// box_cheap: entry / exit estimates t = ((q_org + q * q_step) - o) / d from one convert and one fused multiply-add per
// plane, t = q * A + B with the per-ray constants A = q_step / d, B = (q_org - o) / d.  Against the exact value the
// estimate is off by at most 2^-22 S (roundings of 1/d, A, B and the difference) plus 2^-24 |t| (the multiply-add's own
// rounding), S = (|q_org - o| + extent) / |d| on that axis; the reference's own quotient differs from the exact one by
// another 2^-23 |t|.  Every plane of an axis lies inside the scene bounds, so |t| <= S there and all of it is below
// 1.75 * 2^-22 S: an absolute error on the scale of THAT axis (huge for an axis the ray is nearly parallel to, where the
// slab is just as wide).  It is folded into B per axis: the plane the ray meets first gets B - pad, the other one
// B + pad, pad = kCheapPad S with kCheapPad = 2^-20 (2.3 times the bound; the 16-bit grid step is 2^-16 of the extent) --
// every entry estimate is at or below the reference's quotient, every exit estimate at or above it.
constexpr float kCheapPad = 0x1p-20f;
// Relative width of the distance window inside which two candidates count as "equally near" for the purpose of
// deciding whether visit order can matter (ClosestLane in traverse_wave.h); also the margin by which box and triangle
// distances of the same geometry are assumed to agree.
#ifndef PT_NEAR_TIE_LOG2
#define PT_NEAR_TIE_LOG2 13
#endif
constexpr float kNearTie = 1.0f / float(1u << PT_NEAR_TIE_LOG2);

// per-ray constants of box_cheap.  Which of a slot's two planes on an axis the ray meets first is decided by the sign of
// the direction alone (lo <= hi on the grid, and the padding below keeps that order), so instead of a minimum and a
// maximum per axis and slot -- half-rate instructions on this chip, like the conversions -- the two 16-bit halves of the
// plane word are swapped for a negative direction (one rotate by `swap` = 0 or 16 bits): the entry distance then always
// comes from the low half, the exit distance from the high half.  The values are the ones min / max would have picked.
struct CheapRay {
    V3 a;       // q_step / d
    V3 b_in;    // addend of the plane the ray meets first:  (q_org - o) / d - pad
    V3 b_out;   // addend of the other plane:                (q_org - o) / d + pad
    PTD void set(const DevScene& sc, const RayPre& p) {
        const V3 org{sc.q_org[0], sc.q_org[1], sc.q_org[2]}, step{sc.q_step[0], sc.q_step[1], sc.q_step[2]};
        a = V3{step.x * p.y.x, step.y * p.y.y, step.z * p.y.z};
        const V3 rel = org - p.o;
        const V3 c{rel.x * p.y.x, rel.y * p.y.y, rel.z * p.y.z};
        const V3 e{kCheapPad * (fabsf(c.x) + 65535.0f * fabsf(a.x)), kCheapPad * (fabsf(c.y) + 65535.0f * fabsf(a.y)),
                   kCheapPad * (fabsf(c.z) + 65535.0f * fabsf(a.z))};
        b_in = c - e;
        b_out = c + e;
    }
    // rotate amounts that bring the first-met plane into the low half: 16 for a negative direction
    PTD uint32_t swap_x() const { return (as_u(a.x) >> 31) << 4; }
    PTD uint32_t swap_y() const { return (as_u(a.y) >> 31) << 4; }
    PTD uint32_t swap_z() const { return (as_u(a.z) >> 31) << 4; }
};

struct BoxC {
    float tn, tf;  // entry and exit distance over all three (widened) slabs
};

// the swaps of a ray, once per node
struct PlaneSwap {
    uint32_t x, y, z;
    PTD explicit PlaneSwap(const CheapRay& r) : x(r.swap_x()), y(r.swap_y()), z(r.swap_z()) {}
};

// cx, cy, cz: the three plane words of one child (lo in the low half, hi in the high half)
PTD BoxC box_cheap(uint32_t cx, uint32_t cy, uint32_t cz, const CheapRay& r, const PlaneSwap& sw) {
    cx = __builtin_amdgcn_alignbit(cx, cx, sw.x);
    cy = __builtin_amdgcn_alignbit(cy, cy, sw.y);
    cz = __builtin_amdgcn_alignbit(cz, cz, sw.z);
    const float x0 = __fmaf_rn(float(cx & 0xffffu), r.a.x, r.b_in.x), x1 = __fmaf_rn(float(cx >> 16), r.a.x, r.b_out.x);
    const float y0 = __fmaf_rn(float(cy & 0xffffu), r.a.y, r.b_in.y), y1 = __fmaf_rn(float(cy >> 16), r.a.y, r.b_out.y);
    const float z0 = __fmaf_rn(float(cz & 0xffffu), r.a.z, r.b_in.z), z1 = __fmaf_rn(float(cz >> 16), r.a.z, r.b_out.z);
    BoxC b;
    b.tn = fmaxf(fmaxf(x0, y0), z0);
    b.tf = fminf(fminf(x1, y1), z1);
    return b;
}
PTD BoxC box_cheap(uint32_t cx, uint32_t cy, uint32_t cz, const CheapRay& r) { return box_cheap(cx, cy, cz, r, PlaneSwap(r)); }

// Can the subtree of a box with estimates b hold a triangle that is accepted in [tmin, far]?  `far` is the far bound
// already widened by the caller (near-tie window).  Three conditions: the (widened) slabs overlap; the ray enters the box
// before the far bound; it leaves the box after its near end, with a margin of kNearTie on the scale of the entry
// distance (for a box that ends behind the ray's origin the entry is the larger of its two distances, the scale of the
// rounding errors of the triangle distances inside the box).
// As one comparison: the later of "enters the box" and "near end, less the margin" against the earlier of "leaves the
// box" and "far bound" (the near end never lies beyond the far bound, so measuring it against the far bound as well
// changes nothing).
PTD bool cheap_go(const BoxC& b, float tmin, float far) {
    return fmaxf(b.tn, __fmaf_rn(-kNearTie, fabsf(b.tn), tmin)) <= fminf(b.tf, far);
}

constexpr uint32_t kEmptyRef = 0xffffffffu;  // no navigation node (host side); unused slot of an 8-wide node

// 8-wide navigation nodes (PT_NAV_WIDTH = 8), 80 bytes = 5 x uint4.  A node's slots are the result of opening the
// interior slot with the largest surface until there are eight (or only leaves are left); their boxes are stored on an
// 8-bit grid over the NODE's own box -- origin p = the box's lower corner, spacing 2^k per axis -- rounded outwards, so
// deep nodes are resolved finer than a grid over the whole scene could, in half the bytes per slot:
//   v0 {bits(p.x) bits(p.y) bits(p.z)  ex | ey << 8 | ez << 16}      e = biased exponent of the spacing: 2^k = bits(e << 23)
//   v1 {child_base  leaf_base  meta[0..3]  meta[4..7]}
//   v2 {lo.x[0..3] lo.x[4..7] hi.x[0..3] hi.x[4..7]}   v3 the same for y   v4 for z      (one byte per slot)
// meta byte of a slot: 0 = unused; 0x80 = interior; otherwise the triangle count of a leaf (1..15).  References are
// implicit: the interior slots of a node are navigation nodes child_base, child_base + 1, ... in slot order, and its leaf
// slots lie one behind the other in the leaf array from record leaf_base on (one header record + count triangles each).
// Entry and exit estimates: t = ((p + q 2^k) - o) / d = q * a + b with a = 2^k / d (exact: a power of two times the
// ray's reciprocal) and b = (p - o) / d per node and axis; the padding of box_cheap applies unchanged, since p lies
// inside the scene bounds and the error terms are bounded on the scale of the scene's extent over |d|.
constexpr uint32_t kNav8Interior = 0x80u;

// The top of the navigation tree in LDS.  A CU serves per-lane 16-byte loads of different addresses at about one lane
// per clock, whatever cache level they hit in (measured: tools/gather_roof.hip; 64 lanes x 4 loads per node), and that
// rate is what the ordered traversal kernels run at.  LDS reads go through another path at the same time, so every node
// visit served from LDS is taken off the binding resource: each workgroup copies navigation nodes [0, kTreelet) -- the
// scene builder lists the most visited nodes first -- into LDS when it starts, and a lane whose current node is one of
// them reads it with four ds_read_b128.  Results do not depend on where a node's words come from.
#ifndef PT_TREELET
#define PT_TREELET 32
#endif
constexpr uint32_t kTreelet = PT_TREELET;
static_assert(kTreelet <= kTreeletMax, "the scene builder orders only the first kTreeletMax navigation nodes by visit probability");
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u4;

struct Treelet {
    const lds_u4* nodes;  // LDS copies of navigation nodes [0, n)
    uint32_t n;
};

// By all threads of a workgroup, before any of them traverses.  s_nodes: kTreelet * kNavVec4 uint4 of LDS.
PTD Treelet treelet_fill(const DevScene& sc, uint4* s_nodes) {
    const uint32_t n = min(kTreelet, sc.n_treelet);
    for (uint32_t i = threadIdx.x; i < n * kNavVec4; i += blockDim.x) s_nodes[i] = sc.qnodes[i];
    __syncthreads();
    return Treelet{(const lds_u4*)reinterpret_cast<u32x4*>(s_nodes), n};
}

// Diagnostic builds (-DPT_NODE_HIST): visits per navigation node (the last counter takes all nodes beyond the array).
#ifdef PT_NODE_HIST
constexpr uint32_t kNodeHist = 1u << 20;
__device__ uint32_t g_node_hist[kNodeHist];
#endif

// the four slots of 4-wide navigation node `cur`
PTD void nav_fetch(const DevScene& sc, const Treelet& tl, uint32_t cur, uint4& c0, uint4& c1, uint4& c2, uint4& c3) {
#ifdef PT_NODE_HIST
    atomicAdd(&g_node_hist[min(cur, kNodeHist - 1u)], 1u);
#endif
    if (cur < tl.n) {
        const lds_u4* w = tl.nodes + 4 * cur;
        const u32x4 a = w[0], b = w[1], c = w[2], d = w[3];
        c0 = make_uint4(a.x, a.y, a.z, a.w);
        c1 = make_uint4(b.x, b.y, b.z, b.w);
        c2 = make_uint4(c.x, c.y, c.z, c.w);
        c3 = make_uint4(d.x, d.y, d.z, d.w);
    } else {
        // a 32-bit byte offset from the (wave-uniform) array base: one shift instead of 64-bit address arithmetic per lane
        // (the scene builder refuses trees with 2^26 navigation nodes or more)
        const uint4* w = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.qnodes) + uint32_t(cur << 6));
        c0 = w[0];
        c1 = w[1];
        c2 = w[2];
        c3 = w[3];
    }
}

// the five quarters of 8-wide navigation node `cur`
struct Nav8 {
    uint4 v0, v1, v2, v3, v4;
    PTD void fetch(const DevScene& sc, const Treelet& tl, uint32_t cur) {
#ifdef PT_NODE_HIST
        atomicAdd(&g_node_hist[min(cur, kNodeHist - 1u)], 1u);
#endif
        if (cur < tl.n) {
            const lds_u4* w = tl.nodes + 5 * cur;
            const u32x4 a = w[0], b = w[1], c = w[2], d = w[3], e = w[4];
            v0 = make_uint4(a.x, a.y, a.z, a.w);
            v1 = make_uint4(b.x, b.y, b.z, b.w);
            v2 = make_uint4(c.x, c.y, c.z, c.w);
            v3 = make_uint4(d.x, d.y, d.z, d.w);
            v4 = make_uint4(e.x, e.y, e.z, e.w);
        } else {
            const uint4* w = sc.qnodes + 5 * size_t(cur);
            v0 = w[0];
            v1 = w[1];
            v2 = w[2];
            v3 = w[3];
            v4 = w[4];
        }
    }
};

// Per-ray constants of the 8-wide box test: the reciprocal direction and the per-axis padding of box_cheap, signed like
// the direction (d > 0: the lo plane is met first and gets b - pad, the hi plane b + pad).
struct CheapRay8 {
    V3 y, e;
    PTD void set(const DevScene& sc, const RayPre& p) {
        y = p.y;
        const V3 org{sc.q_org[0], sc.q_org[1], sc.q_org[2]}, step{sc.q_step[0], sc.q_step[1], sc.q_step[2]};
        const V3 rel = org - p.o;
        const V3 c{rel.x * p.y.x, rel.y * p.y.y, rel.z * p.y.z}, a{step.x * p.y.x, step.y * p.y.y, step.z * p.y.z};
        e = V3{copysignf(kCheapPad * (fabsf(c.x) + 65535.0f * fabsf(a.x)), p.d.x), copysignf(kCheapPad * (fabsf(c.y) + 65535.0f * fabsf(a.y)), p.d.y),
               copysignf(kCheapPad * (fabsf(c.z) + 65535.0f * fabsf(a.z)), p.d.z)};
    }
};

// per-node constants of the 8-wide box test
struct NodeFrame8 {
    V3 a, b_lo, b_hi;
    PTD NodeFrame8(const Nav8& n, const V3& o, const CheapRay8& r) {
        const V3 p{as_f(n.v0.x), as_f(n.v0.y), as_f(n.v0.z)};
        const float sx = as_f((n.v0.w & 0xffu) << 23), sy = as_f((n.v0.w & 0xff00u) << 15), sz = as_f((n.v0.w & 0xff0000u) << 7);
        a = V3{sx * r.y.x, sy * r.y.y, sz * r.y.z};
        const V3 rel = p - o;
        const V3 c{rel.x * r.y.x, rel.y * r.y.y, rel.z * r.y.z};
        b_lo = c - r.e;
        b_hi = c + r.e;
    }
};

PTD uint32_t byte_of(uint32_t w, int k) { return (w >> (8 * k)) & 0xffu; }

// slot k (0..7) of node n: its box estimates
PTD BoxC box_cheap8(const Nav8& n, const NodeFrame8& f, int k) {
    const uint32_t lx = k < 4 ? n.v2.x : n.v2.y, hx = k < 4 ? n.v2.z : n.v2.w;
    const uint32_t ly = k < 4 ? n.v3.x : n.v3.y, hy = k < 4 ? n.v3.z : n.v3.w;
    const uint32_t lz = k < 4 ? n.v4.x : n.v4.y, hz = k < 4 ? n.v4.z : n.v4.w;
    const int j = k & 3;
    const float x0 = __fmaf_rn(float(byte_of(lx, j)), f.a.x, f.b_lo.x), x1 = __fmaf_rn(float(byte_of(hx, j)), f.a.x, f.b_hi.x);
    const float y0 = __fmaf_rn(float(byte_of(ly, j)), f.a.y, f.b_lo.y), y1 = __fmaf_rn(float(byte_of(hy, j)), f.a.y, f.b_hi.y);
    const float z0 = __fmaf_rn(float(byte_of(lz, j)), f.a.z, f.b_lo.z), z1 = __fmaf_rn(float(byte_of(hz, j)), f.a.z, f.b_hi.z);
    BoxC b;
    b.tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    b.tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    return b;
}

// Header record of a leaf in the triangle array: {P.lo.xyz, P.hi.x} {P.hi.yz, bits(flags), 0} {unused}, P = the leaf's
// parent node.  kLeafUntested: the leaf is the root (the reference tests no box at all).
constexpr uint32_t kLeafUntested = 1u;

// The reference's decision on the parent box of the leaf at `slot`, with the ray's own far end as the bound (the bound
// the reference holds when it pops the parent is never larger).  entry = the distance at which the ray enters the box.
PTD bool leaf_parent_accepts(const DevScene& sc, uint32_t slot, const RayPre& pre, float tmax, float& entry) {
    const float4 h0 = sc.tris[3 * size_t(slot)], h1 = sc.tris[3 * size_t(slot) + 1];
    const BoxT b = box_decide<false>(h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, pre);
    const bool untested = (as_u(h1.z) & kLeafUntested) != 0u;
    // the entry over all three slabs: what a replay in reference order bounded by distance (closest_exact_t) culls with
    entry = untested ? -__builtin_huge_valf() : b.order;
    return untested || (b.overlap && b.tn < tmax);
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

// One triangle record held in registers.  pin() is an empty instruction sequence that names all twelve words as
// operands: the three 16-byte loads of fetch() then have to be issued before it, together, instead of the compiler
// postponing the first vertex until the determinant test has passed (a second dependent memory round trip per triangle).
struct TriRecord {
    float4 a, b, c;
    PTD void fetch(const DevScene& sc, uint32_t slot) {
        a = sc.tris[3 * size_t(slot) + 0];
        b = sc.tris[3 * size_t(slot) + 1];
        c = sc.tris[3 * size_t(slot) + 2];
    }
    PTD void pin() {
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(c.x), "+v"(c.y), "+v"(c.z));
    }
    PTD V3 p0() const { return V3{a.x, a.y, a.z}; }
    PTD V3 e1() const { return V3{b.x, b.y, b.z}; }
    PTD V3 e2() const { return V3{c.x, c.y, c.z}; }
    PTD int32_t prim() const { return int32_t(as_u(a.w)); }
};

// ------------------------------------------------------------------------------------------------ reference order

// `bound` < +inf is only used when an ordered pass has already established that the nearest accepted triangle lies at
// or before it: candidates and subtrees beyond the bound cannot influence which of the nearer candidates the reference
// ends up with (they never make it skip a box that contains a nearer candidate), so leaving them out reproduces the
// reference's result at a fraction of the work.  With bound = +inf this is the plain reference traversal.
// ROWS: rows of the [row][lane] LDS array at lds_stack that this walk may use (the rest of its stack is private memory).
template <bool SPECIAL, class C, int ROWS = kLdsDepth>
PTD bool closest_exact_t(const DevScene& sc, const RayD& ray, const RayPre& pre, float bound, HitD& hit, uint32_t* lds_stack, C& wc) {
    uint32_t spill[kMaxStack - ROWS];
    StackN<ROWS> st;
    st.init(lds_stack, spill);
    st.push(0u);
    float max_dis = fminf(ray.tmax, bound);
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
                if (tri_hit(p0, e1, e2, ray.o, ray.d, ray.tmin, max_dis, t, u, v)) {
                    max_dis = t;
                    found = true;
                    hit.prim = prim;
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                }
            }
        } else {
            const BoxT b = box_decide<SPECIAL>(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, pre);
            if (b.overlap && b.tn < max_dis && !(b.order > bound)) {
                st.push(first);
                st.push(first + 1u);
            }
        }
    }
    return found;
}

// Closest hit in the reference's visit order.  Returns true when something was hit.
template <class C, int ROWS = kLdsDepth>
PTD bool closest_exact(const DevScene& sc, const RayD& ray, HitD& hit, uint32_t* lds_stack, C& wc) {
    const RayPre pre = make_pre(sc, ray);
    if (pre.special) return closest_exact_t<true, C, ROWS>(sc, ray, pre, __builtin_huge_valf(), hit, lds_stack, wc);
    return closest_exact_t<false, C, ROWS>(sc, ray, pre, __builtin_huge_valf(), hit, lds_stack, wc);
}

template <bool SPECIAL, class C, int ROWS = kLdsDepth>
PTD bool any_exact_t(const DevScene& sc, const RayD& ray, const RayPre& pre, int32_t& first_prim, uint32_t* lds_stack, C& wc) {
    uint32_t spill[kMaxStack - ROWS];
    StackN<ROWS> st;
    st.init(lds_stack, spill);
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
                if (tri_hit(p0, e1, e2, ray.o, ray.d, ray.tmin, ray.tmax, t, u, v)) {
                    first_prim = prim;
                    return true;
                }
            }
        } else {
            const BoxT b = box_decide<SPECIAL>(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, pre);
            if (b.overlap && b.tn < ray.tmax) {
                st.push(first);
                st.push(first + 1u);
            }
        }
    }
    return false;
}

// Any hit in the reference's visit order; first_prim receives the first accepted triangle or -1.
template <class C, int ROWS = kLdsDepth>
PTD bool any_exact(const DevScene& sc, const RayD& ray, int32_t& first_prim, uint32_t* lds_stack, C& wc) {
    const RayPre pre = make_pre(sc, ray);
    if (pre.special) return any_exact_t<true, C, ROWS>(sc, ray, pre, first_prim, lds_stack, wc);
    return any_exact_t<false, C, ROWS>(sc, ray, pre, first_prim, lds_stack, wc);
}

// ------------------------------------------------------------------------------------------------ self test

// 1 when div_refine differs from the division instruction sequence for this operand pair (expected: never).
PTD uint32_t div5_mismatch(float n, float d) {
    if (!(fabsf(d) > kDirLo && fabsf(d) < kDirHi) || !(fabsf(n) < 2.0f * kCoordHi) || !(fabsf(n * (1.0f / d)) >= kTinyQ)) return 0u;
    const float y = 1.0f / d;
    const float a = div_refine(n * y, n, d, y), b = n / d;
    return (as_u(a) != as_u(b) && !(a == 0.0f && b == 0.0f)) ? 1u : 0u;
}

}  // namespace ptd
