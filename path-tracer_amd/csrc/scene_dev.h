// Device-resident scene: what every kernel receives by value.
#pragma once

#include "dev_math.h"

namespace ptd {

// Leaf reference in the navigation nodes: bit 31 set, triangle count in bits 27..30, the leaf's header slot in the leaf array
// in bits 0..26 (its triangles follow in the next `count` slots).
constexpr uint32_t kLeafBit = 0x80000000u;
// Unused slot of a 4-wide navigation node: the box is the single grid point 0 and the reference is a leaf with no
// triangles whose header is record `first` of the leaf array (any record but the last one and record 0: a triangle loop
// requests record first + 1 before it looks at the count, and first = 0 would read as "walk ended").  The kernels treat
// such a slot like any other: a ray that passes through that point of the scene's corner visits an empty leaf.
constexpr uint32_t empty_slot_ref(uint32_t first) { return kLeafBit | first; }
constexpr uint32_t kLeafCountShift = 27;
constexpr uint32_t kLeafFirstMask = (1u << kLeafCountShift) - 1u;
// Fat leaves: an interior node of the reference's tree whose two children are leaves with at most this many triangles
// between them is ONE leaf of the navigation structure (one header -- both children have the same parent box, which is
// all the exact decision at the end of a query needs -- followed by the triangles of both).  The reference's builder
// splits down to one or two triangles per leaf, mostly the two halves of a quad with nearly the same box: visiting the
// pair's node and then both leaves costs a navigation step and a leaf phase more than testing both triangles at once.
// Trying a triangle the reference's walk would not have reached is harmless: candidates are filtered by the decision on
// their leaf's parent box when the query ends.
#ifndef PT_FAT_LEAF_TRIS
#define PT_FAT_LEAF_TRIS 3
#endif
constexpr uint32_t kFatLeafTris = PT_FAT_LEAF_TRIS;
static_assert(kFatLeafTris <= 15u, "the leaf reference has four bits for the triangle count");

// Slots per navigation node of the ordered traversal: 4 (64-byte records, boxes on a 16-bit grid over the scene) or 8
// (80-byte records, boxes on an 8-bit grid over the node's own box).  See traverse.h.
#ifndef PT_NAV_WIDTH
#define PT_NAV_WIDTH 4
#endif
constexpr int kWide = PT_NAV_WIDTH;
static_assert(kWide == 4 || kWide == 8, "navigation nodes are 4- or 8-wide");
constexpr int kNavVec4 = kWide == 8 ? 5 : 4;  // uint4 per navigation node

// The first navigation nodes are the ones rays visit most (the root, then by surface); the traversal kernels serve a
// prefix of them from LDS.  kTreeletMax bounds what the scene builder orders that way.
constexpr uint32_t kTreeletMax = 1024;

// Shading records of one triangle (the reference chases Triangle* -> TriangleMesh -> index arrays -> vertex arrays, and
// Material*, AreaLight*: include/pt/shape.h:58-63, include/pt/mesh.h:31-55).  What every hit needs sits in ONE 64-byte
// sector, so that a hit costs one sector of memory traffic however large the scene is:
//   trec, 4 x float4:  r0 {v0.x v0.y v0.z v1.x}  r1 {v1.y v1.z v2.x v2.y}  r2 {v2.z bits(flags) bits(light_id) bits(material)}
//                      r3 {inv_area lemit.r lemit.g lemit.b}   (read for emitters only; zero otherwise)
//   tnrm, 3 x float4:  {n0.x n0.y n0.z n1.x} {n1.y n1.z n2.x n2.y} {n2.z 0 0 0}   (read for triangles with vertex normals)
//   mats, 2 x float4 per MATERIAL: {kd.r kd.g kd.b ns} {ks.r ks.g ks.b bits(tex)} -- a table of a few hundred bytes that
//                      stays in the nearest cache; its read depends on r2 and overlaps with the light record's
constexpr uint32_t kTriHasNormals = 1u;  // mesh has normals and this face's first normal index is non-zero (mesh.h:37-39)
constexpr uint32_t kTriHasUvs = 2u;      // same for texture coordinates (mesh.h:46-48)
constexpr int kTriRecVec4 = 4;
constexpr int kTriNormalVec4 = 3;
constexpr int kMaterialVec4 = 2;
// The primitive word of a leaf-array record (and of a hit-list entry) carries the triangle's index in its low bits and,
// above them, which of the optional quarters of its shading record a hit on it needs -- known together with the hit, so
// that the shade step can ask for exactly those in one round of loads.  Bit 31 stays clear: -1 means "no hit".
constexpr uint32_t kPrimMask = (1u << 29) - 1u;
constexpr uint32_t kPrimHasNormals = 1u << 29;
constexpr uint32_t kPrimEmitter = 1u << 30;
PTD int32_t prim_index(int32_t word) { return word < 0 ? word : int32_t(uint32_t(word) & kPrimMask); }

// Sampling record of one area light, 6 x float4 (AreaLight + its Triangle, src/light.cpp:10-33, src/shape.cpp:95-113):
//   l0 {v0.x v0.y v0.z v1.x}  l1 {v1.y v1.z v2.x v2.y}  l2 {v2.z n0.x n0.y n0.z}  l3 {n1.x n1.y n1.z n2.x}
//   l4 {n2.y n2.z bits(flags) inv_area}  l5 {lemit.r lemit.g lemit.b 0}
// inv_area = float(1.0 / double(area)) (include/pt/shape.h:37, src/shape.cpp:111).
constexpr int kLightRecVec4 = 6;

struct DevTexture {
    uint32_t w, h;
    uint64_t offset;  // in floats into tex_data
};

struct DevCamera {
    float s2c[16];
    float c2w[16];
    float eye[3];
    float proj_nume;
    uint32_t width, height;
};

struct DevScene {
    // ---- traversal, reference order: binary nodes, 2 x float4 each:
    //      {lo.x, lo.y, lo.z, hi.x} {hi.y, hi.z, bits(first), bits(count)}
    const float4* nodes;
    // ---- traversal, ordered: navigation nodes, kNavVec4 x uint4 each (see box_cheap in traverse.h).  4-wide:
    //      per slot {c.x, c.y, c.z, ref}, c.x = lo.x | hi.x << 16 on the grid q_org + q * q_step.  8-wide: see Nav8 there.
    const uint4* qnodes;
    float q_org[3], q_step[3];
    uint32_t wide_root;  // reference of the root (an interior index, or a leaf reference for tiny scenes)
    uint32_t n_treelet;  // navigation nodes [0, n_treelet) are listed most-visited first (at most kTreeletMax)
    // ---- leaf array, 3 x float4 per slot: per leaf one header slot {P.lo.xyz, P.hi.x} {P.hi.yz, bits(flags), 0} {0}
    //      (P = the leaf's parent node) followed by its triangles {p0.xyz, bits(prim)} {e1.xyz, 0} {e2.xyz, 0};
    //      a binary leaf node's `first` is the slot of its first triangle
    const float4* tris;
    // ---- shading records (original triangle order) and light records (reference light order)
    const float4* trec;
    const float4* tnrm;  // one record when no triangle has normals
    const float4* mats;
    const float4* lrec;
    // ---- only for textured materials
    const float* uvs;
    const uint32_t* ti;
    const DevTexture* textures;
    const float* tex_data;
    uint32_t n_tris, n_lights;
    uint32_t all_special;  // scene bounds too large for the fast slab arithmetic: every ray takes the plain division
    DevCamera cam;
    // ---- Sobol' sampler: word(dim) = F[frame][dim] ^ X[px][dim] ^ Y[py][dim]  (all XOR-linear in the index bits);
    //      rows of 1024 words, so the consecutive dimensions of one bounce sit in one cache line per table
    const uint32_t* sob_x;  // scale x 1024
    const uint32_t* sob_y;  // scale x 1024
    uint32_t sob_scale;     // roundUpPow2(max(W,H)) (src/sampler.cpp:92)
    float filter[33];       // tabulated Gaussian (src/block.cpp:14-25)
};

}  // namespace ptd
