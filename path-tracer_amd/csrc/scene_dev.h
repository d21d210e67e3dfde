// Device-resident scene: what every kernel receives by value.
#pragma once

#include "dev_math.h"

namespace ptd {

// Leaf reference in the 2-wide layout: bit 31 set, triangle count in bits 27..30, first leaf slot in bits 0..26.
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafCountShift = 27;
constexpr uint32_t kLeafFirstMask = (1u << kLeafCountShift) - 1u;

struct DevMaterial {
    float kd[3];
    float ks[3];
    float ns;
    int32_t tex;
};

struct DevLight {
    uint32_t tri;
    float lemit[3];
    float inv_area;  // float(1.0 / double(area)): include/pt/shape.h:37, src/shape.cpp:111
};

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
    // ---- traversal, ordered: one record per interior node, 4 x float4:
    //      {c0.lo.xyz, c0.hi.x} {c0.hi.yz, c1.lo.xy} {c1.lo.z, c1.hi.xyz} {bits(ref0), bits(ref1), 0, 0}
    const float4* wide;
    uint32_t wide_root;  // reference of the root (an interior index, or a leaf reference for tiny scenes)
    // ---- triangles in leaf order, 3 x float4 each: {p0.xyz, bits(prim)} {e1.xyz, 0} {e2.xyz, 0}
    const float4* tris;
    // ---- shading data (original triangle order)
    const float* verts;
    const float* normals;
    const float* uvs;
    const uint32_t* vi;
    const uint32_t* ni;
    const uint32_t* ti;
    const uint32_t* mat_id;
    const int32_t* light_id;
    const DevMaterial* mats;
    const DevLight* lights;
    const DevTexture* textures;
    const float* tex_data;
    uint32_t n_tris, n_normals, n_uvs, n_lights;
    DevCamera cam;
    // ---- Sobol' sampler: value(dim) = F[frame][dim] ^ X[dim][px] ^ Y[dim][py]  (all XOR-linear in the index bits)
    const uint32_t* sob_x;  // 1024 x scale
    const uint32_t* sob_y;  // 1024 x scale
    uint32_t sob_scale;     // roundUpPow2(max(W,H)) (src/sampler.cpp:92)
    float filter[33];       // tabulated Gaussian (src/block.cpp:14-25)
};

}  // namespace ptd
