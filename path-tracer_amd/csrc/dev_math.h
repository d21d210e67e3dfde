// Device-side vector arithmetic with the evaluation order the reference's Eigen expressions have.
//
// The geometry results (hit index, t, barycentrics) must agree bit for bit with a CPU that runs the reference
// algorithm, so every expression here is written in the exact association it has there and the translation unit is
// compiled with -ffp-contract=off (no silent fused multiply-add).  Division and sqrt are IEEE-correct on gfx950 with
// hipcc's default flags.  std::min / std::max / std::clamp are spelled out as the comparisons they are, because the
// hardware min/max instructions treat NaN differently.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

#define PTD __device__ __forceinline__

PTD V3 mk3(float a, float b, float c) { return V3{a, b, c}; }
PTD V3 mk3(float a) { return V3{a, a, a}; }
PTD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
PTD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
PTD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
PTD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
PTD V3 cmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
// 3-element reductions split in halves: c0 + (c1 + c2)
PTD float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
PTD float sum3(V3 a) { return a.x + (a.y + a.z); }
PTD float sqnorm(V3 a) { return dot(a, a); }
PTD float norm(V3 a) { return sqrtf(sqnorm(a)); }
PTD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PTD V3 normalized(V3 a) {
    const float z = sqnorm(a);
    if (z > 0.0f) return a / sqrtf(z);
    return a;
}
PTD float std_max(float a, float b) { return (a < b) ? b : a; }
PTD float std_min(float a, float b) { return (b < a) ? b : a; }
PTD float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
PTD float maxc(V3 a) { return std_max(a.x, std_max(a.y, a.z)); }

constexpr float kEps = 1e-4f;  // include/pt/common.h:36
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kOneMinusEps = 0x1.fffffep-1f;  // include/pt/sampler.h:8

// Orthonormal basis around a unit normal (include/pt/tangent.h:12-33)
struct Frame {
    V3 t, b, n;
    PTD Frame() {}
    PTD explicit Frame(V3 nn) : n(nn) {
        const float sign = copysignf(1.0f, nn.z);
        const float a = -1.0f / (sign + nn.z);
        const float bb = nn.x * nn.y * a;
        t = V3{1.0f + sign * nn.x * nn.x * a, sign * bb, -sign * nn.x};
        b = V3{bb, sign + nn.y * nn.y * a, -nn.y};
    }
    PTD V3 to_world(V3 v) const { return t * v.x + b * v.y + n * v.z; }
};

PTD float as_f(uint32_t u) { return __uint_as_float(u); }
PTD uint32_t as_u(float f) { return __float_as_uint(f); }

}  // namespace ptd
