// Camera set-up on the host: restates Camera::Camera of the reference (src/camera.cpp:12-80).
// Only the three quantities Camera::sampleRay reads (src/camera.cpp:82-89) leave this file: sample2camera,
// camera2world and proj_nume.  All arithmetic is single precision except where the reference promotes to double.
#include <cmath>
#include <cstring>

#include "pt_host_internal.h"

namespace pt_host {

namespace {

constexpr float kPiF = 3.14159265358979323846f;  // include/pt/common.h:37 (a float constant in the reference)

struct Vec {
    float x, y, z;
};
inline Vec sub(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec crossp(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec unit(Vec a) {
    const float s = a.x * a.x + (a.y * a.y + a.z * a.z);
    if (s > 0.0f) {
        const float l = std::sqrt(s);
        return {a.x / l, a.y / l, a.z / l};
    }
    return a;
}

using Mat = float[16];  // row-major

void multiply(const Mat a, const Mat b, Mat out) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = a[4 * r] * b[c];
            for (int k = 1; k < 4; ++k) s += a[4 * r + k] * b[4 * k + c];
            out[4 * r + c] = s;
        }
}

// 3x3 minor of m with row r and column c removed
float minor3(const Mat m, int r, int c) {
    int rr[3], cc[3];
    for (int i = 0, k = 0; i < 4; ++i)
        if (i != r) rr[k++] = i;
    for (int i = 0, k = 0; i < 4; ++i)
        if (i != c) cc[k++] = i;
    auto e = [&](int i, int j) { return m[4 * rr[i] + cc[j]]; };
    return e(0, 0) * e(1, 1) * e(2, 2) - e(0, 0) * e(1, 2) * e(2, 1) - e(1, 0) * e(0, 1) * e(2, 2) + e(1, 0) * e(0, 2) * e(2, 1) +
           e(2, 0) * e(0, 1) * e(1, 2) - e(2, 0) * e(0, 2) * e(1, 1);
}

void invert(const Mat m, Mat out) {
    Mat adj;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            const float mn = minor3(m, c, r);  // adjugate = transposed cofactors
            adj[4 * r + c] = ((r + c) & 1) ? -mn : mn;
        }
    const float det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    for (int i = 0; i < 16; ++i) out[i] = adj[i] / det;
}

inline float deg2rad(float v) { return v * (kPiF / 180.0f); }  // src/camera.cpp:12-14

}  // namespace

void make_camera(uint32_t width, uint32_t height, float fovy, const float eye[3], const float lookat[3], const float up[3],
                 pt_camera& out) {
    out.width = width;
    out.height = height;
    std::memcpy(out.eye, eye, sizeof(float) * 3);

    // src/camera.cpp:20-31
    const Vec e{eye[0], eye[1], eye[2]}, l{lookat[0], lookat[1], lookat[2]}, u{up[0], up[1], up[2]};
    const Vec fwd = unit(sub(l, e));
    const Vec right = unit(crossp(fwd, u));
    const Vec up2 = crossp(right, fwd);
    const float c2w[16] = {-right.x, up2.x, fwd.x, e.x, -right.y, up2.y, fwd.y, e.y, -right.z, up2.z, fwd.z, e.z, 0.0f, 0.0f, 0.0f, 1.0f};
    std::memcpy(out.camera2world, c2w, sizeof(c2w));

    // src/camera.cpp:33-44, 55-65
    const float zn = 0.01f, zf = 10000.0f;
    const float aspect = float(width) / float(height);
    const float fovx = fovy * aspect;
    const float cotx = 1.0f / std::tan(deg2rad(float(0.5 * fovx)));
    const float coty = 1.0f / std::tan(deg2rad(float(0.5 * fovy)));
    const Mat proj = {cotx, 0, 0, 0, 0, coty, 0, 0, 0, 0, zf / (zf - zn), -(zf * zn) / (zf - zn), 0, 0, 1.0f, 0};
    const float hw = float(0.5 * float(width)), hh = float(0.5 * float(height));
    const Mat raster = {-hw, 0, 0, hw, 0, -hh, 0, hh, 0, 0, 1.0f, 0, 0, 0, 0, 1.0f};
    Mat c2s;
    multiply(raster, proj, c2s);
    invert(c2s, out.sample2camera);

    // src/camera.cpp:76-79: z of the unprojected image centre at depth 0.5
    const float* m = out.sample2camera;
    const float sx = 0.5f * width, sy = 0.5f * height, sz = 0.5f;
    const float z = ((m[8] * sx + m[9] * sy) + m[10] * sz) + m[11] * 1.0f;
    const float w = ((m[12] * sx + m[13] * sy) + m[14] * sz) + m[15] * 1.0f;
    out.proj_nume = float(1.0 / (z / w));
}

// Triangle::surfaceArea (src/shape.cpp:32-36)
float triangle_area(const float* verts, const uint32_t* vi, uint32_t tri) {
    const float* a = verts + 3 * size_t(vi[3 * tri + 0] - 1);
    const float* b = verts + 3 * size_t(vi[3 * tri + 1] - 1);
    const float* c = verts + 3 * size_t(vi[3 * tri + 2] - 1);
    const Vec ab{b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac{c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const Vec n = crossp(ab, ac);
    return 0.5f * std::sqrt(n.x * n.x + (n.y * n.y + n.z * n.z));
}

// ImageBlock's tabulated GaussianFilter(radius 2, stddev 0.5) (src/block.cpp:14-25, include/pt/filter.h:26-35)
void filter_table(float* out33) {
    const float radius = 2.0f, stddev = 0.5f;
    const float alpha = -1.0f / (2.0f * stddev * stddev);
    const float tail = std::exp(alpha * radius * radius);
    for (int i = 0; i < 32; ++i) {
        const float x = (radius * i) / 32;
        out33[i] = std::max(0.0f, std::exp(alpha * x * x) - tail);
    }
    out33[32] = 0.0f;
}

}  // namespace pt_host
