// Surface completion, Phong material, area lights, camera ray and the Sobol' sampler on the device.
// Restates (paths relative to the reference root): src/shape.cpp:95-166,186-189, src/material.cpp:9-115,
// include/pt/common.h:82-112, src/light.cpp:14-50, src/scene.cpp:213-220, src/camera.cpp:82-89,
// src/sampler.cpp:96-123, src/bitmap.cpp:146-166.
#pragma once

#include "scene_dev.h"

namespace ptd {

// ---------------------------------------------------------------------------------------------------- mesh access

PTD V3 ld3(const float* p, uint32_t i) { return V3{p[3 * size_t(i)], p[3 * size_t(i) + 1], p[3 * size_t(i) + 2]}; }

PTD void tri_verts(const DevScene& sc, uint32_t f, V3& a, V3& b, V3& c) {  // include/pt/mesh.h:31-35
    a = ld3(sc.verts, sc.vi[3 * size_t(f) + 0] - 1u);
    b = ld3(sc.verts, sc.vi[3 * size_t(f) + 1] - 1u);
    c = ld3(sc.verts, sc.vi[3 * size_t(f) + 2] - 1u);
}
PTD bool tri_normals(const DevScene& sc, uint32_t f, V3& a, V3& b, V3& c) {  // include/pt/mesh.h:37-44
    if (sc.n_normals == 0u) return false;
    const uint32_t i0 = sc.ni[3 * size_t(f)];
    if (i0 == 0u) return false;
    a = ld3(sc.normals, i0 - 1u);
    b = ld3(sc.normals, sc.ni[3 * size_t(f) + 1] - 1u);
    c = ld3(sc.normals, sc.ni[3 * size_t(f) + 2] - 1u);
    return true;
}
PTD bool tri_uvs(const DevScene& sc, uint32_t f, V2& a, V2& b, V2& c) {  // include/pt/mesh.h:46-53
    if (sc.n_uvs == 0u) return false;
    const uint32_t i0 = sc.ti[3 * size_t(f)];
    if (i0 == 0u) return false;
    const uint32_t i1 = sc.ti[3 * size_t(f) + 1], i2 = sc.ti[3 * size_t(f) + 2];
    a = V2{sc.uvs[2 * size_t(i0 - 1u)], sc.uvs[2 * size_t(i0 - 1u) + 1]};
    b = V2{sc.uvs[2 * size_t(i1 - 1u)], sc.uvs[2 * size_t(i1 - 1u) + 1]};
    c = V2{sc.uvs[2 * size_t(i2 - 1u)], sc.uvs[2 * size_t(i2 - 1u) + 1]};
    return true;
}

// ---------------------------------------------------------------------------------------------------- surface point

struct Surf {
    V3 p, n, ng;
    V2 uv;
    Frame ts;
    uint32_t prim;
};

// Intersection::complete (src/shape.cpp:145-166); bary = (1 - u - v, u, v) (src/shape.cpp:86)
PTD void complete(const DevScene& sc, uint32_t prim, float u, float v, Surf& o) {
    const float b0 = 1 - u - v, b1 = u, b2 = v;
    V3 v0, v1, v2;
    tri_verts(sc, prim, v0, v1, v2);
    o.prim = prim;
    o.p = v0 * b0 + v1 * b1 + v2 * b2;
    const V3 g = cross(v0 - v2, v1 - v2);
    V3 n = g, n0, n1, n2;
    if (tri_normals(sc, prim, n0, n1, n2)) n = n0 * b0 + n1 * b1 + n2 * b2;
    o.n = normalized(n);
    o.ng = normalized(g);
    o.ts = Frame(o.n);
    V2 a{0.0f, 0.0f}, b{1.0f, 0.0f}, c{1.0f, 1.0f};
    tri_uvs(sc, prim, a, b, c);
    o.uv.x = a.x * b0 + b.x * b1 + c.x * b2;
    o.uv.y = a.y * b0 + b.y * b1 + c.y * b2;
}

// ---------------------------------------------------------------------------------------------------- texture

PTD V3 texel(const DevScene& sc, const DevTexture& t, int y, int x) {
    const float* p = sc.tex_data + t.offset + (size_t(y) * t.w + size_t(x)) * 3;
    return V3{p[0], p[1], p[2]};
}
// mix() on colours: the double weight is narrowed to float before multiplying (include/pt/common.h:82-85)
PTD V3 mix_rgb(V3 a, V3 b, float f) {
    const float w0 = float(1.0 - double(f));
    return a * w0 + b * f;
}
PTD V3 tex_sample(const DevScene& sc, const DevTexture& t, V2 uv) {  // src/bitmap.cpp:146-166
    const float u = std_clamp(uv.x, 0.0f, 1.0f);
    const float v = std_clamp(uv.y, 0.0f, 1.0f);
    const float ss = std_max(u * float(t.w) - 0.5f, 0.0f);
    const float tt = std_max(v * float(t.h) - 0.5f, 0.0f);
    const int x0 = int(ss), y0 = int(tt);
    const int x1 = min(x0 + 1, int(t.w) - 1), y1 = min(y0 + 1, int(t.h) - 1);
    const float alpha = ss - float(x0), beta = tt - float(y0);
    const V3 a = mix_rgb(texel(sc, t, y0, x0), texel(sc, t, y0, x1), alpha);
    const V3 b = mix_rgb(texel(sc, t, y1, x0), texel(sc, t, y1, x1), alpha);
    return mix_rgb(a, b, beta);
}

// ---------------------------------------------------------------------------------------------------- Phong material

struct Mat {
    V3 kd, ks;
    float ns;
    int32_t tex;
};
PTD Mat load_mat(const DevScene& sc, uint32_t prim) {
    const DevMaterial& m = sc.mats[sc.mat_id[prim]];
    return Mat{V3{m.kd[0], m.kd[1], m.kd[2]}, V3{m.ks[0], m.ks[1], m.ks[2]}, m.ns, m.tex};
}
PTD bool is_mirror(const Mat& m) { return m.ks.x > 999.0f || m.ks.y > 999.0f || m.ks.z > 999.0f; }
PTD V3 base_color(const DevScene& sc, const Mat& m, V2 uv) {  // src/material.cpp:9-16
    if (m.tex >= 0) return tex_sample(sc, sc.textures[m.tex], uv);
    return m.kd;
}
// mix() on floats (include/pt/common.h:82-85): a * (1.0 - f) is a double product, b * f a float product
PTD float mixf(float a, float b, float f) { return float(double(a) * (1.0 - double(f)) + double(b * f)); }
PTD V3 reflect(V3 w, V3 n) { return -w + (2.0f * dot(w, n)) * n; }  // include/pt/common.h:87-89
PTD V3 cosine_hemisphere(V2 u) {                                     // include/pt/common.h:100-104
    const float su0 = sqrtf(u.x);
    const float phi = 2.0f * kPi * u.y;
    return V3{su0 * cosf(phi), su0 * sinf(phi), sqrtf(1.0f - u.x)};
}
PTD V3 phong_lobe(V2 u, float sh) {  // include/pt/common.h:107-112
    const float ct = powf(u.x, 1.0f / (sh + 1.0f));
    const float st = sqrtf(1.0f - ct * ct);
    const float phi = 2.0f * kPi * u.y;
    return V3{st * cosf(phi), st * sinf(phi), ct};
}

// Material::BRDF (src/material.cpp:18-38); `kd` is the base colour already looked up at the hit
PTD V3 brdf_eval(const Mat& m, V3 kd, V3 n, V3 wo, V3 wi) {
    if (is_mirror(m)) return mk3(0.0f);
    const V3 diffuse = kd * kInvPi;
    const V3 r = reflect(wo, n);
    const float cosRV = std_max(dot(wi, r), 0.0f);
    const float normalization = (m.ns + 2.0f) * kInvTwoPi;
    const V3 spec = m.ks * normalization * powf(cosRV, m.ns);
    return diffuse + spec;
}

// Material::pdf (src/material.cpp:92-115)
PTD float brdf_pdf(const Mat& m, V3 kd, V3 n, V3 wo, V3 wi) {
    if (is_mirror(m)) return 0.0f;
    const float sumKd = sum3(kd), sumKs = sum3(m.ks);
    const float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) return 0.0f;
    const float specProb = sumKs / sumKdKs;
    const V3 r = reflect(wo, n);
    const float cosRV = std_max(dot(wi, r), 0.0f);
    const float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * powf(cosRV, m.ns);
    const float cosTheta = fabsf(dot(wi, n));
    const float pdf_diff = cosTheta * kInvPi;
    return mixf(pdf_diff, pdf_spec, specProb);
}

struct BSample {
    V3 wi, f;
    float pdf;
    bool specular;
};

// Material::sampleBRDF (src/material.cpp:40-90)
PTD BSample brdf_sample(const Mat& m, V3 kd, const Surf& it, V3 wo, float uc, V2 u) {
    BSample out{mk3(0.0f), mk3(0.0f), 0.0f, false};
    if (is_mirror(m)) {
        out.wi = reflect(wo, it.n);
        out.f = mk3(1.0f);
        out.specular = true;
        return out;
    }
    const float sumKd = sum3(kd), sumKs = sum3(m.ks);
    const float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) return out;
    const float specProb = sumKs / sumKdKs;
    const V3 r = reflect(wo, it.n);
    V3 wi;
    if (uc < specProb) {
        const Frame ts(r);
        wi = ts.to_world(phong_lobe(u, m.ns));
    } else {
        wi = it.ts.to_world(cosine_hemisphere(u));
    }
    wi = normalized(wi);
    const float cosTheta = dot(wi, it.n);
    if (cosTheta < 0.0f) return out;
    const float cosRV = std_max(dot(wi, r), 0.0f);
    const float powRV = powf(cosRV, m.ns);
    const float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * powRV;
    const float pdf_diff = cosTheta * kInvPi;
    out.f = kd * kInvPi + m.ks * (m.ns + 2.0f) * kInvTwoPi * powRV;
    out.pdf = mixf(pdf_diff, pdf_spec, specProb);
    out.wi = wi;
    return out;
}

// ---------------------------------------------------------------------------------------------------- area lights

struct LiSample {
    V3 L, wi, p, n;
    float pdfDir;
};

// AreaLight::sampleLi (src/light.cpp:18-33) with Triangle::sample (src/shape.cpp:95-113)
PTD LiSample light_sample(const DevScene& sc, const DevLight& l, V3 sp, V3 sn, V2 u) {
    LiSample o{mk3(0.0f), mk3(0.0f), mk3(0.0f), mk3(0.0f), 0.0f};
    const float su0 = sqrtf(u.x);
    const float b0 = 1 - su0, b1 = u.y * su0;
    const float b2 = 1 - b0 - b1;
    V3 v0, v1, v2, n0, n1, n2;
    tri_verts(sc, l.tri, v0, v1, v2);
    const V3 lp = v0 * b0 + v1 * b1 + v2 * b2;
    V3 ln;
    if (tri_normals(sc, l.tri, n0, n1, n2))
        ln = n0 * b0 + n1 * b1 + n2 * b2;
    else
        ln = cross(v0 - v2, v1 - v2);
    ln = normalized(ln);
    V3 wi = lp - sp;
    const float dist = norm(wi);
    wi = wi / dist;
    const float cos_lw = dot(ln, -wi);
    const float cos_sw = dot(sn, wi);
    if (cos_lw > 0.0f && cos_sw > 0.0f && dist > 0.0f) {
        o.L = V3{l.lemit[0], l.lemit[1], l.lemit[2]};
        o.wi = wi;
        o.p = lp;
        o.n = ln;
        o.pdfDir = l.inv_area * dist * dist / cos_lw;
    }
    return o;
}

// AreaLight::pdfLi (src/light.cpp:46-50)
PTD float light_pdf(const DevLight& l, V3 lp, V3 ln, V3 ray_o, V3 ray_d) {
    const float dist = norm(lp - ray_o);
    const float cos_lw = dot(ln, -ray_d);
    return l.inv_area * dist * dist / cos_lw;
}

PTD float power_heuristic(float f, float g) {  // src/integrator.cpp:15-18
    const float f2 = f * f, g2 = g * g;
    return f2 / (f2 + g2);
}

// ---------------------------------------------------------------------------------------------------- camera

PTD V3 xf_point(const float* m, V3 v) {  // Transform::apply(v, Scaler), include/pt/transform.h:29-32
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((m[i * 4 + 0] * v.x + m[i * 4 + 1] * v.y) + m[i * 4 + 2] * v.z) + m[i * 4 + 3] * 1.0f;
    return V3{r[0] / r[3], r[1] / r[3], r[2] / r[3]};
}
PTD V3 xf_vec(const float* m, V3 v) {  // Transform::apply(v, Vector), include/pt/transform.h:26-28
    return V3{(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}

// Camera::sampleRay (src/camera.cpp:82-89)
PTD void camera_ray(const DevCamera& c, V2 sp, V3& o, V3& d, float& tmin, float& tmax) {
    V3 dc = xf_point(c.s2c, V3{sp.x, sp.y, 0.5f});
    dc = normalized(dc);
    const float proj = c.proj_nume / dc.z;
    d = xf_vec(c.c2w, dc);
    o = V3{c.eye[0], c.eye[1], c.eye[2]};
    tmin = 0.01f * proj;
    tmax = 10000.0f * proj;
}

// ---------------------------------------------------------------------------------------------------- sampler

// SobolSampler (src/sampler.cpp:96-123).  The 32-bit Sobol' word of a dimension is XOR-linear in the bits of the
// sample index, and the index is XOR-linear in (sample number, pixel x, pixel y) (sobolIntervalToIndex,
// src/sampler.cpp:16-36), so the word splits into three table look-ups.  frame_tab points at this sample number's row
// of 1024 words.
struct Sampler {
    const uint32_t* frame_tab;
    const uint32_t* sx;
    const uint32_t* sy;
    uint32_t scale, px, py;
    int dim;
    PTD void start(const DevScene& sc, const uint32_t* frame_row, uint32_t x, uint32_t y, int first_dim) {
        frame_tab = frame_row;
        sx = sc.sob_x;
        sy = sc.sob_y;
        scale = sc.sob_scale;
        px = x;
        py = y;
        dim = first_dim;
    }
    PTD float value(int d) const {  // sobolSample, src/sampler.cpp:6-14
        const uint32_t v = frame_tab[d] ^ sx[size_t(d) * scale + px] ^ sy[size_t(d) * scale + py];
        return std_min(float(v) * 0x1p-32f, kOneMinusEps);
    }
    PTD float s1() {
        if (dim >= 1024) dim = 2;
        return value(dim++);
    }
    PTD V2 s2() {
        if (dim + 1 >= 1024) dim = 2;
        const V2 u{value(dim), value(dim + 1)};
        dim += 2;
        return u;
    }
    PTD V2 pixel2() const {  // samplePixel2D, src/sampler.cpp:116-123
        V2 u{value(0), value(1)};
        u.x = std_clamp(u.x * float(int(scale)) - float(int(px)), 0.0f, kOneMinusEps);
        u.y = std_clamp(u.y * float(int(scale)) - float(int(py)), 0.0f, kOneMinusEps);
        return u;
    }
};

}  // namespace ptd
