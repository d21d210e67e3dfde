// Surface completion, Phong material, area lights, camera ray and the Sobol' sampler on the device.
// Restates (paths relative to the reference root): src/shape.cpp:95-166,186-189, src/material.cpp:9-115,
// include/pt/common.h:82-112, src/light.cpp:14-50, src/scene.cpp:213-220, src/camera.cpp:82-89,
// src/sampler.cpp:96-123, src/bitmap.cpp:146-166.
#pragma once

#include "scene_dev.h"

namespace ptd {

// ---------------------------------------------------------------------------------------------------- records

// What the shade step needs about the triangle it hit: the 64-byte record, the normals and the emitter quarter only when
// the hit's primitive word asks for them, and the material's two quarters from the material table.
struct TriRec {
    float4 r[kTriRecVec4];
    float4 nr[kTriNormalVec4];
    float4 m[kMaterialVec4];
    uint32_t index;  // the triangle's number (the primitive word without its flag bits)
    PTD void load(const DevScene& sc, uint32_t prim_word) {
        index = prim_word & kPrimMask;
        const float4* p = sc.trec + size_t(index) * kTriRecVec4;
        r[0] = p[0];
        r[1] = p[1];
        r[2] = p[2];
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        r[3] = nr[0] = nr[1] = nr[2] = zero;
        if (prim_word & kPrimEmitter) r[3] = p[3];
        if (prim_word & kPrimHasNormals) {
            const float4* q = sc.tnrm + size_t(index) * kTriNormalVec4;
            nr[0] = q[0];
            nr[1] = q[1];
            nr[2] = q[2];
        }
        const float4* mt = sc.mats + size_t(as_u(r[2].w)) * kMaterialVec4;
        m[0] = mt[0];
        m[1] = mt[1];
    }
    PTD V3 v0() const { return V3{r[0].x, r[0].y, r[0].z}; }
    PTD V3 v1() const { return V3{r[0].w, r[1].x, r[1].y}; }
    PTD V3 v2() const { return V3{r[1].z, r[1].w, r[2].x}; }
    PTD uint32_t flags() const { return as_u(r[2].y); }
    PTD int32_t light_id() const { return int32_t(as_u(r[2].z)); }
    PTD float inv_area() const { return r[3].x; }
    PTD V3 lemit() const { return V3{r[3].y, r[3].z, r[3].w}; }
    PTD V3 n0() const { return V3{nr[0].x, nr[0].y, nr[0].z}; }
    PTD V3 n1() const { return V3{nr[0].w, nr[1].x, nr[1].y}; }
    PTD V3 n2() const { return V3{nr[1].z, nr[1].w, nr[2].x}; }
    PTD V3 kd() const { return V3{m[0].x, m[0].y, m[0].z}; }
    PTD float ns() const { return m[0].w; }
    PTD V3 ks() const { return V3{m[1].x, m[1].y, m[1].z}; }
    PTD int32_t tex() const { return int32_t(as_u(m[1].w)); }
};

struct LightRec {
    float4 l[kLightRecVec4];
    PTD void load(const DevScene& sc, uint32_t li) {
        const float4* p = sc.lrec + size_t(li) * kLightRecVec4;
#pragma unroll
        for (int k = 0; k < kLightRecVec4; ++k) l[k] = p[k];
    }
    PTD V3 v0() const { return V3{l[0].x, l[0].y, l[0].z}; }
    PTD V3 v1() const { return V3{l[0].w, l[1].x, l[1].y}; }
    PTD V3 v2() const { return V3{l[1].z, l[1].w, l[2].x}; }
    PTD V3 n0() const { return V3{l[2].y, l[2].z, l[2].w}; }
    PTD V3 n1() const { return V3{l[3].x, l[3].y, l[3].z}; }
    PTD V3 n2() const { return V3{l[3].w, l[4].x, l[4].y}; }
    PTD uint32_t flags() const { return as_u(l[4].z); }
    PTD float inv_area() const { return l[4].w; }
    PTD V3 lemit() const { return V3{l[5].x, l[5].y, l[5].z}; }
};

// ---------------------------------------------------------------------------------------------------- surface point

struct Surf {
    V3 p, n, ng;
    Frame ts;
};

// Intersection::complete (src/shape.cpp:145-166) without the uv part; bary = (1 - u - v, u, v) (src/shape.cpp:86)
PTD void complete(const TriRec& t, float u, float v, Surf& o) {
    const float b0 = 1 - u - v, b1 = u, b2 = v;
    const V3 v0 = t.v0(), v1 = t.v1(), v2 = t.v2();
    o.p = v0 * b0 + v1 * b1 + v2 * b2;
    const V3 g = cross(v0 - v2, v1 - v2);
    o.ng = normalized(g);
    // without vertex normals the shading normal is the geometric one: the same vector normalised once, not twice
    o.n = o.ng;
    if (t.flags() & kTriHasNormals) o.n = normalized(t.n0() * b0 + t.n1() * b1 + t.n2() * b2);
    o.ts = Frame(o.n);
}

// uv of Intersection::complete (src/shape.cpp:163-165): only textured materials ever read it
PTD V2 surface_uv(const DevScene& sc, uint32_t prim, uint32_t flags, float u, float v) {
    const float b0 = 1 - u - v, b1 = u, b2 = v;
    V2 a{0.0f, 0.0f}, b{1.0f, 0.0f}, c{1.0f, 1.0f};
    if (flags & kTriHasUvs) {
        const uint32_t i0 = sc.ti[3 * size_t(prim)] - 1u, i1 = sc.ti[3 * size_t(prim) + 1] - 1u, i2 = sc.ti[3 * size_t(prim) + 2] - 1u;
        a = V2{sc.uvs[2 * size_t(i0)], sc.uvs[2 * size_t(i0) + 1]};
        b = V2{sc.uvs[2 * size_t(i1)], sc.uvs[2 * size_t(i1) + 1]};
        c = V2{sc.uvs[2 * size_t(i2)], sc.uvs[2 * size_t(i2) + 1]};
    }
    return V2{a.x * b0 + b.x * b1 + c.x * b2, a.y * b0 + b.y * b1 + c.y * b2};
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
};
// Material::getBaseColor(uv) (src/material.cpp:9-16) folded into the material the shade step works with
PTD Mat surface_material(const DevScene& sc, const TriRec& t, float u, float v) {
    Mat m{t.kd(), t.ks(), t.ns()};
    if (t.tex() >= 0) m.kd = tex_sample(sc, sc.textures[t.tex()], surface_uv(sc, t.index, t.flags(), u, v));
    return m;
}
PTD bool is_mirror(const Mat& m) { return m.ks.x > 999.0f || m.ks.y > 999.0f || m.ks.z > 999.0f; }
// mix() on floats (include/pt/common.h:82-85): a * (1.0 - f) is a double product, b * f a float product
PTD float mixf(float a, float b, float f) { return float(double(a) * (1.0 - double(f)) + double(b * f)); }
PTD V3 reflect(V3 w, V3 n) { return -w + (2.0f * dot(w, n)) * n; }  // include/pt/common.h:87-89
// cosf, sinf and powf of the host's C library (which the reference calls) are correctly rounded in 98.7 - 99.9 % of their
// results (measured against glibc 2.35); the device library's single-precision versions agree with them in 79 - 91 %.
// Evaluating in double and rounding once gives the correctly rounded float in all but a few results per million, so that
// nearly every bounce direction and BRDF value carries the bits the reference computes (a last-bit difference flips
// Russian-roulette and lobe decisions now and then, after which a path has nothing in common with the reference's).
// The double-precision kernels below are sized for that job only (relative error below 1e-13, arguments as the shading
// code produces them) and cost a third of the device library's general ones.
#ifndef PT_LIBM_F64
#define PT_LIBM_F64 2  // 0: device library in single precision, 1: device library in double, 2: the kernels below
#endif

// cos and sin of a float angle: quadrant by Cody-Waite with pi/2 in two pieces (exact up to |x| = 2^18: k * pi/2_hi is
// formed inside the fma and the difference is representable), Taylor polynomials to r^15 / r^14 on |r| <= pi/4
// (remainders below 5e-17 and 1.1e-15).  The shading code only forms angles in [0, 2 pi]; larger ones and NaN go to the
// device library's single-precision functions (kept small on purpose: the general double-precision reduction would
// sit in k_shade unused and cost it registers).
// (a real call returning the pair by value, like pow_k below: with it and the light sampling call k_shade needs 120
// registers instead of 165 and runs four waves per SIMD instead of three)
__device__ __attribute__((noinline)) float2 sincos_call(float xf);
PTD void sincos_r(float xf, float& sn, float& cs) {
    const float2 r = sincos_call(xf);
    sn = r.x;
    cs = r.y;
}
__device__ __attribute__((noinline)) float2 sincos_call(float xf) {
    float sn, cs;
    if (!(__builtin_fabsf(xf) <= 0x1p18f)) {
        sn = sinf(xf);
        cs = cosf(xf);
        return make_float2(sn, cs);
    }
    const double x = double(xf);
    const double k = __builtin_rint(x * 0.63661977236758134308);  // 2 / pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = -7.6471637318198164759e-13;                       // -1/15!
    ps = __builtin_fma(ps, z, 1.6059043836821614599e-10);          //  1/13!
    ps = __builtin_fma(ps, z, -2.5052108385441718775e-08);         // -1/11!
    ps = __builtin_fma(ps, z, 2.7557319223985890653e-06);          //  1/9!
    ps = __builtin_fma(ps, z, -1.9841269841269841270e-04);         // -1/7!
    ps = __builtin_fma(ps, z, 8.3333333333333333333e-03);          //  1/5!
    ps = __builtin_fma(ps, z, -1.6666666666666666667e-01);         // -1/3!
    const double s = __builtin_fma(r * z, ps, r);
    double pc = -1.1470745597729724714e-11;                       // -1/14!
    pc = __builtin_fma(pc, z, 2.0876756987868098979e-09);          //  1/12!
    pc = __builtin_fma(pc, z, -2.7557319223985890653e-07);         // -1/10!
    pc = __builtin_fma(pc, z, 2.4801587301587301587e-05);          //  1/8!
    pc = __builtin_fma(pc, z, -1.3888888888888888889e-03);         // -1/6!
    pc = __builtin_fma(pc, z, 4.1666666666666666667e-02);          //  1/4!
    pc = __builtin_fma(pc, z, -0.5);
    const double c = __builtin_fma(pc, z, 1.0);
    const int q = int(k) & 3;
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    sn = float((q & 2) ? -so : so);
    cs = float(((q + 1) & 2) ? -co : co);
    return make_float2(sn, cs);
}

// x^y for finite x > 0 and finite y as 2^(y log2 x): log through 2 atanh((m - 1) / (m + 1)) with the mantissa m in
// [sqrt(1/2), sqrt(2)) (series to s^20, remainder below 3e-17), 2^f on |f| <= 1/2 by the Taylor series of e^(f ln 2)
// to degree 13 (remainder below 4e-18); the one rounding to float also produces the denormal results.  Results near
// overflow, negative bases, -0 and non-finite operands (none of which the shading code forms) go to the device library's
// single-precision function.
// (a real call, not inlined: three inlined copies cost k_shade 56 registers and with them a third of its waves)
__device__ __attribute__((noinline)) float pow_k(float xf, float yf) {
    const double x = double(xf), y = double(yf);
    if (as_u(xf) == 0u && yf > 0.0f) return 0.0f;  // +0 to a positive power
    if (!(xf > 0.0f && xf < __builtin_huge_valf() && __builtin_fabsf(yf) < __builtin_huge_valf())) return powf(xf, yf);
    int e;
    double m = __builtin_frexp(x, &e);
    if (m < 0.70710678118654752440) {
        m *= 2.0;
        e -= 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    const double ln_m = 2.0 * __builtin_fma(s * z, p, s);
    const double l2 = __builtin_fma(ln_m, 1.44269504088896340736, double(e));  // log2 x
    const double w = y * l2;
    if (w < -150.5) return 0.0f;  // below half of the smallest denormal
    if (!(w < 127.0)) return powf(xf, yf);
    const double n = __builtin_rint(w);
    const double t = (w - n) * 0.69314718055994530942;
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, t, 1.0 / 479001600.0);
    q = __builtin_fma(q, t, 1.0 / 39916800.0);
    q = __builtin_fma(q, t, 1.0 / 3628800.0);
    q = __builtin_fma(q, t, 1.0 / 362880.0);
    q = __builtin_fma(q, t, 1.0 / 40320.0);
    q = __builtin_fma(q, t, 1.0 / 5040.0);
    q = __builtin_fma(q, t, 1.0 / 720.0);
    q = __builtin_fma(q, t, 1.0 / 120.0);
    q = __builtin_fma(q, t, 1.0 / 24.0);
    q = __builtin_fma(q, t, 1.0 / 6.0);
    q = __builtin_fma(q, t, 0.5);
    q = __builtin_fma(q, t, 1.0);
    q = __builtin_fma(q, t, 1.0);
    return float(__builtin_ldexp(q, int(n)));
}

#if PT_LIBM_F64 == 2
PTD void sincos_f(float x, float& s, float& c) { sincos_r(x, s, c); }
PTD float pow_r(float x, float y) { return pow_k(x, y); }
#elif PT_LIBM_F64 == 1
PTD void sincos_f(float x, float& s, float& c) {
    s = float(sin(double(x)));
    c = float(cos(double(x)));
}
PTD float pow_r(float x, float y) { return float(pow(double(x), double(y))); }
#else
PTD void sincos_f(float x, float& s, float& c) {
    s = sinf(x);
    c = cosf(x);
}
PTD float pow_r(float x, float y) { return powf(x, y); }
#endif
// squareToCosineHemisphere and the Phong lobe (include/pt/common.h:100-112) share their azimuth phi = 2 pi u.y: its sine
// and cosine are passed in, so that the two branches of sampleBRDF need them only once
PTD float azimuth(V2 u) { return 2.0f * kPi * u.y; }
PTD V3 cosine_hemisphere(V2 u, float sn, float cs) {  // include/pt/common.h:100-104
    const float su0 = sqrtf(u.x);
    return V3{su0 * cs, su0 * sn, sqrtf(1.0f - u.x)};
}
PTD V3 phong_lobe(V2 u, float sh, float sn, float cs) {  // include/pt/common.h:107-112
    const float ct = pow_r(u.x, 1.0f / (sh + 1.0f));
    const float st = sqrtf(1.0f - ct * ct);
    return V3{st * cs, st * sn, ct};
}

// Material::BRDF (src/material.cpp:18-38) and Material::pdf (src/material.cpp:92-115) for the same pair of directions
// (sampleLd evaluates both, src/integrator.cpp:131-137); the Phong power is shared.
PTD void brdf_eval_pdf(const Mat& m, V3 n, V3 wo, V3 wi, V3& f, float& pdf) {
    if (is_mirror(m)) {
        f = mk3(0.0f);
        pdf = 0.0f;
        return;
    }
    const V3 r = reflect(wo, n);
    const float cosRV = std_max(dot(wi, r), 0.0f);
    const float powRV = pow_r(cosRV, m.ns);
    const float normalization = (m.ns + 2.0f) * kInvTwoPi;
    f = m.kd * kInvPi + m.ks * normalization * powRV;
    const float sumKd = sum3(m.kd), sumKs = sum3(m.ks);
    const float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) {
        pdf = 0.0f;
        return;
    }
    const float specProb = sumKs / sumKdKs;
    const float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * powRV;
    const float pdf_diff = fabsf(dot(wi, n)) * kInvPi;
    pdf = mixf(pdf_diff, pdf_spec, specProb);
}

struct BSample {
    V3 wi, f;
    float pdf;
    bool specular;
};

// Material::sampleBRDF (src/material.cpp:40-90)
PTD BSample brdf_sample(const Mat& m, const Surf& it, V3 wo, float uc, V2 u) {
    BSample out{mk3(0.0f), mk3(0.0f), 0.0f, false};
    if (is_mirror(m)) {
        out.wi = reflect(wo, it.n);
        out.f = mk3(1.0f);
        out.specular = true;
        return out;
    }
    const float sumKd = sum3(m.kd), sumKs = sum3(m.ks);
    const float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) return out;
    const float specProb = sumKs / sumKdKs;
    const V3 r = reflect(wo, it.n);
    float sn, cs;
    sincos_f(azimuth(u), sn, cs);
    V3 wi;
    if (uc < specProb) {
        const Frame ts(r);
        wi = ts.to_world(phong_lobe(u, m.ns, sn, cs));
    } else {
        wi = it.ts.to_world(cosine_hemisphere(u, sn, cs));
    }
    wi = normalized(wi);
    const float cosTheta = dot(wi, it.n);
    if (cosTheta < 0.0f) return out;
    const float cosRV = std_max(dot(wi, r), 0.0f);
    const float powRV = pow_r(cosRV, m.ns);
    const float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * powRV;
    const float pdf_diff = cosTheta * kInvPi;
    out.f = m.kd * kInvPi + m.ks * (m.ns + 2.0f) * kInvTwoPi * powRV;
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
PTD LiSample light_sample(const LightRec& l, V3 sp, V3 sn, V2 u) {
    LiSample o{mk3(0.0f), mk3(0.0f), mk3(0.0f), mk3(0.0f), 0.0f};
    const float su0 = sqrtf(u.x);
    const float b0 = 1 - su0, b1 = u.y * su0;
    const float b2 = 1 - b0 - b1;
    const V3 v0 = l.v0(), v1 = l.v1(), v2 = l.v2();
    const V3 lp = v0 * b0 + v1 * b1 + v2 * b2;
    V3 ln;
    if (l.flags() & kTriHasNormals)
        ln = l.n0() * b0 + l.n1() * b1 + l.n2() * b2;
    else
        ln = cross(v0 - v2, v1 - v2);
    ln = normalized(ln);
    V3 wi = lp - sp;
    const float dist = norm(wi);
    wi = wi / dist;
    const float cos_lw = dot(ln, -wi);
    const float cos_sw = dot(sn, wi);
    if (cos_lw > 0.0f && cos_sw > 0.0f && dist > 0.0f) {
        o.L = l.lemit();
        o.wi = wi;
        o.p = lp;
        o.n = ln;
        o.pdfDir = l.inv_area() * dist * dist / cos_lw;
    }
    return o;
}

// The same as a real call that reads the light's record itself: the record's 24 registers are then held for the length
// of the call only, not from the moment the light is chosen (the record of one of a few lights comes from the nearest
// cache; with the sine / cosine call this takes k_shade from 165 to 120 registers, 3 to 4 waves per SIMD: -9 % of its
// time on the 1.6 M triangle interior, nothing on veach-mis).
__device__ __attribute__((noinline)) LiSample light_sample_at(const float4* rec, V3 sp, V3 sn, V2 u) {
    LightRec l;
#pragma unroll
    for (int k = 0; k < kLightRecVec4; ++k) l.l[k] = rec[k];
    return light_sample(l, sp, sn, u);
}

// AreaLight::pdfLi (src/light.cpp:46-50)
PTD float light_pdf(float inv_area, V3 lp, V3 ln, V3 ray_o, V3 ray_d) {
    const float dist = norm(lp - ray_o);
    const float cos_lw = dot(ln, -ray_d);
    return inv_area * dist * dist / cos_lw;
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
// src/sampler.cpp:16-36), so the word splits into three table look-ups:
//     word(dim) = F[sample number][dim] ^ X[px][dim] ^ Y[py][dim]
// Every table row holds the 1024 dimensions contiguously.  A bounce consumes up to 7 consecutive dimensions, so the
// sampler prefetches a window of 8 words per table (two 16-byte reads each) when it is started.
struct __attribute__((packed, aligned(4))) Words4 {
    uint32_t w[4];
};

PTD float sobol_to_float(uint32_t v) { return std_min(float(v) * 0x1p-32f, kOneMinusEps); }  // src/sampler.cpp:13

struct Sampler {
    const uint32_t* fr;  // F row of this sample number
    const uint32_t* xr;  // X row of this pixel column
    const uint32_t* yr;  // Y row of this pixel row
    uint32_t win[8];     // words of dimensions [base, base + 8)
    int base;
    int dim;
    PTD void start(const DevScene& sc, const uint32_t* frame_row, uint32_t px, uint32_t py, int first_dim) {
        fr = frame_row;
        xr = sc.sob_x + size_t(px) * 1024u;
        yr = sc.sob_y + size_t(py) * 1024u;
        dim = first_dim;
        base = -1;
        if (first_dim + 8 <= 1024) {
            base = first_dim;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Words4 a = *reinterpret_cast<const Words4*>(fr + first_dim + 4 * h);
                const Words4 b = *reinterpret_cast<const Words4*>(xr + first_dim + 4 * h);
                const Words4 c = *reinterpret_cast<const Words4*>(yr + first_dim + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) win[4 * h + k] = a.w[k] ^ b.w[k] ^ c.w[k];
            }
        }
    }
    PTD uint32_t word(int d) const {
        const int k = d - base;
        if (base >= 0 && k >= 0 && k < 8) {
            uint32_t v = win[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) v = (k == j) ? win[j] : v;
            return v;
        }
        return fr[d] ^ xr[d] ^ yr[d];
    }
    PTD float value(int d) const { return sobol_to_float(word(d)); }
    PTD float s1() {  // src/sampler.cpp:102-106
        if (dim >= 1024) dim = 2;
        return value(dim++);
    }
    PTD V2 s2() {  // src/sampler.cpp:108-114
        if (dim + 1 >= 1024) dim = 2;
        const V2 u{value(dim), value(dim + 1)};
        dim += 2;
        return u;
    }
};

// samplePixel2D (src/sampler.cpp:116-123): dimensions 0 and 1, remapped to the pixel
// the pixel's share of the first two Sobol' words: X[px] ^ Y[py] (the same for every sample number)
PTD uint2 pixel_words(const DevScene& sc, uint32_t px, uint32_t py) {
    const uint2 x = *reinterpret_cast<const uint2*>(sc.sob_x + size_t(px) * 1024u);
    const uint2 y = *reinterpret_cast<const uint2*>(sc.sob_y + size_t(py) * 1024u);
    return make_uint2(x.x ^ y.x, x.y ^ y.y);
}
PTD V2 pixel_jitter(const DevScene& sc, const uint32_t* frame_row, uint2 xy, uint32_t px, uint32_t py) {
    const uint2 f = *reinterpret_cast<const uint2*>(frame_row);
    V2 u{sobol_to_float(f.x ^ xy.x), sobol_to_float(f.y ^ xy.y)};
    u.x = std_clamp(u.x * float(int(sc.sob_scale)) - float(int(px)), 0.0f, kOneMinusEps);
    u.y = std_clamp(u.y * float(int(sc.sob_scale)) - float(int(py)), 0.0f, kOneMinusEps);
    return u;
}
PTD V2 pixel_jitter(const DevScene& sc, const uint32_t* frame_row, uint32_t px, uint32_t py) {
    return pixel_jitter(sc, frame_row, pixel_words(sc, px, py), px, py);
}

}  // namespace ptd
