/*
 * pt_oracle.cpp -- CPU oracle: a scalar restatement of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  Nothing here is shared with the HIP product:
 * the two are written independently from the reference's source text and compared by tests.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 *
 * Arithmetic conventions (the reference computes with Eigen, which is absent from this image, so
 * these are decisions of this restatement -- "parity unpinned" at the Eigen boundary):
 *   - 3-vector dot / sum / squaredNorm reduce as c0 + (c1 + c2)   (Eigen's unrolled redux splits in halves)
 *   - cross is the textbook form; normalize divides each component by sqrt(squaredNorm) when it is > 0
 *   - a*s0 + b*s1 + c*s2 is ((a*s0) + (b*s1)) + (c*s2) per component
 *   - small matrix-vector products accumulate left to right
 *   - no fused multiply-add (build with -ffp-contract=off)
 *   - sampleBRDF(wo, sample1D(), sample2D()) draws the 2-D sample first (integrator.cpp:83; right-to-left
 *     argument evaluation, which is what MSVC and g++ do)
 */
#include "pt_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../build/sobol_tables.inc"

namespace {

/* ---------------------------------------------------------------- vectors */

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V2 { float x, y; };

inline V3 mk(float a, float b, float c) { return V3{a, b, c}; }
inline V3 mk(float a) { return V3{a, a, a}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sum3(V3 a) { return a.x + (a.y + a.z); }
inline float sqnorm(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(sqnorm(a)); }
inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 normalized(V3 a) {
    float z = sqnorm(a);
    if (z > 0.0f) return a / std::sqrt(z);
    return a;
}
inline float maxc(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
inline V3 vmin(V3 a, V3 b) { return mk(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return mk(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }

/* common.h:36-42 */
constexpr float kEps = 1e-4f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kOneMinusEps = 0x1.fffffep-1f; /* sampler.h:8 */

/* ---------------------------------------------------------------- ray / box */

struct Ray {
    V3 o, d;
    float tmin, tmax;
};

struct Box {
    V3 lo, hi;
    Box() : lo(mk(std::numeric_limits<float>::infinity())), hi(mk(std::numeric_limits<float>::lowest())) {} /* aabb.h:10-13 */
    void add(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void add(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    /* aabb.h:83-88 */
    float half_area() const {
        if (hi.x < lo.x) return 0.0f;
        float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
        return (x + y) * z + x * y;
    }
};

/* aabb.cpp:6-35 -- six real divisions; the z slab only rejects, it does not tighten the interval */
inline bool box_hit(const Box& b, const Ray& r) {
    float tmin = (b.lo.x - r.o.x) / r.d.x;
    float tmax = (b.hi.x - r.o.x) / r.d.x;
    if (tmin > tmax) std::swap(tmin, tmax);
    float tymin = (b.lo.y - r.o.y) / r.d.y;
    float tymax = (b.hi.y - r.o.y) / r.d.y;
    if (tymin > tymax) std::swap(tymin, tymax);
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b.lo.z - r.o.z) / r.d.z;
    float tzmax = (b.hi.z - r.o.z) / r.d.z;
    if (tzmin > tzmax) std::swap(tzmin, tzmax);
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return (tmin < r.tmax) && (tmax > r.tmin);
}

/* ---------------------------------------------------------------- tangent frame (tangent.h:12-33) */

struct Frame {
    V3 t, b, n;
    Frame() {}
    explicit Frame(V3 nn) : n(nn) {
        float sign = std::copysign(1.0f, nn.z);
        float a = -1.0f / (sign + nn.z);
        float bb = nn.x * nn.y * a;
        t = mk(1.0f + sign * nn.x * nn.x * a, sign * bb, -sign * nn.x);
        b = mk(bb, sign + nn.y * nn.y * a, -nn.y);
    }
    V3 to_world(V3 v) const { return t * v.x + b * v.y + n * v.z; }
};

/* ---------------------------------------------------------------- scene */

struct Node {
    Box box;
    size_t first, count; /* bvh.h:13-45 */
    bool leaf() const { return count != 0; }
};

struct Material {
    V3 kd, ks;
    float ns;
    int tex;
};

struct Texture {
    uint32_t w, h;
    std::vector<float> rgb;
};

struct Light {
    uint32_t tri;
    V3 lemit;
    float area;
};

struct Camera {
    uint32_t W, H;
    V3 eye;
    float s2c[16], c2w[16]; /* row-major */
    float proj_nume;
};

struct Counters {
    uint64_t closest = 0, any = 0, nodes = 0, tris = 0;
};
thread_local Counters tl_cnt;

} // namespace

struct orc_scene {
    std::vector<V3> verts, normals;
    std::vector<V2> uvs;
    std::vector<uint32_t> vi, ni, ti, mat_id;
    std::vector<int32_t> light_of_tri;
    std::vector<Material> mats;
    std::vector<Texture> texs;
    std::vector<Light> lights;
    Camera cam;
    std::vector<Node> nodes;
    std::vector<size_t> prim_ids;
    float filter[33];
    int sobol_scale, sobol_m;
    size_t n_tris() const { return mat_id.size(); }
};

namespace {

/* mesh.h:31-35 */
inline void tri_verts(const orc_scene& s, uint32_t f, V3& a, V3& b, V3& c) {
    a = s.verts[s.vi[3 * f + 0] - 1];
    b = s.verts[s.vi[3 * f + 1] - 1];
    c = s.verts[s.vi[3 * f + 2] - 1];
}
/* mesh.h:37-44 */
inline bool tri_normals(const orc_scene& s, uint32_t f, V3& a, V3& b, V3& c) {
    if (s.normals.empty()) return false;
    if (s.ni[3 * f] == 0) return false;
    a = s.normals[s.ni[3 * f + 0] - 1];
    b = s.normals[s.ni[3 * f + 1] - 1];
    c = s.normals[s.ni[3 * f + 2] - 1];
    return true;
}
/* mesh.h:46-53 */
inline bool tri_uvs(const orc_scene& s, uint32_t f, V2& a, V2& b, V2& c) {
    if (s.uvs.empty()) return false;
    if (s.ti[3 * f] == 0) return false;
    a = s.uvs[s.ti[3 * f + 0] - 1];
    b = s.uvs[s.ti[3 * f + 1] - 1];
    c = s.uvs[s.ti[3 * f + 2] - 1];
    return true;
}

/* shape.cpp:32-36 */
inline float tri_area(const orc_scene& s, uint32_t f) {
    V3 a, b, c;
    tri_verts(s, f, a, b, c);
    return 0.5f * norm(cross(b - a, c - a));
}

/* shape.cpp:50-89 */
inline bool tri_hit(const orc_scene& s, uint32_t f, const Ray& r, V3& bary, float& t) {
    V3 p0, p1, p2;
    tri_verts(s, f, p0, p1, p2);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(r.d, e2);
    float det = dot(e1, pvec);
    if (det > -1e-5f && det < 1e-5f) return false;
    float inv = 1.0f / det;
    V3 tvec = r.o - p0;
    float u = dot(tvec, pvec) * inv;
    if (u < 0.0 || u > 1.0) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(r.d, qvec) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    t = dot(e2, qvec) * inv;
    bary = mk(1 - u - v, u, v);
    return t >= r.tmin && t <= r.tmax;
}

/* ---------------------------------------------------------------- BVH build (bvh.cpp:13-24, 99-234; bvh.h:63-116) */

struct Builder {
    orc_scene& sc;
    std::vector<Box> boxes;
    std::vector<bool> marks;
    std::vector<float> accum;
    std::vector<size_t> ids[3];

    explicit Builder(orc_scene& s) : sc(s) {}

    Box range_box(size_t b, size_t e) const { /* bvh.h:87-92: always over axis-0 order */
        Box r;
        for (size_t i = b; i < e; ++i) r.add(boxes[ids[0][i]]);
        return r;
    }
    static float leaf_cost(size_t b, size_t e, const Box& bx) { return bx.half_area() * (e - b - 1.0f); } /* bvh.h:98-100 */

    struct Split {
        size_t pos;
        float cost;
        size_t axis;
    };

    /* bvh.cpp:201-234 */
    void best_split(size_t axis, size_t begin, size_t end, Split& best) {
        const std::vector<size_t>& id = ids[axis];
        size_t first_right = begin;
        Box rb;
        for (size_t i = end - 1; i > begin;) {
            size_t next = i - std::min(i - begin, size_t(32));
            float rc = 0.0f;
            for (; i > next; --i) {
                rb.add(boxes[id[i]]);
                accum[i] = rc = leaf_cost(i, end, rb);
            }
            if (rc > best.cost) {
                first_right = i;
                break;
            }
        }
        Box lb;
        for (size_t i = begin; i < first_right; ++i) lb.add(boxes[id[i]]);
        for (size_t i = first_right; i < end - 1; ++i) {
            lb.add(boxes[id[i]]);
            float lc = leaf_cost(begin, i + 1, lb);
            float c = lc + accum[i + 1];
            if (c < best.cost)
                best = Split{i + 1, c, axis};
            else if (lc > best.cost)
                break;
        }
    }

    /* bvh.cpp:168-199; returns 0 for "make a leaf" (a valid split position is never 0 relative to begin>=0 && pos>begin) */
    bool try_split(const Box& bx, size_t begin, size_t end, size_t& pos_out) {
        float no_split = bx.half_area() * (end - begin); /* bvh.h:94-96 */
        Split best{(begin + end + 1) / 2, no_split, 0};
        for (size_t a = 0; a < 3; ++a) best_split(a, begin, end, best);
        if (best.cost >= no_split) {
            if (end - begin <= 8) return false;
            best.pos = (begin + end + 1) / 2;
            best.axis = 0; /* AABB::getMaxAxis returns 0 on every branch (aabb.h:123-128) */
        }
        for (size_t i = begin; i < best.pos; ++i) marks[ids[best.axis][i]] = true; /* bvh.h:102-105 */
        for (size_t i = best.pos; i < end; ++i) marks[ids[best.axis][i]] = false;
        for (size_t a = 0; a < 3; ++a) {
            if (a == best.axis) continue;
            std::stable_partition(ids[a].begin() + begin, ids[a].begin() + end, [&](size_t i) { return bool(marks[i]); });
        }
        pos_out = best.pos;
        return true;
    }

    void run() {
        const size_t n = sc.n_tris();
        boxes.resize(n);
        std::vector<V3> centers(n);
        for (size_t i = 0; i < n; ++i) {
            V3 a, b, c;
            tri_verts(sc, (uint32_t)i, a, b, c);
            Box bx; /* aabb.h:27-30 */
            bx.lo = a; bx.hi = a;
            bx.add(b);
            bx.add(c);
            boxes[i] = bx;
            centers[i] = ((a + b) + c) / 3.0f; /* shape.cpp:44-48 */
        }
        marks.resize(n);
        accum.resize(n);
        for (int axis = 0; axis < 3; ++axis) {
            ids[axis].resize(n);
            for (size_t i = 0; i < n; ++i) ids[axis][i] = i;
            std::sort(ids[axis].begin(), ids[axis].end(), [&](size_t i, size_t j) { return centers[i][axis] < centers[j][axis]; });
        }
        std::vector<Node>& nodes = sc.nodes;
        nodes.clear();
        nodes.reserve(2 * n);
        nodes.emplace_back();
        nodes[0].box = range_box(0, n);
        nodes[0].first = nodes[0].count = 0;

        struct Item { size_t node, begin, end; size_t size() const { return end - begin; } };
        std::vector<Item> stack;
        stack.push_back(Item{0, 0, n});
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            if (it.size() > 1) { /* MinLeafSize = 1 */
                size_t pos;
                Box nb = nodes[it.node].box;
                if (try_split(nb, it.begin, it.end, pos)) {
                    size_t fc = nodes.size();
                    nodes[it.node].count = 0;
                    nodes[it.node].first = fc;
                    nodes.resize(fc + 2);
                    Box b0 = range_box(it.begin, pos), b1 = range_box(pos, it.end);
                    size_t r0b = it.begin, r0e = pos, r1b = pos, r1e = it.end;
                    if (b0.half_area() < b1.half_area()) { /* larger child first (bvh.cpp:137-144) */
                        std::swap(b0, b1);
                        std::swap(r0b, r1b);
                        std::swap(r0e, r1e);
                    }
                    Item i0{fc + 0, r0b, r0e}, i1{fc + 1, r1b, r1e};
                    nodes[fc + 0].box = b0;
                    nodes[fc + 1].box = b1;
                    if (i0.size() < i1.size()) std::swap(i0, i1);
                    stack.push_back(i0);
                    stack.push_back(i1);
                    continue;
                }
            }
            nodes[it.node].first = it.begin;
            nodes[it.node].count = it.size();
        }
        sc.prim_ids = ids[0];
        nodes.shrink_to_fit();
    }
};

/* ---------------------------------------------------------------- traversal (bvh.cpp:26-86) */

struct Hit {
    int32_t prim = -1;
    V3 bary;
    float t = 0.0f;
};

inline bool closest_hit(const orc_scene& s, const Ray& ray_in, Hit& h) {
    bool found = false;
    Ray ray = ray_in;
    tl_cnt.closest++;
    size_t stack[256];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const Node& nd = s.nodes[stack[--sp]];
        tl_cnt.nodes++;
        if (nd.leaf()) {
            for (size_t k = nd.first, i = 0; i < nd.count; ++k, ++i) {
                uint32_t prim = (uint32_t)s.prim_ids[k];
                V3 bary;
                float t;
                tl_cnt.tris++;
                if (tri_hit(s, prim, ray, bary, t)) {
                    ray.tmax = t;
                    found = true;
                    h.prim = (int32_t)prim;
                    h.bary = bary;
                    h.t = t;
                }
            }
        } else if (box_hit(nd.box, ray)) {
            stack[sp++] = nd.first;
            stack[sp++] = nd.first + 1;
        }
    }
    return found;
}

inline bool any_hit(const orc_scene& s, const Ray& ray, int32_t* first_prim) {
    tl_cnt.any++;
    size_t stack[256];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const Node& nd = s.nodes[stack[--sp]];
        tl_cnt.nodes++;
        if (nd.leaf()) {
            for (size_t k = nd.first, i = 0; i < nd.count; ++k, ++i) {
                uint32_t prim = (uint32_t)s.prim_ids[k];
                V3 bary;
                float t;
                tl_cnt.tris++;
                if (tri_hit(s, prim, ray, bary, t)) {
                    if (first_prim) *first_prim = (int32_t)prim;
                    return true;
                }
            }
        } else if (box_hit(nd.box, ray)) {
            stack[sp++] = nd.first;
            stack[sp++] = nd.first + 1;
        }
    }
    if (first_prim) *first_prim = -1;
    return false;
}

/* ---------------------------------------------------------------- surface point (shape.cpp:145-166) */

struct Surf {
    V3 p, n, ng;
    V2 uv;
    Frame ts;
    uint32_t prim;
};

inline void complete(const orc_scene& s, uint32_t prim, V3 bary, Surf& o) {
    V3 v0, v1, v2;
    tri_verts(s, prim, v0, v1, v2);
    o.prim = prim;
    o.p = v0 * bary.x + v1 * bary.y + v2 * bary.z;
    V3 n0, n1, n2;
    V3 g = cross(v0 - v2, v1 - v2);
    V3 n = g;
    if (tri_normals(s, prim, n0, n1, n2)) n = n0 * bary.x + n1 * bary.y + n2 * bary.z;
    o.n = normalized(n);
    o.ng = normalized(g);
    o.ts = Frame(o.n);
    V2 a{0.0f, 0.0f}, b{1.0f, 0.0f}, c{1.0f, 1.0f};
    tri_uvs(s, prim, a, b, c);
    o.uv.x = a.x * bary.x + b.x * bary.y + c.x * bary.z;
    o.uv.y = a.y * bary.x + b.y * bary.y + c.y * bary.z;
}

/* ---------------------------------------------------------------- texture (bitmap.cpp:146-166) */

inline V3 texel(const Texture& t, int y, int x) {
    const float* p = &t.rgb[(size_t(y) * t.w + x) * 3];
    return mk(p[0], p[1], p[2]);
}
/* mix on a colour: the double weight (1.0 - f) is narrowed to float before the multiply (common.h:82-85 with Eigen arrays) */
inline V3 mix_rgb(V3 a, V3 b, float f) {
    float w0 = (float)(1.0 - (double)f);
    return a * w0 + b * f;
}
inline V3 tex_sample(const Texture& t, V2 uv) {
    float u = std::clamp(uv.x, 0.0f, 1.0f);
    float v = std::clamp(uv.y, 0.0f, 1.0f);
    float ss = std::max(u * t.w - 0.5f, 0.0f);
    float tt = std::max(v * t.h - 0.5f, 0.0f);
    int x0 = (int)ss, y0 = (int)tt;
    int x1 = std::min(x0 + 1, int(t.w - 1));
    int y1 = std::min(y0 + 1, int(t.h - 1));
    float alpha = ss - x0, beta = tt - y0;
    V3 a = mix_rgb(texel(t, y0, x0), texel(t, y0, x1), alpha);
    V3 b = mix_rgb(texel(t, y1, x0), texel(t, y1, x1), alpha);
    return mix_rgb(a, b, beta);
}

/* ---------------------------------------------------------------- Phong material (material.cpp, common.h:82-112) */

inline float mixf(float a, float b, float f) { return (float)(a * (1.0 - f) + b * f); } /* double arithmetic, common.h:82-85 */
inline V3 reflect(V3 w, V3 n) { return -w + (2.0f * dot(w, n)) * n; }                  /* common.h:87-89 */
inline V3 cosine_hemisphere(V2 u) {                                                    /* common.h:100-104 */
    float su0 = std::sqrt(u.x);
    float phi = 2.0f * kPi * u.y;
    return mk(su0 * std::cos(phi), su0 * std::sin(phi), std::sqrt(1.0f - u.x));
}
inline V3 phong_lobe(V2 u, float sh) { /* common.h:107-112 */
    float ct = std::pow(u.x, 1.0f / (sh + 1.0f));
    float st = std::sqrt(1.0f - ct * ct);
    float phi = 2.0f * kPi * u.y;
    return mk(st * std::cos(phi), st * std::sin(phi), ct);
}

inline bool is_mirror(const Material& m) { return m.ks.x > 999 || m.ks.y > 999 || m.ks.z > 999; }

inline V3 base_color(const orc_scene& s, const Material& m, V2 uv) { /* material.cpp:9-16 */
    if (m.tex >= 0) return tex_sample(s.texs[m.tex], uv);
    return m.kd;
}

inline V3 brdf_eval(const orc_scene& s, const Surf& it, V3 wo, V3 wi) { /* material.cpp:18-38 */
    const Material& m = s.mats[s.mat_id[it.prim]];
    if (is_mirror(m)) return mk(0.0f);
    V3 diffuse = base_color(s, m, it.uv) * kInvPi;
    V3 r = reflect(wo, it.n);
    float cosRV = std::max(dot(wi, r), 0.0f);
    float normalization = (m.ns + 2.0f) * kInvTwoPi;
    V3 spec = m.ks * normalization * std::pow(cosRV, m.ns);
    return diffuse + spec;
}

inline float brdf_pdf(const orc_scene& s, const Surf& it, V3 wo, V3 wi) { /* material.cpp:92-115 */
    const Material& m = s.mats[s.mat_id[it.prim]];
    if (is_mirror(m)) return 0.0f;
    V3 diffuse = base_color(s, m, it.uv);
    float sumKd = sum3(diffuse), sumKs = sum3(m.ks);
    float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) return 0.0f;
    float specProb = sumKs / sumKdKs;
    V3 r = reflect(wo, it.n);
    float cosRV = std::max(dot(wi, r), 0.0f);
    float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * std::pow(cosRV, m.ns);
    float cosTheta = std::abs(dot(wi, it.n));
    float pdf_diff = cosTheta * kInvPi;
    return mixf(pdf_diff, pdf_spec, specProb);
}

struct BSample {
    V3 wi = mk(0.0f);
    float pdf = 0.0f;
    V3 f = mk(0.0f);
    bool specular = false;
};

inline BSample brdf_sample(const orc_scene& s, const Surf& it, V3 wo, float uc, V2 u) { /* material.cpp:40-90 */
    const Material& m = s.mats[s.mat_id[it.prim]];
    BSample out;
    if (is_mirror(m)) {
        out.wi = reflect(wo, it.n);
        out.pdf = 0.0f;
        out.f = mk(1.0f);
        out.specular = true;
        return out;
    }
    V3 diffuse = base_color(s, m, it.uv);
    float sumKd = sum3(diffuse), sumKs = sum3(m.ks);
    float sumKdKs = sumKd + sumKs;
    if (sumKdKs == 0.0f) return out;
    float specProb = sumKs / sumKdKs;
    V3 wi;
    V3 r = reflect(wo, it.n);
    if (uc < specProb) {
        Frame ts(r);
        wi = ts.to_world(phong_lobe(u, m.ns));
    } else {
        wi = it.ts.to_world(cosine_hemisphere(u));
    }
    wi = normalized(wi);
    float cosTheta = dot(wi, it.n);
    if (cosTheta < 0.0f) return out;
    float cosRV = std::max(dot(wi, r), 0.0f);
    float powRV = std::pow(cosRV, m.ns);
    float pdf_spec = (m.ns + 1.0f) * kInvTwoPi * powRV;
    float pdf_diff = cosTheta * kInvPi;
    out.f = diffuse * kInvPi + m.ks * (m.ns + 2.0f) * kInvTwoPi * powRV;
    out.pdf = mixf(pdf_diff, pdf_spec, specProb);
    out.wi = wi;
    return out;
}

/* ---------------------------------------------------------------- area lights (light.cpp:14-50, shape.cpp:95-113) */

struct LiSample {
    V3 L = mk(0.0f), wi = mk(0.0f), p = mk(0.0f), n = mk(0.0f);
    float pdfArea = 0.0f, pdfDir = 0.0f;
};

inline void tri_sample(const orc_scene& s, uint32_t prim, V2 u, V3& p, V3& n, float& pdf) {
    float su0 = std::sqrt(u.x);
    float b0 = 1 - su0, b1 = u.y * su0;
    float b2 = 1 - b0 - b1;
    V3 v0, v1, v2, n0, n1, n2;
    tri_verts(s, prim, v0, v1, v2);
    p = v0 * b0 + v1 * b1 + v2 * b2;
    if (tri_normals(s, prim, n0, n1, n2))
        n = n0 * b0 + n1 * b1 + n2 * b2;
    else
        n = cross(v0 - v2, v1 - v2);
    n = normalized(n);
    pdf = (float)(1.0 / tri_area(s, prim));
}

inline LiSample light_sample(const orc_scene& s, const Light& l, const Surf& it, V2 u) {
    LiSample o;
    V3 lp, ln;
    float pdfA;
    tri_sample(s, l.tri, u, lp, ln, pdfA);
    V3 wi = lp - it.p;
    float dist = norm(wi);
    wi = wi / dist;
    float cos_lw = dot(ln, -wi);
    float cos_sw = dot(it.n, wi);
    if (cos_lw > 0.0 && cos_sw > 0.0 && dist > 0.0) {
        o.L = l.lemit;
        o.wi = wi;
        o.p = lp;
        o.n = ln;
        o.pdfArea = pdfA;
        o.pdfDir = pdfA * dist * dist / cos_lw;
    }
    return o;
}

inline float light_pdf(const orc_scene& s, const Light& l, const Surf& lit, const Ray& ray) {
    float dist = norm(lit.p - ray.o);
    float cos_lw = dot(lit.n, -ray.d);
    float pdf = (float)(1.0 / tri_area(s, l.tri)); /* shape.h:37 */
    return pdf * dist * dist / cos_lw;
}

/* scene.cpp:213-220 */
inline Ray shadow_ray(V3 p0, V3 p1, V3 n0, V3 n1) {
    p0 = p0 + n0 * kEps;
    p1 = p1 + n1 * kEps;
    V3 d = p1 - p0;
    float dist = norm(d);
    return Ray{p0, d / dist, 0.0f, dist * (1 - kEps)};
}

/* ---------------------------------------------------------------- camera (camera.cpp:12-89, transform.h:25-38) */

inline void mat4_mul(const float* a, const float* b, float* o) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a[i * 4 + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; ++k) acc += a[i * 4 + k] * b[k * 4 + j];
            o[i * 4 + j] = acc;
        }
}

/* cofactor inverse in float */
inline void mat4_inv(const float* m, float* o) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    for (int i = 0; i < 16; ++i) o[i] = inv[i] / det;
}

/* Transform::apply(vec, Scaler): homogeneous point with perspective divide (transform.h:29-32) */
inline V3 xf_point(const float* m, V3 v) {
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = ((m[i * 4 + 0] * v.x + m[i * 4 + 1] * v.y) + m[i * 4 + 2] * v.z) + m[i * 4 + 3] * 1.0f;
    return mk(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}
/* Transform::apply(vec, Vector): upper-left 3x3 (transform.h:26-28) */
inline V3 xf_vec(const float* m, V3 v) {
    return mk((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}

inline float deg2rad(float v) { return v * (kPi / 180.0f); }

void camera_setup(Camera& c, uint32_t W, uint32_t H, float fovy, V3 eye, V3 lookat, V3 up) {
    c.W = W;
    c.H = H;
    c.eye = eye;
    float aspect = float(W) / float(H);
    V3 forward = normalized(lookat - eye);
    V3 right = normalized(cross(forward, up));
    V3 newUp = cross(right, forward);
    float c2w[16] = {-right.x, newUp.x, forward.x, eye.x, -right.y, newUp.y, forward.y, eye.y, -right.z, newUp.z, forward.z, eye.z, 0, 0, 0, 1};
    std::memcpy(c.c2w, c2w, sizeof(c2w));
    const float nearz = 0.01f, farz = 10000.0f;
    float fovx = fovy * aspect;
    float cotx = 1.0f / std::tan(deg2rad((float)(0.5 * fovx)));
    float coty = 1.0f / std::tan(deg2rad((float)(0.5 * fovy)));
    float proj[16] = {cotx, 0, 0, 0, 0, coty, 0, 0, 0, 0, farz / (farz - nearz), -(farz * nearz) / (farz - nearz), 0, 0, 1, 0};
    float hw = (float)(0.5 * float(W)), hh = (float)(0.5 * float(H));
    float ndc2pix[16] = {-hw, 0, 0, hw, 0, -hh, 0, hh, 0, 0, 1, 0, 0, 0, 0, 1};
    float c2s[16];
    mat4_mul(ndc2pix, proj, c2s);
    mat4_inv(c2s, c.s2c);
    V3 tmp = xf_point(c.s2c, mk(0.5f * W, 0.5f * H, 0.5f));
    c.proj_nume = (float)(1.0 / tmp.z);
}

inline Ray camera_ray(const Camera& c, V2 sp) { /* camera.cpp:82-89 */
    V3 d = xf_point(c.s2c, mk(sp.x, sp.y, 0.5f));
    d = normalized(d);
    float proj = c.proj_nume / d.z;
    d = xf_vec(c.c2w, d);
    return Ray{c.eye, d, 0.01f * proj, 10000.0f * proj};
}

/* ---------------------------------------------------------------- Sobol' sampler (sampler.cpp) */

inline float sobol_sample(int64_t a, int dim) { /* sampler.cpp:6-14 */
    uint32_t v = 0;
    for (int i = dim * 52; a != 0; a >>= 1, i++)
        if (a & 1) v ^= kSobolMatrices32[i];
    return std::min(v * 0x1p-32f, kOneMinusEps);
}

inline uint64_t sobol_interval_to_index(uint32_t m, uint64_t frame, int32_t px, int32_t py) { /* sampler.cpp:16-36 */
    if (m == 0) return frame;
    const uint32_t m2 = m << 1;
    uint64_t index = uint64_t(frame) << m2;
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= kVdCSobolMatrices[m - 1][c];
    uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= kVdCSobolMatricesInv[m - 1][c];
    return index;
}

inline int32_t round_up_pow2(int32_t v) { /* sampler.cpp:68-76 */
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}
inline int log2int(float v) { /* sampler.cpp:55-66 */
    if (v < 1) return -log2int(1 / v);
    uint32_t b;
    std::memcpy(&b, &v, 4);
    const uint32_t midsignif = 0b00000000001101010000010011110011;
    return int(b >> 23) - 127 + (((b & ((1u << 23) - 1)) >= midsignif) ? 1 : 0);
}

struct Sampler {
    int64_t index = 0;
    int dim = 0, scale = 1;
    int32_t px = 0, py = 0;
    void start(int32_t x, int32_t y, int sample) { /* sampler.cpp:96-100 */
        px = x; py = y;
        dim = 2;
        index = (int64_t)sobol_interval_to_index((uint32_t)log2int((float)scale), (uint64_t)sample, px, py);
    }
    float s1() { /* sampler.cpp:102-106 */
        if (dim >= 1024) dim = 2;
        return sobol_sample(index, dim++);
    }
    V2 s2() { /* sampler.cpp:108-114 */
        if (dim + 1 >= 1024) dim = 2;
        V2 u{sobol_sample(index, dim), sobol_sample(index, dim + 1)};
        dim += 2;
        return u;
    }
    V2 pixel2() { /* sampler.cpp:116-123 */
        V2 u{sobol_sample(index, 0), sobol_sample(index, 1)};
        u.x = std::clamp(u.x * scale - px, 0.0f, kOneMinusEps);
        u.y = std::clamp(u.y * scale - py, 0.0f, kOneMinusEps);
        return u;
    }
};

/* ---------------------------------------------------------------- integrators (integrator.cpp) */

struct RayLog {
    float* buf;
    size_t cap, n;
    void put(const Ray& r, int kind) {
        if (!buf) return;
        if (n < cap) {
            float* p = buf + 9 * n;
            p[0] = r.o.x; p[1] = r.o.y; p[2] = r.o.z; p[3] = r.d.x; p[4] = r.d.y; p[5] = r.d.z; p[6] = r.tmin; p[7] = r.tmax; p[8] = (float)kind;
        }
        n++;
    }
};

inline float power_heuristic(float f, float g) { /* integrator.cpp:15-18 */
    float f2 = f * f, g2 = g * g;
    return f2 / (f2 + g2);
}

inline bool scene_intersect(const orc_scene& s, const Ray& r, Surf& it, RayLog* log) {
    if (log) log->put(r, 0);
    Hit h;
    if (!closest_hit(s, r, h)) return false;
    complete(s, (uint32_t)h.prim, h.bary, it);
    return true;
}

V3 sample_ld(const orc_scene& s, Sampler& smp, const Surf& it, V3 wo, RayLog* log) { /* integrator.cpp:109-141 */
    size_t nl = s.lights.size();
    if (nl == 0) return mk(0.0f);
    float u1 = smp.s1();
    size_t li = (size_t)(int)(u1 * nl); /* light.h:58-60; clamped instead of throwing when u*n rounds up to n */
    if (li >= nl) li = nl - 1;
    const Light& light = s.lights[li];
    float selectPdf = 1.0f / nl;
    LiSample ls = light_sample(s, light, it, smp.s2());
    if (ls.pdfDir == 0.0f) return mk(0.0f);
    Ray sr = shadow_ray(it.p, ls.p, it.n, ls.n);
    if (log) log->put(sr, 1);
    if (any_hit(s, sr, nullptr)) return mk(0.0f);
    V3 f = brdf_eval(s, it, wo, ls.wi);
    float cosTheta = dot(it.n, ls.wi);
    float bp = brdf_pdf(s, it, wo, ls.wi);
    float lp = ls.pdfDir * selectPdf;
    float w = power_heuristic(lp, bp);
    return w * cmul(f, ls.L) * cosTheta / lp;
}

V3 li_path(const orc_scene& s, Sampler& smp, V2 pixelSample, RayLog* log) { /* integrator.cpp:45-107 */
    Ray ray = camera_ray(s.cam, pixelSample);
    V3 L = mk(0.0f), T = mk(1.0f);
    float brdfPdf = 0.0f;
    for (int bounce = 0; bounce < 32; bounce++) {
        Surf it;
        if (!scene_intersect(s, ray, it, log)) break;
        V3 wo = -ray.d;
        int32_t lid = s.light_of_tri[it.prim];
        if (lid >= 0) {
            const Light& light = s.lights[lid];
            V3 Le = dot(it.n, wo) > 0.0 ? light.lemit : mk(0.0f); /* light.cpp:14-16 */
            if (bounce == 0)
                L = L + cmul(T, Le);
            else {
                float lpdf = light_pdf(s, light, it, ray);
                lpdf *= 1.0f / s.lights.size();
                float w = power_heuristic(brdfPdf, lpdf);
                L = L + w * cmul(T, Le);
            }
        }
        V3 Ld = sample_ld(s, smp, it, wo, log);
        L = L + cmul(T, Ld);
        V2 u = smp.s2(); /* drawn before the 1-D sample: see file header */
        float uc = smp.s1();
        BSample bs = brdf_sample(s, it, wo, uc, u);
        if (bs.specular) {
            bounce--;
            brdfPdf = 1.0f;
        } else {
            if (sqnorm(bs.f) == 0.0f || bs.pdf == 0.0f) break;
            V3 thr = bs.f * dot(it.n, bs.wi) / bs.pdf;
            T = cmul(T, thr);
            brdfPdf = bs.pdf;
        }
        ray = Ray{it.p + it.ng * kEps, bs.wi, 0.0f, std::numeric_limits<float>::infinity()}; /* shape.cpp:186-189 */
        if (maxc(T) < 1.0f && bounce > 1) {
            float q = std::max(0.0f, 1.0f - maxc(T));
            if (smp.s1() < q) break;
            T = T / (1.0f - q);
        }
    }
    return L;
}

V3 li_normal(const orc_scene& s, V2 ps, RayLog* log) { /* integrator.cpp:20-31 */
    Surf it;
    if (scene_intersect(s, camera_ray(s.cam, ps), it, log)) return it.n;
    return mk(0.0f);
}
V3 li_albedo(const orc_scene& s, V2 ps, RayLog* log) { /* integrator.cpp:33-43 */
    Surf it;
    if (scene_intersect(s, camera_ray(s.cam, ps), it, log)) return base_color(s, s.mats[s.mat_id[it.prim]], it.uv);
    return mk(0.0f);
}

V3 li_any(const orc_scene& s, int integrator, Sampler& smp, V2 ps, RayLog* log) {
    switch (integrator) {
    case ORC_INTEGRATOR_NORMAL: return li_normal(s, ps, log);
    case ORC_INTEGRATOR_ALBEDO: return li_albedo(s, ps, log);
    default: return li_path(s, smp, ps, log);
    }
}

/* ---------------------------------------------------------------- film (block.cpp:14-110, filter.h:26-50) */

void make_filter(float* f) {
    const float radius = 2.0f, stddev = 0.5f;
    for (int i = 0; i < 32; ++i) {
        float pos = (radius * i) / 32;
        float alpha = -1.0f / (2.0f * stddev * stddev);
        f[i] = std::max(0.0f, std::exp(alpha * pos * pos) - std::exp(alpha * radius * radius));
    }
    f[32] = 0.0f;
}

/* ImageBlock::put for a block whose interior is size_x x size_y pixels at offset (ox, oy); buffer is (size+4)^2 RGBA */
inline void block_put(float* buf, int cols, int rows, int ox, int oy, int sx, int sy, const float* filter, V2 pos, V3 value) {
    const float radius = 2.0f, lookup = 32 / 2.0f;
    const int border = 2;
    float lx = pos.x - (float)ox, ly = pos.y - (float)oy;
    if (lx < 0.0f || lx >= sx || ly < 0.0f || ly >= sy) return;
    lx += border;
    ly += border;
    int bx0 = std::max(int(std::ceil(lx - radius)), 0);
    int by0 = std::max(int(std::ceil(ly - radius)), 0);
    int bx1 = std::min(int(std::floor(lx + radius)), cols - 1);
    int by1 = std::min(int(std::floor(ly + radius)), rows - 1);
    float wx[8], wy[8];
    for (int x = bx0, i = 0; x <= bx1; ++x, ++i) wx[i] = filter[(int)(std::abs(x - lx) * lookup)];
    for (int y = by0, i = 0; y <= by1; ++y, ++i) wy[i] = filter[(int)(std::abs(y - ly) * lookup)];
    for (int y = by0, yr = 0; y <= by1; ++y, ++yr)
        for (int x = bx0, xr = 0; x <= bx1; ++x, ++xr) {
            float* p = buf + (size_t(y) * cols + x) * 4;
            p[0] += value.x * wx[xr] * wy[yr];
            p[1] += value.y * wx[xr] * wy[yr];
            p[2] += value.z * wx[xr] * wy[yr];
            p[3] += 1.0f * wx[xr] * wy[yr];
        }
}

/* block.cpp:116-158: spiral block order */
std::vector<std::pair<int, int>> spiral_blocks(int W, int H, int bs) {
    int nbx = (int)std::ceil(W / (float)bs), nby = (int)std::ceil(H / (float)bs);
    int left = nbx * nby;
    int dir = 0, bx = (nbx - 1) / 2, by = (nby - 1) / 2, steps_left = 1, num_steps = 1;
    std::vector<std::pair<int, int>> out;
    while (left > 0) {
        out.emplace_back(bx, by);
        if (--left == 0) break;
        do {
            switch (dir) {
            case 0: ++bx; break;
            case 1: ++by; break;
            case 2: --bx; break;
            case 3: --by; break;
            }
            if (--steps_left == 0) {
                dir = (dir + 1) % 4;
                if (dir == 2 || dir == 0) ++num_steps;
                steps_left = num_steps;
            }
        } while (bx < 0 || by < 0 || bx >= nbx || by >= nby);
    }
    return out;
}

V3 to_v3(const float* p) { return mk(p[0], p[1], p[2]); }
Ray to_ray(const float* r) { return Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), r[6], r[7]}; }

} // namespace

/* ================================================================ C interface */

extern "C" {

orc_scene* orc_scene_create(const orc_scene_desc* d) {
    orc_scene* s = new orc_scene();
    s->verts.resize(d->n_verts);
    for (uint32_t i = 0; i < d->n_verts; ++i) s->verts[i] = to_v3(d->verts + 3 * i);
    s->normals.resize(d->n_normals);
    for (uint32_t i = 0; i < d->n_normals; ++i) s->normals[i] = to_v3(d->normals + 3 * i);
    s->uvs.resize(d->n_uvs);
    for (uint32_t i = 0; i < d->n_uvs; ++i) s->uvs[i] = V2{d->uvs[2 * i], d->uvs[2 * i + 1]};
    s->vi.assign(d->vi, d->vi + 3 * size_t(d->n_tris));
    s->ni.assign(d->ni, d->ni + 3 * size_t(d->n_tris));
    s->ti.assign(d->ti, d->ti + 3 * size_t(d->n_tris));
    s->mat_id.assign(d->mat_id, d->mat_id + d->n_tris);
    s->mats.resize(d->n_mats);
    for (uint32_t i = 0; i < d->n_mats; ++i)
        s->mats[i] = Material{to_v3(d->mat_kd + 3 * i), to_v3(d->mat_ks + 3 * i), d->mat_ns[i], d->mat_tex ? d->mat_tex[i] : -1};
    s->texs.resize(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        s->texs[i].w = d->tex_w[i];
        s->texs[i].h = d->tex_h[i];
        s->texs[i].rgb.assign(d->tex_rgb[i], d->tex_rgb[i] + size_t(3) * d->tex_w[i] * d->tex_h[i]);
    }
    s->light_of_tri.assign(d->n_tris, -1);
    s->lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) { /* light.cpp:10-12, scene.cpp:251-260 */
        s->lights[i] = Light{d->light_tri[i], to_v3(d->light_lemit + 3 * i), tri_area(*s, d->light_tri[i])};
        s->light_of_tri[d->light_tri[i]] = (int32_t)i;
    }
    camera_setup(s->cam, d->width, d->height, d->fovy, to_v3(d->eye), to_v3(d->lookat), to_v3(d->up));
    s->sobol_scale = round_up_pow2((int32_t)std::max(d->width, d->height)); /* sampler.cpp:89-94 */
    s->sobol_m = log2int((float)s->sobol_scale);
    make_filter(s->filter);
    if (d->n_tris > 0) Builder(*s).run();
    return s;
}

/* Test hook, no counterpart in the reference: the same scene, but traversing a tree that was built elsewhere (nodes in the
 * layout of bvh.h:13-45) instead of running the restated builder.  For scenes of tens of millions of triangles, where the
 * single-threaded builder above takes minutes; that the two builders produce the same tree is asserted on smaller scenes. */
orc_scene* orc_scene_create_with_tree(const orc_scene_desc* d, size_t n_nodes, const float* boxes, const uint32_t* first,
                                      const uint32_t* count, const uint32_t* prim_ids) {
    orc_scene_desc no_tris = *d;
    no_tris.n_tris = 0; /* everything but the builder */
    no_tris.n_lights = 0;
    orc_scene* s = orc_scene_create(&no_tris);
    s->vi.assign(d->vi, d->vi + 3 * size_t(d->n_tris));
    s->ni.assign(d->ni, d->ni + 3 * size_t(d->n_tris));
    s->ti.assign(d->ti, d->ti + 3 * size_t(d->n_tris));
    s->mat_id.assign(d->mat_id, d->mat_id + d->n_tris);
    s->light_of_tri.assign(d->n_tris, -1);
    s->lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        s->lights[i] = Light{d->light_tri[i], to_v3(d->light_lemit + 3 * i), tri_area(*s, d->light_tri[i])};
        s->light_of_tri[d->light_tri[i]] = (int32_t)i;
    }
    s->nodes.resize(n_nodes);
    for (size_t i = 0; i < n_nodes; ++i) {
        s->nodes[i].box.lo = to_v3(boxes + 6 * i);
        s->nodes[i].box.hi = to_v3(boxes + 6 * i + 3);
        s->nodes[i].first = first[i];
        s->nodes[i].count = count[i];
    }
    s->prim_ids.assign(prim_ids, prim_ids + d->n_tris);
    return s;
}

void orc_scene_destroy(orc_scene* s) { delete s; }

size_t orc_bvh_num_nodes(const orc_scene* s) { return s->nodes.size(); }

void orc_bvh_get(const orc_scene* s, float* boxes, uint64_t* first, uint64_t* count, uint64_t* prim_ids) {
    for (size_t i = 0; i < s->nodes.size(); ++i) {
        const Node& n = s->nodes[i];
        float* b = boxes + 6 * i;
        b[0] = n.box.lo.x; b[1] = n.box.lo.y; b[2] = n.box.lo.z; b[3] = n.box.hi.x; b[4] = n.box.hi.y; b[5] = n.box.hi.z;
        first[i] = n.first;
        count[i] = n.count;
    }
    for (size_t i = 0; i < s->prim_ids.size(); ++i) prim_ids[i] = s->prim_ids[i];
}

int orc_tri_intersect(const orc_scene* s, uint32_t prim, const float* ray8, float* bary3, float* t) {
    V3 b = mk(0.0f);
    float tt = 0.0f;
    bool h = tri_hit(*s, prim, to_ray(ray8), b, tt);
    bary3[0] = b.x; bary3[1] = b.y; bary3[2] = b.z;
    *t = tt;
    return h ? 1 : 0;
}

int orc_aabb_intersect(const float* box6, const float* ray8) {
    Box b;
    b.lo = to_v3(box6);
    b.hi = to_v3(box6 + 3);
    return box_hit(b, to_ray(ray8)) ? 1 : 0;
}

void orc_intersect(const orc_scene* s, const float* rays8, size_t n, orc_hit* out, orc_surface* surf) {
    for (size_t i = 0; i < n; ++i) {
        Hit h;
        bool ok = closest_hit(*s, to_ray(rays8 + 8 * i), h);
        out[i].prim = ok ? h.prim : -1;
        out[i].t = ok ? h.t : 0.0f;
        out[i].bary[0] = ok ? h.bary.x : 0.0f;
        out[i].bary[1] = ok ? h.bary.y : 0.0f;
        out[i].bary[2] = ok ? h.bary.z : 0.0f;
        if (surf) {
            std::memset(&surf[i], 0, sizeof(orc_surface));
            if (ok) {
                Surf it;
                complete(*s, (uint32_t)h.prim, h.bary, it);
                orc_surface& o = surf[i];
                o.p[0] = it.p.x; o.p[1] = it.p.y; o.p[2] = it.p.z;
                o.n[0] = it.n.x; o.n[1] = it.n.y; o.n[2] = it.n.z;
                o.ng[0] = it.ng.x; o.ng[1] = it.ng.y; o.ng[2] = it.ng.z;
                o.uv[0] = it.uv.x; o.uv[1] = it.uv.y;
            }
        }
    }
}

void orc_occluded(const orc_scene* s, const float* rays8, size_t n, uint8_t* out, int32_t* first_prim) {
    for (size_t i = 0; i < n; ++i) {
        int32_t fp = -1;
        out[i] = any_hit(*s, to_ray(rays8 + 8 * i), &fp) ? 1 : 0;
        if (first_prim) first_prim[i] = fp;
    }
}

void orc_traversal_counters(const orc_scene*, uint64_t* nodes, uint64_t* tris) {
    *nodes = tl_cnt.nodes;
    *tris = tl_cnt.tris;
}

uint64_t orc_sobol_index(uint32_t m, uint64_t frame, int32_t px, int32_t py) { return sobol_interval_to_index(m, frame, px, py); }
float orc_sobol_sample(int64_t index, int dim) { return sobol_sample(index, dim); }

void orc_pixel_sample(const orc_scene* s, int32_t px, int32_t py, uint32_t sample, float* out) {
    Sampler smp;
    smp.scale = s->sobol_scale;
    smp.start(px, py, (int)sample);
    V2 j = smp.pixel2();
    out[0] = (float)px + j.x;
    out[1] = (float)py + j.y;
}

void orc_camera_ray(const orc_scene* s, float sx, float sy, float* r) {
    Ray ray = camera_ray(s->cam, V2{sx, sy});
    r[0] = ray.o.x; r[1] = ray.o.y; r[2] = ray.o.z; r[3] = ray.d.x; r[4] = ray.d.y; r[5] = ray.d.z; r[6] = ray.tmin; r[7] = ray.tmax;
}

void orc_camera_get(const orc_scene* s, float* s2c, float* c2w, float* proj_nume) {
    std::memcpy(s2c, s->cam.s2c, 64);
    std::memcpy(c2w, s->cam.c2w, 64);
    *proj_nume = s->cam.proj_nume;
}

void orc_brdf_eval(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, const float* wi3, float* f3) {
    Surf it;
    complete(*s, prim, to_v3(bary3), it);
    V3 f = brdf_eval(*s, it, to_v3(wo3), to_v3(wi3));
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
}

float orc_brdf_pdf(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, const float* wi3) {
    Surf it;
    complete(*s, prim, to_v3(bary3), it);
    return brdf_pdf(*s, it, to_v3(wo3), to_v3(wi3));
}

int orc_brdf_sample(const orc_scene* s, uint32_t prim, const float* bary3, const float* wo3, float uc, const float* u2, float* wi3,
                    float* pdf, float* f3) {
    Surf it;
    complete(*s, prim, to_v3(bary3), it);
    BSample b = brdf_sample(*s, it, to_v3(wo3), uc, V2{u2[0], u2[1]});
    wi3[0] = b.wi.x; wi3[1] = b.wi.y; wi3[2] = b.wi.z;
    *pdf = b.pdf;
    f3[0] = b.f.x; f3[1] = b.f.y; f3[2] = b.f.z;
    return b.specular ? 1 : 0;
}

void orc_light_sample(const orc_scene* s, uint32_t light, uint32_t surf_prim, const float* surf_bary3, const float* u2, float* o) {
    Surf it;
    complete(*s, surf_prim, to_v3(surf_bary3), it);
    LiSample l = light_sample(*s, s->lights[light], it, V2{u2[0], u2[1]});
    o[0] = l.L.x; o[1] = l.L.y; o[2] = l.L.z; o[3] = l.wi.x; o[4] = l.wi.y; o[5] = l.wi.z; o[6] = l.p.x; o[7] = l.p.y; o[8] = l.p.z;
    o[9] = l.n.x; o[10] = l.n.y; o[11] = l.n.z; o[12] = l.pdfArea; o[13] = l.pdfDir;
}

float orc_light_pdf(const orc_scene* s, uint32_t light, uint32_t hit_prim, const float* hit_bary3, const float* ray8) {
    Surf it;
    complete(*s, hit_prim, to_v3(hit_bary3), it);
    return light_pdf(*s, s->lights[light], it, to_ray(ray8));
}

void orc_li(const orc_scene* s, int integrator, int32_t px, int32_t py, uint32_t sample, float* rgb3, float* pixel_sample2, float* ray_log,
            size_t max_log, size_t* n_log, int32_t* dims_used) {
    Sampler smp;
    smp.scale = s->sobol_scale;
    smp.start(px, py, (int)sample);
    V2 j = smp.pixel2();
    V2 ps{(float)px + j.x, (float)py + j.y};
    RayLog log{ray_log, max_log, 0};
    V3 L = li_any(*s, integrator, smp, ps, &log);
    rgb3[0] = L.x; rgb3[1] = L.y; rgb3[2] = L.z;
    if (pixel_sample2) { pixel_sample2[0] = ps.x; pixel_sample2[1] = ps.y; }
    if (n_log) *n_log = log.n;
    if (dims_used) *dims_used = smp.dim;
}

void orc_filter_table(float* out33) { make_filter(out33); }

void orc_film_put(float* film, uint32_t W, uint32_t H, const float* pos2, const float* rgb3) {
    float f[33];
    make_filter(f);
    block_put(film, (int)W + 4, (int)H + 4, 0, 0, (int)W, (int)H, f, V2{pos2[0], pos2[1]}, to_v3(rgb3));
}

void orc_render(const orc_scene* s, int integrator, uint32_t spp_begin, uint32_t spp_end, int nthreads, uint32_t max_blocks, float* film,
                orc_stats* stats) {
    const int W = (int)s->cam.W, H = (int)s->cam.H, BS = 32, B = BS + 4;
    const int cols = W + 4;
    std::memset(film, 0, sizeof(float) * 4 * size_t(W + 4) * size_t(H + 4));
    std::vector<std::pair<int, int>> order = spiral_blocks(W, H, BS);
    if (max_blocks > 0 && max_blocks < order.size()) order.resize(max_blocks);
    std::vector<std::vector<float>> results(order.size());
    std::atomic<size_t> next{0};
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    std::vector<Counters> cnt(nthreads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        tl_cnt = Counters();
        Sampler smp;
        smp.scale = s->sobol_scale;
        for (;;) {
            size_t bi = next.fetch_add(1);
            if (bi >= order.size()) break;
            int ox = order[bi].first * BS, oy = order[bi].second * BS;
            int sx = std::min(BS, W - ox), sy = std::min(BS, H - oy);
            std::vector<float>& buf = results[bi];
            buf.assign(size_t(B) * B * 4, 0.0f);
            for (int y = 0; y < sy; ++y)          /* main.cpp:54-74 */
                for (int x = 0; x < sx; ++x)
                    for (uint32_t sidx = spp_begin; sidx < spp_end; ++sidx) {
                        int px = x + ox, py = y + oy;
                        smp.start(px, py, (int)sidx);
                        V2 j = smp.pixel2();
                        V2 ps{(float)px + j.x, (float)py + j.y};
                        V3 L = li_any(*s, integrator, smp, ps, nullptr);
                        block_put(buf.data(), B, B, ox, oy, sx, sy, s->filter, ps, L);
                    }
        }
        cnt[tid] = tl_cnt;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    /* block.cpp:102-110, applied in spiral order */
    for (size_t bi = 0; bi < order.size(); ++bi) {
        int ox = order[bi].first * BS, oy = order[bi].second * BS;
        int sx = std::min(BS, W - ox), sy = std::min(BS, H - oy);
        const std::vector<float>& buf = results[bi];
        for (int y = 0; y < sy + 4; ++y)
            for (int x = 0; x < sx + 4; ++x)
                for (int c = 0; c < 4; ++c) film[(size_t(oy + y) * cols + (ox + x)) * 4 + c] += buf[(size_t(y) * B + x) * 4 + c];
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const Counters& c : cnt) {
            stats->closest_rays += c.closest;
            stats->any_rays += c.any;
            stats->nodes_visited += c.nodes;
            stats->tris_tested += c.tris;
        }
        uint64_t px = 0;
        for (auto& o : order) px += uint64_t(std::min(BS, W - o.first * BS)) * std::min(BS, H - o.second * BS);
        stats->paths = px * (spp_end - spp_begin);
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
}

} // extern "C"
