// Host side of the scene boundary: the Cornell box factory and the flattening of
// a crt_scene_desc into the device-resident DevScene.  Compiled for the host
// only, IEEE binary32, no contraction (see Makefile: -ffp-contract=off).
#include "crt_scene.hpp"

#include <math.h>
#include <cmath>
#include <stdio.h>
#include <string.h>

#include "../include/craytracer/craytracer.hpp"

namespace crt {

namespace {

struct F3 {
    float x, y, z;
};
inline F3 ld(const float* p) { return {p[0], p[1], p[2]}; }
inline void st(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 add(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float comp(F3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline float lensq(F3 a) {
    float r = 0.0f;
    r += a.x * a.x;
    r += a.y * a.y;
    r += a.z * a.z;
    return r;
}
// mstd length()/normalize() (vec.hpp:398-422): 1/(1/sqrt), multiply by 1/len
inline F3 normalize(F3 a) {
    float len = 1.0f / (1.0f / sqrtf(lensq(a)));
    if (len == 0.0f) return a;
    float inv = 1.0f / len;
    return {a.x * inv, a.y * inv, a.z * inv};
}

// The vertex-only part of the bilinear quad test (Quad.hpp:202-250, AreaLight.hpp:382-432).
struct QuadConst {
    F3 a, b, nor;
    uint32_t id, idu, idv;
    float ka[2], kb[2], kg[2], k2, nor_id;
    uint32_t linear;
};
QuadConst quad_constants(F3 p0, F3 p1, F3 p2, F3 p3) {
    static const uint32_t lut[4] = {1, 2, 0, 1};
    QuadConst q;
    q.a = sub(p1, p0);
    q.b = sub(p3, p0);
    F3 c = sub(p2, p0);
    q.nor = cross(q.a, q.b);
    float mx = fabsf(q.nor.x), my = fabsf(q.nor.y), mz = fabsf(q.nor.z);
    q.id = (mx > my && mx > mz) ? 0u : (my > mz) ? 1u : 2u;
    q.idu = lut[q.id];
    q.idv = lut[q.id + 1];
    float kcx = comp(c, q.idu), kcy = comp(c, q.idv);
    q.ka[0] = comp(q.a, q.idu);
    q.ka[1] = comp(q.a, q.idv);
    q.kb[0] = comp(q.b, q.idu);
    q.kb[1] = comp(q.b, q.idv);
    q.kg[0] = (kcx - q.kb[0]) - q.ka[0];
    q.kg[1] = (kcy - q.kb[1]) - q.ka[1];
    float kcbx = kcx - q.kb[0], kcby = kcy - q.kb[1];
    q.k2 = kcbx * q.ka[1] - kcby * q.ka[0];
    q.nor_id = comp(q.nor, q.id);
    q.linear = fabsf(q.k2 - 0.0f) < 1e-4f ? 1u : 0u;
    return q;
}

void fail(char* err, size_t cap, const char* msg) {
    if (err && cap) snprintf(err, cap, "%s", msg);
}

} // namespace

static void fill_uv(DevGeomUV& d, const QuadConst& q) {
    d.idu = q.idu;
    d.idv = q.idv;
    d.linear = q.linear;
    d.nor_id = q.nor_id;
    memcpy(d.ka, q.ka, 8);
    memcpy(d.kb, q.kb, 8);
    memcpy(d.kg, q.kg, 8);
    d.k2 = q.k2;
    d.pad = 0.0f;
}

int flatten_scene(const crt_scene_desc* in, DevScene* out, char* err, size_t errcap) {
    if (!in || !out) {
        fail(err, errcap, "flatten_scene: null argument");
        return CRT_ERR_INVALID;
    }
    if (in->n_geometry > CRT_MAX_GEOMETRY || (in->n_geometry && !in->geometry)) {
        fail(err, errcap, "scene: n_geometry exceeds CRT_MAX_GEOMETRY or geometry is NULL");
        return CRT_ERR_INVALID;
    }
    if (in->n_lights > CRT_MAX_LIGHTS || (in->n_lights && !in->lights)) {
        fail(err, errcap, "scene: n_lights exceeds CRT_MAX_LIGHTS or lights is NULL");
        return CRT_ERR_INVALID;
    }
    memset(out, 0, sizeof *out);

    // Camera (Camera.hpp:86-141)
    const crt_camera& c = in->camera;
    if (c.type != CRT_CAM_ORTHOGRAPHIC && c.type != CRT_CAM_PERSPECTIVE) {
        fail(err, errcap, "scene: unknown camera type");
        return CRT_ERR_INVALID;
    }
    F3 front = ld(c.front);
    F3 right = normalize(cross(front, F3{0.0f, 1.0f, 0.0f}));
    F3 up = normalize(cross(right, front));
    st(out->cam.pos, ld(c.pos));
    st(out->cam.front, front);
    st(out->cam.right, right);
    st(out->cam.up, up);
    out->cam.one_over_tan = 1.0f / tanf(c.fov * 0.5f);
    out->cam.ortho_scale = c.orthoScale;
    out->cam.type = c.type;

    const float LUT_SCALE = (64.0f - 1.0f) / 64.0f, LUT_BIAS = 0.5f / 64.0f; // AreaLight.hpp:47-49
    out->n_geom = in->n_geometry;
    for (uint32_t i = 0; i < in->n_geometry; ++i) {
        const crt_geometry& g = in->geometry[i];
        DevGeomHot& h = out->hot[i];
        DevGeomShade& d = out->shade[i];
        if (g.kind != CRT_GEOM_SPHERE && g.kind != CRT_GEOM_QUAD) {
            fail(err, errcap, "scene: unknown geometry kind");
            return CRT_ERR_INVALID;
        }
        if (g.mat.type > CRT_MAT_REFRACTIVE) {
            fail(err, errcap, "scene: unknown material type");
            return CRT_ERR_INVALID;
        }
        h.kind = g.kind;
        h.mat_type = g.mat.type;
        memcpy(d.ambient, g.mat.ambient, 16);
        memcpy(d.diffuse, g.mat.diffuse, 16);
        memcpy(d.specular, g.mat.specular, 16);
        d.refract_index = g.mat.refractIndex;
        d.shininess = g.mat.shininess;
        // roughness -> LTC texel column (AreaLight.hpp:608,619,625; Texture.hpp:126-152,179-182)
        float roughness = 1.0f - 0.25f * powf(g.mat.shininess, 0.2f);
        float u = roughness * LUT_SCALE + LUT_BIAS;
        if (u < 0.0f || u > 1.0f) u = fminf(fmaxf(u, 0.0f), 1.0f);
        float fx = u * 63.0f;
        uint32_t col = (fx >= 0.0f && fx < 4294967296.0f) ? (uint32_t)fx : 0xFFFFFFFFu;
        d.ltc_col = col < 64u ? col : 0xFFFFFFFFu; // out of range -> getPixelColor returns zeros
        if (g.kind == CRT_GEOM_SPHERE) {
            memcpy(h.o, g.center, 12);
            h.n[0] = g.radius * g.radius;
        } else {
            F3 p0 = ld(g.v[0].pos);
            QuadConst q = quad_constants(p0, ld(g.v[1].pos), ld(g.v[2].pos), ld(g.v[3].pos));
            st(h.o, p0);
            st(h.n, q.nor);
            if (q.nor.y == 0.0f && q.nor.z == 0.0f && q.nor.x != 0.0f) h.kind = CRT_HOT_QUAD_X;
            else if (q.nor.x == 0.0f && q.nor.z == 0.0f && q.nor.y != 0.0f) h.kind = CRT_HOT_QUAD_Y;
            else if (q.nor.x == 0.0f && q.nor.y == 0.0f && q.nor.z != 0.0f) h.kind = CRT_HOT_QUAD_Z;
            fill_uv(out->uv[i], q);
            for (int k = 0; k < 4; ++k) memcpy(d.n[k], g.v[k].normal, 12);
        }
    }

    out->n_lights = in->n_lights;
    for (uint32_t i = 0; i < in->n_lights; ++i) {
        const crt_light& l = in->lights[i];
        DevLight& d = out->lights[i];
        if (l.kind != CRT_LIGHT_AREA && l.kind != CRT_LIGHT_POINT) {
            fail(err, errcap, "scene: unknown light kind");
            return CRT_ERR_INVALID;
        }
        d.kind = l.kind;
        memcpy(d.pos, l.pos, 12);
        memcpy(d.color, l.color, 12);
        d.intensity = l.intensity;
        d.two_sided = l.twoSided;
        d.shadow_samples = l.shadowSamples;
        d.initialized = l.initialized;
        d.horizon_clip = l.horizonClip != 0u ? 1u : 0u;
        if (l.kind == CRT_LIGHT_AREA) {
            F3 pos = ld(l.pos);
            F3 p[4], w[4];
            for (int k = 0; k < 4; ++k) {
                p[k] = ld(l.points[k]);
                w[k] = add(pos, p[k]);
                st(d.pts[k], w[k]);
            }
            st(d.p0, p[0]);
            st(d.p2, p[2]);
            st(d.e01, sub(p[1], p[0]));
            st(d.e03, sub(p[3], p[0]));
            st(d.e23, sub(p[3], p[2]));
            st(d.e21, sub(p[1], p[2]));
            st(d.light_normal, cross(sub(w[1], w[0]), sub(w[3], w[0])));
            {
                double scale = 1e-30, dev = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double p0 = l.points[0][k], p1 = l.points[1][k], p2 = l.points[2][k], p3 = l.points[3][k];
                    d.fB0[k] = (float)(p0 + (double)l.pos[k]);
                    d.fB1[k] = (float)(p2 + (p3 - p2) + (p1 - p2) + (double)l.pos[k]);
                    scale = fmax(scale, fmax(fmax(fabs(p0), fabs(p1)), fmax(fabs(p2), fabs(p3))));
                    dev = fmax(dev, fabs((p1 - p0) + (p3 - p0) - (p2 - p0))); // p2 = p0 + e01 + e03 for a parallelogram
                }
                d.is_par = dev <= 1e-6 * scale ? 1u : 0u;
                double g11 = 0, g12 = 0, g22 = 0;
                for (int k = 0; k < 3; ++k) {
                    g11 += (double)d.e01[k] * d.e01[k];
                    g12 += (double)d.e01[k] * d.e03[k];
                    g22 += (double)d.e03[k] * d.e03[k];
                }
                d.fg[0] = (float)g11;
                d.fg[1] = (float)(2.0 * g12);
                d.fg[2] = (float)g22;
            }
            QuadConst q = quad_constants(p[0], p[1], p[2], p[3]);
            st(d.nor, q.nor);
            fill_uv(d.uv, q);
        }
    }

    // Where a light can only ever add the ambient term (DevLight::dark_lo/hi).
    {
        auto finite3 = [](const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); };
        bool possible = true;
        for (uint32_t l = 0; l < in->n_lights; ++l) {
            DevLight& d = out->lights[l];
            d.dark_lo = d.dark_hi = 0u;
            if (d.kind != CRT_LIGHT_AREA) {
                possible = false;
                continue;
            }
            if (!d.initialized) continue; // adds the ambient product wherever the point is (AreaLight.hpp:586)
            if (d.two_sided || d.horizon_clip || d.shadow_samples == 0u || !std::isfinite(d.intensity) || !finite3(d.color)) {
                possible = false;
                continue;
            }
            uint64_t mask = 0;
            for (uint32_t g = 0; g < in->n_geometry; ++g) {
                const DevGeomShade& m = out->shade[g];
                bool ok = finite3(m.diffuse) && finite3(m.specular);
                for (int k = 0; k < 3 && ok; ++k) {
                    const float amb = m.ambient[k] * d.color[k];
                    uint32_t bits;
                    memcpy(&bits, &amb, 4);
                    ok = bits <= 0x7f800000u; // +0 .. +inf: adding a zero of either sign changes no bit
                }
                if (ok) mask |= 1ull << g;
            }
            d.dark_lo = (uint32_t)mask;
            d.dark_hi = (uint32_t)(mask >> 32);
            if (mask == 0) possible = false;
        }
        out->dark_possible = possible ? 1u : 0u;
    }

    // Per (light, quad) culling constants.  Along a shadow ray from P (+0.01 along the ray) to a light
    // sample X (+0.01 beyond it) the signed plane distance s(.) = nor.(. - v0) is linear, so if P and all
    // of the light are on one side, the plane is crossed neither before the origin nor before the end:
    // the reference's t = -(p.nor / d.nor) is then negative or exceeds the ray's length (Quad.hpp:222-226).
    // Margins: `need` keeps the overshoot of 0.01 plus 1e-3 of the scene size between the light and the
    // plane, which is 500x the rounding of t; `thr` lets P sit on the plane itself (its own quad) as long
    // as the 0.01 step along the ray certainly lifts the origin off it (noise: rounding of p.nor).
    double rabs = 0.0;
    auto grow = [&](double v) { if (fabs(v) > rabs) rabs = fabs(v); };
    for (uint32_t i = 0; i < in->n_geometry; ++i) {
        const crt_geometry& g = in->geometry[i];
        if (g.kind == CRT_GEOM_SPHERE) {
            for (int k = 0; k < 3; ++k) grow(fabs((double)g.center[k]) + fabs((double)g.radius));
        } else {
            for (int v = 0; v < 4; ++v)
                for (int k = 0; k < 3; ++k) grow(g.v[v].pos[k]);
        }
    }
    for (uint32_t l = 0; l < in->n_lights; ++l)
        if (in->lights[l].kind == CRT_LIGHT_AREA)
            for (int v = 0; v < 4; ++v)
                for (int k = 0; k < 3; ++k) grow(out->lights[l].pts[v][k]);
    bool hull_ok[CRT_MAX_GEOMETRY] = {}; // quads whose accepted (u,v) region is known to lie within their vertices' hull
    for (uint32_t i = 0; i < CRT_MAX_GEOMETRY; ++i) {
        DevGeomBox& b = out->box[i];
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = -INFINITY;
            b.hi[k] = INFINITY;
        }
        b.n2 = 0.0f;
        b.sp_min2 = INFINITY;
        for (int k = 0; k < 3; ++k) {
            b.lo_in[k] = INFINITY;
            b.hi_in[k] = -INFINITY;
        }
        b.noise = INFINITY;
        b.pad = 0.0f;
        if (i < in->n_geometry && in->geometry[i].kind == CRT_GEOM_QUAD && rabs > 0.0 && rabs < 1e6) {
            const DevGeomHot& hq = out->hot[i];
            const double n2 = (double)hq.n[0] * hq.n[0] + (double)hq.n[1] * hq.n[1] + (double)hq.n[2] * hq.n[2];
            if (n2 > 1e-24 && n2 < 1e24) {
                b.n2 = (float)n2;
                b.sp_min2 = (float)(1e-8 * n2 * rabs * rabs);
                // |P|, |v0| <= sqrt(3) rabs and |D| <= 2 sqrt(3) rabs: the three products and two sums of a dot
                // product, the rounding of P (or of the ray origin P + 0.01 d) and of the subtraction stay below
                // 7e-7 |nor| rabs; two such computations are compared
                b.noise = (float)(3e-6 * sqrt(n2) * rabs);
            }
        }
        if (i >= in->n_geometry || in->geometry[i].kind != CRT_GEOM_QUAD || !out->uv[i].linear) continue;
        if (!(rabs > 0.0) || !(rabs < 1e6)) continue;
        {
            // The linear branch (Quad.hpp:255) is taken whenever |k2| < 1e-4, which also holds for trapezoids
            // and for any small quad; its u = cross(kp, ka) / k1 with k1 = cross(kp, kg) - nor_id then accepts a
            // region that is not the hull of the four vertices.  Only when kg (= v2 - v1 - v3 + v0 on the kept
            // axes) is negligible against nor_id are (u, v) the affine coordinates of a parallelogram, and only
            // then do the vertices bound the region: the relative change of k1 over the region stays below 1e-4,
            // i.e. the boundary moves by less than a tenth of the margin.
            const DevGeomUV& uvq = out->uv[i];
            const double span = fabs((double)uvq.ka[0]) + fabs((double)uvq.ka[1]) + fabs((double)uvq.kb[0]) + fabs((double)uvq.kb[1]);
            const double skew = fabs((double)uvq.kg[0]) + fabs((double)uvq.kg[1]);
            if (!(skew * span <= 1e-5 * fabs((double)uvq.nor_id))) continue;
        }
        hull_ok[i] = true;
        const DevGeomHot& h = out->hot[i];
        const double nlen = sqrt((double)h.n[0] * h.n[0] + (double)h.n[1] * h.n[1] + (double)h.n[2] * h.n[2]);
        if (!(nlen > 1e-12) || !(nlen < 1e12)) continue;
        const double margin = 1e-3 * rabs;
        const uint32_t keep[2] = {out->uv[i].idu, out->uv[i].idv};
        for (int a = 0; a < 2; ++a) {
            const uint32_t k = keep[a];
            double lo = INFINITY, hi = -INFINITY;
            for (int v = 0; v < 4; ++v) {
                lo = fmin(lo, (double)in->geometry[i].v[v].pos[k]);
                hi = fmax(hi, (double)in->geometry[i].v[v].pos[k]);
            }
            b.lo[k] = (float)(lo - margin);
            b.hi[k] = (float)(hi + margin);
        }
        // axis-aligned rectangle: every vertex sits on a corner of the kept-axes box, all four corners used
        {
            const uint32_t ku = keep[0], kv = keep[1];
            double lo2[2] = {INFINITY, INFINITY}, hi2[2] = {-INFINITY, -INFINITY};
            for (int v = 0; v < 4; ++v) {
                lo2[0] = fmin(lo2[0], (double)in->geometry[i].v[v].pos[ku]);
                hi2[0] = fmax(hi2[0], (double)in->geometry[i].v[v].pos[ku]);
                lo2[1] = fmin(lo2[1], (double)in->geometry[i].v[v].pos[kv]);
                hi2[1] = fmax(hi2[1], (double)in->geometry[i].v[v].pos[kv]);
            }
            const double tol = 1e-6 * rabs;
            unsigned corners = 0;
            bool on_corner = true;
            for (int v = 0; v < 4; ++v) {
                const double pu = in->geometry[i].v[v].pos[ku], pv = in->geometry[i].v[v].pos[kv];
                const bool ulo = fabs(pu - lo2[0]) <= tol, uhi = fabs(pu - hi2[0]) <= tol;
                const bool vlo = fabs(pv - lo2[1]) <= tol, vhi = fabs(pv - hi2[1]) <= tol;
                if (!(ulo || uhi) || !(vlo || vhi)) on_corner = false;
                corners |= 1u << ((uhi ? 1 : 0) | (vhi ? 2 : 0));
            }
            const bool wide = hi2[0] - lo2[0] > 4.0 * margin && hi2[1] - lo2[1] > 4.0 * margin;
            if (on_corner && corners == 15u && wide) {
                for (int k = 0; k < 3; ++k) {
                    b.lo_in[k] = -INFINITY;
                    b.hi_in[k] = INFINITY;
                }
                b.lo_in[ku] = (float)(lo2[0] + margin);
                b.hi_in[ku] = (float)(hi2[0] - margin);
                b.lo_in[kv] = (float)(lo2[1] + margin);
                b.hi_in[kv] = (float)(hi2[1] - margin);
            }
        }
    }
    for (uint32_t i = 0; i < CRT_MAX_GEOMETRY; ++i) {
        DevGeomBox& b = out->box[i];
        for (int k = 0; k < 3; ++k) {
            const bool bounded = b.lo[k] > -INFINITY && b.hi[k] < INFINITY;
            b.mid[k] = bounded ? 0.5f * (b.lo[k] + b.hi[k]) : 0.0f;
            b.half_out[k] = bounded ? 0.5f * (b.hi[k] - b.lo[k]) : INFINITY;
            const bool inner = b.lo_in[0] < INFINITY; // rectangle with an inner box
            b.half_in[k] = !inner ? -1.0f : (bounded ? 0.5f * (b.hi_in[k] - b.lo_in[k]) : INFINITY);
            // lo/hi are the extent widened by 1e-3 rabs, lo_in/hi_in the extent shrunk by it
            const double tight = 3e-5 * rabs, wide_margin = 1e-3 * rabs;
            b.half_out_t[k] = bounded ? (float)((double)b.half_out[k] - wide_margin + tight) : INFINITY;
            b.half_in_t[k] = !inner ? -1.0f : (bounded ? (float)((double)b.half_in[k] + wide_margin - tight) : INFINITY);
        }
        b.pad2 = 0.0f;
        DevGeomWalk& wk = out->walk[i];
        wk.kind_mat = out->hot[i].kind | (out->hot[i].mat_type << 8);
        for (int k = 0; k < 3; ++k) {
            wk.o[k] = out->hot[i].o[k];
            wk.n[k] = out->hot[i].n[k];
            wk.mid[k] = b.mid[k];
            wk.half_out_t[k] = b.half_out_t[k];
            wk.half_in_t[k] = b.half_in_t[k];
        }
    }
    for (uint32_t g = 0; g < CRT_MAX_GEOMETRY; ++g) {
        out->interior_lo[g] = out->interior_hi[g] = 0xffffffffu;
        if (g >= in->n_geometry || in->geometry[g].kind != CRT_GEOM_SPHERE || !(rabs > 0.0) || !(rabs < 1e6)) continue;
        const crt_geometry& sg = in->geometry[g];
        const double reach = fabs((double)sg.radius) + 1e-3 * rabs + 0.02;
        uint64_t mask = 0;
        for (uint32_t j = 0; j < in->n_geometry; ++j) {
            const crt_geometry& o = in->geometry[j];
            double d2 = 0.0; // squared distance from the sphere's centre to primitive j (to its bounding box for a quad)
            if (o.kind == CRT_GEOM_SPHERE) {
                double c2 = 0.0;
                for (int k = 0; k < 3; ++k) c2 += ((double)o.center[k] - sg.center[k]) * ((double)o.center[k] - sg.center[k]);
                const double gap = sqrt(c2) - fabs((double)o.radius);
                d2 = gap > 0.0 ? gap * gap : 0.0;
            } else if (hull_ok[j]) {
                for (int k = 0; k < 3; ++k) {
                    double lo = INFINITY, hi = -INFINITY;
                    for (int v = 0; v < 4; ++v) {
                        lo = fmin(lo, (double)o.v[v].pos[k]);
                        hi = fmax(hi, (double)o.v[v].pos[k]);
                    }
                    const double c = sg.center[k];
                    const double gap = c < lo ? lo - c : (c > hi ? c - hi : 0.0);
                    d2 += gap * gap;
                }
            } // any other quad stays a candidate: the reference's (u,v) solve can accept points outside the
              // hull of its vertices (linear branch with skew, quadratic branch with a tiny k2)
            if (j == g || !(d2 > reach * reach)) mask |= 1ull << j;
        }
        out->interior_lo[g] = (uint32_t)mask;
        out->interior_hi[g] = (uint32_t)(mask >> 32);
    }
    const double dmax = 3.5 * rabs; // > 2*sqrt(3)*rabs: no two scene points are further apart
    for (uint32_t l = 0; l < CRT_MAX_LIGHTS; ++l)
        for (uint32_t i = 0; i < CRT_MAX_GEOMETRY; ++i) {
            out->cull_side[l][i] = 0.0f;
            out->cull_thr[l][i] = INFINITY;
        }
    for (uint32_t l = 0; l < in->n_lights; ++l) {
        if (in->lights[l].kind != CRT_LIGHT_AREA) continue;
        for (uint32_t i = 0; i < in->n_geometry; ++i) {
            const DevGeomHot& h = out->hot[i];
            if (h.kind == CRT_GEOM_SPHERE) continue;
            {
                const DevLight& dl = out->lights[l];
                double d1 = 0, d2 = 0, nn = 0, e1 = 0, e3 = 0;
                for (int k = 0; k < 3; ++k) {
                    d1 += (double)h.n[k] * dl.e01[k];
                    d2 += (double)h.n[k] * dl.e03[k];
                    nn += (double)h.n[k] * h.n[k];
                    e1 += (double)dl.e01[k] * dl.e01[k];
                    e3 += (double)dl.e03[k] * dl.e03[k];
                }
                const bool par = dl.is_par && fabs(d1) <= 1e-6 * sqrt(nn * e1) && fabs(d2) <= 1e-6 * sqrt(nn * e3);
                out->quad_par[l][i] = par ? 1u : 0u;
            }
            const double nlen = sqrt((double)h.n[0] * h.n[0] + (double)h.n[1] * h.n[1] + (double)h.n[2] * h.n[2]);
            if (!(nlen > 1e-12) || !(nlen < 1e12) || !(rabs > 0.0) || !(rabs < 1e6)) continue;
            double smin = INFINITY, smax = -INFINITY;
            for (int v = 0; v < 4; ++v) {
                double sv = 0.0;
                for (int k = 0; k < 3; ++k) sv += (double)h.n[k] * ((double)out->lights[l].pts[v][k] - (double)h.o[k]);
                smin = fmin(smin, sv);
                smax = fmax(smax, sv);
            }
            const double need = nlen * (0.01 + 1e-3 * rabs) * 1.01;
            double side = 0.0, near = 0.0;
            if (smin > need) { side = 1.0; near = smin; }
            else if (smax < -need) { side = -1.0; near = -smax; }
            else continue;
            const double noise = 2e-6 * nlen * rabs; // 4x the rounding of p.nor for |p| <= 2 sqrt(3) rabs
            const double theta = 0.5 * 0.01 * near / dmax - noise; // P may be this far on the wrong side of the plane
            out->cull_side[l][i] = (float)side;
            out->cull_thr[l][i] = (float)(-theta) + (theta < 0 ? 1e-30f : 0.0f);
            if ((double)out->cull_thr[l][i] < -theta) out->cull_thr[l][i] = nextafterf(out->cull_thr[l][i], INFINITY);
        }
    }
    return CRT_OK;
}

} // namespace crt

// create_world (main.cu:370-595) restated through the host-side scene classes.
extern "C" int crt_cornell_scene(uint32_t shadowSamples, crt_camera* camera, crt_geometry* geometry8,
                                 crt_light* light1) {
    if (!camera || !geometry8 || !light1) return CRT_ERR_INVALID;
    craytracer::CornellBox box(shadowSamples);
    craytracer::FlatScene flat = craytracer::flatten(box.camera, box.world, box.lights);
    if (flat.geometry.size() != 8 || flat.lights.size() != 1) return CRT_ERR_INVALID;
    *camera = flat.camera;
    for (size_t i = 0; i < 8; ++i) geometry8[i] = flat.geometry[i];
    *light1 = flat.lights[0];
    return CRT_OK;
}
