/*
 * crt_oracle.c -- CPU restatement of CudaRayTracer's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, imports or
 * calls this file; it exists so that tests/, the smoke check and the CPU-baseline
 * leg of the benchmark can compare the HIP kernels against an independent,
 * expression-by-expression restatement of the reference's algorithm.
 *
 * Arithmetic convention ("canonical arithmetic").  The reference is built with
 * nvcc --use_fast_math (CudaRayTracer.vcxproj:98) whose approximate intrinsics
 * cannot be reproduced off an NVIDIA GPU.  This restatement evaluates every
 * expression in the reference's own order with IEEE-754 binary32 operations and
 * no fused contraction:
 *     __fdividef(a,b) -> a / b            rsqrtf(x) -> 1.0f / sqrtf(x)
 *     __saturatef(x)  -> fminf(fmaxf(x,0),1)
 *     __tanf/__powf   -> libm tanf/powf for the per-camera / per-material constants (evaluated on the host
 *                        by the product as well); orc_pow() below for the point light's specular term
 *     __powf(x, 5)    -> ((x*x)*(x*x))*x in Schlick's reflectance (per glass hit; only compared with 1)
 *     __sinf/__cosf   -> orc_sincos() below (a fixed polynomial, so that CPU and
 *                        GPU can agree bit for bit; |err| < 2e-7)
 * Vector helpers follow mstd: dot/length_sq accumulate from 0 in index order
 * (vec.hpp:390-396,428-434), length() = 1/rsqrt (vec.hpp:398-409), normalize()
 * multiplies by 1/len and is a no-op for len == 0 (vec.hpp:411-422,739-769).
 *
 * Pinning status: checked against (i) the known-answer RNG values and the
 * seed-1234 anchor images recorded from the reference's own headers compiled on
 * the host in IEEE arithmetic, and (ii) the reference's committed renders
 * (renders/render0.hdr etc.) statistically -- see tests/test_oracle_golden.py.
 * Bit-parity with real cuRAND output is unpinned: no artefact of the reference
 * records a seed (it seeds with time(NULL), main.cu:952).
 */
#include "crt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ mstd subset */
typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { float c[3][3]; } m3; /* c[column][row], mat.hpp _values[C][R] */

#define EPS 1e-4f    /* MSTD_CUDA_EPSILON, math_types.hpp:78 */
#define EPS_SQ 1e-8f /* MSTD_CUDA_EPSILON_SQ, math_types.hpp:79 */

static inline float rsqrt_c(float x) { return 1.0f / sqrtf(x); }
static inline float fdiv(float a, float b) { return a / b; }
/* math_functions.hpp:75-78 */
static inline int eps_equal(float a, float b, float e) { return fabsf(a - b) < e; }
/* cuda::std::min/max (return first argument on ties / unordered) */
static inline float min_std(float a, float b) { return (b < a) ? b : a; }
static inline float max_std(float a, float b) { return (a < b) ? b : a; }
static inline float saturatef_c(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v3 v3_from(const float* p) { return V3(p[0], p[1], p[2]); }
static inline v4 v4_from(const float* p) { return V4(p[0], p[1], p[2], p[3]); }
static inline float v3_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_adds(v3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }
static inline v3 v3_neg(v3 a) { return v3_scale(a, -1.0f); } /* vec.hpp:819-821 */
static inline float v3_dot(v3 a, v3 b) {
    float r = 0.0f;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
static inline float v3_length_sq(v3 a) { return v3_dot(a, a); }
static inline float v3_length(v3 a) { return fdiv(1.0f, rsqrt_c(v3_length_sq(a))); }
/* vec.hpp:739-769 */
static inline v3 v3_divs(v3 a, float y) {
    if (y == 0.0f) return a;
    float inv = 1.0f / y;
    return v3_scale(a, inv);
}
/* vec.hpp:411-422 */
static inline v3 v3_normalize(v3 a) {
    float len = v3_length(a);
    if (len == 0.0f) return a;
    return v3_divs(a, len);
}
/* vec.hpp:626-636 */
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec.hpp:460-468: I -= (2 * dot(I,n)) * n */
static inline v3 v3_reflect(v3 i, v3 n) {
    float k = 2.0f * v3_dot(i, n);
    return v3_sub(i, v3_scale(n, k));
}
/* vec.hpp:475-491 (CUDA branch) */
static inline v3 v3_refract(v3 d, v3 n, float eta) {
    float cos_theta = min_std(v3_dot(v3_neg(d), n), 1.0f);
    v3 perp = v3_scale(v3_add(d, v3_scale(n, cos_theta)), eta);
    float lsq = v3_length_sq(perp);
    v3 par = v3_divs(v3_neg(n), rsqrt_c(fabsf(1.0f - lsq)));
    return v3_add(perp, par);
}

static inline v4 v4_add(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 v4_mul(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 v4_scale(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline int v4_equal(v4 a, v4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

/* mat.hpp:1481-1492: res[x][y] += a[i][y] * b[x][i], accumulators start at 0 */
static m3 m3_mul(const m3* a, const m3* b) {
    m3 r;
    for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) {
            float s = 0.0f;
            for (int i = 0; i < 3; ++i) s += a->c[i][y] * b->c[x][i];
            r.c[x][y] = s;
        }
    return r;
}
/* mat.hpp:1522-1530 */
static v3 m3_mulv(const m3* m, v3 v) {
    float in[3] = {v.x, v.y, v.z}, out[3];
    for (int y = 0; y < 3; ++y) {
        float s = 0.0f;
        for (int x = 0; x < 3; ++x) s += m->c[x][y] * in[x];
        out[y] = s;
    }
    return V3(out[0], out[1], out[2]);
}
static m3 m3_from_cols(v3 a, v3 b, v3 c) {
    m3 r = {{{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}}};
    return r;
}
/* mat.hpp:1205-1220 */
static m3 m3_transpose(m3 m) {
    m3 r;
    for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) r.c[x][y] = m.c[y][x];
    return r;
}

/* --------------------------------------------------------------------- sin/cos */
/* Stand-in for __sinf/__cosf (main.cu:64-65): argument in [0, 2*pi + ulp].
 * Quadrant reduction k = round(a*2/pi) <= 4 with a two-term pi/2 whose leading
 * part has 17 significant bits (k*PIO2_1 and a - k*PIO2_1 are exact), then the
 * classic degree-9/8 kernels on [-pi/4, pi/4].  Plain mul/add only, evaluated in
 * this order, so any IEEE machine reproduces the same bits. */
void orc_sincos(float a, float* s, float* c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float PIO2_1 = 1.5707855225e+00f;  /* 0x3fc90f80 */
    const float PIO2_1T = 1.0804334124e-05f; /* pi/2 - PIO2_1 */
    float kf = floorf(a * TWO_OVER_PI + 0.5f);
    int k = (int)kf;
    float r = a - kf * PIO2_1;
    r = r - kf * PIO2_1T;
    float z = r * r;
    /* sin(r) ~ r + r*z*(S1 + z*(S2 + z*(S3 + z*S4))) */
    const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f,
                S4 = 2.7557314297e-06f;
    /* cos(r) ~ 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f;
    float ps = S3 + z * S4;
    ps = S2 + z * ps;
    ps = S1 + z * ps;
    float sr = r + (r * z) * ps;
    float pc = C2 + z * C3;
    pc = C1 + z * pc;
    float cr = (1.0f - 0.5f * z) + (z * z) * pc;
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

/* ------------------------------------------------------------------------- pow */
/* Stand-in for __powf in the point light's Blinn-Phong term (Light.hpp:173): x >= 0 (the base is
 * max(n.h, 0)).  __powf is exp2(y * log2(x)) in approximate device arithmetic that only CUDA reproduces; the
 * canonical form here is the same composition out of IEEE operations in a fixed order -- log2 from the
 * atanh series of (m-1)/(m+1) on m in (sqrt(1/2), sqrt(2)], exp2 from the degree-7 Taylor polynomial on
 * [-1/2, 1/2] -- so that CPU and GPU agree bit for bit.  Relative error about 1e-7 * (1 + |y log2 x|);
 * results below 2^-126 are flushed to zero. */
float orc_pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (!(x > 0.0f)) return y > 0.0f ? 0.0f : INFINITY;
    if (x > 3.0e38f) return y > 0.0f ? INFINITY : 0.0f;
    int e = 0;
    if (x < 1.17549435e-38f) { /* subnormal base */
        x = x * 16777216.0f;
        e = -24;
    }
    uint32_t bits;
    memcpy(&bits, &x, 4);
    e += (int)(bits >> 23) - 127;
    bits = (bits & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &bits, 4);
    if (m > 1.41421354f) {
        m = m * 0.5f;
        e += 1;
    }
    const float sr = (m - 1.0f) / (m + 1.0f);
    const float z = sr * sr;
    float p = 0.111111112f;
    p = 0.142857149f + z * p;
    p = 0.2f + z * p;
    p = 0.333333343f + z * p;
    p = 1.0f + z * p;
    const float lg = (float)e + ((2.0f * sr) * p) * 1.44269502f;
    const float t = y * lg;
    if (!(t < 128.0f)) return INFINITY;
    if (t < -126.0f) return 0.0f;
    const float nf = floorf(t + 0.5f);
    const float q = (t - nf) * 0.693147182f;
    float w = 1.98412701e-4f;
    w = 1.38888892e-3f + q * w;
    w = 8.33333377e-3f + q * w;
    w = 4.16666679e-2f + q * w;
    w = 0.166666672f + q * w;
    w = 0.5f + q * w;
    w = 1.0f + q * w;
    w = 1.0f + q * w;
    const int n = (int)nf, n1 = n / 2, n2 = n - n1;
    uint32_t b1 = (uint32_t)(n1 + 127) << 23, b2 = (uint32_t)(n2 + 127) << 23;
    float s1, s2;
    memcpy(&s1, &b1, 4);
    memcpy(&s2, &b2, 4);
    return (w * s1) * s2;
}

/* ------------------------------------------------------------------------- RNG */
/* cuRAND XORWOW as used through curand_init(seed+idx, 0, 0, &s) (main.cu:324) and
 * curand_uniform (main.cu:177-178, AreaLight.hpp:342-343).  cuRAND's source is
 * not part of the reference tree (CUDA toolkit 12.9, CudaRayTracer.vcxproj:35);
 * this restates its published algorithm: seed scrambling, the xorshift step with
 * a Weyl sequence of stride 362437, and the (0,1] float mapping. */
void orc_rng_init(uint64_t seed, uint32_t st[6]) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st[5] = 6615241u + t1 + t0;
    st[0] = 123456789u + t0;
    st[1] = 362436069u ^ t0;
    st[2] = 521288629u + t1;
    st[3] = 88675123u ^ t1;
    st[4] = 5783321u + t0;
}
uint32_t orc_rng_next(uint32_t st[6]) {
    uint32_t t = st[0] ^ (st[0] >> 2);
    st[0] = st[1];
    st[1] = st[2];
    st[2] = st[3];
    st[3] = st[4];
    st[4] = (st[4] ^ (st[4] << 4)) ^ (t ^ (t << 1));
    st[5] += 362437u;
    return st[4] + st[5];
}
float orc_rng_uniform(uint32_t st[6]) {
    return (float)orc_rng_next(st) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* ------------------------------------------------------------------ ray, hits */
typedef struct {
    v3 o, d;
    float dist;
} ray_t;
typedef struct {
    int isHit;
    v3 p, n;
    float t;
    const crt_material* mat;
} hit_t;

static inline ray_t ray_default(void) { /* Ray.hpp:30-32 */
    ray_t r = {{0, 0, 0}, {0, 0, -1}, 0.0f};
    return r;
}
/* Ray.hpp:37-43 */
static inline ray_t ray_make2(v3 o, v3 d) {
    ray_t r;
    r.o = o;
    r.dist = 0.0f;
    r.d = eps_equal(v3_length_sq(d), 0.0f, EPS_SQ) ? d : v3_normalize(d);
    return r;
}
/* Ray.hpp:45-52,71-78 */
static inline ray_t ray_make3(v3 o, v3 d, float dist) {
    ray_t r = ray_make2(o, d);
    r.dist = max_std(0.0f, dist);
    return r;
}
/* Ray.hpp:62-69 */
static inline void ray_set_direction(ray_t* r, v3 d) {
    if (eps_equal(v3_length_sq(d), 0.0f, EPS_SQ)) return;
    r->d = v3_normalize(d);
}
static inline hit_t hit_default(void) { /* Ray.hpp:19-26 */
    hit_t h;
    h.isHit = 0;
    h.p = V3(0, 0, 0);
    h.n = V3(0, 1, 0);
    h.t = FLT_MAX;
    h.mat = NULL;
    return h;
}

/* Sphere::hit, Sphere.hpp:40-122 (the UV written at :111-119 is never read) */
static int sphere_hit(const crt_geometry* g, const ray_t* ray, hit_t* hit) {
    float t0, t1;
    v3 center = v3_from(g->center);
    float radius = g->radius;
    v3 L = v3_sub(ray->o, center);
    float a = v3_dot(ray->d, ray->d);
    float b = 2.0f * v3_dot(ray->d, L);
    float c = v3_dot(L, L) - radius * radius;
    float discr = b * b - 4.0f * a * c;
    if (discr < 0.0f) return 0;
    else if (eps_equal(discr, 0.0f, EPS)) t0 = t1 = -0.5f * fdiv(b, a);
    else {
        float discr_sq = fdiv(1.0f, rsqrt_c(discr));
        float q = (b > 0) ? -0.5f * (b + discr_sq) : -0.5f * (b - discr_sq);
        t0 = fdiv(q, a);
        t1 = fdiv(c, q);
    }
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0.0f) {
        t0 = t1;
        if (t0 < 0.0f) return 0;
    }
    if (ray->dist > 0.0f && t0 > ray->dist) return 0;
    hit->isHit = 1;
    hit->p = v3_add(ray->o, v3_scale(ray->d, t0));
    hit->mat = &g->mat;
    hit->t = t0;
    hit->n = v3_normalize(v3_sub(hit->p, center));
    return 1;
}

/* Shared body of Quad::hit (Quad.hpp:196-295) and AreaLight::_hittedLight
 * (AreaLight.hpp:378-481): bilinear-quad intersection; p is the ray origin
 * relative to corner 0; a,b,c are corners 1,3,2 relative to corner 0.
 * `half_first` selects _hittedLight's 0.5f * 1.0f / k2 (AreaLight.hpp:456). */
static int quad_uvt(v3 a, v3 b, v3 c, v3 p, const ray_t* ray, int light_form, float* out_t,
                    float* out_u, float* out_v) {
    static const int lut[4] = {1, 2, 0, 1};
    v3 nor = v3_cross(a, b);
    float t = -fdiv(v3_dot(p, nor), v3_dot(ray->d, nor));
    if (t < 0.0f || (ray->dist > 0.0f && t > ray->dist)) return 0;
    v3 pos = v3_add(p, v3_scale(ray->d, t));
    v3 mor = V3(fabsf(nor.x), fabsf(nor.y), fabsf(nor.z));
    int id = (mor.x > mor.y && mor.x > mor.z) ? 0 : (mor.y > mor.z) ? 1 : 2;
    int idu = lut[id], idv = lut[id + 1];
    v2 kp = {v3_get(pos, idu), v3_get(pos, idv)};
    v2 ka = {v3_get(a, idu), v3_get(a, idv)};
    v2 kb = {v3_get(b, idu), v3_get(b, idv)};
    v2 kc = {v3_get(c, idu), v3_get(c, idv)};
    v2 kg = {(kc.x - kb.x) - ka.x, (kc.y - kb.y) - ka.y};
    float k0 = kp.x * kb.y - kp.y * kb.x;
    v2 kcb = {kc.x - kb.x, kc.y - kb.y};
    float k2 = kcb.x * ka.y - kcb.y * ka.x;
    float k1 = (kp.x * kg.y - kp.y * kg.x) - v3_get(nor, id);
    float u, v;
    if (eps_equal(k2, 0.0f, EPS)) {
        v = -fdiv(k0, k1);
        u = fdiv(kp.x * ka.y - kp.y * ka.x, k1);
    } else {
        float w = k1 * k1 - 4.0f * k0 * k2;
        if (w < 0.0f) return 0;
        w = fdiv(1.0f, rsqrt_c(w));
        float ik2 = light_form ? 0.5f * fdiv(1.0f, k2) : fdiv(1.0f, 2.0f * k2);
        v = (-k1 - w) * ik2;
        if (v < 0.0f || v > 1.0f) v = (-k1 + w) * ik2;
        u = fdiv(kp.x - ka.x * v, kb.x + kg.x * v);
    }
    if (min_std(u, v) < 0.0f || max_std(u, v) > 1.0f) return 0;
    *out_t = t;
    *out_u = u;
    *out_v = v;
    return 1;
}

/* Quad::hit, Quad.hpp:196-345 (the UV interpolation at :327-342 is never read) */
static int quad_hit(const crt_geometry* g, const ray_t* ray, hit_t* hit) {
    v3 v0p = v3_from(g->v[0].pos);
    v3 a = v3_sub(v3_from(g->v[1].pos), v0p);
    v3 b = v3_sub(v3_from(g->v[3].pos), v0p);
    v3 c = v3_sub(v3_from(g->v[2].pos), v0p);
    v3 p = v3_sub(ray->o, v0p);
    float t, u, v;
    if (!quad_uvt(a, b, c, p, ray, 0, &t, &u, &v)) return 0;
    hit->isHit = 1;
    hit->p = v3_add(ray->o, v3_scale(ray->d, t));
    hit->mat = &g->mat;
    hit->t = t;
    float l0 = (1 - u) * (1 - v);
    float l1 = u * (1 - v);
    float l2 = u * v;
    float l3 = (1 - u) * v;
    const float *n0 = g->v[0].normal, *n1 = g->v[1].normal, *n2 = g->v[2].normal,
                *n3 = g->v[3].normal;
    hit->n = V3(l0 * n0[0] + l1 * n1[0] + l2 * n2[0] + l3 * n3[0],
                l0 * n0[1] + l1 * n1[1] + l2 * n2[1] + l3 * n3[1],
                l0 * n0[2] + l1 * n1[2] + l2 * n2[2] + l3 * n3[2]);
    return 1;
}

/* GeometryList::hit, GeometryList.hpp:22-35 */
static int world_hit(const crt_scene_desc* sc, const ray_t* ray, hit_t* hit) {
    hit_t temp = hit_default();
    int hit_anything = 0;
    float closest = FLT_MAX;
    for (uint32_t i = 0; i < sc->n_geometry; ++i) {
        const crt_geometry* g = &sc->geometry[i];
        int h = (g->kind == CRT_GEOM_SPHERE) ? sphere_hit(g, ray, &temp) : quad_hit(g, ray, &temp);
        if (h) {
            if (temp.t > closest) continue;
            hit_anything = 1;
            closest = temp.t;
            *hit = temp;
        }
    }
    return hit_anything;
}

/* ---------------------------------------------------------------------- lights */
typedef struct {
    const crt_scene_desc* sc;
    const float* ltc1;
    const float* ltc2;
    uint32_t* rng;
} ctx_t;

/* math_functions.hpp:106-133; its result is only ever compared with > 1.0f.  __powf(x, 5) is an
 * approximation that only the CUDA device reproduces; the canonical form here is ((x*x)*(x*x))*x.
 * For cosine in [0,1] the result cannot exceed 1; it does at grazing hits where the second
 * normalisation of the direction (main.cu:91) flips the sign of d.n and cosine comes out as -6e-8. */
static float reflectance_c(float cosine, float ri) {
    float r0 = fdiv(1.0f - ri, 1.0f + ri);
    r0 = r0 * r0;
    float x = 1.0f - cosine, x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}

/* One coordinate of Texture::_imageSpaceCoordinates, Texture.hpp:129-152 (wrap: 0 CLAMP, 1 MIRROR, 2 REPEAT,
 * Texture.hpp:25-29; fract is x - floor(x), math_functions.hpp:94-97). */
static float texture_wrap(float f, int wrap) {
    if (!(f < 0.0f || f > 1.0f)) return f;
    if (wrap == 0) return saturatef_c(f);
    if (wrap == 2) return f - floorf(f);
    return fabsf((f - floorf(f)) - 1.0f);
}
/* Texture::_imageSpaceCoordinates, Texture.hpp:126-185: the texel a uv pair reads. */
void orc_texture_texel(float u, float v, int wrap_h, int wrap_v, uint32_t width, uint32_t height, uint32_t xy[2]) {
    xy[0] = (uint32_t)(texture_wrap(u, wrap_h) * (float)(width - 1u));
    xy[1] = (uint32_t)(texture_wrap(v, wrap_v) * (float)(height - 1u));
}

/* Texture::sample of an LTC table (64 x 64, CLAMP both ways: AreaLight.hpp:51-54), Texture.hpp:249-252,211-237 */
static v4 ltc_sample(const float* tab, v2 uv) {
    uint32_t xy[2];
    orc_texture_texel(uv.x, uv.y, 0, 0, 64u, 64u, xy);
    const uint32_t x = xy[0], y = xy[1];
    if (x >= 64u || y >= 64u) return V4(0, 0, 0, 0);
    size_t pos0 = ((size_t)y * 64u + (size_t)x) * 4u;
    return V4(tab[pos0], tab[pos0 + 1], tab[pos0 + 2], tab[pos0 + 3]);
}

/* AreaLight::_hittedLight, AreaLight.hpp:378-481 */
static int light_hitted(const crt_light* L, const ray_t* ray) {
    v3 p0 = v3_from(L->points[0]);
    v3 a = v3_sub(v3_from(L->points[1]), p0);
    v3 b = v3_sub(v3_from(L->points[3]), p0);
    v3 c = v3_sub(v3_from(L->points[2]), p0);
    v3 p = v3_sub(v3_sub(ray->o, p0), v3_from(L->pos));
    float t, u, v;
    return quad_uvt(a, b, c, p, ray, 1, &t, &u, &v);
}

/* Ray::isPointOnRay, Ray.hpp:80-105 (Light::_hittedLight, Light.hpp:88-90) */
static int point_on_ray(const ray_t* ray, v3 point) {
    v3 v = v3_sub(point, ray->o);
    v3 cr = v3_cross(ray->d, v);
    if (v3_length_sq(cr) >= EPS_SQ) return 0;
    float d = v3_dot(v, ray->d);
    if (eps_equal(ray->dist, 0.0f, EPS)) return d >= 0.0f;
    return d >= 0.0f && d <= ray->dist;
}

/* Light::_isCovered, Light.hpp:35-86 */
static int orc_trace_enabled(void);
static int is_covered(const ctx_t* cx, const crt_light* L, const ray_t* ray_in, unsigned iterations) {
    hit_t hit = hit_default();
    ray_t r = *ray_in;
    unsigned i = 0;
    do {
        if (!world_hit(cx->sc, &r, &hit)) return 0;
        if (hit.isHit && hit.mat->type != CRT_MAT_REFRACTIVE) {
            i = iterations;
            continue;
        } else {
            if (iterations == 0) return 0;
            int front_face = v3_dot(r.d, hit.n) < 0.0f;
            if (orc_trace_enabled())
                fprintf(stderr, "oracle:     glass walk %u: from (%.9g %.9g %.9g) hit at (%.9g %.9g %.9g) t=%.9g front %d\n", i, r.o.x,
                        r.o.y, r.o.z, hit.p.x, hit.p.y, hit.p.z, hit.t, front_face);
            v3 norm = front_face ? hit.n : v3_neg(hit.n);
            float ratio = hit.mat->refractIndex / 1.0f;
            if (front_face) ratio = fdiv(1.0f, ratio);
            v3 rd = v3_normalize(r.d);
            float cos_theta = fminf(v3_dot(v3_neg(rd), norm), 1.0f);
            float sin_theta = fdiv(1.0f, rsqrt_c(1.0f - cos_theta * cos_theta));
            int cannot_refract = ratio * sin_theta > 1.0f || reflectance_c(cos_theta, ratio) > 1.0f;
            if (cannot_refract) return 1;
            v3 dir = v3_refract(rd, norm, ratio);
            ray_t refracted = ray_make2(v3_add(hit.p, v3_scale(dir, 0.01f)), dir);
            if (!front_face)
                return L->kind == CRT_LIGHT_AREA ? !light_hitted(L, &refracted)
                                                 : !point_on_ray(&refracted, v3_from(L->pos));
            r = refracted;
        }
        ++i;
    } while (i < iterations);
    return hit.isHit;
}

/* AreaLight::_randomPoint, AreaLight.hpp:340-375 */
static v3 area_random_point(const ctx_t* cx, const crt_light* L) {
    v3 pos = v3_from(L->pos);
    float u = orc_rng_uniform(cx->rng);
    float v = orc_rng_uniform(cx->rng);
    v3 p1 = v3_from(L->points[1]);
    v3 p3 = v3_from(L->points[3]);
    if (u + v < 1.0f) {
        v3 p0 = v3_from(L->points[0]);
        return v3_add(v3_add(v3_add(p0, v3_scale(v3_sub(p1, p0), u)), v3_scale(v3_sub(p3, p0), v)), pos);
    } else {
        u = 1.0f - u;
        v = 1.0f - v;
        v3 p2 = v3_from(L->points[2]);
        return v3_add(v3_add(v3_add(p2, v3_scale(v3_sub(p3, p2), u)), v3_scale(v3_sub(p1, p2), v)), pos);
    }
}

/* Debugging aid: with ORC_TRACE set in the environment every shaded point's shadow count is printed
 * (render one pixel with the region arguments of orc_render). */
static int orc_trace_enabled(void) {
    static int state = -1;
    if (state < 0) state = getenv("ORC_TRACE") != NULL;
    return state;
}

/* AreaLight::_calculateVisibility, AreaLight.hpp:484-529 */
static float area_visibility(const ctx_t* cx, const crt_light* L, v3 position) {
    unsigned shadowed = 0;
    for (unsigned i = 0; i < L->shadowSamples; ++i) {
        v3 sample = area_random_point(cx, L);
        v3 dir = v3_sub(sample, position);
        float dist = v3_length(dir);
        if (!eps_equal(dist, 0.0f, EPS)) dir = v3_normalize(dir);
        ray_t ray = ray_make3(v3_add(position, v3_scale(dir, 0.01f)), dir, dist);
        const int covered = is_covered(cx, L, &ray, 10u);
        if (covered) ++shadowed;
        if (orc_trace_enabled())
            fprintf(stderr, "oracle:   sample %u X=(%.9g %.9g %.9g) dist %.9g covered %d\n", i, sample.x, sample.y, sample.z, dist, covered);
    }
    if (orc_trace_enabled())
        fprintf(stderr, "oracle: P=(%.9g %.9g %.9g) light at (%.9g %.9g %.9g): %u of %u shadow rays covered\n", position.x,
                position.y, position.z, L->pos[0], L->pos[1], L->pos[2], shadowed, L->shadowSamples);
    return 1.0f - fdiv((float)shadowed, (float)L->shadowSamples);
}

/* AreaLight::_integrateEdgeVec, AreaLight.hpp:72-98 */
static v3 integrate_edge_vec(v3 v1, v3 v2) {
    float x = v3_dot(v1, v2);
    float y = fabsf(x);
    float a = 0.8543985f + (0.4965155f + 0.0145206f * y) * y;
    float b = 3.4175940f + (4.1616724f + y) * y;
    float v = fdiv(a, b);
    float ts = (x > 0.0f) ? v : 0.5f * rsqrt_c(max_std(1.0f - x * x, EPS)) - v;
    return v3_scale(v3_cross(v1, v2), ts);
}

/* AreaLight::_clipQuadToHorizon, AreaLight.hpp:104-214: clips the transformed quad against z = 0; returns the
 * number of vertices (0, 3, 4 or 5) and closes the polygon (L[n] = L[0] for n = 3, 4). */
static int clip_quad_to_horizon(v3 L[5]) {
#define ORC_CLIP(keep, drop) v3_add(v3_scale(keep, -(drop).z), v3_scale(drop, (keep).z)) /* -drop.z * keep + keep.z * drop */
    int config = 0;
    if (L[0].z > 0.0f) config += 1;
    if (L[1].z > 0.0f) config += 2;
    if (L[2].z > 0.0f) config += 4;
    if (L[3].z > 0.0f) config += 8;
    int n = 0;
    switch (config) {
    case 0: break;
    case 1: n = 3; L[1] = ORC_CLIP(L[0], L[1]); L[2] = ORC_CLIP(L[0], L[3]); break;
    case 2: n = 3; L[0] = ORC_CLIP(L[1], L[0]); L[2] = ORC_CLIP(L[1], L[2]); break;
    case 3: n = 4; L[2] = ORC_CLIP(L[1], L[2]); L[3] = ORC_CLIP(L[0], L[3]); break;
    case 4: n = 3; L[0] = ORC_CLIP(L[2], L[3]); L[1] = ORC_CLIP(L[2], L[1]); break;
    case 5: n = 0; break;
    case 6: n = 4; L[0] = ORC_CLIP(L[1], L[0]); L[3] = ORC_CLIP(L[2], L[3]); break;
    case 7: n = 5; L[4] = ORC_CLIP(L[0], L[3]); L[3] = ORC_CLIP(L[2], L[3]); break;
    case 8: n = 3; L[0] = ORC_CLIP(L[3], L[0]); L[1] = ORC_CLIP(L[3], L[2]); L[2] = L[3]; break;
    case 9: n = 4; L[1] = ORC_CLIP(L[0], L[1]); L[2] = ORC_CLIP(L[3], L[2]); break;
    case 10: n = 0; break;
    case 11: n = 5; L[4] = L[3]; L[3] = ORC_CLIP(L[3], L[2]); L[2] = ORC_CLIP(L[1], L[2]); break;
    case 12: n = 4; L[1] = ORC_CLIP(L[2], L[1]); L[0] = ORC_CLIP(L[3], L[0]); break;
    case 13: n = 5; L[4] = L[3]; L[3] = L[2]; L[2] = ORC_CLIP(L[2], L[1]); L[1] = ORC_CLIP(L[0], L[1]); break;
    case 14: n = 5; L[4] = ORC_CLIP(L[3], L[0]); L[0] = ORC_CLIP(L[1], L[0]); break;
    default: n = 4; break; /* 15: nothing below the horizon */
    }
#undef ORC_CLIP
    if (n == 3) L[3] = L[0];
    if (n == 4) L[4] = L[0];
    return n;
}

/* The horizon-clipping branch of AreaLight::_evaluateLTC, AreaLight.hpp:298-334, on the transformed corners.
 * The reference never takes it (AreaLight::_clipless is true and has no setter, AreaLight.hpp:42); crt_light's
 * horizonClip selects it. */
static float ltc_clipped_sum(const v3 corners[4], int two_sided) {
    v3 L[5] = {corners[0], corners[1], corners[2], corners[3], V3(0, 0, 0)};
    int n = clip_quad_to_horizon(L);
    if (n == 0) return 0.0f; /* AreaLight.hpp:302-303 */
    for (int i = 0; i < 5; ++i) L[i] = eps_equal(v3_length_sq(L[i]), 0.0f, EPS_SQ) ? L[i] : v3_normalize(L[i]);
    float sum = 0.0f;
    sum += integrate_edge_vec(L[0], L[1]).z;
    sum += integrate_edge_vec(L[1], L[2]).z;
    sum += integrate_edge_vec(L[2], L[3]).z;
    if (n >= 4) sum += integrate_edge_vec(L[3], L[4]).z;
    if (n == 5) sum += integrate_edge_vec(L[4], L[0]).z;
    return two_sided ? fabsf(sum) : max_std(0.0f, sum);
}
/* Test hook: the clipped integral of four already transformed corners. */
float orc_ltc_clipped(const float corners[4][3], int two_sided) {
    v3 c[4] = {v3_from(corners[0]), v3_from(corners[1]), v3_from(corners[2]), v3_from(corners[3])};
    return ltc_clipped_sum(c, two_sided);
}

/* AreaLight::_evaluateLTC, AreaLight.hpp:216-336 */
static float evaluate_ltc(const ctx_t* cx, const crt_light* L, v3 norm, v3 view, v3 frag,
                          const m3* minV, const v3 pts[4]) {
    v3 T1 = v3_sub(view, v3_scale(norm, v3_dot(view, norm)));
    if (!eps_equal(v3_length_sq(T1), 0.0f, EPS_SQ)) T1 = v3_normalize(T1);
    v3 T2 = v3_cross(norm, T1);
    m3 basis = m3_transpose(m3_from_cols(T1, T2, norm));
    m3 Minv = m3_mul(minV, &basis);
    v3 Lv[4];
    for (int i = 0; i < 4; ++i) Lv[i] = m3_mulv(&Minv, v3_sub(pts[i], frag));
    if (L->horizonClip) return ltc_clipped_sum(Lv, L->twoSided != 0);
    v3 dir = v3_sub(pts[0], frag);
    v3 lightNormal = v3_cross(v3_sub(pts[1], pts[0]), v3_sub(pts[3], pts[0]));
    int behind = v3_dot(dir, lightNormal) < 0.0f;
    for (int i = 0; i < 4; ++i)
        Lv[i] = eps_equal(v3_length_sq(Lv[i]), 0.0f, EPS_SQ) ? Lv[i] : v3_normalize(Lv[i]);
    v3 vsum = V3(0, 0, 0);
    vsum = v3_add(vsum, integrate_edge_vec(Lv[0], Lv[1]));
    vsum = v3_add(vsum, integrate_edge_vec(Lv[1], Lv[2]));
    vsum = v3_add(vsum, integrate_edge_vec(Lv[2], Lv[3]));
    vsum = v3_add(vsum, integrate_edge_vec(Lv[3], Lv[0]));
    float len = v3_length(vsum);
    float z = !eps_equal(len, 0.0f, EPS) ? vsum.z / len : 0.0f;
    if (behind) z = -z;
    v2 uv = {z * 0.5f + 0.5f, len};
    const float LUT_SCALE = (64.0f - 1.0f) / 64.0f, LUT_BIAS = 0.5f / 64.0f; /* AreaLight.hpp:47-49 */
    uv.x = uv.x * LUT_SCALE + LUT_BIAS;
    uv.y = uv.y * LUT_SCALE + LUT_BIAS;
    float scale = ltc_sample(cx->ltc2, uv).w;
    float sum = len * scale;
    if (!behind && !L->twoSided) sum = 0.0f;
    return sum;
}

typedef struct {
    v3 fragPos, norm, viewDir;
    const crt_material* mat;
} light_input_t;

/* AreaLight::calculateColor, AreaLight.hpp:571-650 */
static v4 area_calculate_color(const ctx_t* cx, const crt_light* L, const light_input_t* in) {
    v3 lcol = V3(L->color[0], L->color[1], L->color[2]);
    float intensity = L->intensity;
    v3 pos = v3_from(L->pos);
    v3 ambient = v3_mul(V3(in->mat->ambient[0], in->mat->ambient[1], in->mat->ambient[2]), lcol);
    if (!L->initialized) return V4(ambient.x, ambient.y, ambient.z, 1.0f); /* AreaLight.hpp:586 */
    float visibility = area_visibility(cx, L, in->fragPos);
    v3 pts[4];
    for (int i = 0; i < 4; ++i) pts[i] = v3_add(pos, v3_from(L->points[i]));
    v3 dcol = V3(in->mat->diffuse[0], in->mat->diffuse[1], in->mat->diffuse[2]);
    v3 scol = V3(in->mat->specular[0], in->mat->specular[1], in->mat->specular[2]);
    float roughness = 1.0f - 0.25f * powf(in->mat->shininess, 0.2f);
    float ndotv = saturatef_c(v3_dot(in->norm, in->viewDir));
    const float LUT_SCALE = (64.0f - 1.0f) / 64.0f, LUT_BIAS = 0.5f / 64.0f;
    v2 uv = {roughness, fdiv(1.0f, rsqrt_c(1.0f - ndotv))};
    uv.x = uv.x * LUT_SCALE + LUT_BIAS;
    uv.y = uv.y * LUT_SCALE + LUT_BIAS;
    v4 t1 = ltc_sample(cx->ltc1, uv);
    v4 t2 = ltc_sample(cx->ltc2, uv);
    m3 Minv = m3_from_cols(V3(t1.x, 0.0f, t1.y), V3(0.0f, 1.0f, 0.0f), V3(t1.z, 0.0f, t1.w));
    m3 ident = m3_from_cols(V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1));
    float d = evaluate_ltc(cx, L, in->norm, in->viewDir, in->fragPos, &ident, pts);
    v3 diffuse = v3_mul(v3_mul(lcol, V3(d, d, d)), dcol);
    float s = evaluate_ltc(cx, L, in->norm, in->viewDir, in->fragPos, &Minv, pts);
    v3 spec = V3(s, s, s);
    v3 fres = v3_add(v3_scale(scol, t2.x), v3_scale(v3_sub(V3(1, 1, 1), scol), t2.y));
    spec = v3_mul(spec, fres);
    v3 specular = v3_mul(lcol, spec);
    v3 result = v3_add(ambient, v3_scale(v3_scale(v3_add(diffuse, specular), intensity), visibility));
    return V4(result.x, result.y, result.z, 1.0f);
}

/* Light::calculateColor (point light, Blinn-Phong), Light.hpp:92-111,119-182 */
static v4 point_calculate_color(const ctx_t* cx, const crt_light* L, const light_input_t* in) {
    v3 lcol = V3(L->color[0], L->color[1], L->color[2]);
    float intensity = L->intensity;
    v3 pos = v3_from(L->pos);
    v3 ambient = v3_mul(V3(in->mat->ambient[0], in->mat->ambient[1], in->mat->ambient[2]), lcol);
    /* _calculateVisibility, Light.hpp:92-111 */
    v3 rayDir = v3_sub(pos, in->fragPos);
    float lightDist = v3_length(rayDir);
    if (!eps_equal(lightDist, 0.0f, EPS)) rayDir = v3_normalize(rayDir);
    ray_t ray = ray_make3(v3_add(in->fragPos, v3_scale(rayDir, 0.01f)), rayDir, lightDist);
    if (is_covered(cx, L, &ray, 10u)) return V4(ambient.x, ambient.y, ambient.z, 1.0f);
    v3 lightDir = v3_sub(pos, in->fragPos);
    if (!eps_equal(v3_length_sq(lightDir), 0.0f, EPS_SQ)) lightDir = v3_normalize(lightDir);
    float diff = max_std(v3_dot(in->norm, lightDir), 0.0f);
    v3 mdiff = V3(in->mat->diffuse[0], in->mat->diffuse[1], in->mat->diffuse[2]);
    v3 mspec = V3(in->mat->specular[0], in->mat->specular[1], in->mat->specular[2]);
    v3 diffuse = v3_scale(v3_mul(lcol, v3_scale(mdiff, diff)), intensity);
    v3 halfway = v3_normalize(v3_add(lightDir, in->viewDir));
    float spec = orc_pow(max_std(v3_dot(in->norm, halfway), 0.0f), in->mat->shininess);
    v3 specular = v3_scale(v3_mul(lcol, v3_scale(mspec, spec)), intensity);
    v3 result = v3_add(v3_add(ambient, diffuse), specular);
    return V4(result.x, result.y, result.z, 1.0f);
}

/* LightList::calculateColor, LightList.hpp:23-36 */
static v4 lights_calculate_color(const ctx_t* cx, const light_input_t* in) {
    v4 color = V4(0, 0, 0, 0);
    for (uint32_t i = 0; i < cx->sc->n_lights; ++i) {
        const crt_light* L = &cx->sc->lights[i];
        v4 c = (L->kind == CRT_LIGHT_AREA) ? area_calculate_color(cx, L, in)
                                           : point_calculate_color(cx, L, in);
        color = v4_add(color, c);
    }
    return color;
}

/* ------------------------------------------------------------------ integrator */
/* main.cu:50-58 */
static void surface_coordinate_system(v3 n, v3* Nx, v3* Nz) {
    if (fabsf(n.x) > fabsf(n.y))
        *Nx = v3_scale(V3(n.z, 0.0f, -n.x), rsqrt_c(n.x * n.x + n.z * n.z));
    else
        *Nx = v3_scale(V3(0.0f, -n.z, n.y), rsqrt_c(n.y * n.y + n.z * n.z));
    *Nz = v3_cross(n, *Nx);
}
/* main.cu:60-74 */
static void hemisphere_ray(float r1, float r2, v3 hp, v3 n, v3 Nx, v3 Nz, v3* o, v3* d) {
    const float PI_2 = 6.28318530717958647692f; /* MSTD_CUDA_PI_2, math_types.hpp:77 */
    float sinTheta = 1.0f - r1 * r1;
    float phi = r2 * PI_2;
    float sp, cp;
    orc_sincos(phi, &sp, &cp);
    float x = sinTheta * cp;
    float z = sinTheta * sp;
    v3 w = V3(x * Nz.x + r1 * n.x + z * Nx.x, x * Nz.y + r1 * n.y + z * Nx.y,
              x * Nz.z + r1 * n.z + z * Nx.z);
    *o = v3_add(hp, v3_scale(w, 0.01f));
    *d = w;
}
/* main.cu:76-81 */
static void reflect_ray(const ray_t* in, v3 hp, v3 hn, v3* o, v3* d) {
    v3 I = v3_normalize(in->d);
    v3 rd = v3_reflect(I, v3_normalize(hn));
    *o = v3_add(hp, v3_scale(rd, 0.01f));
    *d = rd;
}
/* main.cu:83-108 */
static void refraction_ray(const ray_t* in, v3 hp, v3 hn, float refIndex, v3* o, v3* d) {
    int front_face = v3_dot(in->d, hn) < 0.0f;
    v3 norm = front_face ? hn : v3_neg(hn);
    float ratio = refIndex / 1.0f;
    if (front_face) ratio = fdiv(1.0f, ratio);
    v3 rd = v3_normalize(in->d);
    float cos_theta = fminf(v3_dot(v3_neg(rd), norm), 1.0f);
    float sin_theta = fdiv(1.0f, rsqrt_c(1.0f - cos_theta * cos_theta));
    int cannot_refract = ratio * sin_theta > 1.0f || reflectance_c(cos_theta, ratio) > 1.0f;
    v3 dir = cannot_refract ? v3_reflect(rd, norm) : v3_refract(rd, norm, ratio);
    *o = v3_add(hp, v3_scale(dir, 0.01f));
    *d = dir;
}

#define MAX_STACK 128 /* main.cu:111 */
typedef struct {
    v3 o, d;
    v4 att;
    unsigned ref_left, depth;
} raystate_t;

/* color(), main.cu:110-227 */
static v4 color(const ctx_t* cx, v3 ro, v3 rd, unsigned ref_iter, unsigned gl_iter, unsigned ind_rays,
                orc_counters* cnt) {
    const v4 sky = V4(0, 0, 0, 1); /* Color::black(), Color.hpp:51 */
    raystate_t stack[MAX_STACK];
    size_t sp = 0;
    stack[sp].o = ro;
    stack[sp].d = rd;
    stack[sp].att = V4(1, 1, 1, 1);
    stack[sp].ref_left = ref_iter;
    stack[sp].depth = gl_iter;
    ++sp;
    v4 final = V4(0, 0, 0, 0);
    ray_t ray = ray_default();
    while (sp > 0) {
        raystate_t st = stack[--sp];
        if (st.ref_left == 0) continue;
        ray.o = st.o;
        ray_set_direction(&ray, st.d);
        hit_t hit = hit_default();
        if (cnt) cnt->rays++;
        if (!world_hit(cx->sc, &ray, &hit)) {
            final = v4_add(final, v4_mul(st.att, sky));
            continue;
        }
        if (orc_trace_enabled())
            fprintf(stderr, "oracle: ray o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) depth %u ref %u: hit t=%.9g at (%.9g %.9g %.9g) material %u\n",
                    ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, st.depth, st.ref_left, hit.t, hit.p.x, hit.p.y, hit.p.z,
                    (unsigned)hit.mat->type);
        v4 mat_diff = v4_from(hit.mat->diffuse);
        float mat_ref_idx = hit.mat->refractIndex;
        switch (hit.mat->type) {
            case CRT_MAT_DIFFUSE: {
                light_input_t in;
                in.fragPos = hit.p;
                in.norm = hit.n;
                in.viewDir = v3_neg(st.d);
                in.mat = hit.mat;
                if (cnt) cnt->shade_points++;
                v4 direct = lights_calculate_color(cx, &in);
                if (st.depth > 0u) {
                    v3 Nx, Nz;
                    surface_coordinate_system(hit.n, &Nx, &Nz);
                    for (unsigned n = 0; n < ind_rays; ++n) {
                        float r1 = orc_rng_uniform(cx->rng);
                        float r2 = orc_rng_uniform(cx->rng);
                        if (cnt) cnt->gi_children++;
                        if (sp < MAX_STACK) {
                            v3 o, d;
                            hemisphere_ray(r1, r2, hit.p, hit.n, Nx, Nz, &o, &d);
                            float inv = 1.0f / (float)ind_rays;
                            stack[sp].o = o;
                            stack[sp].d = d;
                            stack[sp].att = v4_scale(v4_mul(v4_scale(mat_diff, 2.0f * r1), st.att), inv);
                            stack[sp].ref_left = ref_iter;
                            stack[sp].depth = st.depth - 1u;
                            ++sp;
                        }
                    }
                }
                final = v4_add(final, v4_mul(st.att, direct));
                break;
            }
            case CRT_MAT_REFLECT: {
                if (sp < MAX_STACK) {
                    v3 o, d;
                    reflect_ray(&ray, hit.p, hit.n, &o, &d);
                    stack[sp].o = o;
                    stack[sp].d = d;
                    stack[sp].att = v4_mul(st.att, mat_diff);
                    stack[sp].ref_left = st.ref_left - 1u;
                    stack[sp].depth = st.depth;
                    ++sp;
                }
                break;
            }
            case CRT_MAT_REFRACTIVE: {
                if (sp < MAX_STACK) {
                    v3 o, d;
                    refraction_ray(&ray, hit.p, hit.n, mat_ref_idx, &o, &d);
                    stack[sp].o = o;
                    stack[sp].d = d;
                    stack[sp].att = v4_mul(st.att, mat_diff);
                    stack[sp].ref_left = st.ref_left - 1u;
                    stack[sp].depth = st.depth;
                    ++sp;
                }
                break;
            }
            default: break;
        }
    }
    return final;
}

/* Camera::getRay, Camera.hpp:86-144 */
static ray_t camera_get_ray(const crt_camera* cam, float x, float y, float width, float height) {
    v3 front = v3_from(cam->front), pos = v3_from(cam->pos);
    v3 rightDir = v3_normalize(v3_cross(front, V3(0.0f, 1.0f, 0.0f)));
    v3 upDir = v3_normalize(v3_cross(rightDir, front));
    if (cam->type == CRT_CAM_ORTHOGRAPHIC) {
        float aspect = fdiv(height, width);
        float ow = cam->orthoScale * (aspect > 1.0f ? aspect : 1.0f);
        float oh = cam->orthoScale * (aspect > 1.0f ? aspect : 1.0f);
        v3 o = v3_add(v3_scale(v3_scale(rightDir, x), ow), v3_scale(v3_scale(upDir, y), oh));
        o = v3_add(o, pos);
        return ray_make2(o, front);
    } else if (cam->type == CRT_CAM_PERSPECTIVE) {
        v3 o = v3_add(v3_scale(rightDir, x), v3_scale(upDir, y));
        o = v3_add(o, pos);
        float one_over_tan = fdiv(1.0f, tanf(cam->fov * 0.5f));
        v3 camPos = v3_sub(pos, v3_scale(front, width * 0.5f * one_over_tan));
        return ray_make2(o, v3_sub(o, camPos));
    }
    return ray_default();
}

/* aa_color(), main.cu:230-304 */
static v4 aa_color(const ctx_t* cx, float centerX, float centerY, v2 size, float width, float height,
                   const crt_settings* st, orc_counters* cnt) {
    static const float offs[4][2] = {{-1.f, -1.f}, {1.f, -1.f}, {-1.f, 1.f}, {1.f, 1.f}};
    unsigned aa_iter = st->aa_iter > 4 ? 4 : st->aa_iter;
    struct task { float cx, cy; v2 size; unsigned sample; } stack[256];
    size_t top = 0;
    stack[top].cx = centerX;
    stack[top].cy = centerY;
    stack[top].size = size;
    stack[top].sample = aa_iter;
    ++top;
    v4 acc = V4(0, 0, 0, 0);
    const crt_camera* cam = &cx->sc->camera;
    while (top > 0) {
        struct task t = stack[--top];
        v2 half = {t.size.x * 0.5f, t.size.y * 0.5f};
        float mult = fdiv(1.0f, (float)(1 << (2 * (aa_iter - t.sample))));
        if (t.sample == 0u) {
            ray_t r = camera_get_ray(cam, t.cx, t.cy, width, height);
            acc = v4_add(acc, v4_scale(color(cx, r.o, r.d, st->ref_iter, st->gl_iter, st->ind_rays, cnt), mult));
            continue;
        }
        if (t.sample == 1u) {
            for (int i = 0; i < 4; ++i) {
                ray_t r = camera_get_ray(cam, t.cx + offs[i][0] * half.x * 0.5f,
                                         t.cy + offs[i][1] * half.y * 0.5f, width, height);
                v4 c = color(cx, r.o, r.d, st->ref_iter, st->gl_iter, st->ind_rays, cnt);
                acc = v4_add(acc, v4_scale(v4_scale(c, mult), 0.25f));
            }
            continue;
        }
        v4 colors[4];
        int allEqual = 1;
        for (int i = 0; i < 4; ++i) {
            ray_t r = camera_get_ray(cam, t.cx + offs[i][0] * half.x, t.cy + offs[i][1] * half.y, width, height);
            colors[i] = color(cx, r.o, r.d, st->ref_iter, st->gl_iter, st->ind_rays, cnt);
            if (i != 0 && allEqual && !v4_equal(colors[i], colors[0])) allEqual = 0;
        }
        if (allEqual) {
            acc = v4_add(acc, v4_scale(colors[0], mult));
        } else {
            if (top + 4 > 256) {
                for (int i = 0; i < 4; ++i) acc = v4_add(acc, v4_scale(v4_scale(colors[i], mult), 0.25f));
                continue;
            }
            for (int i = 0; i < 4; ++i) {
                stack[top].cx = t.cx + offs[i][0] * half.x * 0.5f;
                stack[top].cy = t.cy + offs[i][1] * half.y * 0.5f;
                stack[top].size = half;
                stack[top].sample = t.sample - 1u;
                ++top;
            }
        }
    }
    return acc;
}

/* math_functions.hpp:31-45 */
static float remap_c(float input, float cs, float ce, float es, float ee) {
    float denom = ce - cs;
    if (denom == 0.0f) return input;
    return es + fdiv(ee - es, denom) * (input - cs);
}

/* body of render(), main.cu:328-347 (with get_world_coordinates, :306-314) */
void orc_render_pixel(const crt_scene_desc* sc, const crt_settings* st, const float* ltc1,
                      const float* ltc2, uint32_t i, uint32_t j, uint32_t rng[6], float out[4],
                      orc_counters* cnt) {
    ctx_t cx = {sc, ltc1, ltc2, rng};
    float height_world = 2.0f;
    float width_world = height_world * fdiv((float)st->nx, (float)st->ny);
    float x_world = remap_c((float)i, 0.0f, (float)st->nx, -width_world * 0.5f, width_world * 0.5f);
    float y_world = remap_c((float)j, 0.0f, (float)st->ny, 1.0f, -1.0f);
    v2 pixel = {fdiv(width_world, (float)st->nx), fdiv(height_world, (float)st->ny)};
    v4 c = aa_color(&cx, x_world, y_world, pixel, width_world, height_world, st, cnt);
    out[0] = saturatef_c(c.x);
    out[1] = saturatef_c(c.y);
    out[2] = saturatef_c(c.z);
    out[3] = saturatef_c(c.w);
}

int orc_render(const crt_scene_desc* sc, const crt_settings* st, uint64_t seed, const float* ltc_tables,
               float* fb, uint32_t* rng_inout6, int fresh_rng, uint32_t x0, uint32_t y0, uint32_t x1,
               uint32_t y1, int nthreads, orc_counters* total) {
    if (!sc || !st || !ltc_tables || !fb) return -1;
    if (x1 > st->nx) x1 = st->nx;
    if (y1 > st->ny) y1 = st->ny;
    const float* ltc1 = ltc_tables;
    const float* ltc2 = ltc_tables + 64 * 64 * 4;
    orc_counters sum = {0, 0, 0, 0};
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    uint64_t s_rays = 0, s_shade = 0, s_child = 0, s_draws = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : s_rays, s_shade, s_child, s_draws)
    for (uint32_t j = y0; j < y1; ++j) {
        for (uint32_t i = x0; i < x1; ++i) {
            size_t idx = (size_t)j * st->nx + i;
            uint32_t rng[6];
            if (fresh_rng || !rng_inout6) orc_rng_init(seed + (uint64_t)idx, rng);
            else memcpy(rng, rng_inout6 + idx * 6, sizeof rng);
            uint32_t d0 = rng[5];
            orc_counters c = {0, 0, 0, 0};
            orc_render_pixel(sc, st, ltc1, ltc2, i, j, rng, fb + idx * 4, &c);
            if (rng_inout6) memcpy(rng_inout6 + idx * 6, rng, sizeof rng);
            s_rays += c.rays;
            s_shade += c.shade_points;
            s_child += c.gi_children;
            s_draws += (uint32_t)((rng[5] - d0) * 945708813u); /* 362437^-1 mod 2^32 */
        }
    }
    sum.rays = s_rays;
    sum.shade_points = s_shade;
    sum.gi_children = s_child;
    sum.rng_draws = s_draws;
    if (total) *total = sum;
    return 0;
}

/* -------------------------------------------------------- function-level probes */
int orc_camera_ray(const crt_camera* cam, float x, float y, float w, float h, float o[3], float d[3]) {
    ray_t r = camera_get_ray(cam, x, y, w, h);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
    d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
    return 0;
}
int orc_world_hit(const crt_scene_desc* sc, const float o[3], const float d[3], float maxdist,
                  orc_hit_record* out) {
    ray_t r = maxdist > 0.0f ? ray_make3(v3_from(o), v3_from(d), maxdist) : ray_make2(v3_from(o), v3_from(d));
    hit_t h = hit_default();
    int any = world_hit(sc, &r, &h);
    out->hit = any;
    out->t = h.t;
    out->p[0] = h.p.x; out->p[1] = h.p.y; out->p[2] = h.p.z;
    out->n[0] = h.n.x; out->n[1] = h.n.y; out->n[2] = h.n.z;
    out->geom = any ? (int)(((const char*)h.mat - (const char*)sc->geometry) / sizeof(crt_geometry)) : -1;
    return 0;
}
int orc_shade(const crt_scene_desc* sc, const float* ltc_tables, const float p[3], const float n[3],
              const float view[3], uint32_t geom, uint32_t rng[6], float out[4]) {
    ctx_t cx = {sc, ltc_tables, ltc_tables + 64 * 64 * 4, rng};
    light_input_t in = {v3_from(p), v3_from(n), v3_from(view), &sc->geometry[geom].mat};
    v4 c = lights_calculate_color(&cx, &in);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    return 0;
}
int orc_color(const crt_scene_desc* sc, const crt_settings* st, const float* ltc_tables, const float o[3],
              const float d[3], uint32_t rng[6], float out[4]) {
    ctx_t cx = {sc, ltc_tables, ltc_tables + 64 * 64 * 4, rng};
    v4 c = color(&cx, v3_from(o), v3_from(d), st->ref_iter, st->gl_iter, st->ind_rays, NULL);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    return 0;
}

/* ---------------------------------------------------------------- Cornell scene */
/* quat<float> pieces used by vec3::rotate (vec.hpp:639-655, quat.hpp:60-88,111-167,258-263) */
typedef struct { float s; v3 v; } quat_t;
static float quat_mag(quat_t q) { return fdiv(1.0f, rsqrt_c(q.s * q.s + v3_dot(q.v, q.v))); }
static quat_t quat_mul(quat_t a, quat_t b) {
    quat_t r;
    float t = a.s;
    r.s = a.s * b.s - v3_dot(a.v, b.v);
    r.v = v3_add(v3_add(v3_scale(b.v, t), v3_scale(a.v, b.s)), v3_cross(a.v, b.v));
    return r;
}
static int v3_is_zero(v3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
static v3 v3_rotate(v3 p, v3 axis, float radians) {
    quat_t qp = {0.0f, p};
    v3 na = axis;
    if (!v3_is_zero(na)) na = v3_normalize(na);
    quat_t q;
    if (!v3_is_zero(na)) {
        q.s = cosf(radians * 0.5f);
        q.v = v3_scale(v3_normalize(na), sinf(radians * 0.5f));
    } else {
        q.s = cosf(radians * 0.5f);
        q.v = na;
    }
    float m = quat_mag(q);
    if (m != 0.0f) { /* quat::normalize -> operator/=(T), quat.hpp:124-128,278-289 */
        q.s /= m;
        q.v = v3_divs(q.v, m);
    }
    quat_t inv = q; /* quat::invert, quat.hpp:143-163 */
    float mm = quat_mag(inv);
    mm *= mm;
    mm = fdiv(1.0f, mm);
    inv.v = v3_scale(inv.v, -1.0f);
    inv.s *= mm;
    inv.v = v3_scale(inv.v, mm);
    return quat_mul(quat_mul(q, qp), inv).v;
}

static void set4(float* d, float a, float b, float c, float e) { d[0] = a; d[1] = b; d[2] = c; d[3] = e; }
static crt_material mk_mat(uint32_t type, const float col[4], float amb_scale, float shin, float ri) {
    crt_material m;
    memset(&m, 0, sizeof m);
    m.type = type;
    for (int i = 0; i < 4; ++i) {
        m.ambient[i] = col[i] * amb_scale; /* Color::x() * 0.1f scales alpha too */
        m.diffuse[i] = col[i];
        m.specular[i] = col[i];
    }
    m.shininess = shin;
    m.refractIndex = ri;
    return m;
}
static void mk_quad(crt_geometry* g, const float p[4][3], const float n[3], crt_material m) {
    static const float tex[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    memset(g, 0, sizeof *g);
    g->kind = CRT_GEOM_QUAD;
    g->mat = m;
    for (int k = 0; k < 4; ++k) {
        memcpy(g->v[k].pos, p[k], 12);
        memcpy(g->v[k].tex, tex[k], 8);
        memcpy(g->v[k].normal, n, 12);
    }
}

/* create_world(), main.cu:370-595 */
void orc_cornell_scene(uint32_t shadowSamples, crt_camera* cam, crt_geometry* g, crt_light* light) {
    const float white[4] = {1, 1, 1, 1}, red[4] = {1, 0, 0, 1}, blue[4] = {0, 0, 1, 1};
    memset(cam, 0, sizeof *cam);
    cam->front[2] = -1.0f;
    cam->type = CRT_CAM_PERSPECTIVE;
    cam->fov = fmaxf(45.0f * 0.01745329251994329576f, EPS); /* deg_to_rad<float,true>, setFov */
    cam->orthoScale = fmaxf(2.0f, EPS);
    const float qp[4][3] = {{-.25f, .98f, -1.25f}, {.25f, .98f, -1.25f}, {.25f, .98f, -.75f}, {-.25f, .98f, -.75f}};
    crt_material reflect = mk_mat(CRT_MAT_REFLECT, white, 0.1f, 0.0f, 0.0f);
    crt_material refractive = mk_mat(CRT_MAT_REFRACTIVE, white, 0.1f, 0.0f, 1.5f);
    crt_material emissive = mk_mat(CRT_MAT_DIFFUSE, white, 1.0f, 0.0f, 0.0f);
    crt_material mwhite = mk_mat(CRT_MAT_DIFFUSE, white, 0.1f, 5.0f, 0.0f);
    crt_material mred = mk_mat(CRT_MAT_DIFFUSE, red, 0.1f, 5.0f, 0.0f);
    crt_material mblue = mk_mat(CRT_MAT_DIFFUSE, blue, 0.1f, 5.0f, 0.0f);
    memset(&g[0], 0, sizeof g[0]);
    g[0].kind = CRT_GEOM_SPHERE; g[0].mat = reflect;
    g[0].center[0] = -0.25f; g[0].center[1] = -.72f; g[0].center[2] = -1.1f; g[0].radius = .275f;
    memset(&g[1], 0, sizeof g[1]);
    g[1].kind = CRT_GEOM_SPHERE; g[1].mat = refractive;
    g[1].center[0] = 0.3f; g[1].center[1] = -.72f; g[1].center[2] = -.6f; g[1].radius = .275f;
    const float back[4][3] = {{-1, -1, -2}, {1, -1, -2}, {1, 1, -2}, {-1, 1, -2}}, nb[3] = {0, 0, 1};
    const float top[4][3] = {{-1, 1, -2}, {1, 1, -2}, {1, 1, 0}, {-1, 1, 0}}, nt[3] = {0, -1, 0};
    const float bot[4][3] = {{-1, -1, -2}, {1, -1, -2}, {1, -1, 0}, {-1, -1, 0}}, nbo[3] = {0, 1, 0};
    const float rgt[4][3] = {{1, -1, -2}, {1, -1, 0}, {1, 1, 0}, {1, 1, -2}}, nr[3] = {-1, 0, 0};
    const float lft[4][3] = {{-1, -1, -2}, {-1, 1, -2}, {-1, 1, 0}, {-1, -1, 0}}, nl[3] = {1, 0, 0};
    float lq[4][3];
    for (int k = 0; k < 4; ++k) { lq[k][0] = qp[k][0]; lq[k][1] = qp[k][1] + 0.01f; lq[k][2] = qp[k][2]; }
    mk_quad(&g[2], back, nb, mwhite);
    mk_quad(&g[3], top, nt, mwhite);
    mk_quad(&g[4], bot, nbo, mwhite);
    mk_quad(&g[5], rgt, nr, mblue);
    mk_quad(&g[6], lft, nl, mred);
    mk_quad(&g[7], lq, nt, emissive);
    /* AreaLight ctor (CCW), AreaLight.hpp:534-556 + _defineCenterPoint :56-70 */
    memset(light, 0, sizeof *light);
    light->kind = CRT_LIGHT_AREA;
    set4(light->color, 1, 1, 1, 1);
    light->intensity = 10.0f;
    light->twoSided = 0;
    light->shadowSamples = shadowSamples;
    light->initialized = 1;
    v3 pts[4] = {v3_from(qp[0]), v3_from(qp[3]), v3_from(qp[2]), v3_from(qp[1])};
    v3 pos = V3(0, 0, 0);
    for (int k = 0; k < 4; ++k) { pos.x += pts[k].x; pos.y += pts[k].y; pos.z += pts[k].z; }
    pos.x *= 0.25f; pos.y *= 0.25f; pos.z *= 0.25f;
    for (int k = 0; k < 4; ++k) pts[k] = v3_sub(pts[k], pos);
    /* rotate(vec3(1,0,0), deg_to_rad(180)), main.cu:592, AreaLight.hpp:558-568 */
    v3 axis = V3(1.0f, 0.0f, 0.0f);
    if (!eps_equal(v3_length_sq(axis), 0.0f, EPS_SQ)) {
        v3 rot = v3_normalize(axis);
        float rad = 180.0f * 0.01745329251994329576f;
        for (int k = 0; k < 4; ++k) pts[k] = v3_rotate(pts[k], rot, rad);
    }
    light->pos[0] = pos.x; light->pos[1] = pos.y; light->pos[2] = pos.z;
    for (int k = 0; k < 4; ++k) { light->points[k][0] = pts[k].x; light->points[k][1] = pts[k].y; light->points[k][2] = pts[k].z; }
}
