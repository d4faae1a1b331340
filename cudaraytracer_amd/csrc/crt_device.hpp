// Device-side building blocks of the per-pixel render path for gfx950.
//
// One lane owns one pixel at a time and walks that pixel's sample tree in the
// reference's order, so the pixel's XORWOW stream is consumed draw for draw as
// the reference consumes it.  All arithmetic is IEEE binary32 evaluated in the
// reference's expression order without contraction (the build passes
// -ffp-contract=off; the only fused operations are the explicit ones inside
// rcp_exact, whose result is the correctly rounded quotient).  Per-primitive
// quantities that the reference recomputes per ray are read from the flattened
// scene instead (crt_scene.hpp) -- same operations on the same inputs, so the
// same bits.  Tests that reject a primitive without dividing are conservative:
// they fire only when the exact quotient is certain to fail the reference's
// own comparison, and fall through to the exact path otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crt_scene.hpp"

#define DI __device__ __forceinline__
// Constant address space: loads through it with a wave-uniform address become scalar loads.
#define CRT_CONST __attribute__((address_space(4)))

namespace crt {

// Measuring aid (build with -DCRT_LANE_PROFILE): PROF_LANES(k) counts, per section k, how often a wave runs the
// section and with how many active lanes; the trace launch prints the table (crt_wavefront.hip).
#ifdef CRT_LANE_PROFILE
static __device__ unsigned long long g_lane_prof[2 * 32];
#define PROF_LANES(k)                                                                             \
    do {                                                                                          \
        const unsigned long long m_ = __ballot(1);                                                \
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) {                               \
            atomicAdd(&g_lane_prof[2 * (k)], 1ull);                                               \
            atomicAdd(&g_lane_prof[2 * (k) + 1], (unsigned long long)__popcll(m_));              \
        }                                                                                         \
    } while (0)
#else
#define PROF_LANES(k)
#endif

typedef const DevScene CRT_CONST* ScenePtr;
typedef const float CRT_CONST* CFloatPtr;

constexpr float kEps = 1e-4f;   // MSTD_CUDA_EPSILON (math_types.hpp:78)
constexpr float kEpsSq = 1e-8f; // MSTD_CUDA_EPSILON_SQ (math_types.hpp:79)
constexpr float kFltMax = 3.402823466e+38f;

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

DI V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
DI V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
DI V3 ld3c(CFloatPtr p) { return V3{p[0], p[1], p[2]}; }
DI V4 ld4(const float* p) { return V4{p[0], p[1], p[2], p[3]}; }
DI V4 ld4c(CFloatPtr p) { return V4{p[0], p[1], p[2], p[3]}; }
DI V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DI V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DI V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DI V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
DI V3 neg(V3 a) { return a * -1.0f; } // vec::operator-() multiplies by -1 (vec.hpp:819-821)
DI V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DI V4 operator*(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
DI V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
DI bool equal4(V4 a, V4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
DI float comp(V3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Correctly rounded 1/x.  For |x| in [2^-100, 2^100] a hardware reciprocal
// estimate (1 ulp) is refined by two fused Newton steps, which yields the
// IEEE-rounded quotient for every such x (verified exhaustively over all
// 2^32 bit patterns by crt_selftest); anything else takes the full division.
DI float rcp_exact(float x) {
#if CRT_FAST_RCP
    float ax = __builtin_fabsf(x);
    if (ax >= 0x1p-100f && ax <= 0x1p100f) {
        float r = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        return r;
    }
#endif
    return 1.0f / x;
}

// Unchecked variants for operands the caller has already bounded to [2^-100, 2^100].
DI float rcp_nr(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
// Correctly rounded square root of a normal-range operand: hardware estimate (1 ulp), then pick
// among it and its two neighbours by the sign of the fused residuals (verified exhaustively by
// crt_selftest against sqrtf).
DI float sqrt_nr(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    float r_dn = __builtin_fmaf(-s_dn, s, x);
    float r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}

// vec::dot accumulates from zero in index order (vec.hpp:428-434).
DI float dot3(V3 a, V3 b) {
    float r = 0.0f + a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
DI float len_sq(V3 a) { return dot3(a, a); }
DI float rsqrt_c(float x) { return rcp_exact(__builtin_sqrtf(x)); } // rsqrtf in IEEE form
// vec::length (vec.hpp:398-409) = 1 / rsqrt(len^2) and vec::normalize (vec.hpp:411-422) = v * (1/len): the
// reference takes a square root and then three dependent reciprocals, s -> r1 = 1/s -> len = 1/r1 -> 1/len.
// Under round-to-nearest the third one returns the first: 1/(1/(1/s)) == 1/s for every binary32 s whose chain
// stays in the normal range (all 2^24 significands checked on the host, all bit patterns by crt_selftest_rcp on
// the device; 1/(1/s) == s fails for a sixth of them, so `len` keeps its two reciprocals).  One range test on
// len^2 covers the whole chain (len^2 in [2^-80, 2^80] keeps every intermediate in [2^-40, 2^40]).
__device__ __attribute__((noinline)) float length_slow(float lsq) { return 1.0f / (1.0f / __builtin_sqrtf(lsq)); }
DI float length_from_sq(float lsq) {
    if (lsq >= 0x1p-80f && lsq <= 0x1p80f) return rcp_nr(rcp_nr(sqrt_nr(lsq)));
    return length_slow(lsq);
}
DI float length3(V3 a) { return length_from_sq(len_sq(a)); }
// vec::operator/=(T) (vec.hpp:739-769)
DI V3 div_s(V3 a, float y) {
    if (y == 0.0f) return a;
    float inv = rcp_exact(y);
    return a * inv;
}
// length and normalised vector together; `len` is what length3(a) returns.
DI V3 normalize3_len(V3 a, float& len) {
    float lsq = len_sq(a);
    if (lsq >= 0x1p-80f && lsq <= 0x1p80f) {
        const float r1 = rcp_nr(sqrt_nr(lsq));
        len = rcp_nr(r1);
        return a * r1; // r1 == 1/len, see above
    }
    len = length_slow(lsq);
    if (len == 0.0f) return a;
    return a * (1.0f / len);
}
DI V3 normalize3(V3 a) {
    float len;
    return normalize3_len(a, len);
}
DI V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DI bool eps_equal(float a, float b, float e) { return __builtin_fabsf(a - b) < e; } // math_functions.hpp:75-78
DI float min_std(float a, float b) { return (b < a) ? b : a; }                      // cuda::std::min
DI float max_std(float a, float b) { return (a < b) ? b : a; }                      // cuda::std::max
DI float saturate_c(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
// vec::reflect (vec.hpp:460-468)
DI V3 reflect3(V3 i, V3 n) {
    float k = 2.0f * dot3(i, n);
    return i - n * k;
}
// vec::refracted, device branch (vec.hpp:475-491)
DI V3 refract3(V3 d, V3 n, float eta) {
    float cos_theta = min_std(dot3(neg(d), n), 1.0f);
    V3 perp = (d + n * cos_theta) * eta;
    float lsq = len_sq(perp);
    V3 par = div_s(neg(n), rsqrt_c(__builtin_fabsf(1.0f - lsq)));
    return perp + par;
}
// Ray(origin, dir) keeps dir as given when it is shorter than 1e-4, else normalises (Ray.hpp:37-43).
DI V3 ray_dir(V3 d) { return eps_equal(len_sq(d), 0.0f, kEpsSq) ? d : normalize3(d); }

// ---------------------------------------------------------------- XORWOW (cuRAND)
// State words {v0..v4, d}; rng_uniform() is curand_uniform()
// (main.cu:177-178,324; AreaLight.hpp:342-343).
struct Rng {
    uint32_t v0, v1, v2, v3, v4, d;
};
DI void rng_seed(Rng& s, uint64_t seed) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v0 = 123456789u + t0;
    s.v1 = 362436069u ^ t0;
    s.v2 = 521288629u + t1;
    s.v3 = 88675123u ^ t1;
    s.v4 = 5783321u + t0;
}
DI void rng_step(Rng& s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
}
// x * 2^-32 is exact, so the multiply-add below rounds once, fused or not.
DI float rng_to_uniform(uint32_t x) { return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }
DI float rng_uniform(Rng& s) {
    rng_step(s);
    return rng_to_uniform(s.v4 + s.d);
}
// Value the generator returned when it arrived in state s.
DI float rng_current_uniform(const Rng& s) { return rng_to_uniform(s.v4 + s.d); }
// Exact inverse of rng_step (the xorshift is a bijection on the five words).
DI void rng_unstep(Rng& s) {
    uint32_t y = s.v4 ^ (s.v3 ^ (s.v3 << 4)); // = t ^ (t << 1)
    uint32_t t = y;
    t ^= t << 1;
    t ^= t << 2;
    t ^= t << 4;
    t ^= t << 8;
    t ^= t << 16;
    uint32_t u = t; // t = u ^ (u >> 2)
    u ^= u >> 2;
    u ^= u >> 4;
    u ^= u >> 8;
    u ^= u >> 16;
    s.v4 = s.v3;
    s.v3 = s.v2;
    s.v2 = s.v1;
    s.v1 = s.v0;
    s.v0 = u;
    s.d -= 362437u;
}

// n steps at once.  The five xorshift words evolve linearly over GF(2), so the state after n steps is
// the XOR of per-nibble table entries: entry e = (word * 8 + nibble) * 16 + value holds the image of that
// nibble value, words 0..3 at tab[4e .. 4e+3] (one 16-byte read) and word 4 at tab[4 * 640 + e]; the
// table is built on the host by stepping basis states.  The Weyl word just advances by n * 362437.
constexpr uint32_t kJumpEntries = 640, kJumpWords = 5 * kJumpEntries;
DI void rng_jump(Rng& s, const uint32_t* tab, uint32_t weyl_step) {
    const uint32_t in[5] = {s.v0, s.v1, s.v2, s.v3, s.v4};
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
    const uint4* tab4 = reinterpret_cast<const uint4*>(tab);
#pragma unroll
    for (int wd = 0; wd < 5; ++wd) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const uint32_t idx = (uint32_t)(wd * 8 + nb) * 16u + ((in[wd] >> (4 * nb)) & 15u);
            const uint4 q = tab4[idx];
            o0 ^= q.x;
            o1 ^= q.y;
            o2 ^= q.z;
            o3 ^= q.w;
            o4 ^= tab[4u * kJumpEntries + idx];
        }
    }
    s.v0 = o0;
    s.v1 = o1;
    s.v2 = o2;
    s.v3 = o3;
    s.v4 = o4;
    s.d += weyl_step;
}

// ------------------------------------------------------------------ sin / cos
// Stand-in for __sinf/__cosf on [0, 2*pi] (main.cu:64-65): two-term Cody-Waite
// reduction by pi/2 and fixed polynomial kernels, mul/add only, so the host
// checker can reproduce it bit for bit.  |error| < 2e-7.
DI void sincos_c(float a, float& s, float& c) {
    const float TWO_OVER_PI = 0.63661977236758134308f;
    const float PIO2_1 = 1.5707855225e+00f;
    const float PIO2_1T = 1.0804334124e-05f;
    float kf = __builtin_floorf(a * TWO_OVER_PI + 0.5f);
    int k = (int)kf;
    float r = a - kf * PIO2_1;
    r = r - kf * PIO2_1T;
    float z = r * r;
    const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f, S4 = 2.7557314297e-06f;
    const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f;
    float ps = S3 + z * S4;
    ps = S2 + z * ps;
    ps = S1 + z * ps;
    float sr = r + (r * z) * ps;
    float pc = C2 + z * C3;
    pc = C1 + z * pc;
    float cr = (1.0f - 0.5f * z) + (z * z) * pc;
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
}

// ------------------------------------------------------------------------ pow
// Stand-in for __powf(x, y) with x >= 0 in the point light's Blinn-Phong term (Light.hpp:173):
// exp2(y * log2(x)) out of IEEE operations in a fixed order (log2 from the atanh series of (m-1)/(m+1) on
// m in (sqrt(1/2), sqrt(2)], exp2 from the degree-7 Taylor polynomial on [-1/2, 1/2]), so that the host
// checker reproduces it bit for bit.  Relative error about 1e-7 * (1 + |y log2 x|); results below 2^-126
// are flushed to zero.
DI float pow_c(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (!(x > 0.0f)) return y > 0.0f ? 0.0f : __builtin_inff();
    if (x > 3.0e38f) return y > 0.0f ? __builtin_inff() : 0.0f;
    int e = 0;
    if (x < 1.17549435e-38f) { // subnormal base
        x = x * 16777216.0f;
        e = -24;
    }
    uint32_t bits = __float_as_uint(x);
    e += (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
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
    if (!(t < 128.0f)) return __builtin_inff();
    if (t < -126.0f) return 0.0f;
    const float nf = __builtin_floorf(t + 0.5f);
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
    return (w * __uint_as_float((uint32_t)(n1 + 127) << 23)) * __uint_as_float((uint32_t)(n2 + 127) << 23);
}

// ------------------------------------------------------------ intersection tests
// Sphere::hit (Sphere.hpp:40-98): returns the accepted root in t.  dd = d.d.
DI bool sphere_t(V3 center, float radius_sq, V3 o, V3 d, float dd, float dist, float& t) {
    V3 L = o - center;
    float a = dd;
    float b = 2.0f * dot3(d, L);
    float c = dot3(L, L) - radius_sq;
    float discr = b * b - 4.0f * a * c;
    if (discr < 0.0f) return false;
    PROF_LANES(15);
    float t0, t1;
    if (eps_equal(discr, 0.0f, kEps)) {
        t0 = t1 = -0.5f * (b / a);
    } else {
        float sq = rcp_exact(rsqrt_c(discr));
        float q = (b > 0.0f) ? -0.5f * (b + sq) : -0.5f * (b - sq);
        t0 = q / a;
        t1 = c / q;
    }
    if (t0 > t1) {
        float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t0 < 0.0f) {
        t0 = t1;
        if (t0 < 0.0f) return false;
    }
    if (dist > 0.0f && t0 > dist) return false;
    t = t0;
    return true;
}

// Plane part of the quad test: t = -(p.nor / d.nor), rejected when t < 0 or
// (dist > 0 and t > dist) (Quad.hpp:222-226).  `closest` is the caller's
// best-so-far distance: a candidate with t > closest is discarded by
// GeometryList::hit (GeometryList.hpp:28) whatever its (u,v), so it may be
// dropped here as well.  The three quick tests decide from signs and a
// multiplication when the quotient is certainly out of range (margin 1e-6
// against the 6e-8 rounding of the quotient); everything else, including any
// NaN or zero operand, goes through the exact division.
DI bool quad_plane_t(V3 p, V3 d, V3 nor, float dist, float closest, float& t) {
    float pn = dot3(p, nor), dn = dot3(d, nor);
    float apn = __builtin_fabsf(pn), adn = __builtin_fabsf(dn);
    bool same_sign = (pn > 0.0f && dn > 0.0f) || (pn < 0.0f && dn < 0.0f);
    if (same_sign && apn > 1e-30f && adn < 1e6f) return false; // quotient > 1e-36 > 0  =>  t < 0
    float lim = (dist > 0.0f && dist < closest) ? dist : closest;
    float m = (adn * lim) * 1.000001f;
    if (!same_sign && apn > m && m > 1e-30f && lim < 1e30f) return false; // t = |pn/dn| > lim
    PROF_LANES(14);
    float tt = -(pn / dn);
    if (tt < 0.0f || (dist > 0.0f && tt > dist)) return false;
    t = tt;
    return true;
}

// Conservative filters used by the shadow-ray loop: they return false only when
// the reference's test of that primitive certainly fails for this ray, using the
// same operands as the exact test but no division or square root.
//   sphere (Sphere.hpp:48-96): discr < 0 is the reference's own early-out; with
//   b > 0 and c > 0 both roots q/a and c/q are negative (q = -0.5(b + sqrt(discr)) < 0),
//   so the hit is rejected at Sphere.hpp:93-96.  The magnitude bounds keep the
//   quotients away from underflow, where a rounded -0 would not compare < 0.
DI bool sphere_may_hit(V3 center, float radius_sq, V3 o, V3 d, float dd4, bool dd_ok) {
    V3 L = o - center;
    float b = 2.0f * (d.x * L.x + d.y * L.y + d.z * L.z); // a zero's sign cannot matter below
    float c = (L.x * L.x + L.y * L.y + L.z * L.z) - radius_sq;
    float discr = b * b - dd4 * c; // dd4 = 4*(d.d), hoisted by the caller
    bool behind = dd_ok && b > 1e-30f && b < 1e6f && c > 1e-30f && c < 1e6f;
    return !(discr < 0.0f || behind);
}
//   quad (Quad.hpp:222-226): t = -(pn/dn) is certainly negative when pn*dn > 0, and |t| certainly
//   exceeds dist when |pn| > |dn|*dist*(1+1e-6) (either it is > dist or it is negative).  The
//   magnitude bounds keep the rounded quotient away from underflow.  dist_ok = dist in (1e-30, 1e30).
DI bool plane_may_hit(float pn, float dn, float dist, bool dist_ok) {
    float adn = __builtin_fabsf(dn);
    bool neg_t = pn * dn > 1e-24f && adn < 1e6f;
    float m = (adn * dist) * 1.000001f;
    bool far_t = dist_ok && __builtin_fabsf(pn) > m && m > 1e-30f;
    return !(neg_t || far_t);
}
DI bool quad_may_hit(V3 p, V3 d, V3 nor, float dist, bool dist_ok) {
    float pn = p.x * nor.x + p.y * nor.y + p.z * nor.z; // operands as in the exact test; a zero's sign cannot matter
    float dn = d.x * nor.x + d.y * nor.y + d.z * nor.z;
    return plane_may_hit(pn, dn, dist, dist_ok);
}

// Three-way classification of a sphere for a shadow ray: 0 = the reference's test certainly fails,
// 2 = it certainly succeeds, 1 = undecided (run the exact test).  "Certainly succeeds" (Sphere.hpp:48-96):
// origin outside (c > 0), approaching (b < 0), discriminant well clear of both the negative early-out and
// the |discr| < 1e-4 double-root branch; then q = -0.5(b - sqrt(discr)) >= |b|/2 > 0, both roots q/a and
// c/q are positive and the smaller is at most c/q <= 2c/|b|, which the last condition keeps below `dist`
// with a margin of 1e-3 against rounding of a few 1e-7.
DI uint32_t sphere_class(V3 center, float radius_sq, V3 o, V3 d, float dd4, bool dd_ok, float dist) {
    V3 L = o - center;
    float b = 2.0f * (d.x * L.x + d.y * L.y + d.z * L.z);
    float c = (L.x * L.x + L.y * L.y + L.z * L.z) - radius_sq;
    float bb = b * b, ac4 = dd4 * c;
    float discr = bb - ac4;
    bool behind = dd_ok && b > 1e-30f && b < 1e6f && c > 1e-30f && c < 1e6f;
    if (discr < 0.0f || behind) return 0u;
    bool sure = dd_ok && c > 1e-4f && c < 1e6f && b < -1e-4f && b > -1e6f &&
                discr > 2e-4f + 1e-4f * (bb + __builtin_fabsf(ac4)) && dist < 1e6f &&
                (2.0f * c) < ((-b) * dist) * 0.999f;
    return sure ? 2u : 1u;
}

// Conservative use of a quad's bounding box (DevGeomBox): returns false only when the point where the
// ray meets the plane, o - (pn/dn) d, is certainly outside the box on some axis.  The comparison is
// multiplied through by |dn|, so nothing is divided; lo/hi already contain the safety margin.
DI bool box_may_hit(V3 lo, V3 hi, float min_adn, V3 o, V3 d, float pn, float dn) {
    const float a = __builtin_fabsf(dn);
    const float sg = dn < 0.0f ? -1.0f : 1.0f;
    const float qx = (o.x * dn - pn * d.x) * sg, qy = (o.y * dn - pn * d.y) * sg, qz = (o.z * dn - pn * d.z) * sg;
    const bool outside = qx < lo.x * a || qx > hi.x * a || qy < lo.y * a || qy > hi.y * a || qz < lo.z * a || qz > hi.z * a;
    return !(a > min_adn && outside);
}

// Bilinear part of the quad test (Quad.hpp:228-295; AreaLight.hpp:406-480 with
// LIGHT, whose only difference is 0.5f * (1/k2), AreaLight.hpp:456).
// p = ray origin relative to corner 0, tt = accepted plane distance.
template <bool LIGHT, class UVT>
DI bool quad_uv(const UVT& g, V3 p, V3 d, float tt, float& u, float& v) {
    V3 pos = p + d * tt;
    float kpx = comp(pos, g.idu), kpy = comp(pos, g.idv);
    float kax = g.ka[0], kay = g.ka[1], kbx = g.kb[0], kby = g.kb[1], kgx = g.kg[0], kgy = g.kg[1];
    float k0 = kpx * kby - kpy * kbx;
    float k2 = g.k2;
    float k1 = (kpx * kgy - kpy * kgx) - g.nor_id;
    float uu, vv;
    if (g.linear) {
        vv = -(k0 / k1);
        uu = (kpx * kay - kpy * kax) / k1;
    } else {
        float w = k1 * k1 - 4.0f * k0 * k2;
        if (w < 0.0f) return false;
        w = rcp_exact(rsqrt_c(w));
        float ik2 = LIGHT ? 0.5f * rcp_exact(k2) : rcp_exact(2.0f * k2);
        vv = (-k1 - w) * ik2;
        if (vv < 0.0f || vv > 1.0f) vv = (-k1 + w) * ik2;
        uu = (kpx - kax * vv) / (kbx + kgx * vv);
    }
    if (min_std(uu, vv) < 0.0f || max_std(uu, vv) > 1.0f) return false;
    u = uu;
    v = vv;
    return true;
}

// Register copy of a DevGeomUV read through the constant address space.
struct UVRec {
    uint32_t idu, idv, linear;
    float nor_id, ka[2], kb[2], kg[2], k2;
};
DI UVRec load_uv(const DevGeomUV CRT_CONST* q) {
    UVRec r;
    r.idu = q->idu;
    r.idv = q->idv;
    r.linear = q->linear;
    r.nor_id = q->nor_id;
    r.ka[0] = q->ka[0];
    r.ka[1] = q->ka[1];
    r.kb[0] = q->kb[0];
    r.kb[1] = q->kb[1];
    r.kg[0] = q->kg[0];
    r.kg[1] = q->kg[1];
    r.k2 = q->k2;
    return r;
}

struct Hit {
    int idx; // -1: nothing hit
    uint32_t mat_type;
    float t, u, v;
};

// reflectance<true>(cosine, ri) > 1.0f (math_functions.hpp:106-133; main.cu:97, Light.hpp:61).  (1 - cosine)^5
// is evaluated as ((x*x)*(x*x))*x.  With cosine in [0,1] the test cannot hold, but it is not dead: at a
// grazing hit the direction, normalised once more (main.cu:91), can change the sign of d.n, cosine comes
// out as -6e-8 and the ray is mirrored instead of refracted.
// Early out: for cosine >= 0 and ri >= 0 every factor lies in [0,1] and r0 + fl(1 - r0) rounds to at most 1
// (1 - r0 is exact for r0 >= 1/2 and off by at most 2^-25 otherwise), so the comparison is false.
DI bool reflectance_over_one(float cosine, float ri) {
    if (cosine >= 0.0f && ri >= 0.0f) return false;
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    const float x = 1.0f - cosine, x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x) > 1.0f;
}

// GeometryList::hit (GeometryList.hpp:22-35): closest hit, ties and unordered
// distances go to the later primitive.  The loop index is wave-uniform, so each
// primitive costs one 8-dword scalar load.
DI Hit closest_hit(ScenePtr sc, V3 o, V3 d, float dist) {
    Hit h;
    h.idx = -1;
    h.mat_type = 0;
    h.t = kFltMax;
    h.u = 0.0f;
    h.v = 0.0f;
    const float dd = dot3(d, d);
    const uint32_t n = sc->n_geom;
    for (uint32_t i = 0; i < n; ++i) {
        const DevGeomHot CRT_CONST* g = &sc->hot[i];
        const uint32_t kind = g->kind, mat = g->mat_type;
        const V3 go = ld3c(g->o), gn = ld3c(g->n);
        float t = 0.0f, u = 0.0f, v = 0.0f;
        bool hit;
        if (kind == CRT_GEOM_SPHERE) {
            hit = sphere_t(go, gn.x, o, d, dd, dist, t);
        } else {
            V3 p = o - go;
            hit = quad_plane_t(p, d, gn, dist, h.t, t);
            if (hit) {
                PROF_LANES(5);
                UVRec q = load_uv(&sc->uv[i]);
                hit = quad_uv<false>(q, p, d, t, u, v);
#ifdef CRT_DOUBLE_UV
                float u2 = 0.0f, v2 = 0.0f;
                if (quad_uv<false>(q, p, d, t * 1.0000001f, u2, v2) && u2 == 12345.0f) hit = false;
#endif
            }
        }
        if (hit && !(t > h.t)) {
            h.idx = (int)i;
            h.mat_type = mat;
            h.t = t;
            h.u = u;
            h.v = v;
        }
    }
    return h;
}

// closest_hit for the tree walk: a plane hit of a parallelogram is first classified by where the hit point lies
// against the quad's extent on its two kept axes (DevGeomBox::half_out_t / half_in_t): clearly outside is the
// reference's rejection (Quad.hpp:281), clearly inside an axis-aligned rectangle is its acceptance, and only the
// band of 6e-5 of the scene size between the two runs the exact (u,v) solve inside the loop.  A lane's plane
// hits are mostly rejections (the planes of the other walls), and each exact solve costs two divisions.  For an
// accepted hit that was only classified, `uv_pending` is set and the caller computes (u,v) for the final hit
// alone (they are needed for the shading normal, Quad.hpp:299-325).
DI Hit closest_hit_classified(ScenePtr sc, V3 o, V3 d, float dist, bool& uv_pending) {
    Hit h;
    h.idx = -1;
    h.mat_type = 0;
    h.t = kFltMax;
    h.u = 0.0f;
    h.v = 0.0f;
    bool pending = false;
    const float dd = dot3(d, d);
    const uint32_t n = sc->n_geom;
    for (uint32_t i = 0; i < n; ++i) {
        const DevGeomWalk CRT_CONST* g = &sc->walk[i]; // one 16-dword scalar load, nothing more inside the branches
        const uint32_t kind = g->kind_mat & 0xffu, mat = g->kind_mat >> 8;
        const V3 go = ld3c(g->o), gn = ld3c(g->n);
        const V3 mid = ld3c(g->mid), half_out = ld3c(g->half_out_t), half_in = ld3c(g->half_in_t);
        float t = 0.0f, u = 0.0f, v = 0.0f;
        bool hit, deferred = false;
        if (kind == CRT_GEOM_SPHERE) {
            hit = sphere_t(go, gn.x, o, d, dd, dist, t);
        } else {
            V3 p = o - go;
            hit = quad_plane_t(p, d, gn, dist, h.t, t);
            if (hit) {
                // |hit point - centre| per axis (fused: this is a classification with a margin, not the reference's arithmetic)
                const float ax = __builtin_fabsf(__builtin_fmaf(d.x, t, o.x) - mid.x);
                const float ay = __builtin_fabsf(__builtin_fmaf(d.y, t, o.y) - mid.y);
                const float az = __builtin_fabsf(__builtin_fmaf(d.z, t, o.z) - mid.z);
                const bool outside = (ax > half_out.x) | (ay > half_out.y) | (az > half_out.z);
                const bool inside = (ax < half_in.x) & (ay < half_in.y) & (az < half_in.z);
                if (outside) {
                    hit = false;
                } else if (inside) {
                    deferred = true;
                } else {
                    PROF_LANES(5);
                    UVRec q = load_uv(&sc->uv[i]);
                    hit = quad_uv<false>(q, p, d, t, u, v);
                }
            }
        }
        if (hit && !(t > h.t)) {
            h.idx = (int)i;
            h.mat_type = mat;
            h.t = t;
            h.u = u;
            h.v = v;
            pending = deferred;
        }
    }
    uv_pending = pending;
    return h;
}

// closest_hit restricted to the primitives in `mask` (per lane); primitives outside it are known not to be
// hit by this ray, so the result is that of closest_hit.
DI Hit closest_hit_masked(ScenePtr sc, V3 o, V3 d, float dist, uint64_t mask) {
    Hit h;
    h.idx = -1;
    h.mat_type = 0;
    h.t = kFltMax;
    h.u = 0.0f;
    h.v = 0.0f;
    const float dd = dot3(d, d);
    const uint32_t n = sc->n_geom;
    for (uint32_t i = 0; i < n; ++i) {
        const bool mine = (mask >> i) & 1ull;
        if (!__any(mine)) continue;
        const DevGeomHot CRT_CONST* g = &sc->hot[i];
        const uint32_t kind = g->kind, mat = g->mat_type;
        const V3 go = ld3c(g->o), gn = ld3c(g->n);
        float t = 0.0f, u = 0.0f, v = 0.0f;
        bool hit = false;
        if (mine) {
            PROF_LANES(16);
            if (kind == CRT_GEOM_SPHERE) {
                hit = sphere_t(go, gn.x, o, d, dd, dist, t);
            } else {
                V3 p = o - go;
                hit = quad_plane_t(p, d, gn, dist, h.t, t);
                if (hit) {
                    PROF_LANES(17);
                    UVRec q = load_uv(&sc->uv[i]);
                    hit = quad_uv<false>(q, p, d, t, u, v);
                }
            }
        }
        if (hit && !(t > h.t)) {
            h.idx = (int)i;
            h.mat_type = mat;
            h.t = t;
            h.u = u;
            h.v = v;
        }
    }
    return h;
}

// Hit point and shading normal of an accepted hit (Sphere.hpp:102-107, Quad.hpp:299-325).
DI void hit_point_normal(ScenePtr sc, const Hit& h, V3 o, V3 d, V3& P, V3& N) {
    P = o + d * h.t;
    if (sc->hot[h.idx].kind == CRT_GEOM_SPHERE) {
        N = normalize3(P - ld3c(sc->hot[h.idx].o));
    } else {
        const DevGeomShade CRT_CONST* g = &sc->shade[h.idx];
        float u = h.u, v = h.v;
        float l0 = (1 - u) * (1 - v);
        float l1 = u * (1 - v);
        float l2 = u * v;
        float l3 = (1 - u) * v;
        N = V3{l0 * g->n[0][0] + l1 * g->n[1][0] + l2 * g->n[2][0] + l3 * g->n[3][0],
               l0 * g->n[0][1] + l1 * g->n[1][1] + l2 * g->n[2][1] + l3 * g->n[3][1],
               l0 * g->n[0][2] + l1 * g->n[1][2] + l2 * g->n[2][2] + l3 * g->n[3][2]};
    }
}

// Glass on the way to the light: the part of Light::_isCovered (Light.hpp:47-80)
// entered when the closest hit is Refractive.  Returns "covered".
DI bool covered_through_glass(ScenePtr sc, const DevLight CRT_CONST* L, V3 o, V3 d, Hit h) {
    // The do/while of the reference runs at most `iterations` = 10 passes; pass 0 found h.
    for (unsigned i = 0;;) {
        PROF_LANES(18);
        V3 P, N;
        hit_point_normal(sc, h, o, d, P, N);
        bool front_face = dot3(d, N) < 0.0f;
        V3 norm = front_face ? N : neg(N);
        float ratio = sc->shade[h.idx].refract_index / 1.0f;
        if (front_face) ratio = rcp_exact(ratio);
        V3 rd = normalize3(d);
        float cos_theta = __builtin_fminf(dot3(neg(rd), norm), 1.0f);
        float sin_theta = rcp_exact(rsqrt_c(1.0f - cos_theta * cos_theta));
        if (ratio * sin_theta > 1.0f || reflectance_over_one(cos_theta, ratio)) return true; // Light.hpp:61
        V3 dir = refract3(rd, norm, ratio);
        V3 ro = P + dir * 0.01f;
        V3 rdir = ray_dir(dir);
        if (!front_face) {
            if (L->kind == CRT_LIGHT_AREA) { // AreaLight::_hittedLight (AreaLight.hpp:378-481)
                float t, u, v;
                V3 p = (ro - ld3c(L->p0)) - ld3c(L->pos);
                if (!quad_plane_t(p, rdir, ld3c(L->nor), 0.0f, kFltMax, t)) return true;
                UVRec q = load_uv(&L->uv);
                return !quad_uv<true>(q, p, rdir, t, u, v);
            } else { // Ray::isPointOnRay (Ray.hpp:80-105) with distance 0
                V3 w = ld3c(L->pos) - ro;
                V3 cr = cross3(rdir, w);
                if (len_sq(cr) >= kEpsSq) return true;
                return !(dot3(w, rdir) >= 0.0f);
            }
        }
        o = ro;
        d = rdir;
        ++i;
        if (!(i < 10u)) return true; // loop ends with hit.isHit still set
        // Inside a glass sphere the ray leaves the ball within 2r: only primitives that reach into the
        // ball can be hit first (DevScene::interior_*).  Robustly inside: |o - C|^2 < r^2 (1 - 1e-5).
        uint64_t candidates = ~0ull;
        {
            const DevGeomHot* g = (const DevGeomHot*)&sc->hot[h.idx];
            if (g->kind == CRT_GEOM_SPHERE) {
                const V3 L = o - V3{g->o[0], g->o[1], g->o[2]};
                const float r2 = g->n[0];
                if (len_sq(L) < r2 - 1e-5f * r2)
                    candidates = ((uint64_t)((const uint32_t*)sc->interior_hi)[h.idx] << 32) | ((const uint32_t*)sc->interior_lo)[h.idx];
            }
        }
        h = closest_hit_masked(sc, o, d, 0.0f, candidates);
        if (h.idx < 0) return false;
        if (h.mat_type != CRT_MAT_REFRACTIVE) return true;
    }
}

// ------------------------------------------------------------------ LTC shading
// Texture::sample row coordinate with CLAMP wrap (Texture.hpp:154-182); 64 = out of range.
DI uint32_t ltc_texel(float f) {
    if (f < 0.0f || f > 1.0f) f = saturate_c(f);
    uint32_t i = (uint32_t)(f * 63.0f);
    return i < 64u ? i : 64u;
}

// AreaLight::_integrateEdgeVec (AreaLight.hpp:72-98)
DI V3 integrate_edge_vec(V3 v1, V3 v2) {
    float x = dot3(v1, v2);
    float y = __builtin_fabsf(x);
    float a = 0.8543985f + (0.4965155f + 0.0145206f * y) * y;
    float b = 3.4175940f + (4.1616724f + y) * y;
    float v = a / b;
    float ts = (x > 0.0f) ? v : 0.5f * rsqrt_c(max_std(1.0f - x * x, kEps)) - v;
    return cross3(v1, v2) * ts;
}

// Tail of AreaLight::_evaluateLTC's clipless branch (AreaLight.hpp:254-297) on the four transformed corners.
DI float ltc_polygon(const float* __restrict__ ltc2, V3 L0, V3 L1, V3 L2, V3 L3, bool behind, bool two_sided) {
    L0 = ray_dir(L0); // same "normalise unless shorter than 1e-4" rule as Ray
    L1 = ray_dir(L1);
    L2 = ray_dir(L2);
    L3 = ray_dir(L3);
    V3 vsum = mk3(0.0f, 0.0f, 0.0f);
    vsum = vsum + integrate_edge_vec(L0, L1);
    vsum = vsum + integrate_edge_vec(L1, L2);
    vsum = vsum + integrate_edge_vec(L2, L3);
    vsum = vsum + integrate_edge_vec(L3, L0);
    float len = length3(vsum);
    float z = !eps_equal(len, 0.0f, kEps) ? vsum.z / len : 0.0f;
    if (behind) z = -z;
    const float LUT_SCALE = 63.0f / 64.0f, LUT_BIAS = 0.5f / 64.0f; // AreaLight.hpp:47-49
    float ux = (z * 0.5f + 0.5f) * LUT_SCALE + LUT_BIAS;
    float uy = len * LUT_SCALE + LUT_BIAS;
    uint32_t tx = ltc_texel(ux), ty = ltc_texel(uy);
    float scale = (tx < 64u && ty < 64u) ? ltc2[(ty * 64u + tx) * 4u + 3u] : 0.0f;
    float sum = len * scale;
    if (!behind && !two_sided) sum = 0.0f;
    return sum;
}

// Out-of-line copy for the few shaded points the paired form below hands back.
__device__ __attribute__((noinline)) float ltc_polygon_rare(const float* __restrict__ ltc2, V3 L0, V3 L1, V3 L2, V3 L3, bool behind,
                                                            bool two_sided) {
    return ltc_polygon(ltc2, L0, L1, L2, L3, behind, two_sided);
}

// ---- the same tail for the two calls of AreaLight::calculateColor at once (AreaLight.hpp:638,642).
// _evaluateLTC runs twice per shaded point on different matrices and the same instruction sequence, so the
// two evaluations ride in the halves of packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32
// issue at the rate of their scalar forms): component 0 = the identity call, component 1 = the Minv call.
// Every operation rounds as in the scalar code above; what cannot be packed (reciprocal and square-root
// estimates, compares, selects) is done per component.
typedef float F2 __attribute__((ext_vector_type(2)));
typedef uint32_t U2 __attribute__((ext_vector_type(2)));
struct P3 { // a pair of vectors, component-wise
    F2 x, y, z;
};
DI F2 splat2(float a) { return F2{a, a}; }
DI F2 fma2(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }
DI P3 operator+(P3 a, P3 b) { return P3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DI P3 operator*(P3 a, F2 s) { return P3{a.x * s, a.y * s, a.z * s}; }
DI F2 dot3(P3 a, P3 b) {
    F2 r = splat2(0.0f) + a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
DI F2 dot3(P3 a, V3 b) {
    F2 r = splat2(0.0f) + a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
DI P3 cross3(P3 a, P3 b) { return P3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DI V3 half3(P3 a, int k) { return V3{a.x[k], a.y[k], a.z[k]}; }
DI F2 rcp_nr(F2 x) { // rcp_nr of both components
    F2 r = F2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    F2 e = fma2(-x, r, splat2(1.0f));
    r = fma2(e, r, r);
    e = fma2(-x, r, splat2(1.0f));
    r = fma2(e, r, r);
    return r;
}
DI F2 sqrt_nr(F2 x) { // sqrt_nr of both components
    F2 s = F2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const U2 sb = __builtin_bit_cast(U2, s);
    const F2 s_dn = __builtin_bit_cast(F2, sb - 1u), s_up = __builtin_bit_cast(F2, sb + 1u);
    const F2 r_dn = fma2(-s_dn, s, x), r_up = fma2(-s_up, s, x);
    s.x = (r_dn.x <= 0.0f) ? s_dn.x : s.x;
    s.y = (r_dn.y <= 0.0f) ? s_dn.y : s.y;
    s.x = (r_up.x > 0.0f) ? s_up.x : s.x;
    s.y = (r_up.y > 0.0f) ? s_up.y : s.y;
    return s;
}
// a / b, correctly rounded, for normal operands whose quotient is normal too (here a in [0.85, 1.4], b in
// [3.4, 8.6]): with the correctly rounded reciprocal y, q = a*y refined once is within half an ulp and a bit, and
// the second residual step then rounds to the nearest (Markstein's division theorem) -- the value the IEEE
// division of the scalar code returns.
DI F2 div_nr(F2 a, F2 b) {
    const F2 y = rcp_nr(b);
    F2 q = a * y;
    F2 r = fma2(-b, q, a);
    q = fma2(r, y, q);
    r = fma2(-b, q, a);
    return fma2(r, y, q);
}
// integrate_edge_vec for a pair of unit-vector pairs (|x| <= 1 + a few ulp)
DI P3 integrate_edge_vec(P3 v1, P3 v2) {
    const F2 x = dot3(v1, v2);
    const F2 y = F2{__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const F2 a = splat2(0.8543985f) + (splat2(0.4965155f) + splat2(0.0145206f) * y) * y;
    const F2 b = splat2(3.4175940f) + (splat2(4.1616724f) + y) * y;
    const F2 v = div_nr(a, b);
    const F2 om = splat2(1.0f) - x * x;
    const F2 m = F2{max_std(om.x, kEps), max_std(om.y, kEps)}; // in [1e-4, 1]
    const F2 alt = splat2(0.5f) * rcp_nr(sqrt_nr(m)) - v;
    const F2 ts = F2{(x.x > 0.0f) ? v.x : alt.x, (x.y > 0.0f) ? v.y : alt.y};
    return cross3(v1, v2) * ts;
}
// Both ltc_polygon calls of one shaded point: L*.x/y/z[0] are the corners under the identity matrix, [1] under
// Minv.  Returns false, and leaves the sums alone, when some corner does not take the fast path of
// ray_dir/normalize3 (shorter than 1e-4, huge, or NaN): the caller then evaluates the scalar form.
DI bool ltc_polygon_pair(const float* __restrict__ ltc2, P3 L0, P3 L1, P3 L2, P3 L3, bool behind, bool two_sided, float& sum0,
                         float& sum1) {
    const F2 q0 = dot3(L0, L0), q1 = dot3(L1, L1), q2 = dot3(L2, L2), q3 = dot3(L3, L3);
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fminf(q0.x, q0.y), __builtin_fminf(q1.x, q1.y)),
                                     __builtin_fminf(__builtin_fminf(q2.x, q2.y), __builtin_fminf(q3.x, q3.y)));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(q0.x, q0.y), __builtin_fmaxf(q1.x, q1.y)),
                                     __builtin_fmaxf(__builtin_fmaxf(q2.x, q2.y), __builtin_fmaxf(q3.x, q3.y)));
    const bool ordered = q0.x == q0.x && q0.y == q0.y && q1.x == q1.x && q1.y == q1.y && q2.x == q2.x && q2.y == q2.y &&
                         q3.x == q3.x && q3.y == q3.y;
    if (!(ordered && lo >= kEpsSq && hi <= 0x1p80f)) return false;
    // normalize3: v * (1/len) with 1/len == 1/sqrt(len^2) (see normalize3_len)
    L0 = L0 * rcp_nr(sqrt_nr(q0));
    L1 = L1 * rcp_nr(sqrt_nr(q1));
    L2 = L2 * rcp_nr(sqrt_nr(q2));
    L3 = L3 * rcp_nr(sqrt_nr(q3));
    P3 vsum = P3{splat2(0.0f), splat2(0.0f), splat2(0.0f)};
    vsum = vsum + integrate_edge_vec(L0, L1);
    vsum = vsum + integrate_edge_vec(L1, L2);
    vsum = vsum + integrate_edge_vec(L2, L3);
    vsum = vsum + integrate_edge_vec(L3, L0);
    const F2 lsq = dot3(vsum, vsum);
    const float LUT_SCALE = 63.0f / 64.0f, LUT_BIAS = 0.5f / 64.0f; // AreaLight.hpp:47-49
    float out[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float len = length_from_sq(lsq[k]);
        float z = !eps_equal(len, 0.0f, kEps) ? vsum.z[k] / len : 0.0f;
        if (behind) z = -z;
        const float ux = (z * 0.5f + 0.5f) * LUT_SCALE + LUT_BIAS;
        const float uy = len * LUT_SCALE + LUT_BIAS;
        const uint32_t tx = ltc_texel(ux), ty = ltc_texel(uy);
        const float scale = (tx < 64u && ty < 64u) ? ltc2[(ty * 64u + tx) * 4u + 3u] : 0.0f;
        float sum = len * scale;
        if (!behind && !two_sided) sum = 0.0f;
        out[k] = sum;
    }
    sum0 = out[0];
    sum1 = out[1];
    return true;
}

// AreaLight::_clipQuadToHorizon + the clipped branch of _evaluateLTC (AreaLight.hpp:104-214,298-334) on the four
// transformed corners.  The reference never takes this branch (AreaLight::_clipless is true and has no setter);
// crt_light::horizonClip selects it.  keep/drop: the edge from a vertex above the horizon to one below is cut at
// z = 0, -drop.z * keep + keep.z * drop.
DI V3 clip_edge(V3 keep, V3 drop) { return keep * (-drop.z) + drop * keep.z; }
DI float ltc_polygon_clipped(V3 L0, V3 L1, V3 L2, V3 L3, bool two_sided) {
    uint32_t config = 0;
    if (L0.z > 0.0f) config += 1u;
    if (L1.z > 0.0f) config += 2u;
    if (L2.z > 0.0f) config += 4u;
    if (L3.z > 0.0f) config += 8u;
    V3 L4 = mk3(0.0f, 0.0f, 0.0f);
    uint32_t n = 0;
    switch (config) {
    case 0: break;
    case 1: n = 3; L1 = clip_edge(L0, L1); L2 = clip_edge(L0, L3); break;
    case 2: n = 3; L0 = clip_edge(L1, L0); L2 = clip_edge(L1, L2); break;
    case 3: n = 4; L2 = clip_edge(L1, L2); L3 = clip_edge(L0, L3); break;
    case 4: n = 3; L0 = clip_edge(L2, L3); L1 = clip_edge(L2, L1); break;
    case 5: break;
    case 6: n = 4; L0 = clip_edge(L1, L0); L3 = clip_edge(L2, L3); break;
    case 7: n = 5; L4 = clip_edge(L0, L3); L3 = clip_edge(L2, L3); break;
    case 8: n = 3; L0 = clip_edge(L3, L0); L1 = clip_edge(L3, L2); L2 = L3; break;
    case 9: n = 4; L1 = clip_edge(L0, L1); L2 = clip_edge(L3, L2); break;
    case 10: break;
    case 11: n = 5; L4 = L3; L3 = clip_edge(L3, L2); L2 = clip_edge(L1, L2); break;
    case 12: n = 4; L1 = clip_edge(L2, L1); L0 = clip_edge(L3, L0); break;
    case 13: n = 5; L4 = L3; L3 = L2; L2 = clip_edge(L2, L1); L1 = clip_edge(L0, L1); break;
    case 14: n = 5; L4 = clip_edge(L3, L0); L0 = clip_edge(L1, L0); break;
    default: n = 4; break; // 15: nothing below the horizon
    }
    if (n == 0u) return 0.0f;
    if (n == 3u) L3 = L0;
    if (n == 4u) L4 = L0;
    L0 = ray_dir(L0); // "normalise unless shorter than 1e-4"
    L1 = ray_dir(L1);
    L2 = ray_dir(L2);
    L3 = ray_dir(L3);
    L4 = ray_dir(L4);
    float sum = 0.0f;
    sum += integrate_edge_vec(L0, L1).z;
    sum += integrate_edge_vec(L1, L2).z;
    sum += integrate_edge_vec(L2, L3).z;
    if (n >= 4u) sum += integrate_edge_vec(L3, L4).z;
    if (n == 5u) sum += integrate_edge_vec(L4, L0).z;
    return two_sided ? __builtin_fabsf(sum) : max_std(0.0f, sum);
}

// main.cu:50-58
DI void surface_frame(V3 n, V3& Nx, V3& Nz) {
    if (__builtin_fabsf(n.x) > __builtin_fabsf(n.y)) Nx = mk3(n.z, 0.0f, -n.x) * rsqrt_c(n.x * n.x + n.z * n.z);
    else Nx = mk3(0.0f, -n.z, n.y) * rsqrt_c(n.y * n.y + n.z * n.z);
    Nz = cross3(n, Nx);
}

// ---------------------------------------------------------------- wave helpers
DI uint32_t lane_rank(uint64_t mask) { // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

} // namespace crt
