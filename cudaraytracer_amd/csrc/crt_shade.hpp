// Device code shared by the render kernels: work-unit -> pixel mapping and the per-light shading
// routines (LightList::calculateColor's two cases).
#pragma once
#include "crt_device.hpp"
#include "crt_render.hpp"

namespace crt {

// Which pixel slot `idx` (0..63) of work unit `unit` is, and where its result goes.
struct PixelMap {
    bool valid;
    uint32_t i, j;     // pixel coordinates
    uint32_t out_index; // RGBA slot in the output buffer (in pixels)
};

DI PixelMap map_pixel(const RenderArgs& a, uint32_t unit, uint32_t idx) {
    PixelMap m;
    m.valid = false;
    m.i = m.j = 0;
    m.out_index = 0;
    if (a.parts > 1u) unit = unit * a.parts + a.part; // a piece of a larger launch: that launch's unit
    if (a.mode == MODE_FULL || a.mode == MODE_SHARD) {
        uint32_t tile = (a.mode == MODE_SHARD) ? unit * a.world + a.rank : unit;
        uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        m.i = tx * 8u + (idx & 7u);
        m.j = ty * 8u + (idx >> 3);
        m.valid = (m.i < a.nx) && (m.j < a.ny) && (tile < a.tiles_x * a.tiles_y);
        m.out_index = (a.mode == MODE_SHARD) ? unit * 64u + idx : m.j * a.nx + m.i;
    } else { // MODE_PARTIAL: window of reference blocks (render_partial, main.cu:349-353)
        uint32_t per_block = a.tx * a.ty;
        uint32_t p = unit * 64u + idx;
        uint32_t b = p / per_block, q = p - b * per_block;
        if (b < a.win_bx * a.win_by) {
            uint32_t by = b / a.win_bx, bx = b - by * a.win_bx;
            uint32_t ly = q / a.tx, lx = q - ly * a.tx;
            m.i = lx + ((a.start_bx + bx) % a.blocks_x) * a.tx;
            m.j = ly + ((a.start_by + by) % a.blocks_y) * a.ty;
            m.valid = (m.i < a.nx) && (m.j < a.ny);
            m.out_index = m.j * a.nx + m.i;
        }
    }
    return m;
}

// LTC part of AreaLight::calculateColor (AreaLight.hpp:595-649) once the visibility is known.  CLIP: the scene has
// a light with crt_light::horizonClip set (a separate instantiation, so that kernels for ordinary scenes do not
// carry the clipping code's registers).
template <bool CLIP>
DI V4 area_light_shade(const float* __restrict__ ltc, const DevLight CRT_CONST* L,
                       const DevGeomShade CRT_CONST* g, V3 P, V3 N, V3 view, float visibility) {
    V3 lcol = ld3c(L->color);
    V3 ambient = mk3(g->ambient[0], g->ambient[1], g->ambient[2]) * lcol;
    const float* ltc1 = ltc;
    const float* ltc2 = ltc + 64 * 64 * 4;
    V3 dcol = mk3(g->diffuse[0], g->diffuse[1], g->diffuse[2]);
    V3 scol = mk3(g->specular[0], g->specular[1], g->specular[2]);
    float ndotv = saturate_c(dot3(N, view));
    const float LUT_SCALE = 63.0f / 64.0f, LUT_BIAS = 0.5f / 64.0f;
    float uvy = rcp_exact(rsqrt_c(1.0f - ndotv)) * LUT_SCALE + LUT_BIAS;
    uint32_t ty = ltc_texel(uvy), tx = g->ltc_col;
    V4 t1 = V4{0, 0, 0, 0}, t2 = V4{0, 0, 0, 0};
    if (tx < 64u && ty < 64u) {
        const float4 a = *reinterpret_cast<const float4*>(ltc1 + (ty * 64u + tx) * 4u);
        const float4 b = *reinterpret_cast<const float4*>(ltc2 + (ty * 64u + tx) * 4u);
        t1 = V4{a.x, a.y, a.z, a.w};
        t2 = V4{b.x, b.y, b.z, b.w};
    }
    // _evaluateLTC twice (AreaLight.hpp:216-297): identity, then Minv
    V3 T1 = view - N * dot3(view, N);
    if (!eps_equal(len_sq(T1), 0.0f, kEpsSq)) T1 = normalize3(T1);
    V3 T2 = cross3(N, T1);
    V3 w0 = ld3c(L->pts[0]) - P, w1 = ld3c(L->pts[1]) - P, w2 = ld3c(L->pts[2]) - P, w3 = ld3c(L->pts[3]) - P;
    bool behind = dot3(w0, ld3c(L->light_normal)) < 0.0f;
    bool two_sided = L->two_sided != 0u;
    const bool clip = CLIP && L->horizon_clip != 0u; // wave-uniform: the clipped branch of _evaluateLTC (AreaLight.hpp:298-334)
    // Minv(t1) * transpose(T1,T2,N): row0 = t1.x*T1 + t1.z*N, row1 = T2, row2 = t1.y*T1 + t1.w*N
    // (mat3 product accumulates 0 + a*T1 + 0*T2 + c*N per entry, mat.hpp:1481-1492)
    V3 r0 = mk3((0.0f + t1.x * T1.x) + 0.0f * T2.x + t1.z * N.x, (0.0f + t1.x * T1.y) + 0.0f * T2.y + t1.z * N.y,
                (0.0f + t1.x * T1.z) + 0.0f * T2.z + t1.z * N.z);
    V3 r1 = mk3((0.0f + 0.0f * T1.x) + 1.0f * T2.x + 0.0f * N.x, (0.0f + 0.0f * T1.y) + 1.0f * T2.y + 0.0f * N.y,
                (0.0f + 0.0f * T1.z) + 1.0f * T2.z + 0.0f * N.z);
    V3 r2 = mk3((0.0f + t1.y * T1.x) + 0.0f * T2.x + t1.w * N.x, (0.0f + t1.y * T1.y) + 0.0f * T2.y + t1.w * N.y,
                (0.0f + t1.y * T1.z) + 0.0f * T2.z + t1.w * N.z);
    // Both evaluations side by side: rows (T1, T2, N) of identity * transpose(T1,T2,N) in component 0, rows
    // (r0, r1, r2) in component 1; corner k under both matrices is (row0 . wk, row1 . wk, row2 . wk).
    const P3 row0 = P3{F2{T1.x, r0.x}, F2{T1.y, r0.y}, F2{T1.z, r0.z}};
    const P3 row1 = P3{F2{T2.x, r1.x}, F2{T2.y, r1.y}, F2{T2.z, r1.z}};
    const P3 row2 = P3{F2{N.x, r2.x}, F2{N.y, r2.y}, F2{N.z, r2.z}};
    const P3 c0 = P3{dot3(row0, w0), dot3(row1, w0), dot3(row2, w0)}, c1 = P3{dot3(row0, w1), dot3(row1, w1), dot3(row2, w1)};
    const P3 c2 = P3{dot3(row0, w2), dot3(row1, w2), dot3(row2, w2)}, c3 = P3{dot3(row0, w3), dot3(row1, w3), dot3(row2, w3)};
    float dsum = 0.0f, ssum = 0.0f;
#ifdef CRT_SHADE_SCALAR // measuring aid: both evaluations through the scalar form, inlined
    const bool paired = false;
#define ltc_polygon_rare ltc_polygon
#else
    const bool paired = !clip && ltc_polygon_pair(ltc2, c0, c1, c2, c3, behind, two_sided, dsum, ssum);
#endif
    if (!paired) {
        const V3 d0 = half3(c0, 0), d1 = half3(c1, 0), d2 = half3(c2, 0), d3 = half3(c3, 0);
        const V3 s0 = half3(c0, 1), s1 = half3(c1, 1), s2 = half3(c2, 1), s3 = half3(c3, 1);
        dsum = clip ? ltc_polygon_clipped(d0, d1, d2, d3, two_sided) : ltc_polygon_rare(ltc2, d0, d1, d2, d3, behind, two_sided);
        ssum = clip ? ltc_polygon_clipped(s0, s1, s2, s3, two_sided) : ltc_polygon_rare(ltc2, s0, s1, s2, s3, behind, two_sided);
    }
#ifdef CRT_SHADE_SCALAR
#undef ltc_polygon_rare
#endif
    V3 diffuse = (lcol * mk3(dsum, dsum, dsum)) * dcol;
    V3 spec = mk3(ssum, ssum, ssum) * (scol * t2.x + (mk3(1.0f, 1.0f, 1.0f) - scol) * t2.y);
    V3 specular = lcol * spec;
    V3 result = ambient + ((diffuse + specular) * L->intensity) * visibility;
    return V4{result.x, result.y, result.z, 1.0f};
}

// Light::calculateColor, point light with Blinn-Phong (Light.hpp:92-111,119-182).
DI V4 point_light_color(ScenePtr sc, const DevLight CRT_CONST* L, const DevGeomShade CRT_CONST* g, V3 P, V3 N, V3 view) {
    V3 lcol = ld3c(L->color);
    V3 ambient = mk3(g->ambient[0], g->ambient[1], g->ambient[2]) * lcol;
    V3 lpos = ld3c(L->pos);
    V3 rayDir = lpos - P;
    float lightDist = length3(rayDir);
    if (!eps_equal(lightDist, 0.0f, kEps)) rayDir = normalize3(rayDir);
    V3 so = P + rayDir * 0.01f, sd = ray_dir(rayDir);
    Hit h = closest_hit(sc, so, sd, max_std(0.0f, lightDist));
    bool covered = h.idx >= 0 && (h.mat_type != CRT_MAT_REFRACTIVE || covered_through_glass(sc, L, so, sd, h));
    if (covered) return V4{ambient.x, ambient.y, ambient.z, 1.0f};
    V3 lightDir = lpos - P;
    if (!eps_equal(len_sq(lightDir), 0.0f, kEpsSq)) lightDir = normalize3(lightDir);
    float diff = max_std(dot3(N, lightDir), 0.0f);
    V3 diffuse = (lcol * (mk3(g->diffuse[0], g->diffuse[1], g->diffuse[2]) * diff)) * L->intensity;
    V3 halfway = normalize3(lightDir + view);
    float spec = pow_c(max_std(dot3(N, halfway), 0.0f), g->shininess);
    V3 specular = (lcol * (mk3(g->specular[0], g->specular[1], g->specular[2]) * spec)) * L->intensity;
    V3 result = (ambient + diffuse) + specular;
    return V4{result.x, result.y, result.z, 1.0f};
}

} // namespace crt
