// Render kernels for gfx950: random_init and the per-pixel integrator
// (render -> aa_color -> color -> GeometryList::hit / AreaLight::calculateColor,
// main.cu:110-347).
//
// Execution shape.  One lane = one pixel; a wave = an 8x8 pixel tile; one wave
// per workgroup, so the dispatcher balances uneven tiles over the 1024 SIMDs.
// The reference's two explicit stacks (RayState[128] in color(), AA_Task[256] in
// aa_color()) are replaced by O(depth) per-lane state:
//   * a pending mirror/glass ray (a Reflect/Refractive hit pushes one ray that is
//     popped at once),
//   * one Frame per nested global-illumination level.  A diffuse hit at depth>0
//     draws its ind_rays (r1,r2) pairs exactly as the reference does, but keeps
//     only the generator state reached after the last *accepted* child; children
//     are popped last-to-first by stepping that state backwards (XORWOW's
//     xorshift is a bijection), so no child is ever stored.  The reference's
//     128-entry truncation is a function of how many entries are on its stack,
//     which here is the sum of the frames' remaining-children counters.
//   * a 10-entry adaptive-AA task list (aa_iter <= 4 bounds it: 1 + 3*3).
// Each lane alternates between a cheap phase that walks rays until it owns a
// diffuse hit to shade, and the expensive shading phase (shadowSamples shadow
// rays + two LTC polygon integrals) that all lanes of the wave enter together.
#include <hip/hip_runtime.h>

#include "crt_device.hpp"
#include "crt_render.hpp"

namespace crt {

namespace {

struct Frame {
    float P[3];   // hit point the children leave from
    float N[3];   // its shading normal
    float att[4]; // attenuation of the expanding ray (RayState::attenuation)
    uint32_t rng[6]; // generator state right after the draws of child `remaining-1`
    uint32_t geom;   // primitive hit (for Material::diffuse)
    uint32_t remaining; // children still on the reference's stack
    uint32_t child_depth; // global_depth of the children
};

struct AATask {
    float cx, cy;
    uint32_t level; // AA_Task::sample
};

// Which pixel a lane renders and where its result goes.
struct PixelMap {
    bool valid;
    uint32_t i, j;    // pixel coordinates
    size_t out_index; // RGBA slot in the output buffer (in pixels)
    bool store;       // write even when !valid (shard padding)
};

DI PixelMap map_pixel(const RenderArgs& a, uint32_t wave, uint32_t lane) {
    PixelMap m;
    m.valid = false;
    m.store = false;
    m.i = m.j = 0;
    m.out_index = 0;
    if (a.mode == MODE_FULL || a.mode == MODE_SHARD) {
        uint32_t tile = (a.mode == MODE_SHARD) ? wave * a.world + a.rank : wave;
        uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        m.i = tx * 8u + (lane & 7u);
        m.j = ty * 8u + (lane >> 3);
        m.valid = (m.i < a.nx) && (m.j < a.ny) && (tile < a.tiles_x * a.tiles_y);
        if (a.mode == MODE_SHARD) {
            m.out_index = (size_t)wave * 64u + lane;
            m.store = true;
        } else {
            m.out_index = (size_t)m.j * a.nx + m.i;
        }
    } else { // MODE_PARTIAL: window of reference blocks (render_partial, main.cu:349-353)
        uint32_t per_block = a.tx * a.ty;
        uint32_t p = wave * 64u + lane;
        uint32_t b = p / per_block, q = p - b * per_block;
        if (b < a.win_bx * a.win_by) {
            uint32_t by = b / a.win_bx, bx = b - by * a.win_bx;
            uint32_t ly = q / a.tx, lx = q - ly * a.tx;
            m.i = lx + ((a.start_bx + bx) % a.blocks_x) * a.tx;
            m.j = ly + ((a.start_by + by) % a.blocks_y) * a.ty;
            m.valid = (m.i < a.nx) && (m.j < a.ny);
            m.out_index = (size_t)m.j * a.nx + m.i;
        }
    }
    return m;
}

// AreaLight::calculateColor (AreaLight.hpp:571-650) incl. _calculateVisibility (:484-529).
DI V4 area_light_color(const DevScene* __restrict__ sc, const float* __restrict__ ltc, const DevLight& L,
                       const DevGeom& g, V3 P, V3 N, V3 view, Rng& rng) {
    V3 lcol = ld3(L.color);
    V3 ambient = mk3(g.ambient[0], g.ambient[1], g.ambient[2]) * lcol;
    if (!L.initialized) return V4{ambient.x, ambient.y, ambient.z, 1.0f};

    // ---- soft shadow: shadow_samples rays to uniformly drawn points of the quad
    const uint32_t S = L.shadow_samples;
    uint32_t shadowed = 0;
    const V3 lpos = ld3(L.pos);
    for (uint32_t s = 0; s < S; ++s) {
        float u = rng_uniform(rng);
        float v = rng_uniform(rng);
        V3 sample; // _randomPoint (AreaLight.hpp:340-375)
        if (u + v < 1.0f) {
            sample = ((ld3(L.p[0]) + ld3(L.e01) * u) + ld3(L.e03) * v) + lpos;
        } else {
            u = 1.0f - u;
            v = 1.0f - v;
            sample = ((ld3(L.p[2]) + ld3(L.e23) * u) + ld3(L.e21) * v) + lpos;
        }
        V3 dir = sample - P;
        float dist = length3(dir);
        if (!eps_equal(dist, 0.0f, kEps)) dir = normalize3(dir);
        V3 o = P + dir * 0.01f;
        V3 d = ray_dir(dir);                  // Ray ctor normalises once more (Ray.hpp:45-52)
        float maxd = max_std(0.0f, dist);     // Ray::setDistance
        if (is_covered(sc, L, o, d, maxd)) ++shadowed;
    }
    float visibility = 1.0f - ((float)shadowed / (float)S);

    // ---- LTC lookups (AreaLight.hpp:604-635)
    const float* ltc1 = ltc;
    const float* ltc2 = ltc + 64 * 64 * 4;
    V3 dcol = mk3(g.diffuse[0], g.diffuse[1], g.diffuse[2]);
    V3 scol = mk3(g.specular[0], g.specular[1], g.specular[2]);
    float ndotv = saturate_c(dot3(N, view));
    const float LUT_SCALE = 63.0f / 64.0f, LUT_BIAS = 0.5f / 64.0f;
    float uvy = (1.0f / rsqrt_c(1.0f - ndotv)) * LUT_SCALE + LUT_BIAS;
    uint32_t ty = ltc_texel(uvy), tx = g.ltc_col;
    V4 t1 = V4{0, 0, 0, 0}, t2 = V4{0, 0, 0, 0};
    if (tx < 64u && ty < 64u) {
        const float4 a = *reinterpret_cast<const float4*>(ltc1 + (ty * 64u + tx) * 4u);
        const float4 b = *reinterpret_cast<const float4*>(ltc2 + (ty * 64u + tx) * 4u);
        t1 = V4{a.x, a.y, a.z, a.w};
        t2 = V4{b.x, b.y, b.z, b.w};
    }

    // ---- _evaluateLTC twice (AreaLight.hpp:216-297): identity, then Minv
    V3 T1 = view - N * dot3(view, N);
    if (!eps_equal(len_sq(T1), 0.0f, kEpsSq)) T1 = normalize3(T1);
    V3 T2 = cross3(N, T1);
    V3 w0 = ld3(L.pts[0]) - P, w1 = ld3(L.pts[1]) - P, w2 = ld3(L.pts[2]) - P, w3 = ld3(L.pts[3]) - P;
    bool behind = dot3(w0, ld3(L.light_normal)) < 0.0f;
    bool two_sided = L.two_sided != 0u;
    // identity * transpose(T1,T2,N): rows T1, T2, N
    float dsum = ltc_polygon(ltc2, mk3(dot3(T1, w0), dot3(T2, w0), dot3(N, w0)),
                             mk3(dot3(T1, w1), dot3(T2, w1), dot3(N, w1)),
                             mk3(dot3(T1, w2), dot3(T2, w2), dot3(N, w2)),
                             mk3(dot3(T1, w3), dot3(T2, w3), dot3(N, w3)), behind, two_sided);
    // Minv(t1) * transpose(T1,T2,N): row0 = t1.x*T1 + t1.z*N, row1 = T2, row2 = t1.y*T1 + t1.w*N
    // (mat3 product accumulates 0 + a*T1 + 0*T2 + c*N per entry, mat.hpp:1481-1492)
    V3 r0 = mk3((0.0f + t1.x * T1.x) + 0.0f * T2.x + t1.z * N.x, (0.0f + t1.x * T1.y) + 0.0f * T2.y + t1.z * N.y,
                (0.0f + t1.x * T1.z) + 0.0f * T2.z + t1.z * N.z);
    V3 r1 = mk3((0.0f + 0.0f * T1.x) + 1.0f * T2.x + 0.0f * N.x, (0.0f + 0.0f * T1.y) + 1.0f * T2.y + 0.0f * N.y,
                (0.0f + 0.0f * T1.z) + 1.0f * T2.z + 0.0f * N.z);
    V3 r2 = mk3((0.0f + t1.y * T1.x) + 0.0f * T2.x + t1.w * N.x, (0.0f + t1.y * T1.y) + 0.0f * T2.y + t1.w * N.y,
                (0.0f + t1.y * T1.z) + 0.0f * T2.z + t1.w * N.z);
    float ssum = ltc_polygon(ltc2, mk3(dot3(r0, w0), dot3(r1, w0), dot3(r2, w0)),
                             mk3(dot3(r0, w1), dot3(r1, w1), dot3(r2, w1)),
                             mk3(dot3(r0, w2), dot3(r1, w2), dot3(r2, w2)),
                             mk3(dot3(r0, w3), dot3(r1, w3), dot3(r2, w3)), behind, two_sided);

    V3 diffuse = (lcol * mk3(dsum, dsum, dsum)) * dcol;
    V3 spec = mk3(ssum, ssum, ssum) * (scol * t2.x + (mk3(1.0f, 1.0f, 1.0f) - scol) * t2.y);
    V3 specular = lcol * spec;
    V3 result = ambient + ((diffuse + specular) * L.intensity) * visibility;
    return V4{result.x, result.y, result.z, 1.0f};
}

// Light::calculateColor, point light with Blinn-Phong (Light.hpp:92-111,119-182).
DI V4 point_light_color(const DevScene* __restrict__ sc, const DevLight& L, const DevGeom& g, V3 P, V3 N, V3 view) {
    V3 lcol = ld3(L.color);
    V3 ambient = mk3(g.ambient[0], g.ambient[1], g.ambient[2]) * lcol;
    V3 lpos = ld3(L.pos);
    V3 rayDir = lpos - P;
    float lightDist = length3(rayDir);
    if (!eps_equal(lightDist, 0.0f, kEps)) rayDir = normalize3(rayDir);
    if (is_covered(sc, L, P + rayDir * 0.01f, ray_dir(rayDir), max_std(0.0f, lightDist)))
        return V4{ambient.x, ambient.y, ambient.z, 1.0f};
    V3 lightDir = lpos - P;
    if (!eps_equal(len_sq(lightDir), 0.0f, kEpsSq)) lightDir = normalize3(lightDir);
    float diff = max_std(dot3(N, lightDir), 0.0f);
    V3 diffuse = (lcol * (mk3(g.diffuse[0], g.diffuse[1], g.diffuse[2]) * diff)) * L.intensity;
    V3 halfway = normalize3(lightDir + view);
    float spec = __builtin_powf(max_std(dot3(N, halfway), 0.0f), g.shininess);
    V3 specular = (lcol * (mk3(g.specular[0], g.specular[1], g.specular[2]) * spec)) * L.intensity;
    V3 result = (ambient + diffuse) + specular;
    return V4{result.x, result.y, result.z, 1.0f};
}

} // namespace

__global__ void __launch_bounds__(64) random_init_kernel(uint32_t npix, uint64_t seed, uint32_t* __restrict__ state) {
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    if (idx >= npix) return;
    Rng r;
    rng_seed(r, seed + (uint64_t)idx); // curand_init(seed + idx, 0, 0) (main.cu:324)
    state[0 * (size_t)npix + idx] = r.v0;
    state[1 * (size_t)npix + idx] = r.v1;
    state[2 * (size_t)npix + idx] = r.v2;
    state[3 * (size_t)npix + idx] = r.v3;
    state[4 * (size_t)npix + idx] = r.v4;
    state[5 * (size_t)npix + idx] = r.d;
}

__global__ void __launch_bounds__(64) render_kernel(RenderArgs a) {
    const DevScene* __restrict__ sc = a.scene;
    const float* __restrict__ ltc = a.ltc;
    const uint32_t lane = threadIdx.x;
    const PixelMap pm = map_pixel(a, blockIdx.x, lane);
    float4* __restrict__ out = reinterpret_cast<float4*>(a.out);
    if (!pm.valid) {
        if (pm.store) out[pm.out_index] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const size_t npix = (size_t)a.nx * a.ny;
    const size_t pix = (size_t)pm.j * a.nx + pm.i;
    Rng rng;
    rng.v0 = a.rng[0 * npix + pix];
    rng.v1 = a.rng[1 * npix + pix];
    rng.v2 = a.rng[2 * npix + pix];
    rng.v3 = a.rng[3 * npix + pix];
    rng.v4 = a.rng[4 * npix + pix];
    rng.d = a.rng[5 * npix + pix];

    // get_world_coordinates (main.cu:306-314) and pixel_size (main.cu:337)
    const float height_world = 2.0f;
    const float width_world = height_world * ((float)a.nx / (float)a.ny);
    float x_world, y_world;
    {
        float denom = (float)a.nx - 0.0f; // remap<float,true> (math_functions.hpp:31-45)
        float es = -width_world * 0.5f, ee = width_world * 0.5f;
        x_world = (denom == 0.0f) ? (float)pm.i : es + ((ee - es) / denom) * ((float)pm.i - 0.0f);
        float denomy = (float)a.ny - 0.0f;
        y_world = (denomy == 0.0f) ? (float)pm.j : 1.0f + ((-1.0f - 1.0f) / denomy) * ((float)pm.j - 0.0f);
    }
    const float pix_w = width_world / (float)a.nx, pix_h = height_world / (float)a.ny;

    // Camera::getRay constants (Camera.hpp:97-98,133-134)
    const DevCamera& cam = sc->cam;
    const V3 cam_pos = ld3(cam.pos), cam_front = ld3(cam.front), cam_right = ld3(cam.right), cam_up = ld3(cam.up);
    const V3 eye = cam_pos - cam_front * (width_world * 0.5f * cam.one_over_tan);
    float ortho = 0.0f;
    {
        float aspect = height_world / width_world;
        ortho = cam.ortho_scale * (aspect > 1.0f ? aspect : 1.0f);
    }

    const uint32_t aa_iter = a.aa_iter > 4u ? 4u : a.aa_iter; // main.cu:233
    const float inv_ind = 1.0f / (float)a.ind_rays;            // main.cu:182

    // ---- adaptive-AA task list (aa_color, main.cu:230-304)
    AATask aa_stack[10];
    uint32_t aa_top = 0;
    aa_stack[aa_top++] = AATask{x_world, y_world, aa_iter};
    V4 acc = V4{0, 0, 0, 0};
    bool task_active = false;
    float t_cx = 0, t_cy = 0;
    uint32_t t_level = 0, t_corner = 0;
    bool t_all_equal = true;
    V4 t_c0 = V4{0, 0, 0, 0};

    // ---- color() state (main.cu:110-227)
    Frame frames[CRT_MAX_GL_ITER];
    uint32_t nframes = 0;
    uint32_t on_stack = 0; // entries the reference's stack would hold
    bool color_active = false;
    V4 final_c = V4{0, 0, 0, 0};
    V3 ray_d = mk3(0.0f, 0.0f, -1.0f); // Ray::_direction survives a degenerate setDirection (Ray.hpp:62-69)
    // pending ray (popped next)
    bool cur_valid = false;
    V3 cur_o = mk3(0, 0, 0), cur_d = mk3(0, 0, 0);
    V4 cur_att = V4{1, 1, 1, 1};
    uint32_t cur_ref = 0, cur_depth = 0;

    // node to shade
    V3 n_P = mk3(0, 0, 0), n_N = mk3(0, 0, 0), n_view = mk3(0, 0, 0);
    V4 n_att = V4{0, 0, 0, 0};
    uint32_t n_geom = 0, n_depth = 0;
    uint32_t cnt_shade = 0, cnt_expand = 0;

    for (;;) {
        // ================= phase A: walk rays until this lane owns a diffuse hit
        bool have_node = false;
        bool done = false;
        while (!have_node) {
            if (!color_active) {
                if (!task_active) {
                    if (aa_top == 0) {
                        done = true;
                        break;
                    }
                    AATask t = aa_stack[--aa_top];
                    t_cx = t.cx;
                    t_cy = t.cy;
                    t_level = t.level;
                    t_corner = 0;
                    t_all_equal = true;
                    task_active = true;
                }
                // size of this task's box: pixel size halved (aa_iter - level) times (main.cu:251,297)
                float sx = pix_w, sy = pix_h;
                for (uint32_t k = t_level; k < aa_iter; ++k) {
                    sx = sx * 0.5f;
                    sy = sy * 0.5f;
                }
                float hx = sx * 0.5f, hy = sy * 0.5f;
                float ox = (t_corner & 1u) ? 1.0f : -1.0f, oy = (t_corner & 2u) ? 1.0f : -1.0f;
                float rx, ry;
                if (t_level == 0u) {
                    rx = t_cx;
                    ry = t_cy;
                } else if (t_level == 1u) {
                    rx = t_cx + ox * hx * 0.5f;
                    ry = t_cy + oy * hy * 0.5f;
                } else {
                    rx = t_cx + ox * hx;
                    ry = t_cy + oy * hy;
                }
                // Camera::getRay (Camera.hpp:86-144)
                V3 ro, rd;
                if (cam.type == CRT_CAM_ORTHOGRAPHIC) {
                    ro = ((cam_right * rx) * ortho + (cam_up * ry) * ortho) + cam_pos;
                    rd = ray_dir(cam_front);
                } else {
                    ro = (cam_right * rx + cam_up * ry) + cam_pos;
                    rd = ray_dir(ro - eye);
                }
                cur_o = ro;
                cur_d = rd;
                cur_att = V4{1.0f, 1.0f, 1.0f, 1.0f};
                cur_ref = a.ref_iter;
                cur_depth = a.gl_iter;
                cur_valid = true;
                color_active = true;
                final_c = V4{0, 0, 0, 0};
                ray_d = mk3(0.0f, 0.0f, -1.0f);
                nframes = 0;
                on_stack = 0;
            }
            if (!cur_valid) {
                if (nframes == 0) {
                    // color() returned: hand the colour to the AA task (main.cu:254-300)
                    color_active = false;
                    float mult = 1.0f / (float)(1u << (2u * (aa_iter - t_level)));
                    if (t_level == 0u) {
                        acc = acc + final_c * mult;
                        task_active = false;
                    } else if (t_level == 1u) {
                        acc = acc + (final_c * mult) * 0.25f;
                        if (++t_corner == 4u) task_active = false;
                    } else {
                        if (t_corner == 0u) t_c0 = final_c;
                        else if (t_all_equal && !equal4(final_c, t_c0)) t_all_equal = false;
                        if (++t_corner == 4u) {
                            task_active = false;
                            if (t_all_equal) {
                                acc = acc + t_c0 * mult;
                            } else {
                                float sx = pix_w, sy = pix_h;
                                for (uint32_t k = t_level; k < aa_iter; ++k) {
                                    sx = sx * 0.5f;
                                    sy = sy * 0.5f;
                                }
                                float hx = sx * 0.5f, hy = sy * 0.5f;
                                for (uint32_t c = 0; c < 4u; ++c) {
                                    float ox = (c & 1u) ? 1.0f : -1.0f, oy = (c & 2u) ? 1.0f : -1.0f;
                                    aa_stack[aa_top++] = AATask{t_cx + ox * hx * 0.5f, t_cy + oy * hy * 0.5f, t_level - 1u};
                                }
                            }
                        }
                    }
                    continue;
                }
                // pop the next child of the innermost frame
                Frame& f = frames[nframes - 1];
                Rng rv;
                rv.v0 = f.rng[0];
                rv.v1 = f.rng[1];
                rv.v2 = f.rng[2];
                rv.v3 = f.rng[3];
                rv.v4 = f.rng[4];
                rv.d = f.rng[5];
                float r2 = rng_current_uniform(rv);
                rng_unstep(rv);
                float r1 = rng_current_uniform(rv);
                rng_unstep(rv);
                V3 hp = ld3(f.P), hn = ld3(f.N);
                V4 patt = ld4(f.att);
                const DevGeom& pg = sc->geom[f.geom];
                uint32_t cdepth = f.child_depth;
                uint32_t rem = f.remaining - 1u;
                if (rem == 0u) {
                    --nframes;
                } else {
                    f.remaining = rem;
                    f.rng[0] = rv.v0;
                    f.rng[1] = rv.v1;
                    f.rng[2] = rv.v2;
                    f.rng[3] = rv.v3;
                    f.rng[4] = rv.v4;
                    f.rng[5] = rv.d;
                }
                --on_stack;
                // get_random_ray_values_in_hemisphere (main.cu:60-74)
                V3 Nx, Nz;
                surface_frame(hn, Nx, Nz);
                const float PI_2 = 6.28318530717958647692f;
                float sinTheta = 1.0f - r1 * r1;
                float phi = r2 * PI_2;
                float sp, cp;
                sincos_c(phi, sp, cp);
                float x = sinTheta * cp, z = sinTheta * sp;
                V3 w = mk3(x * Nz.x + r1 * hn.x + z * Nx.x, x * Nz.y + r1 * hn.y + z * Nx.y, x * Nz.z + r1 * hn.z + z * Nx.z);
                cur_o = hp + w * 0.01f;
                cur_d = w;
                cur_att = ((ld4(pg.diffuse) * (2.0f * r1)) * patt) * inv_ind; // main.cu:186
                cur_ref = a.ref_iter;
                cur_depth = cdepth;
                cur_valid = true;
            }
            // ---- pop (main.cu:132-147)
            cur_valid = false;
            if (cur_ref == 0u) continue;
            if (!eps_equal(len_sq(cur_d), 0.0f, kEpsSq)) ray_d = normalize3(cur_d);
            const V3 ro = cur_o;
            Hit h = closest_hit(sc, ro, ray_d, 0.0f);
            if (h.idx < 0) {
                final_c = final_c + cur_att * V4{0.0f, 0.0f, 0.0f, 1.0f}; // Color::black()
                continue;
            }
            const DevGeom& g = sc->geom[h.idx];
            V3 P, N;
            hit_point_normal(sc, h, ro, ray_d, P, N);
            if (g.mat_type == CRT_MAT_DIFFUSE) {
                have_node = true;
                n_P = P;
                n_N = N;
                n_view = neg(cur_d); // viewDir = -state.r_dir, the stored (un-normalised) direction
                n_att = cur_att;
                n_geom = (uint32_t)h.idx;
                n_depth = cur_depth;
            } else if (g.mat_type == CRT_MAT_REFLECT) {
                // get_reflect_ray (main.cu:76-81)
                V3 I = normalize3(ray_d);
                V3 rdir = reflect3(I, normalize3(N));
                cur_o = P + rdir * 0.01f;
                cur_d = rdir;
                cur_att = cur_att * ld4(g.diffuse);
                cur_ref = cur_ref - 1u;
                cur_valid = true;
            } else {
                // get_refraction_ray (main.cu:83-108)
                bool front_face = dot3(ray_d, N) < 0.0f;
                V3 norm = front_face ? N : neg(N);
                float ratio = g.refract_index / 1.0f;
                if (front_face) ratio = 1.0f / ratio;
                V3 rdn = normalize3(ray_d);
                float cos_theta = __builtin_fminf(dot3(neg(rdn), norm), 1.0f);
                float sin_theta = 1.0f / rsqrt_c(1.0f - cos_theta * cos_theta);
                bool cannot_refract = ratio * sin_theta > 1.0f; // reflectance(...) > 1 never holds
                V3 dir = cannot_refract ? reflect3(rdn, norm) : refract3(rdn, norm, ratio);
                cur_o = P + dir * 0.01f;
                cur_d = dir;
                cur_att = cur_att * ld4(g.diffuse);
                cur_ref = cur_ref - 1u;
                cur_valid = true;
            }
        }
        if (done) break;

        // ================= phase B: shade the diffuse hit (main.cu:153-195)
        const DevGeom& g = sc->geom[n_geom];
        ++cnt_shade;
        V4 direct = V4{0, 0, 0, 0}; // LightList::calculateColor (LightList.hpp:23-36)
        for (uint32_t li = 0; li < sc->n_lights; ++li) {
            const DevLight& L = sc->lights[li];
            V4 c = (L.kind == CRT_LIGHT_AREA) ? area_light_color(sc, ltc, L, g, n_P, n_N, n_view, rng)
                                              : point_light_color(sc, L, g, n_P, n_N, n_view);
            direct = direct + c;
        }
        if (n_depth > 0u) {
            ++cnt_expand;
            // The reference draws (r1,r2) for every child and pushes while its stack has room (main.cu:176-191).
            uint32_t room = on_stack < CRT_RAY_STACK ? CRT_RAY_STACK - on_stack : 0u;
            uint32_t accepted = a.ind_rays < room ? a.ind_rays : room;
            for (uint32_t k = 0; k < 2u * accepted; ++k) rng_step(rng);
            if (accepted > 0u) {
                Frame& f = frames[nframes++];
                f.P[0] = n_P.x; f.P[1] = n_P.y; f.P[2] = n_P.z;
                f.N[0] = n_N.x; f.N[1] = n_N.y; f.N[2] = n_N.z;
                f.att[0] = n_att.x; f.att[1] = n_att.y; f.att[2] = n_att.z; f.att[3] = n_att.w;
                f.rng[0] = rng.v0; f.rng[1] = rng.v1; f.rng[2] = rng.v2;
                f.rng[3] = rng.v3; f.rng[4] = rng.v4; f.rng[5] = rng.d;
                f.geom = n_geom;
                f.remaining = accepted;
                f.child_depth = n_depth - 1u;
                on_stack += accepted;
            }
            for (uint32_t k = 2u * accepted; k < 2u * a.ind_rays; ++k) rng_step(rng);
        }
        final_c = final_c + n_att * direct;
    }

    // render() epilogue (main.cu:339-346)
    out[pm.out_index] = make_float4(saturate_c(acc.x), saturate_c(acc.y), saturate_c(acc.z), saturate_c(acc.w));
    a.rng[0 * npix + pix] = rng.v0;
    a.rng[1 * npix + pix] = rng.v1;
    a.rng[2 * npix + pix] = rng.v2;
    a.rng[3 * npix + pix] = rng.v3;
    a.rng[4 * npix + pix] = rng.v4;
    a.rng[5 * npix + pix] = rng.d;
    a.counters[pix] += cnt_shade;
    a.counters[npix + pix] += cnt_expand;
}

// Scatter of gathered shard buffers into the row-major framebuffer.
__global__ void __launch_bounds__(64) scatter_kernel(uint32_t nx, uint32_t ny, uint32_t tiles_x, uint32_t tiles_y,
                                                     uint32_t world, uint32_t tiles_max,
                                                     const float4* __restrict__ gathered, float4* __restrict__ fb) {
    uint32_t tile = blockIdx.x, lane = threadIdx.x;
    if (tile >= tiles_x * tiles_y) return;
    uint32_t rank = tile % world, k = tile / world;
    uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    uint32_t i = tx * 8u + (lane & 7u), j = ty * 8u + (lane >> 3);
    if (i < nx && j < ny) fb[(size_t)j * nx + i] = gathered[((size_t)rank * tiles_max + k) * 64u + lane];
}

void launch_random_init(uint32_t npix, uint64_t seed, uint32_t* state, hipStream_t stream) {
    hipLaunchKernelGGL(random_init_kernel, dim3((npix + 63u) / 64u), dim3(64), 0, stream, npix, seed, state);
}
void launch_render(const RenderArgs& args, uint32_t waves, hipStream_t stream) {
    hipLaunchKernelGGL(render_kernel, dim3(waves), dim3(64), 0, stream, args);
}
void launch_scatter(uint32_t nx, uint32_t ny, uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_max,
                    const float* gathered, float* fb, hipStream_t stream) {
    hipLaunchKernelGGL(scatter_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, stream, nx, ny, tiles_x, tiles_y, world,
                       tiles_max, reinterpret_cast<const float4*>(gathered), reinterpret_cast<float4*>(fb));
}

} // namespace crt
