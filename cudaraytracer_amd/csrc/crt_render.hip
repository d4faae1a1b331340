// Render kernels for gfx950: random_init and the per-pixel integrator
// (render -> aa_color -> color -> GeometryList::hit / AreaLight::calculateColor,
// main.cu:110-347).
//
// Execution shape.  A workgroup is one wave.  The grid is persistent: every wave
// pulls work units (8x8 pixel tiles) from a global counter, and every *lane*
// pulls the next pixel of its wave's current unit as soon as it has finished
// its previous pixel, so lanes stay busy although pixels cost very different
// amounts (an escaping ray costs nothing, a diffuse hit costs shadowSamples
// shadow rays and two LTC integrals, times ind_rays^depth).
//
// The reference's two explicit stacks (RayState[128] in color(), AA_Task[256]
// in aa_color()) are replaced by O(depth) per-lane state:
//   * a pending mirror/glass ray (a Reflect/Refractive hit pushes one ray that
//     is popped at once),
//   * one Frame per nested global-illumination level.  A diffuse hit at depth>0
//     draws its ind_rays (r1,r2) pairs exactly as the reference does, but keeps
//     only the generator state reached after the last *accepted* child;
//     children are popped last-to-first by stepping that state backwards
//     (XORWOW's xorshift is a bijection), so no child is ever stored.  The
//     reference's 128-entry truncation is a function of how many entries are on
//     its stack, which here is the sum of the frames' remaining-children counters.
//   * a 10-entry adaptive-AA task list (aa_iter <= 4 bounds it: 1 + 3*3).
//
// Each wave alternates between phase A, in which lanes step rays (camera rays,
// GI children, mirror/glass bounces) until each owns a diffuse hit, and phase B,
// in which all lanes shade their hit together: shadowSamples shadow rays, then
// two LTC polygon integrals.  The shadow loop is two-stage.  Stage 1 runs per
// lane and per sample and only *filters*: each primitive (scalar registers) is
// tested with the conservative, division-free tests of crt_device.hpp; a ray
// that certainly misses everything is done.  Every other ray is appended to a
// queue in LDS.  Stage 2 drains that queue 64 rays at a time with all lanes busy
// and evaluates the reference's exact closest-hit test; rays whose closest hit is
// glass go to a second queue whose batches run the refraction walk of
// Light::_isCovered.  Results come back through per-lane counters in LDS.  The
// rare, long, divergent paths therefore never idle the wave.
#include <hip/hip_runtime.h>

#include "crt_device.hpp"
#include "crt_render.hpp"
#include "crt_shade.hpp"

namespace crt {

namespace {

constexpr uint32_t kRayQueueCap = 192;   // < 64 left over + 2*64 new rays per sample round
constexpr uint32_t kGlassQueueCap = 128; // < 64 left over + 64 new entries per resolved batch

struct Frame {
    float P[3];      // hit point the children leave from
    float N[3];      // its shading normal
    float att[4];    // attenuation of the expanding ray (RayState::attenuation)
    uint32_t rng[6]; // generator state right after the draws of child `remaining-1`
    uint32_t geom;   // primitive hit (for Material::diffuse)
    uint32_t remaining;   // children still on the reference's stack
    uint32_t child_depth; // global_depth of the children
};

struct AATask {
    float cx, cy;
    uint32_t level; // AA_Task::sample
};

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

#ifndef CRT_WAVES_PER_EU
#define CRT_WAVES_PER_EU 1
#endif
template <bool CLIP>
__global__ void __launch_bounds__(64, CRT_WAVES_PER_EU) render_kernel(RenderArgs a) {
    ScenePtr sc = (ScenePtr)a.scene;
    const float* __restrict__ ltc = a.ltc;
    const uint32_t lane = threadIdx.x;
    float4* __restrict__ out = reinterpret_cast<float4*>(a.out);
    const uint32_t npix = a.nx * a.ny;

    // stage-2 queues: shadow rays that may hit something, and rays whose closest hit is glass
    __shared__ float rq_f[7][kRayQueueCap];     // origin, direction, max distance
    __shared__ uint32_t rq_owner[kRayQueueCap]; // lane that owns the sample
    __shared__ float gq_f[9][kGlassQueueCap];   // origin, direction, first hit (t,u,v)
    __shared__ uint32_t gq_meta[kGlassQueueCap]; // primitive | owner << 16
    __shared__ uint32_t s_shadow[64];           // covered samples found by stage 2, per owner lane

    // get_world_coordinates (main.cu:306-314) and pixel_size (main.cu:337): image-wide constants
    const float height_world = 2.0f;
    const float width_world = height_world * ((float)a.nx / (float)a.ny);
    const float pix_w = width_world / (float)a.nx, pix_h = height_world / (float)a.ny;

    // Camera::getRay constants (Camera.hpp:97-98,133-134)
    const V3 cam_pos = ld3c(sc->cam.pos), cam_front = ld3c(sc->cam.front), cam_right = ld3c(sc->cam.right),
             cam_up = ld3c(sc->cam.up);
    const V3 eye = cam_pos - cam_front * (width_world * 0.5f * sc->cam.one_over_tan);
    const uint32_t cam_type = sc->cam.type;
    float ortho;
    {
        float aspect = height_world / width_world;
        ortho = sc->cam.ortho_scale * (aspect > 1.0f ? aspect : 1.0f);
    }
    const uint32_t aa_iter = a.aa_iter > 4u ? 4u : a.aa_iter; // main.cu:233
    const float inv_ind = 1.0f / (float)a.ind_rays;            // main.cu:182

    // ---- wave-uniform work pool
    uint32_t unit = blockIdx.x;
    bool unit_valid = unit < a.n_units;
    bool exhausted = !unit_valid;
    uint32_t next = 0;

    // ---- per-lane pixel state
    bool pixel_active = false, lane_dead = false;
    uint32_t pix = 0, out_index = 0;
    Rng rng = {0, 0, 0, 0, 0, 0};
    uint32_t cnt_shade = 0, cnt_expand = 0;
    uint32_t w_stage2 = 0, w_glass = 0; // wave-uniform tallies of queued rays

    // adaptive-AA task list (aa_color, main.cu:230-304)
    AATask aa_stack[10];
    uint32_t aa_top = 0;
    V4 acc = V4{0, 0, 0, 0};
    bool task_active = false;
    float t_cx = 0, t_cy = 0;
    uint32_t t_level = 0, t_corner = 0;
    bool t_all_equal = true;
    V4 t_c0 = V4{0, 0, 0, 0};

    // color() state (main.cu:110-227)
    Frame frames[CRT_MAX_GL_ITER];
    uint32_t nframes = 0;
    uint32_t on_stack = 0; // entries the reference's stack would hold
    bool color_active = false;
    V4 final_c = V4{0, 0, 0, 0};
    V3 ray_d = mk3(0.0f, 0.0f, -1.0f); // Ray::_direction survives a degenerate setDirection (Ray.hpp:62-69)
    bool cur_valid = false;            // pending ray (popped next)
    V3 cur_o = mk3(0, 0, 0), cur_d = mk3(0, 0, 0);
    V4 cur_att = V4{1, 1, 1, 1};
    uint32_t cur_ref = 0, cur_depth = 0;

    // diffuse hit waiting to be shaded
    bool have_node = false;
    V3 n_P = mk3(0, 0, 0), n_N = mk3(0, 0, 0), n_view = mk3(0, 0, 0);
    V4 n_att = V4{0, 0, 0, 0};
    uint32_t n_geom = 0, n_depth = 0;

    for (;;) {
        // ===================== phase A: step rays until every live lane owns a diffuse hit
        for (;;) {
            // ---- (1) hand pixels to lanes that have none
            for (;;) {
                const bool want = !pixel_active && !lane_dead;
                const uint64_t m = __ballot(want);
                if (m == 0) break;
                if (unit_valid && next < 64u) {
                    const uint32_t idx = next + lane_rank(m);
                    next += (uint32_t)__popcll(m);
                    if (want && idx < 64u) {
                        const PixelMap pm = map_pixel(a, unit, idx);
                        if (pm.valid) {
                            pixel_active = true;
                            pix = pm.j * a.nx + pm.i;
                            out_index = pm.out_index;
                            rng.v0 = a.rng[0 * (size_t)npix + pix];
                            rng.v1 = a.rng[1 * (size_t)npix + pix];
                            rng.v2 = a.rng[2 * (size_t)npix + pix];
                            rng.v3 = a.rng[3 * (size_t)npix + pix];
                            rng.v4 = a.rng[4 * (size_t)npix + pix];
                            rng.d = a.rng[5 * (size_t)npix + pix];
                            // remap<float,true> (math_functions.hpp:31-45)
                            float denom = (float)a.nx - 0.0f;
                            float es = -width_world * 0.5f, ee = width_world * 0.5f;
                            float x_world = (denom == 0.0f) ? (float)pm.i : es + ((ee - es) / denom) * ((float)pm.i - 0.0f);
                            float denomy = (float)a.ny - 0.0f;
                            float y_world = (denomy == 0.0f) ? (float)pm.j
                                                             : 1.0f + ((-1.0f - 1.0f) / denomy) * ((float)pm.j - 0.0f);
                            aa_top = 0;
                            aa_stack[aa_top++] = AATask{x_world, y_world, aa_iter};
                            acc = V4{0, 0, 0, 0};
                            task_active = false;
                            color_active = false;
                            cur_valid = false;
                            nframes = 0;
                            cnt_shade = 0;
                            cnt_expand = 0;
                        }
                    }
                } else if (!exhausted) {
                    uint32_t u = 0;
                    if (lane == 0) u = gridDim.x + atomicAdd(a.unit_counter, 1u);
                    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
                    if (u < a.n_units) {
                        unit = u;
                        next = 0;
                        unit_valid = true;
                    } else {
                        exhausted = true;
                        unit_valid = false;
                    }
                } else {
                    if (want) lane_dead = true;
                }
            }

            // ---- (2) one ray step for every lane that still looks for a diffuse hit
            const bool need = pixel_active && !have_node;
            if (!__any(need)) break;
            if (need) {
                if (!cur_valid) {
                    if (!color_active) {
                        // aa_color: next camera ray of the current task, or finish the pixel
                        if (!task_active && aa_top == 0) {
                            // render() epilogue (main.cu:339-346)
                            out[out_index] = make_float4(saturate_c(acc.x), saturate_c(acc.y), saturate_c(acc.z),
                                                         saturate_c(acc.w));
                            a.rng[0 * (size_t)npix + pix] = rng.v0;
                            a.rng[1 * (size_t)npix + pix] = rng.v1;
                            a.rng[2 * (size_t)npix + pix] = rng.v2;
                            a.rng[3 * (size_t)npix + pix] = rng.v3;
                            a.rng[4 * (size_t)npix + pix] = rng.v4;
                            a.rng[5 * (size_t)npix + pix] = rng.d;
                            a.counters[pix] += cnt_shade;
                            a.counters[npix + pix] += cnt_expand;
                            pixel_active = false;
                        } else {
                            if (!task_active) {
                                AATask t = aa_stack[--aa_top];
                                t_cx = t.cx;
                                t_cy = t.cy;
                                t_level = t.level;
                                t_corner = 0;
                                t_all_equal = true;
                                task_active = true;
                            }
                            // box of this task: pixel size halved (aa_iter - level) times (main.cu:251,297)
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
                            if (cam_type == CRT_CAM_ORTHOGRAPHIC) {
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
                    } else if (nframes == 0) {
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
                                        aa_stack[aa_top++] =
                                            AATask{t_cx + ox * hx * 0.5f, t_cy + oy * hy * 0.5f, t_level - 1u};
                                    }
                                }
                            }
                        }
                    } else {
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
                        V4 pdiff = ld4c(sc->shade[f.geom].diffuse);
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
                        V3 w = mk3(x * Nz.x + r1 * hn.x + z * Nx.x, x * Nz.y + r1 * hn.y + z * Nx.y,
                                   x * Nz.z + r1 * hn.z + z * Nx.z);
                        cur_o = hp + w * 0.01f;
                        cur_d = w;
                        cur_att = ((pdiff * (2.0f * r1)) * patt) * inv_ind; // main.cu:186
                        cur_ref = a.ref_iter;
                        cur_depth = cdepth;
                        cur_valid = true;
                    }
                }
                if (cur_valid) {
                    // ---- pop (main.cu:132-147)
                    cur_valid = false;
                    if (cur_ref != 0u) {
                        if (!eps_equal(len_sq(cur_d), 0.0f, kEpsSq)) ray_d = normalize3(cur_d);
                        const V3 ro = cur_o;
                        Hit h = closest_hit(sc, ro, ray_d, 0.0f);
                        if (h.idx < 0) {
                            final_c = final_c + cur_att * V4{0.0f, 0.0f, 0.0f, 1.0f}; // Color::black()
                        } else {
                            V3 P, N;
                            hit_point_normal(sc, h, ro, ray_d, P, N);
                            if (h.mat_type == CRT_MAT_DIFFUSE) {
                                have_node = true;
                                n_P = P;
                                n_N = N;
                                n_view = neg(cur_d); // viewDir = -state.r_dir, the stored (un-normalised) direction
                                n_att = cur_att;
                                n_geom = (uint32_t)h.idx;
                                n_depth = cur_depth;
                            } else if (h.mat_type == CRT_MAT_REFLECT) {
                                // get_reflect_ray (main.cu:76-81)
                                V3 I = normalize3(ray_d);
                                V3 rdir = reflect3(I, normalize3(N));
                                cur_o = P + rdir * 0.01f;
                                cur_d = rdir;
                                cur_att = cur_att * ld4c(sc->shade[h.idx].diffuse);
                                cur_ref = cur_ref - 1u;
                                cur_valid = true;
                            } else {
                                // get_refraction_ray (main.cu:83-108)
                                bool front_face = dot3(ray_d, N) < 0.0f;
                                V3 norm = front_face ? N : neg(N);
                                float ratio = sc->shade[h.idx].refract_index / 1.0f;
                                if (front_face) ratio = rcp_exact(ratio);
                                V3 rdn = normalize3(ray_d);
                                float cos_theta = __builtin_fminf(dot3(neg(rdn), norm), 1.0f);
                                float sin_theta = rcp_exact(rsqrt_c(1.0f - cos_theta * cos_theta));
                                bool cannot_refract = ratio * sin_theta > 1.0f || reflectance_over_one(cos_theta, ratio); // main.cu:97
                                V3 dir = cannot_refract ? reflect3(rdn, norm) : refract3(rdn, norm, ratio);
                                cur_o = P + dir * 0.01f;
                                cur_d = dir;
                                cur_att = cur_att * ld4c(sc->shade[h.idx].diffuse);
                                cur_ref = cur_ref - 1u;
                                cur_valid = true;
                            }
                        }
                    }
                }
            }
        }
        if (!__any(have_node)) break; // every lane is out of pixels

        // ===================== phase B: shade the diffuse hits (main.cu:153-195)
        const bool act = have_node;
        if (act) ++cnt_shade;
        const DevGeomShade CRT_CONST* gs = &sc->shade[act ? n_geom : 0u];
        V4 direct = V4{0, 0, 0, 0}; // LightList::calculateColor (LightList.hpp:23-36)
        const uint32_t n_lights = sc->n_lights;
        for (uint32_t li = 0; li < n_lights; ++li) {
            const DevLight CRT_CONST* L = &sc->lights[li];
            if (L->kind != CRT_LIGHT_AREA) {
                if (act) direct = direct + point_light_color(sc, L, gs, n_P, n_N, n_view);
                continue;
            }
            if (!L->initialized) { // AreaLight.hpp:586
                if (act) {
                    V3 amb = mk3(gs->ambient[0], gs->ambient[1], gs->ambient[2]) * ld3c(L->color);
                    direct = direct + V4{amb.x, amb.y, amb.z, 1.0f};
                }
                continue;
            }
            // ---- _calculateVisibility (AreaLight.hpp:484-529)
            const uint32_t S = L->shadow_samples;
            const uint32_t n_geom_all = sc->n_geom;
            const V3 lpos = ld3c(L->pos);
            const V3 lp0 = ld3c(L->p0), lp2 = ld3c(L->p2);
            const V3 le01 = ld3c(L->e01), le03 = ld3c(L->e03), le23 = ld3c(L->e23), le21 = ld3c(L->e21);
            uint32_t shadowed = 0; // stays 0 here: every covered sample is counted by stage 2
            uint32_t rq = 0, gq = 0; // queue fills, wave-uniform
            s_shadow[lane] = 0u;
            for (uint32_t s = 0; s < S; s += 2u) {
                const bool two = s + 1u < S; // second sample of this round exists (wave-uniform)
                bool may0 = false, may1 = false;
                V3 o0 = mk3(0, 0, 0), d0 = mk3(0, 0, 0), o1 = mk3(0, 0, 0), d1 = mk3(0, 0, 0);
                float x0 = 0.0f, x1 = 0.0f;
                if (act) {
                    // ---- stage 1: draw, build the shadow rays, filter against every primitive
                    {
                        float u = rng_uniform(rng);
                        float v = rng_uniform(rng);
                        V3 sample; // _randomPoint (AreaLight.hpp:340-375)
                        if (u + v < 1.0f) {
                            sample = ((lp0 + le01 * u) + le03 * v) + lpos;
                        } else {
                            u = 1.0f - u;
                            v = 1.0f - v;
                            sample = ((lp2 + le23 * u) + le21 * v) + lpos;
                        }
                        V3 dir = sample - n_P;
                        float dist;
                        V3 ndir = normalize3_len(dir, dist);
                        if (!eps_equal(dist, 0.0f, kEps)) dir = ndir;
                        o0 = n_P + dir * 0.01f;
                        d0 = ray_dir(dir);            // the Ray ctor normalises once more (Ray.hpp:45-52)
                        x0 = max_std(0.0f, dist);     // Ray::setDistance
                    }
                    if (two) {
                        float u = rng_uniform(rng);
                        float v = rng_uniform(rng);
                        V3 sample;
                        if (u + v < 1.0f) {
                            sample = ((lp0 + le01 * u) + le03 * v) + lpos;
                        } else {
                            u = 1.0f - u;
                            v = 1.0f - v;
                            sample = ((lp2 + le23 * u) + le21 * v) + lpos;
                        }
                        V3 dir = sample - n_P;
                        float dist;
                        V3 ndir = normalize3_len(dir, dist);
                        if (!eps_equal(dist, 0.0f, kEps)) dir = ndir;
                        o1 = n_P + dir * 0.01f;
                        d1 = ray_dir(dir);
                        x1 = max_std(0.0f, dist);
                    }
                    const float dd0 = dot3(d0, d0), dd1 = dot3(d1, d1);
                    const float dd40 = 4.0f * dd0, dd41 = 4.0f * dd1; // 4.f * a (Sphere.hpp:61)
                    const bool ddok0 = dd0 > 1e-6f && dd0 < 1e6f, ddok1 = dd1 > 1e-6f && dd1 < 1e6f;
                    const bool xok0 = x0 > 1e-30f && x0 < 1e30f, xok1 = x1 > 1e-30f && x1 < 1e30f;
                    // the record of primitive i+1 is requested before primitive i is tested
                    const DevGeomHot CRT_CONST* g = &sc->hot[0];
                    uint32_t kind = g->kind;
                    V3 go = ld3c(g->o), gn = ld3c(g->n);
                    for (uint32_t i = 0; i < n_geom_all; ++i) {
                        const uint32_t ckind = kind;
                        const V3 cgo = go, cgn = gn;
                        if (i + 1u < n_geom_all) {
                            g = &sc->hot[i + 1u];
                            kind = g->kind;
                            go = ld3c(g->o);
                            gn = ld3c(g->n);
                        }
                        bool h0, h1;
                        if (ckind == CRT_GEOM_SPHERE) {
                            h0 = sphere_may_hit(cgo, cgn.x, o0, d0, dd40, ddok0);
                            h1 = sphere_may_hit(cgo, cgn.x, o1, d1, dd41, ddok1);
                        } else if (ckind == CRT_HOT_QUAD_X) {
                            h0 = plane_may_hit((o0.x - cgo.x) * cgn.x, d0.x * cgn.x, x0, xok0);
                            h1 = plane_may_hit((o1.x - cgo.x) * cgn.x, d1.x * cgn.x, x1, xok1);
                        } else if (ckind == CRT_HOT_QUAD_Y) {
                            h0 = plane_may_hit((o0.y - cgo.y) * cgn.y, d0.y * cgn.y, x0, xok0);
                            h1 = plane_may_hit((o1.y - cgo.y) * cgn.y, d1.y * cgn.y, x1, xok1);
                        } else if (ckind == CRT_HOT_QUAD_Z) {
                            h0 = plane_may_hit((o0.z - cgo.z) * cgn.z, d0.z * cgn.z, x0, xok0);
                            h1 = plane_may_hit((o1.z - cgo.z) * cgn.z, d1.z * cgn.z, x1, xok1);
                        } else {
                            h0 = quad_may_hit(o0 - cgo, d0, cgn, x0, xok0);
                            h1 = quad_may_hit(o1 - cgo, d1, cgn, x1, xok1);
                        }
                        may0 = may0 | h0;
                        may1 = may1 | h1;
                    }
                    may1 = may1 && two;
                }
                // ---- append the surviving rays to the stage-2 queue
                const uint64_t m0 = __ballot(may0);
                const uint64_t m1 = __ballot(may1);
                if ((m0 | m1) != 0) {
                    if (may0) {
                        const uint32_t slot = rq + lane_rank(m0);
                        rq_f[0][slot] = o0.x;
                        rq_f[1][slot] = o0.y;
                        rq_f[2][slot] = o0.z;
                        rq_f[3][slot] = d0.x;
                        rq_f[4][slot] = d0.y;
                        rq_f[5][slot] = d0.z;
                        rq_f[6][slot] = x0;
                        rq_owner[slot] = lane;
                    }
                    rq += (uint32_t)__popcll(m0);
                    if (may1) {
                        const uint32_t slot = rq + lane_rank(m1);
                        rq_f[0][slot] = o1.x;
                        rq_f[1][slot] = o1.y;
                        rq_f[2][slot] = o1.z;
                        rq_f[3][slot] = d1.x;
                        rq_f[4][slot] = d1.y;
                        rq_f[5][slot] = d1.z;
                        rq_f[6][slot] = x1;
                        rq_owner[slot] = lane;
                    }
                    rq += (uint32_t)__popcll(m1);
                    w_stage2 += (uint32_t)(__popcll(m0) + __popcll(m1));
                }
                // ---- stage 2: full batches during the loop, everything after the last round
                const bool last = s + 2u >= S;
                while (rq >= 64u || gq >= 64u || (last && (rq | gq) != 0u)) {
                    __syncthreads();
                    if (gq < 64u && (rq >= 64u || (last && rq != 0u))) { // a full glass batch goes first: its queue must stay below 64
                        // exact closest hit of Light::_isCovered's first pass (Light.hpp:41-46)
                        const uint32_t n = rq < 64u ? rq : 64u;
                        const uint32_t base = rq - n;
                        bool glass = false;
                        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
                        Hit h;
                        h.idx = -1;
                        h.mat_type = 0;
                        h.t = h.u = h.v = 0.0f;
                        uint32_t owner = 0;
                        if (lane < n) {
                            const uint32_t slot = base + lane;
                            o = mk3(rq_f[0][slot], rq_f[1][slot], rq_f[2][slot]);
                            d = mk3(rq_f[3][slot], rq_f[4][slot], rq_f[5][slot]);
                            owner = rq_owner[slot];
                            h = closest_hit(sc, o, d, rq_f[6][slot]);
                            if (h.idx >= 0) {
                                if (h.mat_type != CRT_MAT_REFRACTIVE) atomicAdd(&s_shadow[owner], 1u);
                                else glass = true;
                            }
                        }
                        rq = base;
                        const uint64_t gm = __ballot(glass);
                        if (gm != 0) {
                            if (glass) {
                                const uint32_t slot = gq + lane_rank(gm);
                                gq_f[0][slot] = o.x;
                                gq_f[1][slot] = o.y;
                                gq_f[2][slot] = o.z;
                                gq_f[3][slot] = d.x;
                                gq_f[4][slot] = d.y;
                                gq_f[5][slot] = d.z;
                                gq_f[6][slot] = h.t;
                                gq_f[7][slot] = h.u;
                                gq_f[8][slot] = h.v;
                                gq_meta[slot] = (uint32_t)h.idx | (owner << 16);
                            }
                            gq += (uint32_t)__popcll(gm);
                            w_glass += (uint32_t)__popcll(gm);
                        }
                    } else {
                        // refraction walk of Light::_isCovered (Light.hpp:47-80)
                        const uint32_t n = gq < 64u ? gq : 64u;
                        const uint32_t base = gq - n;
                        if (lane < n) {
                            const uint32_t slot = base + lane;
                            V3 o = mk3(gq_f[0][slot], gq_f[1][slot], gq_f[2][slot]);
                            V3 d = mk3(gq_f[3][slot], gq_f[4][slot], gq_f[5][slot]);
                            Hit qh;
                            const uint32_t meta = gq_meta[slot];
                            qh.idx = (int)(meta & 0xffffu);
                            qh.mat_type = CRT_MAT_REFRACTIVE;
                            qh.t = gq_f[6][slot];
                            qh.u = gq_f[7][slot];
                            qh.v = gq_f[8][slot];
                            if (covered_through_glass(sc, L, o, d, qh)) atomicAdd(&s_shadow[meta >> 16], 1u);
                        }
                        gq = base;
                    }
                    __syncthreads();
                }
            }
            if (act) {
                shadowed += s_shadow[lane];
                float visibility = 1.0f - ((float)shadowed / (float)S);
                direct = direct + area_light_shade<CLIP>(ltc, L, gs, n_P, n_N, n_view, visibility);
            }
            __syncthreads(); // s_shadow is reset at the top of the next light
        }
        if (act) {
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
            have_node = false;
        }
    }
    if (lane == 0 && a.wave_tallies) {
        atomicAdd(&a.wave_tallies[0], (unsigned long long)w_stage2);
        atomicAdd(&a.wave_tallies[1], (unsigned long long)w_glass);
    }
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

// Exhaustive check of the fast correctly-rounded helpers against the IEEE results over all 2^32 bit
// patterns: rcp_exact vs 1.0f/x everywhere; rcp_nr vs 1.0f/x and sqrt_nr vs sqrtf on [2^-100, 2^100]; there
// also the identity normalize3 rests on, 1/(1/(1/x)) == 1/x; and div_nr against the IEEE quotient for every
// argument integrate_edge_vec can hand it (the two polynomials of y = |x| in [0, 1.00001]).
__global__ void __launch_bounds__(256) selftest_rcp_kernel(unsigned long long* __restrict__ mismatches,
                                                           uint32_t* __restrict__ first_bad) {
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        float x = __uint_as_float((uint32_t)b);
        float r = rcp_exact(x);
        float q = 1.0f / x;
        bool same = (__float_as_uint(r) == __float_as_uint(q)) || (r != r && q != q);
        float ax = __builtin_fabsf(x);
        if (ax >= 0x1p-100f && ax <= 0x1p100f) {
            same = same && (__float_as_uint(rcp_nr(x)) == __float_as_uint(q));
            if (x > 0.0f) same = same && (__float_as_uint(sqrt_nr(x)) == __float_as_uint(__builtin_sqrtf(x)));
            same = same && (__float_as_uint(1.0f / (1.0f / q)) == __float_as_uint(q));
        }
        if (x >= 0.0f && x <= 1.00001f) {
            const float pa = 0.8543985f + (0.4965155f + 0.0145206f * x) * x;
            const float pb = 3.4175940f + (4.1616724f + x) * x;
            const F2 dq = div_nr(F2{pa, pa}, F2{pb, pb});
            same = same && (__float_as_uint(dq.x) == __float_as_uint(pa / pb)) && (__float_as_uint(dq.y) == __float_as_uint(pa / pb));
        }
        if (!same) {
            ++bad;
            atomicMin(first_bad, (uint32_t)b);
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

void launch_random_init(uint32_t npix, uint64_t seed, uint32_t* state, hipStream_t stream) {
    hipLaunchKernelGGL(random_init_kernel, dim3((npix + 63u) / 64u), dim3(64), 0, stream, npix, seed, state);
}
// Persistent grid: as many one-wave workgroups as the device keeps resident, never more than there are units.
static int g_resident_waves[64] = {0};
hipError_t prime_render_grid() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (g_resident_waves[dev] == 0) {
        int per_cu = 0, cus = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel<false>, 64, 0);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 16) per_cu = 16; // 4 waves per SIMD saturate VALU issue
        g_resident_waves[dev] = per_cu * cus;
    }
    return hipSuccess;
}
hipError_t launch_render(const RenderArgs& args, hipStream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (g_resident_waves[dev] == 0 && (e = prime_render_grid()) != hipSuccess) return e;
    if (args.n_units == 0) return hipSuccess;
    uint32_t grid = args.n_units < (uint32_t)g_resident_waves[dev] ? args.n_units : (uint32_t)g_resident_waves[dev];
    e = hipMemsetAsync(args.unit_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (args.horizon_clip) hipLaunchKernelGGL(render_kernel<true>, dim3(grid), dim3(64), 0, stream, args);
    else hipLaunchKernelGGL(render_kernel<false>, dim3(grid), dim3(64), 0, stream, args);
    return hipGetLastError();
}
void launch_scatter(uint32_t nx, uint32_t ny, uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_max,
                    const float* gathered, float* fb, hipStream_t stream) {
    hipLaunchKernelGGL(scatter_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, stream, nx, ny, tiles_x, tiles_y, world,
                       tiles_max, reinterpret_cast<const float4*>(gathered), reinterpret_cast<float4*>(fb));
}
void launch_selftest_rcp(unsigned long long* mismatches, uint32_t* first_bad, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_rcp_kernel, dim3(4096), dim3(256), 0, stream, mismatches, first_bad);
}

} // namespace crt
