/*
 * crt_oracle.h -- entry points of the CPU restatement (test infrastructure only;
 * see the header of crt_oracle.c).  Scene and settings structs are the ones of
 * the product's C ABI so that both sides consume identical inputs.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H
#include <stdint.h>

#include "../include/crt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_counters {
    uint64_t rays;         /* closest-hit queries issued by color() */
    uint64_t shade_points; /* diffuse hits shaded */
    uint64_t gi_children;  /* (r1,r2) pairs drawn */
    uint64_t rng_draws;    /* curand_uniform calls */
} orc_counters;

typedef struct orc_hit_record {
    int hit;
    int geom;
    float t;
    float p[3];
    float n[3];
} orc_hit_record;

/* RNG state layout everywhere: {v0,v1,v2,v3,v4,d}. */
void orc_rng_init(uint64_t seed, uint32_t st[6]);
uint32_t orc_rng_next(uint32_t st[6]);
float orc_rng_uniform(uint32_t st[6]);
void orc_sincos(float a, float* s, float* c);

void orc_cornell_scene(uint32_t shadowSamples, crt_camera* cam, crt_geometry* geometry8, crt_light* light1);

/* One pixel of render() (main.cu:328-347). rng is advanced in place. */
void orc_render_pixel(const crt_scene_desc* sc, const crt_settings* st, const float* ltc1, const float* ltc2,
                      uint32_t i, uint32_t j, uint32_t rng[6], float out[4], orc_counters* cnt);
/* Pixels [x0,x1) x [y0,y1) of an nx*ny image into fb (full-size, row-major RGBA).
 * fresh_rng != 0: seed every pixel with seed + idx first; otherwise continue from
 * rng_inout6 (nx*ny*6 words).  rng_inout6 may be NULL.  nthreads <= 0: OpenMP default. */
int orc_render(const crt_scene_desc* sc, const crt_settings* st, uint64_t seed, const float* ltc_tables,
               float* fb, uint32_t* rng_inout6, int fresh_rng, uint32_t x0, uint32_t y0, uint32_t x1,
               uint32_t y1, int nthreads, orc_counters* total);

/* Canonical stand-in for __powf(x, y), x >= 0 (the point light's specular term, Light.hpp:173). */
float orc_pow(float x, float y);

/* The horizon-clipped LTC integral (AreaLight.hpp:104-214,298-334) of four transformed corners. */
float orc_ltc_clipped(const float corners[4][3], int two_sided);

/* Texture::_imageSpaceCoordinates (Texture.hpp:126-185): texel of (u, v); wrap 0 CLAMP, 1 MIRROR, 2 REPEAT. */
void orc_texture_texel(float u, float v, int wrap_h, int wrap_v, uint32_t width, uint32_t height, uint32_t xy[2]);

/* Function-level probes. */
int orc_camera_ray(const crt_camera* cam, float x, float y, float w, float h, float o[3], float d[3]);
int orc_world_hit(const crt_scene_desc* sc, const float o[3], const float d[3], float maxdist, orc_hit_record* out);
int orc_shade(const crt_scene_desc* sc, const float* ltc_tables, const float p[3], const float n[3],
              const float view[3], uint32_t geom, uint32_t rng[6], float out[4]);
int orc_color(const crt_scene_desc* sc, const crt_settings* st, const float* ltc_tables, const float o[3],
              const float d[3], uint32_t rng[6], float out[4]);

#ifdef __cplusplus
}
#endif
#endif
