/*
 * crt_hip.h -- C ABI of libcrt_hip.so: the MI355X (gfx950) implementation of
 * CudaRayTracer's per-pixel render path.
 *
 * The reference has no FFI layer; its host/device boundary is the set of kernel
 * launches made from main() (main.cu:614-894).  Each entry point below replaces
 * one of those launches (or the host helper wrapped around it) and cites it.
 * Everything is plain C: POD structs, raw pointers, sizes, integer status codes.
 * No function here terminates the process (the reference's checkCuda() calls
 * exit(99), main.cu:896-903); errors come back as negative codes and a message
 * retrievable with crt_last_error().
 *
 * Threading: a crt_ctx belongs to one host thread and one GPU.  Work is queued
 * on the HIP stream handed to each call (NULL = the default stream); calls do
 * not synchronise unless their comment says so.  The launches of one context
 * (crt_rng_init and the crt_render* calls) share its RNG states and work
 * counters, so the library runs them one after the other: a call on another
 * stream than the context's previous launch is queued behind that launch (an
 * event per context).  Buffers the caller owns are the caller's to order.
 * A large launch (1 000 units of 64 pixels or more) is rendered as two pieces
 * side by side on two streams of the context, by two host threads for the
 * duration of the call; the caller's stream waits for both.
 *
 * Memory: "dev" pointers are device (HBM) addresses owned by the caller;
 * "host" pointers are ordinary host memory.  The context owns the flattened
 * scene, the LTC tables and the per-pixel RNG state.
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define CRT_OK 0
#define CRT_ERR_INVALID (-1)     /* bad argument (NULL, zero size, out of range) */
#define CRT_ERR_HIP (-2)         /* a HIP runtime call failed (see crt_last_error) */
#define CRT_ERR_NO_SCENE (-3)    /* render before crt_scene_upload */
#define CRT_ERR_NO_RNG (-4)      /* render before crt_rng_init, or size mismatch */
#define CRT_ERR_UNSUPPORTED (-5) /* setting outside the implemented range */
#define CRT_ERR_IO (-6)          /* file could not be written */
#define CRT_ERR_TRUNCATED (-7)   /* the frame was stopped by crt_set_round_limit before it was finished */

/* ---- limits ------------------------------------------------------------- */
#define CRT_MAX_GEOMETRY 64u /* primitives per GeometryList */
#define CRT_MAX_LIGHTS 8u    /* lights per LightList */
#define CRT_MAX_GL_ITER 8u   /* nested global-illumination levels held per lane */
#define CRT_MAX_AA_ITER 4u   /* aa_color caps aa_iter at 4 (main.cu:233) */
#define CRT_RAY_STACK 128u   /* MAX_STACK of color() (main.cu:111); drives child truncation */

/* ---- scene description (flattened scene classes) ------------------------ */
/* MaterialType, Material.hpp:18 */
#define CRT_MAT_DIFFUSE 0u
#define CRT_MAT_REFLECT 1u
#define CRT_MAT_REFRACTIVE 2u

/* Material, Material.hpp:20-27 */
typedef struct crt_material {
    uint32_t type;
    float ambient[4];
    float diffuse[4];
    float specular[4];
    float shininess;
    float refractIndex;
} crt_material;

/* Vertex, Vertex.hpp:14-18 */
typedef struct crt_vertex {
    float pos[3];
    float tex[2];
    float normal[3];
} crt_vertex;

#define CRT_GEOM_SPHERE 0u /* Sphere.hpp:18-31 */
#define CRT_GEOM_QUAD 1u   /* Quad.hpp:21-27 */

/* One entry of the GeometryList (GeometryList.hpp:14-20); list order is kept
 * because a distance tie goes to the later primitive (GeometryList.hpp:28). */
typedef struct crt_geometry {
    uint32_t kind;
    crt_material mat; /* Geometry::_mat, Geometry.hpp:18 */
    float center[3];  /* sphere */
    float radius;     /* sphere */
    crt_vertex v[4];  /* quad: _v0.._v3 */
} crt_geometry;

#define CRT_LIGHT_POINT 0u /* Light, Light.hpp:29-33,116 */
#define CRT_LIGHT_AREA 1u  /* AreaLight, AreaLight.hpp:27-44,534-556 */

/* One entry of the LightList (LightList.hpp:14-20).  For an area light
 * `points` are the four corners relative to `pos` in the order AreaLight keeps
 * them (after the CCW reorder, centroid subtraction and any rotate()). */
typedef struct crt_light {
    uint32_t kind;
    float pos[3];
    float color[4];
    float intensity;
    float points[4][3];
    uint32_t twoSided;
    uint32_t shadowSamples;
    uint32_t initialized; /* AreaLight::_initialized (AreaLight.hpp:36,488,586): 0 -> ambient term only */
    uint32_t horizonClip; /* !AreaLight::_clipless (AreaLight.hpp:42,245,298): 0 -> the clipless form-factor branch,
                             which is what the reference always runs (_clipless is true and has no setter);
                             1 -> clip the transformed quad to the horizon and sum the edge integrals
                             (AreaLight.hpp:104-214,298-334) */
} crt_light;

#define CRT_CAM_ORTHOGRAPHIC 0u /* CameraType, Camera.hpp:19-22 */
#define CRT_CAM_PERSPECTIVE 1u

/* Camera, Camera.hpp:24-31 (front already normalised by the constructor) */
typedef struct crt_camera {
    float pos[3];
    float front[3];
    float fov;
    float orthoScale;
    uint32_t type;
} crt_camera;

typedef struct crt_scene_desc {
    crt_camera camera;
    uint32_t n_geometry;
    const crt_geometry* geometry;
    uint32_t n_lights;
    const crt_light* lights;
} crt_scene_desc;

/* The render settings of main() (main.cu:626-636), same names. */
typedef struct crt_settings {
    uint32_t renderAllAtOnce;
    uint32_t blocksPerDraw;
    uint32_t nx, ny;
    uint32_t tx, ty; /* block shape of the reference launch; only render_partial's
                        window arithmetic depends on it here */
    uint32_t aa_iter;
    uint32_t ref_iter;
    uint32_t gl_iter;
    uint32_t ind_rays;
    uint32_t shadowSamples; /* consumed by crt_cornell_scene, not by the renderer */
} crt_settings;

/* Work counters of one render (exact, derived from the RNG draw counters). */
typedef struct crt_stats {
    uint64_t rng_draws;      /* total curand_uniform draws */
    uint64_t shadow_samples; /* _calculateVisibility samples */
    uint64_t shade_points;   /* AreaLight::calculateColor calls */
    uint64_t gi_children;    /* (r1,r2) pairs drawn at expanding diffuse hits */
    uint64_t exact_tests;    /* shadow rays that needed the exact closest-hit test (not culled by the filter) */
    uint64_t glass_walks;    /* shadow rays that entered the refraction walk of Light::_isCovered */
} crt_stats;

typedef struct crt_ctx crt_ctx;

/* ---- context ------------------------------------------------------------ */
/* Replaces initCuda() (main.cu:905-943): binds the context to HIP device
 * `device`, uploads the LTC tables (ltc.hpp:24,4127; _initLTCTextures,
 * AreaLight.hpp:51-54).  Synchronous. */
int crt_create(crt_ctx** out, int device);
/* Replaces free_world + the cudaFree block (main.cu:863-878). Synchronous. */
void crt_destroy(crt_ctx* ctx);
/* Message of the last failing call on this context ("" if none). */
const char* crt_last_error(const crt_ctx* ctx);

/* ---- scene -------------------------------------------------------------- */
/* Fills the Cornell box of create_world (main.cu:373-593): camera, 2 spheres,
 * 6 quads, 1 area light (already reordered, centred and rotated by 180 deg
 * about X as main.cu:591-592 does).  geometry must hold 8 entries, lights 1.
 * Pure host function. */
int crt_cornell_scene(uint32_t shadowSamples, crt_camera* camera, crt_geometry* geometry8,
                      crt_light* light1);
/* Replaces the create_world<<<1,1>>> launch (main.cu:694): flattens the scene
 * into the device-resident form the kernels read.  Synchronous. */
int crt_scene_upload(crt_ctx* ctx, const crt_scene_desc* scene);

/* ---- RNG ---------------------------------------------------------------- */
/* Replaces randomInit/random_init (main.cu:946-960, 317-325): one XORWOW
 * state per pixel, state(idx) = curand_init(seed + idx, 0, 0), idx = y*nx + x.
 * The reference passes time(NULL) as the seed (main.cu:952). */
int crt_rng_init(crt_ctx* ctx, uint32_t nx, uint32_t ny, uint64_t seed, void* stream);
/* Copies the per-pixel states to host as nx*ny rows of 6 words
 * {v0,v1,v2,v3,v4,d}.  Synchronous.  d advances by 362437 per draw, so
 * (d_after - d_before) * 362437^-1 mod 2^32 is the pixel's draw count. */
int crt_rng_export(crt_ctx* ctx, uint32_t* host_state6);

/* ---- render ------------------------------------------------------------- */
/* Replaces renderWithCuda/render (main.cu:963-977, 328-347): renders every
 * pixel once into fb_dev (nx*ny RGBA float, row-major, index (j*nx+i)*4+c) and
 * advances the per-pixel RNG states.  Whether the call waits for the GPU depends
 * on the pipeline (crt_set_pipeline); work queued later on `stream` sees the
 * finished image either way. */
int crt_render(crt_ctx* ctx, const crt_settings* st, float* fb_dev, void* stream);
/* crt_render for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same image and same RNG end
 * states as crt_render, always through the persistent kernel (CRT_PIPELINE_PERSISTENT), asynchronous on
 * `stream`.  The kernel's scratch (work-unit cursor, wave tallies, two work counters per pixel) lives in a
 * static device array of 64 + 2 * CRT_STATIC_MAX_PIXELS words (8 MiB) that every GPU the library runs on
 * holds until the process ends; the call makes no hipMalloc/hipFree and no allocating library call (set-up
 * -- crt_create, crt_scene_upload, crt_rng_init -- has allocated what the context owns).  Images of more than
 * CRT_STATIC_MAX_PIXELS pixels are refused with CRT_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's
 * array (the part a call uses is cleared on `stream` first).  A call on another stream or context than the
 * previous one on that GPU is queued behind it (a per-device event, made by crt_create), so such calls run one
 * after the other instead of overlapping on the array; calls made from two host threads at the same moment
 * still race on that bookkeeping and need the caller's own lock.  The work counters of these calls stay in the
 * array and are not part of crt_stats_since_init. */
#define CRT_STATIC_MAX_PIXELS (1u << 20)
int crt_render_static(crt_ctx* ctx, const crt_settings* st, float* fb_dev, void* stream);
/* Replaces one render_partial launch of the progressive loop (main.cu:349-368,
 * 719-759): renders the window of bx*by reference blocks (tx*ty pixels each)
 * that starts at block (tileIndex % blocksX, min(tileIndex / blocksX, blocksY))
 * and wraps around modulo (blocksX, blocksY), blocksX = nx/tx+1, blocksY = ny/ty+1,
 * bx = min(blocksPerDraw, blocksX), by = min(max(1, blocksPerDraw/blocksX+1), blocksY).
 * Because bx <= blocksX and by <= blocksY no pixel is covered twice
 * by one call; successive calls do overlap (blocksPerDraw is not a multiple of
 * blocksX in general) and re-rendered pixels continue their RNG stream, as in
 * the reference. */
int crt_render_partial(crt_ctx* ctx, const crt_settings* st, uint32_t tileIndex, float* fb_dev,
                       void* stream);
/* Multi-GPU shard: renders the pixels of the 8x8 tiles owned by `rank` out of
 * `world` ranks (tile t = ty*tilesX + tx belongs to rank crt_tile_owner(t))
 * into a compact buffer: owned tile k (ascending t) occupies
 * compact_dev[k*256 .. k*256+255] as 64 RGBA pixels, row-major inside the tile.
 * Pixels of a tile that fall outside the image are written as zeros.
 * No counterpart in the reference (single GPU, main.cu:914-940). */
int crt_render_shard(crt_ctx* ctx, const crt_settings* st, uint32_t rank, uint32_t world,
                     float* compact_dev, void* stream);
/* Number of 8x8 tiles rank owns, and the compact buffer length in floats. */
uint32_t crt_shard_tiles(uint32_t nx, uint32_t ny, uint32_t rank, uint32_t world);
/* Upper bound of crt_shard_tiles over ranks (every rank pads to this for a gather). */
uint32_t crt_shard_tiles_max(uint32_t nx, uint32_t ny, uint32_t world);
/* Scatters `world` gathered compact buffers (each padded to tiles_max*256
 * floats, rank-major) into the row-major framebuffer fb_dev. */
int crt_scatter_shards(crt_ctx* ctx, uint32_t nx, uint32_t ny, uint32_t world,
                       const float* gathered_dev, float* fb_dev, void* stream);
/* Exact work counters of everything rendered since the last crt_rng_init
 * (computed from the RNG draw counters and the settings).  Synchronous. */
int crt_stats_since_init(crt_ctx* ctx, const crt_settings* st, crt_stats* out);

/* Which kernels crt_render / crt_render_partial / crt_render_shard run.  Both give the same bits.
 *   CRT_PIPELINE_AUTO        the wavefront pipeline (default); the persistent kernel for launches the
 *                            pipeline cannot hold (more than about 31 million pixels at once, e.g. 5800 x 5800,
 *                            or no device memory left for its workspace)
 *   CRT_PIPELINE_PERSISTENT  one persistent kernel per launch, asynchronous on `stream`, no workspace
 *   CRT_PIPELINE_WAVEFRONT   trace / filter / resolve / shade / sum kernels with records in device
 *                            memory, sized from device-side counters.  A frame of one round whose ray
 *                            queue cannot overflow (gl_iter 0 at aa_iter <= 1) is asynchronous on
 *                            `stream` (its device-side capacity flags are read by the next call on the
 *                            context that looks: a render, crt_rng_export, crt_stats_since_init; a render on
 *                            another stream is ordered behind it); any other call returns after the image is complete (rounds are
 *                            queued back to back, the host waits for the last one, and a frame that
 *                            left more shadow rays undecided than the queue holds is rendered a second
 *                            time from its saved generator states)
 * The reference has one kernel and no such switch (main.cu:328-347). */
#define CRT_PIPELINE_AUTO 0
#define CRT_PIPELINE_PERSISTENT 1
#define CRT_PIPELINE_WAVEFRONT 2
int crt_set_pipeline(crt_ctx* ctx, int pipeline);
/* How the wavefront pipeline's trace kernel walks the last level of the global-illumination tree
 * (the ind_rays children of main.cu:176-191 that expand no further).  Same bits either way.
 *   CRT_TRACE_PER_PIXEL    one lane per pixel pops the children one after the other: fewest
 *                          instructions, best when the launch has enough pixels to fill the GPU
 *   CRT_TRACE_COOPERATIVE  a wave owns 8 pixels and its 64 lanes take one child ray each: twice the
 *                          instructions, but a pixel's dependent chain gets 10x shorter, which wins when
 *                          every wave would otherwise sit alone on its SIMD (the shards of a multi-GPU job)
 *   CRT_TRACE_AUTO         by launch size (default): below 200 000 pixels all units cooperative; up to
 *                          400 000 the few per cent with the highest cost estimate, next to the per-pixel
 *                          kernel on a second stream (they are what the launch would end up waiting for);
 *                          above that per-pixel only
 * No counterpart in the reference. */
#define CRT_TRACE_AUTO 0
#define CRT_TRACE_PER_PIXEL 1
#define CRT_TRACE_COOPERATIVE 2
int crt_set_trace_mode(crt_ctx* ctx, int mode);
/* Per-kernel GPU time of the wavefront pipeline, measured with HIP events on the launch stream.
 * crt_set_timing(ctx, 1) starts (and zeroes) the accumulation, crt_get_kernel_times returns
 * milliseconds and launch counts in the order trace, filter, resolve, shade, sum.  A workspace
 * reallocation (new image size) keeps the setting but restarts the totals.  For a launch rendered in two
 * pieces the figures are sums over both pieces, whose kernels overlap.  Measurement only; the
 * reference prints one wall-clock "Render time" (main.cu:712-716). */
int crt_set_timing(crt_ctx* ctx, int enabled);
int crt_get_kernel_times(crt_ctx* ctx, double ms[5], uint64_t launches[5]);
/* Test instrument of the wavefront pipeline.  Its filter kernel decides most shadow rays ("free" or
 * "blocked by an opaque primitive") with conservative tests and sends the rest to the exact test.
 * With the audit enabled every decided ray is also put through the exact test
 * (Light::_isCovered, Light.hpp:35-86) and contradictions are counted; crt_audit_result returns the
 * totals since the last crt_rng_init.  A correct build reports 0 contradictions.  Slow; off by default. */
int crt_set_audit(crt_ctx* ctx, int enabled);
int crt_audit_result(crt_ctx* ctx, uint64_t* checked, uint64_t* contradictions);
/* Allocates the wavefront pipeline's workspace for crt_render (world == 1) or for
 * crt_render_shard(rank, world) with these settings now, instead of inside the first render call.
 * Optional; a no-op for the persistent pipeline.  Synchronous. */
int crt_reserve_workspace(crt_ctx* ctx, const crt_settings* st, uint32_t rank, uint32_t world);
/* Upper bound in bytes of the wavefront pipeline's device workspace (default 96 GiB, of which a launch takes
 * what its rounds can use; if the device has less left, the library halves the bound until it fits); a smaller
 * bound means more rounds per image, never a different image.  No counterpart in the reference. */
int crt_set_workspace_limit(crt_ctx* ctx, size_t bytes);
/* Measuring aid: wavefront frames of this context stop after `rounds` pipeline rounds (0, the default: no
 * limit), to time the steady state of a launch whose whole frame takes an hour.  A frame that is cut short
 * returns CRT_ERR_TRUNCATED: its image is unfinished and its RNG states are left in the middle of their trees
 * (crt_rng_init before anything else is rendered).  Frames that finish within the limit return CRT_OK.  No
 * counterpart in the reference. */
int crt_set_round_limit(crt_ctx* ctx, uint32_t rounds);

/* Device self-test: compares the kernels' fast correctly-rounded reciprocal with
 * the IEEE quotient 1.0f/x for all 2^32 float bit patterns.  Returns the number
 * of mismatches (expected 0) and the smallest mismatching bit pattern
 * (0xFFFFFFFF if none).  Synchronous, about a second.  No counterpart in the
 * reference. */
int crt_selftest_rcp(crt_ctx* ctx, uint64_t* mismatches, uint32_t* first_bad_bits);

/* ---- device-buffer helpers for hosts without their own allocator -------- */
/* Allocates and zero-fills n_floats floats of device memory.  Synchronous. */
int crt_fb_alloc(crt_ctx* ctx, size_t n_floats, float** out_dev);
int crt_fb_free(crt_ctx* ctx, float* dev);
/* Synchronous device-to-host copy (the reference reads managed memory, main.cu:816,856). */
int crt_fb_download(crt_ctx* ctx, const float* dev, float* host, size_t n_floats, void* stream);

/* ---- output ------------------------------------------------------------- */
/* Replaces stbi_write_hdr("file.hdr", nx, ny, 4, fb) (main.cu:856): Radiance
 * RGBE, "-Y h +X w", new-style RLE when 8 <= w < 32768, first three of `comp`
 * channels, float -> RGBE by truncation.  Pure host function; returns CRT_OK or
 * CRT_ERR_IO/CRT_ERR_INVALID. */
int crt_write_hdr(const char* path, int w, int h, int comp, const float* data);
/* Same encoder into memory; returns the byte count (or a negative status).
 * Pass out=NULL to query the size bound. */
long crt_encode_hdr(int w, int h, int comp, const float* data, unsigned char* out, size_t cap);

/* 8-bit preview of a float image, the conversion the reference applies before handing the frame to its
 * SFML texture: out[i] = (uint8_t)clamp(in[i] * 255.0f, 0.f, 255.f) for every channel value
 * (main.cu:816-818); NaN maps to 0.  Host code, n values in and out. */
int crt_to_rgba8(const float* in, size_t n, uint8_t* out);
/* Writes the RGB part of an RGBA8 image as a binary PPM (P6), rows top to bottom -- the headless
 * stand-in for the reference's preview window (main.cu:812-853). */
int crt_write_ppm(const char* path, int w, int h, const uint8_t* rgba8);

#ifdef __cplusplus
}
#endif
#endif /* CRT_HIP_H */
