// Wavefront form of the per-pixel render path for gfx950
// (render -> aa_color -> color -> GeometryList::hit / AreaLight::calculateColor, main.cu:110-347).
//
// The reference walks each pixel's sample tree depth-first inside one thread.  Its control flow
// (which rays exist, which generator draws they use) does not depend on any shading result as long
// as aa_iter <= 1, and colours are only ever *added*, in tree order.  The work is therefore split
// into dense kernels that hand records to each other through HBM:
//
//   trace    one lane per pixel walks the tree (camera rays, mirror/glass bounces, GI children) with
//            the pixel's XORWOW stream, but shades nothing: every diffuse hit becomes a shade record
//            {P, N, view, attenuation, primitive, generator state, result slot} and the generator is
//            stepped past the 2*shadowSamples draws the hit will consume.  Misses write their
//            (attenuation x black) term straight into their result slot.  A pixel fills at most
//            `quota` slots per round and keeps its traversal state in HBM between rounds.  In small
//            launches (and for the most expensive pixels of medium ones) a second form of the kernel
//            expands the last GI level with one lane per child ray instead (see kCoopBatch).
//   cost     ordering only: estimates every 64-pixel unit's number of tree steps, so that the trace
//            kernel can start the expensive units first.
//   filter   one lane per shade record draws the light samples and classifies every shadow ray with
//            division-free tests against the primitives that survive a per-record cull: certainly
//            free, certainly blocked by an opaque primitive, or undecided.  Undecided rays (a few per
//            cent) are compacted through LDS into a queue.
//   resolve  one lane per queued ray runs the reference's exact closest-hit test and, for glass, the
//            refraction walk of Light::_isCovered.
//   shade    one lane per shade record evaluates the two LTC integrals with the final visibility and
//            writes attenuation x colour into the record's result slot.
//   sum      one lane per pixel adds its slots in tree order, folds finished color() calls into the
//            pixel (aa_color's weights) and writes finished pixels.
//
// Every kernel runs with all 64 lanes busy on identical work, needs few registers (many waves per
// SIMD), and the sample tree, the draw order and every floating-point operation stay those of the
// reference, so results are bit-identical to the single-kernel path in crt_render.hip.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "crt_device.hpp"
#include "crt_render.hpp"
#include "crt_shade.hpp"

namespace crt {

namespace {

// slot tags: type in bits 0..1; an end-of-call tag carries its AA task's level (bits 2..4) and corner (bits 5..6)
enum : uint32_t { TAG_ADD = 0, TAG_END_CALL = 1, TAG_END_PIXEL = 2, TAG_SKIP = 3 /* a slot without a result */ };
// verdict of the sum kernel on a finished level >= 2 AA task (main.cu:283-300)
enum : uint32_t { AA_PENDING = 0, AA_ALL_EQUAL = 1, AA_SUBDIVIDE = 2 };
constexpr uint32_t kWorkChunk = 256;         // shade-record indices a trace wave reserves at a time
constexpr uint32_t kMaxFilterWaves = 16384;   // upper bound of the filter kernel's grid
constexpr uint32_t kNoRecord = 0xffffffffu;  // primitive index of an unused record slot
constexpr uint32_t kFromCounter = 0xffffffffu; // kernel argument: take the record count from WF_CTR_WORK

// state planes
enum : uint32_t {
    ST_CTL = 0,   // bit0 done, bit1 color_active, bit2 cur_valid, bit3 task_active, bit4 waiting for the AA verdict,
                  // bits 5..7 task level, bits 8..10 task corner, bits 12..15 AA stack fill, bits 16..19 nframes
    ST_CUR_O = 1, ST_CUR_D = 4, ST_CUR_ATT = 7, ST_CUR_REF = 11, ST_CUR_DEPTH = 12, ST_RAY_D = 13, ST_ON_STACK = 16,
    ST_T_CX = 17, ST_T_CY = 18, // centre of the running AA task
    ST_AA = 19,                 // AA task stack: 10 x {cx, cy, level} (aa_color, main.cu:235-242; aa_iter <= 4 bounds it)
};
// frame planes
// A frame is five 16-byte groups, each an array over the pixel slots (a wave reads or writes 1 KB per group):
enum : uint32_t {
    FG_POINT = 0, // {P.x, P.y, P.z, primitive}
    FG_NORMAL,    // {N.x, N.y, N.z, gl depth of the children}
    FG_ATT,       // attenuation of the node
    FG_GEN,       // {v0, v1, v2, v3} of the generator state behind the last child still on the stack
    FG_GEN2,      // {v4, d, children still on the stack, -}
    FG_GROUPS
};
static_assert(WF_FRAME_PLANES == 4u * FG_GROUPS, "frame = FG_GROUPS groups of four words");
// Shade records: six arrays of 16-byte groups behind WfArgs::work, each indexed by record (cap_work entries), so that
// a wave's consecutive records are one contiguous kilobyte per array and access: every store instruction of the trace
// kernel then writes whole lines (they are streamed past the caches, store_streaming), every load reads whole lines.
//   0  {P.x, P.y, P.z, primitive | ordinal << 8}     filter, resolve, shade
//   1  {N.x, N.y, N.z, result slot}                  shade
//   2  {view.x, view.y, view.z, att.w}               shade
//   3  {att.x, att.y, att.z, -}                      shade
//   4  {v0, v1, v2, v3}   5  {v4, d, -, -}           filter (the records it samples)
// (Word planes, one per field, had every kernel gather 256-byte pieces from up to 21 streams 650 MB apart: the shade
// kernel ran at 2.5 TB/s whatever its arithmetic.  One record's surface groups next to each other, 48 bytes apart
// from lane to lane, made three instructions share every line: 0.2 ms slower per 720x720 gl_iter 1 frame.)
static_assert(WF_WORK_PLANES == 24, "six groups of four words per record");

struct WfArgs {
    RenderArgs r;
    uint32_t n_lp, quota, cap_work, qcap;
    uint32_t* ctrs;
    uint32_t* state;
    uint32_t* frames;
    uint8_t* tags;
    float4* contrib;
    uint32_t* slot_count;
    float4* px_acc;
    uint32_t* aa_verdict; // [n_lp]: AA_* | all-equal-so-far << 8
    uint32_t* work;
    uint32_t* shadow_cnt;
    const uint32_t* jump_tabs; // rng_jump table for jump_n[0] = the draws of one shaded point (copied to LDS by the trace kernels)
    uint32_t jump_n[1];        // 0: no table
    uint32_t* queue;
    uint32_t qchunk;           // queue entries a filter wave reserves at a time (a multiple of 64)
    const uint32_t* unit_order; // trace: launch slot -> work unit, most expensive first; null = identity
    float* draws;              // cooperative trace: [2 * CRT_RAY_STACK][n_lp] child draws of the innermost leaf node
    uint32_t work_static;      // record indices [0, work_static): the first chunk of every per-pixel trace wave, by its place
                               // in the launch; WF_CTR_WORK counts the chunks handed out behind them
    const uint32_t* draw_tabs; // per-pixel trace: draw_tabs_n rng_jump tables in device memory, table a-1 jumps the 2a draws of a child rays
    uint32_t draw_tabs_n;
    const uint32_t* coop_tabs; // cooperative trace: kCoopTables rng_jump tables in device memory, table k-1 jumps k shaded points' worth of draws
};

// Measuring aid, off by default: -DCRT_EARLY_ROUND_END=K ends a trace wave's round once the pixels still filling
// slots are fewer than 1/K of those that have filled theirs.  With 141 slots per pixel and round (8 GiB workspace) a
// 720x720 gl_iter 2 frame goes from 1467 to 1370 ms (K = 4); with the default 508 it is even (1322 ms): the rounds
// the cut-off pixels need at the end cost what the ragged round ends did.
#ifndef CRT_EARLY_ROUND_END
#define CRT_EARLY_ROUND_END 0
#endif
DI float u2f(uint32_t x) { return __uint_as_float(x); }
DI uint32_t f2u(float x) { return __float_as_uint(x); }
// Result slot k of pixel slot lp (tags and contributions).  Unit-major: the quota rows of one 64-pixel unit lie
// together, a row being the unit's 64 lanes, so what a wave of the trace, shade or sum kernel touches at a time is
// a few neighbouring kilobytes instead of pieces one whole image apart.
DI size_t slot_at(const WfArgs& w, uint32_t k, size_t lp) { return ((lp >> 6) * w.quota + k) * 64u + (lp & 63u); }
DI size_t slot_pixel(const WfArgs& w, size_t slot) { return (slot / ((size_t)w.quota * 64u)) * 64u + (slot & 63u); }
DI uint32_t records_handed_out(const WfArgs& w) { // record indices below this may hold a record
    const uint32_t n = w.work_static + w.ctrs[WF_CTR_WORK];
    return n < w.cap_work ? n : w.cap_work;
}
DI uint4* frame_at(const WfArgs& w, uint32_t level, size_t lp) { // group g of the frame: [g * w.n_lp]
    return reinterpret_cast<uint4*>(w.frames) + ((size_t)level * FG_GROUPS) * w.n_lp + lp;
}
// A 16-byte store of data this kernel does not read again (shade records, finished results): written with the
// non-temporal hint, so that the stream does not push the GI frames, which the trace kernel reads again for every
// child ray, out of the caches.  Measured on the 720x720 gl_iter 1 frame: trace 9.15 -> 8.7 ms, 5.4 -> 0.7 GB fetched.  (The same hint on
// the shade kernel's scattered result stores costs 1.4 ms, on its record loads 0.25 ms: those lines are touched by
// several instructions and want to be combined in the L2.)
typedef uint32_t X4 __attribute__((ext_vector_type(4)));
DI void store_streaming(uint4* p, uint4 v) { __builtin_nontemporal_store(X4{v.x, v.y, v.z, v.w}, reinterpret_cast<X4*>(p)); }
DI void store_streaming(float4* p, float4 v) { store_streaming(reinterpret_cast<uint4*>(p), make_uint4(f2u(v.x), f2u(v.y), f2u(v.z), f2u(v.w))); }
DI uint4* rec_point(const WfArgs& w) { return reinterpret_cast<uint4*>(w.work); }
DI uint4* rec_surface(const WfArgs& w) { return reinterpret_cast<uint4*>(w.work) + (size_t)w.cap_work; }
DI uint4* rec_gen(const WfArgs& w) { return reinterpret_cast<uint4*>(w.work) + (size_t)w.cap_work * 4u; }
DI Rng rec_load_gen(const WfArgs& w, uint32_t wi) {
    const uint4 g0 = rec_gen(w)[wi], g1 = rec_gen(w)[(size_t)w.cap_work + wi];
    return Rng{g0.x, g0.y, g0.z, g0.w, g1.x, g1.y};
}

// ------------------------------------------------------------------------- shadow-ray classification
// Constants of AreaLight::_randomPoint (AreaLight.hpp:340-375) in scalar registers.
struct LightC {
    V3 pos, p0, p2, e01, e03, e23, e21;
};
DI LightC load_lightc(const DevLight CRT_CONST* L) {
    LightC c;
    c.pos = ld3c(L->pos);
    c.p0 = ld3c(L->p0);
    c.p2 = ld3c(L->p2);
    c.e01 = ld3c(L->e01);
    c.e03 = ld3c(L->e03);
    c.e23 = ld3c(L->e23);
    c.e21 = ld3c(L->e21);
    return c;
}
// Shadow ray of one light sample (AreaLight.hpp:340-375,495-510; Ray.hpp:45-52).
DI void shadow_ray(const LightC& lc, V3 P, float u, float v, V3& o, V3& d, float& x) {
    V3 sample;
    if (u + v < 1.0f) {
        sample = ((lc.p0 + lc.e01 * u) + lc.e03 * v) + lc.pos;
    } else {
        u = 1.0f - u;
        v = 1.0f - v;
        sample = ((lc.p2 + lc.e23 * u) + lc.e21 * v) + lc.pos;
    }
    V3 dir = sample - P;
    float dist;
    V3 ndir = normalize3_len(dir, dist);
    if (!eps_equal(dist, 0.0f, kEps)) dir = ndir;
    o = P + dir * 0.01f;
    d = ray_dir(dir);         // the Ray ctor normalises once more
    x = max_std(0.0f, dist);  // Ray::setDistance
}

constexpr uint32_t kStageCap = 128; // < 64 left over + 64 new entries per append

struct Stage {
    uint32_t head, fill; // wave-uniform ring cursor
    uint32_t written;    // entries already moved to the queue
    uint32_t q_next, q_end; // the wave's piece of the queue: next free entry, end (q_next == q_end: none)
};
constexpr uint32_t kNoRay = 0xffffffffu; // queue key of an entry that holds no ray
// A queued ray's key: its shade record's index in the low kRecBits bits, its light above them.
#ifndef CRT_REC_BITS
#define CRT_REC_BITS 29
#endif
constexpr uint32_t kRecBits = CRT_REC_BITS, kRecMask = (1u << kRecBits) - 1u;
constexpr uint64_t kRecLimit = 1ull << kRecBits;
static_assert(CRT_MAX_LIGHTS <= (1u << (32u - kRecBits)), "the light must fit above the record index");

// Moves `n` staged entries (n <= 64) to the queue.  A wave reserves the queue in pieces of w.qchunk entries
// (one atomic on the device counter per piece: atomics on one address are served one after the other, about
// 9 ns each, which at one per 64 rays was three times the kernel's own time); what it leaves unused of its
// last piece is marked "no ray" by stage_close.  When the queue is full the rays are dropped and the frame is
// marked (the host renders it again with launches sized for the worst case).
DI void stage_flush(const WfArgs& w, uint32_t (*stg)[kStageCap], Stage& st, uint32_t n, uint32_t lane) {
    __syncthreads();
    if (st.q_next == st.q_end) { // pieces are multiples of 64 and flushes before the last one are 64: no straddling
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(&w.ctrs[WF_CTR_QUEUE], w.qchunk);
        st.q_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        st.q_end = st.q_next + w.qchunk;
    }
    if (lane < n) {
        const uint32_t pos = (st.head + lane) & (kStageCap - 1u);
        const uint32_t idx = st.q_next + lane;
        if (idx < w.qcap && st.q_end > st.q_next) { // (q_end < q_next: the 32-bit counter wrapped)
            w.queue[idx] = stg[0][pos];
            w.queue[(size_t)w.qcap + idx] = stg[1][pos];
            w.queue[2 * (size_t)w.qcap + idx] = stg[2][pos];
        } else {
            w.ctrs[WF_CTR_QOVERFLOW] = 1u;
        }
    }
    st.q_next += n;
    st.head = (st.head + n) & (kStageCap - 1u);
    st.fill -= n;
    st.written += n;
    __syncthreads();
}
// End of the kernel: the last entries, then "no ray" marks on the rest of the wave's piece.
DI void stage_close(const WfArgs& w, uint32_t (*stg)[kStageCap], Stage& st, uint32_t lane) {
    if (st.fill != 0u) stage_flush(w, stg, st, st.fill, lane);
    if (st.q_end > st.q_next)
        for (uint32_t idx = st.q_next + lane; idx < st.q_end; idx += 64u)
            if (idx < w.qcap) w.queue[idx] = kNoRay;
}

// Conservative classification of NB shadow rays of one shaded point against the primitives in
// `prim_mask`.  Nothing here reproduces the reference's rounding: every statement carries a margin
// (1e-3 relative, or 2e-5 of ray length) that is orders of magnitude above the 1e-6 rounding of the
// reference's own test, and every ray that is not decided with that margin is left undecided for
// the exact test.  The ray is handled un-normalised: points P + s D, D = X - P towards the light
// sample X, a = D.D.  The reference's ray starts at s0 = 0.01/|D| and ends at 1 + s0
// (AreaLight.hpp:495-510: origin P + 0.01 dir, length |X - P|).
//
// Sphere, Lp = P - C, bq = D.Lp, c0 = Lp.Lp - r^2, e0 = Lp.Lp + r^2, disc = bq^2 - a c0
// (= a^2/4 times the reference's discriminant, Sphere.hpp:56-63):
//   free   disc < -margin (the line misses), or bq > 0 and c0 > 0 with margins (sphere behind the origin:
//          b > 0, c > 0 in the reference, both roots negative, Sphere.hpp:85-96);
//   sure   P further than 0.02 from the surface, disc > margin (so bq^2 > a c0 > 0 and bq is far from 0),
//          bq < 0: both roots positive, the first is s1 = c0 / (-bq + sqrt(disc)) < c0 / -bq <= 0.999,
//          i.e. before the light sample, and more than 0.01 beyond the ray origin.
// Quad, sP = nor.(P - v0), dnq = nor.D; the reference's t = -(p.nor / d.nor) = -sP |D| / dnq - 0.01
// (Quad.hpp:222-226):
//   free   sP dnq > 0 with P robustly off the plane (t < 0);
//          or |sP| |D| < 0.005 |dnq| and the ray not grazing (P on the plane: the 0.01 step decides, t < 0);
//          or opposite signs and (|sP| - |dnq|) |D| > 0.01002 |dnq| (t > |D|);
//          or the plane point P - (sP/dnq) D is outside the quad's bounding box (DevGeomBox).
//   In the second and third statement |sP| and |dnq| are replaced by the end of [x - N, x + N] that makes the
//   statement harder, N = DevGeomBox::noise: rounding errors of the two dot products are absolute (they grow
//   with the scene's coordinates), so for a grazing ray that meets the plane near its end they can exceed
//   any relative margin.
struct Classes {
    uint32_t sure_opaque, maybe_any, maybe_glass; // bit j = ray j
};

template <int NB>
DI Classes classify_block(ScenePtr sc, uint64_t prim_mask, V3 P, const V3 (&D)[NB], const float (&a)[NB]) {
    Classes k = {0u, 0u, 0u};
    uint64_t m = prim_mask;
    while (m != 0) {
        const uint32_t i = (uint32_t)__builtin_ctzll(m);
        m &= m - 1u;
        const DevGeomHot CRT_CONST* g = &sc->hot[i];
        const uint32_t kind = g->kind;
        const bool glass = g->mat_type == CRT_MAT_REFRACTIVE;
        const V3 go = ld3c(g->o), gn = ld3c(g->n);
        if (kind == CRT_GEOM_SPHERE) {
            const float r2 = gn.x;
            const V3 Lp = P - go;
            const float l2 = __builtin_fmaf(Lp.x, Lp.x, __builtin_fmaf(Lp.y, Lp.y, Lp.z * Lp.z));
            const float c0 = l2 - r2, e0 = l2 + r2;
            // |Lp| > r + 0.02  <=>  c0 > 0.04 r + 4e-4; with r <= sqrt(e0): c0 > 0.04 sqrt(e0) + 4e-4 is sufficient,
            // tested without the root as c0 > 5e-4 and (c0 - 4e-4)^2 > 1.6e-3 e0
            const float cm = c0 - 4e-4f;
            const bool outside = c0 > 5e-4f && cm * cm > 1.61e-3f * e0 && e0 < 1e12f;
            const bool c_pos = c0 > 1e-3f * e0 && c0 > 1e-30f && e0 < 1e12f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float bq = __builtin_fmaf(D[j].x, Lp.x, __builtin_fmaf(D[j].y, Lp.y, D[j].z * Lp.z));
                const float bq2 = bq * bq;
                const float disc = __builtin_fmaf(-a[j], c0, bq2);
                const float marg = 1e-3f * __builtin_fmaf(a[j], e0, bq2);
                const bool a_ok = a[j] > 1e-6f && a[j] < 1e12f;
                const bool free_ = a_ok && (disc < -marg || (c_pos && bq > 0.0f && bq2 > 1e-6f * (a[j] * e0)));
                const bool sure = a_ok && outside && bq < 0.0f && disc > __builtin_fmaf(1e-4f, a[j], marg) &&
                                  c0 < -bq * 0.999f;
                if (!free_) {
                    if (glass) {
                        k.maybe_any |= 1u << j;
                        k.maybe_glass |= 1u << j;
                    } else if (sure) {
                        k.sure_opaque |= 1u << j;
                    } else {
                        k.maybe_any |= 1u << j;
                    }
                }
            }
        } else {
            const DevGeomBox CRT_CONST* bx = &sc->box[i];
            const V3 blo = ld3c(bx->lo), bhi = ld3c(bx->hi);
            const float n2 = bx->n2, sp_min2 = bx->sp_min2, nz = bx->noise;
            const V3 q = P - go;
            const float sP = __builtin_fmaf(q.x, gn.x, __builtin_fmaf(q.y, gn.y, q.z * gn.z));
            const float sP2 = sP * sP;
            const bool off_plane = sP2 > sp_min2;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                float dnq;
                if (kind == CRT_HOT_QUAD_X) dnq = D[j].x * gn.x;
                else if (kind == CRT_HOT_QUAD_Y) dnq = D[j].y * gn.y;
                else if (kind == CRT_HOT_QUAD_Z) dnq = D[j].z * gn.z;
                else dnq = __builtin_fmaf(D[j].x, gn.x, __builtin_fmaf(D[j].y, gn.y, D[j].z * gn.z));
                const float dn2 = dnq * dnq;
                const bool a_ok = a[j] > 1e-6f && a[j] < 1e12f;
                const bool same = sP * dnq > 0.0f;
                const bool behind = same && off_plane;
                const float ad = __builtin_fabsf(dnq), as = __builtin_fabsf(sP);
                const float s_hi = as + nz, d_lo = ad - nz, d_hi = ad + nz;
                const bool lifted = d_lo > 0.0f && (s_hi * s_hi) * a[j] < 2.5e-5f * (d_lo * d_lo) && dn2 > 4e-6f * (n2 * a[j]);
                const float e = (as - nz) - d_hi * 1.000004f;
                const bool beyond = !same && off_plane && e > 0.0f && (e * e) * a[j] > 1.004004e-4f * (d_hi * d_hi);
                // plane point times dnq, against the box times |dnq|
                const float sg = dnq < 0.0f ? -1.0f : 1.0f;
                const float qx = __builtin_fmaf(P.x, dnq, -(sP * D[j].x)) * sg;
                const float qy = __builtin_fmaf(P.y, dnq, -(sP * D[j].y)) * sg;
                const float qz = __builtin_fmaf(P.z, dnq, -(sP * D[j].z)) * sg;
                const bool outside = qx < blo.x * ad || qx > bhi.x * ad || qy < blo.y * ad || qy > bhi.y * ad ||
                                     qz < blo.z * ad || qz > bhi.z * ad;
                const bool boxed_out = outside && dn2 > 1e-4f * (n2 * a[j]);
                const bool free_ = a_ok && (behind || lifted || beyond || boxed_out);
                if (!free_) {
                    k.maybe_any |= 1u << j;
                    if (glass) k.maybe_glass |= 1u << j;
                }
            }
        }
    }
    return k;
}

// Light-sample offsets D = X - P for the filter (approximate: fused operations, the second triangle
// folded algebraically; the exact sample point is rebuilt from (u, v) by whoever needs it).
struct LightF {
    V3 B0, B1, e01, e03, e23, e21;
    float g11, g12x2, g22; // |e01|^2, 2 e01.e03, |e03|^2
    V3 nor;                // the light's plane normal (un-normalised)
    bool is_par;
};
DI LightF load_lightf(const DevLight CRT_CONST* L) {
    LightF f;
    f.B0 = ld3c(L->fB0);
    f.B1 = ld3c(L->fB1);
    f.e01 = ld3c(L->e01);
    f.e03 = ld3c(L->e03);
    f.e23 = ld3c(L->e23);
    f.e21 = ld3c(L->e21);
    f.g11 = L->fg[0];
    f.g12x2 = L->fg[1];
    f.g22 = L->fg[2];
    f.nor = ld3c(L->nor);
    f.is_par = L->is_par != 0u;
    return f;
}
DI V3 sample_offset(const LightF& lf, V3 Bp0, V3 Bp1, float u, float v) {
    V3 D = mk3(__builtin_fmaf(lf.e03.x, v, __builtin_fmaf(lf.e01.x, u, Bp0.x)),
               __builtin_fmaf(lf.e03.y, v, __builtin_fmaf(lf.e01.y, u, Bp0.y)),
               __builtin_fmaf(lf.e03.z, v, __builtin_fmaf(lf.e01.z, u, Bp0.z)));
    if (!lf.is_par && !(u + v < 1.0f)) {
        D = mk3(__builtin_fmaf(-lf.e21.x, v, __builtin_fmaf(-lf.e23.x, u, Bp1.x)),
                __builtin_fmaf(-lf.e21.y, v, __builtin_fmaf(-lf.e23.y, u, Bp1.y)),
                __builtin_fmaf(-lf.e21.z, v, __builtin_fmaf(-lf.e23.z, u, Bp1.z)));
    }
    return D;
}
// Quotient for the conservative tests only (1 ulp reciprocal estimate, no correction steps).
DI float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DI float fdot(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }

// For a parallelogram light D = Bp + e01 u + e03 v is affine in the two draws, so every quantity the
// classification needs is a polynomial in (u, v) whose coefficients depend only on the shaded point:
//   a  = D.D    = h0 + 2h1 u + 2h2 v + g11 u^2 + 2g12 uv + g22 v^2
//   bq = D.Lp   = b0 + b1 u + b2 v
//   disc = bq^2 - a c0 = A u^2 + B uv + C v^2 + Du u + Ev v + F
//   dnq = nor.D = d0 + d1 u + d2 v,   plane point times dnq:  q_k = qk0 + qk1 u + qk2 v
// Two spheres and one quad of the wave's candidate set get such coefficient sets ("slots") per lane;
// further candidates take the generic path (classify_block).  The per-sample margins of
// classify_block are replaced by their maxima over the light (a_max, max bq^2: both are attained at
// corners), which only widens the undecided band.
struct SphSlot {
    float A, B, C, Du, Ev, F, b0, b1, b2;
    float M;  // free:  disc < -M
    float M2; // sure:  disc > M2  (+inf when the point is not clearly outside the sphere)
    float kb; // free:  bq > kb    (+inf unless c0 is clearly positive)
    float kc; // sure:  bq < -kc
};
struct QuadSlot {
    float sP, d0, d1, d2;
    float s_lo, s_hi; // |sP| -+ the noise bound
    float q[3][3]; // [axis][1, u, v]
    bool offp;
};
struct PolyA {
    float h0, h1x2, h2x2;
};
// Quad parallel to the light: dnq = nor.D is one number per shaded point, so the plane is met at the
// fixed ray parameter s_c = -sP/dnq and the plane point P + s_c D is affine in (u, v); the statements
// of quad_slot_class that involve a = D.D turn into thresholds on a.
struct QuadParSlot {
    float r[3][3];        // [axis][1, u, v]: plane point minus the box centre
    float t_lift, t_beyond, t_steep, t_between; // free if a < t_lift or a > t_beyond; box tests need a < t_steep; sure needs a > t_between
    bool behind;
};

DI SphSlot make_sph_slot(V3 P, V3 center, float r2, V3 Bp, const LightF& lf, const PolyA& pa, float a_max) {
    SphSlot s;
    const V3 Lp = P - center;
    const float l2 = fdot(Lp, Lp);
    const float c0 = l2 - r2, e0 = l2 + r2;
    s.b0 = fdot(Bp, Lp);
    s.b1 = fdot(lf.e01, Lp);
    s.b2 = fdot(lf.e03, Lp);
    s.A = __builtin_fmaf(s.b1, s.b1, -(c0 * lf.g11));
    s.B = __builtin_fmaf(2.0f * s.b1, s.b2, -(c0 * lf.g12x2));
    s.C = __builtin_fmaf(s.b2, s.b2, -(c0 * lf.g22));
    s.Du = __builtin_fmaf(2.0f * s.b0, s.b1, -(c0 * pa.h1x2));
    s.Ev = __builtin_fmaf(2.0f * s.b0, s.b2, -(c0 * pa.h2x2));
    s.F = __builtin_fmaf(s.b0, s.b0, -(c0 * pa.h0));
    const float q00 = s.b0, q10 = s.b0 + s.b1, q01 = s.b0 + s.b2, q11 = q10 + s.b2;
    const float bqmax2 = __builtin_fmaxf(__builtin_fmaxf(q00 * q00, q10 * q10), __builtin_fmaxf(q01 * q01, q11 * q11));
    s.M = 1e-3f * __builtin_fmaf(a_max, e0, bqmax2);
    const float cm = c0 - 4e-4f;
    const bool outside = c0 > 5e-4f && cm * cm > 1.61e-3f * e0 && e0 < 1e12f; // further than 0.02 from the surface
    const bool c_pos = c0 > 1e-3f * e0 && c0 > 1e-30f && e0 < 1e12f;
    s.M2 = outside ? __builtin_fmaf(1e-4f, a_max, s.M) : __builtin_inff();
    s.kb = c_pos ? __builtin_amdgcn_sqrtf(1e-6f * (a_max * e0)) * 1.001f : __builtin_inff();
    s.kc = c0 * 1.0011f;
    return s;
}
// returns 0 free, 1 undecided, 2 sure
DI uint32_t sph_slot_class(const SphSlot& s, float u, float v) {
    const float bq = __builtin_fmaf(s.b2, v, __builtin_fmaf(s.b1, u, s.b0));
    const float t1 = __builtin_fmaf(s.A, u, __builtin_fmaf(s.B, v, s.Du));
    const float t2 = __builtin_fmaf(s.C, v, s.Ev);
    const float disc = __builtin_fmaf(t1, u, __builtin_fmaf(t2, v, s.F));
    if (disc < -s.M || bq > s.kb) return 0u;
    return (disc > s.M2 && bq < -s.kc) ? 2u : 1u;
}

DI QuadSlot make_quad_slot(V3 P, V3 v0, V3 nor, float sp_min2, float nz, V3 Bp, const LightF& lf) {
    QuadSlot q;
    const V3 pq = P - v0;
    q.sP = fdot(pq, nor);
    q.offp = q.sP * q.sP > sp_min2;
    q.s_lo = __builtin_fabsf(q.sP) - nz;
    q.s_hi = __builtin_fabsf(q.sP) + nz;
    q.d0 = fdot(nor, Bp);
    q.d1 = fdot(nor, lf.e01);
    q.d2 = fdot(nor, lf.e03);
    const float Pk[3] = {P.x, P.y, P.z}, Bk[3] = {Bp.x, Bp.y, Bp.z}, e1[3] = {lf.e01.x, lf.e01.y, lf.e01.z},
                e3[3] = {lf.e03.x, lf.e03.y, lf.e03.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        q.q[k][0] = __builtin_fmaf(Pk[k], q.d0, -(q.sP * Bk[k]));
        q.q[k][1] = __builtin_fmaf(Pk[k], q.d1, -(q.sP * e1[k]));
        q.q[k][2] = __builtin_fmaf(Pk[k], q.d2, -(q.sP * e3[k]));
    }
    return q;
}
// 0 free, 1 undecided, 2 sure.  "Free" repeats the quad statements of classify_block.  "Sure" (axis-aligned
// rectangles only): the plane lies between P and the sample (s_c = |sP|/|dnq| < 0.999), more than 0.011 of
// ray length away from P (so t = s_c |D| - 0.01 > 0), the ray is not grazing, and the plane point is
// inside the rectangle shrunk by the margin, where the reference's 0 <= u,v <= 1 certainly holds.
DI uint32_t quad_slot_class(const QuadSlot& q, V3 blo, V3 bhi, V3 ilo, V3 ihi, float n2, float nz, const PolyA& pa,
                            const LightF& lf, float u, float v) {
    const float dnq = __builtin_fmaf(q.d2, v, __builtin_fmaf(q.d1, u, q.d0));
    const float a = __builtin_fmaf(__builtin_fmaf(lf.g11, u, __builtin_fmaf(lf.g12x2, v, pa.h1x2)), u,
                                   __builtin_fmaf(__builtin_fmaf(lf.g22, v, pa.h2x2), v, pa.h0));
    const float dn2 = dnq * dnq;
    const float sd = q.sP * dnq;
    const bool same = sd > 0.0f;
    const bool behind = same && q.offp;
    const float ad = __builtin_fabsf(dnq);
    const float d_lo = ad - nz, d_hi = ad + nz;
    const float dl2 = d_lo * d_lo, dh2 = d_hi * d_hi;
    const bool lifted = d_lo > 0.0f && (q.s_hi * q.s_hi) * a < 2.5e-5f * dl2 && dn2 > 4e-6f * (n2 * a);
    const float e = q.s_lo - d_hi * 1.000004f;
    const bool beyond = !same && q.offp && e > 0.0f && (e * e) * a > 1.004004e-4f * dh2;
    const float sg = dnq < 0.0f ? -1.0f : 1.0f;
    const float qx = __builtin_fmaf(q.q[0][2], v, __builtin_fmaf(q.q[0][1], u, q.q[0][0])) * sg;
    const float qy = __builtin_fmaf(q.q[1][2], v, __builtin_fmaf(q.q[1][1], u, q.q[1][0])) * sg;
    const float qz = __builtin_fmaf(q.q[2][2], v, __builtin_fmaf(q.q[2][1], u, q.q[2][0])) * sg;
    const bool outside = qx < blo.x * ad || qx > bhi.x * ad || qy < blo.y * ad || qy > bhi.y * ad || qz < blo.z * ad ||
                         qz > bhi.z * ad;
    const bool steep = dn2 > 1e-4f * (n2 * a);
    if (behind || lifted || beyond || (outside && steep)) return 0u;
    const bool inside = qx > ilo.x * ad && qx < ihi.x * ad && qy > ilo.y * ad && qy < ihi.y * ad && qz > ilo.z * ad &&
                        qz < ihi.z * ad;
    const bool between = sd < 0.0f && q.offp && steep && q.s_hi < 0.999f * d_lo && (q.s_lo * q.s_lo) * a > 1.21e-4f * dh2;
    return (inside && between) ? 2u : 1u;
}

DI QuadParSlot make_quad_par_slot(V3 P, V3 v0, V3 nor, float sp_min2, float n2, float nz, V3 mid, V3 Bp, const LightF& lf) {
    QuadParSlot q;
    const V3 pq = P - v0;
    const float sP = fdot(pq, nor), dnq = fdot(nor, Bp);
    const float sP2 = sP * sP, dn2 = dnq * dnq, ad = __builtin_fabsf(dnq);
    const bool offp = sP2 > sp_min2;
    const float sd = sP * dnq;
    q.behind = sd > 0.0f && offp;
    const float inf = __builtin_inff();
    // |sP| and |dnq| at the end of [x - nz, x + nz] that makes each statement harder
    const float s_lo = __builtin_fabsf(sP) - nz, s_hi = __builtin_fabsf(sP) + nz, d_lo = ad - nz, d_hi = ad + nz;
    q.t_lift = d_lo > 0.0f ? __builtin_fminf(fdiv(2.5e-5f * (d_lo * d_lo), s_hi * s_hi) * 0.9999f, fdiv(dn2, 4e-6f * n2) * 0.9999f) : 0.0f;
    const float e = s_lo - d_hi * 1.000004f;
    q.t_beyond = (!(sd > 0.0f) && offp && e > 0.0f) ? fdiv(1.004004e-4f * (d_hi * d_hi), e * e) * 1.0001f : inf;
    q.t_steep = fdiv(dn2, 1e-4f * n2) * 0.9999f;
    q.t_between = (sd < 0.0f && offp && s_hi < 0.999f * d_lo) ? fdiv(1.21e-4f * (d_hi * d_hi), s_lo * s_lo) * 1.0001f : inf;
    const float sc_ = -fdiv(sP, dnq); // ray parameter of the plane (box margins cover the 1e-7 error)
    const float Pk[3] = {P.x, P.y, P.z}, Bk[3] = {Bp.x, Bp.y, Bp.z}, e1[3] = {lf.e01.x, lf.e01.y, lf.e01.z},
                e3[3] = {lf.e03.x, lf.e03.y, lf.e03.z}, mk[3] = {mid.x, mid.y, mid.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        q.r[k][0] = __builtin_fmaf(sc_, Bk[k], Pk[k] - mk[k]);
        q.r[k][1] = sc_ * e1[k];
        q.r[k][2] = sc_ * e3[k];
    }
    return q;
}
DI uint32_t quad_par_slot_class(const QuadParSlot& q, V3 hout, V3 hin, const PolyA& pa, const LightF& lf, float u, float v) {
    const float a = __builtin_fmaf(__builtin_fmaf(lf.g11, u, __builtin_fmaf(lf.g12x2, v, pa.h1x2)), u,
                                   __builtin_fmaf(__builtin_fmaf(lf.g22, v, pa.h2x2), v, pa.h0));
    const float rx = __builtin_fabsf(__builtin_fmaf(q.r[0][2], v, __builtin_fmaf(q.r[0][1], u, q.r[0][0])));
    const float ry = __builtin_fabsf(__builtin_fmaf(q.r[1][2], v, __builtin_fmaf(q.r[1][1], u, q.r[1][0])));
    const float rz = __builtin_fabsf(__builtin_fmaf(q.r[2][2], v, __builtin_fmaf(q.r[2][1], u, q.r[2][0])));
    const bool steep = a < q.t_steep;
    const bool outside = rx > hout.x || ry > hout.y || rz > hout.z;
    if (q.behind || a < q.t_lift || a > q.t_beyond || (outside && steep)) return 0u;
    const bool inside = rx < hin.x && ry < hin.y && rz < hin.z;
    return (inside && steep && a > q.t_between) ? 2u : 1u;
}

// Everything the filter derives once per (shaded point, light) before its sample loop.
struct RecordFilter {
    // wave-uniform
    bool poly;
    int sph_i[2], quad_i;
    bool sph_glass[2], quad_glass, quad_par;
    V3 blo, bhi, ilo, ihi;
    float n2, nz;
    uint64_t rest_mask;
    // per lane
    bool rec_ok;
    V3 P, Bp0, Bp1;
    PolyA pa;
    SphSlot s[2];
    union { // one of the two, by quad_par
        QuadSlot q;
        QuadParSlot qp;
    };
    V3 hout, hin;
};

// Primitives that may stop a shadow ray from P towards light `li` (per lane): all spheres, and the quads
// the per-light plane test of DevScene::cull_side / cull_thr does not exclude.
DI uint64_t shadow_candidates(ScenePtr sc, uint32_t li, V3 P) {
    uint64_t mask = 0;
    const uint32_t n_geom = sc->n_geom;
    for (uint32_t i = 0; i < n_geom; ++i) {
        const DevGeomHot CRT_CONST* g = &sc->hot[i];
        bool keep = true;
        if (g->kind != CRT_GEOM_SPHERE) {
            const V3 q = P - ld3c(g->o);
            const float s = (q.x * g->n[0] + q.y * g->n[1]) + q.z * g->n[2];
            keep = !(s * sc->cull_side[li][i] > sc->cull_thr[li][i]);
        }
        if (keep) mask |= 1ull << i;
    }
    return mask;
}

DI RecordFilter make_record_filter(ScenePtr sc, uint32_t li, const LightF& lf, V3 P) {
    RecordFilter f;
    f.P = P;
    f.Bp0 = lf.B0 - P;
    f.Bp1 = lf.B1 - P;
    // primitives that may stop a shadow ray of some record of this wave
    const uint64_t lane_mask = shadow_candidates(sc, li, P);
    uint64_t prim_mask = 0;
    const uint32_t n_geom = sc->n_geom;
    for (uint32_t i = 0; i < n_geom; ++i)
        if (__any((lane_mask >> i) & 1ull)) prim_mask |= 1ull << i;
    f.poly = lf.is_par;
    f.sph_i[0] = f.sph_i[1] = f.quad_i = -1;
    f.sph_glass[0] = f.sph_glass[1] = f.quad_glass = f.quad_par = false;
    f.hout = f.hin = mk3(0, 0, 0);
    f.blo = f.bhi = f.ilo = f.ihi = mk3(0, 0, 0);
    f.n2 = 0.0f;
    f.nz = __builtin_inff();
    f.rest_mask = prim_mask;
    f.rec_ok = true;
    f.pa.h0 = f.pa.h1x2 = f.pa.h2x2 = 0.0f;
    if (!f.poly) return f;
    // a(u,v) and its range over the light
    f.pa.h0 = fdot(f.Bp0, f.Bp0);
    f.pa.h1x2 = 2.0f * fdot(f.Bp0, lf.e01);
    f.pa.h2x2 = 2.0f * fdot(f.Bp0, lf.e03);
    const float a10 = f.pa.h0 + f.pa.h1x2 + lf.g11, a01 = f.pa.h0 + f.pa.h2x2 + lf.g22;
    const float a11 = a10 + f.pa.h2x2 + lf.g22 + lf.g12x2;
    const float a_max = __builtin_fmaxf(__builtin_fmaxf(f.pa.h0, a10), __builtin_fmaxf(a01, a11));
    // |D| >= distance to the light's plane; the statements below need |D| > 1e-3
    const float pl = fdot(lf.nor, f.Bp0);
    const float nl2 = fdot(lf.nor, lf.nor);
    f.rec_ok = pl * pl > 1.0001e-6f * nl2 && a_max < 1e12f && nl2 > 1e-30f && nl2 < 1e30f;
    // slot assignment (wave-uniform): the first two spheres and the first quad of the candidate set
    int i0 = -1, i1 = -1, iq = -1;
    for (uint64_t m = prim_mask; m != 0; m &= m - 1u) {
        const int i = __builtin_ctzll(m);
        if (sc->hot[i].kind == CRT_GEOM_SPHERE) {
            if (i0 < 0) i0 = i;
            else if (i1 < 0) i1 = i;
        } else if (iq < 0) {
            iq = i;
        }
    }
    if (i0 >= 0) {
        const DevGeomHot CRT_CONST* g = &sc->hot[i0];
        f.s[0] = make_sph_slot(P, ld3c(g->o), g->n[0], f.Bp0, lf, f.pa, a_max);
        f.sph_i[0] = i0;
        f.sph_glass[0] = g->mat_type == CRT_MAT_REFRACTIVE;
        f.rest_mask &= ~(1ull << i0);
    }
    if (i1 >= 0) {
        const DevGeomHot CRT_CONST* g = &sc->hot[i1];
        f.s[1] = make_sph_slot(P, ld3c(g->o), g->n[0], f.Bp0, lf, f.pa, a_max);
        f.sph_i[1] = i1;
        f.sph_glass[1] = g->mat_type == CRT_MAT_REFRACTIVE;
        f.rest_mask &= ~(1ull << i1);
    }
    if (iq >= 0) {
        const DevGeomHot CRT_CONST* g = &sc->hot[iq];
        const DevGeomBox CRT_CONST* bx = &sc->box[iq];
        f.quad_par = sc->quad_par[li][iq] != 0u;
        if (f.quad_par) {
            f.qp = make_quad_par_slot(P, ld3c(g->o), ld3c(g->n), bx->sp_min2, bx->n2, bx->noise, ld3c(bx->mid), f.Bp0, lf);
            f.hout = ld3c(bx->half_out);
            f.hin = ld3c(bx->half_in);
        } else {
            f.q = make_quad_slot(P, ld3c(g->o), ld3c(g->n), bx->sp_min2, bx->noise, f.Bp0, lf);
        }
        f.blo = ld3c(bx->lo);
        f.bhi = ld3c(bx->hi);
        f.ilo = ld3c(bx->lo_in);
        f.ihi = ld3c(bx->hi_in);
        f.n2 = bx->n2;
        f.nz = bx->noise;
        f.quad_i = iq;
        f.quad_glass = g->mat_type == CRT_MAT_REFRACTIVE;
        f.rest_mask &= ~(1ull << iq);
    }
    return f;
}

// SLOTS_ONLY: the caller has checked (wave-uniformly) that f.poly holds and f.rest_mask is empty.
template <int NB, bool SLOTS_ONLY>
DI Classes classify_samples(ScenePtr sc, const LightF& lf, const RecordFilter& f, const float (&us)[NB],
                            const float (&vs)[NB]) {
    Classes k = {0u, 0u, 0u};
    if (!SLOTS_ONLY && f.rest_mask != 0) {
        V3 D[NB];
        float a[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            D[j] = sample_offset(lf, f.Bp0, f.Bp1, us[j], vs[j]);
            a[j] = fdot(D[j], D[j]);
        }
        k = classify_block<NB>(sc, f.rest_mask, f.P, D, a);
    }
    if (SLOTS_ONLY || f.poly) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (f.sph_i[n] < 0) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const uint32_t c = sph_slot_class(f.s[n], us[j], vs[j]);
                if (f.sph_glass[n]) {
                    if (c != 0u) {
                        k.maybe_any |= 1u << j;
                        k.maybe_glass |= 1u << j;
                    }
                } else {
                    if (c == 1u) k.maybe_any |= 1u << j;
                    if (c == 2u) k.sure_opaque |= 1u << j;
                }
            }
        }
        if (f.quad_i >= 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const uint32_t c = f.quad_par ? quad_par_slot_class(f.qp, f.hout, f.hin, f.pa, lf, us[j], vs[j])
                                              : quad_slot_class(f.q, f.blo, f.bhi, f.ilo, f.ihi, f.n2, f.nz, f.pa, lf, us[j], vs[j]);
                if (f.quad_glass) {
                    if (c != 0u) {
                        k.maybe_any |= 1u << j;
                        k.maybe_glass |= 1u << j;
                    }
                } else {
                    if (c == 1u) k.maybe_any |= 1u << j;
                    if (c == 2u) k.sure_opaque |= 1u << j;
                }
            }
        }
        if (!f.rec_ok) { // the shaded point is (almost) in the light's plane: decide nothing
            k.maybe_any = (1u << NB) - 1u;
            k.sure_opaque = 0u;
        }
    }
    return k;
}

template <int NB, bool SLOTS_ONLY>
DI void filter_block(const WfArgs& w, ScenePtr sc, const LightF& lf, const RecordFilter& f, bool act, Rng& rng,
                     uint32_t key, uint32_t& shadowed, uint32_t (*stg)[kStageCap], Stage& st, uint32_t lane,
                     uint32_t& tally) {
    float us[NB], vs[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        us[j] = rng_uniform(rng);
        vs[j] = rng_uniform(rng);
    }
    const Classes k = classify_samples<NB, SLOTS_ONLY>(sc, lf, f, us, vs);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const bool so = (k.sure_opaque >> j) & 1u, ma = (k.maybe_any >> j) & 1u, mg = (k.maybe_glass >> j) & 1u;
        const bool covered = act && so && !mg; // the closest hit is opaque whichever primitive it is
        const bool unresolved = act && !covered && (ma || so);
        if (covered) ++shadowed;
        const uint64_t bm = __ballot(unresolved);
        if (bm != 0) {
            if (unresolved) {
                const uint32_t pos = (st.head + st.fill + lane_rank(bm)) & (kStageCap - 1u);
                stg[0][pos] = key;
                stg[1][pos] = f2u(us[j]);
                stg[2][pos] = f2u(vs[j]);
            }
            st.fill += (uint32_t)__popcll(bm);
            tally += (uint32_t)__popcll(bm);
            if (st.fill >= 64u) stage_flush(w, stg, st, 64u, lane);
        }
    }
}

// Decision for all samples of one (shaded point, light) at once: 1 = every shadow ray is free,
// 2 = every shadow ray is blocked by an opaque primitive, 0 = the samples have to be looked at one by
// one.  Uses the slot polynomials: over the unit square of draws a polynomial c0 + cu s + cv t + A s^2 +
// B st + C t^2 (s, t in [-1/2, 1/2] around the centre) lies within c0 +- (|cu| + |cv|)/2 +- the
// quadratic terms' extremes, affine functions take their extremes at the four corners, and a = D.D is
// convex.  Only parallelogram lights, spheres and quads parallel to the light are decided here.
DI uint32_t record_light_class(ScenePtr sc, uint32_t li, const LightF& lf, V3 P) {
    if (!lf.is_par) return 0u;
    const V3 Bp = lf.B0 - P;
    PolyA pa;
    pa.h0 = fdot(Bp, Bp);
    pa.h1x2 = 2.0f * fdot(Bp, lf.e01);
    pa.h2x2 = 2.0f * fdot(Bp, lf.e03);
    const float a10 = pa.h0 + pa.h1x2 + lf.g11, a01 = pa.h0 + pa.h2x2 + lf.g22;
    const float a11 = a10 + pa.h2x2 + lf.g22 + lf.g12x2;
    const float a_max = __builtin_fmaxf(__builtin_fmaxf(pa.h0, a10), __builtin_fmaxf(a01, a11));
    const float pl = fdot(lf.nor, Bp);
    const float nl2 = fdot(lf.nor, lf.nor);
    const bool rec_ok = pl * pl > 1.0001e-6f * nl2 && a_max < 1e12f && nl2 > 1e-30f && nl2 < 1e30f;
    if (!rec_ok) return 0u;
    // lower bound of a: exact minimum over the square for a rectangle, else the squared distance to the light's plane
    float a_min = fdiv(pl * pl, nl2);
    if (lf.g12x2 * lf.g12x2 <= 1e-12f * (lf.g11 * lf.g22)) {
        const float us = __builtin_fminf(__builtin_fmaxf(fdiv(-0.5f * pa.h1x2, lf.g11), 0.0f), 1.0f);
        const float vs = __builtin_fminf(__builtin_fmaxf(fdiv(-0.5f * pa.h2x2, lf.g22), 0.0f), 1.0f);
        const float am = __builtin_fmaf(__builtin_fmaf(lf.g11, us, pa.h1x2), us, __builtin_fmaf(__builtin_fmaf(lf.g22, vs, pa.h2x2), vs, pa.h0));
        a_min = __builtin_fmaxf(a_min, am);
    }
    a_min *= 0.9999f;
    const uint64_t cand = shadow_candidates(sc, li, P);
    bool undecided = false, have_sure = false;
    const uint32_t n_geom = sc->n_geom;
    for (uint32_t i = 0; i < n_geom; ++i) {
        const bool mine = (cand >> i) & 1ull;
        if (!__any(mine)) continue;
        const DevGeomHot CRT_CONST* g = &sc->hot[i];
        const bool glass = g->mat_type == CRT_MAT_REFRACTIVE;
        bool all_free = false, all_sure = false;
        if (g->kind == CRT_GEOM_SPHERE) {
            const SphSlot s = make_sph_slot(P, ld3c(g->o), g->n[0], Bp, lf, pa, a_max);
            const float q00 = s.b0, q10 = s.b0 + s.b1, q01 = s.b0 + s.b2, q11 = q10 + s.b2;
            const float bqmin = __builtin_fminf(__builtin_fminf(q00, q10), __builtin_fminf(q01, q11));
            const float bqmax = __builtin_fmaxf(__builtin_fmaxf(q00, q10), __builtin_fmaxf(q01, q11));
            // disc around the centre of the square
            const float c0 = __builtin_fmaf(0.25f, s.A + s.B + s.C, __builtin_fmaf(0.5f, s.Du + s.Ev, s.F));
            const float cu = s.A + 0.5f * s.B + s.Du, cv = s.C + 0.5f * s.B + s.Ev;
            const float lin = 0.5f * (__builtin_fabsf(cu) + __builtin_fabsf(cv));
            const float qb = 0.25f * __builtin_fabsf(s.B);
            const float ub = c0 + lin + qb + 0.25f * (__builtin_fmaxf(s.A, 0.0f) + __builtin_fmaxf(s.C, 0.0f));
            const float lb = c0 - lin - qb + 0.25f * (__builtin_fminf(s.A, 0.0f) + __builtin_fminf(s.C, 0.0f));
            all_free = bqmin > s.kb || ub < -s.M;
            all_sure = lb > s.M2 && bqmax < -s.kc;
        } else if (sc->quad_par[li][i] != 0u) {
            const DevGeomBox CRT_CONST* bx = &sc->box[i];
            const QuadParSlot q = make_quad_par_slot(P, ld3c(g->o), ld3c(g->n), bx->sp_min2, bx->n2, bx->noise, ld3c(bx->mid), Bp, lf);
            const float hout[3] = {bx->half_out[0], bx->half_out[1], bx->half_out[2]};
            const float hin[3] = {bx->half_in[0], bx->half_in[1], bx->half_in[2]};
            bool out_all = false, in_all = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float r00 = q.r[k][0], r10 = r00 + q.r[k][1], r01 = r00 + q.r[k][2], r11 = r10 + q.r[k][2];
                const float rmin = __builtin_fminf(__builtin_fminf(r00, r10), __builtin_fminf(r01, r11));
                const float rmax = __builtin_fmaxf(__builtin_fmaxf(r00, r10), __builtin_fmaxf(r01, r11));
                out_all = out_all || rmin > hout[k] || rmax < -hout[k];
                in_all = in_all && rmin > -hin[k] && rmax < hin[k];
            }
            const bool steep_all = a_max < q.t_steep;
            all_free = q.behind || a_max < q.t_lift || a_min > q.t_beyond || (out_all && steep_all);
            all_sure = in_all && steep_all && a_min > q.t_between;
        }
        if (mine) {
            if (glass) {
                if (!all_free) undecided = true;
            } else {
                if (all_sure) have_sure = true;
                else if (!all_free) undecided = true;
            }
        }
    }
    if (undecided) return 0u;
    return have_sure ? 2u : 1u;
}

// Hands a wave the groups (of 64 records or rays) of a launch, a few at a time: cursor set `set` of the counter
// block, one cursor per partition, partition p = the groups g with g % WF_CURSOR_PARTS == p.  A wave starts at the
// partition of its workgroup and moves on to the next three when one is used up, so the waves end together although
// the groups' costs differ widely (a fixed stride left the last waves running alone; a wave that looked into every
// partition before it ends would add tens of thousands of atomics to a small launch).
struct GroupFeed {
    uint32_t part, parts_left, next, end, chunk, n_groups;
    uint32_t* cursors;
};
DI GroupFeed feed_begin(const WfArgs& w, uint32_t set, uint32_t n_groups) {
    GroupFeed f;
    f.part = blockIdx.x % WF_CURSOR_PARTS;
    f.parts_left = 4u; // its own partition and the next three: every partition is the same sample of the launch's groups
    f.next = f.end = 0u;
    uint32_t chunk = n_groups / (gridDim.x * 8u); // about eight turns per wave
    f.chunk = chunk < 1u ? 1u : (chunk > 8u ? 8u : chunk);
    f.n_groups = n_groups;
    f.cursors = w.ctrs + WF_CTR_COUNT + (size_t)set * WF_CURSOR_PARTS * WF_CURSOR_STRIDE;
    return f;
}
DI bool feed_next(GroupFeed& f, uint32_t lane, uint32_t& g) { // wave-uniform
    while (f.next == f.end) {
        if (f.parts_left == 0u) return false;
        const uint32_t in_part = f.part < f.n_groups ? (f.n_groups - f.part + WF_CURSOR_PARTS - 1u) / WF_CURSOR_PARTS : 0u;
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&f.cursors[f.part * WF_CURSOR_STRIDE], f.chunk);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if (c < in_part) {
            f.next = c;
            f.end = c + f.chunk < in_part ? c + f.chunk : in_part;
        } else {
            f.part = (f.part + 1u) % WF_CURSOR_PARTS;
            --f.parts_left;
        }
    }
    g = f.next * WF_CURSOR_PARTS + f.part;
    ++f.next;
    return true;
}

// ---------------------------------------------------------------------------------------- trace
#ifndef CRT_TRACE_BLOCK_WAVES
#define CRT_TRACE_BLOCK_WAVES 4
#endif
constexpr uint32_t kTraceWaves = CRT_TRACE_BLOCK_WAVES; // waves per trace workgroup (they share the generator jump tables in LDS)
// Cooperative expansion takes at most 63 children of a pixel at a time.  The k-th diffuse hit among them starts k
// shaded points' worth of draws after the first; its record keeps the pixel's generator state and k (bits 8..13 of
// the primitive word), and whoever draws from it (the filter kernel, for the fifth of the records whose samples it
// has to look at) jumps there with table k-1 of 63 in device memory.  The pixel itself continues from the state
// after all of those hits; 63 children keep every ordinal and the hit count within the 63 tables.
constexpr uint32_t kCoopBatch = 63, kCoopTables = 63;
DI void ordinal_jump(Rng& rng, uint32_t k, const WfArgs& w) { // k shaded points ahead; k = 0 for records of the per-pixel walk
    if (k != 0u && w.jump_n[0] != 0u) rng_jump(rng, w.coop_tabs + (size_t)(k - 1u) * kJumpWords, k * w.jump_n[0] * 362437u);
}
// In the cooperative kernel a wave owns 8 of its unit's 64 pixels (one row of the tile) and kCoopSplit
// waves share the unit: the batches of one wave run back to back, so fewer owners per wave means more
// of them in flight.
constexpr uint32_t kCoopSplit = 8;
constexpr uint32_t kCoopOwners = 64u / kCoopSplit;
constexpr uint32_t kCoopList = 512; // >= kCoopOwners * kCoopBatch: children one step of a wave can have in its list
static_assert(kCoopOwners * kCoopBatch <= kCoopList, "list positions");
static_assert(kCoopOwners <= (1u << (32u - kRecBits)), "a record index and its owner share a word");
// bits [lo, hi) set in a bit array
DI uint32_t count_bits(const uint32_t* bits, uint32_t lo, uint32_t hi) {
    uint32_t n = 0;
    for (uint32_t wd = lo >> 5; wd <= (hi >> 5) && wd < kCoopList / 32u; ++wd) {
        uint32_t m = bits[wd];
        if (wd == (lo >> 5)) m &= ~0u << (lo & 31u);
        if (wd == (hi >> 5)) m &= (hi & 31u) ? ~(~0u << (hi & 31u)) : 0u;
        n += (uint32_t)__popc(m);
    }
    return n;
}
// what the cooperative expansion keeps per owner in LDS while the wave works through the owners' children
enum : uint32_t { OW_PRE = 0, OW_N, OW_LP, OW_NFRAMES, OW_CNT, OW_REM, OW_RNG, OW_RAY_D = OW_RNG + 6, OW_LANE = OW_RAY_D + 3, OW_SLOTS, OW_RECS, OW_WORDS };
DI uint64_t lanes_below(uint32_t k) { return k >= 64u ? ~0ull : (1ull << k) - 1ull; }

// XOR over the 64 lanes of a wave, returned to every lane (wave-uniform): within each row of 16 lanes by DPP
// shifts (a lane without a source reads 0), then the rows' last lanes.
DI uint32_t wave_xor(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 15) ^ (uint32_t)__builtin_amdgcn_readlane((int)x, 31) ^
           (uint32_t)__builtin_amdgcn_readlane((int)x, 47) ^ (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// rng_jump of ONE generator state by the whole wave: the state's five words are wave-uniform, lane j < 40 looks up
// the image of nibble j, the images are XORed across the lanes.  A lane that walks its pixel needs such a jump only
// when it opens a frame (one lane in a hundred per step at gl_iter 1, one per step and wave at gl_iter 2); done by
// that lane alone it costs the wave 40 look-ups of its own, done by the wave one.  The Weyl word is the caller's.
DI void wave_jump(uint32_t (&v)[5], const uint32_t* tab, uint32_t lane) {
    const uint32_t wd = lane >> 3;
    const uint32_t word = wd == 0u ? v[0] : (wd == 1u ? v[1] : (wd == 2u ? v[2] : (wd == 3u ? v[3] : v[4])));
    const uint32_t idx = (lane < 40u ? lane : 0u) * 16u + ((word >> (4u * (lane & 7u))) & 15u);
    uint4 q = reinterpret_cast<const uint4*>(tab)[idx];
    uint32_t q4 = tab[4u * kJumpEntries + idx];
    if (lane >= 40u) {
        q = make_uint4(0u, 0u, 0u, 0u);
        q4 = 0u;
    }
    v[0] = wave_xor(q.x);
    v[1] = wave_xor(q.y);
    v[2] = wave_xor(q.z);
    v[3] = wave_xor(q.w);
    v[4] = wave_xor(q4);
}

// rng_jump with the state cut into 32 chunks of five bits instead of 40 of four: 32 look-ups per jump, a table of
// 32 x 32 entries (20 KB, what a trace workgroup's LDS has room for next to the primitive table).  Entry
// e = chunk * 32 + value: words 0..3 at tab[4e .. 4e+3], word 4 at tab[4 * 1024 + e]; chunk c is bits [5c, 5c+5) of
// the 160-bit string v0 | v1 << 32 | ..., four of the chunks straddle two words.
#ifndef CRT_JUMP5
#define CRT_JUMP5 1
#endif
constexpr uint32_t kJump5Entries = 32u * 32u, kJump5Words = 5u * kJump5Entries;
constexpr uint32_t kTraceJumpWords = CRT_JUMP5 ? kJump5Words : kJumpWords; // the table the trace kernels keep in LDS
DI void rng_jump5(Rng& s, const uint32_t* tab, uint32_t weyl_step) {
    const uint32_t in[6] = {s.v0, s.v1, s.v2, s.v3, s.v4, 0u};
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
    const uint4* tab4 = reinterpret_cast<const uint4*>(tab);
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const int wd = (5 * c) >> 5, sh = (5 * c) & 31;
        const uint32_t bits = sh <= 27 ? in[wd] >> sh : __builtin_amdgcn_alignbit(in[wd + 1], in[wd], (uint32_t)sh);
        const uint32_t idx = (uint32_t)c * 32u + (bits & 31u);
        const uint4 q = tab4[idx];
        o0 ^= q.x;
        o1 ^= q.y;
        o2 ^= q.z;
        o3 ^= q.w;
        o4 ^= tab[4u * kJump5Entries + idx];
        // the sums are made here, not folded into whatever uses the state next: without this (and with it behind the
        // loop instead of in its last pass) the compiler keeps the 32 look-ups' results alive past the jump, 128
        // registers and spills instead of 95
        if (c == 31) asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(o4));
    }
    s = Rng{o0, o1, o2, o3, o4, s.d + weyl_step};
}

// Children of the innermost frame leave last-to-first (main.cu:176-191 pushes, :132 pops):
// get_random_ray_values_in_hemisphere (main.cu:60-74) and the child's attenuation (main.cu:186).
DI void child_ray(V3 hp, V3 hn, V4 patt, V4 pdiff, float r1, float r2, float inv_ind, V3& o, V3& d, V4& att) {
    V3 Nx, Nz;
    surface_frame(hn, Nx, Nz);
    const float PI_2 = 6.28318530717958647692f;
    float sinTheta = 1.0f - r1 * r1;
    float phi = r2 * PI_2;
    float sp, cp;
    sincos_c(phi, sp, cp);
    float x = sinTheta * cp, z = sinTheta * sp;
    V3 wd = mk3(x * Nz.x + r1 * hn.x + z * Nx.x, x * Nz.y + r1 * hn.y + z * Nx.y, x * Nz.z + r1 * hn.z + z * Nx.z);
    o = hp + wd * 0.01f;
    d = wd;
    att = ((pdiff * (2.0f * r1)) * patt) * inv_ind;
}

// What the tree walk reads of a primitive by per-lane index, copied to LDS once per workgroup: a lone
// wave spends most of a step waiting, and an LDS read returns in a quarter of the time of a cached
// global one.  Row layout: vertex normals [0..11], Material::diffuse [12..15], refractIndex [16],
// kind [17], sphere centre / quad corner 0 [18..20], Material::ambient rgb [21..23], the quad's (u,v) constants
// (DevGeomUV: idu, idv, linear, nor_id, ka, kb, kg, k2) [24..34].
constexpr uint32_t kPrimWords = 36;
typedef float PrimRow[kPrimWords];
typedef const PrimRow* PrimTable; // n_geom rows
DI void fill_prim_table(ScenePtr sc, PrimRow* t, uint32_t tid, uint32_t nthreads) {
    const uint32_t n = sc->n_geom;
    for (uint32_t k = tid; k < n * kPrimWords; k += nthreads) {
        const uint32_t i = k / kPrimWords, wd = k % kPrimWords;
        float v = 0.0f;
        if (wd < 12u) v = sc->shade[i].n[wd / 3u][wd % 3u];
        else if (wd < 16u) v = sc->shade[i].diffuse[wd - 12u];
        else if (wd == 16u) v = sc->shade[i].refract_index;
        else if (wd == 17u) v = u2f(sc->hot[i].kind);
        else if (wd < 21u) v = sc->hot[i].o[wd - 18u];
        else if (wd < 24u) v = sc->shade[i].ambient[wd - 21u];
        else if (wd == 24u) v = u2f(sc->uv[i].idu);
        else if (wd == 25u) v = u2f(sc->uv[i].idv);
        else if (wd == 26u) v = u2f(sc->uv[i].linear);
        else if (wd == 27u) v = sc->uv[i].nor_id;
        else if (wd < 30u) v = sc->uv[i].ka[wd - 28u];
        else if (wd < 32u) v = sc->uv[i].kb[wd - 30u];
        else if (wd < 34u) v = sc->uv[i].kg[wd - 32u];
        else if (wd == 34u) v = sc->uv[i].k2;
        t[i][wd] = v;
    }
}

// One pop of color()'s loop (main.cu:132-221) for a ray with bounces left, without the shading.
enum : uint32_t { POP_MISS = 0, POP_DIFFUSE = 1, POP_BOUNCE = 2 };
DI uint32_t pop_ray(ScenePtr sc, PrimTable prim, V3& cur_o, V3& cur_d, V4& cur_att, uint32_t& cur_ref, V3& ray_d,
                    V3& P, V3& N, uint32_t& geom) {
    if (!eps_equal(len_sq(cur_d), 0.0f, kEpsSq)) ray_d = normalize3(cur_d);
    const V3 ro = cur_o;
    bool uv_pending = false;
#ifdef CRT_NO_CLASSIFY
    Hit h = closest_hit(sc, ro, ray_d, 0.0f);
#else
    Hit h = closest_hit_classified(sc, ro, ray_d, 0.0f, uv_pending);
#endif
    if (h.idx < 0) return POP_MISS;
    const float* pr = prim[h.idx];
    if (uv_pending) { // the final hit was accepted by where its point lies: its (u,v) for the normal below
        UVRec q;
        q.idu = f2u(pr[24]);
        q.idv = f2u(pr[25]);
        q.linear = f2u(pr[26]);
        q.nor_id = pr[27];
        q.ka[0] = pr[28];
        q.ka[1] = pr[29];
        q.kb[0] = pr[30];
        q.kb[1] = pr[31];
        q.kg[0] = pr[32];
        q.kg[1] = pr[33];
        q.k2 = pr[34];
        (void)quad_uv<false>(q, ro - mk3(pr[18], pr[19], pr[20]), ray_d, h.t, h.u, h.v);
    }
    // hit point and shading normal (Sphere.hpp:102-107, Quad.hpp:299-325)
    P = ro + ray_d * h.t;
    if (f2u(pr[17]) == CRT_GEOM_SPHERE) {
        PROF_LANES(6);
        N = normalize3(P - mk3(pr[18], pr[19], pr[20]));
    } else {
        PROF_LANES(7);
        const float u = h.u, v = h.v;
        const float l0 = (1 - u) * (1 - v), l1 = u * (1 - v), l2 = u * v, l3 = (1 - u) * v;
        N = V3{l0 * pr[0] + l1 * pr[3] + l2 * pr[6] + l3 * pr[9], l0 * pr[1] + l1 * pr[4] + l2 * pr[7] + l3 * pr[10],
               l0 * pr[2] + l1 * pr[5] + l2 * pr[8] + l3 * pr[11]};
    }
    geom = (uint32_t)h.idx;
    if (h.mat_type == CRT_MAT_DIFFUSE) return POP_DIFFUSE;
    if (h.mat_type == CRT_MAT_REFLECT) {
        PROF_LANES(8);
        // get_reflect_ray (main.cu:76-81)
        V3 I = normalize3(ray_d);
        V3 rdir = reflect3(I, normalize3(N));
        cur_o = P + rdir * 0.01f;
        cur_d = rdir;
    } else {
        // get_refraction_ray (main.cu:83-108)
        PROF_LANES(9);
        bool front_face = dot3(ray_d, N) < 0.0f;
        V3 norm = front_face ? N : neg(N);
        float ratio = pr[16] / 1.0f;
        if (front_face) ratio = rcp_exact(ratio);
        V3 rdn = normalize3(ray_d);
        float cos_theta = __builtin_fminf(dot3(neg(rdn), norm), 1.0f);
        float sin_theta = rcp_exact(rsqrt_c(1.0f - cos_theta * cos_theta));
        bool cannot_refract = ratio * sin_theta > 1.0f || reflectance_over_one(cos_theta, ratio); // main.cu:97
        V3 dir = cannot_refract ? reflect3(rdn, norm) : refract3(rdn, norm, ratio);
        cur_o = P + dir * 0.01f;
        cur_d = dir;
    }
    cur_att = cur_att * V4{pr[12], pr[13], pr[14], pr[15]};
    cur_ref = cur_ref - 1u;
    return POP_BOUNCE;
}

// A diffuse hit that no light can do more for than add the material's ambient term needs neither shadow
// rays nor the LTC integrals: LightList::calculateColor (LightList.hpp:23-36) then sums, per light,
// vec4(ambient * light colour, 1) -- an uninitialised area light returns just that (AreaLight.hpp:586-589), and
// a one-sided area light that the point is not "behind" has both polygon sums set to zero
// (AreaLight.hpp:249-251,292-296), which leaves ambient + ((0 + 0) * intensity) * visibility = ambient for the
// lights and materials DevLight::dark_lo/hi admits.  In the Cornell box that is every hit on the ceiling.  The
// hit still counts as a shaded point and still moves the generator past its light samples; it just writes
// attenuation x that sum into its result slot now instead of becoming a record.  Returns false when some
// light needs the full evaluation.
DI bool ambient_only(ScenePtr sc, PrimTable prim, V3 P, uint32_t geom, V4 att, float4& out) {
    if (sc->dark_possible == 0u) return false;
    const float* pr = prim[geom];
    const V3 mat_ambient = mk3(pr[21], pr[22], pr[23]);
    V4 direct = V4{0, 0, 0, 0};
    bool all = true;
    const uint32_t n_lights = sc->n_lights;
    for (uint32_t li = 0; li < n_lights; ++li) {
        const DevLight CRT_CONST* L = &sc->lights[li];
        if (L->initialized) {
            const V3 w0 = ld3c(L->pts[0]) - P;
            const bool behind = dot3(w0, ld3c(L->light_normal)) < 0.0f;
            const uint32_t word = geom < 32u ? L->dark_lo : L->dark_hi;
            if (behind || ((word >> (geom & 31u)) & 1u) == 0u) all = false;
        }
        const V3 amb = mat_ambient * ld3c(L->color);
        direct = direct + V4{amb.x, amb.y, amb.z, 1.0f};
    }
    const V4 c = att * direct; // final_color += attenuation * light (main.cu:195)
    out = make_float4(c.x, c.y, c.z, c.w);
    return all;
}

// Stores one shade record.
DI void store_record(const WfArgs& w, uint32_t wi, V3 P, V3 N, V3 view, V4 att, uint32_t geom, const Rng& rng, uint32_t slot,
                     uint32_t ordinal = 0u) {
    PROF_LANES(11);
    if (wi < w.cap_work) {
        store_streaming(rec_point(w) + wi, make_uint4(f2u(P.x), f2u(P.y), f2u(P.z), geom | (ordinal << 8)));
        const size_t cw_ = w.cap_work;
        uint4* sf = rec_surface(w) + wi;
        store_streaming(sf, make_uint4(f2u(N.x), f2u(N.y), f2u(N.z), slot));
        store_streaming(sf + cw_, make_uint4(f2u(view.x), f2u(view.y), f2u(view.z), f2u(att.w)));
        store_streaming(sf + 2 * cw_, make_uint4(f2u(att.x), f2u(att.y), f2u(att.z), 0u));
        uint4* gn = rec_gen(w) + wi;
        store_streaming(gn, make_uint4(rng.v0, rng.v1, rng.v2, rng.v3));
        store_streaming(gn + cw_, make_uint4(rng.v4, rng.d, 0u, 0u));
    } else {
        w.ctrs[WF_CTR_ERROR] = 1u;
    }
}

// The same without the word that says what the record is (primitive | ordinal): the cooperative kernel writes that
// word at the end of its step (disjoint bytes: the two stores cannot overtake each other's data).
DI void store_record_body(const WfArgs& w, uint32_t wi, V3 P, V3 N, V3 view, V4 att, const Rng& rng, uint32_t slot) {
    if (wi < w.cap_work) {
        uint32_t* pt = reinterpret_cast<uint32_t*>(rec_point(w) + wi);
        __builtin_nontemporal_store(f2u(P.x), pt + 0);
        __builtin_nontemporal_store(f2u(P.y), pt + 1);
        __builtin_nontemporal_store(f2u(P.z), pt + 2);
        const size_t cw_ = w.cap_work;
        uint4* sf = rec_surface(w) + wi;
        store_streaming(sf, make_uint4(f2u(N.x), f2u(N.y), f2u(N.z), slot));
        store_streaming(sf + cw_, make_uint4(f2u(view.x), f2u(view.y), f2u(view.z), f2u(att.w)));
        store_streaming(sf + 2 * cw_, make_uint4(f2u(att.x), f2u(att.y), f2u(att.z), 0u));
        uint4* gn = rec_gen(w) + wi;
        store_streaming(gn, make_uint4(rng.v0, rng.v1, rng.v2, rng.v3));
        store_streaming(gn + cw_, make_uint4(rng.v4, rng.d, 0u, 0u));
    } else {
        w.ctrs[WF_CTR_ERROR] = 1u;
    }
}

// The wave's cursor into the shade-record array: records are handed out in chunks of kWorkChunk
// consecutive indices, so the one global counter sees one atomic per chunk instead of one per step.
struct WorkCursor {
    uint32_t base, used; // wave-uniform
    bool have;
};
// Record index for every lane of `em` (all lanes call this together).
DI uint32_t reserve_records(const WfArgs& w, WorkCursor& wc, uint64_t em, uint32_t lane) {
    const uint32_t n_emit = (uint32_t)__popcll(em);
    const bool need_chunk = wc.used + n_emit > kWorkChunk;
    uint32_t new_base = 0;
    if (need_chunk) {
        if (lane == (uint32_t)__builtin_ctzll(em)) new_base = w.work_static + atomicAdd(&w.ctrs[WF_CTR_WORK], kWorkChunk);
        new_base = (uint32_t)__builtin_amdgcn_readlane((int)new_base, __builtin_ctzll(em));
    }
    const uint32_t my_pos = wc.used + lane_rank(em);
    const uint32_t my_wi = my_pos < kWorkChunk ? wc.base + my_pos : new_base + (my_pos - kWorkChunk);
    if (need_chunk) {
        wc.base = new_base;
        wc.used = wc.used + n_emit - kWorkChunk;
        wc.have = true;
    } else {
        wc.used += n_emit;
    }
    return my_wi;
}

// get_world_coordinates (main.cu:306-314), pixel_size (main.cu:337) and Camera::getRay (Camera.hpp:86-144).
struct CameraRays {
    V3 pos, front, right, up, eye;
    uint32_t type;
    float ortho, pix_w, pix_h, width_world;
    DI void ray(float rx, float ry, V3& ro, V3& rd) const {
        if (type == CRT_CAM_ORTHOGRAPHIC) {
            ro = ((right * rx) * ortho + (up * ry) * ortho) + pos;
            rd = ray_dir(front);
        } else {
            ro = (right * rx + up * ry) + pos;
            rd = ray_dir(ro - eye);
        }
    }
    // centre of pixel (i, j): remap<float,true> (math_functions.hpp:31-45)
    DI void pixel_centre(const RenderArgs& a, uint32_t i, uint32_t j, float& x_world, float& y_world) const {
        float denom = (float)a.nx - 0.0f;
        float es = -width_world * 0.5f, ee = width_world * 0.5f;
        x_world = (denom == 0.0f) ? (float)i : es + ((ee - es) / denom) * ((float)i - 0.0f);
        float denomy = (float)a.ny - 0.0f;
        y_world = (denomy == 0.0f) ? (float)j : 1.0f + ((-1.0f - 1.0f) / denomy) * ((float)j - 0.0f);
    }
};
DI CameraRays load_camera(ScenePtr sc, const RenderArgs& a) {
    CameraRays c;
    const float height_world = 2.0f;
    c.width_world = height_world * ((float)a.nx / (float)a.ny);
    c.pix_w = c.width_world / (float)a.nx;
    c.pix_h = height_world / (float)a.ny;
    c.pos = ld3c(sc->cam.pos);
    c.front = ld3c(sc->cam.front);
    c.right = ld3c(sc->cam.right);
    c.up = ld3c(sc->cam.up);
    c.eye = c.pos - c.front * (c.width_world * 0.5f * sc->cam.one_over_tan); // Camera.hpp:97-98,133-134
    c.type = sc->cam.type;
    float aspect = height_world / c.width_world;
    c.ortho = sc->cam.ortho_scale * (aspect > 1.0f ? aspect : 1.0f);
    return c;
}

// Launch order of the trace kernel's work units.  A wave lasts as long as its slowest pixel, and a pixel's
// walk is one dependent chain, so a tile of expensive pixels that starts last leaves the GPU idle behind
// it.  This kernel gives every unit a cost estimate: the primary ray of each pixel centre is followed to
// its first diffuse hit, and the estimate grows with the part of the hemisphere there that mirrors and
// glass cover (every GI child that meets one takes up to ref_iter more steps).  Ordering only; no result
// depends on it.
__global__ void __launch_bounds__(64) wf_cost_kernel(WfArgs w, uint32_t* unit_cost) {
    const RenderArgs& a = w.r;
    ScenePtr sc = (ScenePtr)a.scene;
    const uint32_t lane = threadIdx.x, unit = blockIdx.x;
    __shared__ __attribute__((aligned(16))) PrimRow s_prim[CRT_MAX_GEOMETRY];
    fill_prim_table(sc, s_prim, lane, 64u);
    __syncthreads();
    const PixelMap pm = map_pixel(a, unit, lane);
    const CameraRays cam = load_camera(sc, a);
    float cost = 0.0f;
    if (pm.valid) {
        float xw, yw;
        cam.pixel_centre(a, pm.i, pm.j, xw, yw);
        V3 o, d, ray_d = mk3(0.0f, 0.0f, -1.0f), P = mk3(0, 0, 0), N = mk3(0, 0, 1);
        cam.ray(xw, yw, o, d);
        V4 att = V4{1, 1, 1, 1};
        uint32_t ref = a.ref_iter, geom = 0, what = POP_MISS;
        float steps = 0.0f;
        while (ref != 0u) {
            what = pop_ray(sc, s_prim, o, d, att, ref, ray_d, P, N, geom);
            steps += 1.0f;
            if (what != POP_BOUNCE) break;
        }
        cost = steps;
        if (what == POP_DIFFUSE && a.gl_iter != 0u) {
            float covered = 0.0f; // fraction of the hemisphere over P that bounces a child ray on
            const uint32_t n = sc->n_geom;
            for (uint32_t i = 0; i < n; ++i) {
                const DevGeomHot CRT_CONST* g = &sc->hot[i];
                if (g->mat_type == CRT_MAT_DIFFUSE) continue;
                if (g->kind == CRT_GEOM_SPHERE) {
                    const V3 c = ld3c(g->o) - P;
                    const float d2 = fdot(c, c) + 1e-12f, r2 = g->n[0];
                    const float dist = __builtin_amdgcn_sqrtf(d2), r = __builtin_amdgcn_sqrtf(r2);
                    const float omega = 1.0f - __builtin_amdgcn_sqrtf(__builtin_fmaxf(0.0f, 1.0f - fdiv(r2, d2)));
                    const float above = __builtin_fminf(__builtin_fmaxf(fdiv(fdot(N, c) + r, dist + r), 0.0f), 1.0f);
                    covered += omega * above;
                } else {
                    const DevGeomBox CRT_CONST* bx = &sc->box[i];
                    const V3 c = ld3c(g->o) - P, nq = ld3c(g->n); // |nq| = area of the spanning parallelogram
                    const float d2 = fdot(c, c) + 1e-12f;
                    const float area_cos = __builtin_fabsf(fdot(nq, c)) * __builtin_amdgcn_rsqf(d2);
                    covered += __builtin_fminf(fdiv(area_cos, 6.2831853f * d2), 0.5f) * (bx->n2 > 0.0f ? 1.0f : 0.0f);
                }
            }
            covered = __builtin_fminf(covered, 1.0f);
            cost += (float)a.ind_rays * (1.0f + covered * (float)a.ref_iter);
        }
    }
    for (int off = 32; off > 0; off >>= 1) cost = __builtin_fmaxf(cost, __shfl_xor(cost, off, 64));
    if (lane == 0) unit_cost[unit] = f2u(cost);
}

#ifdef CRT_TRACE_MAXWAVES
#define CRT_TRACE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(CRT_TRACE_MAXWAVES, CRT_TRACE_MAXWAVES)))
#else
#define CRT_TRACE_WAVES_ATTR
#endif
#ifndef CRT_COOP_WAVES
#define CRT_COOP_WAVES 4
#endif
template <bool COOP>
__global__ void CRT_TRACE_WAVES_ATTR __launch_bounds__(64 * kTraceWaves, COOP ? CRT_COOP_WAVES : 4) wf_trace_kernel(WfArgs w, uint32_t first, uint32_t slot_begin,
                                                                                   uint32_t slot_end) {
    const RenderArgs& a = w.r;
    ScenePtr sc = (ScenePtr)a.scene;
    __shared__ __attribute__((aligned(16))) uint32_t s_jump[1][kTraceJumpWords]; // the draws of one shaded point
    // one row per primitive of the scene (dynamic, sized by the launch): with the two jump tables the workgroup
    // stays under 32 KB for scenes of up to 49 primitives, so that five workgroups fit a CU's LDS
    __shared__ uint32_t s_coop[COOP ? kTraceWaves : 1u][COOP ? kCoopOwners * OW_WORDS : 1u];
    __shared__ uint32_t s_coop_acc[COOP ? kTraceWaves : 1u][COOP ? kCoopOwners * 5u : 1u];
    __shared__ uint32_t s_pend[COOP ? kTraceWaves : 1u][COOP ? kCoopList : 1u];
    __shared__ uint8_t s_pgeom[COOP ? kTraceWaves : 1u][COOP ? kCoopList : 4u];
    __shared__ uint32_t s_hitbits[COOP ? kTraceWaves : 1u][COOP ? kCoopList / 32u : 1u];
    extern __shared__ __attribute__((aligned(16))) float s_prim_words[];
    PrimRow* s_prim = reinterpret_cast<PrimRow*>(s_prim_words);
    for (uint32_t k = threadIdx.x; k < kTraceJumpWords; k += 64u * kTraceWaves) s_jump[0][k] = w.jump_tabs[k];
    fill_prim_table(sc, s_prim, threadIdx.x, 64u * kTraceWaves);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_slot = blockIdx.x * kTraceWaves + (threadIdx.x >> 6);
    const uint32_t launch_slot = slot_begin + (COOP ? wave_slot / kCoopSplit : wave_slot); // this launch walks [slot_begin, slot_end)
    if (launch_slot >= slot_end) return;
    const uint32_t unit = w.unit_order ? w.unit_order[launch_slot] : launch_slot; // most expensive units first
    const bool own = !COOP || (lane >> 3) == wave_slot % kCoopSplit; // the pixels this wave walks
    const size_t n_lp = w.n_lp;
    const size_t lp = (size_t)unit * 64u + lane;
    const uint32_t npix = a.nx * a.ny;
    const PixelMap pm = map_pixel(a, unit, lane);
    const uint32_t pix = pm.valid ? pm.j * a.nx + pm.i : 0u;
    const CameraRays cam = load_camera(sc, a);
    const float pix_w = cam.pix_w, pix_h = cam.pix_h;
    const uint32_t aa_iter = a.aa_iter > 4u ? 4u : a.aa_iter; // main.cu:233
    const float inv_ind = 1.0f / (float)a.ind_rays;    // main.cu:182
    // generator draws one diffuse hit consumes in LightList::calculateColor (AreaLight.hpp:342-343,495)
    uint32_t skip_draws = 0;
    for (uint32_t li = 0; li < sc->n_lights; ++li) {
        const DevLight CRT_CONST* L = &sc->lights[li];
        if (L->kind == CRT_LIGHT_AREA && L->initialized) skip_draws += 2u * L->shadow_samples;
    }

    // ---- traversal state
    uint32_t ctl = 0;
    if (!first && own) ctl = w.state[ST_CTL * n_lp + lp];
    bool done = !pm.valid || !own || (ctl & 1u);
    const bool live_at_entry = !done;
    bool color_active = (ctl >> 1) & 1u, cur_valid = (ctl >> 2) & 1u;
    bool task_active = (ctl >> 3) & 1u, waiting = (ctl >> 4) & 1u;
    uint32_t t_level = (ctl >> 5) & 7u, t_corner = (ctl >> 8) & 7u, aa_top = (ctl >> 12) & 15u, nframes = (ctl >> 16) & 15u;
    float t_cx = 0.0f, t_cy = 0.0f;
    Rng rng = {0, 0, 0, 0, 0, 0};
    V3 cur_o = mk3(0, 0, 0), cur_d = mk3(0, 0, 0), ray_d = mk3(0.0f, 0.0f, -1.0f);
    V4 cur_att = V4{1, 1, 1, 1};
    uint32_t cur_ref = 0, cur_depth = 0, on_stack = 0;
    if (live_at_entry) {
        rng.v0 = a.rng[0 * (size_t)npix + pix];
        rng.v1 = a.rng[1 * (size_t)npix + pix];
        rng.v2 = a.rng[2 * (size_t)npix + pix];
        rng.v3 = a.rng[3 * (size_t)npix + pix];
        rng.v4 = a.rng[4 * (size_t)npix + pix];
        rng.d = a.rng[5 * (size_t)npix + pix];
        if (!first) {
            const uint32_t* s = w.state + lp;
            cur_o = mk3(u2f(s[(ST_CUR_O + 0) * n_lp]), u2f(s[(ST_CUR_O + 1) * n_lp]), u2f(s[(ST_CUR_O + 2) * n_lp]));
            cur_d = mk3(u2f(s[(ST_CUR_D + 0) * n_lp]), u2f(s[(ST_CUR_D + 1) * n_lp]), u2f(s[(ST_CUR_D + 2) * n_lp]));
            cur_att = V4{u2f(s[(ST_CUR_ATT + 0) * n_lp]), u2f(s[(ST_CUR_ATT + 1) * n_lp]),
                         u2f(s[(ST_CUR_ATT + 2) * n_lp]), u2f(s[(ST_CUR_ATT + 3) * n_lp])};
            cur_ref = s[ST_CUR_REF * n_lp];
            cur_depth = s[ST_CUR_DEPTH * n_lp];
            ray_d = mk3(u2f(s[(ST_RAY_D + 0) * n_lp]), u2f(s[(ST_RAY_D + 1) * n_lp]), u2f(s[(ST_RAY_D + 2) * n_lp]));
            on_stack = s[ST_ON_STACK * n_lp];
            t_cx = u2f(s[ST_T_CX * n_lp]);
            t_cy = u2f(s[ST_T_CY * n_lp]);
        }
    }
    float x_world, y_world;
    cam.pixel_centre(a, pm.i, pm.j, x_world, y_world);

    // aa_color's task list (main.cu:230-304) lives in the state planes; it starts with the whole pixel
    uint32_t* aa_stack = w.state + (size_t)ST_AA * n_lp + lp; // entry e: planes 3e, 3e+1, 3e+2
    if (first && live_at_entry) {
        aa_stack[0 * n_lp] = f2u(x_world);
        aa_stack[1 * n_lp] = f2u(y_world);
        aa_stack[2 * n_lp] = aa_iter;
        aa_top = 1;
    }
    if (waiting) {
        // the sum kernel has compared the four corner colours of the level >= 2 task that ended last round
        const uint32_t verdict = w.aa_verdict[lp] & 0xffu;
        if (verdict == AA_SUBDIVIDE) { // main.cu:293-299: four tasks one level down, at the quarter points
            float sx = pix_w, sy = pix_h;
            for (uint32_t k = t_level; k < aa_iter; ++k) {
                sx = sx * 0.5f;
                sy = sy * 0.5f;
            }
            const float hx = sx * 0.5f, hy = sy * 0.5f;
            for (uint32_t c = 0; c < 4u; ++c) {
                const float ox = (c & 1u) ? 1.0f : -1.0f, oy = (c & 2u) ? 1.0f : -1.0f;
                aa_stack[(3u * aa_top + 0u) * n_lp] = f2u(t_cx + ox * hx * 0.5f);
                aa_stack[(3u * aa_top + 1u) * n_lp] = f2u(t_cy + oy * hy * 0.5f);
                aa_stack[(3u * aa_top + 2u) * n_lp] = t_level - 1u;
                ++aa_top;
            }
        }
        waiting = false;
    }
    uint32_t cnt = 0; // result slots filled this round
    uint32_t cnt_shade = 0, cnt_expand = 0;
    const uint32_t quota = w.quota;
    // A per-pixel wave's first chunk of record indices is its own by its place in the launch: one atomic per wave on
    // one counter at the start of the kernel, 8 100 of them at 10 ns each, was a third of a gl_iter 0 frame's trace
    // time.  (The cooperative kernel has eight times the waves, and most of their first chunks would stay empty.)
    WorkCursor wc = {COOP ? 0u : wave_slot * kWorkChunk, COOP ? kWorkChunk : 0u, !COOP};
#ifdef CRT_TRACE_PROFILE // build with -DCRT_TRACE_PROFILE: every 37th wave prints its steps and the cycles per section
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = 0, prof_iters = 0;
#define PROF_MARK(k)                                   \
    do {                                               \
        const unsigned long long now_ = clock64();    \
        prof[k] += now_ - prof_t;                      \
        prof_t = now_;                                 \
    } while (0)
    const unsigned long long wall0 = wall_clock64(), cyc0 = clock64();
    prof_t = clock64();
#else
#define PROF_MARK(k)
#endif

    for (;;) {
        const bool go = !done && !waiting && cnt < quota;
        const unsigned long long go_mask = __ballot(go);
        if (go_mask == 0ull) break;
#if CRT_EARLY_ROUND_END
        {
            // a round may end anywhere in a pixel's walk: the few pixels that have not filled their slots go on next round
            const unsigned long long full_mask = __ballot(!done && !waiting && cnt >= quota);
            if (full_mask != 0ull && (uint32_t)__popcll(go_mask) * (uint32_t)(CRT_EARLY_ROUND_END) < (uint32_t)__popcll(full_mask)) break;
        }
#endif
#ifdef CRT_TRACE_PROFILE
        ++prof_iters;
#endif
        bool emit = false;      // a diffuse hit to record
        bool want_coop = false; // COOP: the innermost frame's children are leaves
        V3 n_P = mk3(0, 0, 0), n_N = mk3(0, 0, 0), n_view = mk3(0, 0, 0);
        V4 n_att = V4{0, 0, 0, 0};
        uint32_t n_geom = 0, n_depth = 0;
        if (go) {
            PROF_LANES(0);
            if (!cur_valid) {
                if (!color_active) {
                    PROF_LANES(1);
                    if (!task_active && aa_top != 0u) { // next task of aa_color's list
                        --aa_top;
                        t_cx = u2f(aa_stack[(3u * aa_top + 0u) * n_lp]);
                        t_cy = u2f(aa_stack[(3u * aa_top + 1u) * n_lp]);
                        t_level = aa_stack[(3u * aa_top + 2u) * n_lp];
                        t_corner = 0;
                        task_active = true;
                    }
                    if (!task_active) {
                        // render() epilogue (main.cu:339-346) happens in the sum kernel
                        w.tags[slot_at(w, cnt, lp)] = (uint8_t)TAG_END_PIXEL;
                        ++cnt;
                        done = true;
                    } else {
                        // box of this task: pixel size halved (aa_iter - level) times (main.cu:251,297)
                        float sx = pix_w, sy = pix_h;
                        for (uint32_t k = t_level; k < aa_iter; ++k) {
                            sx = sx * 0.5f;
                            sy = sy * 0.5f;
                        }
                        const float hx = sx * 0.5f, hy = sy * 0.5f;
                        const float ox = (t_corner & 1u) ? 1.0f : -1.0f, oy = (t_corner & 2u) ? 1.0f : -1.0f;
                        float rx, ry;
                        if (t_level == 0u) { // one sample at the centre
                            rx = t_cx;
                            ry = t_cy;
                        } else if (t_level == 1u) { // four samples at the quarter points
                            rx = t_cx + ox * hx * 0.5f;
                            ry = t_cy + oy * hy * 0.5f;
                        } else { // four samples at the box corners, compared afterwards
                            rx = t_cx + ox * hx;
                            ry = t_cy + oy * hy;
                        }
                        {
                            // the camera's constants are needed here only (a fifth of the iterations, four lanes): read
                            // them where they are used instead of holding twenty scalar registers through the walk (the
                            // empty asm keeps the compiler from hoisting the loads back out of the loop)
                            ScenePtr sc_here = sc;
                            asm volatile("" : "+s"(sc_here));
                            const CameraRays cam_here = load_camera(sc_here, a);
                            cam_here.ray(rx, ry, cur_o, cur_d);
                        }
                        cur_att = V4{1.0f, 1.0f, 1.0f, 1.0f};
                        cur_ref = a.ref_iter;
                        cur_depth = a.gl_iter;
                        cur_valid = true;
                        color_active = true;
                        ray_d = mk3(0.0f, 0.0f, -1.0f);
                        nframes = 0;
                        on_stack = 0;
                    }
                } else if (nframes == 0) {
                    // color() returned (main.cu:225): the sum kernel folds it into the pixel
                    w.tags[slot_at(w, cnt, lp)] = (uint8_t)(TAG_END_CALL | (t_level << 2) | (t_corner << 5));
                    ++cnt;
                    color_active = false;
                    if (t_level == 0u || ++t_corner == 4u) {
                        task_active = false;
                        waiting = t_level >= 2u; // whether to subdivide depends on the four colours (main.cu:283-300)
                    }
                } else {
                    uint4* f = frame_at(w, nframes - 1u, lp);
                    const uint4 fn = f[FG_NORMAL * n_lp];
                    const uint32_t cdepth = fn.w;
                    if (COOP && cdepth == 0u) {
                        want_coop = true; // the wave expands these children together, below
                    } else {
                        // pop the next child of the innermost frame (children leave last-to-first); its two
                        // draws are the last ones before the frame's generator state
                        PROF_LANES(2);
                        const uint4 fp = f[FG_POINT * n_lp], fa = f[FG_ATT * n_lp], fg = f[FG_GEN * n_lp], fg2 = f[FG_GEN2 * n_lp];
                        Rng rv = Rng{fg.x, fg.y, fg.z, fg.w, fg2.x, fg2.y};
                        float r2 = rng_current_uniform(rv);
                        rng_unstep(rv);
                        float r1 = rng_current_uniform(rv);
                        rng_unstep(rv);
                        V3 hp = mk3(u2f(fp.x), u2f(fp.y), u2f(fp.z));
                        V3 hn = mk3(u2f(fn.x), u2f(fn.y), u2f(fn.z));
                        V4 patt = V4{u2f(fa.x), u2f(fa.y), u2f(fa.z), u2f(fa.w)};
                        const float* fpr = s_prim[fp.w];
                        V4 pdiff = V4{fpr[12], fpr[13], fpr[14], fpr[15]};
                        uint32_t rem = fg2.z - 1u;
                        if (rem == 0u) {
                            --nframes;
                        } else {
                            f[FG_GEN * n_lp] = make_uint4(rv.v0, rv.v1, rv.v2, rv.v3);
                            f[FG_GEN2 * n_lp] = make_uint4(rv.v4, rv.d, rem, 0u);
                        }
                        --on_stack;
                        child_ray(hp, hn, patt, pdiff, r1, r2, inv_ind, cur_o, cur_d, cur_att);
                        cur_ref = a.ref_iter;
                        cur_depth = cdepth;
                        cur_valid = true;
                    }
                }
            }
            PROF_MARK(0);
            if (cur_valid) {
                // ---- pop (main.cu:132-147)
                cur_valid = false;
                if (cur_ref != 0u) {
                    PROF_LANES(3);
                    const V3 view = neg(cur_d); // viewDir = -state.r_dir, the stored (un-normalised) direction
                    const V4 att = cur_att;
                    V3 P = mk3(0, 0, 0), N = mk3(0, 0, 0);
                    uint32_t geom = 0;
#ifdef CRT_SHADE_TRACE
                    const V3 o_ = cur_o;
                    const uint32_t ref_ = cur_ref;
#endif
                    const uint32_t what = pop_ray(sc, s_prim, cur_o, cur_d, cur_att, cur_ref, ray_d, P, N, geom);
#ifdef CRT_SHADE_TRACE // the rays of that pixel, in the form of the CPU checker's ORC_TRACE lines
                    if (pm.valid && pm.j * a.nx + pm.i == (uint32_t)(CRT_SHADE_TRACE))
                        printf("gpu: ray o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) depth %u ref %u: %s at (%.9g %.9g %.9g) primitive %u N=(%.9g %.9g %.9g)\n",
                               o_.x, o_.y, o_.z, ray_d.x, ray_d.y, ray_d.z, cur_depth, ref_,
                               what == POP_MISS ? "miss" : (what == POP_DIFFUSE ? "diffuse hit" : "bounce"), P.x, P.y, P.z, geom, N.x, N.y, N.z);
#endif
                    if (what == POP_MISS) {
                        // final_color += attenuation * Color::black() (main.cu:149-151)
                        const V4 c = att * V4{0.0f, 0.0f, 0.0f, 1.0f};
                        store_streaming(&w.contrib[slot_at(w, cnt, lp)], make_float4(c.x, c.y, c.z, c.w));
                        ++cnt;
                    } else if (what == POP_DIFFUSE) {
                        emit = true;
                        n_P = P;
                        n_N = N;
                        n_view = view;
                        n_att = att;
                        n_geom = geom;
                        n_depth = cur_depth;
                    } else {
                        cur_valid = true;
                    }
                }
            }
        }
        PROF_MARK(1);
        // ---- record the diffuse hits of this step (all lanes are here together)
        const uint64_t em = __ballot(emit);
        if (em != 0) {
            float4 dark_c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const bool dark = emit && ambient_only(sc, s_prim, n_P, n_geom, n_att, dark_c);
            const uint64_t rm = __ballot(emit && !dark);
            uint32_t my_wi = 0;
            if (rm != 0) my_wi = reserve_records(w, wc, rm, lane);
            if (emit) {
                PROF_LANES(10);
                if (dark) store_streaming(&w.contrib[slot_at(w, cnt, lp)], dark_c);
#ifdef CRT_ABLATE_STORE
                else if (n_P.x == 1234.5f) store_record(w, my_wi, n_P, n_N, n_view, n_att, n_geom, rng, (uint32_t)slot_at(w, cnt, lp));
#else
                else store_record(w, my_wi, n_P, n_N, n_view, n_att, n_geom, rng, (uint32_t)slot_at(w, cnt, lp));
#endif
                ++cnt;
                ++cnt_shade;
                // the light samples of this hit (drawn by the filter kernel from the recorded state)
#ifdef CRT_ABLATE_JUMP
                rng.d += skip_draws * 362437u; rng.v0 ^= rng.v1 + rng.d;
#else
                if (skip_draws != 0u && skip_draws == w.jump_n[0]) {
                    if (CRT_JUMP5) rng_jump5(rng, s_jump[0], skip_draws * 362437u);
                    else rng_jump(rng, s_jump[0], skip_draws * 362437u);
                }
                else for (uint32_t k = 0; k < skip_draws; ++k) rng_step(rng);
#endif
            }
            {
                // ---- expanding hits: the reference draws (r1,r2) for every one of the ind_rays children and pushes
                // while its stack has room (main.cu:176-191).  The frame keeps the generator state behind the draws of
                // the `accepted` children it holds (the per-pixel kernel steps them back out of it one by one), the
                // pixel goes on behind all 2 ind_rays draws.  Both are jumps of one lane's state, made by the whole wave.
                // The cooperative kernel needs the draws of a leaf node's children as numbers (its lanes pick them up
                // from w.draws): those it steps through, and jumps only over the draws of the children that found no room.
                const bool push = emit && n_depth > 0u;
                uint32_t accepted = 0, to_jump = 0; // children kept; of those, the ones whose draws are still to be jumped over
                if (push) {
                    PROF_LANES(13);
                    ++cnt_expand;
                    const uint32_t room = on_stack < CRT_RAY_STACK ? CRT_RAY_STACK - on_stack : 0u;
                    accepted = a.ind_rays < room ? a.ind_rays : room;
                    to_jump = accepted;
                    if (COOP && n_depth == 1u) {
                        for (uint32_t k = 0; k < 2u * accepted; ++k) w.draws[(size_t)k * n_lp + lp] = rng_uniform(rng);
                        to_jump = 0u;
                    }
                }
                Rng fgen = rng; // generator state the frame keeps
                for (uint64_t pmask = __ballot(push); pmask != 0ull; pmask &= pmask - 1ull) {
                    const int src = __builtin_ctzll(pmask);
                    uint32_t v[5] = {(uint32_t)__builtin_amdgcn_readlane((int)rng.v0, src), (uint32_t)__builtin_amdgcn_readlane((int)rng.v1, src),
                                     (uint32_t)__builtin_amdgcn_readlane((int)rng.v2, src), (uint32_t)__builtin_amdgcn_readlane((int)rng.v3, src),
                                     (uint32_t)__builtin_amdgcn_readlane((int)rng.v4, src)};
                    const uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)to_jump, src);
                    const uint32_t rest = a.ind_rays - (uint32_t)__builtin_amdgcn_readlane((int)accepted, src);
                    for (uint32_t left = acc; left != 0u;) { // (more children than tables: ind_rays above CRT_RAY_STACK never are accepted)
                        const uint32_t n = left < w.draw_tabs_n ? left : w.draw_tabs_n;
                        wave_jump(v, w.draw_tabs + (size_t)(n - 1u) * kJumpWords, lane);
                        left -= n;
                    }
                    if (lane == (uint32_t)src) fgen = Rng{v[0], v[1], v[2], v[3], v[4], rng.d + 2u * acc * 362437u};
                    for (uint32_t left = rest; left != 0u;) {
                        const uint32_t n = left < w.draw_tabs_n ? left : w.draw_tabs_n;
                        wave_jump(v, w.draw_tabs + (size_t)(n - 1u) * kJumpWords, lane);
                        left -= n;
                    }
                    if (lane == (uint32_t)src) rng = Rng{v[0], v[1], v[2], v[3], v[4], rng.d + 2u * (acc + rest) * 362437u};
                }
                if (push && accepted > 0u) {
                    uint4* f = frame_at(w, nframes, lp);
                    ++nframes;
                    f[FG_POINT * n_lp] = make_uint4(f2u(n_P.x), f2u(n_P.y), f2u(n_P.z), n_geom);
                    f[FG_NORMAL * n_lp] = make_uint4(f2u(n_N.x), f2u(n_N.y), f2u(n_N.z), n_depth - 1u);
                    f[FG_ATT * n_lp] = make_uint4(f2u(n_att.x), f2u(n_att.y), f2u(n_att.z), f2u(n_att.w));
                    f[FG_GEN * n_lp] = make_uint4(fgen.v0, fgen.v1, fgen.v2, fgen.v3);
                    f[FG_GEN2 * n_lp] = make_uint4(fgen.v4, fgen.d, accepted, 0u);
                    on_stack += accepted;
                }
            }
        }
        PROF_MARK(2);
        if (COOP) {
            // ---- leaf children: one lane per child ray.  The owners (pixels) that want their innermost frame expanded
            // put up to kCoopBatch children each into one list, in lane order, and the wave's lanes work through the
            // list (below).  Every child takes one result slot, so n <= quota - cnt children of an owner respect its
            // quota.
            const uint64_t cm = __ballot(want_coop);
            if (cm != 0) {
                uint32_t* own_tab = s_coop[threadIdx.x >> 6]; // [kCoopOwners][OW_WORDS], this wave's
                uint32_t my_rem = 0, my_n = 0;
                if (want_coop) {
                    my_rem = frame_at(w, nframes - 1u, lp)[FG_GEN2 * n_lp].z;
                    my_n = my_rem < quota - cnt ? my_rem : quota - cnt;
                    if (my_n > kCoopBatch) my_n = kCoopBatch;
                }
                uint32_t my_pre = 0, total = 0, n_own = 0; // list position of an owner's first child; list length; owners
                for (uint64_t m = cm; m != 0; m &= m - 1u) {
                    const int o = __builtin_ctzll(m);
                    if (lane == (uint32_t)o) my_pre = total;
                    total += (uint32_t)__builtin_amdgcn_readlane((int)my_n, o);
                    ++n_own;
                }
                if (want_coop) {
                    uint32_t* oj = own_tab + lane_rank(cm) * OW_WORDS;
                    oj[OW_PRE] = my_pre;
                    oj[OW_N] = my_n;
                    oj[OW_LP] = (uint32_t)lp;
                    oj[OW_NFRAMES] = nframes;
                    oj[OW_CNT] = cnt;
                    oj[OW_REM] = my_rem;
                    oj[OW_RNG + 0] = rng.v0;
                    oj[OW_RNG + 1] = rng.v1;
                    oj[OW_RNG + 2] = rng.v2;
                    oj[OW_RNG + 3] = rng.v3;
                    oj[OW_RNG + 4] = rng.v4;
                    oj[OW_RNG + 5] = rng.d;
                    oj[OW_RAY_D + 0] = f2u(ray_d.x);
                    oj[OW_RAY_D + 1] = f2u(ray_d.y);
                    oj[OW_RAY_D + 2] = f2u(ray_d.z);
                    oj[OW_LANE] = lane;
                    oj[OW_RECS] = 0u; // diffuse hits among its children of this step (filled in at the end)
                }
                // what the list's children leave behind for the end of the step: the record of a child that ended in a
                // diffuse hit (its index and primitive; the record's ordinal is known only when every earlier child
                // of its owner has ended), and which children ended in a diffuse hit at all
                uint32_t* pend = s_pend[threadIdx.x >> 6];
                uint8_t* pgeom = s_pgeom[threadIdx.x >> 6];
                uint32_t* hitbits = s_hitbits[threadIdx.x >> 6];
                for (uint32_t q = lane; q < total; q += 64u) pend[q] = kNoRecord;
                if (lane < kCoopList / 32u) hitbits[lane] = 0u;
                // ---- the wave's lanes take the list's children in order, one each; a lane keeps its child until the
                // child's chain of mirror / glass bounces has ended (main.cu:132-221 with gl_depth 0) and then takes the
                // next one, so every pass over the scene runs on 64 rays while children are left -- working through the
                // list 64 at a time, each batch until its longest chain has ended, took 2.2 passes per batch.
                uint32_t next = 0;     // list position of the next child to hand out (wave-uniform)
                bool running = false;  // this lane holds a child whose chain has not ended
                uint32_t c_pos = 0, c_j = 0, c_ref = 0, c_lp = 0;
                V3 c_o = mk3(0, 0, 0), c_d = mk3(0, 0, 0), c_ray_d = mk3(0, 0, -1.0f);
                V4 c_att = V4{0, 0, 0, 0};
                for (;;) {
                    const uint64_t fm = __ballot(!running);
                    const uint32_t left = total - next;
                    if (left == 0u && fm == ~0ull) break;
                    const uint32_t fr = lane_rank(fm);
                    if (!running && fr < left) {
                        const uint32_t pp = next + fr;
                        uint32_t j = 0; // the owner whose children cover list position pp
                        for (uint32_t jj = 1; jj < n_own; ++jj)
                            if (pp >= own_tab[jj * OW_WORDS + OW_PRE]) j = jj;
                        const uint32_t* oj = own_tab + j * OW_WORDS;
                        const uint32_t o_lp = oj[OW_LP], o_rem = oj[OW_REM];
                        c_ray_d = mk3(u2f(oj[OW_RAY_D + 0]), u2f(oj[OW_RAY_D + 1]), u2f(oj[OW_RAY_D + 2]));
                        const uint4* f = frame_at(w, oj[OW_NFRAMES] - 1u, o_lp);
                        const uint4 fp = f[FG_POINT * n_lp], fn = f[FG_NORMAL * n_lp], fa = f[FG_ATT * n_lp];
                        const V3 hp = mk3(u2f(fp.x), u2f(fp.y), u2f(fp.z));
                        const V3 hn = mk3(u2f(fn.x), u2f(fn.y), u2f(fn.z));
                        const V4 patt = V4{u2f(fa.x), u2f(fa.y), u2f(fa.z), u2f(fa.w)};
                        const float* fpr = s_prim[fp.w];
                        const V4 pdiff = V4{fpr[12], fpr[13], fpr[14], fpr[15]};
                        // the c-th child of its owner to leave the stack: child rem-1-c, drawn as pair rem-1-c
                        const uint32_t child = o_rem - 1u - (pp - oj[OW_PRE]);
                        const float r1 = w.draws[(size_t)(2u * child) * n_lp + o_lp];
                        const float r2 = w.draws[(size_t)(2u * child + 1u) * n_lp + o_lp];
                        child_ray(hp, hn, patt, pdiff, r1, r2, inv_ind, c_o, c_d, c_att);
                        c_ref = a.ref_iter;
                        c_pos = pp;
                        c_j = j;
                        c_lp = o_lp;
                        running = true;
                    }
                    next += (uint32_t)__popcll(fm) < left ? (uint32_t)__popcll(fm) : left;
                    PROF_MARK(4);
                    // ---- one pop for every lane that holds a ray
                    uint32_t outcome = 3u; // 0 ended without a result (bounces used up), 1 miss, 2 diffuse hit, 3 still running / idle
                    V3 c_P = mk3(0, 0, 0), c_N = mk3(0, 0, 0), c_view = mk3(0, 0, 0);
                    V4 r_att = V4{0, 0, 0, 0};
                    uint32_t c_geom = 0;
                    if (running) {
                        if (c_ref == 0u) {
                            outcome = 0u;
                        } else {
                            c_view = neg(c_d);
                            r_att = c_att;
                            const uint32_t what = pop_ray(sc, s_prim, c_o, c_d, c_att, c_ref, c_ray_d, c_P, c_N, c_geom);
                            if (what == POP_MISS) outcome = 1u;
                            else if (what == POP_DIFFUSE) outcome = 2u;
                        }
                        if (outcome != 3u) running = false;
                    }
                    PROF_MARK(5);
                    // ---- the children that ended in this pass.  A child's result slot is its place in the list (the
                    // sum kernel adds a pixel's slots in order, which is the order the children leave the stack in; a
                    // child that ended without a result marks its slot as one to skip).
                    const uint32_t* oj = own_tab + c_j * OW_WORDS;
                    const uint32_t o_pre = oj[OW_PRE], o_n = oj[OW_N];
                    const size_t slot = slot_at(w, oj[OW_CNT] + (c_pos - o_pre), c_lp);
                    float4 dark_c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    const bool dark = outcome == 2u && ambient_only(sc, s_prim, c_P, c_geom, r_att, dark_c);
                    const uint64_t m_store = __ballot(outcome == 2u && !dark); // those that become records
                    if (outcome == 0u) w.tags[slot] = (uint8_t)TAG_SKIP;
                    if (outcome == 1u) {
                        const V4 c = r_att * V4{0.0f, 0.0f, 0.0f, 1.0f};
                        store_streaming(&w.contrib[slot], make_float4(c.x, c.y, c.z, c.w));
                    }
                    if (dark) store_streaming(&w.contrib[slot], dark_c);
                    if (outcome == 2u) atomicOr(&hitbits[c_pos >> 5], 1u << (c_pos & 31u)); // dark or not: a shaded point
                    if (m_store != 0) {
                        // the record keeps the owner's generator state; its ordinal (how many shaded points further on
                        // its own draws start) follows at the end of the step
                        const uint32_t my_wi = reserve_records(w, wc, m_store, lane);
                        if (outcome == 2u && !dark) {
                            const Rng o_rng = Rng{oj[OW_RNG + 0], oj[OW_RNG + 1], oj[OW_RNG + 2], oj[OW_RNG + 3], oj[OW_RNG + 4], oj[OW_RNG + 5]};
                            store_record_body(w, my_wi, c_P, c_N, c_view, r_att, o_rng, (uint32_t)slot);
                            pend[c_pos] = my_wi < w.cap_work ? (my_wi | (c_j << kRecBits)) : kNoRecord; // record and owner (< 8)
                            pgeom[c_pos] = (uint8_t)c_geom;
                        }
                    }
                    // the owner's last child leaves the direction register (main.cu:138: the last ray that had a usable direction)
                    if (outcome != 3u && c_pos + 1u == o_pre + o_n) {
                        uint32_t* ow = own_tab + c_j * OW_WORDS;
                        ow[OW_RAY_D + 0] = f2u(c_ray_d.x);
                        ow[OW_RAY_D + 1] = f2u(c_ray_d.y);
                        ow[OW_RAY_D + 2] = f2u(c_ray_d.z);
                    }
                    PROF_MARK(6);
                }
                // ---- every child has ended: the records' ordinals (diffuse hits among the earlier children of the same
                // owner) and, per owner, its number of diffuse hits
                for (uint32_t q = lane; q < total; q += 64u) {
                    const uint32_t wj = pend[q];
                    if (wj == kNoRecord) continue;
                    const uint32_t ordinal = count_bits(hitbits, own_tab[(wj >> kRecBits) * OW_WORDS + OW_PRE], q);
                    __builtin_nontemporal_store((uint32_t)pgeom[q] | (ordinal << 8), &rec_point(w)[wj & kRecMask].w);
                }
                if (lane < n_own) {
                    uint32_t* ow = own_tab + lane * OW_WORDS;
                    ow[OW_RECS] = count_bits(hitbits, ow[OW_PRE], ow[OW_PRE] + ow[OW_N]);
                }
                PROF_MARK(7);
                // ---- every owner continues as many shaded points further on as its children had diffuse hits: the
                // jumps of all owners at once, one table look-up per (owner, nibble) spread over the lanes, their
                // images added up in LDS
                uint32_t* acc = s_coop_acc[threadIdx.x >> 6];
                if (lane < kCoopOwners * 5u) acc[lane] = 0u;
                if (w.jump_n[0] != 0u) {
                    for (uint32_t q = lane; q < n_own * 40u; q += 64u) {
                        const uint32_t o = q / 40u, e = q - o * 40u; // owner, nibble of its state
                        const uint32_t* ou = own_tab + o * OW_WORDS;
                        const uint32_t k = ou[OW_RECS];
                        if (k != 0u) {
                            const uint32_t* tab = w.coop_tabs + (size_t)(k - 1u) * kJumpWords;
                            const uint32_t idx = e * 16u + ((ou[OW_RNG + (e >> 3)] >> (4u * (e & 7u))) & 15u);
                            const uint4 t4 = reinterpret_cast<const uint4*>(tab)[idx];
                            atomicXor(&acc[o * 5u + 0u], t4.x);
                            atomicXor(&acc[o * 5u + 1u], t4.y);
                            atomicXor(&acc[o * 5u + 2u], t4.z);
                            atomicXor(&acc[o * 5u + 3u], t4.w);
                            atomicXor(&acc[o * 5u + 4u], tab[4u * kJumpEntries + idx]);
                        }
                    }
                }
                if (want_coop) {
                    const uint32_t o = lane_rank(cm);
                    const uint32_t* ow = own_tab + o * OW_WORDS;
                    const uint32_t recs = ow[OW_RECS];
                    cnt += my_n; // every child has a slot
                    cnt_shade += recs;
                    if (recs != 0u && w.jump_n[0] != 0u) {
                        rng.v0 = acc[o * 5u + 0u];
                        rng.v1 = acc[o * 5u + 1u];
                        rng.v2 = acc[o * 5u + 2u];
                        rng.v3 = acc[o * 5u + 3u];
                        rng.v4 = acc[o * 5u + 4u];
                        rng.d += recs * w.jump_n[0] * 362437u;
                    }
                    ray_d = mk3(u2f(ow[OW_RAY_D + 0]), u2f(ow[OW_RAY_D + 1]), u2f(ow[OW_RAY_D + 2]));
                    on_stack -= my_n;
                    if (my_rem == my_n) --nframes;
                    else frame_at(w, nframes - 1u, lp)[FG_GEN2 * n_lp].z = my_rem - my_n;
                }
            }
        }
        PROF_MARK(3);
    }

#ifdef CRT_TRACE_PROFILE
    if (lane == 0 && unit % 37u == 0u)
        printf("wave %u: %llu steps, %llu cycles, %llu ticks of 10 ns; next ray %llu, pop %llu, record %llu, cooperative %llu "
               "(children %llu, chains %llu, jumps %llu, results %llu)\n", unit, prof_iters, clock64() - cyc0, wall_clock64() - wall0,
               prof[0], prof[1], prof[2], prof[3] + prof[4] + prof[5] + prof[6] + prof[7], prof[4], prof[5], prof[6], prof[7]);
#endif
    // ---- the unused tail of the last chunk holds no record
    if (wc.have)
        for (uint32_t k = wc.used + lane; k < kWorkChunk; k += 64u)
            if (wc.base + k < w.cap_work) rec_point(w)[wc.base + k].w = kNoRecord;

    // ---- park the traversal
    if (own) w.slot_count[lp] = cnt;
    if (live_at_entry) {
        a.rng[0 * (size_t)npix + pix] = rng.v0;
        a.rng[1 * (size_t)npix + pix] = rng.v1;
        a.rng[2 * (size_t)npix + pix] = rng.v2;
        a.rng[3 * (size_t)npix + pix] = rng.v3;
        a.rng[4 * (size_t)npix + pix] = rng.v4;
        a.rng[5 * (size_t)npix + pix] = rng.d;
        a.counters[pix] += cnt_shade;
        a.counters[npix + pix] += cnt_expand;
        uint32_t* s = w.state + lp;
        s[ST_CTL * n_lp] = (done ? 1u : 0u) | (color_active ? 2u : 0u) | (cur_valid ? 4u : 0u) | (task_active ? 8u : 0u) |
                           (waiting ? 16u : 0u) | (t_level << 5) | (t_corner << 8) | (aa_top << 12) | (nframes << 16);
        if (!done) {
            s[(ST_CUR_O + 0) * n_lp] = f2u(cur_o.x);
            s[(ST_CUR_O + 1) * n_lp] = f2u(cur_o.y);
            s[(ST_CUR_O + 2) * n_lp] = f2u(cur_o.z);
            s[(ST_CUR_D + 0) * n_lp] = f2u(cur_d.x);
            s[(ST_CUR_D + 1) * n_lp] = f2u(cur_d.y);
            s[(ST_CUR_D + 2) * n_lp] = f2u(cur_d.z);
            s[(ST_CUR_ATT + 0) * n_lp] = f2u(cur_att.x);
            s[(ST_CUR_ATT + 1) * n_lp] = f2u(cur_att.y);
            s[(ST_CUR_ATT + 2) * n_lp] = f2u(cur_att.z);
            s[(ST_CUR_ATT + 3) * n_lp] = f2u(cur_att.w);
            s[ST_CUR_REF * n_lp] = cur_ref;
            s[ST_CUR_DEPTH * n_lp] = cur_depth;
            s[(ST_RAY_D + 0) * n_lp] = f2u(ray_d.x);
            s[(ST_RAY_D + 1) * n_lp] = f2u(ray_d.y);
            s[(ST_RAY_D + 2) * n_lp] = f2u(ray_d.z);
            s[ST_ON_STACK * n_lp] = on_stack;
            s[ST_T_CX * n_lp] = f2u(t_cx);
            s[ST_T_CY * n_lp] = f2u(t_cy);
        }
    } else if (first && own) {
        w.state[ST_CTL * n_lp + lp] = 1u; // pixel slots outside the image never run
    }
    const uint64_t alive = __ballot(!done);
    if (lane == 0 && alive != 0) atomicAdd(&w.ctrs[WF_CTR_ACTIVE], (uint32_t)__popcll(alive));
}

// --------------------------------------------------------------------------------------- filter
#ifndef CRT_WF_FILTER_WAVES
#define CRT_WF_FILTER_WAVES 4
#endif
constexpr uint32_t kRecQueueCap = 128; // < 64 left over + 64 new entries per group

// Runs the sample loop of light `li` for up to 64 queued records (one per lane).
template <bool ORDINALS>
DI void filter_batch(const WfArgs& w, ScenePtr sc, uint32_t li, const LightF& lf, uint32_t S, uint32_t (*rq)[kRecQueueCap],
                     uint32_t& rqn, uint32_t (*stg)[kStageCap], Stage& st, uint32_t lane, uint32_t& tally) {
    const uint32_t n = rqn < 64u ? rqn : 64u;
    const uint32_t base = rqn - n;
    __syncthreads();
    const bool act = lane < n;
    const uint32_t slot = base + (act ? lane : 0u); // idle lanes shadow the first entry and discard the result
    const uint32_t wi = rq[0][slot];
    const V3 P = mk3(u2f(rq[1][slot]), u2f(rq[2][slot]), u2f(rq[3][slot]));
    Rng rng;
    rng.v0 = rq[4][slot];
    rng.v1 = rq[5][slot];
    rng.v2 = rq[6][slot];
    rng.v3 = rq[7][slot];
    rng.v4 = rq[8][slot];
    rng.d = rq[9][slot];
    const uint32_t ordinal = ORDINALS ? rq[10][slot] : 0u;
    rqn = base;
    __syncthreads();
    if (ORDINALS && __any(ordinal != 0u)) ordinal_jump(rng, ordinal, w); // a record of a cooperative batch (powers of one map: order is free)
    const RecordFilter rf = make_record_filter(sc, li, lf, P);
    uint32_t shadowed = 0;
    const uint32_t key = wi | (li << kRecBits);
    uint32_t s = 0;
    if (rf.poly && rf.rest_mask == 0) { // every candidate has a slot: the sample loop is polynomial evaluation only
        for (; s + 5u <= S; s += 5u) filter_block<5, true>(w, sc, lf, rf, act, rng, key, shadowed, stg, st, lane, tally);
        for (; s < S; ++s) filter_block<1, true>(w, sc, lf, rf, act, rng, key, shadowed, stg, st, lane, tally);
    } else {
        for (; s + 5u <= S; s += 5u) filter_block<5, false>(w, sc, lf, rf, act, rng, key, shadowed, stg, st, lane, tally);
        for (; s < S; ++s) filter_block<1, false>(w, sc, lf, rf, act, rng, key, shadowed, stg, st, lane, tally);
    }
    if (act) w.shadow_cnt[(size_t)li * w.cap_work + wi] = shadowed;
}

// For every light: each group of 64 records is first classified as a whole (record_light_class, all
// lanes busy); records whose samples have to be looked at are compacted through an LDS queue, and the
// sample loop runs on full batches of 64 of them.
// ORDINALS: the records may come from the cooperative trace kernel (see kCoopBatch).
// w_end == kFromCounter: all records of the round (the count is read from the device counter, so the host
// does not have to know it).
template <bool ORDINALS>
__global__ void __launch_bounds__(64, CRT_WF_FILTER_WAVES) wf_filter_kernel(WfArgs w, uint32_t w_begin, uint32_t w_end) {
    ScenePtr sc = (ScenePtr)w.r.scene;
    const uint32_t lane = threadIdx.x;
    __shared__ uint32_t stg[3][kStageCap];
    __shared__ uint32_t rq[11][kRecQueueCap]; // record index, P, generator state (draws_before steps after the record's), the record's ordinal
    Stage st = {0u, 0u, 0u, 0u, 0u};
    uint32_t rqn = 0; // wave-uniform fill of rq
    uint32_t tally = 0;
    const size_t cw = w.cap_work;
    if (w_end == kFromCounter) w_end = records_handed_out(w);
    const uint32_t n_lights = sc->n_lights;
    const uint32_t n_groups = w_begin < w_end ? (w_end - w_begin + 63u) / 64u : 0u;
    uint32_t draws_before = 0; // draws of one record that precede light li (AreaLight.hpp:342-343 per sample)
    for (uint32_t li = 0; li < n_lights; ++li) {
        const DevLight CRT_CONST* L = &sc->lights[li];
        if (L->kind != CRT_LIGHT_AREA || !L->initialized) continue; // no draws (Light.hpp:92-111, AreaLight.hpp:586)
        const uint32_t S = L->shadow_samples;
        const LightF lf = load_lightf(L);
        GroupFeed feed = feed_begin(w, li, n_groups);
        for (uint32_t g = 0; feed_next(feed, lane, g);) {
            const uint32_t base = g * 64u;
            const uint32_t wi = w_begin + base + lane;
            const uint32_t wr = wi < w_end ? wi : w_begin;
            const uint4 pt = rec_point(w)[wr];
            const uint32_t gword = pt.w;
            const bool act = wi < w_end && gword != kNoRecord;
            const V3 P = mk3(u2f(pt.x), u2f(pt.y), u2f(pt.z));
            const uint32_t c = record_light_class(sc, li, lf, P);
            if (act && c != 0u) w.shadow_cnt[(size_t)li * cw + wi] = (c == 2u) ? S : 0u;
            const bool queue = act && c == 0u;
            const uint64_t qm = __ballot(queue);
            if (qm != 0) {
                if (queue) {
                    Rng rng = rec_load_gen(w, wr);
                    for (uint32_t k = 0; k < draws_before; ++k) rng_step(rng);
                    const uint32_t slot = rqn + lane_rank(qm);
                    rq[0][slot] = wi;
                    rq[1][slot] = f2u(P.x);
                    rq[2][slot] = f2u(P.y);
                    rq[3][slot] = f2u(P.z);
                    rq[4][slot] = rng.v0;
                    rq[5][slot] = rng.v1;
                    rq[6][slot] = rng.v2;
                    rq[7][slot] = rng.v3;
                    rq[8][slot] = rng.v4;
                    rq[9][slot] = rng.d;
                    if (ORDINALS) rq[10][slot] = (gword >> 8) & 63u;
                }
                rqn += (uint32_t)__popcll(qm);
                if (rqn >= 64u) filter_batch<ORDINALS>(w, sc, li, lf, S, rq, rqn, stg, st, lane, tally);
            }
        }
        while (rqn != 0u) filter_batch<ORDINALS>(w, sc, li, lf, S, rq, rqn, stg, st, lane, tally);
        draws_before += 2u * S;
    }
    stage_close(w, stg, st, lane);
    if (lane == 0 && tally != 0u) atomicAdd(&w.r.wave_tallies[0], (unsigned long long)tally);
}

// ---------------------------------------------------------------------------------------- audit
// Test instrument: repeats the filter's classification for every shadow ray of the records in
// [w_begin, w_end) and compares each decided ray with the exact test.  tallies[2] += decided rays,
// tallies[3] += contradictions (must stay 0).
__global__ void __launch_bounds__(64) wf_audit_kernel(WfArgs w, uint32_t w_begin, uint32_t w_end) {
    ScenePtr sc = (ScenePtr)w.r.scene;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_lights = sc->n_lights;
    unsigned long long decided = 0, bad = 0;
    if (w_end == kFromCounter) w_end = records_handed_out(w);
    for (uint32_t base = blockIdx.x * 64u; w_begin < w_end && base < w_end - w_begin; base += gridDim.x * 64u) {
        const uint32_t wi = w_begin + base + lane;
        const uint32_t wr = wi < w_end ? wi : w_begin;
        const uint4 pt = rec_point(w)[wr];
        const uint32_t gword = pt.w;
        const bool act = wi < w_end && gword != kNoRecord;
        const V3 P = mk3(u2f(pt.x), u2f(pt.y), u2f(pt.z));
        Rng rng = rec_load_gen(w, wr);
        if (act) ordinal_jump(rng, (gword >> 8) & 63u, w);
        for (uint32_t li = 0; li < n_lights; ++li) {
            const DevLight CRT_CONST* L = &sc->lights[li];
            if (L->kind != CRT_LIGHT_AREA || !L->initialized) continue;
            const uint32_t S = L->shadow_samples;
            const LightF lf = load_lightf(L);
            const LightC lc = load_lightc(L);
            const RecordFilter rf = make_record_filter(sc, li, lf, P);
            const uint32_t rec_class = record_light_class(sc, li, lf, P); // 1 all free, 2 all blocked, 0 per sample
            for (uint32_t s = 0; s < S; ++s) {
                const float u = rng_uniform(rng), v = rng_uniform(rng);
                const float us[1] = {u}, vs[1] = {v};
                bool covered, unresolved = false;
                if (rec_class != 0u) { // all samples of this record were decided at once
                    covered = rec_class == 2u;
                } else {
                    const Classes k = classify_samples<1, false>(sc, lf, rf, us, vs);
                    covered = k.sure_opaque && !k.maybe_glass;
                    unresolved = !covered && (k.maybe_any || k.sure_opaque);
                }
#ifdef CRT_SHADE_TRACE
                if (act) {
                    const uint32_t lp_ = (uint32_t)slot_pixel(w, rec_surface(w)[wr].w);
                    const PixelMap pm_ = map_pixel(w.r, lp_ / 64u, lp_ % 64u);
                    if (pm_.valid && pm_.j * w.r.nx + pm_.i == (uint32_t)(CRT_SHADE_TRACE)) {
                        V3 o_, d_;
                        float x_;
                        shadow_ray(lc, P, u, v, o_, d_, x_);
                        const Hit h_ = closest_hit(sc, o_, d_, x_);
                        bool cov_ = h_.idx >= 0;
                        if (cov_ && h_.mat_type == CRT_MAT_REFRACTIVE) cov_ = covered_through_glass(sc, L, o_, d_, h_);
                        printf("gpu:   P=(%.9g %.9g %.9g) light %u sample %u u=%.9g v=%.9g dist %.9g class %u filter %s, exact: hit %d t=%.9g covered %d\n",
                               P.x, P.y, P.z, li, s, u, v, x_, rec_class, unresolved ? "undecided" : (covered ? "blocked" : "free"), h_.idx, h_.t,
                               (int)cov_);
                    }
                }
#endif
                if (act && !unresolved) {
                    V3 o, d;
                    float x;
                    shadow_ray(lc, P, u, v, o, d, x);
                    const Hit h = closest_hit(sc, o, d, x);
                    const bool exact_opaque = h.idx >= 0 && h.mat_type != CRT_MAT_REFRACTIVE;
                    const bool exact_free = h.idx < 0;
                    ++decided;
                    if (covered ? !exact_opaque : !exact_free) {
                        ++bad;
#ifdef CRT_AUDIT_TRACE // build with -DCRT_AUDIT_TRACE to see the offending rays
                        printf("audit: record %u light %u sample %u P=(%.9g %.9g %.9g) u=%.9g v=%.9g decided %s (record class %u), "
                               "exact hit %d mat %u t=%.9g of %.9g, origin (%.9g %.9g %.9g) dir (%.9g %.9g %.9g)\n",
                               wi, li, s, P.x, P.y, P.z, u, v, covered ? "blocked" : "free", rec_class, h.idx, h.mat_type, h.t, x,
                               o.x, o.y, o.z, d.x, d.y, d.z);
#endif
                    }
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        decided += __shfl_down(decided, off, 64);
        bad += __shfl_down(bad, off, 64);
    }
    if (lane == 0) {
        atomicAdd(&w.r.wave_tallies[2], decided);
        atomicAdd(&w.r.wave_tallies[3], bad);
    }
}

// -------------------------------------------------------------------------------------- resolve
constexpr uint32_t kGlassCap = 128; // < 64 left over + 64 new entries per group of rays

// Runs the refraction walk of Light::_isCovered (Light.hpp:47-80) on up to 64 queued glass hits.
DI void glass_batch(const WfArgs& w, ScenePtr sc, float (*gq_f)[kGlassCap], uint32_t (*gq_u)[kGlassCap], uint32_t& gq,
                    uint32_t lane) {
    const uint32_t n = gq < 64u ? gq : 64u;
    const uint32_t base = gq - n;
    __syncthreads();
    const bool act = lane < n;
    const uint32_t slot = base + (act ? lane : 0u);
    const V3 o = mk3(gq_f[0][slot], gq_f[1][slot], gq_f[2][slot]);
    const V3 d = mk3(gq_f[3][slot], gq_f[4][slot], gq_f[5][slot]);
    Hit h;
    h.mat_type = CRT_MAT_REFRACTIVE;
    h.t = gq_f[6][slot];
    h.u = gq_f[7][slot];
    h.v = gq_f[8][slot];
    const uint32_t key = gq_u[0][slot];
    h.idx = (int)gq_u[1][slot];
    const uint32_t my_light = key >> kRecBits;
    const uint32_t n_lights = sc->n_lights;
    for (uint32_t li = 0; li < n_lights; ++li) {
        const bool mine = act && my_light == li;
        if (!__any(mine)) continue;
        const DevLight CRT_CONST* L = &sc->lights[li];
        if (mine && covered_through_glass(sc, L, o, d, h))
            atomicAdd(&w.shadow_cnt[(size_t)li * w.cap_work + (key & kRecMask)], 1u);
    }
    gq = base;
    __syncthreads();
}

__global__ void __launch_bounds__(64, 4) wf_resolve_kernel(WfArgs w) {
    ScenePtr sc = (ScenePtr)w.r.scene;
    const uint32_t lane = threadIdx.x;
    const size_t cw = w.cap_work;
    const uint32_t n_lights = sc->n_lights;
    // rays whose closest hit is glass wait here until 64 of them can walk together
    __shared__ float gq_f[9][kGlassCap];    // origin, direction, first hit (t, u, v)
    __shared__ uint32_t gq_u[2][kGlassCap]; // queue key (record | light << kRecBits), primitive
    uint32_t gq = 0;                        // wave-uniform fill
    // the queue's fill is whatever the filter kernel left in the device counter
    const uint32_t qcount = min(w.ctrs[WF_CTR_QUEUE], w.qcap);
    uint32_t glass_tally = 0;
    // (a ray that meets glass costs several times one that does not)
    GroupFeed feed = feed_begin(w, CRT_MAX_LIGHTS, (qcount + 63u) / 64u);
    for (uint32_t rg = 0; feed_next(feed, lane, rg);) {
        const uint32_t base = rg * 64u;
        bool act = base + lane < qcount;
        const size_t idx = (size_t)base + lane;
        uint32_t key = 0;
        float u = 0.0f, v = 0.0f;
        V3 P = mk3(0, 0, 0);
        if (act) key = w.queue[idx];
        act = act && key != kNoRay;
        if (act) {
            u = u2f(w.queue[(size_t)w.qcap + idx]);
            v = u2f(w.queue[2 * (size_t)w.qcap + idx]);
            const uint4 pt = rec_point(w)[key & kRecMask];
            P = mk3(u2f(pt.x), u2f(pt.y), u2f(pt.z));
        } else {
            key = 0u;
        }
        const uint32_t my_light = key >> kRecBits;
        bool glass = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
        Hit h;
        h.idx = -1;
        h.mat_type = 0;
        h.t = h.u = h.v = 0.0f;
        for (uint32_t li = 0; li < n_lights; ++li) {
            const bool mine = act && my_light == li;
            if (!__any(mine)) continue;
            const DevLight CRT_CONST* L = &sc->lights[li];
            const LightC lc = load_lightc(L);
            if (mine) {
                PROF_LANES(19);
                float x;
                shadow_ray(lc, P, u, v, o, d, x);
                // first pass of Light::_isCovered (Light.hpp:35-46)
                h = closest_hit_masked(sc, o, d, x, shadow_candidates(sc, li, P));
                if (h.idx >= 0) {
                    if (h.mat_type != CRT_MAT_REFRACTIVE) atomicAdd(&w.shadow_cnt[(size_t)li * cw + (key & kRecMask)], 1u);
                    else glass = true;
                }
            }
        }
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
                gq_u[0][slot] = key;
                gq_u[1][slot] = (uint32_t)h.idx;
            }
            gq += (uint32_t)__popcll(gm);
            glass_tally += (uint32_t)__popcll(gm);
            if (gq >= 64u) glass_batch(w, sc, gq_f, gq_u, gq, lane);
        }
    }
    while (gq != 0u) glass_batch(w, sc, gq_f, gq_u, gq, lane);
    if (lane == 0 && glass_tally != 0u) atomicAdd(&w.r.wave_tallies[1], (unsigned long long)glass_tally);
}

// ---------------------------------------------------------------------------------------- shade
// One lane per record, workgroups of 256 = one chunk of record indices (kWorkChunk); the grid covers what the launch
// can produce at most and workgroups beyond the round's record count (read from the device counter) leave at once.
// The kernel is bound by its scattered 16-byte result stores, not by arithmetic (with the LTC evaluation compiled
// out it takes as long); measured per workgroup size 64 / 128 / 256 / 512 / 1024: 3.4 / 3.2 / 3.0 / 3.2 / 3.5 ms
// per 720x720 gl_iter 1 frame, and several groups per wave in a loop are no faster than one.
// CLIP: some light takes the horizon-clipping LTC branch (crt_light::horizonClip).
template <bool CLIP>
DI void shade_record(const WfArgs& w, ScenePtr sc, uint32_t wi);
#ifndef CRT_SHADE_WAVES
#define CRT_SHADE_WAVES 4
#endif
#ifndef CRT_SHADE_BLOCK
#define CRT_SHADE_BLOCK 256
#endif
template <bool CLIP>
__global__ void __launch_bounds__(CRT_SHADE_BLOCK, CRT_SHADE_WAVES) wf_shade_kernel(WfArgs w) {
    const uint32_t wi = blockIdx.x * (uint32_t)CRT_SHADE_BLOCK + threadIdx.x;
    if (wi >= records_handed_out(w)) return;
    shade_record<CLIP>(w, (ScenePtr)w.r.scene, wi);
}
template <bool CLIP>
DI void shade_record(const WfArgs& w, ScenePtr sc, uint32_t wi) {
    const size_t cw = w.cap_work;
    const uint4 pt = rec_point(w)[wi];
    const uint4* sf = rec_surface(w) + wi;
    const uint4 s0 = sf[0], s1 = sf[cw], s2 = sf[2 * cw];
    const V3 P = mk3(u2f(pt.x), u2f(pt.y), u2f(pt.z));
    const V3 N = mk3(u2f(s0.x), u2f(s0.y), u2f(s0.z));
    const V3 view = mk3(u2f(s1.x), u2f(s1.y), u2f(s1.z));
    const V4 att = V4{u2f(s2.x), u2f(s2.y), u2f(s2.z), u2f(s1.w)};
    const uint32_t gword = pt.w;
    if (gword == kNoRecord) return;
    const uint32_t geom = gword & 0xffu; // bits 8..13: the record's ordinal in a cooperative batch
    const uint32_t slot = s0.w;
    const DevGeomShade CRT_CONST* gs = &sc->shade[geom];
    V4 direct = V4{0, 0, 0, 0}; // LightList::calculateColor (LightList.hpp:23-36)
    const uint32_t n_lights = sc->n_lights;
    for (uint32_t li = 0; li < n_lights; ++li) {
        const DevLight CRT_CONST* L = &sc->lights[li];
        if (L->kind != CRT_LIGHT_AREA) {
            direct = direct + point_light_color(sc, L, gs, P, N, view);
        } else if (!L->initialized) { // AreaLight.hpp:586
            V3 amb = mk3(gs->ambient[0], gs->ambient[1], gs->ambient[2]) * ld3c(L->color);
            direct = direct + V4{amb.x, amb.y, amb.z, 1.0f};
        } else {
            const uint32_t S = L->shadow_samples;
            const uint32_t shadowed = w.shadow_cnt[(size_t)li * cw + wi];
            float visibility = 1.0f - ((float)shadowed / (float)S); // AreaLight.hpp:527
#ifdef CRT_SHADE_TRACE // build with -DCRT_SHADE_TRACE=<pixel index j*nx+i>: the shadow counts of that pixel's shaded points
            {
                const uint32_t lp = (uint32_t)slot_pixel(w, slot);
                const PixelMap pm = map_pixel(w.r, lp / 64u, lp % 64u);
                if (pm.valid && pm.j * w.r.nx + pm.i == (uint32_t)(CRT_SHADE_TRACE))
                    printf("gpu: P=(%.9g %.9g %.9g) light at (%.9g %.9g %.9g): %u of %u shadow rays covered\n", P.x, P.y, P.z,
                           L->pos[0], L->pos[1], L->pos[2], shadowed, S);
            }
#endif
#ifdef CRT_ABLATE_SHADE // measuring aid: the kernel's loads and store without the LTC arithmetic
            direct = direct + V4{P.x + N.x + view.x, P.y + N.y + view.y, P.z + N.z + view.z, visibility + gs->ambient[0]};
#else
            direct = direct + area_light_shade<CLIP>(w.r.ltc, L, gs, P, N, view, visibility);
#endif
        }
    }
    const V4 c = att * direct; // final_color += attenuation * light (main.cu:195)
#ifdef CRT_ABLATE_SCATTER // measuring aid (wrong image): results stored in record order, what the kernel takes without the permutation
    const size_t n_slots = (size_t)w.quota * w.n_lp;
    w.contrib[wi < n_slots ? wi : slot] = make_float4(c.x, c.y, c.z, c.w);
#else
    w.contrib[slot] = make_float4(c.x, c.y, c.z, c.w);
#endif
}

// ------------------------------------------------------------------------------------------ sum
// AHEAD: slots fetched four at a time (small launches, where the chain of dependent loads is what the
// kernel waits for; a large launch has enough waves to hide it and only pays for the extra selects).
template <bool AHEAD>
__global__ void __launch_bounds__(64) wf_sum_kernel(WfArgs w, uint32_t first) {
    const RenderArgs& a = w.r;
    const uint32_t lane = threadIdx.x, unit = blockIdx.x;
    const size_t n_lp = w.n_lp;
    const size_t lp = (size_t)unit * 64u + lane;
    const uint32_t cnt = w.slot_count[lp];
    if (cnt == 0u && !first) return;
    V4 final_c = V4{0, 0, 0, 0}, acc = V4{0, 0, 0, 0}, c0 = V4{0, 0, 0, 0};
    bool all_equal = true;
    if (!first) {
        const float4 f = w.px_acc[lp], g = w.px_acc[n_lp + lp], h = w.px_acc[2 * n_lp + lp];
        final_c = V4{f.x, f.y, f.z, f.w};
        acc = V4{g.x, g.y, g.z, g.w};
        c0 = V4{h.x, h.y, h.z, h.w};
        all_equal = (w.aa_verdict[lp] >> 8) & 1u;
    }
    const uint32_t aa_iter = a.aa_iter > 4u ? 4u : a.aa_iter; // main.cu:233
    uint32_t verdict = AA_PENDING;
    // AHEAD: tag and contribution alike are fetched four slots at a time (a slot that is not an addition
    // has an unused, but allocated, contribution) and consumed in order: one round trip to memory per four
    // slots instead of two per slot.
    constexpr uint32_t kAhead = AHEAD ? 4 : 1;
    uint32_t tags[kAhead];
    float4 contribs[kAhead];
    for (uint32_t k = 0; k < cnt; ++k) {
        if (k % kAhead == 0u) {
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                const uint32_t kk = k + j < cnt ? k + j : cnt - 1u;
                tags[j] = w.tags[slot_at(w, kk, lp)];
                contribs[j] = w.contrib[slot_at(w, kk, lp)];
            }
        }
        uint32_t tag = tags[0];
        float4 c = contribs[0];
#pragma unroll
        for (uint32_t j = 1; j < kAhead; ++j)
            if (k % kAhead == j) {
                tag = tags[j];
                c = contribs[j];
            }
        const uint32_t type = tag & 3u;
        if (type == TAG_SKIP) continue;
        if (type == TAG_ADD) {
            final_c = final_c + V4{c.x, c.y, c.z, c.w};
        } else if (type == TAG_END_CALL) {
            // aa_color's use of a finished color() call (main.cu:254-300)
            const uint32_t level = (tag >> 2) & 7u, corner = (tag >> 5) & 3u;
            const float mult = 1.0f / (float)(1u << (2u * (aa_iter - level)));
            if (level == 0u) {
                acc = acc + final_c * mult;
            } else if (level == 1u) {
                acc = acc + (final_c * mult) * 0.25f;
            } else {
                if (corner == 0u) {
                    c0 = final_c;
                    all_equal = true;
                } else if (all_equal && !equal4(final_c, c0)) {
                    all_equal = false;
                }
                if (corner == 3u) {
                    if (all_equal) acc = acc + c0 * mult;
                    verdict = all_equal ? AA_ALL_EQUAL : AA_SUBDIVIDE; // the trace kernel waits for this
                }
            }
            final_c = V4{0, 0, 0, 0};
        } else {
            // render() epilogue (main.cu:339-346)
            const PixelMap pm = map_pixel(a, unit, lane);
            reinterpret_cast<float4*>(a.out)[pm.out_index] =
                make_float4(saturate_c(acc.x), saturate_c(acc.y), saturate_c(acc.z), saturate_c(acc.w));
        }
    }
    w.px_acc[lp] = make_float4(final_c.x, final_c.y, final_c.z, final_c.w);
    w.px_acc[n_lp + lp] = make_float4(acc.x, acc.y, acc.z, acc.w);
    w.px_acc[2 * n_lp + lp] = make_float4(c0.x, c0.y, c0.z, c0.w);
    w.aa_verdict[lp] = verdict | (all_equal ? 0x100u : 0u);
}

int g_resident_cus[64] = {0};

// Pairs of HIP events around kernel launches; read back once the stream has been synchronised.
struct EventTimer {
    struct Span {
        hipEvent_t a, b;
        uint32_t kernel;
    };
    std::vector<Span> pool;
    size_t used = 0;
    Span* begin(uint32_t kernel, hipStream_t stream) {
        if (used == pool.size()) {
            Span s;
            if (hipEventCreate(&s.a) != hipSuccess) return nullptr;
            if (hipEventCreate(&s.b) != hipSuccess) {
                (void)hipEventDestroy(s.a);
                return nullptr;
            }
            pool.push_back(s);
        }
        Span* s = &pool[used++];
        s->kernel = kernel;
        (void)hipEventRecord(s->a, stream);
        return s;
    }
    static void end(Span* s, hipStream_t stream) {
        if (s) (void)hipEventRecord(s->b, stream);
    }
    void collect(WfWorkspace& ws) { // the stream is idle
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pool[i].a, pool[i].b) == hipSuccess) {
                ws.kernel_ms[pool[i].kernel] += ms;
                ws.kernel_launches[pool[i].kernel] += 1;
            }
        }
        used = 0;
    }
    ~EventTimer() {
        for (auto& s : pool) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
    }
};
#define WF_TIMED(kernel_id, launch_stmt)                                                    \
    do {                                                                                    \
        EventTimer::Span* span_ = timer ? timer->begin(kernel_id, stream) : nullptr;        \
        launch_stmt;                                                                        \
        EventTimer::end(span_, stream);                                                     \
    } while (0)

// Table for rng_jump: images of every nibble value of every state word under `nsteps` XORWOW steps.
void build_jump_table(uint32_t nsteps, uint32_t* tab) {
    for (uint32_t pos = 0; pos < 40u; ++pos) {
        uint32_t basis[4][5]; // images of the nibble's four bits
        for (uint32_t bit = 0; bit < 4u; ++bit) {
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[pos / 8u] = 1u << (4u * (pos % 8u) + bit);
            for (uint32_t n = 0; n < nsteps; ++n) { // the xorshift part of rng_step
                const uint32_t t = v[0] ^ (v[0] >> 2);
                v[0] = v[1];
                v[1] = v[2];
                v[2] = v[3];
                v[3] = v[4];
                v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
            }
            for (int wd = 0; wd < 5; ++wd) basis[bit][wd] = v[wd];
        }
        for (uint32_t val = 0; val < 16u; ++val)
            for (int wd = 0; wd < 5; ++wd) {
                uint32_t x = 0;
                for (uint32_t bit = 0; bit < 4u; ++bit)
                    if ((val >> bit) & 1u) x ^= basis[bit][wd];
                const uint32_t e = pos * 16u + val;
                if (wd < 4) tab[4u * e + (uint32_t)wd] = x;
                else tab[4u * kJumpEntries + e] = x;
            }
    }
}
// Table of "jump a, then jump b": every entry of table a, itself a generator state, is put through table b.
void compose_jump_tables(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    for (uint32_t en = 0; en < kJumpEntries; ++en) {
        const uint32_t in[5] = {a[4u * en], a[4u * en + 1u], a[4u * en + 2u], a[4u * en + 3u], a[4u * kJumpEntries + en]};
        uint32_t o[5] = {0, 0, 0, 0, 0};
        for (uint32_t wd = 0; wd < 5u; ++wd)
            for (uint32_t nb = 0; nb < 8u; ++nb) {
                const uint32_t idx = (wd * 8u + nb) * 16u + ((in[wd] >> (4u * nb)) & 15u);
                for (uint32_t k = 0; k < 4u; ++k) o[k] ^= b[4u * idx + k];
                o[4] ^= b[4u * kJumpEntries + idx];
            }
        for (uint32_t k = 0; k < 4u; ++k) out[4u * en + k] = o[k];
        out[4u * kJumpEntries + en] = o[4];
    }
}

// The five-bit form of a jump table (rng_jump5): every chunk value, as a generator state, put through the nibble table.
void regroup_jump_table5(const uint32_t* nib, uint32_t* out) {
    for (uint32_t c = 0; c < 32u; ++c)
        for (uint32_t val = 0; val < 32u; ++val) {
            uint32_t in[5] = {0, 0, 0, 0, 0};
            for (uint32_t b = 0; b < 5u; ++b)
                if ((val >> b) & 1u) in[(5u * c + b) >> 5] |= 1u << ((5u * c + b) & 31u);
            uint32_t o[5] = {0, 0, 0, 0, 0};
            for (uint32_t wd = 0; wd < 5u; ++wd)
                for (uint32_t nb = 0; nb < 8u; ++nb) {
                    const uint32_t idx = (wd * 8u + nb) * 16u + ((in[wd] >> (4u * nb)) & 15u);
                    for (uint32_t k = 0; k < 4u; ++k) o[k] ^= nib[4u * idx + k];
                    o[4] ^= nib[4u * kJumpEntries + idx];
                }
            const uint32_t e = c * 32u + val;
            for (uint32_t k = 0; k < 4u; ++k) out[4u * e + k] = o[k];
            out[4u * kJump5Entries + e] = o[4];
        }
}

// Result slots one pixel can fill without a break: per color() call one per popped ray that can end in a hit
// or a miss (1 + ind + ind^2 + ..) and one end-of-call marker; one end-of-pixel marker.  Without a break a
// pixel makes 1 (aa_iter 0) or 4 (aa_iter 1) color() calls, else a level-2 task's four subtasks of four calls
// before it has to wait for a verdict on its colours.
uint64_t slots_per_pixel(const RenderArgs& a) {
    const uint64_t calls = a.aa_iter == 0u ? 1u : (a.aa_iter == 1u ? 4u : 16u);
    uint64_t rays = 1, level = 1;
    for (uint32_t g = 0; g < a.gl_iter && rays < (1ull << 32); ++g) {
        level *= a.ind_rays;
        rays += level;
    }
    return calls * (rays + 1u) + 1u;
}
constexpr uint32_t kMinQuota = 8;

} // namespace

// A shade record's index shares a 32-bit word with its light (29 + 3 bits), and every pixel slot needs at
// least 8 result slots plus its share of partly used record chunks per round: launches of more than about
// 31 million pixel slots (for example 5800 x 5800) do not fit.
bool wavefront_supports(const RenderArgs& a) {
    const uint64_t n_lp = (uint64_t)a.n_units * 64u;
    if (n_lp == 0) return true;
    const uint64_t chunk_slack = (a.cooperative ? kCoopSplit : 1u) * 2u * kWorkChunk / 64u;
    return (kRecLimit - 1u) / n_lp >= chunk_slack + 1u + kMinQuota;
}

// Whether two launches map work units to pixels and rays the same way and walk trees of the same shape
// (everything wf_cost_kernel reads, apart from the scene's contents).
static bool same_pixel_mapping(const RenderArgs& a, const RenderArgs& b) {
    return a.scene == b.scene && a.n_units == b.n_units && a.nx == b.nx && a.ny == b.ny && a.ref_iter == b.ref_iter &&
           a.gl_iter == b.gl_iter && a.ind_rays == b.ind_rays && a.mode == b.mode && a.tiles_x == b.tiles_x &&
           a.tiles_y == b.tiles_y && a.rank == b.rank && a.world == b.world && a.tx == b.tx && a.ty == b.ty &&
           a.blocks_x == b.blocks_x && a.blocks_y == b.blocks_y && a.win_bx == b.win_bx && a.win_by == b.win_by &&
           a.start_bx == b.start_bx && a.start_by == b.start_by;
}

void wavefront_release(WfWorkspace& ws) {
    (void)hipFree(ws.ctrs);
    (void)hipHostFree(ws.ctrs_host);
    (void)hipFree(ws.state);
    (void)hipFree(ws.frames);
    (void)hipFree(ws.tags);
    (void)hipFree(ws.contrib);
    (void)hipFree(ws.slot_count);
    (void)hipFree(ws.px_acc);
    (void)hipFree(ws.aa_verdict);
    (void)hipFree(ws.work);
    (void)hipFree(ws.shadow_cnt);
    (void)hipFree(ws.queue);
    (void)hipFree(ws.jump_dev);
    (void)hipFree(ws.coop_tabs);
    (void)hipFree(ws.draw_tabs);
    (void)hipFree(ws.backup);
    delete static_cast<EventTimer*>(ws.timer);
    (void)hipFree(ws.draws);
    if (ws.side_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(ws.side_stream));
    if (ws.fork_event) (void)hipEventDestroy(static_cast<hipEvent_t>(ws.fork_event));
    if (ws.join_event) (void)hipEventDestroy(static_cast<hipEvent_t>(ws.join_event));
    if (ws.round_events) {
        hipEvent_t* ev = static_cast<hipEvent_t*>(ws.round_events);
        for (uint32_t k = 0; k < WF_CTR_RING; ++k)
            if (ev[k]) (void)hipEventDestroy(ev[k]);
        delete[] ev;
    }
    (void)hipFree(ws.unit_cost);
    (void)hipFree(ws.unit_order);
    (void)hipHostFree(ws.unit_host);
    const bool timing = ws.timing, async_error = ws.async_error;
    const uint64_t redone = ws.frames_redone;
    const uint32_t max_rounds = ws.max_rounds;
    ws = WfWorkspace();
    ws.timing = timing;
    ws.frames_redone = redone;
    ws.async_error = async_error;
    ws.max_rounds = max_rounds;
}

bool wavefront_collect_async(WfWorkspace& ws, bool wait) {
    hipEvent_t* ev = static_cast<hipEvent_t*>(ws.round_events);
    for (uint32_t k = 0; k < WF_CTR_RING && ev && ws.ctrs_host; ++k) {
        if (!((ws.async_pending >> k) & 1u)) continue;
        const hipError_t e = wait ? hipEventSynchronize(ev[k]) : hipEventQuery(ev[k]);
        if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            continue;
        }
        const uint32_t* host = ws.ctrs_host + (size_t)k * WF_CTR_COUNT;
        if (e != hipSuccess || host[WF_CTR_ERROR] != 0u || host[WF_CTR_QOVERFLOW] != 0u) ws.async_error = true;
        ws.async_pending &= ~(1u << k);
    }
    const bool bad = ws.async_error;
    ws.async_error = false;
    return bad;
}

// Frames left running on another stream share this context's workspace, counters and generator states: what is
// queued on `stream` next is ordered behind the latest of them.  (Frames that returned after their last round have
// been waited for.)
hipError_t wavefront_order_behind_async(WfWorkspace& ws, hipStream_t stream) {
    if (ws.async_pending != 0u && ws.last_stream != (void*)stream && ws.round_events) {
        const uint32_t latest = (ws.async_next + WF_CTR_RING - 1u) % WF_CTR_RING;
        const hipError_t e = hipStreamWaitEvent(stream, static_cast<hipEvent_t*>(ws.round_events)[latest], 0);
        if (e != hipSuccess) return e;
    }
    ws.last_stream = (void*)stream;
    return hipSuccess;
}

hipError_t wavefront_reserve(WfWorkspace& ws, const RenderArgs& args, uint32_t n_lights, size_t budget_bytes) {
    const uint32_t n_lp = args.n_units * 64u;
    if (n_lp == 0) return hipSuccess;
    const uint32_t npix = args.nx * args.ny;
    uint64_t want = slots_per_pixel(args);
    const uint32_t lights = n_lights ? n_lights : 1u;
    // bytes per pixel slot and quota unit: tag 1 + contribution 16 + shade record planes + shadow counts
    const uint64_t per_slot = 1u + 16u + 4u * WF_WORK_PLANES + 4u * lights;
    uint64_t quota_max = budget_bytes / (per_slot * n_lp);
    if (quota_max < 16u) quota_max = 16u; // what a round needs to make progress, whatever the limit says
    if (want > quota_max) want = quota_max;
    const uint64_t chunk_slack = (args.cooperative ? kCoopSplit : 1u) * 2u * kWorkChunk / 64u; // partly used chunks, per pixel slot
    if ((want + chunk_slack + 1u) * n_lp >= kRecLimit) { // record index and light share one word
        const uint64_t room = (kRecLimit - 1u) / n_lp;
        if (room < chunk_slack + 1u + kMinQuota) return hipErrorOutOfMemory; // see wavefront_supports
        want = room - chunk_slack - 1u;
    }
    const uint32_t quota = (uint32_t)want;
    const uint32_t levels = args.gl_iter ? args.gl_iter : 1u;
    if (ws.n_lp == n_lp && ws.quota >= quota && ws.gl_levels >= levels && ws.n_lights >= lights &&
        ws.max_shadow_rays >= args.max_shadow_rays && ws.npix == npix && (!args.cooperative || ws.draws != nullptr))
        return hipSuccess;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    if (wavefront_collect_async(ws, true)) ws.async_error = true; // kept across the release for the next call that looks
    wavefront_release(ws);
    // + every trace wave's two partly used chunks; a multiple of the chunk size so that both streams stay aligned
    const uint32_t trace_waves = args.cooperative ? args.n_units * kCoopSplit : args.n_units;
    const uint32_t cap_work = ((quota * n_lp + trace_waves * kWorkChunk + kWorkChunk - 1u) / kWorkChunk) * kWorkChunk;
    // the queue of undecided shadow rays: room for every ray of a round if that is less than 2^30 rays
    // (12.9 GB); normally a few per cent of the rays get there
    uint64_t qwant = (uint64_t)cap_work * (args.max_shadow_rays ? args.max_shadow_rays : 1u) +
                     (uint64_t)kMaxFilterWaves * 2048u; // + the piece every filter wave may leave partly unused
    if (qwant > (1ull << 30)) qwant = 1ull << 30;
    if (qwant > budget_bytes / 36u) qwant = budget_bytes / 36u; // at most a third of the budget (12 B per ray)
    if (qwant < (1ull << 20)) qwant = 1ull << 20;
    const uint32_t qcap = (uint32_t)qwant;
    size_t total = 0;
    auto grab = [&](void** p, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(p, bytes);
        total += bytes;
    };
    grab((void**)&ws.ctrs, (WF_CTR_COUNT + WF_CURSOR_WORDS) * sizeof(uint32_t));
    grab((void**)&ws.state, (size_t)WF_STATE_PLANES * n_lp * 4u);
    grab((void**)&ws.frames, (size_t)levels * WF_FRAME_PLANES * n_lp * 4u);
    grab((void**)&ws.tags, (size_t)quota * n_lp);
    grab((void**)&ws.contrib, (size_t)quota * n_lp * 16u);
    grab((void**)&ws.slot_count, (size_t)n_lp * 4u);
    grab((void**)&ws.px_acc, (size_t)3u * n_lp * 16u);
    grab((void**)&ws.aa_verdict, (size_t)n_lp * 4u);
    grab((void**)&ws.work, (size_t)WF_WORK_PLANES * cap_work * 4u);
    grab((void**)&ws.shadow_cnt, (size_t)lights * cap_work * 4u);
    grab((void**)&ws.queue, (size_t)3u * qcap * 4u);
    grab((void**)&ws.jump_dev, (size_t)kTraceJumpWords * 4u);
    grab((void**)&ws.backup, ((size_t)8u * npix + 8u) * 4u);
    grab((void**)&ws.unit_cost, (size_t)args.n_units * 4u);
    grab((void**)&ws.unit_order, (size_t)args.n_units * 4u);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ws.unit_host, (size_t)2u * args.n_units * sizeof(uint32_t), hipHostMallocDefault);
    if (args.cooperative) {
        grab((void**)&ws.draws, (size_t)2u * CRT_RAY_STACK * n_lp * 4u);
        grab((void**)&ws.coop_tabs, (size_t)kCoopTables * kJumpWords * 4u);
    }
    if (e == hipSuccess)
        e = hipHostMalloc((void**)&ws.ctrs_host, (size_t)WF_CTR_RING * WF_CTR_COUNT * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) {
        hipEvent_t* ev = new hipEvent_t[WF_CTR_RING]();
        ws.round_events = ev;
        for (uint32_t k = 0; k < WF_CTR_RING && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        wavefront_release(ws);
        return e;
    }
    ws.n_lp = n_lp;
    ws.quota = quota;
    ws.gl_levels = levels;
    ws.cap_work = cap_work;
    ws.n_lights = lights;
    ws.qcap = qcap;
    ws.max_shadow_rays = args.max_shadow_rays;
    ws.npix = npix;
    ws.bytes = total;
    return hipSuccess;
}

namespace {

// One round's kernels behind the trace kernel, sized without the host knowing the round's counts: fixed grids
// that read the number of records and queued rays from the device counters.
struct Grids {
    uint32_t filter, resolve, shade;
    uint32_t qusable; // queue entries left when every filter wave wastes a whole piece
};

// The trace kernel(s) of one round.  The first n_coop units of the launch order (the most expensive ones) go
// through the cooperative kernel, the rest through the per-pixel one; when both run, the cooperative kernel
// goes to a side stream so that the two share the GPU.  (Measured on the half-frame launch, trace time: this
// way round, launched into an idle GPU after the pre-pass's host round trip, 5.2 ms; the kernels' streams
// swapped, or the launch queued behind the previous frame without the round trip, 5.9 ms.)
hipError_t launch_trace(const RenderArgs& args, WfWorkspace& ws, const WfArgs& w, uint32_t first, uint32_t n_coop,
                        hipStream_t stream, EventTimer* timer) {
    hipError_t e = hipSuccess;
    EventTimer::Span* span = timer ? timer->begin(WF_K_TRACE, stream) : nullptr;
    const dim3 tblock(64 * kTraceWaves);
    const uint32_t prim_lds = (args.n_geom ? args.n_geom : 1u) * (uint32_t)sizeof(PrimRow); // the kernel's primitive table
    const uint32_t rest = args.n_units - n_coop;
    hipStream_t cstream = stream;
    if (n_coop != 0u && rest != 0u) {
        cstream = static_cast<hipStream_t>(ws.side_stream);
        e = hipEventRecord(static_cast<hipEvent_t>(ws.fork_event), stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cstream, static_cast<hipEvent_t>(ws.fork_event), 0);
        if (e != hipSuccess) return e;
    }
    if (n_coop != 0u)
        hipLaunchKernelGGL(wf_trace_kernel<true>, dim3((n_coop * kCoopSplit + kTraceWaves - 1u) / kTraceWaves), tblock, prim_lds,
                           cstream, w, first, 0u, n_coop);
    if (rest != 0u)
        hipLaunchKernelGGL(wf_trace_kernel<false>, dim3((rest + kTraceWaves - 1u) / kTraceWaves), tblock, prim_lds, stream, w, first,
                           n_coop, args.n_units);
    if (cstream != stream) {
        e = hipEventRecord(static_cast<hipEvent_t>(ws.join_event), cstream);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, static_cast<hipEvent_t>(ws.join_event), 0);
        if (e != hipSuccess) return e;
    }
    EventTimer::end(span, stream);
    return hipSuccess;
}

// Renders the frame.  safe == false: rounds are queued back to back, every kernel takes its sizes from the
// device counters, the host only looks (one round late) at whether pixels are still unfinished.  If more
// shadow rays stay undecided than the queue holds, the device drops rays and raises WF_CTR_QOVERFLOW.
// safe == true: the host waits for each round's record count and runs filter and resolve over batches of
// records small enough for the queue to hold every ray of a batch.
hipError_t run_frame(const RenderArgs& args, WfWorkspace& ws, WfArgs& w, const Grids& g, uint32_t n_coop, bool safe,
                     bool leave_running, hipStream_t stream, EventTimer* timer, uint32_t flags_out[2]) {
    const bool coop = n_coop != 0u; // records may carry ordinals
    hipEvent_t* ev = static_cast<hipEvent_t*>(ws.round_events);
    hipError_t e = hipMemsetAsync(ws.ctrs, 0, WF_CTR_COUNT * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    flags_out[0] = flags_out[1] = 0u;
    const uint64_t rays_per_record = args.max_shadow_rays ? args.max_shadow_rays : 1u;
    // Measuring aid (crt_set_round_limit; bench.py --workload c5): stop after this many rounds to time the steady
    // state of a launch whose whole frame takes an hour.  The caller reports a frame cut short as such.
    const uint32_t max_rounds = ws.max_rounds ? ws.max_rounds : 0xffffffffu;
    uint32_t first = 1u;
    for (uint32_t round = 0;; ++round) {
        uint32_t* host = ws.ctrs_host + (size_t)(round % WF_CTR_RING) * WF_CTR_COUNT;
        e = hipMemsetAsync(ws.ctrs, 0, WF_CTR_ERROR * sizeof(uint32_t), stream); // the flags behind it are sticky
        if (e == hipSuccess) e = hipMemsetAsync(ws.ctrs + WF_CTR_COUNT, 0, WF_CURSOR_WORDS * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        // slot tags default to TAG_ADD (0): the trace kernel stores only the rare end-of-call / end-of-pixel marks
        e = hipMemsetAsync(ws.tags, 0, (size_t)w.quota * ws.n_lp, stream);
        if (e != hipSuccess) return e;
        e = launch_trace(args, ws, w, first, n_coop, stream, timer);
        if (e != hipSuccess) return e;
#ifdef CRT_LANE_PROFILE
        {
            unsigned long long prof[64];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_lane_prof), sizeof prof);
            static const char* names[32] = {"iteration (lanes walking)", "next sample / end of call", "child pop", "pop_ray", "", "quad uv test",
                                            "sphere normal", "quad normal", "reflect", "refract", "emit", "store_record", "", "frame push",
                                            "plane division", "sphere roots", "resolve: primitive test", "resolve: quad uv", "glass walk step",
                                            "resolve: ray"};
            for (int k = 0; k < 32; ++k)
                if (prof[2 * k] != 0ull)
                    printf("lanes: section %2d %-28s runs %12llu  mean active lanes %.1f\n", k, names[k] ? names[k] : "", prof[2 * k],
                           (double)prof[2 * k + 1] / (double)prof[2 * k]);
            unsigned long long zero[64] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_prof), zero, sizeof zero);
        }
#endif
        if (!safe) {
            if (coop) WF_TIMED(WF_K_FILTER, hipLaunchKernelGGL(wf_filter_kernel<true>, dim3(g.filter), dim3(64), 0, stream, w, 0u, kFromCounter));
            else WF_TIMED(WF_K_FILTER, hipLaunchKernelGGL(wf_filter_kernel<false>, dim3(g.filter), dim3(64), 0, stream, w, 0u, kFromCounter));
            if (args.audit) hipLaunchKernelGGL(wf_audit_kernel, dim3(g.filter), dim3(64), 0, stream, w, 0u, kFromCounter);
            WF_TIMED(WF_K_RESOLVE, hipLaunchKernelGGL(wf_resolve_kernel, dim3(g.resolve), dim3(64), 0, stream, w));
        } else {
            e = hipMemcpyAsync(host, ws.ctrs, WF_CTR_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            const uint64_t handed_out = (uint64_t)w.work_static + host[WF_CTR_WORK];
            const uint32_t n_work = handed_out < ws.cap_work ? (uint32_t)handed_out : ws.cap_work;
            uint64_t batch = g.qusable / rays_per_record / 64u * 64u;
            if (batch == 0) return hipErrorInvalidValue; // a single group's shadow rays exceed the queue
            if (batch > 0x40000000ull) batch = 0x40000000ull;
            for (uint32_t b0 = 0; b0 < n_work; b0 += (uint32_t)batch) {
                const uint32_t b1 = n_work - b0 > batch ? b0 + (uint32_t)batch : n_work;
                const uint32_t groups = (b1 - b0 + 63u) / 64u;
                const uint32_t fgrid = groups < g.filter ? groups : g.filter;
                e = hipMemsetAsync(ws.ctrs + WF_CTR_QUEUE, 0, sizeof(uint32_t), stream);
                if (e == hipSuccess) e = hipMemsetAsync(ws.ctrs + WF_CTR_COUNT, 0, WF_CURSOR_WORDS * sizeof(uint32_t), stream);
                if (e != hipSuccess) return e;
                if (coop) WF_TIMED(WF_K_FILTER, hipLaunchKernelGGL(wf_filter_kernel<true>, dim3(fgrid), dim3(64), 0, stream, w, b0, b1));
                else WF_TIMED(WF_K_FILTER, hipLaunchKernelGGL(wf_filter_kernel<false>, dim3(fgrid), dim3(64), 0, stream, w, b0, b1));
                if (args.audit) hipLaunchKernelGGL(wf_audit_kernel, dim3(fgrid), dim3(64), 0, stream, w, b0, b1);
                WF_TIMED(WF_K_RESOLVE, hipLaunchKernelGGL(wf_resolve_kernel, dim3(g.resolve), dim3(64), 0, stream, w));
            }
        }
        const dim3 sgrid((g.shade * 64u + CRT_SHADE_BLOCK - 1u) / CRT_SHADE_BLOCK), sblock(CRT_SHADE_BLOCK);
        if (args.horizon_clip) WF_TIMED(WF_K_SHADE, hipLaunchKernelGGL(wf_shade_kernel<true>, sgrid, sblock, 0, stream, w));
        else WF_TIMED(WF_K_SHADE, hipLaunchKernelGGL(wf_shade_kernel<false>, sgrid, sblock, 0, stream, w));
        if (args.n_units < 4096u) WF_TIMED(WF_K_SUM, hipLaunchKernelGGL(wf_sum_kernel<true>, dim3(args.n_units), dim3(64), 0, stream, w, first));
        else WF_TIMED(WF_K_SUM, hipLaunchKernelGGL(wf_sum_kernel<false>, dim3(args.n_units), dim3(64), 0, stream, w, first));
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        first = 0u;
        if (leave_running) {
            // one round, nothing that can overflow: nothing to wait for.  The round's flags still go to the host, into
            // the next slot of the ring, and are looked at by a later call (wavefront_collect_async).
            const uint32_t k = ws.async_next % WF_CTR_RING;
            if ((ws.async_pending >> k) & 1u) { // the frame four calls back: long finished
                e = hipEventSynchronize(ev[k]);
                if (e != hipSuccess) return e;
                (void)wavefront_collect_async(ws, false);
                if ((ws.async_pending >> k) & 1u) return hipErrorUnknown;
            }
            e = hipMemcpyAsync(ws.ctrs_host + (size_t)k * WF_CTR_COUNT, ws.ctrs, WF_CTR_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(ev[k], stream);
            if (e != hipSuccess) return e;
            ws.async_pending |= 1u << k;
            ws.async_next = k + 1u;
            return hipSuccess;
        }
        if (round + 1u >= max_rounds) break;
        if (safe) {
            if (host[WF_CTR_ACTIVE] == 0u) break;
            continue;
        }
        // this round's counters, flags included, to the host; the host reads them one round late
        e = hipMemcpyAsync(host, ws.ctrs, WF_CTR_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(ev[round % WF_CTR_RING], stream);
        if (e != hipSuccess) return e;
        if (ws.single_round) break;
        if (round >= 1u) {
            const uint32_t* prev = ws.ctrs_host + (size_t)((round - 1u) % WF_CTR_RING) * WF_CTR_COUNT;
            e = hipEventSynchronize(ev[(round - 1u) % WF_CTR_RING]);
            if (e != hipSuccess) return e;
            // nothing was left after the round before: the round just queued finds no work
            if (prev[WF_CTR_ACTIVE] == 0u || prev[WF_CTR_ERROR] != 0u || prev[WF_CTR_QOVERFLOW] != 0u) break;
        }
    }
    e = hipMemcpyAsync(ws.ctrs_host, ws.ctrs, WF_CTR_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    flags_out[0] = ws.ctrs_host[WF_CTR_ERROR];
    flags_out[1] = ws.ctrs_host[WF_CTR_QOVERFLOW];
    return hipSuccess;
}

} // namespace

hipError_t launch_wavefront(const RenderArgs& args, WfWorkspace& ws, hipStream_t stream, bool* overflow, bool* truncated,
                            bool no_redo, bool* needs_redo) {
    *overflow = false;
    *truncated = false;
    if (needs_redo) *needs_redo = false;
    if (args.n_units == 0) return hipSuccess;
    if (ws.timing && !ws.timer) ws.timer = new EventTimer();
    EventTimer* timer = ws.timing ? static_cast<EventTimer*>(ws.timer) : nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (g_resident_cus[dev] == 0) {
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        g_resident_cus[dev] = cus > 0 ? cus : 1;
    }
    const uint32_t cus = (uint32_t)g_resident_cus[dev];
    if (ws.npix != args.nx * args.ny) return hipErrorInvalidValue; // wavefront_reserve was given another image
    e = wavefront_order_behind_async(ws, stream);
    if (e != hipSuccess) return e;

    WfArgs w;
    w.r = args;
    w.n_lp = ws.n_lp;
    // What this launch can use of the workspace: a smaller tree than the workspace was sized for (the same
    // context rendering gl_iter 0 after gl_iter 1) fills fewer slots per pixel, and the grids and the per-round
    // clearing follow that, not the allocation.
    {
        const uint64_t slots = slots_per_pixel(args);
        w.quota = slots < ws.quota ? (uint32_t)slots : ws.quota;
    }
    w.cap_work = ws.cap_work; // the planes' stride stays the allocation's
    const uint32_t launch_waves = args.cooperative ? args.n_units * kCoopSplit : args.n_units;
    const uint64_t used64 = (uint64_t)w.quota * ws.n_lp + (uint64_t)launch_waves * kWorkChunk;
    const uint32_t cap_used = used64 < ws.cap_work ? (uint32_t)used64 : ws.cap_work; // record indices this launch can reach
    w.qcap = ws.qcap;
    w.ctrs = ws.ctrs;
    w.state = ws.state;
    w.frames = ws.frames;
    w.tags = ws.tags;
    w.contrib = reinterpret_cast<float4*>(ws.contrib);
    w.slot_count = ws.slot_count;
    w.px_acc = reinterpret_cast<float4*>(ws.px_acc);
    w.aa_verdict = ws.aa_verdict;
    w.work = ws.work;
    w.shadow_cnt = ws.shadow_cnt;
    w.queue = ws.queue;
    w.jump_tabs = ws.jump_dev;
    w.jump_n[0] = ws.jump_n[0];
    w.draws = ws.draws;
    w.coop_tabs = ws.coop_tabs;
    w.draw_tabs = ws.draw_tabs;
    w.draw_tabs_n = ws.draw_tabs_n;
    w.work_static = 0u;
    w.unit_order = nullptr;
    ws.single_round = args.aa_iter <= 1u && slots_per_pixel(args) <= ws.quota;
    const bool coop = args.cooperative != 0u && args.gl_iter != 0u;
    if (coop && (!ws.draws || !ws.coop_tabs)) return hipErrorInvalidValue; // wavefront_reserve was given other arguments
    uint32_t n_coop = 0; // units for the cooperative kernel: that share of the launch, at least one unit
    if (coop) {
        n_coop = (uint32_t)(((uint64_t)args.n_units * (args.cooperative > 1000u ? 1000u : args.cooperative) + 999u) / 1000u);
        if (n_coop > args.n_units) n_coop = args.n_units;
        if (n_coop != args.n_units && !ws.side_stream) {
            hipStream_t s2 = nullptr;
            hipEvent_t ev1 = nullptr, ev2 = nullptr;
            e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev1, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev2, hipEventDisableTiming);
            if (e != hipSuccess) return e;
            ws.side_stream = s2;
            ws.fork_event = ev1;
            ws.join_event = ev2;
        }
    }

    // Grids that do not depend on the round's counts, only on what the workspace can hold.
    Grids g;
    {
        const uint32_t groups = (cap_used + 63u) / 64u;
        g.filter = cus * 4u * CRT_WF_FILTER_WAVES;
        if (g.filter > kMaxFilterWaves) g.filter = kMaxFilterWaves;
        if (g.filter > groups) g.filter = groups;
        // resolve: as many waves as the device keeps resident (57 registers: eight per SIMD); they take the queue's
        // groups of 64 rays from a cursor
        uint64_t rgrid = (uint64_t)cus * 32u;
        if (rgrid > (ws.qcap + 63u) / 64u) rgrid = (ws.qcap + 63u) / 64u;
        g.resolve = (uint32_t)rgrid;
        g.shade = groups;
        // the filter waves reserve the queue in pieces: large enough to keep the atomics rare, small enough
        // that the waves' unused remainders stay below a quarter of the queue
        uint32_t qchunk = ws.qcap / (4u * g.filter) / 64u * 64u;
        w.qchunk = qchunk < 64u ? 64u : (qchunk > 2048u ? 2048u : qchunk);
        const uint64_t waste = (uint64_t)g.filter * w.qchunk;
        g.qusable = waste < ws.qcap ? (uint32_t)(ws.qcap - waste) : 0u;
    }

    // generator jump tables for this scene and these settings
    {
        const uint32_t want[1] = {2u * args.max_shadow_rays};
        for (int t = 0; t < 1; ++t) {
            if (ws.jump_n[t] == want[t]) continue;
            std::vector<uint32_t> tab(kJumpWords, 0u), tab5(kJump5Words, 0u);
            if (want[t] != 0u) build_jump_table(want[t], tab.data());
            if (CRT_JUMP5) regroup_jump_table5(tab.data(), tab5.data());
            const std::vector<uint32_t>& up = CRT_JUMP5 ? tab5 : tab;
            e = hipMemcpyAsync(ws.jump_dev, up.data(), (size_t)kTraceJumpWords * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream); // `tab` is pageable and about to go out of scope
            if (e != hipSuccess) return e;
            ws.jump_n[t] = want[t];
        }
        w.jump_n[0] = ws.jump_n[0];
        // one table per number of children an expanding hit can keep, 2a draws each
        const uint32_t n_child_tabs = args.gl_iter == 0u ? 0u : (args.ind_rays < CRT_RAY_STACK ? args.ind_rays : CRT_RAY_STACK);
        if (n_child_tabs != 0u && (ws.draw_tabs_n != n_child_tabs || !ws.draw_tabs)) {
            if (ws.draw_tabs) {
                e = hipStreamSynchronize(stream);
                if (e != hipSuccess) return e;
                (void)hipFree(ws.draw_tabs);
                ws.draw_tabs = nullptr;
                ws.draw_tabs_n = 0;
            }
            std::vector<uint32_t> tabs((size_t)n_child_tabs * kJumpWords, 0u);
            build_jump_table(2u, tabs.data());
            for (uint32_t t = 1; t < n_child_tabs; ++t)
                compose_jump_tables(tabs.data() + (size_t)(t - 1u) * kJumpWords, tabs.data(), tabs.data() + (size_t)t * kJumpWords);
            e = hipMalloc((void**)&ws.draw_tabs, tabs.size() * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpyAsync(ws.draw_tabs, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            ws.draw_tabs_n = n_child_tabs;
        }
        w.draw_tabs = ws.draw_tabs;
        w.draw_tabs_n = ws.draw_tabs_n;
        if (coop && ws.coop_n != want[0]) {
            std::vector<uint32_t> tabs((size_t)kCoopTables * kJumpWords, 0u);
            // table k-1 = k times the jump of one shaded point
            build_jump_table(want[0], tabs.data());
            for (uint32_t t = 1; t < kCoopTables; ++t)
                compose_jump_tables(tabs.data() + (size_t)(t - 1u) * kJumpWords, tabs.data(), tabs.data() + (size_t)t * kJumpWords);
            e = hipMemcpyAsync(ws.coop_tabs, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            ws.coop_n = want[0];
        }
    }
    // Launch order of the trace kernel's units (most expensive first).  Worth a host round trip only when
    // pixels take many steps and the launch is large enough to have late starters.
    // The order is reused while scene, camera and mapping stay the same (same estimate), except by mixed
    // launches: a half-frame launch with both trace kernels measured 0.6 ms slower when it is queued behind
    // the previous frame instead of entering an idle GPU after the pre-pass's host round trip.
    w.work_static = (args.n_units - n_coop) * kWorkChunk; // the per-pixel waves' first chunks
    const bool mixed = n_coop != 0u && n_coop != args.n_units;
    if (args.gl_iter != 0u && args.n_units > 1u && !mixed && ws.order_valid && same_pixel_mapping(args, ws.order_key)) {
        w.unit_order = ws.unit_order;
    } else if (args.gl_iter != 0u && args.n_units > 1u) {
        ws.order_valid = false;
        hipLaunchKernelGGL(wf_cost_kernel, dim3(args.n_units), dim3(64), 0, stream, w, ws.unit_cost);
        e = hipMemcpyAsync(ws.unit_host, ws.unit_cost, (size_t)args.n_units * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        const uint32_t* cost = ws.unit_host; // non-negative floats: their bit patterns order like the values
        uint32_t* order = ws.unit_host + args.n_units;
        for (uint32_t u = 0; u < args.n_units; ++u) order[u] = u;
        std::stable_sort(order, order + args.n_units, [cost](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
        e = hipMemcpyAsync(ws.unit_order, order, (size_t)args.n_units * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        w.unit_order = ws.unit_order;
        ws.order_key = args;
        ws.order_valid = true;
    }

    // Can the queue overflow at all?  Not when it holds every shadow ray of a round's records.
    const uint64_t rays_per_record = args.max_shadow_rays ? args.max_shadow_rays : 1u;
    const bool can_overflow = (uint64_t)cap_used * rays_per_record + (uint64_t)g.filter * w.qchunk > ws.qcap;
    const size_t npix = ws.npix;
    if (can_overflow && !no_redo) {
        // what a second rendering of this frame starts from: generator states, work counters, tallies
        e = hipMemcpyAsync(ws.backup, args.rng, npix * 6u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ws.backup + 6u * npix, args.counters, npix * 2u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ws.backup + 8u * npix, args.wave_tallies, 8u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    const bool leave_running = ws.single_round && !can_overflow && !timer && !args.audit && ws.max_rounds == 0u;
    // flags of earlier frames that were left running: all of them before a frame that uses the ring for its rounds
    if (wavefront_collect_async(ws, !leave_running)) *overflow = true;
    uint32_t flags[2] = {0u, 0u};
    e = run_frame(args, ws, w, g, n_coop, false, leave_running, stream, timer, flags);
    if (e != hipSuccess) return e;
    if (!leave_running && ws.max_rounds != 0u && ws.ctrs_host[WF_CTR_ACTIVE] != 0u) *truncated = true;
    if (flags[0] == 0u && flags[1] != 0u && can_overflow && no_redo) {
        if (needs_redo) *needs_redo = true; // the caller starts the frame again (its other pieces too)
        if (timer) timer->collect(ws);
        return hipSuccess;
    }
    if (flags[0] == 0u && flags[1] != 0u && can_overflow) {
        // rays were dropped: the same frame again from where it started, with launches that cannot overflow
        e = hipMemcpyAsync(args.rng, ws.backup, npix * 6u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(args.counters, ws.backup + 6u * npix, npix * 2u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(args.wave_tallies, ws.backup + 8u * npix, 8u * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        if (timer) timer->collect(ws);
        ws.frames_redone += 1;
        e = run_frame(args, ws, w, g, n_coop, true, false, stream, timer, flags);
        if (e != hipSuccess) return e;
    }
    if (flags[0] != 0u || flags[1] != 0u) *overflow = true;
    if (timer) timer->collect(ws);
    return hipSuccess;
}

} // namespace crt
