// C ABI of libcrt_hip.so (declarations and contracts: include/crt_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/crt_hip.h"
#include "crt_render.hpp"
#include "crt_scene.hpp"

// 2 x 64x64x4 float32 LTC tables, embedded from data/ltc_tables.f32 (crt_ltc_blob.cpp).
extern "C" const unsigned char crt_ltc_tables_blob[];
extern "C" const unsigned char crt_ltc_tables_blob_end[];

// Scratch of crt_render_static, the entry point that allocates nothing: the persistent kernel's work-unit
// cursor and wave tallies (first 256 bytes) and its two per-pixel work counters.  Every GPU the library runs
// on holds its own copy of this array until the process ends; crt_create looks up that copy's address once
// per device.  In host code the array's name is only a handle for that lookup, never an address to use.
#define CRT_STATIC_HEAD_WORDS 64u
static __device__ __align__(256) uint32_t scratch[CRT_STATIC_HEAD_WORDS + 2u * CRT_STATIC_MAX_PIXELS];
static uint32_t* g_scratch_ptr[64] = {nullptr}; // per device: device address of `scratch` there
// Per device: an event recorded behind the latest crt_render_static launch there, and the stream that launch went
// to.  A call on another stream (or from another context) first makes its stream wait for that event, so calls
// that would have overlapped on the shared array run one after the other instead.  Created by crt_create (the
// entry point itself allocates nothing); calls from two host threads at once still need the caller's own lock.
static hipEvent_t g_static_done[64] = {nullptr};
static void* g_static_stream[64] = {nullptr};
static bool g_static_used[64] = {false};

// The context's second host thread: it runs the second piece of a launch that is rendered in two pieces, and sleeps
// between frames (starting a thread per frame costs a few per cent of a 3 ms frame).
class PieceWorker {
public:
    PieceWorker() : th_([this] { loop(); }) {}
    ~PieceWorker() {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void submit(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = std::move(job);
            pending_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return done_; });
    }

private:
    void loop() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return pending_ || quit_; });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            pending_ = false;
            g.unlock();
            job();
            g.lock();
            done_ = true;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool pending_ = false, done_ = true, quit_ = false;
    std::thread th_; // last: the thread starts in the constructor and uses the members above
};

struct crt_ctx {
    int device = 0;
    uint32_t* scratch_ptr = nullptr; // g_scratch_ptr[device]
    char err[512] = {0};
    float* ltc_dev = nullptr;
    crt::DevScene* scene_dev = nullptr;
    crt::DevScene scene_host;
    bool has_scene = false;
    uint32_t* rng_dev = nullptr;      // 6 planes of npix words
    uint32_t* rng_init_d = nullptr;   // plane 5 (d) as seeded, for draw counting
    uint32_t* counters_dev = nullptr; // 2 planes of npix words: shade points, expansions
    uint32_t* unit_counter = nullptr; // work-unit cursor of the persistent render grid
    uint32_t rng_nx = 0, rng_ny = 0;
    int pipeline = CRT_PIPELINE_AUTO;
    int audit = 0;
    int trace_mode = CRT_TRACE_AUTO;
    int coop_permille = -1; // CRT_COOP_PERMILLE as read by crt_create (tools/mix_sweep.sh); -1: by launch size
    size_t workspace_limit = (size_t)96 << 30;
    crt::WfWorkspace wf;
    // Launches of one context share its generator states, work counters and workspace, so they run one after the
    // other whatever streams they are given: every launch records `last_done` behind itself, and a call on another
    // stream than the previous one first makes its stream wait for that event.
    hipEvent_t last_done = nullptr;
    void* last_stream = nullptr;
    bool launched = false;
    // A large launch through the wavefront pipeline is rendered as two interleaved pieces side by side (see
    // wants_pieces): a workspace, a stream and, while a frame runs, a host thread per piece.
    static constexpr uint32_t kPieces = 2;
    crt::WfWorkspace wf_piece[kPieces];
    hipStream_t piece_stream[kPieces] = {nullptr, nullptr};
    hipEvent_t piece_fork = nullptr, piece_join[kPieces] = {nullptr, nullptr};
    std::unique_ptr<PieceWorker> piece_worker; // started with the first launch that runs in pieces
    uint32_t* frame_backup = nullptr; // 8 planes of npix words + 8: generator states, work counters, tallies at the frame's start
    size_t frame_backup_npix = 0;
    bool piece_overflowed = false;       // the pieces of launch `piece_overflow_key` overflowed their ray queues
    crt::RenderArgs piece_overflow_key{};
    int piece_min_units = 1000; // launches of at least this many 64-pixel units are rendered in pieces; 0 = never (CRT_FRAME_PIECES_MIN_UNITS)
};

namespace {

int fail(crt_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return fail(ctx, CRT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));    \
    } while (0)

int bind(crt_ctx* ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return CRT_OK;
}

// Flags of frames that were left running, whichever workspace they ran in (waits for them).
bool collect_async_all(crt_ctx* ctx) {
    bool bad = crt::wavefront_collect_async(ctx->wf, true);
    for (crt::WfWorkspace& w : ctx->wf_piece) bad = crt::wavefront_collect_async(w, true) || bad;
    return bad;
}

int order_behind_last_launch(crt_ctx* ctx, hipStream_t stream) {
    if (ctx->launched && ctx->last_stream != (void*)stream) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->last_done, 0));
    return CRT_OK;
}

int mark_launched(crt_ctx* ctx, hipStream_t stream) {
    HIP_TRY(ctx, hipEventRecord(ctx->last_done, stream));
    ctx->last_stream = (void*)stream;
    ctx->launched = true;
    return CRT_OK;
}

int check_settings(crt_ctx* ctx, const crt_settings* st) {
    if (!st) return fail(ctx, CRT_ERR_INVALID, "settings is NULL");
    if (st->nx == 0 || st->ny == 0) return fail(ctx, CRT_ERR_INVALID, "nx and ny must be positive");
    if ((uint64_t)st->nx * st->ny > 0x7fffffffull) return fail(ctx, CRT_ERR_INVALID, "image too large");
    if (st->gl_iter > CRT_MAX_GL_ITER)
        return fail(ctx, CRT_ERR_UNSUPPORTED, "gl_iter %u exceeds CRT_MAX_GL_ITER (%u)", st->gl_iter, CRT_MAX_GL_ITER);
    if (st->gl_iter > 0 && st->ind_rays == 0)
        return fail(ctx, CRT_ERR_INVALID, "ind_rays must be positive when gl_iter > 0");
    if (!ctx->has_scene) return fail(ctx, CRT_ERR_NO_SCENE, "no scene uploaded");
    if (!ctx->rng_dev || ctx->rng_nx != st->nx || ctx->rng_ny != st->ny)
        return fail(ctx, CRT_ERR_NO_RNG, "RNG states missing or sized for %ux%u, not %ux%u", ctx->rng_nx, ctx->rng_ny,
                    st->nx, st->ny);
    return CRT_OK;
}

crt::RenderArgs base_args(crt_ctx* ctx, const crt_settings* st, float* out) {
    crt::RenderArgs a;
    memset(&a, 0, sizeof a);
    a.scene = ctx->scene_dev;
    a.ltc = ctx->ltc_dev;
    a.rng = ctx->rng_dev;
    a.counters = ctx->counters_dev;
    a.unit_counter = ctx->unit_counter;
    a.wave_tallies = reinterpret_cast<unsigned long long*>(ctx->unit_counter + 16);
    a.out = out;
    a.nx = st->nx;
    a.ny = st->ny;
    a.aa_iter = st->aa_iter;
    a.ref_iter = st->ref_iter;
    a.gl_iter = st->gl_iter;
    a.ind_rays = st->ind_rays;
    a.tiles_x = (st->nx + 7u) / 8u;
    a.tiles_y = (st->ny + 7u) / 8u;
    a.world = 1;
    a.audit = ctx->audit ? 1u : 0u;
    a.n_geom = ctx->scene_host.n_geom;
    for (uint32_t l = 0; l < ctx->scene_host.n_lights; ++l) {
        const crt::DevLight& L = ctx->scene_host.lights[l];
        if (L.kind == CRT_LIGHT_AREA && L.initialized) a.max_shadow_rays += L.shadow_samples;
        if (L.kind == CRT_LIGHT_AREA && L.initialized && L.horizon_clip) a.horizon_clip = 1u;
    }
    return a;
}

// Share (per mille) of this launch's work units, most expensive first, that go through the cooperative
// trace kernel.  Measured at 720x720 gl_iter 1 (trace kernel, ms): 64 800 pixels, an eighth of the frame:
// all cooperative 1.8, none 6.7; 129 600 pixels: all 3.3, 15-60 % 3.8, 6 % 4.1, none 7.1; 259 200 pixels: 6 % 4.8,
// 15 % 4.9, all 6.4, none 7.7; the full frame: none 9.0, 3 % 8.9, all 12.4.  The environment variable
// CRT_COOP_PERMILLE, read once by crt_create, is for such measurements (tools/mix_sweep.sh); it changes speed only.
uint32_t pick_cooperative(const crt_ctx* ctx, const crt::RenderArgs& a) {
    if (a.gl_iter == 0) return 0u;
    if (ctx->trace_mode == CRT_TRACE_COOPERATIVE) return 1000u;
    if (ctx->trace_mode == CRT_TRACE_PER_PIXEL) return 0u;
    if (ctx->coop_permille >= 0) return (uint32_t)ctx->coop_permille > 1000u ? 1000u : (uint32_t)ctx->coop_permille;
    const uint64_t pixels = (uint64_t)a.n_units * 64u;
    return pixels < 200000ull ? 1000u : pixels < 400000ull ? 60u : 0u;
}

// wavefront_reserve within the context's workspace limit; when the device has less memory left than that, with
// half the limit, a quarter, ... (a smaller workspace means more rounds, never another image).
hipError_t reserve_workspace(crt_ctx* ctx, const crt::RenderArgs& a) {
    hipError_t e = hipSuccess;
    for (size_t budget = ctx->workspace_limit;; budget /= 2) {
        e = crt::wavefront_reserve(ctx->wf, a, ctx->scene_host.n_lights, budget);
        if (e != hipErrorOutOfMemory || budget < ((size_t)1 << 30)) break;
        (void)hipGetLastError();
    }
    return e;
}

// Whether a launch is rendered as two interleaved pieces side by side, each a wavefront frame of its own (workspace,
// stream, host thread).  The stages of one frame run one after the other, and each leaves part of the GPU idle at
// its start and end; the trace kernel of a multi-round frame's late rounds waits more than it computes.  Two
// half-frames fill those gaps with each other's kernels: 720x720, gl_iter 1: 17.9 against 18.6 ms; gl_iter 2: 1 070
// against 1 200 ms; an eighth of that frame (a rank of an 8-GPU job, 1 013 units): 3.00 against 3.04 ms and 178
// against 191 ms; a sixteenth: 2.04 against 1.90 ms, hence the threshold.  Three pieces of the whole frame are slower
// than one launch.  The pieces run the trace kernels that the whole launch's size picks (pick_cooperative): halves
// of the whole frame with the per-pixel kernel alone, 1 074 ms at gl_iter 2, with the mix a half-frame would pick
// for itself, 1 158.
bool wants_pieces(const crt_ctx* ctx, const crt::RenderArgs& a) {
    if (ctx->piece_min_units <= 0 || a.n_units < (uint32_t)ctx->piece_min_units) return false;
    return ctx->pipeline != CRT_PIPELINE_PERSISTENT && a.gl_iter != 0u && a.audit == 0u && a.parts <= 1u;
}

crt::RenderArgs piece_args(const crt::RenderArgs& a, uint32_t k) {
    crt::RenderArgs p = a;
    p.parts = crt_ctx::kPieces;
    p.part = k;
    p.n_units = (a.n_units + crt_ctx::kPieces - 1u - k) / crt_ctx::kPieces;
    return p; // `cooperative` stays what the whole launch's size picked
}

void release_pieces(crt_ctx* ctx) {
    for (uint32_t k = 0; k < crt_ctx::kPieces; ++k) crt::wavefront_release(ctx->wf_piece[k]);
}

// A launch whose pieces overflowed their ray queues is rendered whole from then on (until the scene changes): it
// would only overflow again.
bool same_launch(const crt::RenderArgs& a, const crt::RenderArgs& b) {
    return a.n_units == b.n_units && a.nx == b.nx && a.ny == b.ny && a.aa_iter == b.aa_iter && a.ref_iter == b.ref_iter &&
           a.gl_iter == b.gl_iter && a.ind_rays == b.ind_rays && a.mode == b.mode && a.rank == b.rank && a.world == b.world &&
           a.tx == b.tx && a.ty == b.ty && a.win_bx == b.win_bx && a.win_by == b.win_by && a.start_bx == b.start_bx &&
           a.start_by == b.start_by;
}

// Streams, events and workspaces of the pieces; false when one of them cannot be had (the launch then runs whole).
bool reserve_pieces(crt_ctx* ctx, const crt::RenderArgs& a) {
    for (uint32_t k = 0; k < crt_ctx::kPieces; ++k) {
        if (!ctx->piece_stream[k] && hipStreamCreateWithFlags(&ctx->piece_stream[k], hipStreamNonBlocking) != hipSuccess) return false;
        if (!ctx->piece_join[k] && hipEventCreateWithFlags(&ctx->piece_join[k], hipEventDisableTiming) != hipSuccess) return false;
    }
    if (!ctx->piece_fork && hipEventCreateWithFlags(&ctx->piece_fork, hipEventDisableTiming) != hipSuccess) return false;
    if (!ctx->piece_worker) {
        try {
            ctx->piece_worker.reset(new PieceWorker());
        } catch (...) { // no memory, or no thread to be had: the launch runs whole
            return false;
        }
    }
    const size_t npix = (size_t)a.nx * a.ny;
    if (ctx->frame_backup_npix != npix) {
        (void)hipFree(ctx->frame_backup);
        ctx->frame_backup = nullptr;
        ctx->frame_backup_npix = 0;
        if (hipMalloc((void**)&ctx->frame_backup, (8u * npix + 8u) * sizeof(uint32_t)) != hipSuccess) return false;
        ctx->frame_backup_npix = npix;
    }
    for (uint32_t k = 0; k < crt_ctx::kPieces; ++k) {
        const crt::RenderArgs p = piece_args(a, k);
        if (!crt::wavefront_supports(p)) return false;
        hipError_t e = hipSuccess;
        for (size_t budget = ctx->workspace_limit / crt_ctx::kPieces;; budget /= 2) {
            e = crt::wavefront_reserve(ctx->wf_piece[k], p, ctx->scene_host.n_lights, budget);
            if (e != hipErrorOutOfMemory || budget < ((size_t)1 << 30)) break;
            (void)hipGetLastError();
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            release_pieces(ctx); // the launch runs whole and wants the memory
            return false;
        }
    }
    return true;
}

// Copies generator states, work counters and tallies to (save) or from the frame backup, on `stream`.
hipError_t frame_backup_copy(crt_ctx* ctx, const crt::RenderArgs& a, bool save, hipStream_t stream) {
    const size_t npix = (size_t)a.nx * a.ny;
    uint32_t* b = ctx->frame_backup;
    auto copy = [&](void* live, uint32_t* saved, size_t bytes) {
        return save ? hipMemcpyAsync(saved, live, bytes, hipMemcpyDeviceToDevice, stream)
                    : hipMemcpyAsync(live, saved, bytes, hipMemcpyDeviceToDevice, stream);
    };
    hipError_t e = copy(a.rng, b, npix * 6u * sizeof(uint32_t));
    if (e == hipSuccess) e = copy(a.counters, b + 6u * npix, npix * 2u * sizeof(uint32_t));
    if (e == hipSuccess) e = copy(a.wave_tallies, b + 8u * npix, 8u * sizeof(uint32_t));
    return e;
}

struct PieceResult {
    hipError_t err = hipSuccess;
    bool overflow = false, truncated = false, needs_redo = false;
};

// The launch as two pieces side by side.  *done is false when a piece's ray queue overflowed: everything is back
// where the frame started and the caller renders the launch whole (that path saves and redoes by itself).
int run_pieces(crt_ctx* ctx, const crt::RenderArgs& a, hipStream_t stream, bool* done) {
    *done = false;
    HIP_TRY(ctx, frame_backup_copy(ctx, a, true, stream));
    HIP_TRY(ctx, hipEventRecord(ctx->piece_fork, stream));
    PieceResult res[crt_ctx::kPieces];
    auto run = [&](uint32_t k) {
        PieceResult& r = res[k];
        r.err = hipSetDevice(ctx->device);
        if (r.err == hipSuccess) r.err = hipStreamWaitEvent(ctx->piece_stream[k], ctx->piece_fork, 0);
        if (r.err == hipSuccess)
            r.err = crt::launch_wavefront(piece_args(a, k), ctx->wf_piece[k], ctx->piece_stream[k], &r.overflow, &r.truncated, true,
                                          &r.needs_redo);
        const hipError_t e2 = hipEventRecord(ctx->piece_join[k], ctx->piece_stream[k]); // also after a failure: part may be queued
        if (r.err == hipSuccess) r.err = e2;
    };
    ctx->piece_worker->submit([&run] { run(1u); });
    run(0u);
    ctx->piece_worker->wait();
    for (uint32_t k = 0; k < crt_ctx::kPieces; ++k) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->piece_join[k], 0));
    bool overflow = false, redo = false, truncated = false;
    for (const PieceResult& r : res) {
        if (r.err != hipSuccess) return fail(ctx, CRT_ERR_HIP, "wavefront pipeline (piece of a frame): %s", hipGetErrorString(r.err));
        overflow = overflow || r.overflow;
        redo = redo || r.needs_redo;
        truncated = truncated || r.truncated;
    }
    if (overflow) return fail(ctx, CRT_ERR_HIP, "wavefront pipeline: a device queue overflowed (in this frame or one left running before it)");
    if (redo) {
        HIP_TRY(ctx, frame_backup_copy(ctx, a, false, stream));
        return CRT_OK;
    }
    *done = true;
    if (truncated)
        return fail(ctx, CRT_ERR_TRUNCATED, "frame stopped after %u rounds (crt_set_round_limit): image unfinished, RNG states mid-tree",
                    ctx->wf.max_rounds);
    return CRT_OK;
}

// Runs one launch through the pipeline the context selects.
int run_pipeline(crt_ctx* ctx, const crt::RenderArgs& args, hipStream_t stream) {
    crt::RenderArgs a = args;
    a.cooperative = pick_cooperative(ctx, a);
    if (wants_pieces(ctx, a) && !(ctx->piece_overflowed && same_launch(a, ctx->piece_overflow_key)) && reserve_pieces(ctx, a)) {
        bool done = false;
        const int rc = run_pieces(ctx, a, stream, &done);
        if (rc != CRT_OK || done) return rc;
        release_pieces(ctx); // make room: the launch is rendered whole below, which needs the workspace limit once more
        ctx->piece_overflowed = true;
        ctx->piece_overflow_key = a;
    }
    const bool can_wf = crt::wavefront_supports(a);
    if (ctx->pipeline == CRT_PIPELINE_WAVEFRONT && !can_wf)
        return fail(ctx, CRT_ERR_UNSUPPORTED, "the wavefront pipeline does not implement these settings");
    if (ctx->pipeline == CRT_PIPELINE_PERSISTENT || !can_wf) {
        HIP_TRY(ctx, crt::launch_render(a, stream));
        return CRT_OK;
    }
    const hipError_t reserved = reserve_workspace(ctx, a);
    if (reserved != hipSuccess) {
        // no workspace (device memory exhausted): the persistent kernel needs none, so the default pipeline
        // still renders; only an explicit request for the wavefront kernels reports the failure
        (void)hipGetLastError();
        if (ctx->pipeline == CRT_PIPELINE_WAVEFRONT)
            return fail(ctx, CRT_ERR_HIP, "wavefront workspace: %s", hipGetErrorString(reserved));
        HIP_TRY(ctx, crt::launch_render(a, stream));
        return CRT_OK;
    }
    bool overflow = false, truncated = false;
    HIP_TRY(ctx, crt::launch_wavefront(a, ctx->wf, stream, &overflow, &truncated));
    if (overflow) return fail(ctx, CRT_ERR_HIP, "wavefront pipeline: a device queue overflowed (in this frame or one left running before it)");
    if (truncated)
        return fail(ctx, CRT_ERR_TRUNCATED, "frame stopped after %u rounds (crt_set_round_limit): image unfinished, RNG states mid-tree",
                    ctx->wf.max_rounds);
    return CRT_OK;
}

int run_render(crt_ctx* ctx, const crt::RenderArgs& args, hipStream_t stream) {
    int rc = order_behind_last_launch(ctx, stream);
    if (rc != CRT_OK) return rc;
    rc = run_pipeline(ctx, args, stream);
    const int marked = mark_launched(ctx, stream); // also after a failure: part of the launch may be queued
    return rc != CRT_OK ? rc : marked;
}

} // namespace

extern "C" {

int crt_create(crt_ctx** out, int device) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CRT_ERR_HIP;
    crt_ctx* ctx = new (std::nothrow) crt_ctx();
    if (!ctx) return CRT_ERR_INVALID;
    ctx->device = device;
    if (const char* env = getenv("CRT_COOP_PERMILLE")) ctx->coop_permille = atoi(env) < 0 ? 0 : atoi(env);
    if (const char* env = getenv("CRT_FRAME_PIECES_MIN_UNITS")) ctx->piece_min_units = atoi(env); // a measuring and testing aid
    const size_t ltc_bytes = (size_t)(crt_ltc_tables_blob_end - crt_ltc_tables_blob);
    if (ltc_bytes != 2u * 64u * 64u * 4u * sizeof(float) || hipSetDevice(device) != hipSuccess ||
        hipMalloc((void**)&ctx->ltc_dev, ltc_bytes) != hipSuccess ||
        hipMemcpy(ctx->ltc_dev, crt_ltc_tables_blob, ltc_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void**)&ctx->scene_dev, sizeof(crt::DevScene)) != hipSuccess ||
        hipMalloc((void**)&ctx->unit_counter, 256) != hipSuccess) {
        crt_destroy(ctx);
        return CRT_ERR_HIP;
    }
    // this device's copy of the static scratch array (the device is current), and the persistent kernel's
    // grid size: neither is looked up again by crt_render_static
    if (device < 64 && g_scratch_ptr[device] == nullptr) {
        uint32_t* ptr = nullptr;
        if (hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || ptr == nullptr) {
            crt_destroy(ctx);
            return CRT_ERR_HIP;
        }
        g_scratch_ptr[device] = ptr;
    }
    if (device < 64 && g_static_done[device] == nullptr &&
        hipEventCreateWithFlags(&g_static_done[device], hipEventDisableTiming) != hipSuccess) {
        g_static_done[device] = nullptr;
        crt_destroy(ctx);
        return CRT_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&ctx->last_done, hipEventDisableTiming) != hipSuccess) {
        ctx->last_done = nullptr;
        crt_destroy(ctx);
        return CRT_ERR_HIP;
    }
    if (crt::prime_render_grid() != hipSuccess) {
        crt_destroy(ctx);
        return CRT_ERR_HIP;
    }
    ctx->scratch_ptr = device < 64 ? g_scratch_ptr[device] : nullptr; // null: crt_render_static reports it
    *out = ctx;
    return CRT_OK;
}

void crt_destroy(crt_ctx* ctx) {
    if (!ctx) return;
    if (hipSetDevice(ctx->device) == hipSuccess) {
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->ltc_dev);
        (void)hipFree(ctx->scene_dev);
        (void)hipFree(ctx->rng_dev);
        (void)hipFree(ctx->rng_init_d);
        (void)hipFree(ctx->counters_dev);
        (void)hipFree(ctx->unit_counter);
        if (ctx->last_done) (void)hipEventDestroy(ctx->last_done);
        crt::wavefront_release(ctx->wf);
        release_pieces(ctx);
        for (uint32_t k = 0; k < crt_ctx::kPieces; ++k) {
            if (ctx->piece_stream[k]) (void)hipStreamDestroy(ctx->piece_stream[k]);
            if (ctx->piece_join[k]) (void)hipEventDestroy(ctx->piece_join[k]);
        }
        if (ctx->piece_fork) (void)hipEventDestroy(ctx->piece_fork);
        (void)hipFree(ctx->frame_backup);
    }
    delete ctx;
}

const char* crt_last_error(const crt_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int crt_scene_upload(crt_ctx* ctx, const crt_scene_desc* scene) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!scene) return fail(ctx, CRT_ERR_INVALID, "scene is NULL");
    // Flattened into a scratch copy first: a scene that is rejected half way must leave the context's host
    // mirror (which sizes the pipeline's queues and jump tables) and the device copy describing the same,
    // previous scene.
    auto flat = std::make_unique<crt::DevScene>();
    int rc = crt::flatten_scene(scene, flat.get(), ctx->err, sizeof ctx->err);
    if (rc != CRT_OK) return rc;
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize()); // no render may still be reading the old scene
    ctx->has_scene = false;               // until the device holds the new one
    ctx->wf.order_valid = false;          // the trace kernel's launch order was estimated on the old scene
    for (crt::WfWorkspace& w : ctx->wf_piece) w.order_valid = false;
    ctx->piece_overflowed = false;
    HIP_TRY(ctx, hipMemcpy(ctx->scene_dev, flat.get(), sizeof(crt::DevScene), hipMemcpyHostToDevice));
    ctx->scene_host = *flat;
    ctx->has_scene = true;
    return CRT_OK;
}

int crt_rng_init(crt_ctx* ctx, uint32_t nx, uint32_t ny, uint64_t seed, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (nx == 0 || ny == 0 || (uint64_t)nx * ny > 0x7fffffffull) return fail(ctx, CRT_ERR_INVALID, "bad image size");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    const size_t npix = (size_t)nx * ny;
    if (ctx->rng_nx != nx || ctx->rng_ny != ny || !ctx->rng_dev) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        (void)hipFree(ctx->rng_dev);
        (void)hipFree(ctx->rng_init_d);
        (void)hipFree(ctx->counters_dev);
        ctx->rng_dev = ctx->rng_init_d = ctx->counters_dev = nullptr;
        ctx->rng_nx = ctx->rng_ny = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rng_dev, npix * 6 * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rng_init_d, npix * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->counters_dev, npix * 2 * sizeof(uint32_t)));
        ctx->rng_nx = nx;
        ctx->rng_ny = ny;
    }
    hipStream_t s = (hipStream_t)stream;
    if ((rc = order_behind_last_launch(ctx, s)) != CRT_OK) return rc; // a frame still running elsewhere draws from these states
    crt::launch_random_init((uint32_t)npix, seed, ctx->rng_dev, s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->rng_init_d, ctx->rng_dev + 5 * npix, npix * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters_dev, 0, npix * 2 * sizeof(uint32_t), s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->unit_counter + 16, 0, 32, s));
    return mark_launched(ctx, s);
}

int crt_rng_export(crt_ctx* ctx, uint32_t* host_state6) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!host_state6) return fail(ctx, CRT_ERR_INVALID, "host_state6 is NULL");
    if (!ctx->rng_dev) return fail(ctx, CRT_ERR_NO_RNG, "no RNG states");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    const size_t npix = (size_t)ctx->rng_nx * ctx->rng_ny;
    std::vector<uint32_t> planes(npix * 6);
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (collect_async_all(ctx))
        return fail(ctx, CRT_ERR_HIP, "wavefront pipeline: a device queue overflowed in an earlier frame");
    HIP_TRY(ctx, hipMemcpy(planes.data(), ctx->rng_dev, npix * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t p = 0; p < npix; ++p)
        for (int k = 0; k < 6; ++k) host_state6[p * 6 + k] = planes[(size_t)k * npix + p];
    return CRT_OK;
}

int crt_render(crt_ctx* ctx, const crt_settings* st, float* fb_dev, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!fb_dev) return fail(ctx, CRT_ERR_INVALID, "fb_dev is NULL");
    int rc = check_settings(ctx, st);
    if (rc != CRT_OK) return rc;
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    crt::RenderArgs a = base_args(ctx, st, fb_dev);
    a.mode = crt::MODE_FULL;
    a.n_units = a.tiles_x * a.tiles_y;
    return run_render(ctx, a, (hipStream_t)stream);
}

int crt_render_static(crt_ctx* ctx, const crt_settings* st, float* fb_dev, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!fb_dev) return fail(ctx, CRT_ERR_INVALID, "fb_dev is NULL");
    int rc = check_settings(ctx, st);
    if (rc != CRT_OK) return rc;
    if ((uint64_t)st->nx * st->ny > CRT_STATIC_MAX_PIXELS)
        return fail(ctx, CRT_ERR_UNSUPPORTED, "crt_render_static: %ux%u pixels exceed CRT_STATIC_MAX_PIXELS (%u)", st->nx,
                    st->ny, (unsigned)CRT_STATIC_MAX_PIXELS);
    if (!ctx->scratch_ptr || ctx->device >= 64 || !g_static_done[ctx->device])
        return fail(ctx, CRT_ERR_HIP, "crt_render_static: no scratch address for device %d", ctx->device);
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    crt::RenderArgs a = base_args(ctx, st, fb_dev);
    a.mode = crt::MODE_FULL;
    a.n_units = a.tiles_x * a.tiles_y;
    // the kernel's scratch comes from the static array: cursor at word 0, wave tallies at word 16, then the
    // per-pixel counters, which start at zero for this call (the cursor is cleared by launch_render)
    uint32_t* sp = ctx->scratch_ptr;
    const size_t npix = (size_t)st->nx * st->ny;
    a.unit_counter = sp;
    a.wave_tallies = reinterpret_cast<unsigned long long*>(sp + 16);
    a.counters = sp + CRT_STATIC_HEAD_WORDS;
    a.audit = 0u;
    // the previous call on this GPU may still be using the array on another stream: queue behind it
    const int dv = ctx->device;
    if ((rc = order_behind_last_launch(ctx, (hipStream_t)stream)) != CRT_OK) return rc;
    if (g_static_used[dv] && g_static_stream[dv] != stream) HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)stream, g_static_done[dv], 0));
    HIP_TRY(ctx, hipMemsetAsync(sp, 0, (CRT_STATIC_HEAD_WORDS + 2u * npix) * sizeof(uint32_t), (hipStream_t)stream));
    HIP_TRY(ctx, crt::launch_render(a, (hipStream_t)stream));
    HIP_TRY(ctx, hipEventRecord(g_static_done[dv], (hipStream_t)stream));
    g_static_stream[dv] = stream;
    g_static_used[dv] = true;
    return mark_launched(ctx, (hipStream_t)stream);
}

int crt_render_partial(crt_ctx* ctx, const crt_settings* st, uint32_t tileIndex, float* fb_dev, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!fb_dev) return fail(ctx, CRT_ERR_INVALID, "fb_dev is NULL");
    int rc = check_settings(ctx, st);
    if (rc != CRT_OK) return rc;
    if (st->tx == 0 || st->ty == 0 || st->blocksPerDraw == 0)
        return fail(ctx, CRT_ERR_INVALID, "tx, ty and blocksPerDraw must be positive");
    if ((uint64_t)st->tx * st->ty > 1024u) return fail(ctx, CRT_ERR_INVALID, "tx*ty exceeds 1024 threads per block");
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    // window arithmetic of the progressive loop (main.cu:719-727,752-754)
    const uint32_t blocksX = st->nx / st->tx + 1, blocksY = st->ny / st->ty + 1;
    const uint32_t bx = st->blocksPerDraw < blocksX ? st->blocksPerDraw : blocksX;
    uint32_t by = st->blocksPerDraw / blocksX + 1;
    if (by < 1) by = 1;
    if (by > blocksY) by = blocksY;
    crt::RenderArgs a = base_args(ctx, st, fb_dev);
    a.mode = crt::MODE_PARTIAL;
    a.tx = st->tx;
    a.ty = st->ty;
    a.blocks_x = blocksX;
    a.blocks_y = blocksY;
    a.win_bx = bx;
    a.win_by = by;
    a.start_bx = tileIndex % blocksX;
    a.start_by = (tileIndex / blocksX) < blocksY ? (tileIndex / blocksX) : blocksY;
    const uint64_t lanes = (uint64_t)bx * by * st->tx * st->ty;
    a.n_units = (uint32_t)((lanes + 63u) / 64u);
    return run_render(ctx, a, (hipStream_t)stream);
}

uint32_t crt_shard_tiles(uint32_t nx, uint32_t ny, uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) return 0;
    const uint32_t tiles = ((nx + 7u) / 8u) * ((ny + 7u) / 8u);
    return tiles / world + (rank < tiles % world ? 1u : 0u);
}

uint32_t crt_shard_tiles_max(uint32_t nx, uint32_t ny, uint32_t world) {
    if (world == 0) return 0;
    const uint32_t tiles = ((nx + 7u) / 8u) * ((ny + 7u) / 8u);
    return (tiles + world - 1u) / world;
}

int crt_render_shard(crt_ctx* ctx, const crt_settings* st, uint32_t rank, uint32_t world, float* compact_dev,
                     void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!compact_dev) return fail(ctx, CRT_ERR_INVALID, "compact_dev is NULL");
    if (world == 0 || rank >= world) return fail(ctx, CRT_ERR_INVALID, "rank %u out of world %u", rank, world);
    int rc = check_settings(ctx, st);
    if (rc != CRT_OK) return rc;
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    crt::RenderArgs a = base_args(ctx, st, compact_dev);
    a.mode = crt::MODE_SHARD;
    a.rank = rank;
    a.world = world;
    // padding (pixels outside the image, tiles past this rank's last one) stays zero
    const uint32_t tiles_max = crt_shard_tiles_max(st->nx, st->ny, world);
    HIP_TRY(ctx, hipMemsetAsync(compact_dev, 0, (size_t)tiles_max * 256u * sizeof(float), (hipStream_t)stream));
    a.n_units = crt_shard_tiles(st->nx, st->ny, rank, world);
    return run_render(ctx, a, (hipStream_t)stream);
}

int crt_scatter_shards(crt_ctx* ctx, uint32_t nx, uint32_t ny, uint32_t world, const float* gathered_dev,
                       float* fb_dev, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!gathered_dev || !fb_dev || world == 0 || nx == 0 || ny == 0)
        return fail(ctx, CRT_ERR_INVALID, "bad argument to crt_scatter_shards");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    crt::launch_scatter(nx, ny, (nx + 7u) / 8u, (ny + 7u) / 8u, world, crt_shard_tiles_max(nx, ny, world),
                        gathered_dev, fb_dev, (hipStream_t)stream);
    HIP_TRY(ctx, hipGetLastError());
    return CRT_OK;
}

int crt_stats_since_init(crt_ctx* ctx, const crt_settings* st, crt_stats* out) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!out || !st) return fail(ctx, CRT_ERR_INVALID, "NULL argument");
    if (!ctx->rng_dev) return fail(ctx, CRT_ERR_NO_RNG, "no RNG states");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    const size_t npix = (size_t)ctx->rng_nx * ctx->rng_ny;
    std::vector<uint32_t> d_now(npix), d_init(npix), cnt(npix * 2);
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (collect_async_all(ctx))
        return fail(ctx, CRT_ERR_HIP, "wavefront pipeline: a device queue overflowed in an earlier frame");
    HIP_TRY(ctx, hipMemcpy(d_now.data(), ctx->rng_dev + 5 * npix, npix * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(d_init.data(), ctx->rng_init_d, npix * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(cnt.data(), ctx->counters_dev, npix * 8, hipMemcpyDeviceToHost));
    uint64_t draws = 0, shade = 0, expand = 0;
    for (size_t p = 0; p < npix; ++p) {
        draws += (uint32_t)((d_now[p] - d_init[p]) * 945708813u); // 362437^-1 mod 2^32
        shade += cnt[p];
        expand += cnt[npix + p];
    }
    out->rng_draws = draws;
    out->shade_points = shade;
    out->gi_children = expand * (uint64_t)st->ind_rays;
    out->shadow_samples = (draws - 2u * out->gi_children) / 2u;
    unsigned long long tallies[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(tallies, ctx->unit_counter + 16, 16, hipMemcpyDeviceToHost));
    out->exact_tests = tallies[0];
    out->glass_walks = tallies[1];
    return CRT_OK;
}

int crt_set_pipeline(crt_ctx* ctx, int pipeline) {
    if (!ctx) return CRT_ERR_INVALID;
    if (pipeline != CRT_PIPELINE_AUTO && pipeline != CRT_PIPELINE_PERSISTENT && pipeline != CRT_PIPELINE_WAVEFRONT)
        return fail(ctx, CRT_ERR_INVALID, "unknown pipeline %d", pipeline);
    ctx->pipeline = pipeline;
    return CRT_OK;
}

int crt_set_timing(crt_ctx* ctx, int enabled) {
    if (!ctx) return CRT_ERR_INVALID;
    crt::WfWorkspace* all[] = {&ctx->wf, &ctx->wf_piece[0], &ctx->wf_piece[1]};
    for (crt::WfWorkspace* w : all) {
        w->timing = enabled != 0;
        for (uint32_t k = 0; k < crt::WF_KERNEL_COUNT; ++k) {
            w->kernel_ms[k] = 0.0;
            w->kernel_launches[k] = 0;
        }
    }
    return CRT_OK;
}

int crt_get_kernel_times(crt_ctx* ctx, double ms[5], uint64_t launches[5]) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!ms || !launches) return fail(ctx, CRT_ERR_INVALID, "NULL argument");
    // a launch rendered in two pieces has run every kernel twice, side by side: the times add up to more than the frame's
    for (uint32_t k = 0; k < 5u; ++k) {
        ms[k] = ctx->wf.kernel_ms[k] + ctx->wf_piece[0].kernel_ms[k] + ctx->wf_piece[1].kernel_ms[k];
        launches[k] = ctx->wf.kernel_launches[k] + ctx->wf_piece[0].kernel_launches[k] + ctx->wf_piece[1].kernel_launches[k];
    }
    return CRT_OK;
}

int crt_set_trace_mode(crt_ctx* ctx, int mode) {
    if (!ctx) return CRT_ERR_INVALID;
    if (mode != CRT_TRACE_AUTO && mode != CRT_TRACE_PER_PIXEL && mode != CRT_TRACE_COOPERATIVE)
        return fail(ctx, CRT_ERR_INVALID, "unknown trace mode %d", mode);
    ctx->trace_mode = mode;
    return CRT_OK;
}

int crt_set_audit(crt_ctx* ctx, int enabled) {
    if (!ctx) return CRT_ERR_INVALID;
    ctx->audit = enabled ? 1 : 0;
    return CRT_OK;
}

int crt_audit_result(crt_ctx* ctx, uint64_t* checked, uint64_t* contradictions) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!checked || !contradictions) return fail(ctx, CRT_ERR_INVALID, "NULL argument");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    unsigned long long t[2] = {0, 0};
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(t, ctx->unit_counter + 20, 16, hipMemcpyDeviceToHost));
    *checked = t[0];
    *contradictions = t[1];
    return CRT_OK;
}

int crt_reserve_workspace(crt_ctx* ctx, const crt_settings* st, uint32_t rank, uint32_t world) {
    if (!ctx) return CRT_ERR_INVALID;
    if (world == 0 || rank >= world) return fail(ctx, CRT_ERR_INVALID, "rank %u out of world %u", rank, world);
    int rc = check_settings(ctx, st);
    if (rc != CRT_OK) return rc;
    if ((rc = bind(ctx)) != CRT_OK) return rc;
    if (ctx->pipeline == CRT_PIPELINE_PERSISTENT) return CRT_OK;
    crt::RenderArgs a = base_args(ctx, st, nullptr);
    a.mode = world == 1 ? crt::MODE_FULL : crt::MODE_SHARD;
    a.rank = rank;
    a.world = world;
    a.n_units = world == 1 ? a.tiles_x * a.tiles_y : crt_shard_tiles(st->nx, st->ny, rank, world);
    a.cooperative = pick_cooperative(ctx, a); // before the check: the cooperative kernel needs more record slack
    if (wants_pieces(ctx, a) && !(ctx->piece_overflowed && same_launch(a, ctx->piece_overflow_key)) && reserve_pieces(ctx, a))
        return CRT_OK; // crt_render takes the launch in two pieces
    if (!crt::wavefront_supports(a)) return CRT_OK; // crt_render takes the persistent kernel (or refuses, if told to use the pipeline)
    const hipError_t reserved = reserve_workspace(ctx, a);
    if (reserved != hipSuccess) {
        (void)hipGetLastError();
        // as in crt_render: without a workspace the default configuration renders with the persistent kernel
        if (ctx->pipeline == CRT_PIPELINE_WAVEFRONT)
            return fail(ctx, CRT_ERR_HIP, "wavefront workspace: %s", hipGetErrorString(reserved));
    }
    return CRT_OK;
}

int crt_set_round_limit(crt_ctx* ctx, uint32_t rounds) {
    if (!ctx) return CRT_ERR_INVALID;
    ctx->wf.max_rounds = rounds;
    for (crt::WfWorkspace& w : ctx->wf_piece) w.max_rounds = rounds; // each piece of a launch stops after that many rounds of its own
    return CRT_OK;
}

int crt_set_workspace_limit(crt_ctx* ctx, size_t bytes) {
    if (!ctx) return CRT_ERR_INVALID;
    if (bytes < ((size_t)1 << 20)) return fail(ctx, CRT_ERR_INVALID, "workspace limit below 1 MiB");
    ctx->workspace_limit = bytes;
    return CRT_OK;
}

int crt_selftest_rcp(crt_ctx* ctx, uint64_t* mismatches, uint32_t* first_bad_bits) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!mismatches || !first_bad_bits) return fail(ctx, CRT_ERR_INVALID, "NULL argument");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    unsigned long long* d_bad = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_bad, 16));
    uint32_t* d_first = reinterpret_cast<uint32_t*>(d_bad + 1);
    hipError_t e = hipMemset(d_bad, 0, 8);
    if (e == hipSuccess) e = hipMemset(d_first, 0xff, 4);
    if (e == hipSuccess) {
        crt::launch_selftest_rcp(d_bad, d_first, nullptr);
        e = hipGetLastError();
    }
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(first_bad_bits, d_first, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(ctx, CRT_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    *mismatches = bad;
    return CRT_OK;
}

int crt_fb_alloc(crt_ctx* ctx, size_t n_floats, float** out_dev) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!out_dev || n_floats == 0) return fail(ctx, CRT_ERR_INVALID, "bad argument to crt_fb_alloc");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    HIP_TRY(ctx, hipMalloc((void**)out_dev, n_floats * sizeof(float)));
    HIP_TRY(ctx, hipMemset(*out_dev, 0, n_floats * sizeof(float))); // pixels a progressive walk never reaches read as 0
    return CRT_OK;
}

int crt_fb_free(crt_ctx* ctx, float* dev) {
    if (!ctx) return CRT_ERR_INVALID;
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    HIP_TRY(ctx, hipFree(dev));
    return CRT_OK;
}

int crt_fb_download(crt_ctx* ctx, const float* dev, float* host, size_t n_floats, void* stream) {
    if (!ctx) return CRT_ERR_INVALID;
    if (!dev || !host) return fail(ctx, CRT_ERR_INVALID, "NULL buffer");
    int rc = bind(ctx);
    if (rc != CRT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(host, dev, n_floats * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CRT_OK;
}

} // extern "C"
