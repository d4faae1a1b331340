// Launch interface between the C ABI (crt_api.cpp) and the kernels (crt_render.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "crt_scene.hpp"

namespace crt {

enum RenderMode : uint32_t {
    MODE_FULL = 0,    // every pixel, wave w = 8x8 tile w (render, main.cu:328-347)
    MODE_SHARD = 1,   // tiles rank, rank+world, ... into a compact buffer
    MODE_PARTIAL = 2, // a wrapped window of tx*ty blocks (render_partial, main.cu:349-368)
};

struct RenderArgs {
    const DevScene* scene;
    const float* ltc; // ltc_1 then ltc_2, 64*64*4 floats each
    uint32_t* rng;    // 6 planes of nx*ny words: v0..v4, d
    uint32_t* counters; // 2 planes of nx*ny words: shaded diffuse hits, expanding hits (accumulated)
    float* out;
    unsigned long long* wave_tallies; // [0] rays sent to the exact test, [1] rays sent to the glass walk, [2] rays the filter audit checked, [3] audit contradictions (accumulated)
    uint32_t* unit_counter; // global work-unit cursor (zeroed by launch_render)
    uint32_t n_units;       // 64-pixel work units in this launch
    uint32_t nx, ny;
    uint32_t aa_iter, ref_iter, gl_iter, ind_rays;
    uint32_t mode;
    uint32_t tiles_x, tiles_y; // 8x8 tiles
    uint32_t rank, world;      // MODE_SHARD
    uint32_t tx, ty, blocks_x, blocks_y, win_bx, win_by, start_bx, start_by; // MODE_PARTIAL
    uint32_t audit;           // wavefront pipeline: also run the filter audit kernel
    uint32_t max_shadow_rays; // shadow rays one diffuse hit casts: sum of shadowSamples over initialised area lights
    uint32_t n_geom;          // primitives in the scene (sizes the trace kernel's LDS table)
    uint32_t horizon_clip;    // some light takes the horizon-clipping LTC branch (crt_light::horizonClip)
    uint32_t cooperative;     // wavefront pipeline: per mille of the work units (most expensive first) whose last GI
                              // level the trace kernel expands with one lane per child ray; 0 = none, 1000 = all
    uint32_t part, parts;     // parts > 1: this launch is one of `parts` interleaved pieces of a larger one; its unit u
                              // is unit u * parts + part of that launch (n_units counts this piece's units)
};

// ---- wavefront pipeline (crt_wavefront.hip)
enum WfKernel : uint32_t { WF_K_TRACE = 0, WF_K_FILTER, WF_K_RESOLVE, WF_K_SHADE, WF_K_SUM, WF_KERNEL_COUNT };
// Device workspace of the multi-kernel pipeline; all planes are structure-of-arrays.
struct WfWorkspace {
    uint32_t n_lp = 0;      // pixel slots (work units * 64) the planes are sized for
    uint32_t quota = 0;     // result slots a pixel may fill per round
    bool single_round = false; // every pixel finishes within one round (quota covers its whole tree, aa_iter <= 1)
    uint32_t gl_levels = 0; // frame levels allocated
    uint32_t cap_work = 0;  // shade records per round
    uint32_t n_lights = 0;  // shadow-count planes
    uint32_t qcap = 0;      // undecided shadow rays the queue holds
    uint32_t max_shadow_rays = 0; // shadow rays per record the queue was sized for
    uint32_t npix = 0;      // pixels of the image the generator backup is sized for
    uint32_t* ctrs = nullptr;       // device counters (WF_CTR_*)
    uint32_t* ctrs_host = nullptr;  // pinned mirror: WF_CTR_RING copies, one per round in flight
    uint32_t* state = nullptr;      // [WF_STATE_PLANES][n_lp] traversal state between rounds
    uint32_t* frames = nullptr;     // [gl_levels][WF_FRAME_PLANES / 4][n_lp] groups of four words
    uint8_t* tags = nullptr;        // [quota][n_lp]
    float* contrib = nullptr;       // [quota][n_lp] float4
    uint32_t* slot_count = nullptr; // [n_lp]
    float* px_acc = nullptr;        // [3][n_lp] float4: colour of the running color() call, pixel accumulator, first corner of the running AA task
    uint32_t* aa_verdict = nullptr; // [n_lp]: the sum kernel's verdict on a finished level >= 2 AA task
    uint32_t* work = nullptr;       // shade records: cap_work x WF_WORK_PLANES words, in six arrays of 16-byte groups
    uint32_t* shadow_cnt = nullptr; // [n_lights][cap_work]
    uint32_t* queue = nullptr;      // [3][qcap]
    uint32_t* jump_dev = nullptr;   // generator jump table: the draws of one shaded point
    uint32_t jump_n[1] = {0};       // step count the table was built for
    uint32_t* unit_cost = nullptr;  // [n_units] cost estimate per work unit (float bits)
    uint32_t* unit_order = nullptr; // [n_units] launch order of the trace kernel
    uint32_t* unit_host = nullptr;  // pinned: [n_units] costs, then [n_units] order
    bool order_valid = false;       // unit_order holds the order for order_key (same scene upload)
    RenderArgs order_key{};         // launch whose order unit_order holds: the estimate depends on scene, camera and pixel mapping only
    uint32_t* backup = nullptr;     // [8][npix] + 8 words: generator states, work counters and wave tallies as they were
                                    // when the frame started (a frame whose ray queue overflows is rendered again)
    void* round_events = nullptr;   // hipEvent_t[WF_CTR_RING]: the counter copy of round r has landed
    // cooperative trace only
    float* draws = nullptr;         // [2 * CRT_RAY_STACK][n_lp]: the (r1, r2) pairs of the innermost node's children
    uint32_t* coop_tabs = nullptr;  // 63 jump tables: 1..63 shaded points' worth of draws
    uint32_t* draw_tabs = nullptr;  // per-pixel trace: draw_tabs_n jump tables, table a-1 = the 2a draws of a child rays (a jump by any
    uint32_t draw_tabs_n = 0;       // number of steps is the same map whatever it jumps over, so they do not depend on the settings)
    uint32_t coop_n = 0xffffffffu;  // draws per shaded point the tables were built for
    void* side_stream = nullptr;    // hipStream_t: the cooperative kernel of a mixed launch runs here
    void* fork_event = nullptr;     // hipEvent_t pair ordering the side stream against the caller's
    void* join_event = nullptr;
    size_t bytes = 0;
    uint64_t frames_redone = 0;     // frames rendered a second time because the ray queue overflowed
    // frames that were left running on their stream (one round, nothing that can overflow): their counter copies land
    // in the ring slots of `async_pending` (bit k = slot k); wavefront_collect_async reads the flags once they have
    uint32_t async_pending = 0, async_next = 0;
    bool async_error = false;       // one of those frames raised a device flag (reported by the next call that looks)
    void* last_stream = nullptr;    // stream of the last launch: a launch on another one is ordered behind it
    // measuring aid (crt_set_round_limit): stop a frame after this many rounds; 0 = no limit
    uint32_t max_rounds = 0;
    // optional per-kernel timing with HIP events on the launch stream (crt_set_timing)
    bool timing = false;
    void* timer = nullptr;   // event pool, owned by crt_wavefront.hip
    double kernel_ms[WF_KERNEL_COUNT] = {0}; // accumulated since the last reset
    uint64_t kernel_launches[WF_KERNEL_COUNT] = {0};
};
constexpr uint32_t WF_STATE_PLANES = 49, WF_FRAME_PLANES = 20, WF_WORK_PLANES = 24; // words per shade record: six 16-byte groups (crt_wavefront.hip)
constexpr uint32_t WF_CTR_RING = 4;
enum WfCounter : uint32_t {
    WF_CTR_WORK = 0,    // record indices handed out this round
    WF_CTR_ACTIVE = 1,  // pixels not finished after this round
    WF_CTR_QUEUE = 2,   // undecided shadow rays queued by the filter kernel (cleared per round, or per filter batch)
    WF_CTR_ERROR = 14,  // non-zero: a capacity was exceeded (results invalid); sticky for the frame
    WF_CTR_QOVERFLOW = 15, // non-zero: the ray queue was full and rays were dropped; sticky for the frame
    WF_CTR_COUNT = 16,
};
// Behind the counters: the cursors the filter waves (one set per light) and the resolve waves (the last set) take
// their groups of 64 records / rays from.  A set is WF_CURSOR_PARTS cursors, each in a 128-byte line of its own and
// each over every WF_CURSOR_PARTS-th group: atomics on one address take about 10 ns each, whoever issues them.
constexpr uint32_t WF_CURSOR_PARTS = 32, WF_CURSOR_STRIDE = 32, WF_CURSOR_SETS = CRT_MAX_LIGHTS + 1;
constexpr uint32_t WF_CURSOR_WORDS = WF_CURSOR_SETS * WF_CURSOR_PARTS * WF_CURSOR_STRIDE;
// True when the wavefront pipeline implements these settings (all of them).
bool wavefront_supports(const RenderArgs& args);
// Frees / (re)allocates the workspace for this launch; returns hipSuccess or the allocation error.
hipError_t wavefront_reserve(WfWorkspace& ws, const RenderArgs& args, uint32_t n_lights, size_t budget_bytes);
void wavefront_release(WfWorkspace& ws);
// Runs the pipeline on `stream`.  Rounds are queued without waiting for each other; the call waits for the
// frame's end to read the device's error flags, except for launches that need one round and cannot
// overflow the ray queue (gl_iter 0 at aa_iter <= 1), which are left running on `stream`.  A frame whose
// ray queue overflowed is rendered again from the saved generator states with launches sized for the worst
// case.  *overflow is set when a capacity check on the device failed for good.
// *truncated is set when the round limit (WfWorkspace::max_rounds) ended the frame with pixels unfinished.
// no_redo: the caller takes care of a frame whose ray queue overflowed (it renders pieces of a frame side by side and
// has to start all of them again): nothing is saved beforehand, and *needs_redo is set instead of rendering again.
hipError_t launch_wavefront(const RenderArgs& args, WfWorkspace& ws, hipStream_t stream, bool* overflow, bool* truncated,
                            bool no_redo = false, bool* needs_redo = nullptr);
// Looks at the counter copies of frames that were left running: with `wait` the call waits for them, without
// it takes those that have landed.  Returns true (once) if one of them had raised a device flag.
bool wavefront_collect_async(WfWorkspace& ws, bool wait);
// Orders `stream` behind the frames that were left running on another stream: they share the context's generator
// states, work counters and workspace with whatever the caller queues next.
hipError_t wavefront_order_behind_async(WfWorkspace& ws, hipStream_t stream);

void launch_random_init(uint32_t npix, uint64_t seed, uint32_t* state, hipStream_t stream);
hipError_t launch_render(const RenderArgs& args, hipStream_t stream);
// Looks up the persistent kernel's grid size for the current device (cached); launch_render does it on first
// use otherwise.
hipError_t prime_render_grid();
void launch_selftest_rcp(unsigned long long* mismatches, uint32_t* first_bad, hipStream_t stream);
void launch_scatter(uint32_t nx, uint32_t ny, uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_max,
                    const float* gathered, float* fb, hipStream_t stream);

} // namespace crt
