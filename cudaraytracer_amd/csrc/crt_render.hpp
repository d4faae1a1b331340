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
    unsigned long long* wave_tallies; // [0] rays sent to stage 2, [1] rays sent to the glass walk (accumulated)
    uint32_t* unit_counter; // global work-unit cursor (zeroed by launch_render)
    uint32_t n_units;       // 64-pixel work units in this launch
    uint32_t nx, ny;
    uint32_t aa_iter, ref_iter, gl_iter, ind_rays;
    uint32_t mode;
    uint32_t tiles_x, tiles_y; // 8x8 tiles
    uint32_t rank, world;      // MODE_SHARD
    uint32_t tx, ty, blocks_x, blocks_y, win_bx, win_by, start_bx, start_by; // MODE_PARTIAL
};

void launch_random_init(uint32_t npix, uint64_t seed, uint32_t* state, hipStream_t stream);
hipError_t launch_render(const RenderArgs& args, hipStream_t stream);
void launch_selftest_rcp(unsigned long long* mismatches, uint32_t* first_bad, hipStream_t stream);
void launch_scatter(uint32_t nx, uint32_t ny, uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_max,
                    const float* gathered, float* fb, hipStream_t stream);

} // namespace crt
