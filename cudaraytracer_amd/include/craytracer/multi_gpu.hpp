// One process driving several GPUs of a node: the image is split into interleaved 8x8 tiles
// (tile t belongs to rank t % N, see crt_render_shard), every GPU renders its tiles, one RCCL gather over
// xGMI brings the compact shard buffers to the first GPU, which de-interleaves them into the row-major
// frame.  The reference renders on a single device (main.cu:914-940); this is the host side of the
// multi-GPU split for C++ callers -- Python callers get the same from cudaraytracer_amd/distributed.py with
// one process per GPU.
//
// Pixels are keyed by their global index in the generator seeding (main.cu:323-324), so any partition
// renders identical pixels: the frame is bit-identical to Renderer::render on one GPU.
//
// Needs the HIP runtime and RCCL headers (compile with hipcc or -I/opt/rocm/include -D__HIP_PLATFORM_AMD__,
// link -lcrt_hip -lrccl -lamdhip64).
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "craytracer.hpp"

namespace craytracer {

class MultiGpuRenderer {
    std::vector<int> _devices;
    std::vector<crt_ctx*> _ctx;
    std::vector<ncclComm_t> _comms;
    std::vector<hipStream_t> _streams;
    // device buffers of render(), kept from call to call and re-made only when the image size changes: every
    // rank's compact shard, and on the first GPU the gathered shards and the row-major frame
    std::vector<float*> _compact;
    float *_gathered = nullptr, *_fb = nullptr;
    unsigned int _buf_nx = 0, _buf_ny = 0;

    static void check(int rc, crt_ctx* ctx, const char* what) {
        if (rc != CRT_OK)
            throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " +
                                     (ctx ? crt_last_error(ctx) : ""));
    }
    static void hip(hipError_t e, const char* what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
    static void nccl(ncclResult_t r, const char* what) {
        if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
    }
    // Runs fn(rank) for every rank on its own host thread (a crt_ctx belongs to one thread at a time, and the
    // render call of the multi-kernel pipeline waits for its GPU) and rethrows the first failure.
    template <class F>
    void for_each_rank(F fn) {
        std::vector<std::thread> threads;
        std::vector<std::string> errors(_ctx.size());
        for (size_t r = 0; r < _ctx.size(); ++r)
            threads.emplace_back([&, r] {
                try {
                    fn(r);
                } catch (const std::exception& e) {
                    errors[r] = e.what()[0] ? e.what() : "error";
                }
            });
        for (auto& t : threads) t.join();
        for (size_t r = 0; r < errors.size(); ++r)
            if (!errors[r].empty()) throw std::runtime_error("GPU " + std::to_string(_devices[r]) + ": " + errors[r]);
    }

public:
    explicit MultiGpuRenderer(std::vector<int> devices) : _devices(std::move(devices)) {
        if (_devices.empty()) throw std::invalid_argument("no devices");
        _ctx.assign(_devices.size(), nullptr);
        _streams.assign(_devices.size(), nullptr);
        try {
            for (size_t r = 0; r < _devices.size(); ++r) {
                check(crt_create(&_ctx[r], _devices[r]), nullptr, "crt_create");
                hip(hipSetDevice(_devices[r]), "hipSetDevice");
                hip(hipStreamCreate(&_streams[r]), "hipStreamCreate");
            }
            _comms.assign(_devices.size(), nullptr);
            nccl(ncclCommInitAll(_comms.data(), static_cast<int>(_devices.size()), _devices.data()), "ncclCommInitAll");
        } catch (...) {
            release();
            throw;
        }
    }
    ~MultiGpuRenderer() { release(); }
    MultiGpuRenderer(const MultiGpuRenderer&) = delete;
    MultiGpuRenderer& operator=(const MultiGpuRenderer&) = delete;
    size_t size() const { return _ctx.size(); }

    void setScene(const Camera& cam, const Geometry& world, const Light& lights) {
        FlatScene f = flatten(cam, world, lights);
        crt_scene_desc d = f.desc();
        for (crt_ctx* c : _ctx) check(crt_scene_upload(c, &d), c, "crt_scene_upload");
    }
    // Every GPU seeds all nx*ny streams (12 MB at 720x720): a rank only ever touches those of its own tiles.
    void randomInit(unsigned int nx, unsigned int ny, unsigned long long seed) {
        for (size_t r = 0; r < _ctx.size(); ++r) check(crt_rng_init(_ctx[r], nx, ny, seed, _streams[r]), _ctx[r], "crt_rng_init");
    }

    // Allocates render()'s device buffers for an nx x ny image (render() does it on first use; calling it
    // beforehand keeps the allocation out of a timed render).
    void reserve(unsigned int nx, unsigned int ny) {
        if (_buf_nx == nx && _buf_ny == ny && _fb) return;
        free_buffers();
        const uint32_t world = static_cast<uint32_t>(_ctx.size());
        const size_t shard_floats = static_cast<size_t>(crt_shard_tiles_max(nx, ny, world)) * 256u;
        _compact.assign(world, nullptr);
        try {
            for (uint32_t r = 0; r < world; ++r) check(crt_fb_alloc(_ctx[r], shard_floats, &_compact[r]), _ctx[r], "crt_fb_alloc");
            check(crt_fb_alloc(_ctx[0], shard_floats * world, &_gathered), _ctx[0], "crt_fb_alloc");
            check(crt_fb_alloc(_ctx[0], static_cast<size_t>(nx) * ny * 4, &_fb), _ctx[0], "crt_fb_alloc");
        } catch (...) {
            free_buffers();
            throw;
        }
        _buf_nx = nx;
        _buf_ny = ny;
    }

    // Renders into a host RGBA float image (nx*ny*4).
    std::vector<float> render(const RenderSettings& s) {
        if (!s.renderAllAtOnce) throw std::invalid_argument("the progressive walk is a single-GPU mode");
        crt_settings st = s.flatten();
        const uint32_t world = static_cast<uint32_t>(_ctx.size());
        const size_t shard_floats = static_cast<size_t>(crt_shard_tiles_max(s.nx, s.ny, world)) * 256u;
        const size_t n = static_cast<size_t>(s.nx) * s.ny * 4;
        std::vector<float> host(n);
        reserve(s.nx, s.ny);
        // every rank renders its tiles
        for_each_rank([&](size_t r) {
            check(crt_render_shard(_ctx[r], &st, static_cast<uint32_t>(r), world, _compact[r], _streams[r]), _ctx[r],
                  "crt_render_shard");
        });
        // one gather to rank 0, queued behind each rank's render on its stream
        nccl(ncclGroupStart(), "ncclGroupStart");
        for (uint32_t r = 0; r < world; ++r)
            nccl(ncclGather(_compact[r], r == 0 ? _gathered : nullptr, shard_floats, ncclFloat, 0, _comms[r], _streams[r]),
                 "ncclGather");
        nccl(ncclGroupEnd(), "ncclGroupEnd");
        check(crt_scatter_shards(_ctx[0], s.nx, s.ny, world, _gathered, _fb, _streams[0]), _ctx[0], "crt_scatter_shards");
        check(crt_fb_download(_ctx[0], _fb, host.data(), n, _streams[0]), _ctx[0], "crt_fb_download");
        for (uint32_t r = 1; r < world; ++r) { // the senders' part of the gather
            hip(hipSetDevice(_devices[r]), "hipSetDevice");
            hip(hipStreamSynchronize(_streams[r]), "hipStreamSynchronize");
        }
        return host;
    }

private:
    void free_buffers() {
        for (size_t r = 0; r < _compact.size(); ++r)
            if (_compact[r] && r < _ctx.size() && _ctx[r]) crt_fb_free(_ctx[r], _compact[r]);
        _compact.clear();
        if (_gathered && !_ctx.empty() && _ctx[0]) crt_fb_free(_ctx[0], _gathered);
        if (_fb && !_ctx.empty() && _ctx[0]) crt_fb_free(_ctx[0], _fb);
        _gathered = _fb = nullptr;
        _buf_nx = _buf_ny = 0;
    }
    void release() {
        free_buffers();
        for (ncclComm_t c : _comms)
            if (c) ncclCommDestroy(c);
        _comms.clear();
        for (size_t r = 0; r < _ctx.size(); ++r) {
            if (_streams[r] && hipSetDevice(_devices[r]) == hipSuccess) {
                (void)hipStreamSynchronize(_streams[r]);
                (void)hipStreamDestroy(_streams[r]);
            }
            crt_destroy(_ctx[r]);
        }
        _ctx.clear();
        _streams.clear();
    }
};

} // namespace craytracer
