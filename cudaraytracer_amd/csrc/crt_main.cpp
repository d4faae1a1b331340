// Headless counterpart of the reference's main() (main.cu:614-894) without the
// SFML window: same settings (main.cu:626-636, here also settable from the
// command line), same two timed regions ("Create World", "Render"), result
// written to file.hdr.
//
//   crt_main [--nx N] [--ny N] [--aa_iter N] [--ref_iter N] [--gl_iter N] [--ind_rays N]
//            [--shadowSamples N] [--tx N] [--ty N] [--blocksPerDraw N] [--progressive]
//            [--seed S] [--device D] [--gpus N] [--out file.hdr] [--preview file.ppm]
//
// --gpus N renders on devices 0..N-1 of the node: interleaved 8x8 tiles, one RCCL gather to device 0
// (craytracer/multi_gpu.hpp); the image is the same, bit for bit, as on one GPU.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>

#include "../include/craytracer/craytracer.hpp"
#include "../include/craytracer/multi_gpu.hpp"

using namespace craytracer;

static void printTime(const char* prefix, double milliseconds) { // main.cu:979-989
    int total_ms = static_cast<int>(milliseconds);
    std::printf("%s time: %02d:%02d:%02d.%03d (hh:mm:ss.mmm)\n", prefix, total_ms / 3600000, (total_ms / 60000) % 60,
                (total_ms / 1000) % 60, total_ms % 1000);
}

int main(int argc, char** argv) {
    RenderSettings s;
    unsigned long long seed = static_cast<unsigned long long>(std::time(nullptr)); // main.cu:952
    int device = 0;
    int gpus = 0; // 0: the single-GPU path on `device`
    std::string out = "file.hdr";
    std::string preview; // 8-bit view of the frame, what the reference shows in its window (main.cu:812-818)
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--nx") s.nx = std::strtoul(next(), nullptr, 10);
        else if (a == "--ny") s.ny = std::strtoul(next(), nullptr, 10);
        else if (a == "--tx") s.tx = std::strtoul(next(), nullptr, 10);
        else if (a == "--ty") s.ty = std::strtoul(next(), nullptr, 10);
        else if (a == "--aa_iter") s.aa_iter = std::strtoul(next(), nullptr, 10);
        else if (a == "--ref_iter") s.ref_iter = std::strtoul(next(), nullptr, 10);
        else if (a == "--gl_iter") s.gl_iter = std::strtoul(next(), nullptr, 10);
        else if (a == "--ind_rays") s.ind_rays = std::strtoul(next(), nullptr, 10);
        else if (a == "--shadowSamples") s.shadowSamples = std::strtoul(next(), nullptr, 10);
        else if (a == "--blocksPerDraw") s.blocksPerDraw = std::strtoul(next(), nullptr, 10);
        else if (a == "--progressive") s.renderAllAtOnce = false;
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--preview") preview = next();
        else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    try {
        using clock = std::chrono::steady_clock;
        auto t0 = clock::now();
        CornellBox box(s.shadowSamples);
        std::vector<float> fb;
        if (gpus > 0) {
            std::vector<int> devices;
            for (int d = 0; d < gpus; ++d) devices.push_back(d);
            MultiGpuRenderer renderer(devices);
            std::printf("HIP initialized on %d GPU%s.\n", gpus, gpus == 1 ? "" : "s");
            renderer.randomInit(s.nx, s.ny, seed);
            renderer.setScene(box.camera, box.world, box.lights);
            renderer.reserve(s.nx, s.ny); // the frame is allocated inside "Create World" (main.cu:699-700)
            auto t1 = clock::now();
            printTime("Create World", std::chrono::duration<double, std::milli>(t1 - t0).count());
            fb = renderer.render(s);
            printTime("Render", std::chrono::duration<double, std::milli>(clock::now() - t1).count());
        } else {
            Renderer renderer(device);
            std::printf("HIP initialized.\n");
            renderer.randomInit(s.nx, s.ny, seed);
            renderer.setScene(box.camera, box.world, box.lights);
            renderer.reserve(s.nx, s.ny); // the frame is allocated inside "Create World" (main.cu:699-700)
            auto t1 = clock::now();
            printTime("Create World", std::chrono::duration<double, std::milli>(t1 - t0).count());
            fb = renderer.render(s); // returns after the device finished
            printTime("Render", std::chrono::duration<double, std::milli>(clock::now() - t1).count());
        }

        std::printf("Saving image...");
        if (crt_write_hdr(out.c_str(), static_cast<int>(s.nx), static_cast<int>(s.ny), 4, fb.data()) != CRT_OK) {
            std::fprintf(stderr, "\ncould not write %s\n", out.c_str());
            return 1;
        }
        std::printf("\nImage saved as '%s'.\n", out.c_str());
        if (!preview.empty()) {
            std::vector<uint8_t> px(fb.size());
            if (crt_to_rgba8(fb.data(), fb.size(), px.data()) != CRT_OK ||
                crt_write_ppm(preview.c_str(), static_cast<int>(s.nx), static_cast<int>(s.ny), px.data()) != CRT_OK) {
                std::fprintf(stderr, "could not write %s\n", preview.c_str());
                return 1;
            }
            std::printf("Preview saved as '%s'.\n", preview.c_str());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 99; // the reference exits with 99 on a device error (main.cu:901)
    }
    return 0;
}
