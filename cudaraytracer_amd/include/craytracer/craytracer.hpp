// Umbrella header of the host-side scene API plus the pieces that tie it to the
// C ABI: RenderSettings (the constants of the reference's main(), main.cu:626-636),
// flatten() (scene objects -> POD records), CornellBox (create_world,
// main.cu:370-595) and Renderer (the launch sequence of main(), main.cu:668-716,856).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/crt_hip.h"
#include "AreaLight.hpp"
#include "Camera.hpp"
#include "Color.hpp"
#include "Geometry.hpp"
#include "GeometryList.hpp"
#include "Light.hpp"
#include "LightList.hpp"
#include "Material.hpp"
#include "Quad.hpp"
#include "Ray.hpp"
#include "Sphere.hpp"
#include "Vertex.hpp"

namespace craytracer {

struct RenderSettings {
    bool renderAllAtOnce = true;
    unsigned int blocksPerDraw = 200;
    unsigned int nx = 720;
    unsigned int ny = 720;
    unsigned int tx = 19;
    unsigned int ty = 19;
    unsigned int aa_iter = 1;
    unsigned int ref_iter = 4;
    unsigned int gl_iter = 2;
    unsigned int ind_rays = 75;
    unsigned int shadowSamples = 50;

    crt_settings flatten() const {
        return crt_settings{renderAllAtOnce ? 1u : 0u, blocksPerDraw, nx, ny, tx, ty, aa_iter, ref_iter, gl_iter,
                            ind_rays, shadowSamples};
    }
};

// Owns the POD arrays a crt_scene_desc points into.
struct FlatScene {
    crt_camera camera{};
    std::vector<crt_geometry> geometry;
    std::vector<crt_light> lights;
    crt_scene_desc desc() const {
        return crt_scene_desc{camera, static_cast<uint32_t>(geometry.size()), geometry.data(),
                              static_cast<uint32_t>(lights.size()), lights.data()};
    }
};

inline FlatScene flatten(const Camera& cam, const Geometry& world, const Light& lights) {
    FlatScene f;
    f.camera = cam.flatten();
    world.flatten(f.geometry);
    lights.flatten(f.lights);
    return f;
}

// The scene built by create_world (main.cu:373-593): same objects, same order.
struct CornellBox {
    Camera camera;
    std::vector<std::unique_ptr<Geometry>> objects;
    std::vector<Geometry*> object_ptrs;
    GeometryList world;
    std::unique_ptr<AreaLight> area;
    Light* light_ptrs[1] = {nullptr};
    LightList lights;

    explicit CornellBox(unsigned int shadowSamples = 50u) {
        camera = Camera(vec3::zero(), vec3(0.f, 0.f, -1.f), CameraType::PERSPECTIVE, deg_to_rad(45.f), 2.f);
        const vec3 quadPoints[4] = {vec3(-.25f, .98f, -1.25f), vec3(.25f, .98f, -1.25f), vec3(.25f, .98f, -.75f),
                                    vec3(-.25f, .98f, -.75f)};
        auto diffuse = [](vec4 c, float ambientScale, float shininess) {
            return Material{MaterialType::Diffuse, c * ambientScale, c, c, shininess, 0.f};
        };
        const Material reflect{MaterialType::Reflect, Color::white() * 0.1f, Color::white(), Color::white(), 0.f, 0.f};
        const Material refractive{MaterialType::Refractive, Color::white() * 0.1f, Color::white(), Color::white(), 0.f, 1.5f};
        const Material whiteEmissive{MaterialType::Diffuse, Color::white(), Color::white(), Color::white(), 0.f, 0.f};
        const Material white = diffuse(Color::white(), 0.1f, 5.f);
        const Material red = diffuse(Color::red(), 0.1f, 5.f);
        const Material blue = diffuse(Color::blue(), 0.1f, 5.f);

        auto add = [&](Geometry* g, const Material& m) {
            g->setMaterial(m);
            objects.emplace_back(g);
        };
        auto wall = [&](vec3 a, vec3 b, vec3 c, vec3 d, vec3 n, const Material& m) {
            add(new Quad(Vertex{a, vec2(0.f, 0.f), n}, Vertex{b, vec2(1.f, 0.f), n}, Vertex{c, vec2(1.f, 1.f), n},
                         Vertex{d, vec2(0.f, 1.f), n}),
                m);
        };
        add(new Sphere(vec3(-0.25f, -.72f, -1.1f), .275f), reflect);
        add(new Sphere(vec3(0.3f, -.72f, -.6f), .275f), refractive);
        wall(vec3(-1.f, -1.f, -2.f), vec3(1.f, -1.f, -2.f), vec3(1.f, 1.f, -2.f), vec3(-1.f, 1.f, -2.f), vec3(0.f, 0.f, 1.f), white); // back
        wall(vec3(-1.f, 1.f, -2.f), vec3(1.f, 1.f, -2.f), vec3(1.f, 1.f, 0.f), vec3(-1.f, 1.f, 0.f), vec3(0.f, -1.f, 0.f), white);   // top
        wall(vec3(-1.f, -1.f, -2.f), vec3(1.f, -1.f, -2.f), vec3(1.f, -1.f, 0.f), vec3(-1.f, -1.f, 0.f), vec3(0.f, 1.f, 0.f), white); // bottom
        wall(vec3(1.f, -1.f, -2.f), vec3(1.f, -1.f, 0.f), vec3(1.f, 1.f, 0.f), vec3(1.f, 1.f, -2.f), vec3(-1.f, 0.f, 0.f), blue);    // right
        wall(vec3(-1.f, -1.f, -2.f), vec3(-1.f, 1.f, -2.f), vec3(-1.f, 1.f, 0.f), vec3(-1.f, -1.f, 0.f), vec3(1.f, 0.f, 0.f), red);  // left
        auto lifted = [&](int k) { return vec3(quadPoints[k].x(), quadPoints[k].y() + 0.01f, quadPoints[k].z()); };
        wall(lifted(0), lifted(1), lifted(2), lifted(3), vec3(0.f, -1.f, 0.f), whiteEmissive); // emissive panel

        for (auto& o : objects) object_ptrs.push_back(o.get());
        world = GeometryList(object_ptrs.data(), static_cast<int>(object_ptrs.size()));

        area = std::make_unique<AreaLight>(quadPoints[0], quadPoints[1], quadPoints[2], quadPoints[3], Color::white(),
                                           shadowSamples, 10.f);
        area->rotate(vec3(1.f, 0.f, 0.f), deg_to_rad(180.f));
        light_ptrs[0] = area.get();
        lights = LightList(light_ptrs, 1);
    }
    CornellBox(const CornellBox&) = delete;
    CornellBox& operator=(const CornellBox&) = delete;
};

// RAII wrapper over the C ABI following the order of main(): create context,
// upload scene, seed the per-pixel RNG, render, read back, write file.hdr.
class Renderer {
    crt_ctx* _ctx = nullptr;
    float* _fb = nullptr; // device frame of render(), kept from call to call
    size_t _fb_floats = 0;
    static void check(int rc, crt_ctx* ctx, const char* what) {
        if (rc != CRT_OK)
            throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " +
                                     (ctx ? crt_last_error(ctx) : ""));
    }

public:
    explicit Renderer(int device = 0) { check(crt_create(&_ctx, device), nullptr, "crt_create"); }
    ~Renderer() {
        if (_fb) crt_fb_free(_ctx, _fb);
        crt_destroy(_ctx);
    }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    crt_ctx* handle() const { return _ctx; }

    void setScene(const Camera& cam, const Geometry& world, const Light& lights) {
        FlatScene f = flatten(cam, world, lights);
        crt_scene_desc d = f.desc();
        check(crt_scene_upload(_ctx, &d), _ctx, "crt_scene_upload");
    }
    void randomInit(unsigned int nx, unsigned int ny, unsigned long long seed) {
        check(crt_rng_init(_ctx, nx, ny, seed, nullptr), _ctx, "crt_rng_init");
    }
    // Allocates render()'s device frame for an nx x ny image (render() does it on first use; the reference
    // allocates its frame before the render timer starts, main.cu:699-700).  Pixels a progressive walk never
    // reaches read as 0.
    void reserve(unsigned int nx, unsigned int ny) {
        const size_t n = static_cast<size_t>(nx) * ny * 4;
        if (_fb && _fb_floats == n) return;
        if (_fb) crt_fb_free(_ctx, _fb);
        _fb = nullptr;
        _fb_floats = 0;
        check(crt_fb_alloc(_ctx, n, &_fb), _ctx, "crt_fb_alloc");
        _fb_floats = n;
    }
    // Renders into a host RGBA float image (nx*ny*4), honouring renderAllAtOnce / blocksPerDraw.
    std::vector<float> render(const RenderSettings& s) {
        crt_settings st = s.flatten();
        const size_t n = static_cast<size_t>(s.nx) * s.ny * 4;
        reserve(s.nx, s.ny);
        float* dev = _fb;
        std::vector<float> host(n);
        if (s.renderAllAtOnce) {
            check(crt_render(_ctx, &st, dev, nullptr), _ctx, "crt_render");
        } else {
            // the window arithmetic below divides by the block shape; the ABI would refuse these, too
            if (s.tx == 0 || s.ty == 0 || s.blocksPerDraw == 0)
                throw std::invalid_argument("tx, ty and blocksPerDraw must be positive");
            const unsigned int blocksX = s.nx / s.tx + 1, blocksY = s.ny / s.ty + 1;
            const unsigned int totalTiles = blocksX * blocksY;
            for (unsigned int tileIndex = 0; tileIndex < totalTiles; tileIndex += s.blocksPerDraw)
                check(crt_render_partial(_ctx, &st, tileIndex, dev, nullptr), _ctx, "crt_render_partial");
        }
        check(crt_fb_download(_ctx, dev, host.data(), n, nullptr), _ctx, "crt_fb_download");
        return host;
    }
};

} // namespace craytracer
