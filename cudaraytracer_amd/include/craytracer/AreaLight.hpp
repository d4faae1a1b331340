// Quad area light shaded with linearly transformed cosines and stochastic soft
// shadows (reference: AreaLight.hpp:22-70,534-568).  The host object keeps the
// four corners relative to their centroid, exactly as the reference's
// constructor and rotate() leave them; shading runs on the GPU.
#pragma once
#include "Light.hpp"

namespace craytracer {

enum class VERTEX_ORDER : uint8_t { CCW = 0, CW = 1 };

class AreaLight : public Light {
    vec3 _points[4] = {vec3(-1.f, -1.f, 0.f), vec3(-1.f, 1.f, 0.f), vec3(1.f, 1.f, 0.f), vec3(1.f, -1.f, 0.f)};
    bool _twoSided = false;
    bool _initialized = false;
    bool _clipless = true; // no setter, as in the reference (AreaLight.hpp:42): the clipped branch is reachable through crt_light only
    unsigned int _shadowSamples = 10u;

    // Moves the light position to the centroid and makes the corners relative to it (AreaLight.hpp:56-70).
    void _defineCenterPoint() {
        for (const vec3& pt : _points) {
            _pos.x() += pt.x();
            _pos.y() += pt.y();
            _pos.z() += pt.z();
        }
        _pos.x() *= 0.25f;
        _pos.y() *= 0.25f;
        _pos.z() *= 0.25f;
        for (vec3& pt : _points) pt = pt - _pos;
    }

public:
    AreaLight() = default;
    AreaLight(vec3 p0, vec3 p1, vec3 p2, vec3 p3, vec4 color, unsigned int shadowSamples = 10u, float intensity = 1.f,
              bool twoSided = false, VERTEX_ORDER order = VERTEX_ORDER::CCW)
        : Light(vec3::zero(), color, intensity), _twoSided(twoSided), _shadowSamples(shadowSamples) {
        _points[0] = p0;
        _points[2] = p2;
        if (order == VERTEX_ORDER::CCW) {
            _points[1] = p3;
            _points[3] = p1;
        } else {
            _points[1] = p1;
            _points[3] = p3;
        }
        _defineCenterPoint();
        _initialized = true;
    }

    void rotate(vec3 axis, float radians) {
        if (epsilon_equal(axis.length_sq(), 0.f, EPSILON_SQ)) return;
        vec3 rotAxis = axis.normalized();
        for (vec3& pt : _points) pt.rotate(rotAxis, radians);
    }

    bool isTwoSided() const { return _twoSided; }
    unsigned int getShadowSamples() const { return _shadowSamples; }
    const vec3* getPoints() const { return _points; }

    void flatten(std::vector<crt_light>& out) const override {
        crt_light l{};
        l.kind = CRT_LIGHT_AREA;
        for (int i = 0; i < 3; ++i) l.pos[i] = _pos[i];
        for (int i = 0; i < 4; ++i) l.color[i] = _color[i];
        l.intensity = _intensity;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 3; ++i) l.points[k][i] = _points[k][i];
        l.twoSided = _twoSided ? 1u : 0u;
        l.shadowSamples = _shadowSamples;
        l.initialized = _initialized ? 1u : 0u;
        l.horizonClip = _clipless ? 0u : 1u;
        out.push_back(l);
    }
};
} // namespace craytracer
