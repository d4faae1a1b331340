// Point light and the shading input record (reference: Light.hpp:21-33,114-116).
#pragma once
#include <vector>

#include "../../../include/crt_hip.h"
#include "Color.hpp"
#include "Geometry.hpp"

namespace craytracer {

struct LightInput {
    vec3 fragPos = vec3::zero();
    vec2 uv = vec2::zero();
    vec3 norm = vec3(0.0f, 1.0f, 0.0f);
    vec3 viewDir = vec3(0.0f, 0.0f, -1.0f);
    const Material* mat = nullptr;
};

class Light {
protected:
    vec3 _pos = vec3::zero();
    vec4 _color = Color::white();
    float _intensity = 1.f;

public:
    Light() = default;
    virtual ~Light() = default;
    Light(vec3 position, vec4 color, float intensity) : _pos(position), _color(color), _intensity(intensity) {}
    vec3 getPosition() const { return _pos; }
    vec4 getColor() const { return _color; }
    float getIntensity() const { return _intensity; }

    virtual void flatten(std::vector<crt_light>& out) const {
        crt_light l{};
        l.kind = CRT_LIGHT_POINT;
        for (int i = 0; i < 3; ++i) l.pos[i] = _pos[i];
        for (int i = 0; i < 4; ++i) l.color[i] = _color[i];
        l.intensity = _intensity;
        out.push_back(l);
    }
};
} // namespace craytracer
