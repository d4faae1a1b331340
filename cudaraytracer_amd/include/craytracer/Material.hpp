// Material aggregate of the scene API (reference: Material.hpp:13-27).
#pragma once
#include <cstdint>

#include "mstd.hpp"

#define AIR_INDEX 1.0f

namespace craytracer {
using namespace mstd;

enum class MaterialType : uint8_t { Diffuse = 0, Reflect = 1, Refractive = 2 };

struct Material {
    MaterialType type = MaterialType::Diffuse;
    vec4 ambient;
    vec4 diffuse;
    vec4 specular;
    float shininess = 0.f;
    float refractIndex = 0.f;
};
} // namespace craytracer
