// Quad corner (reference: Vertex.hpp:14-18).
#pragma once
#include "mstd.hpp"

namespace craytracer {
using namespace mstd;

struct Vertex {
    vec3 pos;
    vec2 tex;
    vec3 normal;
};
} // namespace craytracer
