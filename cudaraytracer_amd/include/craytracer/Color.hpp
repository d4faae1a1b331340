// Named RGBA constants (same names and values as the reference's Color class, Color.hpp:20-51).
#pragma once
#include "mstd.hpp"

namespace craytracer {
using namespace mstd;

class Color {
public:
#define CRT_COLOR(name, r, g, b) static vec4 name() { return vec4(r, g, b, 1.000f); }
    CRT_COLOR(red, 1.000f, 0.000f, 0.000f)
    CRT_COLOR(orange, 1.000f, 0.647f, 0.000f)
    CRT_COLOR(yellow, 1.000f, 1.000f, 0.000f)
    CRT_COLOR(lime, 0.196f, 0.803f, 0.196f)
    CRT_COLOR(green, 0.000f, 1.000f, 0.000f)
    CRT_COLOR(teal, 0.000f, 0.502f, 0.502f)
    CRT_COLOR(cyan, 0.000f, 1.000f, 1.000f)
    CRT_COLOR(turquoise, 0.251f, 0.878f, 0.816f)
    CRT_COLOR(lapis, 0.149f, 0.380f, 0.612f)
    CRT_COLOR(blue, 0.000f, 0.000f, 1.000f)
    CRT_COLOR(indigo, 0.294f, 0.000f, 0.510f)
    CRT_COLOR(violet, 0.933f, 0.510f, 0.933f)
    CRT_COLOR(purple, 0.502f, 0.000f, 0.502f)
    CRT_COLOR(magenta, 1.000f, 0.000f, 1.000f)
    CRT_COLOR(pink, 1.000f, 0.753f, 0.796f)
    CRT_COLOR(brown, 0.647f, 0.165f, 0.165f)
    CRT_COLOR(maroon, 0.502f, 0.000f, 0.000f)
    CRT_COLOR(olive, 0.502f, 0.502f, 0.000f)
    CRT_COLOR(gold, 1.000f, 0.843f, 0.000f)
    CRT_COLOR(silver, 0.753f, 0.753f, 0.753f)
    CRT_COLOR(navy, 0.000f, 0.000f, 0.502f)
    CRT_COLOR(mint, 0.741f, 0.988f, 0.788f)
    CRT_COLOR(beige, 0.961f, 0.961f, 0.863f)
    CRT_COLOR(salmon, 0.980f, 0.502f, 0.447f)
    CRT_COLOR(coral, 1.000f, 0.498f, 0.314f)
    CRT_COLOR(white, 1.000f, 1.000f, 1.000f)
    CRT_COLOR(gray, 0.500f, 0.500f, 0.500f)
    CRT_COLOR(black, 0.000f, 0.000f, 0.000f)
#undef CRT_COLOR
};
} // namespace craytracer
