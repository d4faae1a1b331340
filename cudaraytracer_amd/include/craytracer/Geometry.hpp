// Base of the scene primitives (reference: Geometry.hpp:16-38).  Host-side: a
// primitive carries its Material and knows how to write itself into the flat
// crt_geometry record the renderer consumes; intersection runs on the GPU.
#pragma once
#include <vector>

#include "../../../include/crt_hip.h"
#include "Color.hpp"
#include "Material.hpp"
#include "Ray.hpp"

namespace craytracer {

inline crt_material to_flat(const Material& m) {
    crt_material f{};
    f.type = static_cast<uint32_t>(m.type);
    for (int i = 0; i < 4; ++i) {
        f.ambient[i] = m.ambient[i];
        f.diffuse[i] = m.diffuse[i];
        f.specular[i] = m.specular[i];
    }
    f.shininess = m.shininess;
    f.refractIndex = m.refractIndex;
    return f;
}

class Geometry {
protected:
    // Default material of a primitive (Geometry.hpp:18-29)
    Material _mat = {MaterialType::Diffuse, Color::white() * 0.1f, Color::white(), Color::white(), 0.f, 0.f};

public:
    Geometry() = default;
    virtual ~Geometry() = default;
    Material getMaterial() const { return _mat; }
    void setMaterial(Material value) { _mat = value; }
    // Appends this primitive (a list appends its members in order).
    virtual void flatten(std::vector<crt_geometry>& out) const = 0;
};
} // namespace craytracer
