// Sphere primitive (reference: Sphere.hpp:18-38).
#pragma once
#include "Geometry.hpp"

namespace craytracer {

class Sphere : public Geometry {
    vec3 _center;
    float _radius;

public:
    Sphere() : _center(vec3::zero()), _radius(0.0f) {}
    explicit Sphere(float radius) : _center(vec3::zero()), _radius(radius) {}
    explicit Sphere(vec3 center) : _center(center), _radius(1.0f) {}
    Sphere(vec3 center, float radius) : _center(center), _radius(radius) {}

    vec3 getCenter() const { return _center; }
    float getRadius() const { return _radius; }
    void setCenter(vec3 value) { _center = value; }
    void setRadius(float value) { _radius = value; }

    void flatten(std::vector<crt_geometry>& out) const override {
        crt_geometry g{};
        g.kind = CRT_GEOM_SPHERE;
        g.mat = to_flat(_mat);
        for (int i = 0; i < 3; ++i) g.center[i] = _center[i];
        g.radius = _radius;
        out.push_back(g);
    }
};
} // namespace craytracer
