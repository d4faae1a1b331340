// Ray and RayHit value types (reference: Ray.hpp:19-106).
#pragma once
#include <cfloat>

#include "Material.hpp"
#include "mstd.hpp"

namespace craytracer {
using namespace mstd;

struct RayHit {
    bool isHit = false;
    vec3 hitPoint = vec3::zero();
    vec3 hitNormal = vec3(0.0f, 1.0f, 0.0f);
    vec2 hitUV = vec2::zero();
    float hitDist = FLT_MAX;
    const Material* hitMat = nullptr;
};

class Ray {
    vec3 _origin = vec3::zero();
    vec3 _direction = vec3(0.0f, 0.0f, -1.0f);
    float _distance = 0.0f;

public:
    Ray() = default;
    // A direction shorter than 1e-4 is kept as given; anything else is normalised (Ray.hpp:37-52).
    Ray(vec3 origin, vec3 direction) : _origin(origin) {
        _direction = epsilon_equal(direction.length_sq(), 0.f, EPSILON_SQ) ? direction : direction.normalized();
    }
    Ray(vec3 origin, vec3 direction, float distance) : Ray(origin, direction) { setDistance(distance); }

    vec3 getOrigin() const { return _origin; }
    vec3 getDirection() const { return _direction; }
    float getDistance() const { return _distance; }
    void setOrigin(vec3 value) { _origin = value; }
    void setDirection(vec3 value) {
        if (epsilon_equal(value.length_sq(), 0.f, EPSILON_SQ)) return;
        _direction = value.normalized();
    }
    void setDistance(float value) { _distance = value > 0.f ? value : 0.f; }

    // Ray.hpp:80-105
    bool isPointOnRay(vec3 point) const {
        vec3 v = point - _origin;
        vec3 c = _direction.cross(v);
        if (c.length_sq() >= EPSILON_SQ) return false;
        float d = v.dot(_direction);
        if (epsilon_equal(_distance, 0.f, EPSILON)) return d >= 0.f;
        return d >= 0.f && d <= _distance;
    }
};
} // namespace craytracer
