// Pinhole / orthographic camera (reference: Camera.hpp:19-145).
#pragma once
#include <cstdint>

#include "../../../include/crt_hip.h"
#include "Ray.hpp"

namespace craytracer {

enum class CameraType : uint8_t { ORTHOGRAPHIC = 0, PERSPECTIVE = 1 };

class Camera {
    float _fov = 0.785f;
    float _orthoScale = 2.0f;
    CameraType _type = CameraType::PERSPECTIVE;
    vec3 _pos = vec3::zero();
    vec3 _front = vec3(0.0f, 0.0f, -1.0f);

public:
    Camera() = default;
    Camera(vec3 position, vec3 front, CameraType type, float fovRad, float orthoScale) {
        _pos = position;
        _type = type;
        _front = epsilon_equal(front.length_sq(), 0.f, EPSILON_SQ) ? vec3(0.f, 0.f, -1.f) : front.normalized();
        setFov(fovRad);
        setOrthographicScale(orthoScale);
    }

    float getFov() const { return _fov; }
    float getOrthographicScale() const { return _orthoScale; }
    vec3 getPosition() const { return _pos; }
    vec3 getFront() const { return _front; }
    CameraType getType() const { return _type; }
    void setFov(float radians) { _fov = radians < EPSILON ? EPSILON : radians; }
    void setOrthographicScale(float value) { _orthoScale = value < EPSILON ? EPSILON : value; }
    void setPosition(vec3 value) { _pos = value; }
    void setFront(vec3 value) {
        if (epsilon_equal(value.length_sq(), 0.f, EPSILON_SQ)) return;
        _front = value.normalized();
    }

    // Primary ray through the image-plane point (x, y); width/height are the
    // image-plane extents (Camera.hpp:86-144).  The renderer evaluates the same
    // expression per sample on the GPU; this host version serves tools and tests.
    Ray getRay(float x, float y, float width, float height) const {
        vec3 rightDir = _front.cross(vec3(0.f, 1.f, 0.f)).normalized();
        vec3 upDir = rightDir.cross(_front).normalize();
        switch (_type) {
            case CameraType::ORTHOGRAPHIC: {
                float aspect = height / width;
                float orthoWidth = _orthoScale * (aspect > 1.f ? aspect : 1.f);
                float orthoHeight = _orthoScale * (aspect > 1.f ? aspect : 1.f);
                vec3 rayOrigin = x * rightDir * orthoWidth + y * upDir * orthoHeight;
                rayOrigin += _pos;
                return Ray(rayOrigin, _front);
            }
            case CameraType::PERSPECTIVE: {
                vec3 rayOrigin = x * rightDir + y * upDir;
                rayOrigin += _pos;
                float one_over_tan = 1.0f / std::tan(_fov * 0.5f);
                vec3 camPos = _pos - ((width * 0.5f * one_over_tan) * _front);
                return Ray(rayOrigin, rayOrigin - camPos);
            }
        }
        return Ray();
    }

    crt_camera flatten() const {
        crt_camera c{};
        for (int i = 0; i < 3; ++i) {
            c.pos[i] = _pos[i];
            c.front[i] = _front[i];
        }
        c.fov = _fov;
        c.orthoScale = _orthoScale;
        c.type = static_cast<uint32_t>(_type);
        return c;
    }
};
} // namespace craytracer
