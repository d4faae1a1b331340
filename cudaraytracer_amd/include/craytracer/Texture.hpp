// RGBA float texture with nearest-texel lookup and the three wrap modes (reference: Texture.hpp:25-29,
// 31-38,82-124,126-185,197-252).  On the render path the only textures are the two LTC tables, which the
// library keeps on the device and always reads with CLAMP (AreaLight.hpp:51-54); this host class is the rest of
// the reference's Texture surface for callers that use it directly.  Image files are not read here (the
// reference goes through stb_image, Texture.hpp:254-330): a texture is built from float data.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#include "mstd.hpp"

namespace craytracer {
using namespace mstd;

enum class WRAP_MODE : uint8_t { CLAMP = 0, MIRROR = 1, REPEAT = 2 };

class Texture {
    std::vector<float> _data; // RGBA, row-major
    unsigned int _width = 0u;
    unsigned int _height = 0u;
    unsigned int _channels = 4u;
    WRAP_MODE _verticalWrapMode = WRAP_MODE::REPEAT;
    WRAP_MODE _horizontalWrapMode = WRAP_MODE::REPEAT;

    static float _fract(float x) { return x - std::floor(x); } // math_functions.hpp:94-97

    // One coordinate outside [0, 1] under a wrap mode (Texture.hpp:129-152).
    static float _wrap(float f, WRAP_MODE mode) {
        if (!(f < 0.f || f > 1.f)) return f;
        switch (mode) {
            case WRAP_MODE::CLAMP: return saturate(f);
            case WRAP_MODE::REPEAT: return _fract(f);
            case WRAP_MODE::MIRROR: return std::fabs(_fract(f) - 1.f);
        }
        return f;
    }

    // Texel of a uv pair: wrap, then scale by (size - 1) and truncate (Texture.hpp:126-185).
    uvec2 _imageSpaceCoordinates(vec2 uv) const {
        const float x = _wrap(uv.x(), _horizontalWrapMode), y = _wrap(uv.y(), _verticalWrapMode);
        return uvec2(static_cast<unsigned int>(x * static_cast<float>(_width - 1u)),
                     static_cast<unsigned int>(y * static_cast<float>(_height - 1u)));
    }

public:
    Texture() = default;
    // 1 channel: grey, alpha 1; 2: red and green; 3: RGB, alpha 1; 4: as given (Texture.hpp:82-124,197-203).
    Texture(const float* data, unsigned int width, unsigned int height, unsigned int channels,
            WRAP_MODE verticalWrapMode = WRAP_MODE::REPEAT, WRAP_MODE horizontalWrapMode = WRAP_MODE::REPEAT)
        : _width(width), _height(height), _verticalWrapMode(verticalWrapMode), _horizontalWrapMode(horizontalWrapMode) {
        const size_t size = static_cast<size_t>(width) * height;
        _data.resize(size * 4u);
        for (size_t i = 0; i < size; ++i) {
            float r = 0.f, g = 0.f, b = 0.f, a = 1.f;
            const float* px = data + i * channels;
            switch (channels) {
                case 1: r = g = b = px[0]; break;
                case 2: r = px[0]; g = px[1]; break;
                case 3: r = px[0]; g = px[1]; b = px[2]; break;
                case 4: r = px[0]; g = px[1]; b = px[2]; a = px[3]; break;
                default: break;
            }
            _data[i * 4u + 0u] = r;
            _data[i * 4u + 1u] = g;
            _data[i * 4u + 2u] = b;
            _data[i * 4u + 3u] = a;
        }
    }

    unsigned int getWidth() const { return _width; }
    unsigned int getHeight() const { return _height; }
    bool isImageLoaded() const { return !_data.empty(); }
    WRAP_MODE getVerticalWrapMode() const { return _verticalWrapMode; }
    WRAP_MODE getHorizontalWrapMode() const { return _horizontalWrapMode; }
    void setVerticalWrapMode(WRAP_MODE mode) { _verticalWrapMode = mode; }
    void setHorizontalWrapMode(WRAP_MODE mode) { _horizontalWrapMode = mode; }

    vec4 getPixelColor(size_t pos) const { // Texture.hpp:211-222
        if (_data.empty() || pos >= static_cast<size_t>(_width) * _height) return vec4();
        const size_t p = pos * _channels;
        return vec4(_data[p], _data[p + 1u], _data[p + 2u], _data[p + 3u]);
    }
    vec4 getPixelColor(unsigned int x, unsigned int y) const { // Texture.hpp:224-237
        if (_data.empty() || x >= _width || y >= _height) return vec4();
        return getPixelColor(static_cast<size_t>(y) * _width + x);
    }
    vec4 sample(vec2 uv) const { // Texture.hpp:249-252
        const uvec2 pos = _imageSpaceCoordinates(uv);
        return getPixelColor(pos.x(), pos.y());
    }
};
} // namespace craytracer
