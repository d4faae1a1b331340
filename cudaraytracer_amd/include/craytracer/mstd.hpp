// Host-side value types of the scene API: vec<N,T>, mat3, quat.
//
// Only what the scene classes need is here (the reference's mstd library is much
// larger; vec.hpp / mat.hpp / quat.hpp).  Float arithmetic follows the reference's
// device branch evaluated in IEEE binary32: length() is 1/(1/sqrt(len^2))
// (vec.hpp:398-409), division by a scalar multiplies by 1/y and leaves the vector
// unchanged for y == 0 (vec.hpp:739-769), dot products accumulate from zero in
// index order (vec.hpp:428-434).  The flattened scene therefore carries exactly
// the bits the reference's constructors would have produced.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace mstd {

constexpr float EPSILON = 1e-4f;    // math_types.hpp:78
constexpr float EPSILON_SQ = 1e-8f; // math_types.hpp:79
constexpr float DEG_TO_RAD = 0.01745329251994329576f; // math_types.hpp:80

inline bool epsilon_equal(float a, float b, float eps) { return std::fabs(a - b) < eps; } // math_functions.hpp:75-78
inline float deg_to_rad(float deg) { return deg * DEG_TO_RAD; }                            // math_functions.hpp:47-59
inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }

template <size_t N, class T>
class vec {
    static_assert(N > 0, "vec needs at least one component");
    T _v[N] = {};

public:
    static constexpr size_t size = N;
    using value_type = T;

    vec() = default;
    // vecN(x, y, ...) -- missing trailing components are zero (vec.hpp:140-143)
    template <class... Ts, class = std::enable_if_t<(sizeof...(Ts) > 0 && sizeof...(Ts) <= N) &&
                                                    (std::is_arithmetic_v<Ts> && ...)>>
    vec(Ts... xs) {
        T tmp[] = {static_cast<T>(xs)...};
        for (size_t i = 0; i < sizeof...(Ts); ++i) _v[i] = tmp[i];
    }
    // vecN(vecM): truncate or zero-pad (vec.hpp:189-192)
    template <size_t M, class U>
    vec(const vec<M, U>& o) {
        for (size_t i = 0; i < (N < M ? N : M); ++i) _v[i] = static_cast<T>(o[i]);
    }
    // vecN(vecM, z, ...) (vec.hpp:155-159)
    template <size_t M, class U, class... Ts,
              class = std::enable_if_t<(M < N && sizeof...(Ts) > 0 && sizeof...(Ts) <= N - M) &&
                                       (std::is_arithmetic_v<Ts> && ...)>>
    vec(const vec<M, U>& o, Ts... xs) {
        for (size_t i = 0; i < M; ++i) _v[i] = static_cast<T>(o[i]);
        T tmp[] = {static_cast<T>(xs)...};
        for (size_t i = 0; i < sizeof...(Ts); ++i) _v[M + i] = tmp[i];
    }

    static vec zero() { return vec(); }
    static vec fill(T v) {
        vec r;
        for (size_t i = 0; i < N; ++i) r._v[i] = v;
        return r;
    }
    static vec one() { return fill(T(1)); }
    bool is_zero() const {
        for (size_t i = 0; i < N; ++i)
            if (_v[i] != T(0)) return false;
        return true;
    }

    T& operator[](size_t i) { return _v[i]; }
    T operator[](size_t i) const { return _v[i]; }
    const T* data() const { return _v; }
    T& x() { return _v[0]; }
    T x() const { return _v[0]; }
    T& y() { static_assert(N > 1); return _v[1]; }
    T y() const { static_assert(N > 1); return _v[1]; }
    T& z() { static_assert(N > 2); return _v[2]; }
    T z() const { static_assert(N > 2); return _v[2]; }
    T& w() { static_assert(N > 3); return _v[3]; }
    T w() const { static_assert(N > 3); return _v[3]; }
    T& r() { return _v[0]; }
    T r() const { return _v[0]; }
    T& g() { static_assert(N > 1); return _v[1]; }
    T g() const { static_assert(N > 1); return _v[1]; }
    T& b() { static_assert(N > 2); return _v[2]; }
    T b() const { static_assert(N > 2); return _v[2]; }
    T& a() { static_assert(N > 3); return _v[3]; }
    T a() const { static_assert(N > 3); return _v[3]; }

    T length_sq() const { return dot(*this); }
    T dot(const vec& o) const {
        T r = T(0);
        for (size_t i = 0; i < N; ++i) r += _v[i] * o._v[i];
        return r;
    }
    T length() const { return T(1) / (T(1) / std::sqrt(length_sq())); }
    vec& normalize() {
        T len = length();
        if (len == T(0)) return *this;
        return *this /= len;
    }
    vec normalized() const {
        vec r = *this;
        return r.normalize();
    }
    vec cross(const vec& o) const {
        static_assert(N == 3);
        return vec(_v[1] * o._v[2] - _v[2] * o._v[1], _v[2] * o._v[0] - _v[0] * o._v[2],
                   _v[0] * o._v[1] - _v[1] * o._v[0]);
    }
    vec& saturate() {
        for (size_t i = 0; i < N; ++i) _v[i] = ::mstd::saturate(_v[i]);
        return *this;
    }
    vec& rotate(const vec& axis, T radians); // vec.hpp:639-655, defined after quat

    vec& operator+=(const vec& o) { for (size_t i = 0; i < N; ++i) _v[i] += o._v[i]; return *this; }
    vec& operator-=(const vec& o) { for (size_t i = 0; i < N; ++i) _v[i] -= o._v[i]; return *this; }
    vec& operator*=(const vec& o) { for (size_t i = 0; i < N; ++i) _v[i] *= o._v[i]; return *this; }
    vec& operator+=(T s) { for (size_t i = 0; i < N; ++i) _v[i] += s; return *this; }
    vec& operator-=(T s) { for (size_t i = 0; i < N; ++i) _v[i] -= s; return *this; }
    vec& operator*=(T s) { for (size_t i = 0; i < N; ++i) _v[i] *= s; return *this; }
    vec& operator/=(T s) {
        if (s == T(0)) return *this;
        T inv = T(1) / s;
        for (size_t i = 0; i < N; ++i) _v[i] *= inv;
        return *this;
    }
    vec operator+(const vec& o) const { vec r = *this; return r += o; }
    vec operator-(const vec& o) const { vec r = *this; return r -= o; }
    vec operator*(const vec& o) const { vec r = *this; return r *= o; }
    vec operator+(T s) const { vec r = *this; return r += s; }
    vec operator-(T s) const { vec r = *this; return r -= s; }
    vec operator*(T s) const { vec r = *this; return r *= s; }
    vec operator/(T s) const { vec r = *this; return r /= s; }
    friend vec operator*(T s, const vec& v) { return v * s; }
    vec operator-() const { return *this * T(-1); }
    bool operator==(const vec& o) const {
        for (size_t i = 0; i < N; ++i)
            if (!(_v[i] == o._v[i])) return false;
        return true;
    }
    bool operator!=(const vec& o) const { return !(*this == o); }
};

using vec2 = vec<2, float>;
using vec3 = vec<3, float>;
using vec4 = vec<4, float>;
using uvec2 = vec<2, unsigned int>;

template <class T, size_t N> T dot(const vec<N, T>& a, const vec<N, T>& b) { return a.dot(b); }
template <class T> vec<3, T> cross(const vec<3, T>& a, const vec<3, T>& b) { return a.cross(b); }

// Column-major 3x3 (mat.hpp: _values[column][row]).
class mat3 {
    float _m[3][3] = {};

public:
    mat3() = default;
    mat3(const vec3& c0, const vec3& c1, const vec3& c2) {
        for (int r = 0; r < 3; ++r) {
            _m[0][r] = c0[r];
            _m[1][r] = c1[r];
            _m[2][r] = c2[r];
        }
    }
    static mat3 identity() { return mat3(vec3(1.f, 0.f, 0.f), vec3(0.f, 1.f, 0.f), vec3(0.f, 0.f, 1.f)); }
    float* operator[](size_t col) { return _m[col]; }
    const float* operator[](size_t col) const { return _m[col]; }
    mat3& transpose() {
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < c; ++r) {
                float t = _m[c][r];
                _m[c][r] = _m[r][c];
                _m[r][c] = t;
            }
        return *this;
    }
    mat3 operator*(const mat3& o) const { // mat.hpp:1481-1492
        mat3 res;
        for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y)
                for (int i = 0; i < 3; ++i) res._m[x][y] += _m[i][y] * o._m[x][i];
        return res;
    }
    vec3 operator*(const vec3& v) const { // mat.hpp:1522-1530
        vec3 res;
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x) res[y] += _m[x][y] * v[x];
        return res;
    }
};

// quat<float>, the part vec3::rotate uses (quat.hpp:60-88,111-167,258-289).
struct quat {
    float s = 0.f;
    vec3 v;
    quat() = default;
    quat(float s_, const vec3& v_) : s(s_), v(v_) {}
    float magnitude() const { return 1.0f / (1.0f / std::sqrt(s * s + v.dot(v))); }
    quat& operator/=(float m) {
        if (m == 0.f) return *this;
        s /= m;
        v /= m;
        return *this;
    }
    quat& normalize() { return *this /= magnitude(); }
    quat& operator*=(const quat& o) {
        float t = s;
        s = s * o.s - v.dot(o.v);
        v = o.v * t + v * o.s + v.cross(o.v);
        return *this;
    }
    quat operator*(const quat& o) const { quat r = *this; return r *= o; }
    quat inverted() const {
        quat r = *this;
        float m = r.magnitude();
        m *= m;
        m = 1.0f / m;
        r.v *= -1.0f;
        r.s *= m;
        r.v *= m;
        return r;
    }
    static quat rotation(const vec3& axis, float radians) {
        quat q;
        if (!axis.is_zero()) q = quat(std::cos(radians * 0.5f), axis.normalized() * std::sin(radians * 0.5f));
        else q = quat(std::cos(radians * 0.5f), axis);
        if (q.magnitude() != 0.f) q.normalize();
        return q;
    }
};

template <size_t N, class T>
vec<N, T>& vec<N, T>::rotate(const vec<N, T>& axis, T radians) {
    static_assert(N == 3);
    const quat p(0.f, *this);
    vec3 norm_axis = axis;
    if (!norm_axis.is_zero()) norm_axis.normalize();
    const quat q = quat::rotation(norm_axis, radians);
    const quat inv = q.inverted();
    *this = (q * p * inv).v;
    return *this;
}

} // namespace mstd
