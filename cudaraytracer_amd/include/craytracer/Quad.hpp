// Quadrilateral primitive with the reference's four constructors (Quad.hpp:105-182):
// positions only (face normal + planar auto-UV), positions + UVs, positions +
// normals (auto-UV), or four complete vertices.
#pragma once
#include <algorithm>
#include <tuple>

#include "Geometry.hpp"
#include "Vertex.hpp"

namespace craytracer {

class Quad : public Geometry {
    Vertex _v0, _v1, _v2, _v3;

    // Planar UVs normalised to [0,1] on the dominant-axis projection (Quad.hpp:29-100).
    static std::tuple<vec2, vec2, vec2, vec2> _getUVs(vec3 v0, vec3 v1, vec3 v2, vec3 v3, vec3 normal) {
        const float ax = std::fabs(normal.x()), ay = std::fabs(normal.y()), az = std::fabs(normal.z());
        auto project = [&](const vec3& v) {
            if (ax > ay && ax > az) return vec2(v.y(), v.z());
            if (ay > az) return vec2(v.x(), v.z());
            return vec2(v.x(), v.y());
        };
        vec2 uv[4] = {project(v0), project(v1), project(v2), project(v3)};
        float minU = std::min({uv[0].x(), uv[1].x(), uv[2].x(), uv[3].x()});
        float maxU = std::max({uv[0].x(), uv[1].x(), uv[2].x(), uv[3].x()});
        float minV = std::min({uv[0].y(), uv[1].y(), uv[2].y(), uv[3].y()});
        float maxV = std::max({uv[0].y(), uv[1].y(), uv[2].y(), uv[3].y()});
        float rangeU = maxU - minU, rangeV = maxV - minV;
        if (epsilon_equal(rangeU, 0.f, EPSILON)) rangeU = 1.0f;
        if (epsilon_equal(rangeV, 0.f, EPSILON)) rangeV = 1.0f;
        auto norm = [&](const vec2& p) { return vec2((p.x() - minU) / rangeU, (p.y() - minV) / rangeV); };
        return {norm(uv[0]), norm(uv[1]), norm(uv[2]), norm(uv[3])};
    }
    static vec3 _faceNormal(vec3 v0, vec3 v1, vec3 v2) { return ((v1 - v0).cross(v2 - v0)).normalize(); }
    void _setPositions(vec3 v0, vec3 v1, vec3 v2, vec3 v3) {
        _v0.pos = v0;
        _v1.pos = v1;
        _v2.pos = v2;
        _v3.pos = v3;
    }
    void _setNormals(vec3 n0, vec3 n1, vec3 n2, vec3 n3) {
        _v0.normal = n0;
        _v1.normal = n1;
        _v2.normal = n2;
        _v3.normal = n3;
    }
    void _setUVs(const std::tuple<vec2, vec2, vec2, vec2>& t) {
        _v0.tex = std::get<0>(t);
        _v1.tex = std::get<1>(t);
        _v2.tex = std::get<2>(t);
        _v3.tex = std::get<3>(t);
    }

public:
    Quad() = default;
    Quad(vec3 v0, vec3 v1, vec3 v2, vec3 v3) {
        _setPositions(v0, v1, v2, v3);
        vec3 n = _faceNormal(v0, v1, v2);
        _setNormals(n, n, n, n);
        _setUVs(_getUVs(v0, v1, v2, v3, n));
    }
    Quad(vec3 v0, vec3 v1, vec3 v2, vec3 v3, vec2 vt0, vec2 vt1, vec2 vt2, vec2 vt3) {
        _setPositions(v0, v1, v2, v3);
        vec3 n = _faceNormal(v0, v1, v2);
        _setNormals(n, n, n, n);
        _setUVs({vt0, vt1, vt2, vt3});
    }
    Quad(vec3 v0, vec3 v1, vec3 v2, vec3 v3, vec3 vn0, vec3 vn1, vec3 vn2, vec3 vn3) {
        _setPositions(v0, v1, v2, v3);
        _setNormals(vn0, vn1, vn2, vn3);
        _setUVs(_getUVs(v0, v1, v2, v3, _faceNormal(v0, v1, v2)));
    }
    Quad(Vertex v0, Vertex v1, Vertex v2, Vertex v3) : _v0(v0), _v1(v1), _v2(v2), _v3(v3) {}

    std::tuple<Vertex, Vertex, Vertex, Vertex> getVertices() const { return {_v0, _v1, _v2, _v3}; }

    void flatten(std::vector<crt_geometry>& out) const override {
        crt_geometry g{};
        g.kind = CRT_GEOM_QUAD;
        g.mat = to_flat(_mat);
        const Vertex* vs[4] = {&_v0, &_v1, &_v2, &_v3};
        for (int k = 0; k < 4; ++k) {
            for (int i = 0; i < 3; ++i) g.v[k].pos[i] = vs[k]->pos[i];
            for (int i = 0; i < 2; ++i) g.v[k].tex[i] = vs[k]->tex[i];
            for (int i = 0; i < 3; ++i) g.v[k].normal[i] = vs[k]->normal[i];
        }
        out.push_back(g);
    }
};
} // namespace craytracer
