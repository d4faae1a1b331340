// Device-resident ("flattened") form of the scene that the render kernels read,
// and the host-side routine that derives it from a crt_scene_desc.
//
// Every derived field is a value the reference recomputes per ray from the same
// inputs with the same float operations (cited per field), evaluated here once
// on the host in IEEE binary32 without contraction, so hoisting it does not
// change a single bit of the result.
#pragma once
#include <stdint.h>

#include "../../include/crt_hip.h"

namespace crt {

// All members are 4-byte so that a wave-uniform index turns into scalar loads.
struct DevGeom {
    uint32_t kind;       // CRT_GEOM_*
    uint32_t mat_type;   // CRT_MAT_*
    float ambient[4];    // Material::ambient
    float diffuse[4];    // Material::diffuse
    float specular[4];   // Material::specular
    float refract_index; // Material::refractIndex
    uint32_t ltc_col;    // texel column of (roughness*LUT_SCALE+LUT_BIAS): AreaLight.hpp:608,619,625 + Texture.hpp:179-182
    float shininess;
    // sphere (Sphere.hpp:40-61)
    float center[3];
    float radius;
    float radius_sq; // radius * radius (Sphere.hpp:56)
    // quad (Quad.hpp:202-250): everything that depends only on the vertices
    float v0[3];
    float a[3];   // v1 - v0
    float b[3];   // v3 - v0
    float nor[3]; // a x b
    uint32_t id, idu, idv; // projection axis and the two kept axes (lut {1,2,0,1})
    float ka[2], kb[2], kg[2]; // projected a, b and kc - kb - ka
    float k2;        // cross2d(kc - kb, ka)
    float nor_id;    // nor[id]
    uint32_t linear; // |k2| < 1e-4 -> linear branch (Quad.hpp:255)
    float n[4][3];   // vertex normals
};

struct DevLight {
    uint32_t kind;
    float pos[3];
    float color[3];
    float intensity;
    uint32_t two_sided;
    uint32_t shadow_samples;
    uint32_t initialized; // AreaLight::_initialized (AreaLight.hpp:586)
    float pts[4][3]; // pos + points[i] (AreaLight.hpp:595-602)
    float p[4][3];   // points[i] (relative corners)
    float e01[3], e03[3]; // points[1]-points[0], points[3]-points[0] (AreaLight.hpp:359)
    float e23[3], e21[3]; // points[3]-points[2], points[1]-points[2] (AreaLight.hpp:370)
    float light_normal[3]; // (pts[1]-pts[0]) x (pts[3]-pts[0]) (AreaLight.hpp:250)
    // _hittedLight (AreaLight.hpp:382-432), same derivation as a quad
    float a[3], b[3], nor[3];
    uint32_t id, idu, idv;
    float ka[2], kb[2], kg[2];
    float k2, nor_id;
    uint32_t linear;
};

struct DevCamera {
    float pos[3], front[3], right[3], up[3]; // Camera.hpp:97-98
    float one_over_tan;                      // 1 / tan(fov/2) (Camera.hpp:133)
    float ortho_scale;
    uint32_t type;
};

struct DevScene {
    DevCamera cam;
    uint32_t n_geom;
    uint32_t n_lights;
    DevGeom geom[CRT_MAX_GEOMETRY];
    DevLight lights[CRT_MAX_LIGHTS];
};

// Returns CRT_OK or CRT_ERR_INVALID (message in err, if non-null).
int flatten_scene(const crt_scene_desc* in, DevScene* out, const float* ltc_tables, char* err, size_t errcap);

} // namespace crt
