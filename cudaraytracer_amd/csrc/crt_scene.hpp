// Device-resident ("flattened") form of the scene that the render kernels read,
// and the host-side routine that derives it from a crt_scene_desc.
//
// Every derived field is a value the reference recomputes per ray from the same
// inputs with the same float operations (cited per field), evaluated here once
// on the host in IEEE binary32 without contraction, so hoisting it does not
// change a single bit of the result.
//
// Layout follows how the kernels touch the data.  Primitive records are split
// by temperature: `hot` is what every ray-primitive test reads (8 dwords, one
// scalar load per primitive when the loop index is wave-uniform), `uv` is read
// only by rays that survive the plane/range test of a quad, `shade` only for
// the one primitive a ray finally hit.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/crt_hip.h"

namespace crt {

// All members are 4-byte so that a wave-uniform index turns into scalar loads.
// DevGeomHot::kind: CRT_GEOM_SPHERE (0), CRT_GEOM_QUAD (1), or a quad whose plane normal has a
// single non-zero component (the dot products with it collapse to one multiplication).
#define CRT_HOT_QUAD_X 2u
#define CRT_HOT_QUAD_Y 3u
#define CRT_HOT_QUAD_Z 4u

struct alignas(32) DevGeomHot {
    uint32_t kind;     // CRT_GEOM_SPHERE, CRT_GEOM_QUAD or CRT_HOT_QUAD_X/Y/Z
    uint32_t mat_type; // CRT_MAT_*
    float o[3];        // sphere centre (Sphere.hpp:43) / quad corner v0 (Quad.hpp:202)
    float n[3];        // quad: nor = (v1-v0) x (v3-v0) (Quad.hpp:205-220); sphere: {radius^2, 0, 0} (Sphere.hpp:56)
};

// What the tree walk's closest-hit loop reads of a primitive, in one 16-dword scalar load: the hot record plus
// the centre and half-widths that classify a plane hit (DevGeomBox::mid, half_out_t, half_in_t).
struct alignas(64) DevGeomWalk {
    uint32_t kind_mat; // DevGeomHot::kind | mat_type << 8
    float o[3], n[3];  // DevGeomHot::o, n
    float mid[3], half_out_t[3], half_in_t[3];
};

struct alignas(16) DevGeomUV { // quad only (Quad.hpp:232-250): everything that depends only on the vertices
    uint32_t idu, idv;         // the two kept axes (lut {1,2,0,1})
    uint32_t linear;           // |k2| < 1e-4 -> linear branch (Quad.hpp:255)
    float nor_id;              // nor[id]
    float ka[2], kb[2], kg[2]; // projected a, b and kc - kb - ka
    float k2;                  // cross2d(kc - kb, ka)
    float pad;
};

// Bounding box of a parallelogram quad, used only to reject: a plane hit whose point lies outside
// [lo, hi] (already widened by 1e-3 of the scene size) on an axis the reference keeps for its (u,v)
// solve (Quad.hpp:232-250) cannot yield 0 <= u,v <= 1.  The dropped axis and every axis of a sphere or
// of a non-parallelogram quad carry -inf/+inf, which never reject.
struct alignas(64) DevGeomBox {
    float lo[3], hi[3];
    float n2;      // |nor|^2
    float sp_min2; // (1e-4 * |nor| * scene size)^2: a point this far off the plane is robustly off it
    // For an axis-aligned rectangle the same box shrunk by the margin lies inside the quad: a plane
    // hit inside [lo_in, hi_in] on the kept axes certainly has 0 < u,v < 1.  lo_in = +inf otherwise.
    float lo_in[3], hi_in[3];
    // Bound on the absolute error of nor.(P - v0) and of nor.D (D a light-sample offset) in binary32, on either
    // side of a comparison: 3e-6 * |nor| * scene size.  +inf when the quad is degenerate (nothing is decided).
    float noise;
    float pad;
    // The same boxes as centre and half-widths: |x - mid| > half_out rejects, |x - mid| < half_in is
    // inside the rectangle.  Unused axes: mid 0, half-widths +inf; no inner box: half_in -1.
    float mid[3], half_out[3], half_in[3];
    // The same two tests with a margin of 3e-5 of the scene size instead of 1e-3 (still 30x the rounding of a hit
    // point and of the (u,v) solve): the closest-hit loop of the tree walk decides most plane hits with them and
    // leaves the exact (u,v) test to the narrow band in between.
    float half_out_t[3], half_in_t[3];
    float pad2;
};

struct alignas(16) DevGeomShade {
    float n[4][3];       // quad vertex normals (Quad.hpp:310-325)
    float ambient[4];    // Material::ambient
    float diffuse[4];    // Material::diffuse
    float specular[4];   // Material::specular
    float refract_index; // Material::refractIndex
    float shininess;
    uint32_t ltc_col;    // texel column of (roughness*LUT_SCALE+LUT_BIAS): AreaLight.hpp:608,619,625 + Texture.hpp:179-182
    uint32_t pad;
};

struct alignas(64) DevLight {
    uint32_t kind;
    uint32_t two_sided;
    uint32_t shadow_samples;
    uint32_t initialized; // AreaLight::_initialized (AreaLight.hpp:586)
    float pos[3];
    float intensity;
    float color[3];
    float pad0;
    float pts[4][3];       // pos + points[i] (AreaLight.hpp:595-602)
    float p0[3], p2[3];    // points[0], points[2] (relative corners)
    float e01[3], e03[3];  // points[1]-points[0], points[3]-points[0] (AreaLight.hpp:359)
    float e23[3], e21[3];  // points[3]-points[2], points[1]-points[2] (AreaLight.hpp:370)
    float light_normal[3]; // (pts[1]-pts[0]) x (pts[3]-pts[0]) (AreaLight.hpp:250)
    // Shadow-ray filter only (approximate by design, see crt_wavefront.hip): the sample point is
    // fB0 + e01 u + e03 v when u + v < 1 and fB1 - e23 u - e21 v otherwise; for a parallelogram the
    // two agree and `is_par` is set.
    float fB0[3], fB1[3];
    uint32_t is_par;
    float fg[3]; // e01.e01, 2 e01.e03, e03.e03
    // _hittedLight (AreaLight.hpp:382-432), same derivation as a quad
    float nor[3];
    DevGeomUV uv;
    // Primitives (bit g of dark_hi:dark_lo) on which this light leaves nothing but the material's ambient term
    // wherever the shaded point is not "behind" it: a one-sided area light whose LTC sums are then set to zero
    // (AreaLight.hpp:292-296), with every factor of the zeroed terms finite and the ambient product free of
    // negative zeros and NaNs, so that ambient + ((0 + 0) * intensity) * visibility is the ambient product bit
    // for bit.  Zero for every other light.
    uint32_t dark_lo, dark_hi;
    uint32_t horizon_clip; // crt_light::horizonClip: the clipped branch of _evaluateLTC (AreaLight.hpp:298-334)
    uint32_t pad1[13];
};

struct alignas(64) DevCamera {
    float pos[3], front[3], right[3], up[3]; // Camera.hpp:97-98
    float one_over_tan;                      // 1 / tan(fov/2) (Camera.hpp:133)
    float ortho_scale;
    uint32_t type;
    uint32_t pad;
};

struct alignas(64) DevScene {
    DevCamera cam;
    uint32_t n_geom;
    uint32_t n_lights;
    uint32_t dark_possible; // 1: every light is an area light that is uninitialised or has a non-empty dark set
    uint32_t pad[13];
    DevGeomHot hot[CRT_MAX_GEOMETRY];
    DevGeomUV uv[CRT_MAX_GEOMETRY];
    DevGeomShade shade[CRT_MAX_GEOMETRY];
    DevGeomBox box[CRT_MAX_GEOMETRY];
    DevLight lights[CRT_MAX_LIGHTS];
    // Shadow-ray culling of quad `g` for area light `l`, decided once per shaded point P: with
    // s = nor . (P - v0), the quad cannot stop any shadow ray from P towards that light when
    // s * cull_side[l][g] > cull_thr[l][g] (the whole light lies on that side of the quad's plane,
    // further than the 0.01 the ray overshoots, and P is on the same side or on the plane itself).
    // cull_thr = +inf where no such statement holds (spheres, quads the light straddles or touches).
    // For a sphere g: the primitives that come within its ball (itself included).  A ray that starts
    // inside the ball leaves it after at most 2r, so nothing outside this set can be its closest hit.
    // A quad is judged by the hull of its vertices only when it is a true parallelogram; any other quad
    // is always in the set, because the reference's (u,v) solve can accept points outside that hull.
    // All ones for quads.
    uint32_t interior_lo[CRT_MAX_GEOMETRY], interior_hi[CRT_MAX_GEOMETRY];
    // 1 when quad g is parallel to the plane of area light l (nor.D is then the same for every sample)
    uint32_t quad_par[CRT_MAX_LIGHTS][CRT_MAX_GEOMETRY];
    float cull_side[CRT_MAX_LIGHTS][CRT_MAX_GEOMETRY];
    float cull_thr[CRT_MAX_LIGHTS][CRT_MAX_GEOMETRY];
    DevGeomWalk walk[CRT_MAX_GEOMETRY];
};

// Returns CRT_OK or CRT_ERR_INVALID (message in err, if non-null).
int flatten_scene(const crt_scene_desc* in, DevScene* out, char* err, size_t errcap);

} // namespace crt
