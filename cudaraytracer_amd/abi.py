"""ctypes mirror of include/crt_hip.h (struct layouts and constants only)."""
import ctypes as C

CRT_OK = 0
CRT_PIPELINE_AUTO, CRT_PIPELINE_PERSISTENT, CRT_PIPELINE_WAVEFRONT = 0, 1, 2
CRT_TRACE_AUTO, CRT_TRACE_PER_PIXEL, CRT_TRACE_COOPERATIVE = 0, 1, 2
CRT_ERR_INVALID = -1
CRT_ERR_HIP = -2
CRT_ERR_NO_SCENE = -3
CRT_ERR_NO_RNG = -4
CRT_ERR_UNSUPPORTED = -5
CRT_ERR_IO = -6
CRT_ERR_TRUNCATED = -7

CRT_MAX_GEOMETRY = 64
CRT_MAX_LIGHTS = 8
CRT_MAX_GL_ITER = 8
CRT_MAX_AA_ITER = 4
CRT_RAY_STACK = 128

CRT_MAT_DIFFUSE, CRT_MAT_REFLECT, CRT_MAT_REFRACTIVE = 0, 1, 2
CRT_GEOM_SPHERE, CRT_GEOM_QUAD = 0, 1
CRT_LIGHT_POINT, CRT_LIGHT_AREA = 0, 1
CRT_CAM_ORTHOGRAPHIC, CRT_CAM_PERSPECTIVE = 0, 1


class crt_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("ambient", C.c_float * 4), ("diffuse", C.c_float * 4),
                ("specular", C.c_float * 4), ("shininess", C.c_float), ("refractIndex", C.c_float)]


class crt_vertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("tex", C.c_float * 2), ("normal", C.c_float * 3)]


class crt_geometry(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mat", crt_material), ("center", C.c_float * 3),
                ("radius", C.c_float), ("v", crt_vertex * 4)]


class crt_light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("pos", C.c_float * 3), ("color", C.c_float * 4),
                ("intensity", C.c_float), ("points", (C.c_float * 3) * 4), ("twoSided", C.c_uint32),
                ("shadowSamples", C.c_uint32), ("initialized", C.c_uint32), ("horizonClip", C.c_uint32)]


class crt_camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("fov", C.c_float),
                ("orthoScale", C.c_float), ("type", C.c_uint32)]


class crt_scene_desc(C.Structure):
    _fields_ = [("camera", crt_camera), ("n_geometry", C.c_uint32), ("geometry", C.POINTER(crt_geometry)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(crt_light))]


class crt_settings(C.Structure):
    _fields_ = [("renderAllAtOnce", C.c_uint32), ("blocksPerDraw", C.c_uint32), ("nx", C.c_uint32),
                ("ny", C.c_uint32), ("tx", C.c_uint32), ("ty", C.c_uint32), ("aa_iter", C.c_uint32),
                ("ref_iter", C.c_uint32), ("gl_iter", C.c_uint32), ("ind_rays", C.c_uint32),
                ("shadowSamples", C.c_uint32)]


class crt_stats(C.Structure):
    _fields_ = [("rng_draws", C.c_uint64), ("shadow_samples", C.c_uint64), ("shade_points", C.c_uint64),
                ("gi_children", C.c_uint64), ("exact_tests", C.c_uint64), ("glass_walks", C.c_uint64)]


def make_settings(nx=720, ny=720, aa_iter=1, ref_iter=4, gl_iter=2, ind_rays=75, shadowSamples=50,
                  tx=19, ty=19, renderAllAtOnce=True, blocksPerDraw=200):
    """Defaults are the constants of the reference's main() (main.cu:626-636)."""
    return crt_settings(int(bool(renderAllAtOnce)), blocksPerDraw, nx, ny, tx, ty, aa_iter, ref_iter,
                        gl_iter, ind_rays, shadowSamples)
