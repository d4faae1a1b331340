"""ctypes loader for the CPU restatement under oracle/ (test infrastructure)."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from cudaraytracer_amd import abi

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LTC_PATH = ROOT / "cudaraytracer_amd" / "data" / "ltc_tables.f32"


class orc_counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("shade_points", C.c_uint64), ("gi_children", C.c_uint64),
                ("rng_draws", C.c_uint64)]


class orc_hit_record(C.Structure):
    _fields_ = [("hit", C.c_int), ("geom", C.c_int), ("t", C.c_float), ("p", C.c_float * 3),
                ("n", C.c_float * 3)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib():
    global _lib
    if _lib is None:
        so = ORACLE_DIR / "libcrt_oracle.so"
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        L.orc_rng_init.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
        L.orc_rng_next.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rng_next.restype = C.c_uint32
        L.orc_rng_uniform.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_rng_uniform.restype = C.c_float
        L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_cornell_scene.argtypes = [C.c_uint32, C.POINTER(abi.crt_camera), C.POINTER(abi.crt_geometry),
                                        C.POINTER(abi.crt_light)]
        L.orc_render.argtypes = [C.POINTER(abi.crt_scene_desc), C.POINTER(abi.crt_settings), C.c_uint64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_int, C.POINTER(orc_counters)]
        L.orc_render.restype = C.c_int
        L.orc_camera_ray.argtypes = [C.POINTER(abi.crt_camera), C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_world_hit.argtypes = [C.POINTER(abi.crt_scene_desc), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.c_float, C.POINTER(orc_hit_record)]
        L.orc_texture_texel.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32 * 2)]
        L.orc_ltc_clipped.argtypes = [C.POINTER((C.c_float * 3) * 4), C.c_int]
        L.orc_ltc_clipped.restype = C.c_float
        _lib = L
    return _lib


def ltc_tables():
    t = np.fromfile(LTC_PATH, dtype="<f4")
    assert t.size == 2 * 64 * 64 * 4
    return np.ascontiguousarray(t)


class Scene:
    """Owns the arrays a crt_scene_desc points to."""

    def __init__(self, camera, geometry, lights):
        self.camera = camera
        self.geometry = (abi.crt_geometry * len(geometry))(*geometry)
        self.lights = (abi.crt_light * len(lights))(*lights)
        self.desc = abi.crt_scene_desc(camera, len(geometry), self.geometry, len(lights), self.lights)


def cornell(shadow_samples=50):
    cam = abi.crt_camera()
    geo = (abi.crt_geometry * 8)()
    light = (abi.crt_light * 1)()
    lib().orc_cornell_scene(shadow_samples, C.byref(cam), geo, light)
    return Scene(cam, list(geo), list(light))


def initial_state(nx, ny, seed):
    """Per-pixel states as random_init leaves them: curand_init(seed + idx, 0, 0) (main.cu:317-325)."""
    st = np.zeros((ny * nx, 6), dtype=np.uint32)
    buf = (C.c_uint32 * 6)()
    L = lib()
    for idx in range(nx * ny):
        L.orc_rng_init(seed + idx, buf)
        st[idx] = buf[:]
    return st


def render(scene, settings, seed, region=None, rng_state=None, threads=0, want_state=False):
    """Returns (fb[ny,nx,4] float32, counters, rng_state or None)."""
    nx, ny = settings.nx, settings.ny
    fb = np.zeros((ny, nx, 4), dtype=np.float32)
    x0, y0, x1, y1 = region if region else (0, 0, nx, ny)
    ltc = ltc_tables()
    cnt = orc_counters()
    fresh = 1
    st = None
    if rng_state is not None:
        st = np.ascontiguousarray(rng_state, dtype=np.uint32).copy()
        fresh = 0
    elif want_state:
        st = np.zeros((ny * nx, 6), dtype=np.uint32)
    rc = lib().orc_render(C.byref(scene.desc), C.byref(settings), seed, ltc.ctypes.data, fb.ctypes.data,
                          st.ctypes.data if st is not None else None, fresh, x0, y0, x1, y1,
                          threads or (os.cpu_count() or 1), C.byref(cnt))
    assert rc == 0
    return fb, cnt, st
