"""Python host binding of libcrt_hip.so (the C ABI declared in include/crt_hip.h).

The library holds hand-written HIP kernels for gfx950 that implement the
per-pixel render path of CudaRayTracer (render -> aa_color -> color ->
GeometryList::hit / AreaLight::calculateColor, main.cu:110-347).  This module is
a thin ctypes layer for tests, the benchmark and Python callers; C++ callers use
cudaraytracer_amd/include/craytracer/craytracer.hpp.

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from . import abi
from .abi import make_settings  # noqa: F401  (re-export)

_PKG = Path(__file__).resolve().parent
# CRT_HIP_LIB selects another build of the same library (tools/variant_bench.sh compares kernel variants)
LIB_PATH = Path(__import__("os").environ.get("CRT_HIP_LIB", _PKG / "libcrt_hip.so"))

_lib = None


class CrtError(RuntimeError):
    def __init__(self, code, what, message=""):
        super().__init__(f"{what} failed with status {code}: {message}")
        self.code = code


def lib():
    """Loads libcrt_hip.so (built in-tree by `make -C cudaraytracer_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} not found: build it with `make -C {_PKG / 'csrc'}` (hipcc, --offload-arch=gfx950)")
        L = C.CDLL(str(LIB_PATH))
        P = C.POINTER
        sig = {
            "crt_create": (C.c_int, [P(C.c_void_p), C.c_int]),
            "crt_destroy": (None, [C.c_void_p]),
            "crt_last_error": (C.c_char_p, [C.c_void_p]),
            "crt_cornell_scene": (C.c_int, [C.c_uint32, P(abi.crt_camera), P(abi.crt_geometry), P(abi.crt_light)]),
            "crt_scene_upload": (C.c_int, [C.c_void_p, P(abi.crt_scene_desc)]),
            "crt_rng_init": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
            "crt_rng_export": (C.c_int, [C.c_void_p, C.c_void_p]),
            "crt_render": (C.c_int, [C.c_void_p, P(abi.crt_settings), C.c_void_p, C.c_void_p]),
            "crt_render_static": (C.c_int, [C.c_void_p, P(abi.crt_settings), C.c_void_p, C.c_void_p]),
            "crt_render_partial": (C.c_int, [C.c_void_p, P(abi.crt_settings), C.c_uint32, C.c_void_p, C.c_void_p]),
            "crt_render_shard": (C.c_int, [C.c_void_p, P(abi.crt_settings), C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p]),
            "crt_shard_tiles": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
            "crt_shard_tiles_max": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
            "crt_scatter_shards": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
            "crt_stats_since_init": (C.c_int, [C.c_void_p, P(abi.crt_settings), P(abi.crt_stats)]),
            "crt_selftest_rcp": (C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint32)]),
            "crt_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
            "crt_set_trace_mode": (C.c_int, [C.c_void_p, C.c_int]),
            "crt_set_workspace_limit": (C.c_int, [C.c_void_p, C.c_size_t]),
            "crt_set_round_limit": (C.c_int, [C.c_void_p, C.c_uint32]),
            "crt_reserve_workspace": (C.c_int, [C.c_void_p, P(abi.crt_settings), C.c_uint32, C.c_uint32]),
            "crt_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
            "crt_get_kernel_times": (C.c_int, [C.c_void_p, P(C.c_double), P(C.c_uint64)]),
            "crt_set_audit": (C.c_int, [C.c_void_p, C.c_int]),
            "crt_audit_result": (C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]),
            "crt_fb_alloc": (C.c_int, [C.c_void_p, C.c_size_t, P(C.c_void_p)]),
            "crt_fb_free": (C.c_int, [C.c_void_p, C.c_void_p]),
            "crt_fb_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
            "crt_write_hdr": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
            "crt_encode_hdr": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
            "crt_to_rgba8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
            "crt_write_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
        }
        variant = "CRT_HIP_LIB" in __import__("os").environ  # an older build measured for comparison may lack entry points
        for name, (res, args) in sig.items():
            if variant and not hasattr(L, name):
                continue
            fn = getattr(L, name)  # AttributeError here = the library does not export the declared ABI
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


ABI_SYMBOLS = (
    "crt_create", "crt_destroy", "crt_last_error", "crt_cornell_scene", "crt_scene_upload", "crt_rng_init",
    "crt_rng_export", "crt_render", "crt_render_static", "crt_render_partial", "crt_render_shard", "crt_shard_tiles",
    "crt_shard_tiles_max", "crt_scatter_shards", "crt_stats_since_init", "crt_selftest_rcp", "crt_set_pipeline",
    "crt_set_trace_mode",
    "crt_set_workspace_limit", "crt_set_round_limit", "crt_reserve_workspace", "crt_set_audit", "crt_audit_result", "crt_set_timing",
    "crt_get_kernel_times", "crt_fb_alloc", "crt_fb_free",
    "crt_fb_download", "crt_write_hdr", "crt_encode_hdr", "crt_to_rgba8", "crt_write_ppm")


class Scene:
    """Owns the POD arrays a crt_scene_desc points to."""

    def __init__(self, camera, geometry, lights):
        self.camera = camera
        self.geometry = (abi.crt_geometry * max(len(geometry), 1))(*geometry)
        self.lights = (abi.crt_light * max(len(lights), 1))(*lights)
        self.desc = abi.crt_scene_desc(camera, len(geometry), self.geometry, len(lights), self.lights)


def cornell_box(shadow_samples=50):
    """The scene of create_world (main.cu:373-593)."""
    cam = abi.crt_camera()
    geo = (abi.crt_geometry * 8)()
    light = (abi.crt_light * 1)()
    rc = lib().crt_cornell_scene(shadow_samples, C.byref(cam), geo, light)
    if rc != abi.CRT_OK:
        raise CrtError(rc, "crt_cornell_scene")
    return Scene(cam, list(geo), list(light))


class Context:
    """One GPU, one scene, one set of per-pixel RNG states."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().crt_create(C.byref(self._h), device)
        if rc != abi.CRT_OK:
            raise CrtError(rc, "crt_create", "no usable HIP device" if rc == abi.CRT_ERR_HIP else "")
        self.device = device

    def close(self):
        if self._h:
            lib().crt_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != abi.CRT_OK:
            raise CrtError(rc, what, lib().crt_last_error(self._h).decode(errors="replace"))

    # -- scene / rng
    def upload(self, scene):
        self._check(lib().crt_scene_upload(self._h, C.byref(scene.desc)), "crt_scene_upload")

    def rng_init(self, nx, ny, seed, stream=None):
        self._check(lib().crt_rng_init(self._h, nx, ny, seed, stream), "crt_rng_init")

    def rng_export(self, nx, ny):
        st = np.empty((ny * nx, 6), dtype=np.uint32)
        self._check(lib().crt_rng_export(self._h, st.ctypes.data), "crt_rng_export")
        return st

    # -- render into caller-owned device memory (an integer device address)
    def render(self, settings, fb_ptr, stream=None):
        self._check(lib().crt_render(self._h, C.byref(settings), fb_ptr, stream), "crt_render")

    def render_static(self, settings, fb_ptr, stream=None):
        """crt_render without any device allocation (persistent kernel, scratch in a static device array)."""
        self._check(lib().crt_render_static(self._h, C.byref(settings), fb_ptr, stream), "crt_render_static")

    def render_partial(self, settings, tile_index, fb_ptr, stream=None):
        self._check(lib().crt_render_partial(self._h, C.byref(settings), tile_index, fb_ptr, stream),
                    "crt_render_partial")

    def render_shard(self, settings, rank, world, compact_ptr, stream=None):
        self._check(lib().crt_render_shard(self._h, C.byref(settings), rank, world, compact_ptr, stream),
                    "crt_render_shard")

    def scatter_shards(self, nx, ny, world, gathered_ptr, fb_ptr, stream=None):
        self._check(lib().crt_scatter_shards(self._h, nx, ny, world, gathered_ptr, fb_ptr, stream),
                    "crt_scatter_shards")

    def stats(self, settings):
        s = abi.crt_stats()
        self._check(lib().crt_stats_since_init(self._h, C.byref(settings), C.byref(s)), "crt_stats_since_init")
        return s

    def set_pipeline(self, pipeline):
        """abi.CRT_PIPELINE_AUTO / _PERSISTENT / _WAVEFRONT."""
        self._check(lib().crt_set_pipeline(self._h, pipeline), "crt_set_pipeline")

    def set_trace_mode(self, mode):
        """abi.CRT_TRACE_AUTO / _PER_PIXEL / _COOPERATIVE (wavefront pipeline, gl_iter >= 1)."""
        self._check(lib().crt_set_trace_mode(self._h, mode), "crt_set_trace_mode")

    def reserve_workspace(self, settings, rank=0, world=1):
        self._check(lib().crt_reserve_workspace(self._h, C.byref(settings), rank, world), "crt_reserve_workspace")

    def set_workspace_limit(self, nbytes):
        self._check(lib().crt_set_workspace_limit(self._h, nbytes), "crt_set_workspace_limit")

    def set_round_limit(self, rounds):
        """Measuring aid: wavefront frames stop after `rounds` rounds (0 = no limit); a frame cut short raises
        CrtError with code abi.CRT_ERR_TRUNCATED."""
        self._check(lib().crt_set_round_limit(self._h, rounds), "crt_set_round_limit")

    KERNELS = ("trace", "filter", "resolve", "shade", "sum")

    def set_timing(self, enabled):
        self._check(lib().crt_set_timing(self._h, int(bool(enabled))), "crt_set_timing")

    def kernel_times(self):
        """{kernel: (milliseconds, launches)} of the wavefront pipeline since set_timing(True)."""
        ms = (C.c_double * 5)()
        n = (C.c_uint64 * 5)()
        self._check(lib().crt_get_kernel_times(self._h, ms, n), "crt_get_kernel_times")
        return {k: (ms[i], n[i]) for i, k in enumerate(self.KERNELS)}

    def set_audit(self, enabled):
        self._check(lib().crt_set_audit(self._h, int(bool(enabled))), "crt_set_audit")

    def audit_result(self):
        """(rays decided by the conservative filter and re-checked exactly, contradictions)."""
        n, bad = C.c_uint64(), C.c_uint64()
        self._check(lib().crt_audit_result(self._h, C.byref(n), C.byref(bad)), "crt_audit_result")
        return n.value, bad.value

    def selftest_rcp(self):
        """(mismatches, first mismatching bit pattern) of the fast reciprocal vs 1.0f/x over all floats."""
        bad, first = C.c_uint64(), C.c_uint32()
        self._check(lib().crt_selftest_rcp(self._h, C.byref(bad), C.byref(first)), "crt_selftest_rcp")
        return bad.value, first.value

    # -- library-owned device buffers (for hosts without torch)
    def fb_alloc(self, n_floats):
        p = C.c_void_p()
        self._check(lib().crt_fb_alloc(self._h, n_floats, C.byref(p)), "crt_fb_alloc")
        return p.value

    def fb_free(self, ptr):
        self._check(lib().crt_fb_free(self._h, ptr), "crt_fb_free")

    def fb_download(self, ptr, n_floats, stream=None):
        host = np.empty(n_floats, dtype=np.float32)
        self._check(lib().crt_fb_download(self._h, ptr, host.ctypes.data, n_floats, stream), "crt_fb_download")
        return host

    # -- convenience: full image to host
    def render_to_host(self, settings, progressive=False):
        nx, ny = settings.nx, settings.ny
        n = nx * ny * 4
        ptr = self.fb_alloc(n)
        try:
            if not progressive:
                self.render(settings, ptr)
            else:
                bx, by = nx // settings.tx + 1, ny // settings.ty + 1
                for tile in range(0, bx * by, settings.blocksPerDraw):
                    self.render_partial(settings, tile, ptr)
            return self.fb_download(ptr, n).reshape(ny, nx, 4)
        finally:
            self.fb_free(ptr)


def shard_tiles(nx, ny, rank, world):
    return lib().crt_shard_tiles(nx, ny, rank, world)


def shard_tiles_max(nx, ny, world):
    return lib().crt_shard_tiles_max(nx, ny, world)


def encode_hdr(image):
    """Radiance RGBE bytes of a float32 image [h, w, comp] (stbi_write_hdr format, main.cu:856)."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w, comp = img.shape
    n = lib().crt_encode_hdr(w, h, comp, img.ctypes.data, None, 0)
    if n < 0:
        raise CrtError(n, "crt_encode_hdr")
    buf = np.empty(n, dtype=np.uint8)
    n2 = lib().crt_encode_hdr(w, h, comp, img.ctypes.data, buf.ctypes.data, n)
    assert n2 == n
    return buf.tobytes()


def write_hdr(path, image):
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w, comp = img.shape
    rc = lib().crt_write_hdr(str(path).encode(), w, h, comp, img.ctypes.data)
    if rc != abi.CRT_OK:
        raise CrtError(rc, "crt_write_hdr")


def to_rgba8(image):
    """uint8 preview of a float image: (uint8)clamp(c * 255, 0, 255) per value (main.cu:816-818)."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    out = np.empty(img.shape, dtype=np.uint8)
    rc = lib().crt_to_rgba8(img.ctypes.data, img.size, out.ctypes.data)
    if rc != abi.CRT_OK:
        raise CrtError(rc, "crt_to_rgba8")
    return out


def write_ppm(path, rgba8):
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w, comp = img.shape
    if comp != 4:
        raise ValueError("write_ppm expects an RGBA8 image [h, w, 4]")
    rc = lib().crt_write_ppm(str(path).encode(), w, h, img.ctypes.data)
    if rc != abi.CRT_OK:
        raise CrtError(rc, "crt_write_ppm")
