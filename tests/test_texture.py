"""craytracer::Texture (the host mirror of the reference's Texture.hpp: nearest texel, CLAMP / MIRROR / REPEAT) against
the CPU restatement's _imageSpaceCoordinates, and against the wrap rules written out by hand."""
import ctypes as C
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
MODES = {"CLAMP": 0, "MIRROR": 1, "REPEAT": 2}

PROGRAM = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <craytracer/Texture.hpp>
using namespace craytracer;
int main(int argc, char** argv) {
    const unsigned w = 7, h = 5;
    std::vector<float> data(w * h * 2);                 // two channels: texel (x, y) holds (x, y)
    for (unsigned y = 0; y < h; ++y)
        for (unsigned x = 0; x < w; ++x) { data[(y * w + x) * 2] = (float)x; data[(y * w + x) * 2 + 1] = (float)y; }
    const WRAP_MODE modes[3] = {WRAP_MODE::CLAMP, WRAP_MODE::MIRROR, WRAP_MODE::REPEAT};
    for (int mh = 0; mh < 3; ++mh)
        for (int mv = 0; mv < 3; ++mv) {
            Texture t(data.data(), w, h, 2, modes[mv], modes[mh]);
            for (int i = 1; i + 1 < argc; i += 2) {
                const vec4 c = t.sample(vec2((float)atof(argv[i]), (float)atof(argv[i + 1])));
                std::printf("%d %d %s %s %g %g %g %g\n", mh, mv, argv[i], argv[i + 1], c.x(), c.y(), c.z(), c.w());
            }
        }
    Texture grey(data.data(), 3, 1, 1);                  // one channel: grey with alpha 1; data[2] is texel (1, 0)'s x = 1
    const vec4 g = grey.getPixelColor(2u, 0u);
    std::printf("grey %g %g %g %g %d %d\n", g.x(), g.y(), g.z(), g.w(), (int)grey.getVerticalWrapMode(), (int)Texture().isImageLoaded());
    const vec4 out = grey.getPixelColor(3u, 0u);         // outside: zero vector (Texture.hpp:230)
    std::printf("outside %g %g %g %g\n", out.x(), out.y(), out.z(), out.w());
    return 0;
}
"""


def _expected(f, mode, size):
    f = np.float32(f)
    if f < 0 or f > 1:
        if mode == 0:
            f = np.float32(min(max(f, np.float32(0)), np.float32(1)))
        else:
            fr = np.float32(f - np.float32(math.floor(f)))
            f = fr if mode == 2 else np.float32(abs(np.float32(fr - np.float32(1))))
    return int(np.float32(f * np.float32(size - 1)))


def test_texture_sampling_follows_the_reference_wrap_rules(oracle, tmp_path):
    src = tmp_path / "tex.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "tex"
    subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-I", str(ROOT / "cudaraytracer_amd" / "include"),
                    "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    coords = [-2.25, -1.0, -0.3, -0.0001, 0.0, 0.2, 0.5, 0.999, 1.0, 1.0001, 1.4, 2.0, 3.75]
    uvs = [(u, v) for u in coords for v in (-0.6, 0.35, 1.7)] + [(0.5, v) for v in coords]
    args = [repr(x) for uv in uvs for x in uv]
    lines = subprocess.run([str(exe), *args], check=True, capture_output=True, text=True).stdout.splitlines()
    L = oracle.lib()
    n = 0
    for line in lines:
        parts = line.split()
        if parts[0] in ("grey", "outside"):
            continue
        mh, mv, u, v = int(parts[0]), int(parts[1]), float(parts[2]), float(parts[3])
        got = (float(parts[4]), float(parts[5]))
        xy = (C.c_uint32 * 2)()
        L.orc_texture_texel(u, v, mh, mv, 7, 5, C.byref(xy))
        assert got == (float(xy[0]), float(xy[1])), line
        assert got == (float(_expected(u, mh, 7)), float(_expected(v, mv, 5))), line
        assert float(parts[6]) == 0.0 and float(parts[7]) == 1.0     # two channels: blue 0, alpha 1
        n += 1
    assert n == 9 * len(uvs)
    assert "grey 1 1 1 1 2 0" in lines                               # REPEAT is the default wrap mode
    assert "outside 0 0 0 0" in lines
