#!/usr/bin/env python3
"""Freezes whole frames of the CPU restatement (CASES below; first of all the benchmark workload, 720x720
gl_iter=1) as checksums, so that the GPU tests can compare every pixel and final generator state bit for
bit without paying minutes of CPU render in every run.

    python tests/golden/make_oracle_frames.py [case ...]   # updates tests/golden/oracle_full_frames.json

Data only: SHA-256 of the float32 RGBA image and of the uint32 generator states, whole and per band of 16
rows (a mismatch then names the band), plus the work counters.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import oracle_lib as O  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

from frame_digest import BAND, digest  # noqa: E402


CASES = {  # name: (image side or (nx, ny), aa_iter, gl_iter, shadowSamples); ref_iter 4, ind_rays 75, seed 1234
    "cornell_720_gl1": (720, 1, 1, 50),     # the benchmark workload, three minutes on 8 cores
    "cornell_128_gl2": (128, 1, 2, 50),     # two GI levels: ~22 000 shaded points per pixel, many pipeline rounds
    "cornell_256_aa3_gl1": (256, 3, 1, 50), # adaptive anti-aliasing three levels deep on top of one GI level
    # BASELINE's last configuration in miniature: aa_iter 4 over gl_iter 3 with ind_rays 75, i.e. the 128-entry
    # stack truncation three levels deep under nested anti-aliasing tasks; 5e7 shaded points per pixel, so 36
    # pixels and 4 shadow samples keep it at ten minutes on 8 cores
    "cornell_6_aa4_gl3": (6, 4, 3, 4),
    # the same configuration on the image's middle column of eight pixels (1 x 8: floor, both spheres' edges, back
    # wall, light, ceiling).  How long the GPU takes for such a frame is set by the pixels one wave walks, not by
    # their number: here every wave of the cooperative trace kernel owns a single pixel, which makes this the
    # form of the miniature that fits the default GPU test run
    "cornell_1x8_aa4_gl3": ((1, 8), 4, 3, 4),
    "cornell_1x8_aa4_gl3_s50": ((1, 8), 4, 3, 50),  # and with the configuration's own 50 shadow samples
    # the two halves of that configuration at sizes every trace mode finishes in seconds (a lane walks a
    # pixel's tree step by step: 6.5e5 and 1.9e6 steps per pixel here, 5.5e7 above)
    "cornell_12_gl3": (12, 1, 3, 4),       # ind_rays 75 three levels deep: 54 children at level 1, then 1..21 at level 2
    "cornell_4_aa4_gl2": (4, 4, 2, 4),     # 340 color() calls per pixel over the truncated two-level tree
}


def main():
    path = HERE / "oracle_full_frames.json"
    out = json.loads(path.read_text()) if path.exists() else {}
    for name in (sys.argv[1:] or list(CASES)):
        n, aa, gl, S = CASES[name]
        nx, ny = n if isinstance(n, tuple) else (n, n)
        st = crt.make_settings(nx=nx, ny=ny, aa_iter=aa, gl_iter=gl, shadowSamples=S)
        t0 = time.perf_counter()
        fb, cnt, state = O.render(O.cornell(S), st, 1234, want_state=True)
        d = digest(fb, state)
        d.update({"nx": nx, "ny": ny, "aa_iter": aa, "gl_iter": gl, "shadowSamples": S, "seed": 1234, "band_rows": BAND,
                  "rng_draws": int(cnt.rng_draws), "shade_points": int(cnt.shade_points),
                  "gi_children": int(cnt.gi_children),
                  "mean_rgba": [float(x) for x in fb.reshape(-1, 4).astype(np.float64).mean(0)]})
        out[name] = d
        print(f"{name}: {time.perf_counter() - t0:.1f} s", d["image_sha256"], flush=True)
        path.write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
