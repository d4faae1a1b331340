#!/usr/bin/env python3
"""Freezes the CPU restatement's full-size gl_iter=1 render (720x720, seed 1234, the benchmark workload)
as checksums, so that the GPU test can compare all 518400 pixels and final generator states bit for bit
without paying the CPU render (about two minutes on 8 cores) in every run.

    python tests/golden/make_oracle_frames.py        # writes tests/golden/oracle_full_frames.json

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


def main():
    out = {}
    for gl in (1,):
        st = crt.make_settings(gl_iter=gl)
        t0 = time.perf_counter()
        fb, cnt, state = O.render(O.cornell(50), st, 1234, want_state=True)
        d = digest(fb, state)
        d.update({"nx": int(st.nx), "ny": int(st.ny), "seed": 1234, "band_rows": BAND,
                  "rng_draws": int(cnt.rng_draws), "shade_points": int(cnt.shade_points),
                  "gi_children": int(cnt.gi_children),
                  "mean_rgba": [float(x) for x in fb.reshape(-1, 4).astype(np.float64).mean(0)]})
        out[f"cornell_720_gl{gl}"] = d
        print(f"gl_iter={gl}: {time.perf_counter() - t0:.1f} s", d["image_sha256"], flush=True)
    (HERE / "oracle_full_frames.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
