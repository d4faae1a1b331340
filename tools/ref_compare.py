#!/usr/bin/env python3
"""Renders the README's 720x720 frames on cuda:0 and stores them as the reference would have written them
(crt_encode_hdr -> RGBE bytes) under gpurun_out/, for offline comparison with the reference's published
renders (tests/golden/reference_hdr_stats.json):  python tools/ref_compare.py <gl_iter> <seed> [<seed> ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

gl = int(sys.argv[1])
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
st = crt.make_settings(gl_iter=gl)
with crt.Context(0) as ctx:
    ctx.upload(crt.cornell_box(st.shadowSamples))
    for seed in [int(s) for s in sys.argv[2:]]:
        ctx.rng_init(st.nx, st.ny, seed)
        fb = ctx.render_to_host(st)
        (out / f"frame_gl{gl}_seed{seed}.hdr").write_bytes(crt.encode_hdr(fb))
        print("gl", gl, "seed", seed, "mean", fb.reshape(-1, 4).astype(np.float64).mean(0), flush=True)
