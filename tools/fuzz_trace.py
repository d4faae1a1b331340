"""Shadow counts of one pixel of a fuzz scene, oracle (stderr, ORC_TRACE=1) and GPU (a -DCRT_SHADE_TRACE build):
ORC_TRACE=1 CRT_HIP_LIB=... python tools/fuzz_trace.py SEED I J [wild|wide]"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(root), str(root / "tests")]
import cudaraytracer_amd as crt
from cudaraytracer_amd import abi
import oracle_lib
from scene_fuzz import random_scene
seed, i, j = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
import numpy as np
wild = len(sys.argv) > 4 and sys.argv[4] in ("wild", "wide")
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
S = 3 + seed % 7
gl = 2 if seed % 11 == 10 else 1
if wide:  # the settings tools/fuzz_gpu.py draws in its "wide" form
    prng = np.random.default_rng(seed + 13_000_000)
    gl = int(prng.choice([1, 1, 1, 2, 3]))
    ind = int(prng.choice([2, 5, 9, 70, 130])) if gl == 1 else int(prng.choice([2, 3, 9 if gl == 2 else 4]))
    st = crt.make_settings(nx=int(prng.integers(5, 40)), ny=int(prng.integers(5, 30)), aa_iter=int(prng.choice([0, 1, 1, 2, 3])),
                           ref_iter=int(prng.choice([1, 2, 4, 6])), gl_iter=gl, ind_rays=ind, shadowSamples=S)
else:
    st = crt.make_settings(nx=36, ny=28, aa_iter=(1, 1, 1, 1, 2, 1, 3, 1, 0, 4)[seed % 10], gl_iter=gl,
                           ind_rays=5 if gl == 1 else 3, shadowSamples=S)
oracle_lib.render(random_scene(oracle_lib, seed, S, wild), st, seed, region=(i, j, i + 1, j + 1), threads=1)
ctx = crt.Context(0)
ctx.set_pipeline(abi.CRT_PIPELINE_WAVEFRONT)
ctx.set_trace_mode(abi.CRT_TRACE_PER_PIXEL)  # the debug build prints from the per-pixel walk
ctx.set_audit(True)
ctx.upload(random_scene(crt, seed, S, wild))
ctx.rng_init(st.nx, st.ny, seed)
ctx.render_to_host(st)
ctx.close()
