"""Long form of the random-scene parity test: python tools/fuzz_gpu.py FIRST COUNT [wild|wide] [pieces] (needs a GPU).
With `pieces` every launch of two work units or more is rendered as two pieces side by side (the library does that
from 1 000 units on) and the audit, which such launches do not run in pieces, stays off."""
import os
import sys
from pathlib import Path

import numpy as np

root = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(root), str(root / "tests")]
import cudaraytracer_amd as crt  # noqa: E402
from cudaraytracer_amd import abi  # noqa: E402
import oracle_lib  # noqa: E402
from scene_fuzz import random_scene  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
wild = len(sys.argv) > 3 and sys.argv[3] in ("wild", "wide")
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
pieces = "pieces" in sys.argv[3:]
if pieces:
    os.environ["CRT_FRAME_PIECES_MIN_UNITS"] = "2"  # read by crt_create
failures = 0
for seed in range(first, first + count):
    S = 3 + seed % 7
    gl = 2 if seed % 11 == 10 else 1
    if wide:  # settings vary too: children beyond one batch and beyond the ray stack, few or many bounces, odd image sizes
        prng = np.random.default_rng(seed + 13_000_000)
        gl = int(prng.choice([1, 1, 1, 2, 3]))
        ind = int(prng.choice([2, 5, 9, 70, 130])) if gl == 1 else int(prng.choice([2, 3, 9 if gl == 2 else 4]))
        st = crt.make_settings(nx=int(prng.integers(5, 40)), ny=int(prng.integers(5, 30)), aa_iter=int(prng.choice([0, 1, 1, 2, 3])),
                               ref_iter=int(prng.choice([1, 2, 4, 6])), gl_iter=gl, ind_rays=ind, shadowSamples=S)
    else:
        st = crt.make_settings(nx=36, ny=28, aa_iter=(1, 1, 1, 1, 2, 1, 3, 1, 0, 4)[seed % 10], gl_iter=gl,
                               ind_rays=5 if gl == 1 else 3, shadowSamples=S)
    scene = random_scene(crt, seed, S, wild)
    ctx = crt.Context(0)
    ctx.set_pipeline(abi.CRT_PIPELINE_PERSISTENT if seed % 8 == 5 else abi.CRT_PIPELINE_WAVEFRONT)
    ctx.set_audit(not pieces)
    ctx.set_trace_mode(abi.CRT_TRACE_COOPERATIVE if seed % 2 == 0 else abi.CRT_TRACE_PER_PIXEL)
    if seed % 7 == 3:
        ctx.set_workspace_limit(1 << 20)
    ctx.upload(scene)
    ctx.rng_init(st.nx, st.ny, seed)
    fb = ctx.render_to_host(st)
    state = ctx.rng_export(st.nx, st.ny)
    checked, bad = ctx.audit_result()
    ctx.close()
    ref, _, ref_state = oracle_lib.render(random_scene(oracle_lib, seed, S, wild), st, seed, want_state=True)
    same = np.array_equal(fb, ref)
    ok = bad == 0 and same and np.array_equal(state, ref_state)
    if not ok:
        failures += 1
        print(f"seed {seed}: FAIL bad={bad}/{checked} image={same} state={np.array_equal(state, ref_state)} "
              f"maxdiff={np.abs(fb - ref).max()}", flush=True)
    elif seed % 50 == 0:
        print(f"seed {seed}: ok ({checked} decisions audited)", flush=True)
print(f"{count} scenes, {failures} failures")
sys.exit(1 if failures else 0)
