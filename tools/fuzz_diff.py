"""Where a fuzz scene differs from the oracle, per pipeline: python tools/fuzz_diff.py SEED [wild]."""
import sys
from pathlib import Path
import numpy as np
root = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(root), str(root / "tests")]
import cudaraytracer_amd as crt
from cudaraytracer_amd import abi
import oracle_lib
from scene_fuzz import random_scene
seed = int(sys.argv[1])
wild = len(sys.argv) > 2 and sys.argv[2] == "wild"
S = 3 + seed % 7
gl = 2 if seed % 11 == 10 else 1
st = crt.make_settings(nx=36, ny=28, aa_iter=(1, 1, 1, 1, 2, 1, 3, 1, 0, 4)[seed % 10], gl_iter=gl,
                       ind_rays=5 if gl == 1 else 3, shadowSamples=S)
ref, cnt, ref_state = oracle_lib.render(random_scene(oracle_lib, seed, S, wild), st, seed, want_state=True)
for name, pipe, mode, limit in (("persistent", abi.CRT_PIPELINE_PERSISTENT, 0, None), ("per-pixel", abi.CRT_PIPELINE_WAVEFRONT, 1, None),
                                ("per-pixel 1MiB", abi.CRT_PIPELINE_WAVEFRONT, 1, 1 << 20), ("cooperative", abi.CRT_PIPELINE_WAVEFRONT, 2, None)):
    ctx = crt.Context(0)
    ctx.set_pipeline(pipe)
    ctx.set_trace_mode(mode)
    if limit:
        ctx.set_workspace_limit(limit)
    ctx.upload(random_scene(crt, seed, S, wild))
    ctx.rng_init(st.nx, st.ny, seed)
    fb = ctx.render_to_host(st)
    state = ctx.rng_export(st.nx, st.ny)
    ctx.close()
    bad = np.argwhere((fb != ref).any(-1))
    print(f"{name}: {len(bad)} pixels differ, states equal {np.array_equal(state, ref_state)}")
    for r, c in bad[:6]:
        print("   pixel", r, c, "gpu", fb[r, c], "oracle", ref[r, c])
