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
wild = len(sys.argv) > 2 and sys.argv[2] in ("wild", "wide")
wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
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
sc0 = random_scene(crt, seed, S, wild)
print(f"{st.nx}x{st.ny} aa {st.aa_iter} ref {st.ref_iter} gl {st.gl_iter} ind {st.ind_rays} S {S}; lights:",
      [int(l.kind) for l in sc0.lights], "point kind =", int(abi.CRT_LIGHT_POINT))
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
    print(f"{name}: {len(bad)} pixels differ, states equal {np.array_equal(state, ref_state)}, max abs {np.abs(fb - ref).max():.3g}, "
          f"max rel {(np.abs(fb - ref) / np.maximum(np.abs(ref), 1e-30)).max():.3g}")
    for r, c in bad[:6]:
        print("   pixel", r, c, "gpu", fb[r, c], "oracle", ref[r, c])
