"""One scene of tools/fuzz_gpu.py with the scene printed: python tools/fuzz_one.py SEED [wild]."""
import sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(root), str(root / "tests")]
import cudaraytracer_amd as crt
from cudaraytracer_amd import abi
from scene_fuzz import random_scene
seed = int(sys.argv[1])
wild = len(sys.argv) > 2 and sys.argv[2] == "wild"
S = 3 + seed % 7
gl = 2 if seed % 11 == 10 else 1
st = crt.make_settings(nx=36, ny=28, aa_iter=(1, 1, 1, 1, 2, 1, 3, 1, 0, 4)[seed % 10], gl_iter=gl,
                       ind_rays=5 if gl == 1 else 3, shadowSamples=S)
scene = random_scene(crt, seed, S, wild)
print("camera", list(scene.camera.pos), list(scene.camera.front), scene.camera.type)
for i, g in enumerate(scene.geometry):
    if g.kind == 0:
        print(i, "sphere", list(g.center), g.radius, "mat", g.mat.type, g.mat.refractIndex)
    else:
        print(i, "quad", [list(v.pos) for v in g.v], "mat", g.mat.type)
for l in scene.lights:
    print("light", l.kind, list(l.pos), [list(p) for p in l.points], l.twoSided, l.shadowSamples)
ctx = crt.Context(0)
ctx.set_pipeline(abi.CRT_PIPELINE_WAVEFRONT)
ctx.set_audit(True)
ctx.upload(scene)
ctx.rng_init(st.nx, st.ny, seed)
ctx.render_to_host(st)
print("audited, contradictions:", ctx.audit_result())
