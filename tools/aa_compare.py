"""Times one 720x720 render with adaptive AA through both pipelines and checks that the images agree."""
import sys
import time

import numpy as np

import cudaraytracer_amd as crt
from cudaraytracer_amd import abi

aa, gl = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 720
st = crt.make_settings(nx=n, ny=n, aa_iter=aa, gl_iter=gl)
imgs = {}
for name, pipe in (("wavefront", abi.CRT_PIPELINE_WAVEFRONT), ("persistent", abi.CRT_PIPELINE_PERSISTENT)):
    ctx = crt.Context(0)
    ctx.set_pipeline(pipe)
    ctx.upload(crt.cornell_box(50))
    ptr = ctx.fb_alloc(n * n * 4)
    for rep in range(2):
        ctx.rng_init(n, n, 1234)
        ctx.fb_download(ptr, 4)  # drain
        t0 = time.perf_counter()
        ctx.render(st, ptr)
        imgs[name] = ctx.fb_download(ptr, n * n * 4)
        dt = time.perf_counter() - t0
    s = ctx.stats(st)
    print(f"{name:10s} aa{aa} gl{gl} {n}x{n}: {dt * 1e3:9.2f} ms, {s.shadow_samples / dt / 1e9:7.2f} Gsamples/s")
    ctx.close()
print("images identical:", np.array_equal(imgs["wavefront"], imgs["persistent"]))
