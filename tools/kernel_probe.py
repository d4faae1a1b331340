"""python tools/kernel_probe.py N AA_ITER GL_ITER [frames]: wall time per frame of back-to-back renders (no per-kernel
timing, so single-round frames stay asynchronous), then the per-kernel milliseconds of the same frames with the
event timer on."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

n, aa, gl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 20
st = crt.make_settings(nx=n, ny=n, aa_iter=aa, gl_iter=gl)
ctx = crt.Context(0)
ctx.upload(crt.cornell_box(50))
ctx.rng_init(n, n, 1234)
buf = ctx.fb_alloc(n * n * 4)
ctx.reserve_workspace(st)
for _ in range(3):
    ctx.render(st, buf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    ctx.render(st, buf)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / frames * 1e3
ctx.set_timing(True)
for _ in range(frames):
    ctx.render(st, buf)
torch.cuda.synchronize()
k = {name: round(ms / frames, 4) for name, (ms, _) in ctx.kernel_times().items()}
print(f"{n}x{n} aa {aa} gl {gl}: {wall:.3f} ms/frame back to back; kernels {k} (sum {sum(k.values()):.3f})")
