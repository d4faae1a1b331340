"""One render of the Cornell box at the given size and iteration counts, with its wall time and work:
python tools/config_probe.py N AA_ITER GL_ITER   (prints a heartbeat while the render call runs)."""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

n, aa, gl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
st = crt.make_settings(nx=n, ny=n, aa_iter=aa, gl_iter=gl)
ctx = crt.Context(0)
ctx.upload(crt.cornell_box(50))
ctx.rng_init(n, n, 1234)
buf = ctx.fb_alloc(n * n * 4)
ctx.reserve_workspace(st)
stop = threading.Event()


def heartbeat():
    t0 = time.perf_counter()
    while not stop.wait(30.0):
        print(f"... rendering, {time.perf_counter() - t0:.0f} s", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
t0 = time.perf_counter()
ctx.render(st, buf)
torch.cuda.synchronize()
t = time.perf_counter() - t0
stop.set()
s = ctx.stats(st)
img = ctx.fb_download(buf, n * n * 4).reshape(n, n, 4)
print(f"{n}x{n} aa_iter {aa} gl_iter {gl}: {t:.2f} s, {s.shade_points} shaded points ({s.shade_points / (n * n):.0f} per pixel), "
      f"{s.shadow_samples / t / 1e9:.1f} G shadow samples/s, image mean {img[..., :3].mean():.5f}, "
      f"same rate at 720x720: {t * 518400 / (n * n):.1f} s", flush=True)
