"""What one rank of an N-GPU job does, on one GPU: python tools/shard_bench.py WORLD [gl_iter] [steps] [trace mode 0|1|2] [GiB].
Renders shard 0 of WORLD of the 720x720 Cornell box and prints wall and per-kernel milliseconds per frame.
A fifth argument is the workspace limit in GiB (default: the library's)."""
import sys
import time
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
import torch  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

world = int(sys.argv[1])
gl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
st = crt.make_settings(gl_iter=gl)
ctx = crt.Context(0)
ctx.set_trace_mode(mode)
if len(sys.argv) > 5:
    ctx.set_workspace_limit(int(float(sys.argv[5]) * (1 << 30)))
ctx.upload(crt.cornell_box(50))
ctx.rng_init(st.nx, st.ny, 1234)
n = crt.shard_tiles_max(st.nx, st.ny, world) * 64 * 4
buf = ctx.fb_alloc(n)
ctx.reserve_workspace(st, 0, world)
ctx.render_shard(st, 0, world, buf)
torch.cuda.synchronize()
ctx.set_timing(True)
t0 = time.perf_counter()
for _ in range(steps):
    ctx.render_shard(st, 0, world, buf)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
k = {name: round(ms / steps, 3) for name, (ms, _) in ctx.kernel_times().items()}
rounds = {name: n // steps for name, (_, n) in ctx.kernel_times().items()}["trace"]
print(f"world {world} gl {gl} trace mode {mode}: {wall:.2f} ms/frame, {rounds} rounds, kernels {k} (sum {sum(k.values()):.2f})")
