"""Times the first rounds of one rank's launch of BASELINE's last configuration (720x720, aa_iter 4, gl_iter 3):
python tools/c5_probe.py WORLD ROUNDS [trace mode 0|1|2]"""
import os
import sys
import time
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
import torch  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

world, rounds = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
st = crt.make_settings(aa_iter=4, gl_iter=3)
ctx = crt.Context(0)
ctx.set_trace_mode(mode)
ctx.upload(crt.cornell_box(50))
ctx.rng_init(st.nx, st.ny, 1234)
n = crt.shard_tiles_max(st.nx, st.ny, world) * 64 * 4
buf = ctx.fb_alloc(n)
t0 = time.perf_counter()
ctx.reserve_workspace(st, 0, world)
print(f"workspace reserved in {time.perf_counter() - t0:.2f} s", flush=True)
ctx.set_timing(True)
ctx.set_round_limit(rounds)
t0 = time.perf_counter()
try:
    ctx.render_shard(st, 0, world, buf)
except crt.CrtError as err:  # cut short on purpose
    if err.code != crt.abi.CRT_ERR_TRUNCATED:
        raise
torch.cuda.synchronize()
dt = time.perf_counter() - t0
k = {name: (round(ms, 1), int(cnt)) for name, (ms, cnt) in ctx.kernel_times().items()}
pts = int(ctx.stats(st).shade_points)
print(f"world {world}, {rounds} rounds, trace mode {mode}: {dt:.2f} s, {pts} shaded points, {pts / dt / 1e9:.3f} G points/s, kernels {k}", flush=True)
