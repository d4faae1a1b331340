"""Would a frame get faster if its kernels overlapped?
    python tools/overlap_probe.py PARTS [WORLD] [gl_iter] [steps] [trace mode] [stagger ms] [lockstep 0|1]
PARTS contexts, each with its own stream and host thread, render shard (r % WORLD) of WORLD of the 720x720 Cornell box
on ONE GPU at the same time, thread r starting r * stagger ms late, so that one launch's trace kernel runs next to
another's filter, resolve and shade kernels.  WORLD = PARTS: one frame in pieces; WORLD = 1: whole frames side by side.
With lockstep the threads wait for each other after every frame (and every stream is drained), so that the figure is
one frame's time, not the throughput of frames that run into each other.
Prints the wall time per whole frame next to the plain single-launch frame."""
import sys
import threading
import time
from pathlib import Path

root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
import torch  # noqa: E402
import cudaraytracer_amd as crt  # noqa: E402

parts = int(sys.argv[1])
world_arg = int(sys.argv[2]) if len(sys.argv) > 2 else parts
gl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
stagger = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
lockstep = len(sys.argv) > 7 and sys.argv[7] == "1"
st = crt.make_settings(gl_iter=gl)
scene = crt.cornell_box(50)
dev = torch.device("cuda", 0)


def frame_ms(parts, world, stagger):
    ctxs, bufs, streams = [], [], []
    for r in range(parts):
        c = crt.Context(0)
        c.set_trace_mode(mode)
        c.set_workspace_limit((96 << 30) // parts)
        c.upload(scene)
        c.rng_init(st.nx, st.ny, 1234)
        bufs.append(c.fb_alloc(crt.shard_tiles_max(st.nx, st.ny, world) * 64 * 4))
        c.reserve_workspace(st, r % world, world)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(dev))
    gate = threading.Barrier(parts + 1)
    frame_gate = threading.Barrier(parts)

    def run(r):
        s = streams[r].cuda_stream
        ctxs[r].render_shard(st, r % world, world, bufs[r], s)   # warm-up
        torch.cuda.synchronize(dev)
        gate.wait()
        if stagger:
            time.sleep(r * stagger * 1e-3)
        for _ in range(steps):
            ctxs[r].render_shard(st, r % world, world, bufs[r], s)
            if lockstep:
                streams[r].synchronize()
                frame_gate.wait()
        streams[r].synchronize()
        gate.wait()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(parts)]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    ms = (time.perf_counter() - t0) / (steps * parts / world) * 1e3
    for t in threads:
        t.join()
    for c, b in zip(ctxs, bufs):
        c.fb_free(b)
        c.close()
    return ms


one = frame_ms(1, 1, 0.0)
many = frame_ms(parts, world_arg, stagger)
print(f"gl {gl} trace mode {mode}: one launch {one:.2f} ms/frame; {parts} launches of 1/{world_arg} frame side by side, "
      f"{stagger} ms apart: {many:.2f} ms/frame")
