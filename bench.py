#!/usr/bin/env python3
"""Benchmark of the per-pixel render path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload gl0|gl1|gl2]

With N > 1 and no rank environment the script starts its own ranks (`python -m torch.distributed.run`, one
process per GPU, rendezvous on 127.0.0.1) before anything touches a GPU, passes their output through and
exits with their status; started under torch.distributed.run it is one of those ranks.  `--dry-run` runs the
same launcher and the same gather/scatter over gloo on CPU tensors without rendering (no GPU needed).

A step is one render of the README's 720x720 Cornell box (aa_iter=1, ref_iter=4,
ind_rays=75, shadowSamples=50; main.cu:626-636) at the workload's gl_iter, with
the scene, LTC tables, RNG states and framebuffer resident in HBM.  With N>1 the
image is sharded as interleaved 8x8 tiles, one rank per GPU, one RCCL gather to
rank 0 per step (total work fixed => "strong" scaling).  Rank 0 prints one JSON
line; `value` is shadow samples per second over all GPUs (Msamples/s) and
`ms_per_step` the render time the README table quotes.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# README.md:131-134 of the reference (RTX 4070 Ti SUPER), seconds per render
README_4070TIS_S = {"gl0": 0.116, "gl1": 10.438, "gl2": 574.214, "gl3": 23040.602}
PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak (needs packed v_pk_fma_f32; see peak_note)
PEAK_HBM_TBPS = 8.0       # MI355X HBM3E


def algorithmic_flops(stats):
    """FLOPs of the reference's arithmetic as written: 300 per shadow sample, 1200 per shaded
    point (closest hit + LTC shading), 40 per generated GI child."""
    return 300.0 * stats["shadow_samples"] + 1200.0 * stats["shade_points"] + 40.0 * stats["gi_children"]


def kernel_flops(stats):
    """The same model split by the kernel that does the work: the closest-hit query of a shaded point (450 of its
    1200) and the generation of the GI children are the trace kernel's, the shadow samples the filter and resolve
    kernels', AreaLight::calculateColor's LTC part (750) the shade kernel's; the sum kernel only adds."""
    return {"trace": 450.0 * stats["shade_points"] + 40.0 * stats["gi_children"],
            "filter+resolve": 300.0 * stats["shadow_samples"],
            "shade": 750.0 * stats["shade_points"]}


def per_kernel_roofline(ktimes, per_launch):
    """{kernel: ms, model TFLOP/s and fraction of the packed FP32 peak} from per-frame kernel times and work."""
    fl = kernel_flops(per_launch)
    ms = {"trace": ktimes.get("trace", (0, 0))[0], "filter+resolve": ktimes.get("filter", (0, 0))[0] + ktimes.get("resolve", (0, 0))[0],
          "shade": ktimes.get("shade", (0, 0))[0]}
    out = {}
    for k in fl:
        if ms[k] > 0:
            t = fl[k] / (ms[k] * 1e-3) / 1e12
            out[k] = {"ms": ms[k], "model_tflops": t, "frac": t / PEAK_FP32_TFLOPS}
    return out


def stats_dict(s):
    return {"rng_draws": int(s.rng_draws), "shadow_samples": int(s.shadow_samples),
            "shade_points": int(s.shade_points), "gi_children": int(s.gi_children),
            "exact_tests": int(s.exact_tests), "glass_walks": int(s.glass_walks)}


def diff(a, b):
    return {k: b[k] - a[k] for k in a}


def cpu_baseline(workload_gl, budget_s):
    """Times the CPU restatement (oracle/) on the host cores on a centred crop of the same 720^2 render."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    import cudaraytracer_amd as crt
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    crop = {0: 720, 1: 400, 2: 40}[workload_gl]  # about 10-20 s on 16 host threads
    st = crt.make_settings(gl_iter=workload_gl)
    x0 = (720 - crop) // 2
    sc = O.cornell(50)
    t0 = time.perf_counter()
    _, cnt, _ = O.render(sc, st, 1234, region=(x0, x0, x0 + crop, x0 + crop), threads=threads)
    dt = time.perf_counter() - t0
    shadow = (cnt.rng_draws - 2 * cnt.gi_children) // 2
    model, physical = "unknown", set()
    try:
        phys = core = None
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "model name":
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not key and phys is not None:
                physical.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return {"value": shadow / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "host": {"cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": len(physical) or None,
                     "threads_used": threads},
            "sample": f"centred {crop}x{crop} crop of the 720x720 gl_iter={workload_gl} render, seed 1234, "
                      f"{shadow} shadow samples in {dt:.2f} s (OpenMP, dynamic rows)"}


def config5(args, torch, crt):
    """BASELINE's last configuration: 720x720, aa_iter 4, gl_iter 3, ind_rays 75, 50 shadow samples, 8 GPUs.
    The whole frame is about 2.5e13 shaded points -- two hours of one GPU -- so it is measured in parts, all on
    this GPU, and put together:
      work   shaded points of the frame = points(aa_iter 1, gl_iter 3) x points(aa_iter 4, gl_iter 1) /
             points(aa_iter 1, gl_iter 1): the first is the README's 6 h 24 min row rendered in full, the ratio is
             the number of color() calls adaptive anti-aliasing makes per pixel, 4 at aa_iter 1 (it does not
             depend on gl_iter as long as gl_iter >= 1: the GI noise makes every colour comparison fail);
      rate   what one rank of eight launches -- crt_render_shard(rank 0 of 8), 64 800 pixels in interleaved 8x8
             tiles -- timed over its first --c5-rounds pipeline rounds (crt_set_round_limit), with the GPU as full as
             it is in the real job (the end of the frame, where few pixels are left, is not in the sample).
    Reported: the rate in Msamples/s, and work / 8 / rate as the time eight GPUs would need."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream(dev)
    nx = ny = 720
    points = {}
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(50))
        fb = torch.empty((ny, nx, 4), dtype=torch.float32, device=dev)
        for aa, gl in ((1, 1), (4, 1), (1, 3)):
            st = crt.make_settings(nx=nx, ny=ny, aa_iter=aa, gl_iter=gl)
            ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)
            t0 = time.perf_counter()
            ctx.render(st, fb.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            points[(aa, gl)] = {"shade_points": int(ctx.stats(st).shade_points), "s": time.perf_counter() - t0}
            print(f"# aa_iter {aa} gl_iter {gl}: {points[(aa, gl)]}", file=sys.stderr, flush=True)
    calls_ratio = points[(4, 1)]["shade_points"] / points[(1, 1)]["shade_points"]
    frame_points = points[(1, 3)]["shade_points"] * calls_ratio
    st = crt.make_settings(nx=nx, ny=ny, aa_iter=4, gl_iter=3)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(50))
        ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)
        n = crt.shard_tiles_max(nx, ny, 8) * 64 * 4
        compact = torch.empty(n, dtype=torch.float32, device=dev)
        ctx.reserve_workspace(st, 0, 8)
        ctx.set_round_limit(args.c5_rounds)  # this context only; the call below reports the frame as cut short
        torch.cuda.synchronize(dev)
        print(f"# sample: rank 0 of 8, {args.c5_rounds} rounds ...", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        try:
            ctx.render_shard(st, 0, 8, compact.data_ptr(), stream.cuda_stream)
        except crt.CrtError as err:
            if err.code != crt.abi.CRT_ERR_TRUNCATED:
                raise
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        done = int(ctx.stats(st).shade_points)
        print(f"# sample: {done} shaded points in {dt:.2f} s", file=sys.stderr, flush=True)
    rate = done / dt
    est_s = frame_points / 8 / rate
    print(json.dumps({
        "metric": "Cornell 720x720 aa_iter=4 gl_iter=3 (BASELINE configs[4]): shadow-sample throughput of one rank of 8, "
                  "extrapolated render time on 8 GPUs",
        "value": rate * 50 / 1e6, "unit": "Msamples/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
        "data": "synthetic", "vs_baseline": None,
        "config": {"workload": "Cornell box 720x720, aa_iter=4, ref_iter=4, gl_iter=3, ind_rays=75, shadowSamples=50",
                   "sampling": f"rank 0 of 8 (interleaved 8x8 tiles, 64 800 pixels), first {args.c5_rounds} pipeline rounds: "
                               f"{done} shaded points in {dt:.2f} s",
                   "seed": args.seed},
        "frame_shade_points": frame_points, "color_calls_per_pixel": 4 * calls_ratio,
        "extrapolated_s_on_8_gpus": est_s, "extrapolated_s_on_1_gpu": frame_points / rate,
        "parts": {f"aa{aa}_gl{gl}": v for (aa, gl), v in points.items()},
        "readme_aa1_gl3_rtx4070tis_s": README_4070TIS_S["gl3"],
    }), flush=True)


def launch_ranks(n):
    """Re-runs this command line as n ranks under torch.distributed.run and returns their exit status.
    The parent has made no HIP call, so the children are the only processes on the GPUs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver shares device memory through dmabuf only
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """The N-rank plumbing without a GPU: gloo process group, every rank fills its compact shard with a pattern
    keyed by global pixel index, the timed steps are gather + scatter, rank 0 checks the reassembled frame."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cudaraytracer_amd import distributed as shard
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nx, ny = args.nx, args.ny
    idx = shard.compact_pixel_index(nx, ny, rank, world)
    compact = np.zeros((idx.size, 4), np.float32)
    compact[idx >= 0] = (idx[idx >= 0, None] * 4 + np.arange(4)[None, :]).astype(np.float32)
    compact = torch.from_numpy(compact.reshape(-1))
    fb = None
    for _ in range(args.warmup):
        fb = shard.gather_image(compact, nx, ny)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fb = shard.gather_image(compact, nx, ny)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        want = np.arange(nx * ny * 4, dtype=np.float32).reshape(ny, nx, 4)
        ok = fb is not None and np.array_equal(fb.numpy(), want)
        print(json.dumps({"metric": "dry run: gather + scatter of the interleaved-tile shards over gloo, nothing rendered",
                          "dry_run": True, "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "frame_ok": bool(ok),
                          "config": {"workload": f"{nx}x{ny} pattern image", "sharding": "interleaved 8x8 tiles, tile % n_gpus"}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if rank != 0 or ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["gl0", "gl1", "gl2", "gl3", "c5"], default="gl1",
                    help="README benchmark row; gl3 takes about 80 s per step on one GPU; c5 = BASELINE's last "
                         "configuration (aa_iter 4, gl_iter 3), measured on a sample and extrapolated (minutes)")
    ap.add_argument("--c5-rounds", type=int, default=24, help="pipeline rounds timed by --workload c5")
    ap.add_argument("--nx", type=int, default=720)
    ap.add_argument("--ny", type=int, default=720)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--pipeline", choices=["auto", "wavefront", "persistent"], default="auto",
                    help="kernels behind crt_render: auto = wavefront pipeline where it applies")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rows", action="store_true", help="skip the extra gl0/gl2 README rows at N=1")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of N > 1 on a one-GPU box: every rank renders its shard on GPU 0 and the gather goes "
                         "over gloo through host memory (RCCL does not put two ranks on one device); not a scaling measurement")
    ap.add_argument("--frame-sha", action="store_true", help="add the SHA-256 of the last assembled frame to the line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher, process group (gloo) and gather/scatter only, on CPU tensors: no GPU, no render")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # N > 1 without a rank environment: start the ranks (before torch or HIP are touched in this process)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if args.dry_run:
        raise SystemExit(dry_run(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))))

    import torch
    import torch.distributed as dist
    import cudaraytracer_amd as crt
    import cudaraytracer_amd.abi  # noqa: F401
    from cudaraytracer_amd import distributed as shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = torch.device("cpu") if args.share_gpu else dev  # where the scalar reductions below live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload == "c5":
        return config5(args, torch, crt)
    gl = int(args.workload[2])
    nx, ny = args.nx, args.ny
    st = crt.make_settings(nx=nx, ny=ny, gl_iter=gl)
    ctx = crt.Context(dev_index)
    ctx.set_pipeline({"auto": crt.abi.CRT_PIPELINE_AUTO, "wavefront": crt.abi.CRT_PIPELINE_WAVEFRONT,
                      "persistent": crt.abi.CRT_PIPELINE_PERSISTENT}[args.pipeline])
    ctx.upload(crt.cornell_box(st.shadowSamples))
    stream = torch.cuda.current_stream(dev)
    ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)

    ctx.reserve_workspace(st, rank, world)  # allocation is set-up, not part of a step

    tiles_max = shard.shard_tiles_max(nx, ny, world)
    fb = torch.empty((ny, nx, 4), dtype=torch.float32, device=dev)
    compact = torch.empty(tiles_max * 64 * 4, dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        if world == 1:
            ctx.render(st, fb.data_ptr(), stream.cuda_stream)
        else:
            ctx.render_shard(st, rank, world, compact.data_ptr(), stream.cuda_stream)
            shard.gather_image(compact, nx, ny, ctx, fb)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    s0 = stats_dict(ctx.stats(st))
    ctx.set_timing(True)  # per-kernel HIP events on the launch stream, accumulated inside the library
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        if world == 1:
            ctx.render(st, fb.data_ptr(), stream.cuda_stream)
            ev[k][1].record(stream)
        else:
            ctx.render_shard(st, rank, world, compact.data_ptr(), stream.cuda_stream)
            ev[k][1].record(stream)
            shard.gather_image(compact, nx, ny, ctx, fb)
    sync()
    elapsed = time.perf_counter() - t0
    s1 = stats_dict(ctx.stats(st))
    work = diff(s0, s1)  # this rank's work over the K timed steps
    render_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # whole crt_render call, HIP events on its stream
    ktimes = {k: (ms / args.steps, n / args.steps) for k, (ms, n) in ctx.kernel_times().items() if n}
    ctx.set_timing(False)
    # GPU time of the kernels of one render: the wavefront kernels' own events, or the single persistent kernel.  A
    # large launch is rendered as two interleaved pieces side by side, each with its own five kernels per round: their
    # times overlap and add up to more than the frame, so the frame's own events (on the caller's stream, which
    # waits for both pieces) are the GPU time then.
    kernel_sum = sum(ms for ms, _ in ktimes.values()) if ktimes else render_ms
    in_pieces = bool(ktimes) and kernel_sum > 1.2 * render_ms
    kernel_ms = render_ms if in_pieces else kernel_sum

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        w = torch.tensor([work["shadow_samples"], work["shade_points"], work["gi_children"], work["rng_draws"]],
                         dtype=torch.int64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        total = dict(zip(["shadow_samples", "shade_points", "gi_children", "rng_draws"], [int(x) for x in w.tolist()]))
    else:
        total = work

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        per_launch = {k: v / args.steps for k, v in work.items()}
        achieved = algorithmic_flops(per_launch) / (kernel_ms * 1e-3) / 1e12
        traffic = traffic_from = None
        per_kernel = per_kernel_roofline(ktimes, per_launch)
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():  # PMC passes are separate rocprofv3 runs (tools/prof_traffic.sh); the line quotes their result
            try:
                entry = json.loads(tfile.read_text()).get(f"{args.workload}_n{world}_{nx}x{ny}", {})
                traffic, traffic_from = entry.get("hbm_bytes_per_launch"), entry.get("profile")
                # the profiled bytes of each kernel over this run's time for it: which kernels the memory side bounds
                pk = entry.get("per_kernel", {})
                names = {"trace": ["wf_trace_kernel"], "filter+resolve": ["wf_filter_kernel", "wf_resolve_kernel"],
                         "shade": ["wf_shade_kernel"], "sum": ["wf_sum_kernel"]}
                for k, kernels in names.items():
                    nbytes = sum(pk.get(n, {}).get("hbm_bytes_upper", 0.0) for n in kernels)
                    ms = per_kernel.get(k, {}).get("ms") or (ktimes.get(k, (0, 0))[0] if k in ktimes else 0)
                    if nbytes and ms:
                        per_kernel.setdefault(k, {"ms": ms})
                        per_kernel[k].update({"hbm_gb_profiled": nbytes / 1e9, "hbm_tbps": nbytes / (ms * 1e-3) / 1e12,
                                              "hbm_frac": nbytes / (ms * 1e-3) / 1e12 / PEAK_HBM_TBPS})
            except Exception:
                traffic = None
        out = {
            "metric": "Cornell 720x720 render: shadow-sample throughput (render time in ms_per_step)",
            "value": total["shadow_samples"] / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Cornell box {nx}x{ny}, aa_iter=1, ref_iter=4, gl_iter={gl}, ind_rays=75, "
                                   f"shadowSamples=50 (README row {gl})",
                       "seed": args.seed, "sharding": "interleaved 8x8 tiles, tile % n_gpus, one gather to rank 0"
                       if world > 1 else "none",
                       "primary_samples_per_s": nx * ny * 4 / (elapsed / args.steps)},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_profile": traffic_from,
                         "kernel": ("wavefront pipeline (" + "+".join(ktimes) + ")") if ktimes else "crt::render_kernel",
                         "kernel_ms": kernel_ms, "render_call_ms": render_ms,
                         "pieces": 2 if in_pieces else 1,
                         "pieces_note": "the launch runs as two interleaved half-frames side by side (two streams): kernel_ms is "
                                        "the render call's HIP events on the caller's stream; the per-kernel times below are "
                                        "summed over both pieces, which overlap, and exceed it" if in_pieces else None,
                         "kernels": {k: {"ms": ms, "launches": n} for k, (ms, n) in ktimes.items()},
                         "per_kernel": per_kernel,
                         "per_kernel_note": "model FLOPs by the kernel that does them: trace 450*shade_points + 40*gi_children, "
                                            "filter+resolve 300*shadow_samples (97.8 % of the rays are decided by conservative "
                                            "tests that cost a fraction of the exact test they stand for), shade 750*shade_points; "
                                            "hbm_*: the profile's bytes for the kernel (upper bound, see traffic_profile) over this "
                                            "run's time for it, against 8 TB/s",
                         "peak_note": "157.3 = 256 CU x 4 SIMD x 16 lanes/clk x 2 (packed v_pk_fma_f32) x 2 flop x 2.4 GHz; "
                                      "a plain wave64 VALU instruction issues every 4 cycles per SIMD (78.6 TFLOP/s for unpacked fma)",
                         "flops_model": "300*shadow_samples + 1200*shade_points + 40*gi_children (rank 0's launch)"},
            "work_per_step": {k: v / args.steps for k, v in total.items()},
        }
        if args.share_gpu:
            out["rehearsal"] = "all ranks on GPU 0, gather over gloo through host memory: plumbing only, not a scaling number"
        if args.frame_sha:
            import hashlib
            out["frame_sha256"] = hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest()
        if nx == 720 and ny == 720:
            out["speedup_vs_readme_rtx4070tis"] = README_4070TIS_S[args.workload] / (ms_per_step * 1e-3)
        if world == 1 and not args.no_rows and nx == 720 and ny == 720:
            rows = {}
            for other in ("gl0", "gl2"):
                if other == args.workload:
                    continue
                st2 = crt.make_settings(nx=nx, ny=ny, gl_iter=int(other[2]))
                ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)
                ctx.render(st2, fb.data_ptr(), stream.cuda_stream)  # untimed warm-up
                torch.cuda.synchronize(dev)
                frames = 3 if other == "gl2" else 10
                ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(frames):
                    ctx.render(st2, fb.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t1) / frames
                w2 = {k: v / frames for k, v in stats_dict(ctx.stats(st2)).items()}
                rows[other] = {"ms": dt * 1e3, "frames": frames, "Msamples_per_s": w2["shadow_samples"] / dt / 1e6,
                               "tflops_algorithmic": algorithmic_flops(w2) / dt / 1e12,
                               "speedup_vs_readme_rtx4070tis": README_4070TIS_S[other] / dt}
                if other == "gl2":
                    # the north-star workload with a roofline block of its own: the same frames again with the kernels'
                    # HIP events on (which makes every frame wait for its last round, as a gl_iter 2 frame does anyway)
                    ctx.rng_init(nx, ny, args.seed, stream.cuda_stream)
                    ctx.set_timing(True)
                    torch.cuda.synchronize(dev)
                    t2 = time.perf_counter()
                    for _ in range(frames):
                        ctx.render(st2, fb.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    timed_frame_ms = (time.perf_counter() - t2) / frames * 1e3
                    kt = {k: (ms / frames, n / frames) for k, (ms, n) in ctx.kernel_times().items() if n}
                    ctx.set_timing(False)
                    ksum = sum(ms for ms, _ in kt.values())
                    kms = timed_frame_ms if ksum > 1.2 * timed_frame_ms else ksum  # in two pieces the kernels overlap
                    ach = algorithmic_flops(w2) / (kms * 1e-3) / 1e12
                    rows[other]["roofline"] = {"bound": "valu_fp32", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                               "frac": ach / PEAK_FP32_TFLOPS, "kernel_ms": kms, "pieces": 2 if ksum > 1.2 * timed_frame_ms else 1,
                                               "trace_launches": kt["trace"][1],
                                               "kernels": {k: {"ms": ms, "launches": n} for k, (ms, n) in kt.items()},
                                               "per_kernel": per_kernel_roofline(kt, w2), "work_per_frame": w2}
            out["readme_rows"] = rows
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(gl, 20.0)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
