"""The C++ host side end to end (crt_main: scene classes -> flatten -> C ABI -> file.hdr), and error paths of
the C ABI that must leave a context usable."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cudaraytracer_amd as crt
from cudaraytracer_amd import abi

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
CRT_MAIN = ROOT / "cudaraytracer_amd" / "crt_main"


def _run_main(*args):
    return subprocess.run([str(CRT_MAIN), *[str(a) for a in args]], capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("progressive", [False, True])
def test_crt_main_writes_the_file_the_abi_path_would(tmp_path, oracle, progressive):
    """The headless stand-in for the reference's main() (main.cu:614-894) builds the Cornell box from the C++
    scene classes, renders and writes file.hdr; the bytes must be those of the same frame rendered through
    ctypes and encoded by crt_encode_hdr -- and that frame is the CPU restatement's."""
    assert CRT_MAIN.exists(), "build crt_main with make -C cudaraytracer_amd/csrc"
    out, ppm = tmp_path / "file.hdr", tmp_path / "preview.ppm"
    args = ["--nx", 64, "--ny", 48, "--gl_iter", 1, "--ind_rays", 8, "--shadowSamples", 10, "--seed", 5, "--out", out,
            "--preview", ppm]
    kw = {}
    if progressive:
        args += ["--progressive", "--blocksPerDraw", 3]
        kw = dict(renderAllAtOnce=False, blocksPerDraw=3)
    res = _run_main(*args)
    assert res.returncode == 0, res.stderr
    assert "Create World time" in res.stdout and "Render time" in res.stdout and "Image saved" in res.stdout
    st = crt.make_settings(nx=64, ny=48, gl_iter=1, ind_rays=8, shadowSamples=10, **kw)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(10))
        ctx.rng_init(64, 48, 5)
        fb = ctx.render_to_host(st, progressive=progressive)
    assert out.read_bytes() == crt.encode_hdr(fb)
    if not progressive:
        ref, _, _ = oracle.render(oracle.cornell(10), st, 5)
        assert np.array_equal(fb, ref)
    raw = ppm.read_bytes()
    head = b"P6\n64 48\n255\n"
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(48, 64, 3), crt.to_rgba8(fb)[..., :3])


def test_crt_main_multi_gpu_host_path_gives_the_single_gpu_file(tmp_path):
    """--gpus N goes through craytracer::MultiGpuRenderer: one context and one shard per device, an RCCL gather
    to device 0, crt_scatter_shards.  One GPU is what this box has, so N = 1 here (a communicator of one rank,
    the gather to itself); the split itself is covered on one GPU by the shard tests of test_gpu_parity.py."""
    a, b = tmp_path / "one.hdr", tmp_path / "multi.hdr"
    common = ["--nx", 72, "--ny", 40, "--gl_iter", 1, "--ind_rays", 6, "--shadowSamples", 7, "--seed", 9]
    r1 = _run_main(*common, "--out", a)
    r2 = _run_main(*common, "--gpus", 1, "--out", b)
    assert r1.returncode == 0, r1.stderr
    assert r2.returncode == 0, r2.stderr
    assert "1 GPU" in r2.stdout
    assert a.read_bytes() == b.read_bytes()


@pytest.mark.parametrize("bad", [("--tx", 0), ("--ty", 0), ("--blocksPerDraw", 0)])
def test_crt_main_reports_a_bad_block_shape_instead_of_dying(tmp_path, bad):
    """The reference's error convention is exit code 99 with a message (main.cu:896-903); a zero block shape in
    progressive mode must end there, not in a division fault."""
    res = _run_main("--nx", 32, "--ny", 32, "--gl_iter", 0, "--progressive", *bad, "--out", tmp_path / "x.hdr")
    assert res.returncode == 99, (res.returncode, res.stderr)
    assert "must be positive" in res.stderr


def test_a_rejected_scene_upload_leaves_the_previous_scene_in_place(oracle):
    st = crt.make_settings(nx=40, ny=32, aa_iter=1, gl_iter=1, ind_rays=6, shadowSamples=9)
    good = crt.cornell_box(9)
    bad = crt.cornell_box(3)          # fewer shadow samples: would size queues and jump tables differently
    bad.geometry[5].kind = 7          # ... but its sixth primitive is of no known kind
    with crt.Context(0) as ctx:
        ctx.upload(good)
        with pytest.raises(crt.CrtError):
            ctx.upload(bad)
        ctx.rng_init(st.nx, st.ny, 12)
        fb = ctx.render_to_host(st)
        state = ctx.rng_export(st.nx, st.ny)
    ref, _, ref_state = oracle.render(oracle.cornell(9), st, 12, want_state=True)
    assert np.array_equal(state, ref_state)
    assert np.array_equal(fb, ref)


def test_a_launch_too_large_for_the_pipeline_takes_the_persistent_kernel():
    """5800 x 5800 pixels exceed the wavefront pipeline's record index (29 bits next to the light's 3); the default
    configuration renders them with the persistent kernel instead of failing, an explicit request for the
    pipeline is refused."""
    st = crt.make_settings(nx=5800, ny=5800, aa_iter=0, gl_iter=0, shadowSamples=1)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(1))
        ctx.rng_init(st.nx, st.ny, 2)
        n = st.nx * st.ny * 4
        fb = ctx.fb_alloc(n)
        try:
            ctx.render(st, fb)
            img = ctx.fb_download(fb, n).reshape(st.ny, st.nx, 4)
            assert img[..., 3].min() >= 0.0 and img[2900, 2900, 0] > 0.0
            ctx.set_pipeline(abi.CRT_PIPELINE_WAVEFRONT)
            with pytest.raises(crt.CrtError) as err:
                ctx.render(st, fb)
            assert err.value.code == abi.CRT_ERR_UNSUPPORTED
        finally:
            ctx.fb_free(fb)


@pytest.mark.gpu
def test_bench_with_two_ranks_on_one_gpu_assembles_the_single_rank_frame():
    """`python bench.py --gpus 2` starts its own two ranks; with --share-gpu both render their interleaved tiles on
    GPU 0 and rank 0 gathers them (over gloo, RCCL needs a GPU per rank) and scatters them: everything of the
    N > 1 path but the RCCL call itself.  The assembled frame must be the one-rank frame bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--no-cpu", "--no-rows", "--frame-sha", "--nx", "96", "--ny", "64", "--workload", "gl1"]
    lines = []
    for flags in (["--gpus", "1"], ["--gpus", "2", "--share-gpu"]):
        p = subprocess.run([sys.executable, str(root / "bench.py"), *flags, *common], capture_output=True, text=True, timeout=280, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        out = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == 1, p.stdout
        lines.append(json.loads(out[0]))
    one, two = lines
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["steps"] == 2
    assert two["frame_sha256"] == one["frame_sha256"]
    assert two["work_per_step"]["shadow_samples"] == one["work_per_step"]["shadow_samples"]
