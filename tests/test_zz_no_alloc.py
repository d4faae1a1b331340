"""crt_render_static -- the entry point that allocates nothing -- against crt_render, through the C ABI.

Bar: identical floats and identical RNG end states (both run the same persistent kernel on the same inputs;
only where the kernel's scratch lives differs).  Two calls in a row on the same GPU, the second on a smaller
image, so that whatever the first call left in the shared static array must not leak into the second.
"""
import ctypes as C

import numpy as np
import pytest

import cudaraytracer_amd as crt
from cudaraytracer_amd import abi

pytestmark = pytest.mark.gpu


def _free_device_memory():
    """hipMemGetInfo of the current device, straight from the HIP runtime the library is linked against."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _render(ctx, st, seed, static):
    ctx.rng_init(st.nx, st.ny, seed)          # set-up: this is where the context allocates
    n = st.nx * st.ny * 4
    fb = ctx.fb_alloc(n)
    try:
        if static:
            ctx.render_static(st, fb)
        else:
            ctx.render(st, fb)
        img = ctx.fb_download(fb, n).reshape(st.ny, st.nx, 4)
    finally:
        ctx.fb_free(fb)
    return img, ctx.rng_export(st.nx, st.ny)


def test_static_entry_point_matches_crt_render_twice_in_a_row():
    first = crt.make_settings(nx=72, ny=56, aa_iter=2, gl_iter=1, ind_rays=9, shadowSamples=7)
    second = crt.make_settings(nx=40, ny=24, aa_iter=1, gl_iter=2, ind_rays=5, shadowSamples=4)  # smaller input
    scene = crt.cornell_box(7)
    with crt.Context(0) as ref_ctx, crt.Context(0) as ctx:
        ref_ctx.set_pipeline(abi.CRT_PIPELINE_PERSISTENT)  # the existing entry point on the same kernel
        ref_ctx.upload(scene)
        ctx.upload(scene)
        for st, seed in ((first, 31), (second, 32)):
            want, want_state = _render(ref_ctx, st, seed, static=False)
            got, got_state = _render(ctx, st, seed, static=True)
            assert np.array_equal(got_state, want_state), "RNG end states differ"
            assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} floats differ"
            assert got.max() > 0.0


def test_static_entry_point_matches_the_default_pipeline_too(oracle):
    """Same bits as the wavefront pipeline behind the default crt_render, and as the CPU restatement."""
    st = crt.make_settings(nx=33, ny=21, aa_iter=1, gl_iter=1, ind_rays=6, shadowSamples=5)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(5))
        got, got_state = _render(ctx, st, 8, static=True)
        want, want_state = _render(ctx, st, 8, static=False)
    ref, _, ref_state = oracle.render(oracle.cornell(5), st, 8, want_state=True)
    assert np.array_equal(got, want) and np.array_equal(got_state, want_state)
    assert np.array_equal(got, ref) and np.array_equal(got_state, ref_state)


def test_static_entry_point_leaves_device_memory_alone():
    """Free device memory (hipMemGetInfo) is the same before and after a call.  The first launch of a kernel in
    a process can make the HIP runtime set up its own per-queue memory, so one call is made beforehand."""
    st = crt.make_settings(nx=64, ny=48, aa_iter=1, gl_iter=1, ind_rays=8, shadowSamples=6)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(6))
        ctx.rng_init(st.nx, st.ny, 3)
        fb = ctx.fb_alloc(st.nx * st.ny * 4)
        try:
            ctx.render_static(st, fb)
            ctx.fb_download(fb, 4)  # synchronises
            before = _free_device_memory()
            for _ in range(3):
                ctx.render_static(st, fb)
            after_launch = _free_device_memory()
            ctx.fb_download(fb, 4)
            after = _free_device_memory()
            assert after_launch == before and after == before, (before, after_launch, after)
        finally:
            ctx.fb_free(fb)


def test_static_entry_point_refuses_images_beyond_its_scratch():
    st = crt.make_settings(nx=1025, ny=1024, aa_iter=0, gl_iter=0, shadowSamples=1)
    with crt.Context(0) as ctx:
        ctx.upload(crt.cornell_box(1))
        ctx.rng_init(st.nx, st.ny, 1)
        fb = ctx.fb_alloc(16)  # never written: the call must fail before launching anything
        try:
            with pytest.raises(crt.CrtError) as err:
                ctx.render_static(st, fb)
            assert err.value.code == abi.CRT_ERR_UNSUPPORTED and "CRT_STATIC_MAX_PIXELS" in str(err.value)
        finally:
            ctx.fb_free(fb)


def test_static_calls_from_two_contexts_and_streams_are_queued_one_after_the_other():
    """Two contexts on one GPU, each with its own stream, call crt_render_static back to back without waiting:
    the calls share the GPU's scratch array, so the library queues the second behind the first (an event per
    device).  Both images and RNG end states must be what crt_render gives each context on its own."""
    import torch
    sts = [crt.make_settings(nx=96, ny=80, aa_iter=1, gl_iter=1, ind_rays=12, shadowSamples=6),
           crt.make_settings(nx=64, ny=40, aa_iter=2, gl_iter=1, ind_rays=7, shadowSamples=6)]
    scene = crt.cornell_box(6)
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    with crt.Context(0) as a, crt.Context(0) as b, crt.Context(0) as ref_ctx:
        ref_ctx.set_pipeline(abi.CRT_PIPELINE_PERSISTENT)
        ref_ctx.upload(scene)
        want = [_render(ref_ctx, st, 40 + k, static=False) for k, st in enumerate(sts)]
        ctxs = [a, b]
        fbs = []
        for k, (ctx, st) in enumerate(zip(ctxs, sts)):
            ctx.upload(scene)
            ctx.rng_init(st.nx, st.ny, 40 + k)
            fbs.append(torch.zeros((st.ny, st.nx, 4), dtype=torch.float32, device=dev))
        torch.cuda.synchronize(dev)
        for rep in range(3):  # interleaved, no waiting in between; every repetition renders the same frame again
            for k, (ctx, st) in enumerate(zip(ctxs, sts)):
                if rep:
                    ctx.rng_init(st.nx, st.ny, 40 + k, streams[k].cuda_stream)
                ctx.render_static(st, fbs[k].data_ptr(), streams[k].cuda_stream)
        torch.cuda.synchronize(dev)
        for k, (ctx, st) in enumerate(zip(ctxs, sts)):
            assert np.array_equal(fbs[k].cpu().numpy(), want[k][0]), f"context {k}: image differs"
            assert np.array_equal(ctx.rng_export(st.nx, st.ny), want[k][1]), f"context {k}: RNG end states differ"
