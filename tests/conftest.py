import sys
from pathlib import Path

import pytest

try:  # first, so that the process ends up with one HIP runtime whichever test file runs alone
    import torch  # noqa: F401
except ImportError:  # the CPU-only tests of the checker, the texture and the LTC clipping need no torch
    torch = None

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session", params=["auto", "per_pixel", "persistent", "pieces"])
def gpu_ctx(request):
    """One context per pipeline on cuda:0 shared by the GPU tests (fails loudly without the HIP library
    or a GPU).  "auto" is the default configuration: the wavefront kernels, with the cooperative trace
    kernel for launches below 200 000 pixels; "per_pixel" forces the one-lane-per-pixel trace kernel;
    "pieces" renders every launch of two work units or more as two interleaved pieces side by side, which the
    default configuration does from 8 000 units on (the threshold is read from the environment by crt_create)."""
    import os
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import abi
    if request.param == "pieces":
        os.environ["CRT_FRAME_PIECES_MIN_UNITS"] = "2"
    try:
        ctx = crt.Context(0)
    finally:
        os.environ.pop("CRT_FRAME_PIECES_MIN_UNITS", None)
    ctx.set_pipeline(abi.CRT_PIPELINE_PERSISTENT if request.param == "persistent" else abi.CRT_PIPELINE_AUTO)
    if request.param == "per_pixel":
        ctx.set_trace_mode(abi.CRT_TRACE_PER_PIXEL)
    ctx.pipeline_name = request.param
    yield ctx
    ctx.close()
