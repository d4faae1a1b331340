"""N>1 plumbing on CPU: two gloo ranks each render their interleaved tiles, rank 0 gathers and scatters.

The per-pixel renderer here is the CPU restatement (the HIP kernels need a GPU); what is under test is the
sharding layout, the collective and the scatter of cudaraytracer_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudaraytracer_amd import abi, distributed as shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, seed, out_path):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = abi.make_settings(nx=nx, ny=ny, aa_iter=1, gl_iter=0, shadowSamples=6)
        full, _, _ = O.render(O.cornell(6), st, seed, threads=2)  # every rank can render any pixel identically
        idx = shard.compact_pixel_index(nx, ny, rank, world)
        compact = np.zeros((idx.size, 4), np.float32)
        compact[idx >= 0] = full.reshape(-1, 4)[idx[idx >= 0]]
        other = np.setdiff1d(np.arange(nx * ny), idx[idx >= 0])
        assert other.size > 0  # this rank really owns only part of the image
        fb = shard.gather_image(torch.from_numpy(compact.reshape(-1)), nx, ny)
        if rank == 0:
            np.save(out_path, fb.numpy())
        else:
            assert fb is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nx,ny", [(40, 24), (37, 21)])
def test_two_rank_gather_matches_single_render(tmp_path, oracle, nx, ny):
    port = _free_port()
    out = tmp_path / "fb.npy"
    mp.spawn(_worker, args=(2, port, nx, ny, 11, str(out)), nprocs=2, join=True)
    st = abi.make_settings(nx=nx, ny=ny, aa_iter=1, gl_iter=0, shadowSamples=6)
    ref, _, _ = oracle.render(oracle.cornell(6), st, 11, threads=2)
    assert np.array_equal(np.load(out), ref)


def test_single_rank_needs_no_process_group():
    nx, ny = 16, 9
    img = np.arange(nx * ny * 4, dtype=np.float32).reshape(ny, nx, 4)
    idx = shard.compact_pixel_index(nx, ny, 0, 1)
    compact = np.zeros((idx.size, 4), np.float32)
    compact[idx >= 0] = img.reshape(-1, 4)[idx[idx >= 0]]
    fb = shard.gather_image(torch.from_numpy(compact.reshape(-1)), nx, ny)
    assert np.array_equal(fb.numpy(), img)
