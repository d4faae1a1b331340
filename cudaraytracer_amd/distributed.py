"""Image sharding over GPUs: interleaved 8x8 tiles, one gather, one scatter.

Pixels are independent and every pixel's RNG stream is keyed by its global index
(random_init, main.cu:323-324), so any partition of the image renders the same
pixels bit for bit.  Tile t = ty*tiles_x + tx belongs to rank t % world; rank r
renders its k-th tile (t = k*world + r) into slots [k*64, k*64+64) of a compact
RGBA buffer that every rank pads to tiles_max tiles.  Rank 0 gathers the `world`
compact buffers (RCCL when the tensors live on GPUs, gloo on CPU) and scatters
them into the row-major framebuffer.  The reference is single-GPU
(main.cu:914-940); this layer has no counterpart there.
"""
import numpy as np


def tiles_xy(nx, ny):
    return (nx + 7) // 8, (ny + 7) // 8


def tile_owner(tile, world):
    return tile % world


def shard_tiles(nx, ny, rank, world):
    tx, ty = tiles_xy(nx, ny)
    n = tx * ty
    return n // world + (1 if rank < n % world else 0)


def shard_tiles_max(nx, ny, world):
    tx, ty = tiles_xy(nx, ny)
    return (tx * ty + world - 1) // world


def compact_pixel_index(nx, ny, rank, world):
    """int64 array [tiles_max*64]: global pixel index j*nx+i of each compact slot, -1 for padding."""
    tx, ty = tiles_xy(nx, ny)
    kmax = shard_tiles_max(nx, ny, world)
    k = np.arange(kmax)[:, None]
    lane = np.arange(64)[None, :]
    tile = k * world + rank
    tyy, txx = tile // tx, tile % tx
    i = txx * 8 + (lane & 7)
    j = tyy * 8 + (lane >> 3)
    ok = (tile < tx * ty) & (i < nx) & (j < ny)
    return np.where(ok, j * nx + i, -1).reshape(-1).astype(np.int64)


def scatter_numpy(gathered, nx, ny, world):
    """gathered: [world, tiles_max*64, 4] -> framebuffer [ny, nx, 4] (host-side mirror of crt_scatter_shards)."""
    fb = np.zeros((ny * nx, 4), dtype=np.float32)
    for r in range(world):
        idx = compact_pixel_index(nx, ny, r, world)
        ok = idx >= 0
        fb[idx[ok]] = np.asarray(gathered[r]).reshape(-1, 4)[ok]
    return fb.reshape(ny, nx, 4)


def gather_image(compact, nx, ny, ctx=None, fb=None):
    """Gathers every rank's compact tensor on rank 0 and scatters it into a framebuffer.

    compact: torch tensor [tiles_max*64*4] float32 (CUDA with a Context, CPU without).
    Returns the [ny, nx, 4] framebuffer tensor on rank 0, None elsewhere.
    With world_size 1 (or no process group) no collective is issued.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if world > 1:
        # RCCL gathers device buffers directly; under gloo (several ranks rehearsing on one GPU, or CPU tensors)
        # the shard goes through host memory
        staged = compact.is_cuda and dist.get_backend() == "gloo"
        send = compact.cpu() if staged else compact
        gathered = torch.empty((world, send.numel()), dtype=send.dtype, device=send.device) if rank == 0 else None
        parts = [gathered[r] for r in range(world)] if rank == 0 else None
        dist.gather(send, gather_list=parts, dst=0)
        if rank != 0:
            return None
        if staged:
            gathered = gathered.to(compact.device)
    else:
        gathered = compact.unsqueeze(0)
    if compact.is_cuda:
        if ctx is None:
            raise ValueError("a Context is required to scatter device buffers")
        if fb is None:
            fb = torch.empty((ny, nx, 4), dtype=torch.float32, device=compact.device)
        stream = torch.cuda.current_stream(compact.device).cuda_stream
        ctx.scatter_shards(nx, ny, world, gathered.data_ptr(), fb.data_ptr(), stream)
        return fb
    out = scatter_numpy(gathered.numpy().reshape(world, -1, 4), nx, ny, world)
    return torch.from_numpy(out)
