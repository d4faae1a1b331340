"""Host-side checks that need no GPU: the C-ABI library loads and exports what include/crt_hip.h declares,
struct layouts agree between C and ctypes, the host scene classes reproduce create_world, tile sharding."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cudaraytracer_amd as crt
from cudaraytracer_amd import abi, distributed as shard

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "crt_hip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = crt.lib()
    names = declared_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), f"libcrt_hip.so does not export {n}"
    assert sorted(crt.ABI_SYMBOLS) == names


def test_ctypes_layouts_match_the_header(tmp_path):
    structs = ["crt_material", "crt_vertex", "crt_geometry", "crt_light", "crt_camera", "crt_scene_desc",
               "crt_settings", "crt_stats"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "crt_hip.h"\nint main(void){' +
                   "".join(f'printf("{s} %zu\\n", sizeof({s}));' for s in structs) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True,
                                                       text=True).stdout.splitlines())
    for s in structs:
        assert C.sizeof(getattr(abi, s)) == int(out[s]), s


def test_host_scene_classes_reproduce_create_world(oracle):
    """CornellBox (C++ scene classes -> flatten) must equal the restated create_world byte for byte."""
    a, b = crt.cornell_box(50), oracle.cornell(50)
    assert bytes(a.camera) == bytes(b.camera)
    for i in range(8):
        assert bytes(a.geometry[i]) == bytes(b.geometry[i]), i
    assert bytes(a.lights[0]) == bytes(b.lights[0])
    assert crt.cornell_box(7).lights[0].shadowSamples == 7


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(crt.CrtError):
        crt.Context(0)


@pytest.mark.parametrize("nx,ny,world", [(720, 720, 1), (720, 720, 2), (720, 720, 8), (37, 21, 3), (8, 8, 4), (100, 7, 5)])
def test_tile_sharding_covers_every_pixel_once(nx, ny, world):
    L = crt.lib()
    seen = np.zeros(nx * ny, dtype=np.int32)
    kmax = shard.shard_tiles_max(nx, ny, world)
    assert kmax == L.crt_shard_tiles_max(nx, ny, world)
    total = 0
    for r in range(world):
        assert shard.shard_tiles(nx, ny, r, world) == L.crt_shard_tiles(nx, ny, r, world)
        idx = shard.compact_pixel_index(nx, ny, r, world)
        assert idx.size == kmax * 64
        np.add.at(seen, idx[idx >= 0], 1)
        total += shard.shard_tiles(nx, ny, r, world)
    assert (seen == 1).all()
    assert total == ((nx + 7) // 8) * ((ny + 7) // 8)
    assert L.crt_shard_tiles(nx, ny, world, world) == 0  # rank out of range


def test_scatter_numpy_inverts_compact_layout():
    nx, ny, world = 29, 19, 3
    img = np.random.default_rng(0).random((ny, nx, 4), dtype=np.float32)
    gathered = []
    for r in range(world):
        idx = shard.compact_pixel_index(nx, ny, r, world)
        comp = np.zeros((idx.size, 4), np.float32)
        comp[idx >= 0] = img.reshape(-1, 4)[idx[idx >= 0]]
        gathered.append(comp)
    assert np.array_equal(shard.scatter_numpy(np.stack(gathered), nx, ny, world), img)
