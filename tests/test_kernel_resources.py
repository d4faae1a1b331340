"""Registers, scratch and LDS of the built kernels, read from the code object inside libcrt_hip.so (no GPU needed).

The per-pixel trace kernel runs five waves per SIMD only while it needs at most 96 registers (they are handed out in
eights: 97 means 104, and 5 x 104 exceeds the SIMD's 512), and five workgroups per CU only while a workgroup's LDS
stays under 32 KB; both have been lost to innocent-looking source changes before
(the compiler keeping forty table look-ups alive, a second table in LDS), at a cost of 20-30 % of the kernel's time.
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "cudaraytracer_amd" / "libcrt_hip.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def _kernel_table(tmp_path):
    fat, dev = tmp_path / "fat.bin", tmp_path / "dev.co"
    subprocess.run([str(LLVM / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(LIB), str(tmp_path / "host.so")],
                   check=True, capture_output=True)
    # the section holds one offload bundle per translation unit, back to back
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    table = {}
    for n, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part = tmp_path / f"bundle{n}.bin"
        part.write_bytes(blob[lo:hi])
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={part}", f"--output={dev}", "--unbundle"], check=True, capture_output=True)
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(dev)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            def field(key):
                m = re.search(r"\." + key + r":\s*(\S+)", blk)
                return m.group(1) if m else None
            table[field("name")] = {k: int(field(k)) for k in ("vgpr_count", "group_segment_fixed_size", "private_segment_fixed_size")}
    return table


@pytest.mark.skipif(not (LLVM / "llvm-readelf").exists() or shutil.which("true") is None, reason="LLVM tools of ROCm not installed")
def test_trace_kernel_keeps_its_occupancy(tmp_path):
    assert LIB.exists(), "build libcrt_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    table = _kernel_table(tmp_path)
    per_pixel = [v for k, v in table.items() if "wf_trace_kernelILb0" in k]
    coop = [v for k, v in table.items() if "wf_trace_kernelILb1" in k]
    assert len(per_pixel) == 1 and len(coop) == 1, sorted(table)
    pp, co = per_pixel[0], coop[0]
    assert pp["vgpr_count"] <= 96, f"per-pixel trace kernel: {pp['vgpr_count']} registers, five waves per SIMD need <= 96"
    assert pp["private_segment_fixed_size"] == 0, f"per-pixel trace kernel spills: {pp['private_segment_fixed_size']} bytes of scratch"
    assert pp["group_segment_fixed_size"] <= 28 * 1024, "per-pixel trace workgroup: static LDS leaves no room for five workgroups per CU"
    assert co["vgpr_count"] <= 128 and co["private_segment_fixed_size"] <= 64, co
    assert co["group_segment_fixed_size"] <= 36 * 1024, "cooperative trace workgroup: four per CU need <= 40 KB with the primitive table"
    for name, v in table.items():
        if "wf_" in name or "render_kernel" in name:
            assert v["vgpr_count"] <= 128, (name, v)  # nothing below four waves per SIMD
