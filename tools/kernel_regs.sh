#!/bin/bash
# Registers, LDS and scratch of every kernel in the built HIP objects (from the code objects' metadata notes):
#   bash tools/kernel_regs.sh [object ...]      default: the wavefront and persistent kernels' objects
cd "$(dirname "$0")/../cudaraytracer_amd/csrc" || exit 1
OBJS=${@:-crt_wavefront.o crt_render.o}
for o in $OBJS; do
  tmp=$(mktemp -d)
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $o $tmp/host.o || exit 1
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle || exit 1
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    name = re.sub(r"^_ZN3crt\d+_GLOBAL__N_1", "", name)
    print("%-60s vgpr %4s agpr %4s sgpr %4s lds %6s scratch %5s" % (name[:60], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
'
  rm -rf $tmp
done
