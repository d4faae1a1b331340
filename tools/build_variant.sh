# usage: bash tools/build_variant.sh <suffix> <hipcc flags...>  -- builds cudaraytracer_amd/libcrt_hip_<suffix>.so with the
# given flags appended to the kernels' compile line (later flags win), objects under /tmp.  A measuring aid:
# CRT_HIP_LIB=<that file> makes the Python binding load it instead of libcrt_hip.so.
set -e
SUF=$1; shift
HERE=$(cd "$(dirname "$0")/../cudaraytracer_amd/csrc" && pwd)
B=/tmp/crt_variant_$SUF; mkdir -p $B
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DCRT_FAST_RCP=1 -DCRT_WAVES_PER_EU=4 -fno-slp-vectorize -Wno-unused-function $*"
/opt/rocm/bin/hipcc $FLAGS -c $HERE/crt_render.hip -o $B/crt_render.o &
/opt/rocm/bin/hipcc $FLAGS -c $HERE/crt_wavefront.hip -o $B/crt_wavefront.o &
/opt/rocm/bin/hipcc $FLAGS -x hip -c $HERE/crt_api.cpp -o $B/crt_api.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $HERE/../libcrt_hip_$SUF.so $B/crt_render.o $B/crt_wavefront.o $B/crt_api.o $HERE/crt_scene.o $HERE/crt_hdr.o $HERE/crt_ltc_blob.o
ls -la $HERE/../libcrt_hip_$SUF.so
