# usage (GPU box): bash tools/variant_cycle.sh <lib suffix>...  -- per variant: bench line and the world-8 shard timing
for v in "$@"; do
  bash tools/variant_bench.sh $v
  CRT_HIP_LIB=$PWD/cudaraytracer_amd/libcrt_hip_$v.so timeout -k 10 100 python -u tools/shard_bench.py 8 1 5 1 2>&1 | tail -1
done
