# usage (GPU box): bash tools/variant_bench.sh <lib suffix>...   -- bench per-kernel times with alternative builds of the library
for v in "$@"; do
  CRT_HIP_LIB=$PWD/cudaraytracer_amd/libcrt_hip_$v.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu --no-rows 2>/dev/null > /tmp/variant_$v.json || { echo "$v failed"; continue; }
  python3 - $v <<'PY'
import json, sys
d = json.load(open(f"/tmp/variant_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["ms_per_step"], 2), {a: round(b["ms"], 2) for a, b in d["roofline"]["kernels"].items()})
PY
done
