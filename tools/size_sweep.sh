# usage (GPU box): bash tools/size_sweep.sh  -- per-kernel time of the gl1 workload at several image sizes
for n in 256 512 640 720 1024; do
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu --no-rows --nx $n --ny $n 2>/dev/null > /tmp/sweep_$n.json || exit 1
  python3 - $n <<'PY'
import json, sys
n = int(sys.argv[1])
d = json.load(open(f"/tmp/sweep_{n}.json"))
k = d["roofline"]["kernels"]
print(n, round(d["ms_per_step"], 2), {a: round(b["ms"], 2) for a, b in k.items()}, "trace ns/px", round(k["trace"]["ms"] * 1e6 / (n * n), 1))
PY
done
