# usage (GPU box, repo root): bash tools/prof_traffic.sh <tag> [bench args]  -- HBM-side traffic per kernel, separate passes
TAG=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_write.log 2>&1 || exit 1
python3 - $TAG <<'PY'
import csv, re, sys, json
from collections import defaultdict
tag = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
for name in ("fetch", "write"):
    for r in csv.DictReader(open(f"gpurun_out/{tag}_{name}/pmc_counter_collection.csv")):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m:
            tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in tot.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out[k] = {"FETCH_SIZE_kb": f, "WRITE_SIZE_kb": w, "hbm_bytes_upper": (2 * f + w) * 1024}
    print("%-24s FETCH_SIZE %12.0f KB  WRITE_SIZE %12.0f KB  (2*F+W)*1024 = %.3f GB" % (k, f, w, (2 * f + w) * 1024 / 1e9))
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
PY
