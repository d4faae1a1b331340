# usage (GPU box, repo root): bash tools/bench_cycle.sh <tag> [bench args]  -- bench under rocprofv3 kernel-trace only
TAG=$1; shift
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG} -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-rows "$@" > gpurun_out/${TAG}.log 2>&1 || { tail -5 gpurun_out/${TAG}.log; exit 1; }
grep '"metric"' gpurun_out/${TAG}.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],2), 'Gs/s', round(d['value']/1e3,2), 'frac', round(d['roofline']['frac'],4), d['work_per_step'])"
python3 - gpurun_out/${TAG}/kt_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel|__amd\w+)", r["Name"])
    print("%-28s calls %5s  total %9.3f ms  avg %9.3f ms" % (m.group(1) if m else r["Name"][:28], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
