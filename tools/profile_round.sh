# usage (on the GPU box, repo root): bash tools/profile_round.sh r01
# kernel trace + stats of the default bench command, then the PMC passes; summaries land in gpurun_out/
set -e
TAG=$1
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -o kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-rows > gpurun_out/${TAG}_kt.log 2>&1
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_kt.log > gpurun_out/${TAG}_bench_under_rocprof.json
bash tools/prof_pmc.sh ${TAG} > gpurun_out/${TAG}_pmc_summary.txt 2>&1
bash tools/prof_pmc3.sh ${TAG} >> gpurun_out/${TAG}_pmc_summary.txt 2>&1
# HBM traffic, separate passes
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu > gpurun_out/${TAG}_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu > gpurun_out/${TAG}_write.log 2>&1
python3 - <<PY
import csv
for name in ("fetch", "write"):
    for r in csv.DictReader(open(f"gpurun_out/${TAG}_{name}/pmc_counter_collection.csv")):
        if "render_kernel" in r["Kernel_Name"]:
            print(name, r["Counter_Name"], r["Counter_Value"])
PY
cat gpurun_out/${TAG}_kernel_stats.csv
