# usage (GPU box, repo root): bash tools/profile_round.sh <tag>
# rocprofv3 kernel-trace stats of the default bench workload, PMC passes (each its own run), HBM-side traffic.
# Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=$1
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -o kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-rows > gpurun_out/${TAG}_kt.log 2>&1 || exit 1
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
grep '"metric"' gpurun_out/${TAG}_kt.log > gpurun_out/${TAG}_bench_under_rocprof.json
bash tools/prof_pmc.sh ${TAG} > gpurun_out/${TAG}_pmc_summary.txt 2>&1 || exit 1
bash tools/prof_pmc3.sh ${TAG} >> gpurun_out/${TAG}_pmc_summary.txt 2>&1 || exit 1
bash tools/prof_pmc4.sh ${TAG} >> gpurun_out/${TAG}_pmc_summary.txt 2>&1 || exit 1
bash tools/prof_traffic.sh ${TAG} > gpurun_out/${TAG}_traffic.txt 2>&1 || exit 1
cat gpurun_out/${TAG}_traffic.txt
cat gpurun_out/${TAG}_kernel_stats.csv
# what one rank of 2 / 4 / 8 runs (tools/shard_bench.py), kernel-trace stats only
for W in 2 4 8; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_shard${W} -o kt -- python3 tools/shard_bench.py $W 1 10 > gpurun_out/${TAG}_shard${W}.log 2>&1 || exit 1
  cp gpurun_out/${TAG}_shard${W}/kt_kernel_stats.csv gpurun_out/${TAG}_shard_1of${W}_kernel_stats.csv
  cat gpurun_out/${TAG}_shard${W}.log | tail -1
done
