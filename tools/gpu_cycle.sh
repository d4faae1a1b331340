# usage: bash tools/gpu_cycle.sh <tag>   -- parity tests, then bench (gl1, no extras), then PMC summary
set -e
TAG=$1
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/parity_$TAG.log 2>&1 || { tail -30 gpurun_out/parity_$TAG.log; exit 1; }
tail -1 gpurun_out/parity_$TAG.log
timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-rows > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'Msamples/s', d['value'], 'roofline', d['roofline']['frac'])"
bash tools/prof_pmc.sh $TAG 2>&1 | tail -19
