# usage (GPU box, repo root): bash tools/coop_cycle.sh  -- cooperative-trace tests, then shard timings in both trace modes
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cooperative or trace_modes" --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
for world in 8 4 2 1; do
  for mode in 1 2; do
    timeout -k 10 120 python -u tools/shard_bench.py $world 1 5 $mode || exit 1
  done
done
