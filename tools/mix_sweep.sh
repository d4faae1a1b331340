# usage (GPU box): bash tools/mix_sweep.sh WORLD "p1 p2 ..."  -- shard timing with that share (per mille) of the units cooperative
WORLD=$1
for p in $2; do
  echo -n "permille $p: "
  CRT_COOP_PERMILLE=$p timeout -k 10 100 python -u tools/shard_bench.py $WORLD 1 5 0 2>&1 | tail -1
done
