# usage: bash tools/prof_pmc_quick.sh <tag>   (GPU box, repo root) -- one pass: instruction counts and stall share of every kernel
set -e
TAG=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/${TAG}_pmcq -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_pmcq.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmcq/pmc_counter_collection.csv
