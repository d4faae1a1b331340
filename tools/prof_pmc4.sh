# usage: bash tools/prof_pmc4.sh <tag> [bench args...]   (GPU box, repo root) -- occupancy and LDS bank conflicts, one pass
set -e
TAG=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/${TAG}_pmc4 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_pmc4.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc4/pmc_counter_collection.csv
