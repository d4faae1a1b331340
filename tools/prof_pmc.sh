# usage: bash tools/prof_pmc.sh <tag> [bench args...]   (runs on the GPU box from the repo root)
set -e
TAG=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/${TAG}_pmc1 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_pmc1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/${TAG}_pmc2 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_pmc2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc1/pmc_counter_collection.csv gpurun_out/${TAG}_pmc2/pmc_counter_collection.csv
