set -e
TAG=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VSKIPPED GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc3 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu "$@" > gpurun_out/${TAG}_pmc3.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc3/pmc_counter_collection.csv
