set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-rows --no-cpu > gpurun_out/prof_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_pmc1 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu > gpurun_out/prof_pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU_TRANS --output-format csv -d gpurun_out/prof_pmc2 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-rows --no-cpu > gpurun_out/prof_pmc2.log 2>&1
find gpurun_out -name "*.csv" | head -20
