cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FFM_ENGINE_SERIAL=1
for d in zipf uniform; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM -d gpurun_out/pmc_sq_$d -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --dist $d > gpurun_out/pmc_sq_$d.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f_$d -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --dist $d > gpurun_out/pmc_f_$d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w_$d -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --dist $d > gpurun_out/pmc_w_$d.log 2>&1
done
find gpurun_out -name "*.csv" | head -20
