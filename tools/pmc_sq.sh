# SQ counters per kernel, serial streams (no overlap), for the general and the fused row paths
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FFM_ENGINE_SERIAL=1
for fu in 0 1; do
export FFM_ENGINE_FUSE=$fu
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d gpurun_out/pmc_sq_fuse$fu -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/pmc_sq_fuse$fu.log 2>&1
done
