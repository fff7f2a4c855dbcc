#!/bin/bash
# SQ counters per kernel, serial streams, for ONE rank's share of an 8-GPU job (bench.py --emulate-shards 8).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-emu8}
export FFM_ENGINE_SERIAL=1
ARGS="--emulate-shards 8 --emulate-rank 3 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d gpurun_out/pmc_${tag}_a -o p --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/pmc_${tag}_b -o p --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${tag}_b.log 2>&1
