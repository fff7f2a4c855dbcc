#!/bin/bash
# SQ / instruction-cache counters per kernel with serial streams (no overlap).
# usage: tools/pmc_sq.sh <tag> [hipcc -D flags of an experimental build]   (env is inherited;
#        PMC_BENCH_ARGS="--config c4" for another workload)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
if [ -n "$*" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $* ftrl-ffm_amd/csrc/engine.hip -o /tmp/libexp_$tag.so 2>/dev/null || { echo "build failed"; exit 1; }
  export FFM_ENGINE_LIB=/tmp/libexp_$tag.so
fi
export FFM_ENGINE_SERIAL=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d gpurun_out/pmc_${tag}_a -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only $PMC_BENCH_ARGS > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/pmc_${tag}_b -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only $PMC_BENCH_ARGS > gpurun_out/pmc_${tag}_b.log 2>&1
