#!/bin/bash
# Byte budget of the FFM row kernel (VERDICT r05 "missing" #3): HBM-side byte counters and L2 hit
# rate of ffm_row_kernel<train> for the experiment builds of kernels_row.h (FFM_ROW_EXP: each drops
# one class of memory accesses -- wrong results, timing / traffic only), on the workload given.
# usage: tools/row_budget.sh <out tag> "<bench args>" <lib> [<lib> ...]     (lib = tree | name in exp_libs/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; args=$2; shift 2
OUT=gpurun_out/row_budget_$tag
mkdir -p $OUT
for lib in "$@"; do
  if [ "$lib" != "tree" ]; then export FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so; else unset FFM_ENGINE_LIB; fi
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    d=$OUT/${lib}_$(echo $ctr | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o b -- python3 bench.py $args --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-resident --no-eval > /dev/null 2> $d.err
  done
done
python3 tools/row_budget.py $OUT "$@" | tee $OUT/summary.txt
