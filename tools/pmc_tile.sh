#!/bin/bash
# SQ + LDS counters of the update kernels with serial streams.  usage: tools/pmc_tile.sh <tag>   (env is inherited)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
export FFM_ENGINE_SERIAL=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU -d gpurun_out/pmc_${tag}_a -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only $BENCH_ARGS > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/pmc_${tag}_b -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only $BENCH_ARGS > gpurun_out/pmc_${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d gpurun_out/pmc_${tag}_c -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only $BENCH_ARGS > gpurun_out/pmc_${tag}_c.log 2>&1
python3 tools/pmc_sq_summary.py $tag
python3 - $tag <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("gpurun_out/pmc_%s_c/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, v in agg.items():
    if 'update' in k or 'row_kernel' in k:
        print(k, {n: round(x / max(cnt[k][n], 1) / 1e6, 2) for n, x in v.items()})
PY
