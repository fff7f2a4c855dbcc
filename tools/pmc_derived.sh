#!/bin/bash
# Derived busy / stall metrics per kernel with serial streams.  usage: tools/pmc_derived.sh <tag> "<metric list>" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; metrics=$2; shift; shift
export FFM_ENGINE_SERIAL=1
rocprofv3 --kernel-trace --pmc $metrics -d gpurun_out/pmcd_${tag} -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --resident-only --no-eval "$@" > gpurun_out/pmcd_${tag}.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/pmcd_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if any(x in k for x in ('fill','init','verify','rocprim','at::')): continue
    print(k.ljust(42), ' '.join('%s=%.1f' % (n, sum(x)/len(x)) for n,x in sorted(v.items())))
PY
tail -3 gpurun_out/pmcd_${tag}.log | cut -c1-300
