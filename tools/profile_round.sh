#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel-trace + stats of the default
# bench command, then (separate passes, as the microarch guide prescribes) the HBM byte counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/pmc_fetch.json 2>/dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $OUT/pmc_write.json 2>/dev/null
find $OUT -name "*.csv" | head -20
cat $OUT/bench.json
