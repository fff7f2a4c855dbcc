#!/bin/bash
# Kernel + memory-copy timeline of the H2D-inclusive leg of bench.py (no counters in this pass).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-trace_host}
mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-resident --no-profile > $OUT/bench.json 2> $OUT/bench.err
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
ls $OUT/trace/* | head
