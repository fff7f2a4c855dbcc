#!/bin/bash
# Kernel timeline of the HBM-resident leg, one GPU (tools/timeline.py, tools/step_gaps.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-trace_resident}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --resident-only --no-profile > $OUT/bench.json 2> $OUT/bench.err
python3 tools/step_gaps.py $OUT/trace > $OUT/gaps.txt 2>&1
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
