#!/bin/bash
# Kernel timeline of the H2D-inclusive leg of one emulated rank of an 8-GPU job.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-trace_emu_host}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 16 --warmup 4 --no-cpu-baseline --no-resident --no-profile > $OUT/bench.json 2> $OUT/bench.err
python3 tools/step_gaps.py $OUT/trace > $OUT/gaps.txt 2>&1
