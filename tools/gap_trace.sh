#!/bin/bash
# Host submit times against queue start times for the H2D leg: one rocprofv3
# --hip-trace --kernel-trace of bench.py (no counters), joined per dispatch by tools/gap_trace.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gap_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/h2d -o bench -- python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-resident --no-profile "$@" > $OUT/h2d.json 2> $OUT/h2d.err
python3 tools/gap_trace.py $OUT/h2d > $OUT/h2d_gaps.txt 2>&1
FFM_HOST_TIMING=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-profile "$@" > $OUT/host_timing.json 2> $OUT/host_timing.err
tail -40 $OUT/h2d_gaps.txt
