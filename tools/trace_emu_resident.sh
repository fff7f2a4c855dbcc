#!/bin/bash
# Kernel timeline of the HBM-resident leg of one emulated rank of an 8-GPU job (compare with
# tools/trace_emu_host.sh: what the block upload costs the kernels it runs beside).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-trace_emu_resident}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 16 --warmup 4 --no-cpu-baseline --resident-only --no-profile > $OUT/bench.json 2> $OUT/bench.err
python3 tools/step_gaps.py $OUT/trace > $OUT/gaps.txt 2>&1
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
