#!/bin/bash
# Why is the H2D-inclusive leg slower than the resident one on some boxes?  One box: both legs'
# step time, the submitting thread's time per call (FFM_HOST_TIMING), the kernel timeline of the
# host leg with the idle gap before every refresh, and the CPU model.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-hostdiag}
mkdir -p $OUT
grep -m1 "model name" /proc/cpuinfo > $OUT/cpu.txt; nproc >> $OUT/cpu.txt; uptime >> $OUT/cpu.txt
python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 --no-profile > $OUT/bench.json 2> $OUT/bench.err
FFM_HOST_TIMING=1 python3 bench.py --no-cpu-baseline --no-resident --no-profile --steps 200 > $OUT/timing.json 2> $OUT/timing.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-resident --no-profile > $OUT/trace_bench.json 2> $OUT/trace_bench.err
python3 tools/step_gaps.py $OUT/trace > $OUT/gaps.txt 2>&1
python3 tools/timeline.py $OUT/trace > $OUT/timeline.txt 2>&1
cat $OUT/cpu.txt
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print('host ms', d['ms_per_step'], 'resident ms', d['resident']['ms_per_step'])
d=json.load(open('$OUT/trace_bench.json')); print('traced host ms', d['ms_per_step'])"
grep "host timing" $OUT/timing.err
tail -2 $OUT/gaps.txt | cut -c1-600
head -45 $OUT/timeline.txt
