#!/bin/bash
# The staging thread on / off on one box: C5, C2, C3, C4 H2D-inclusive and resident rates.
cd "$GRAFT_REPO_ROOT"
grep -m1 "model name" /proc/cpuinfo; uptime
for t in 1 0 1 0; do
  for c in c5 c2 c3 c4; do
    FFM_STAGE_THREAD=$t python3 bench.py --config $c --no-cpu-baseline --no-profile --steps 150 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('thread=%s %s host ms %.4f rows/s %d | resident ms %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['value'], d.get('resident',{}).get('ms_per_step')))" $t $c
  done
done
