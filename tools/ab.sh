#!/bin/bash
# A/B of the in-tree library against exp_libs/<name>.so builds on one box: C5 step (H2D-inclusive
# and resident) with the per-kernel HIP-event table, C2/C3/C4 rates, one emulated 8-GPU rank.
# usage: tools/ab.sh [name ...]   ("tree" = the in-tree build, always first)
cd "$GRAFT_REPO_ROOT"
run() {
  python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > /tmp/ab.json 2> /tmp/ab.err
  echo "[$1] c5 $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|latent|refresh' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  for c in c2 c3 c4; do python3 bench.py --config $c --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[%s] %s host rows/s %d resident %s' % (sys.argv[1], sys.argv[2], d['value'], d.get('resident',{}).get('value')))" "$1" $c; done
  if [ -n "$AB_EMU" ]; then python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[%s] emu8 rank3 host ms %s resident ms %s' % (sys.argv[1], d['ms_per_step'], d.get('resident',{}).get('ms_per_step')))" "$1"; fi
}
run tree
for l in "$@"; do FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$l.so run $l; done
