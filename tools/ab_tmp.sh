#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  echo -n "$*  "
  E=(); A=(); for t in "$@"; do case $t in *=*) E+=("$t");; *) A+=("$t");; esac; done; env "${E[@]}" python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-profile "${A[@]}" 2> /tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h2d %.4f ms  resident %.4f ms loss %s'%(d['ms_per_step'], d['resident']['ms_per_step'], d.get('train_logloss')))"
}
for rep in 1 2 3; do
run A=0
run FFM_XSWAP=1
done
for rep in 1 2; do
run A=0 --config c3
run FFM_XSWAP=1 --config c3
run A=0 --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6
run FFM_XSWAP=1 --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6
done
