#!/bin/bash
# Compares experimental builds of the library (exp_libs/*.so, built here with extra -D flags)
# against the in-tree one: 1-GPU resident + host legs and one rank's share of the 8-GPU job.
cd "$GRAFT_REPO_ROOT"
one() { python3 bench.py --no-cpu-baseline --steps 150 --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(sys.argv[1], '1gpu host', d['ms_per_step'], 'resident', d.get('resident',{}).get('ms_per_step'))" "$1"; }
emu() { python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(sys.argv[1], 'emu8 host', d['ms_per_step'], 'resident', d.get('resident',{}).get('ms_per_step'))" "$1"; }
one default; emu default
for l in "$@"; do
  export FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/$l.so
  one $l; emu $l
done
