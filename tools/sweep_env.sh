#!/bin/bash
# Compares settings of the library's environment knobs ("A=1 B=2" per argument; "" = defaults):
# 1-GPU host + resident legs and one rank's share of the 8-GPU job.
cd "$GRAFT_REPO_ROOT"
one() { env $1 python3 bench.py --no-cpu-baseline --steps 150 --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[%s]' % sys.argv[1], '1gpu host', d['ms_per_step'], 'resident', d.get('resident',{}).get('ms_per_step'))" "$1"; }
emu() { env $1 python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('[%s]' % sys.argv[1], 'emu8 host', d['ms_per_step'], 'resident', d.get('resident',{}).get('ms_per_step'))" "$1"; }
for v in "$@"; do one "$v"; emu "$v"; done
