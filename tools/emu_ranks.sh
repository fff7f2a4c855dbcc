#!/bin/bash
# One rank's share of the 8-GPU job on one GPU, every rank in turn (the last lines of tools/profile_round.sh alone).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${1:-r06}; mkdir -p $OUT
for r in 0 1 2 3 4 5 6 7; do python3 bench.py --emulate-shards 8 --emulate-rank $r --steps 40 --warmup 6 --no-profile > $OUT/emu8_rank$r.json 2>/dev/null; done
python3 bench.py --emulate-shards 8 --emulate-rank 3 --steps 40 --warmup 6 --scaling strong --no-profile > $OUT/emu8_strong_rank3.json 2>/dev/null
python3 -c "
import json
for r in range(8):
    d=json.load(open('$OUT/emu8_rank%d.json'%r)); print(r, d['ms_per_step'], d.get('resident',{}).get('ms_per_step'))
d=json.load(open('$OUT/emu8_strong_rank3.json')); print('strong', d['ms_per_step'], d['value'])"
