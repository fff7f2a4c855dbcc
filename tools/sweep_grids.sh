#!/bin/bash
# Grid-size sweep of the update kernels (env knobs of the library), resident leg, one box.
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --no-cpu-baseline --resident-only --steps 150 --no-profile | python3 -c "
import json,sys; d=json.load(sys.stdin); print(sys.argv[1], d['ms_per_step'])" "$1"; }
run "default"
for s in 384 512 1024 1536; do FFM_GRID_SINGLE=$s run "single=$s"; done
for h in 1024 3072 4096; do FFM_GRID_HOT=$h run "hot=$h"; done
for h in 1024 3072; do FFM_GRID_HUGE=$h run "huge=$h"; done
for m in 384 1536; do FFM_GRID_SMALL=$m run "small=$m"; done
FFM_GRID_SINGLE=512 FFM_GRID_HOT=3072 run "single=512 hot=3072"
run "default again"
