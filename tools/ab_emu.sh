#!/bin/bash
# One emulated 8-GPU rank (bench.py --emulate-shards 8) for (library, environment) variants on one box.
# usage: tools/ab_emu.sh <rank> "tree|FFM_SUPER_MIN=1024" tree ...
cd "$GRAFT_REPO_ROOT"
rank=$1; shift
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env $L $envs python3 bench.py --emulate-shards 8 --emulate-rank $rank --steps 40 --warmup 6 --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err
  echo "[rank $rank $v] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'job rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|update|refresh|tmp_grad' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
