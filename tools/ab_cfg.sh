#!/bin/bash
# One BASELINE configuration for (library, environment) variants on one box.
# usage: tools/ab_cfg.sh c3 "tree|FFM_SUPER_MIN=1024" "tree" ...   (lib = tree or exp_libs/lib<name>.so)
cd "$GRAFT_REPO_ROOT"
cfg=$1; shift
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env $L $envs python3 bench.py --config $cfg --no-cpu-baseline --steps 100 --warmup 10 > /tmp/ab.json 2> /tmp/ab.err
  echo "[$cfg $v] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|update|refresh' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
