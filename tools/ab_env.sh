#!/bin/bash
# C5 step for (library, environment) variants on one box.  usage: tools/ab_env.sh "lib|ENV=.. ENV=.." ...
# lib = tree or the name of exp_libs/lib<name>.so; EXP_BENCH_ARGS="--config c3" for another workload
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env $L $envs python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 $EXP_BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err
  echo "[$v] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|update|refresh' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
