#!/bin/bash
# Kernel times ALONE (serial streams) for (library, environment) variants.  usage: tools/ab_lib_serial.sh "lib|ENV=.." ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env FFM_ENGINE_SERIAL=1 $L $envs python3 bench.py --no-cpu-baseline --steps 20 --warmup 4 --resident-only $BENCH_ARGS > /tmp/abs.json 2> /tmp/abs.err
  echo "[$v] $(python3 -c "import json;d=json.load(open('/tmp/abs.json'));print('serial ms', d['ms_per_step'])") | $(grep -E 'row_kernel<train>|latent|refresh' /tmp/abs.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
