#!/bin/bash
# Kernel times ALONE (serial streams) of one BASELINE configuration for (library, environment) variants.
# usage: tools/ab_cfg_serial.sh c4 "tree|" "head|ENV=.." ...
cd "$GRAFT_REPO_ROOT"
cfg=$1; shift
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env FFM_ENGINE_SERIAL=1 $L $envs python3 bench.py --config $cfg --no-cpu-baseline --steps 20 --warmup 4 --resident-only > /tmp/abs.json 2> /tmp/abs.err
  echo "[$cfg $v] $(python3 -c "import json;d=json.load(open('/tmp/abs.json'));print('serial ms', d['ms_per_step'])") | $(grep -E 'row_kernel<train>|latent|refresh|bias|linear' /tmp/abs.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
