#!/bin/bash
# Update-kernel times ALONE (serial streams) for environment variants.  usage: tools/ab_serial.sh "ENV=.. ENV=.." ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  env FFM_ENGINE_SERIAL=1 $v python3 bench.py --no-cpu-baseline --steps 20 --warmup 4 --resident-only $BENCH_ARGS > /tmp/abs.json 2> /tmp/abs.err
  echo "[$v] $(python3 -c "import json;d=json.load(open('/tmp/abs.json'));print('serial ms', d['ms_per_step'])") | $(grep -E 'row_kernel<train>|latent|refresh' /tmp/abs.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
