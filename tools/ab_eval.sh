#!/bin/bash
# The evaluation leg (predict + logloss) for (library, environment) variants on one box, with the
# leg's kernel table.  usage: tools/ab_eval.sh "lib|ENV=.. ENV=.." ...   [EXP_BENCH_ARGS="--config c3"]
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=${v%%|*}; envs=${v#*|}; [ "$envs" == "$v" ] && envs=""
  L=""; [ "$lib" != "tree" ] && L="FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$lib.so"
  env $L $envs python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 $EXP_BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err
  echo "[$v] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'))['eval'];print('eval H2D rows/s', round(d['value']), 'resident', round(d['resident']), 'frac', d['roofline_frac'], d['roofline_frac_resident'], 'logloss', d['logloss'])")"
  grep -A12 '^\[eval, resident\]' /tmp/ab.err | grep -E 'predict|loss_sum' 
done
