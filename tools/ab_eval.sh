#!/bin/bash
# Evaluation leg (predict + logloss, H2D included / resident) for environment variants on one box.
# usage: tools/ab_eval.sh "ENV=.. ENV=.. |<bench args>" ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  envs=${v%%|*}; args=${v#*|}
  env $envs python3 bench.py --no-cpu-baseline --no-profile $args > /tmp/ev.json 2> /tmp/ev.err
  echo "[$v] $(python3 -c "import json;d=json.load(open('/tmp/ev.json'));e=d['eval'];print('train ms', d['ms_per_step'], 'eval H2D', round(e['value']/1e6,2), 'M rows/s', e['ms_per_step'], 'ms  frac', e['roofline_frac'], ' resident', round(e['resident']/1e6,2), e['roofline_frac_resident'])")"
done
