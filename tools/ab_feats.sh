#!/bin/bash
# C5 with smaller models on one box: how much of the step is address translation / HBM locality
# of the gathers into a 247 GB tensor.  usage: tools/ab_feats.sh <n_feats> ...
cd "$GRAFT_REPO_ROOT"
for nf in "$@"; do
  python3 bench.py --no-cpu-baseline --no-eval --steps 60 --warmup 6 --n-feats $nf > /tmp/ab.json 2> /tmp/ab.err
  echo "[n_feats $nf] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|update|refresh' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
