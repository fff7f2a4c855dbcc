#!/bin/bash
# A/B of library builds on config 4 (FM): rates + the per-kernel HIP-event table.
cd "$GRAFT_REPO_ROOT"
run() {
  python3 bench.py --config c4 --no-cpu-baseline --steps 200 > /tmp/ab.json 2> /tmp/ab.err
  echo "[$1] c4 $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host rows/s', round(d['value']), 'ms', d['ms_per_step'], 'resident', round(d.get('resident',{}).get('value',0)), d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|latent|linear|bias|tmp_grad' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
}
run tree
for l in "$@"; do FFM_ENGINE_LIB=$GRAFT_REPO_ROOT/exp_libs/lib$l.so run $l; done
