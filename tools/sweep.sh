#!/bin/bash
# overlapped step time for a list of "ENV=.. ENV=.." settings (one per argument)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
  echo "[$v] $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['value'],d['ms_per_step'])") $(grep -E 'latent|row_kernel<train>' /tmp/o.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
