#!/bin/bash
# like sweep.sh, for one rank's share of an N-GPU job (bench.py --emulate-shards N)
cd "$GRAFT_REPO_ROOT"
N=${EMU_N:-8}
for v in "$@"; do
  env $v python bench.py --emulate-shards $N --steps 8 --warmup 2 --no-cpu-baseline --n-blocks 4 > /tmp/o.json 2> /tmp/o.err
  echo "[$v] $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['value'],d['ms_per_step'])") $(grep -E 'latent|row_kernel<train>|refresh' /tmp/o.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
