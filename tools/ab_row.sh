#!/bin/bash
# Row-kernel shapes on one box: workgroup size x LDS parked per row.  usage: tools/ab_row.sh "<threads> <park budget bytes>" ...
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DFFM_ROW_MAXT=1024 ftrl-ffm_amd/csrc/engine.hip -o /tmp/librow.so 2>/dev/null || { echo "build failed"; exit 1; }
for v in "$@"; do
  set -- $v
  FFM_ENGINE_LIB=/tmp/librow.so FFM_ROW_THREADS=$1 FFM_ROW_PARK_BUDGET=$2 python3 bench.py --no-cpu-baseline --no-eval --steps 60 --warmup 6 $EXP_BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err
  echo "[threads $1 park $2] $(python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('host ms', d['ms_per_step'], 'rows/s', round(d['value']), 'resident ms', d.get('resident',{}).get('ms_per_step'))") | $(grep -E 'row_kernel<train>|update_kernel|refresh' /tmp/ab.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
done
