#!/bin/bash
# A/B experiments (not part of the product): builds variants with -D macros on the GPU box and
# prints the overlapped step time (and the spans of the update kernels) of each.
# usage: tools/exp.sh "<flags of variant 1>" "<flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v ftrl-ffm_amd/csrc/engine.hip -o /tmp/libexp.so 2>/dev/null || { echo "[$v] build failed"; continue; }
  FFM_ENGINE_LIB=/tmp/libexp.so python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-eval $EXP_BENCH_ARGS > /tmp/o.json 2> /tmp/o.err
  echo "[$v] fused: $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['value'],d['ms_per_step'],d.get('resident',{}).get('ms_per_step'))") $(grep -E 'update|refresh|row_kernel<train>' /tmp/o.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  if [ -n "$EXP_SPLIT" ]; then
  FFM_ENGINE_LIB=/tmp/libexp.so FFM_UPDATE_SPLIT=1 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-eval --no-resident $EXP_BENCH_ARGS > /tmp/s.json 2> /tmp/s.err
  echo "[$v] split: $(grep -E 'update' /tmp/s.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  fi
  if [ -n "$EXP_SERIAL" ]; then
  FFM_ENGINE_LIB=/tmp/libexp.so FFM_ENGINE_SERIAL=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/s.json 2> /tmp/s.err
  echo "[$v] serial: $(grep -E 'row_kernel<train>|latent|fused|refresh' /tmp/s.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  fi
done
