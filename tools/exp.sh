#!/bin/bash
# A/B experiments (not part of the product): builds variants with -D macros on the GPU box and
# prints the overlapped step time (and the spans of the update kernels) of each.
# usage: tools/exp.sh "<flags of variant 1>" "<flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v ftrl-ffm_amd/csrc/engine.hip -o /tmp/libexp.so 2>/dev/null || { echo "[$v] build failed"; continue; }
  FFM_ENGINE_LIB=/tmp/libexp.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
  echo "[$v] overlapped: $(python -c "import json;d=json.load(open('/tmp/o.json'));print(d['value'],d['ms_per_step'])") $(grep -E 'latent|refresh|row_kernel<train>' /tmp/o.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  if [ -n "$EXP_SERIAL" ]; then
  FFM_ENGINE_LIB=/tmp/libexp.so FFM_ENGINE_SERIAL=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/s.json 2> /tmp/s.err
  echo "[$v] serial: $(grep -E 'row_kernel<train>|latent|fused|refresh' /tmp/s.err | awk '{print $1"="$NF}' | tr '\n' ' ')"
  fi
done
