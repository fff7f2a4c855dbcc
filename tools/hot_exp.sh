#!/bin/bash
# A/B experiment (not part of the product): what bounds one group of touches in the hot kernel?
cd "$GRAFT_REPO_ROOT"
for v in "" "-DHOT_EXP_NOCOMPUTE" "-DHOT_EXP_NOGATHER" "-DHOT_EXP_NOCOMPUTE -DHOT_EXP_NOGATHER"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $v ftrl-ffm_amd/csrc/engine.hip -o /tmp/libexp.so 2>/dev/null
  for cfg in "--dist uniform --n-feats 312" "--dist zipf"; do
    FFM_ENGINE_LIB=/tmp/libexp.so FFM_ENGINE_SERIAL=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $cfg > /tmp/o.json 2> /tmp/o.err
    echo "[$v] [$cfg] $(grep latent_update_hot /tmp/o.err)"
  done
done
