#!/bin/bash
# The parity suites under the engine's scheduling switches: every variant must give the same bits.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "FFM_ENGINE_SERIAL=1" "FFM_UPDATE_SPLIT=2" "FFM_UPDATE_SPLIT=0" "FFM_OWN_SORT=0" "FFM_OWN_SORT=1" "FFM_RANGE_SORT=1" "FFM_RANGE_SORT=0" \
         "FFM_PREP_AFTER_ROW=1 FFM_PULL_AFTER_ROW=1" "FFM_PREP_AFTER_ROW=0 FFM_PULL_AFTER_ROW=0" "FFM_SUPER_WAIT=0" "FFM_EVAL_DEFER=0" "FFM_ROW_PARK=0" "FFM_UPDATE_ORDER=012"; do
  echo "[$v] $(env $v timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_scale.py -q -m gpu -x -k 'not logloss_within and not headline and not full_headline' 2>&1 | tail -1)"
done
