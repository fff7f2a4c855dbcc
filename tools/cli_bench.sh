#!/bin/bash
# End-to-end rate of the C++ CLI (host/ftrl_ffm_main) on a synthetic libffm file: parse + train.
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from ftrl_ffm_amd import synth
g = synth.Generator(39, 39 * 25000, "zipf", seed=42)
t0 = time.time()
with open('/tmp/synth.ffm', 'w') as f:
    for _ in range(8):
        f.write(synth.to_libffm_text(g.block(32768)))
print("wrote 262144 rows in %.1fs" % (time.time() - t0))
PY
ls -la /tmp/synth.ffm
for bs in 8192; do
  time ftrl-ffm_amd/host/ftrl_ffm_main --train_data /tmp/synth.ffm --model_type FFM --n_fields 39 --n_feats 975000 --n_factors 16 --eval_data /tmp/synth.ffm --online false --n_epochs 3 --batch_size $bs --n_threads 8 2>&1 | head -40
done
