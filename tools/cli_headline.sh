#!/bin/bash
# The C++ CLI at headline size: ftrl_ffm_main --n_feats 33000000 --n_fields 39
# --n_factors 16 on one MI355X -- rows/s of the online (streamed, parse in the timed region) and
# offline (in-memory, shuffled) trainers, peak host RSS, model file streaming, parse-only ingest
# rate, and the --n_gpus orchestration with the shards sharing this GPU (FTRL_SAME_DEVICE=1).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-cli_headline}
mkdir -p $OUT
python3 - <<'PY'
import sys, time
sys.path.insert(0, '.')
from ftrl_ffm_amd import synth
nf = 33_000_000 - 33_000_000 % 39
g = synth.Generator(39, nf, "zipf", seed=42)
t0 = time.time()
import os
chunks = int(os.environ.get("CLI_CHUNKS", "64"))  # x 32768 rows: 2 M rows (1.1 GB of text) by default
with open('/tmp/synth33m.ffm', 'w') as f:
    for _ in range(chunks):
        f.write(synth.to_libffm_text(g.block(32768)))
print("wrote %d rows of 39 fields in %.1fs" % (chunks * 32768, time.time() - t0))
g2 = synth.Generator(39, 39 * 25000, "zipf", seed=43)
with open('/tmp/synth1m.ffm', 'w') as f:
    for _ in range(8):
        f.write(synth.to_libffm_text(g2.block(32768)))
PY
ls -la /tmp/synth33m.ffm /tmp/synth1m.ffm
NF=32999967
BIN=ftrl-ffm_amd/host/ftrl_ffm_main
run() { python3 tools/rss_run.py "$@" 2>&1; }
echo "== parse-only ingest (host_tests ingest: csr_reader vs Sample reader), 8 threads"
ftrl-ffm_amd/host/host_tests ingest /tmp/synth33m.ffm libffm 8 | tail -1
echo "== parse-only, chunked stream reader (what online mode uses), 1 / 8 / 32 threads"
for t in 1 8 32; do ftrl-ffm_amd/host/host_tests stream /tmp/synth33m.ffm libffm $t | tail -1; done
echo "== online, 33 M features, 8 parse threads, 2 epochs + eval"
run $BIN --train_data /tmp/synth33m.ffm --eval_data /tmp/synth33m.ffm --model_type FFM --n_fields 39 --n_feats $NF --n_factors 16 --online true --n_epochs 2 --batch_size 8192 --n_threads 8 | grep -E "epoch|wall|lines"
echo "== offline, 33 M features, 8 threads, 2 epochs + eval"
run $BIN --train_data /tmp/synth33m.ffm --eval_data /tmp/synth33m.ffm --model_type FFM --n_fields 39 --n_feats $NF --n_factors 16 --online false --n_epochs 2 --batch_size 8192 --n_threads 8 | grep -E "epoch|wall|parsing"
echo "== online with --model_path (zstd frame + .nz sidecar streamed), 3 M features"
run $BIN --train_data /tmp/synth1m.ffm --model_type FFM --n_fields 39 --n_feats 2999997 --n_factors 16 --online true --n_epochs 1 --batch_size 8192 --n_threads 8 --model_path /tmp/m3.zst | grep -E "epoch|wall|saving"
ls -la /tmp/m3.zst /tmp/m3.zst.nz
echo "== --n_gpus 2 / 4, shards sharing this GPU (dry run of the orchestration), 975 k features"
for n in 1 2 4; do
  FTRL_SAME_DEVICE=1 run $BIN --train_data /tmp/synth1m.ffm --eval_data /tmp/synth1m.ffm --model_type FFM --n_fields 39 --n_feats 975000 --n_factors 16 --online true --n_epochs 1 --batch_size 8192 --n_threads 8 --n_gpus $n --field_ranges uniform | grep -E "epoch|wall|shards"
done
echo "== --n_gpus 1 through RCCL (FFM_GROUP_RCCL=1: ncclAllReduce over one rank on the engine stream)"
FFM_GROUP_RCCL=1 run $BIN --train_data /tmp/synth1m.ffm --model_type FFM --n_fields 39 --n_feats 975000 --n_factors 16 --online true --n_epochs 1 --batch_size 8192 --n_threads 8 --n_gpus 1 | grep -E "epoch|wall|shards"
