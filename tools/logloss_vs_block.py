#!/usr/bin/env python3
"""Logloss of block training vs the strictly sequential reference loop, by block size.

FFM 39x16, reference default hyper-parameters, fresh model, Zipf rows; the host scheduler's policy
(block t = min(B, max(1, rows_seen // 32)) rows) for B = 8192 ... 65536 -- the block sizes of the
1- to 8-GPU weak-scaling runs (the engine's result does not depend on how many shards compute it,
up to the association order of the logit sum).  The oracle's fo_train_rows (checker code) is the
sequential loop.  Prints one JSON object; committed as profiles/rNN_logloss_vs_block.json."""
import json, os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
from oracle.pyoracle import CpuModel

F, K, PER = 39, 16, 2000
N_TRAIN = int(os.environ.get("N_TRAIN", 2_400_000))
N_EVAL, RAMP = 65_536, 32
nf = F * PER
g = synth.Generator(F, nf, "zipf", seed=42)
train, held = g.block(N_TRAIN), g.block(N_EVAL)
rng = np.random.default_rng(1)
o = CpuModel("oracle", "FFM", nf, F, K)
st = o.zero_state()
st["lin_w"][...] = rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
o.set_state(st)
t0 = time.time()
_, seq_train = o.train_rows(train)
_, seq_eval = o.predict_batch(held)
out = {"shape": "FFM F=%d k=%d n_feats=%d, %d train rows + %d held-out, default hyper-parameters, fresh model, ramp %d"
                % (F, K, nf, N_TRAIN, N_EVAL, RAMP),
       "sequential": {"train": seq_train / N_TRAIN, "eval": seq_eval / N_EVAL, "cpu_seconds": round(time.time() - t0, 1)},
       "blocks": {}}
for B in (8192, 16384, 32768, 65536):
    e = fa.Engine("FFM", nf, F, K, skip_init=True, max_batch_rows=B, max_batch_nnz=B * F, max_row_nnz=F)
    e.set_state(st)
    seen, full = 0, 0
    while seen < N_TRAIN:
        rows = min(B, max(1, seen // RAMP), N_TRAIN - seen)
        e.train_batch_async(train.rows(seen, seen + rows))
        full += rows == B
        seen += rows
    tl = e.train_flush()
    el = sum(e.predict_batch(held.rows(r0, min(r0 + B, N_EVAL)))[1] for r0 in range(0, N_EVAL, B))
    e.close()
    out["blocks"][str(B)] = {"full_blocks": int(full), "d_train": tl / N_TRAIN - seq_train / N_TRAIN,
                             "d_eval": el / N_EVAL - seq_eval / N_EVAL}
print(json.dumps(out))
