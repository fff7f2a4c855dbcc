#!/usr/bin/env python3
"""Golden values for the block-size bound at the multi-GPU job's block sizes
(tests/test_gpu_scale.py::test_logloss_within_1e4_at_blocks_of_65536).

The reference is strictly per-sample (src/task/ftrl_offline.cpp:74-83); the engine trains blocks.
This script runs the SEQUENTIAL loop -- the oracle's fo_train_rows, which tests/test_oracle_golden.py
pins to the compiled reference -- over the seeded synthetic rows the GPU test regenerates, and
writes the mean train (progressive) and eval logloss.  ~4 minutes of one CPU core; the GPU test
then needs no CPU loop of its own.

    python tests/golden/make_sequential_logloss.py          # writes g10_sequential_logloss_39x16.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ftrl_ffm_amd import synth  # noqa: E402
from oracle.pyoracle import CpuModel  # noqa: E402

F, K, PER = 39, 16, 2000
N_TRAIN, N_EVAL = 2_400_000, 65_536
HP = dict(w_alpha=1e-4, w_beta=1.0, w_l1=0.1, w_l2=5.0)  # the reference's defaults


def inputs():
    """The rows and the initial weights, shared with the GPU test."""
    g = synth.Generator(F, F * PER, "zipf", seed=42)
    train, held = g.block(N_TRAIN), g.block(N_EVAL)
    rng = np.random.default_rng(1)
    lin_w = rng.normal(0, 0.02, F * PER).astype(np.float32)
    vec_w = rng.normal(0, 0.02, (F * PER, F * K)).astype(np.float32)
    return train, held, lin_w, vec_w


if __name__ == "__main__":
    train, held, lin_w, vec_w = inputs()
    o = CpuModel("oracle", "FFM", F * PER, F, K, **HP)
    st = o.zero_state()
    st["lin_w"][...] = lin_w
    st["vec_w"][...] = vec_w
    o.set_state(st)
    t0 = time.time()
    _, seq_train = o.train_rows(train)
    _, seq_eval = o.predict_batch(held)
    out = {"shape": "FFM F=%d k=%d n_feats=%d, %d train rows + %d held-out, Zipf(1.1) seed 42, "
                    "reference default hyper-parameters, weights N(0, 0.02) seed 1, n = z = 0"
                    % (F, K, F * PER, N_TRAIN, N_EVAL),
           "sequential_train_logloss": seq_train / N_TRAIN, "sequential_eval_logloss": seq_eval / N_EVAL,
           "cpu_seconds": round(time.time() - t0, 1)}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g10_sequential_logloss_39x16.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
