#!/usr/bin/env python3
"""Fixed cost of one timed region of the pipelined host path: K blocks between two syncs, K = 1 .. 64."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
F, K_, B = 39, 16, 8192
nf = int(os.environ.get("NF", 33_000_000)); nf -= nf % F
e = fa.Engine("FFM", nf, F, K_, max_batch_rows=B, max_batch_nnz=B * F, max_row_nnz=F)
e.fill_state()
g = synth.Generator(F, nf, "zipf", seed=42)
blocks = [g.block(B) for _ in range(16)]
keep = []
for b in blocks:
    for name in ("row_ptr", "field", "feat", "val", "label"):
        t = torch.from_numpy(getattr(b, name)).pin_memory(); keep.append(t); setattr(b, name, t.numpy())
loss = torch.zeros(4096, dtype=torch.float64, device="cuda")
def run(n):
    staged = 0
    for i in range(n):
        while staged < min(i + 3, n):
            e.stage_batch(blocks[staged % 16], True); staged += 1
        e.train_staged(None, loss.data_ptr() + 8 * (i % 4096))
    e.sync()
run(30)
for K in (1, 2, 4, 8, 16, 32, 64):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(K); ts.append(time.perf_counter() - t0)
    m = min(ts)
    print("K=%3d  total %8.1f us  per block %7.1f us" % (K, m * 1e6, m * 1e6 / K))
e.close()
