#!/usr/bin/env python3
"""Per-block cost of small blocks through the pipelined host path (the block-size ramp's regime):
blocks of B rows streamed with stage_batch(zero_copy) + train_staged, two staged ahead."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
F, K = 39, 16
nf = 3_000_000 - 3_000_000 % F
e = fa.Engine("FFM", nf, F, K, max_batch_rows=8192, max_batch_nnz=8192 * F, max_row_nnz=F)
e.fill_state()
g = synth.Generator(F, nf, "zipf", seed=42)
loss = torch.zeros(4096, dtype=torch.float64, device="cuda")
for B in (1, 8, 64, 512, 2048, 8192):
    blocks = [g.block(B) for _ in range(16)]
    keep = []
    for b in blocks:
        for name in ("row_ptr", "field", "feat", "val", "label"):
            t = torch.from_numpy(getattr(b, name)).pin_memory(); keep.append(t); setattr(b, name, t.numpy())
    N = 400 if B <= 512 else 100
    def run(n):
        staged = 0
        for i in range(n):
            while staged < min(i + 3, n):
                e.stage_batch(blocks[staged % 16], True); staged += 1
            e.train_staged(None, loss.data_ptr() + 8 * (i % 4096))
        e.sync()
    run(20)
    t0 = time.perf_counter(); run(N); dt = time.perf_counter() - t0
    print("block %5d rows: %7.1f us per block, %9.0f rows/s" % (B, dt / N * 1e6, B * N / dt))
e.close()
