#!/usr/bin/env python3
"""Host-side cost of the staged training calls (where does the submitting thread spend its time?)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
F, K, B = int(os.environ.get("F", 39)), int(os.environ.get("K", 16)), int(os.environ.get("B", 8192))
nf = int(os.environ.get("NF", 3_000_000)); nf -= nf % F
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
e = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, stream=ts.cuda_stream, max_row_nnz=F)
e.fill_state()
g = synth.Generator(F, nf, "zipf", seed=42)
NB = int(os.environ.get('NB', 16))
blocks = [g.block(B) for _ in range(NB)]
keep = []
for b in blocks:
    for name in ("row_ptr", "field", "feat", "val", "label"):
        t = torch.from_numpy(getattr(b, name)).pin_memory(); keep.append(t); setattr(b, name, t.numpy())
loss = torch.zeros(4096, dtype=torch.float64, device="cuda")
N = int(os.environ.get("STEPS", 300))
ts_stage, ts_train = [], []
torch.cuda.synchronize()
t_all = time.perf_counter()
e.stage_batch(blocks[0], True)
for i in range(N):
    t0 = time.perf_counter()
    if i + 1 < N:
        e.stage_batch(blocks[(i + 1) % NB], True)
    t1 = time.perf_counter()
    e.train_staged(None, loss.data_ptr() + 8 * (i % 4096))
    t2 = time.perf_counter()
    ts_stage.append(t1 - t0); ts_train.append(t2 - t1)
t_sub = time.perf_counter() - t_all
e.sync()
t_tot = time.perf_counter() - t_all
a, b = np.array(ts_stage) * 1e6, np.array(ts_train) * 1e6
print("steps %d: submit %.1f ms, total %.1f ms (%.3f ms/step)" % (N, t_sub * 1e3, t_tot * 1e3, t_tot * 1e3 / N))
for name, v in (("stage_batch", a), ("train_staged", b)):
    print("%-13s us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; >1ms: %d" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), (v > 1000).sum()))
print("first 24 stage:", np.round(a[:24]).astype(int).tolist())
print("first 24 train:", np.round(b[:24]).astype(int).tolist())
big = np.flatnonzero((a + b) > 1000)
print("steps with >1 ms of host time:", big[:40].tolist())
