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
kw = {}
if os.environ.get("TS"):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); kw["stream"] = ts.cuda_stream
if not os.environ.get("NOCAP"):
    kw["max_row_nnz"] = F
e = fa.Engine("FFM", nf, F, K_, max_batch_rows=B, max_batch_nnz=B * F, seed=42, **kw)
e.fill_state()
g = synth.Generator(F, nf, "zipf", seed=42)
NB = int(os.environ.get('NB', 16))
blocks = [g.block(B) for _ in range(NB)]
keep = []
for b in blocks:
    for name in ("row_ptr", "field", "feat", "val", "label"):
        t = torch.from_numpy(getattr(b, name)).pin_memory(); keep.append(t); setattr(b, name, t.numpy())
loss = torch.zeros(4096, dtype=torch.float64, device="cuda")
def run(n, base=0):
    staged = 0
    for i in range(n):
        while staged < min(i + 3, n):
            e.stage_batch(blocks[(base + staged) % NB], True); staged += 1
        e.train_staged(None, loss.data_ptr() + 8 * (i % 4096))
    e.sync()
import gc
gc.collect(); gc.disable()
if os.environ.get("PRE") == "csr":
    for b in blocks: e._csr(b)
if os.environ.get("PRE") == "gpu":   # the device reads every block's arrays once
    for t in keep: t.cuda(non_blocking=True)
    torch.cuda.synchronize()
run(5)
base = 5
for K in (20, 20, 20, 20, 20):
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(K, base); ts.append(time.perf_counter() - t0); base += K
    print("K=%3d  per block us: %s" % (K, " ".join("%7.1f" % (t * 1e6 / K) for t in ts)))
e.close()
