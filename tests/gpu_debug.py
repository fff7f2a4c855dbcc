"""Scratch diagnostics for a failing GPU parity case (not collected by pytest)."""
import sys, os
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ftrl_ffm_amd as fa
from oracle.pyoracle import CpuModel, Csr
from util import STRESS_HP, rand_state

def run(dup_mode):
    rng = np.random.default_rng(31)
    F, k, per = 6, 8, 5
    nf = F * per
    rows, labels = [], []
    for r in range(100):
        row = [(f, f * per + int(rng.integers(0, per)), float(np.float32(rng.random() + 0.3)))
               for f in range(F) if rng.random() < 0.8]
        if dup_mode in (1, 3) and row and r % 3 == 0:
            j = int(rng.integers(0, len(row)))
            row.append(row[j])
        if dup_mode in (2, 3) and len(row) > 2 and r % 5 == 0:
            row.append(((row[0][0] + 1) % F, row[1][1], 0.7))
        rows.append(row); labels.append(int(rng.integers(0, 2)))
    csr = Csr.from_rows(rows, labels)
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o); st["vec_n"] += np.float32(0.05); o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=128, max_row_nnz=2 * F + 2, **STRESS_HP)
    e.set_state(st)
    lo, _ = o.train_batch(csr); lg, _ = e.train_batch(csr)
    bad = np.flatnonzero(lo.view(np.uint32) != lg.view(np.uint32))
    print("mode", dup_mode, "bad logits", bad.size, bad[:10])
    so, se = o.get_state(), e.get_state()
    for key in ("vec_w", "lin_w", "vec_n", "vec_z"):
        d = np.argwhere(so[key].view(np.uint32) != se[key].view(np.uint32))
        print(" ", key, "differs at", len(d), d[:8].tolist())
    for r in bad[:3]:
        print("  row", r, rows[r])
    e.close()

for mode in (0, 1, 2, 3):
    run(mode)
