import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
from oracle.pyoracle import CpuModel
from util import DEFAULT_HP, STRESS_HP, rand_state, bits
mt, F, k, nf, B = "FFM", 8, 16, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
per = nf // F
hp = DEFAULT_HP
e = fa.Engine(mt, nf, F, k, max_batch_rows=B, max_batch_nnz=B * F, seed=11, **hp)
e.fill_state(seed=5, n_lo=0.05, n_hi=1.0, z_stddev=0.3)
st = e.get_state()
o = CpuModel("oracle", mt, nf, F, k, **hp)
o.set_state(st)
g = synth.Generator(F, nf, "zipf", seed=42)
for it in range(2):
    sub = g.block(B)
    lo, _ = o.train_batch(sub); lg, _ = e.train_batch(sub)
    so, sg = o.get_state(), e.get_state()
    bad = False
    dl = np.flatnonzero(bits(lo) != bits(lg))
    print("block", it, "logit mismatches", len(dl), dl[:10])
    for key in ("bias3", "vec_n", "vec_z", "vec_w", "lin_n", "lin_z", "lin_w"):
        d = np.argwhere(bits(so[key]) != bits(sg[key]))
        if len(d):
            bad = True
            print("block", it, key, "mismatches", len(d))
            for idx in d[:8]:
                idx = tuple(idx)
                if len(idx) == 2:
                    feat, el = idx
                    c = int((sub.feat == feat).sum())
                    print("   feat", feat, "field", feat // per, "el", el, "fp", el // k, "kk", el % k,
                          "count in block", c, "got", sg[key][idx], "want", so[key][idx], "init", st[key][idx])
                else:
                    c = int((sub.feat == idx[0]).sum()) if key.startswith("lin") else -1
                    print("   ", idx, "count", c, sg[key][idx], so[key][idx])
    if bad or len(dl):
        break
else:
    print("all blocks match")
