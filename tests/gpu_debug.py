import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
from oracle.pyoracle import CpuModel
from util import DEFAULT_HP, STRESS_HP, rand_state, bits
mt,F,k,per,B = "FFM",39,4,30,7
hp = DEFAULT_HP
rng = np.random.default_rng(7)
nf = F*per
o = CpuModel("oracle", mt, nf, F, k, **hp)
st = rand_state(rng, o)
for key in ("vec_n","lin_n"): st[key] += np.float32(0.05)
o.set_state(st)
e = fa.Engine(mt, nf, F, k, skip_init=True, max_batch_rows=512, **hp); e.set_state(st)
blk = synth.Generator(F, nf, "zipf", seed=3).block(512)
for r0 in range(0, 512, B):
    sub = blk.rows(r0, min(r0+B,512))
    lo,_ = o.train_batch(sub); lg,_ = e.train_batch(sub)
    so, sg = o.get_state(), e.get_state()
    bad = False
    for key in ("vec_n","vec_z","vec_w","lin_n","lin_z"):
        d = np.argwhere(bits(so[key]) != bits(sg[key]))
        if len(d):
            bad = True
            print("block at", r0, key, "mismatches", len(d))
            for idx in d[:6]:
                idx = tuple(idx)
                if len(idx)==2:
                    feat, el = idx
                    c = int((sub.feat==feat).sum())
                    print("   feat",feat,"field",feat//per,"el",el,"fp",el//k,"kk",el%k,"count in block",c,"got",sg[key][idx],"want",so[key][idx], "init", st[key][idx])
                else: print("   ", idx, sg[key][idx], so[key][idx])
    if not np.array_equal(bits(lo), bits(lg)): print("logit mismatch at", r0); bad=True
    if bad:
        print("rows of block:"); 
        for r in range(sub.n_rows):
            b,e_ = sub.row_ptr[r], sub.row_ptr[r+1]
            print("  ", list(sub.feat[b:e_][:6]), "...")
        break
else:
    print("all blocks match")
