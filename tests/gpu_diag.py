"""Diagnostic (not a test): how close is the HIP engine to the oracle, bit for bit?
Run on the GPU box:  python tests/gpu_diag.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ftrl_ffm_amd as fa  # noqa: E402
from ftrl_ffm_amd import synth  # noqa: E402
from oracle.pyoracle import CpuModel, Csr  # noqa: E402
from util import STATE_KEYS, bits, golden_cases, load_case, make_cpu, rand_state  # noqa: E402


def diff_report(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    both_nan = np.isnan(a) & np.isnan(b)
    neq = (bits(a) != bits(b)) & ~both_nan
    n = int(neq.sum())
    if n == 0:
        return "%s: bitwise equal (%d)" % (what, a.size)
    with np.errstate(all="ignore"):
        ad = np.abs(a.astype(np.float64) - b.astype(np.float64))
        rel = ad / np.maximum(np.abs(b.astype(np.float64)), 1e-30)
    nanmis = int((np.isnan(a) != np.isnan(b)).sum())
    return "%s: %d/%d differ, max abs %.3e, max rel %.3e, nan-mismatch %d" % (
        what, n, a.size, np.nanmax(ad[neq]), np.nanmax(rel[neq]), nanmis)


def engine_for(c):
    nf, F, k = [int(x) for x in c["dims"]]
    return fa.Engine(c["model_type"], nf, F, k, skip_init=True, max_batch_rows=1024, **c["hp_kw"])


def main():
    for name in golden_cases():
        c = load_case(name)
        e = engine_for(c)
        e.set_state(c["init"])
        if c["mode"] == "rows":
            for ep in range(int(c["epochs"])):
                lg, ls = e.train_rows(c["csr"])
                print(name, diff_report(lg, c["logits"][ep], "logits ep%d" % ep),
                      "| loss", ls, "vs", float(c["loss_sums"][ep]))
            st = e.get_state()
            for k in STATE_KEYS:
                print("   ", diff_report(st[k], c["final"][k], k))
        else:
            pl, pls = e.predict_batch(c["csr"])
            print(name, diff_report(pl, c["predict_logit"], "predict"), pls, float(c["predict_loss"]))
        e.close()

    # batch semantics vs the oracle's fo_train_batch, random injected state, several batch sizes
    rng = np.random.default_rng(7)
    for mt, F, k, per in (("FFM", 8, 16, 40), ("FFM", 39, 4, 30), ("FM", 1, 64, 300), ("LR", 1, 1, 300)):
        nf = F * per if mt == "FFM" else per
        hp = dict(w_alpha=0.1, w_beta=1.0, w_l1=0.01, w_l2=0.1)
        o = CpuModel("oracle", mt, nf, F, k, **hp)
        st = rand_state(rng, o)
        for kk in ("vec_n", "lin_n"):
            st[kk] += np.float32(0.05)
        g = synth.Generator(F if mt == "FFM" else 13, nf, "zipf", seed=3)
        blk = g.block(512)
        if mt != "FFM":
            blk.field[:] = 0
        for B in (1, 7, 64, 512):
            o.set_state(st)
            e = fa.Engine(mt, nf, F, k, skip_init=True, max_batch_rows=512, **hp)
            e.set_state(st)
            lo, lg = [], []
            for r0 in range(0, 512 if B > 1 else 64, B):
                sub = blk.rows(r0, min(r0 + B, 512))
                a, _ = o.train_batch(sub)
                b_, _ = e.train_batch(sub)
                lo.append(a)
                lg.append(b_)
            print(mt, "F%d k%d B=%d" % (F, k, B), diff_report(np.concatenate(lg), np.concatenate(lo), "logits"))
            so, sg = o.get_state(), e.get_state()
            for key in STATE_KEYS:
                if so[key].size:
                    print("   ", diff_report(sg[key], so[key], key))
            e.close()

    # sigmoid exactness: engine tmp_grad path vs glibc over many logits (via LR with one feature)
    # timing + kernel table at a C5-like shape (small n_feats)
    F, k, nf, B = 39, 16, 39 * 2000, 8192
    g = synth.Generator(F, nf, "zipf", seed=42)
    e = fa.Engine("FFM", nf, F, k, max_batch_rows=B)
    blocks = [g.block(B) for _ in range(4)]
    e.train_batch(blocks[0])
    e.profile_enable(True)
    t0 = time.time()
    for b_ in blocks[1:]:
        e.train_batch(b_)
    e.sync()
    dt = time.time() - t0
    print("C5-like (n_feats=%d) host-buffer path: %.1f rows/s" % (nf, 3 * B / dt))
    print(e.profile_dump())
    e.close()


if __name__ == "__main__":
    main()
