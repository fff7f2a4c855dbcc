#!/usr/bin/env python3
"""Generate the golden vectors G1-G9 of SURVEY.md section 8(c) from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference to build oracle/_ref/libftrl_ref.so via
oracle/Makefile).  Outputs are data only -- inputs plus the reference's outputs -- written next to
this script as .npz files, which are committed; the reference itself never travels.

    python tests/golden/make_golden.py
"""
import gzip
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.pyoracle import CpuModel, Csr, REFERENCE_ROOT, build  # noqa: E402

DEFAULT_HP = dict(w_alpha=1e-4, w_beta=1.0, w_l1=0.1, w_l2=5.0)   # cmd_option.h:49-63
STRESS_HP = dict(w_alpha=0.1, w_beta=1.0, w_l1=0.01, w_l2=0.1)


def parse_libffm(lines, libsvm=False):
    """Restates the accepted grammar of src/data/parser.cpp:11-103 for well-formed lines."""
    rows, labels = [], []
    for line in lines:
        t = line.split()
        if not t:
            continue
        labels.append(1 if int(t[0]) > 0 else 0)
        row = []
        for tok in t[1:]:
            p = tok.split(":")
            if libsvm:
                fld, ft, v = 0, int(p[0]), float(p[1])
            else:
                fld, ft, v = int(p[0]), int(p[1]), float(p[2])
            if np.float32(v) != 0:
                row.append((fld, ft, float(np.float32(v))))
        rows.append(row)
    return rows, labels


def rand_state(rng, m, n_hi=1.0, z_sd=0.3, w_sd=0.02):
    st = m.zero_state()
    for k in st:
        if k == "bias3":
            continue
        if k.endswith("_n"):
            st[k][...] = (rng.random(st[k].shape) * n_hi).astype(np.float32)
        elif k.endswith("_z"):
            st[k][...] = rng.normal(0, z_sd, st[k].shape).astype(np.float32)
        else:
            st[k][...] = rng.normal(0, w_sd, st[k].shape).astype(np.float32)
    st["bias3"][...] = np.array([0.013, 0.7, -0.4], np.float32)
    return st


def run_case(name, model_type, dims, hp, state, csr, mode="rows", epochs=1, store_state=True):
    """mode 'rows': sequential train() per row (x epochs); 'predict': predict_batch only."""
    n_feats, n_fields, n_factors = dims
    m = CpuModel("ref", model_type, n_feats, n_fields, n_factors, **hp)
    m.set_state(state)
    out = dict(model_type=np.array(model_type), dims=np.array(dims, np.int32),
               hp=np.array([hp["w_alpha"], hp["w_beta"], hp["w_l1"], hp["w_l2"]], np.float32),
               mode=np.array(mode), epochs=np.array(epochs),
               row_ptr=csr.row_ptr, field=csr.field, feat=csr.feat, val=csr.val, label=csr.label)
    for k, v in state.items():
        out["init_" + k] = v
    if mode == "rows":
        logits, losses = [], []
        for _ in range(epochs):
            lg, ls = m.train_rows(csr)
            logits.append(lg)
            losses.append(ls)
        out["logits"] = np.stack(logits)
        out["loss_sums"] = np.array(losses, np.float64)
        pl, pls = m.predict_batch(csr)
        out["post_predict"] = pl
        out["post_predict_loss"] = np.array(pls, np.float64)
    else:
        pl, pls = m.predict_batch(csr)
        pp, _ = m.predict_batch(csr, output_prob=True)
        out["predict_logit"] = pl
        out["predict_prob"] = pp
        out["predict_loss"] = np.array(pls, np.float64)
    if store_state:
        for k, v in m.get_state().items():
            out["final_" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)
    return m


def main():
    build(ref=True)
    rng = np.random.default_rng(42)

    # ---- G1: scalar helpers -------------------------------------------------------------
    m = CpuModel("ref", "LR", 4, **DEFAULT_HP)
    ms = CpuModel("ref", "LR", 4, **STRESS_HP)
    n = np.concatenate([[0, 0, 0, 0, 1e-12, 1, 4, 1e6, 0.25, 0.25],
                        rng.random(2000) * 10]).astype(np.float32)
    z = np.concatenate([[0, 0.1, -0.1, np.nextafter(np.float32(0.1), np.float32(1)), 0.5, -0.5,
                         3, -3, 0.1000001, -0.0999999], rng.normal(0, 1, 2000)]).astype(np.float32)
    xs = np.concatenate([[0, 1, -2, 20, -20, 88.5, -88.5, 100, -104], rng.normal(0, 5, 500)]).astype(np.float32)
    np.savez_compressed(
        os.path.join(HERE, "g1_scalars.npz"), n=n, z=z,
        w_default=np.array([m.maybe_zero_weight(a, b) for a, b in zip(n, z)], np.float32),
        w_stress=np.array([ms.maybe_zero_weight(a, b) for a, b in zip(n, z)], np.float32),
        hp_default=np.array(list(DEFAULT_HP.values()), np.float32),
        hp_stress=np.array(list(STRESS_HP.values()), np.float32),
        x=xs, sgn=np.array([m.sgn(v) for v in xs], np.float32),
        sgn_int_1_0_m2=np.array([m.lib.fr_sgn_int(1), m.lib.fr_sgn_int(0), m.lib.fr_sgn_int(-2)]),
        sigmoid=np.array([m.sigmoid(v) for v in xs], np.float32),
        loss_y1=np.array([m.loss(1, float(v)) for v in xs], np.float64),
        loss_y0=np.array([m.loss(0, float(v)) for v in xs], np.float64),
        loss_1_2=np.array(m.loss(1, 2.0)), loss_0_1=np.array(m.loss(0, 1.0)))
    print("wrote g1_scalars")

    # ---- bundled data (also committed, gzip'd, as the config-1 / G9 input) --------------
    with open(os.path.join(REFERENCE_ROOT, "data", "libffm_data.txt")) as f:
        ffm_text = f.read()
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    with gzip.GzipFile(os.path.join(HERE, "data", "libffm_data.txt.gz"), "wb", mtime=0) as g:
        g.write(ffm_text.encode())
    ffm_rows, ffm_labels = parse_libffm(ffm_text.splitlines())
    svm_rows = [[(0, ft, v) for (_, ft, v) in r] for r in ffm_rows]

    # ---- G2: LR, fresh state (zero n,z; seeded w), first 256 rows of libsvm data --------
    lr = CpuModel("oracle", "LR", 10000)
    st = lr.zero_state()
    st["lin_w"][...] = rng.normal(0, 0.02, 10000).astype(np.float32)
    run_case("g2_lr_fresh", "LR", (10000, 1, 1), DEFAULT_HP, st,
             Csr.from_rows(svm_rows[:256], ffm_labels[:256]))

    # ---- G3: FFM fresh on the 10 rows of tests/common.h, 2 epochs -----------------------
    common_rows = """0 0:1:1 1:13:1 2:21:1 3:31:1
1 0:4:1 1:11:1 2:23:1 3:32:1
1 0:2:1 1:13:1 2:25:1 3:34:1
0 0:1:1 1:14:1 2:21:1 3:32:1
0 0:2:1 1:15:1 2:22:1 3:34:1
1 0:4:1 1:11:1 2:21:1 3:35:1
1 0:5:1 1:12:1 2:23:1 3:31:1
1 0:5:1 1:12:1 2:25:1 3:38:1
0 0:2:1 1:11:1 2:24:1 3:37:1
1 0:1:1 1:15:1 2:22:1 3:35:1""".splitlines()
    rows, labels = parse_libffm(common_rows)
    f3 = CpuModel("oracle", "FFM", 50, 4, 4)
    st = f3.zero_state()
    st["lin_w"][...] = rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
    st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
    run_case("g3_ffm_fresh_common", "FFM", (50, 4, 4), DEFAULT_HP, st, Csr.from_rows(rows, labels),
             epochs=2)

    # ---- G4: FFM injected state, 64 rows F=8 k=16, distinct ids, non-unit values --------
    F, k, per = 8, 16, 12
    nf = F * per
    f4 = CpuModel("oracle", "FFM", nf, F, k)
    st = rand_state(rng, f4, n_hi=2.0)
    st_nan = {k_: v_.copy() for k_, v_ in st.items()}   # n near 0 somewhere: :118 NaN spreads
    for k_ in ("vec_n", "lin_n"):
        st[k_] += np.float32(0.05)                      # n + g1*g2 stays >= 0: finite case
    rows, labels = [], []
    for _ in range(64):
        rows.append([(f, f * per + int(rng.integers(0, per)), float(np.float32(rng.random() * 2 + 0.05)))
                     for f in range(F)])
        labels.append(int(rng.integers(0, 2)))
    run_case("g4_ffm_injected", "FFM", (nf, F, k), STRESS_HP, st, Csr.from_rows(rows, labels))
    run_case("g4b_ffm_injected_default_hp", "FFM", (nf, F, k), DEFAULT_HP, st,
             Csr.from_rows(rows, labels))
    run_case("g4c_ffm_injected_nan_spread", "FFM", (nf, F, k), STRESS_HP, st_nan,
             Csr.from_rows(rows, labels))

    # ---- G5: the ffm.cpp:118 quirk: sqrtf(negative) -> NaN, and the finite n=1 variant ---
    for tag, n0 in (("nan", 0.0), ("finite", 1.0)):
        f5 = CpuModel("oracle", "FFM", 50, 4, 4)
        st = f5.zero_state()
        st["vec_z"][1, :] = 0.5
        st["vec_z"][13, :] = -0.5
        st["vec_n"][...] = n0
        st["lin_n"][...] = n0
        run_case("g5_quirk_" + tag, "FFM", (50, 4, 4), DEFAULT_HP, st,
                 Csr.from_rows([[(0, 1, 1.0), (1, 13, 1.0)]] * 2, [0, 0]))

    # ---- G6: FM injected state, k=8 and k=64, 32 rows ------------------------------------
    for k6 in (8, 64):
        nf6 = 200
        f6 = CpuModel("oracle", "FM", nf6, 1, k6)
        st = rand_state(rng, f6)
        rows, labels = [], []
        for _ in range(32):
            ids = rng.choice(nf6, size=int(rng.integers(3, 12)), replace=False)
            rows.append([(0, int(i), float(np.float32(rng.random() + 0.1))) for i in ids])
            labels.append(int(rng.integers(0, 2)))
        run_case("g6_fm_injected_k%d" % k6, "FM", (nf6, 1, k6), STRESS_HP, st,
                 Csr.from_rows(rows, labels))

    # ---- G7: multi-valued fields (a slot touched twice within one row) --------------------
    F, k, per = 5, 4, 8
    nf = F * per
    f7 = CpuModel("oracle", "FFM", nf, F, k)
    st = rand_state(rng, f7, n_hi=3.0)
    rows, labels = [], []
    for _ in range(48):
        row = []
        for f in range(F):
            cnt = int(rng.integers(0, 4))  # 0..3 features in this field
            for i in rng.choice(per, size=cnt, replace=False):
                row.append((f, f * per + int(i), float(np.float32(rng.random() + 0.2))))
        order = rng.permutation(len(row))
        rows.append([row[i] for i in order])  # fields interleaved, not sorted
        labels.append(int(rng.integers(0, 2)))
    rows.append([])                          # an empty row
    labels.append(1)
    rows.append([(2, 17, 1.5)])              # a single-entry row (no pairs)
    labels.append(0)
    run_case("g7_ffm_multivalued", "FFM", (nf, F, k), STRESS_HP, st, Csr.from_rows(rows, labels))

    # ---- G8: out-of-range ids: predict + train with filtered rows -------------------------
    f8 = CpuModel("oracle", "FFM", 50, 4, 4)
    st = rand_state(rng, f8)
    rows = [[(1, -1, 3.0), (44, 0, 1.0), (1, 100, 0.5)],
            [(1, 3, 3.0), (1, 0, 1.0), (3, 10, 1.0), (12, 4, 0.7), (111, 1, 0.2), (8, 8, 8.0)],
            [(0, 1, 1.0), (-1, 2, 1.0), (3, 49, 2.0), (2, 50, 1.0), (4, 7, 1.0)],
            [(0, 5, 1.0), (1, 15, 1.0), (2, 25, 1.0), (3, 35, 1.0)]]
    labels = [1, 0, 1, 0]
    c8 = Csr.from_rows(rows, labels)
    run_case("g8_ffm_out_of_range_predict", "FFM", (50, 4, 4), STRESS_HP, st, c8, mode="predict")
    run_case("g8_ffm_out_of_range_train", "FFM", (50, 4, 4), STRESS_HP, st, c8)
    l8 = CpuModel("oracle", "LR", 50)
    st8 = rand_state(rng, l8)
    c8l = Csr.from_rows([[(1, -1, 3.0), (1, 0, 1.0), (1, 100, 0.5)], rows[1]], [1, 0])
    run_case("g8_lr_out_of_range_train", "LR", (50, 1, 1), STRESS_HP, st8, c8l)
    rl = CpuModel("ref", "LR", 50)
    rf = CpuModel("ref", "FFM", 50, 4, 4)
    cc = Csr.from_rows([[(1, -1, 3.0), (1, 0, 1.0), (1, 100, 0.0)]], [0])
    cf = Csr.from_rows([[(1, -1, 3.0), (44, 0, 1.0), (1, 100, 0.0)]], [0])
    from oracle.pyoracle import _i32, _f32
    np.savez_compressed(os.path.join(HERE, "g8_remove_out_range_counts.npz"),
                        lr_keeps=np.array(rl.lib.fr_remove_out_range(rl.h, 3, _i32(cc.field), _i32(cc.feat), _f32(cc.val))),
                        ffm_keeps=np.array(rf.lib.fr_remove_out_range(rf.h, 3, _i32(cf.field), _i32(cf.feat), _f32(cf.val))))

    # ---- G9: end to end, bundled libffm data, FFM defaults, 3 epochs, file order ----------
    nf, F, k = 10000, 8, 16
    f9 = CpuModel("oracle", "FFM", nf, F, k)
    st = f9.zero_state()
    g9rng = np.random.default_rng(9)
    st["lin_w"][...] = g9rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
    st["vec_w"][...] = g9rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
    c9 = Csr.from_rows(ffm_rows, ffm_labels)
    m9 = CpuModel("ref", "FFM", nf, F, k, **DEFAULT_HP)
    m9.set_state(st)
    train_loss, eval_loss = [], []
    for _ in range(3):
        _, ls = m9.train_rows(c9)
        _, es = m9.predict_batch(c9)
        train_loss.append(ls / c9.n_rows)
        eval_loss.append(es / c9.n_rows)
    fs = m9.get_state()
    np.savez_compressed(
        os.path.join(HERE, "g9_bundled_ffm_end_to_end.npz"), dims=np.array([nf, F, k], np.int32),
        init_seed=np.array(9), train_loss=np.array(train_loss), eval_loss=np.array(eval_loss),
        final_bias3=fs["bias3"], final_lin_w=fs["lin_w"], final_lin_n=fs["lin_n"],
        final_lin_z=fs["lin_z"],
        vec_w_sha256=np.array(hashlib.sha256(fs["vec_w"].tobytes()).hexdigest()),
        vec_n_nonzero=np.array(np.count_nonzero(fs["vec_n"])),
        vec_z_nonzero=np.array(np.count_nonzero(fs["vec_z"])),
        vec_w_zero_count=np.array(int((fs["vec_w"] == 0).sum())))
    print("wrote g9: train", train_loss, "eval", eval_loss)


if __name__ == "__main__":
    main()
