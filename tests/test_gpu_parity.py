"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the
committed golden vectors.  Everything here needs a real MI355X (-m gpu).

Tolerances (written here, as the contract asks): logits, tmp_grad-driven state (w, n, z) and
weights are compared BIT FOR BIT with the oracle / the reference's golden outputs -- the kernels
perform the reference's fp32 operations in the reference's order (csrc/ftrl_math.h).  The only
floating-point slack is on double logloss sums, where device exp/log may differ from glibc in the
last ulp: |gpu - cpu| <= 1e-12 * max(1, |cpu|) per row summed, stated as LOSS_RTOL below.  The
north-star bound (epoch logloss within 1e-4 of the reference CPU path) is asserted on top.
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  -- first: torch bundles its own HIP runtime; loading it before the
#                              engine library keeps ONE runtime in the process

import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
from oracle.pyoracle import CpuModel, Csr
from util import (DEFAULT_HP, GOLDEN, STRESS_HP, STATE_KEYS, assert_bitwise, assert_state_bitwise,
                  bundled_rows, golden_cases, load_case, make_cpu, rand_state)

pytestmark = pytest.mark.gpu
LOSS_RTOL = 1e-12


def loss_close(a, b):
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= LOSS_RTOL * max(1.0, abs(b)) * 64


def engine_for(case, **kw):
    nf, F, k = [int(x) for x in case["dims"]]
    return fa.Engine(case["model_type"], nf, F, k, skip_init=True, max_batch_rows=1024,
                     **case["hp_kw"], **kw)


@pytest.mark.parametrize("name", golden_cases())
def test_golden_replay_one_row_per_call(name):
    """n_rows == 1 per call is one reference train(): the reference's own outputs, bit for bit."""
    c = load_case(name)
    e = engine_for(c)
    e.set_state(c["init"])
    if c["mode"] == "rows":
        for ep in range(int(c["epochs"])):
            lg, ls = e.train_rows(c["csr"])
            assert_bitwise(lg, c["logits"][ep], name + " logits")
            assert loss_close(ls, float(c["loss_sums"][ep]))
        pl, pls = e.predict_batch(c["csr"])
        assert_bitwise(pl, c["post_predict"], name + " post-train predict")
        assert loss_close(pls, float(c["post_predict_loss"]))
    else:
        pl, pls = e.predict_batch(c["csr"])
        assert_bitwise(pl, c["predict_logit"], name + " predict")
        pp, _ = e.predict_batch(c["csr"], output_prob=True)
        assert_bitwise(pp, c["predict_prob"], name + " prob")
        assert loss_close(pls, float(c["predict_loss"]))
    assert_state_bitwise(e.get_state(), c["final"], name)
    e.close()


CASES = [("FFM", 8, 16, 40), ("FFM", 39, 4, 30), ("FFM", 12, 8, 25), ("FFM", 5, 3, 20),
         ("FM", 1, 64, 300), ("FM", 1, 7, 100), ("LR", 1, 1, 300)]
CASE_IDS = ["%s-%d-%d" % (c[0], c[1], c[2]) for c in CASES]


@pytest.mark.parametrize("mt,F,k,per", CASES, ids=CASE_IDS)
@pytest.mark.parametrize("B", [1, 7, 64, 512])
@pytest.mark.parametrize("hp", [DEFAULT_HP, STRESS_HP], ids=["default_hp", "stress_hp"])
def test_block_semantics_match_oracle(mt, F, k, per, B, hp):
    """Blocks of B rows against oracle fo_train_batch on the same seeded inputs, injected warm
    state, Zipf ids (so features repeat inside a block): bitwise."""
    rng = np.random.default_rng(7)
    nf = F * per if mt == "FFM" else per
    o = CpuModel("oracle", mt, nf, F, k, **hp)
    st = rand_state(rng, o)
    for key in ("vec_n", "lin_n"):
        st[key] += np.float32(0.05)
    o.set_state(st)
    e = fa.Engine(mt, nf, F, k, skip_init=True, max_batch_rows=512, **hp)
    e.set_state(st)
    blk = synth.Generator(F if mt == "FFM" else 13, nf, "zipf", seed=3).block(512)
    if mt != "FFM":
        blk.field[:] = 0
    n = 512 if B > 1 else 48
    for r0 in range(0, n, B):
        sub = blk.rows(r0, min(r0 + B, n))
        lo, so = o.train_batch(sub)
        lg, sg = e.train_batch(sub)
        assert_bitwise(lg, lo, "logits block at %d" % r0)
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "%s B=%d" % (mt, B))
    pe, _ = e.predict_batch(blk)
    po, _ = o.predict_batch(blk)
    assert_bitwise(pe, po, "predict after training")
    e.close()


@pytest.mark.parametrize("F,k", [(5, 32), (4, 64), (3, 128), (6, 12), (5, 20), (4, 96), (7, 2), (3, 200)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("B", [64, 700])
def test_other_factor_counts_match_oracle(F, k, B):
    """The factor counts BASELINE's configurations do not use: k = 32 / 64 (two / one slot per 64-element
    chunk), k = 128 / 96 / 200 (a slot is several chunks, the last one partial), k = 12 / 20 (whole
    16-byte vectors, not a power of two: evaluation keeps the workgroup-per-row kernel), k = 2 (the
    scalar kernels) -- blocks with once-only, few-occurrence, hot and giant features, n near 0
    (ffm.cpp:118's NaNs), bitwise against the oracle; evaluation too."""
    rng = np.random.default_rng(100 + k)
    per = 25
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o, n_hi=0.05)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=1024, **STRESS_HP)
    e.set_state(st)
    g = synth.Generator(F, nf, "zipf", seed=21)
    for n in (B, B, 1):
        blk = g.block(n)
        if n > 300:
            blk.feat[::F] = 0  # one feature in every row: a giant
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "F=%d k=%d logits of a %d-row block" % (F, k, n))
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "F=%d k=%d" % (F, k))
    blk = g.block(256)
    pe, le = e.predict_batch(blk)
    po, lo_ = o.predict_batch(blk)
    assert_bitwise(pe, po, "F=%d k=%d predict" % (F, k))
    assert loss_close(le, lo_)
    e.close()


@pytest.mark.parametrize("k", [65, 128, 33, 4])
def test_fm_other_factor_counts_match_oracle(k):
    """FM beyond one wave per row (k > 64: fm_row_kernel) and at odd k: blocks with once-only,
    repeated and giant features (ranges of 64 occurrences joined by a second launch), bitwise."""
    rng = np.random.default_rng(200 + k)
    nf = 400
    o = CpuModel("oracle", "FM", nf, 1, k, **STRESS_HP)
    st = rand_state(rng, o, n_hi=0.05)
    o.set_state(st)
    e = fa.Engine("FM", nf, 1, k, skip_init=True, max_batch_rows=1024, **STRESS_HP)
    e.set_state(st)
    g = synth.Generator(11, nf, "zipf", seed=23)
    for n in (700, 64, 700, 1):
        blk = g.block(n)
        blk.field = np.zeros(blk.nnz, np.int32)
        if n > 300:
            blk.feat[::11] = 3  # one feature in every row
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "FM k=%d logits of a %d-row block" % (k, n))
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "FM k=%d" % k)
    blk = g.block(256)
    blk.field = np.zeros(blk.nnz, np.int32)
    pe, _ = e.predict_batch(blk)
    po, _ = o.predict_batch(blk)
    assert_bitwise(pe, po, "FM k=%d predict" % k)
    e.close()


def test_ragged_empty_and_multivalued_rows():
    """Empty rows, single-entry rows, out-of-range entries, several features per field, fields
    out of order, a feature repeated across rows of the block."""
    rng = np.random.default_rng(11)
    F, k, per = 6, 8, 9
    nf = F * per
    rows, labels = [], []
    for r in range(200):
        row = []
        for f in rng.permutation(F):
            for i in rng.choice(per, size=int(rng.integers(0, 3)), replace=False):
                row.append((int(f), int(f) * per + int(i), float(np.float32(rng.random() + 0.2))))
        if r % 9 == 0:
            row += [(F + 3, 1, 1.0), (0, -5, 1.0), (1, nf + 7, 0.5), (-1, 3, 2.0)]
        if r % 13 == 0:
            row = []
        if r % 17 == 0:
            row = row[:1]
        rows.append(row)
        labels.append(int(rng.integers(0, 2)))
    csr = Csr.from_rows(rows, labels)
    for B in (1, 16, 200):
        o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
        st = rand_state(rng, o)
        st["vec_n"] += np.float32(0.05)
        o.set_state(st)
        e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=256, **STRESS_HP)
        e.set_state(st)
        for r0 in range(0, 200, B):
            sub = csr.rows(r0, min(r0 + B, 200))
            lo, _ = o.train_batch(sub)
            lg, _ = e.train_batch(sub)
            assert_bitwise(lg, lo, "ragged logits")
        assert_state_bitwise(e.get_state(), o.get_state(), "ragged B=%d" % B)
        e.close()


@pytest.mark.parametrize("k", [4, 8, 16, 32, 64, 12])
def test_evaluation_rows_one_wave_each(k, monkeypatch):
    """FFM predict (ffm.cpp:24-70 with the stored w; evaluate.cpp:23-33): the wave-per-row kernel
    (kernels_predict.h; k = 12 keeps the workgroup-per-row kernel) against the oracle and against
    the workgroup-per-row kernel, bitwise -- rows longer than a wave (up to 150 entries, several
    per field), empty and one-entry rows, entries outside the feature / field range, logits,
    probabilities and the logloss sum."""
    rng = np.random.default_rng(23)
    F, per = 9, 40
    nf = F * per
    rows, labels = [], []
    for r in range(300):
        row = []
        many = r % 7 == 0
        for f in rng.permutation(F):
            cnt = int(rng.integers(8, 18)) if many else int(rng.integers(0, 3))
            for i in rng.choice(per, size=cnt, replace=False):
                row.append((int(f), int(f) * per + int(i), float(np.float32(rng.normal() * 0.7))))
        if r % 9 == 0:
            row += [(F + 3, 1, 1.0), (0, -5, 1.0), (1, nf + 7, 0.5), (-1, 3, 2.0)]
            row = [row[j] for j in rng.permutation(len(row))]
        if r % 13 == 0:
            row = []
        if r % 17 == 0:
            row = row[:1]
        rows.append(row)
        labels.append(int(rng.integers(0, 2)))
    csr = Csr.from_rows(rows, labels)
    assert max(len(r) for r in rows) > 64
    o = CpuModel("oracle", "FFM", nf, F, k, **DEFAULT_HP)
    st = rand_state(rng, o)
    o.set_state(st)
    po, lo = o.predict_batch(csr)
    pp, _ = o.predict_batch(csr, output_prob=True)
    for waves in ("1", "0"):
        monkeypatch.setenv("FFM_PREDICT_WAVE", waves)
        e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=512, max_row_nnz=160, **DEFAULT_HP)
        e.set_state(st)
        pe, le = e.predict_batch(csr)
        assert_bitwise(pe, po, "logits, FFM_PREDICT_WAVE=" + waves)
        assert loss_close(le, lo)
        pq, _ = e.predict_batch(csr, output_prob=True)
        assert_bitwise(pq, pp, "probabilities, FFM_PREDICT_WAVE=" + waves)
        e.close()


@pytest.mark.parametrize("park", ["0", "96", "1024", "default"])
def test_row_kernel_lds_parking_is_bit_identical(park, monkeypatch):
    """The first vectors of (n, z) of a row's once-only features stay in LDS between the row's refresh
    and its in-row update (FFM_ROW_PARK = bytes; default: what fits 31 KB): none, three vectors (the
    boundary falls inside the first record), 32, and the default must all give the oracle's bits --
    warm and near-zero n (the ffm.cpp:118 NaNs), rows with once-only and hot features mixed."""
    if park != "default":
        monkeypatch.setenv("FFM_ROW_PARK", park)
    rng = np.random.default_rng(29)
    F, k, per = 10, 8, 400
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=256, **STRESS_HP)
    e.set_state(st)
    blk = synth.Generator(F, nf, "zipf", seed=9).block(512)
    for r0 in range(0, 512, 256):
        sub = blk.rows(r0, r0 + 256)
        lo, so = o.train_batch(sub)
        lg, sg = e.train_batch(sub)
        assert_bitwise(lg, lo, "park %s logits at %d" % (park, r0))
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "park %s" % park)
    e.close()


@pytest.mark.parametrize("mode", ["0", "2", "3", "1"])
def test_refresh_modes_are_bit_identical(mode, monkeypatch):
    """Where the lazy weight refresh and the once-only features' update run is a scheduling choice
    (FFM_ENGINE_ROW_REFRESH: 0 = one pass over the block's distinct features, 2 = once-only
    features refreshed by their row, 3 = and updated there, 1 = everything per occurrence in the
    row kernel): every choice must give the oracle's bits, on Zipf blocks (features repeat) and on
    ragged rows with several features per field."""
    monkeypatch.setenv("FFM_ENGINE_ROW_REFRESH", mode)
    rng = np.random.default_rng(23)
    F, k, per = 10, 8, 40
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)  # n near 0: the ffm.cpp:118 NaNs included
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=512, **STRESS_HP)
    e.set_state(st)
    blk = synth.Generator(F, nf, "zipf", seed=5).block(512)
    for r0 in range(0, 512, 128):
        sub = blk.rows(r0, r0 + 128)
        lo, _ = o.train_batch(sub)
        lg, _ = e.train_batch(sub)
        assert_bitwise(lg, lo, "mode %s logits at %d" % (mode, r0))
    rows, labels = [], []
    for r in range(96):  # several features per field, fields out of order, once-only ids among them
        row = []
        for f in rng.permutation(F)[: int(rng.integers(2, F + 1))]:
            for i in rng.choice(per, size=int(rng.integers(1, 3)), replace=False):
                row.append((int(f), int(f) * per + int(i), float(np.float32(rng.random() + 0.2))))
        rows.append(row)
        labels.append(int(rng.integers(0, 2)))
    csr = Csr.from_rows(rows, labels)
    lo, _ = o.train_batch(csr)
    lg, _ = e.train_batch(csr)
    assert_bitwise(lg, lo, "mode %s ragged logits" % mode)
    assert_state_bitwise(e.get_state(), o.get_state(), "mode %s" % mode)
    e.close()


@pytest.mark.parametrize("split", ["0", "0n", "2"], ids=["one_launch", "one_launch_four_waves", "three_side_by_side"])
@pytest.mark.parametrize("k", [4, 8, 16])
def test_update_launch_folds_every_class_of_feature(k, split, monkeypatch):
    """The FFM update of a block (kernels_tile.h: ffm_update_all_kernel) as ONE launch whose
    workgroup ranges fold the bias, the linear terms, the hot features' tiles (k = 4 / 8 / 16: the
    three fact-record shapes), the giants and the few-occurrence features and sum the losses -- and
    as the three launches side by side that large blocks get (FFM_UPDATE_SPLIT=2: hot + bias + linear
    | few + serial walk + loss | giant, instantiated per set of ranges): the oracle's bits on blocks
    with once-only, few-occurrence, hot and very hot features (a feature in every row included:
    sixteen segments of 64 occurrences)."""
    monkeypatch.setenv("FFM_UPDATE_SPLIT", split[0])
    if split == "0n":  # (small blocks at k >= 16 run the one launch with eight-wave workgroups: here with four)
        monkeypatch.setenv("FFM_WIDE_NNZ", "0")
    rng = np.random.default_rng(31 + k)
    F, per = 6, 30
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=1024, **STRESS_HP)
    e.set_state(st)
    g = synth.Generator(F, nf, "zipf", seed=9)
    for n in (1024, 700, 33, 1):
        blk = g.block(n)
        blk.feat[::F] = 0  # field 0's entry of every row: one feature with n occurrences
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "k=%d logits of a %d-row block" % (k, n))
        if np.isnan(so):  # (the ffm.cpp:118 NaNs of n near 0 reach the logits: the loss sums are NaN too)
            assert np.isnan(sg)
        else:
            assert abs(sg - so) <= 1e-9 * max(1.0, abs(so))
    assert_state_bitwise(e.get_state(), o.get_state(), "k=%d" % k)
    e.close()


@pytest.mark.parametrize("hp_name", ["default", "stress"])
@pytest.mark.parametrize("occurrences", [1, 2], ids=["once_only_in_row_refresh", "refresh_kernel"])
@pytest.mark.parametrize("mt", ["FFM", "FM"])
def test_g1_weight_formula_pinned_on_the_device(mt, occurrences, hp_name):
    """G1: FtrlModel::maybe_zero_weight (ftrl_model.h:28-33) on the reference's own
    grid of (n, z) -- z = 0, |z| = l1, nextafter(l1), sgn(0) = -1 (tests/test_utils.cpp:13-18) -- for
    both hyper-parameter sets, evaluated BY THE DEVICE: the 2010 pairs are injected as linear and as
    latent accumulators, one block touches each of them (once: the row kernel refreshes them; twice:
    ffm_refresh_kernel / the FM row kernel does), and the w the engine stored -- the refresh writes
    W(n_0, z_0) before the forward, the update never writes w -- must be the golden w bit for bit."""
    z = np.load(os.path.join(GOLDEN, "g1_scalars.npz"))
    gn, gz, want = z["n"], z["z"], z["w_" + hp_name]
    hp = dict(zip(("w_alpha", "w_beta", "w_l1", "w_l2"), [float(v) for v in z["hp_" + hp_name]]))
    N = len(gn)
    if mt == "FFM":
        F, k, P = 2, 4, 256  # field 0: ids [0, P), field 1: ids [P, 2P); touched: the other field's slot
        nf = 2 * P
    else:
        F, k, P = 1, 8, 252  # every factor of ids [0, P)
        nf = N               # (the linear pairs need N features)
    nf = max(nf, N)
    o = CpuModel("oracle", mt, nf, F, k, **hp)
    st = o.zero_state()
    rng = np.random.default_rng(3)
    st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
    st["lin_w"][...] = rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
    st["lin_n"][:N], st["lin_z"][:N] = gn, gz
    L = st["vec_n"].shape[1]
    if mt == "FFM":
        slots = [(i, (1 if i < P else 0) * k + f) for i in range(2 * P) for f in range(k)]
    else:
        slots = [(i, f) for i in range(P) for f in range(k)]
    idx = np.arange(len(slots)) % N  # (the grid, repeated where there are more touched elements than pairs)
    rows_i = np.array([a for a, _ in slots]), np.array([b for _, b in slots])
    st["vec_n"][rows_i] = gn[idx]
    st["vec_z"][rows_i] = gz[idx]
    e = fa.Engine(mt, nf, F, k, skip_init=True, max_batch_rows=4096, max_row_nnz=16, **hp)
    e.set_state(st)
    rows, labels = [], []
    if mt == "FFM":
        for r in range(P):
            rows += [[(0, r, 1.0), (1, P + r, 0.5)]] * occurrences
    else:
        for r in range(P // 2):
            rows += [[(0, 2 * r, 1.0), (0, 2 * r + 1, 0.5)]] * occurrences
    # ... and every linear pair: 8 features per row
    for b in range(0, N, 8):
        ids = list(range(b, min(N, b + 8)))
        if mt == "FFM":  # (one entry per field and row keeps the FFM slots above untouched by these rows)
            rows += [[(0 if i < P else 1, i, 1.0)] for i in ids if i < 2 * P] * occurrences
        else:
            rows += [[(0, i, 1.0) for i in ids]] * occurrences
    labels = [r % 2 for r in range(len(rows))]
    csr = Csr.from_rows(rows, labels)
    e.train_batch(csr)
    got = e.get_state()
    n_lin = min(N, 2 * P) if mt == "FFM" else N
    assert_bitwise(got["lin_w"][:n_lin], want[:n_lin], "%s lin_w = W(n, z)" % mt)
    assert_bitwise(got["vec_w"].reshape(nf, L)[rows_i], want[idx], "%s vec_w = W(n, z)" % mt)
    assert np.count_nonzero(want) > 500 and np.count_nonzero(want == 0) > 10  # both branches of the formula
    e.close()


@pytest.mark.parametrize("split", ["0", "2"], ids=["one_launch", "three_side_by_side"])
@pytest.mark.parametrize("super_min", ["default", "257"], ids=["workgroup_per_giant", "ranges_across_the_chip"])
@pytest.mark.parametrize("k", [4, 16])
def test_giant_features_fold_the_same_either_way(super_min, k, split, monkeypatch):
    """Features with more than 256 occurrences in a block are folded by the waves of one workgroup
    together (a tile per wave and super-step), the longest ones (FFM_SUPER_MIN, 2048 by default) as
    ranges all over the chip through global partial sums and two more launches: the same tree, so
    the oracle's bits either way -- with n near 0 (ffm.cpp:118's NaNs), a feature in every row, rows
    that lack fields (empty tiles), and blocks whose last tile / range is partial."""
    if super_min != "default":
        monkeypatch.setenv("FFM_SUPER_MIN", super_min)
    monkeypatch.setenv("FFM_UPDATE_SPLIT", split)  # (the update as one launch / as three side by side)
    rng = np.random.default_rng(41 + k)
    F, per = 6, 12
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o, n_hi=0.02)
    st["vec_n"][rng.random(st["vec_n"].shape) < 0.2] = 0.0
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=3000, **STRESS_HP)
    e.set_state(st)
    g = synth.Generator(F, nf, "zipf", seed=13)
    for n in (3000, 1531, 700):
        blk = g.block(n)
        blk.feat[::F] = 0  # field 0's entry of every row: one feature with n occurrences
        # a third of the rows lose their field-2 entry: the slots for partner field 2 skip those rows
        keep = np.ones(blk.nnz, bool)
        keep[2::F] = rng.random(n) > 0.33
        per_row = np.add.reduceat(keep.astype(np.int64), blk.row_ptr[:-1].astype(np.int64))
        blk = synth.Block(np.concatenate([[0], np.cumsum(per_row)]).astype(np.int32), blk.field[keep].copy(),
                          blk.feat[keep].copy(), blk.val[keep].copy(), blk.label)
        lo, _ = o.train_batch(blk)
        lg, _ = e.train_batch(blk)
        assert_bitwise(lg, lo, "super_min=%s k=%d logits of a %d-row block" % (super_min, k, n))
    so = o.get_state()
    assert np.isnan(so["vec_z"]).any() and np.isfinite(so["vec_z"]).any()
    assert_state_bitwise(e.get_state(), so, "super_min=%s k=%d" % (super_min, k))
    e.close()


@pytest.mark.parametrize("own", ["0", "1", "range"], ids=["library_sort", "one_launch_sort", "range_sort"])
@pytest.mark.parametrize("rows", [8192, 65536], ids=["8192_rows", "65536_rows"])
def test_grouping_sorts_agree_on_a_four_pass_block(own, rows, monkeypatch):
    """The grouping's two sorts (csrc/kernels_sort.h: one launch, 8 bits per pass; rocPRIM Onesweep)
    on blocks that need four passes (20 M feature ids = 25 key bits): rows x 39 Zipf entries with
    long runs of equal keys (2.5 M entries at 65 536 rows: 157 tiles per workgroup of the one-launch
    sort), and a ragged tail block: both bitwise against the oracle."""
    if own == "range":  # (forced on a model without fields: ONE range, one workgroup, four passes)
        if rows > 8192:
            pytest.skip("one workgroup sorting 2.5 M entries: seconds; the 320 k-entry block covers it")
        monkeypatch.setenv("FFM_RANGE_SORT", "1")
    else:
        monkeypatch.setenv("FFM_OWN_SORT", own)
    rng = np.random.default_rng(11)
    nf = 20_000_003
    o = CpuModel("oracle", "LR", nf, 1, 1, **STRESS_HP)
    e = fa.Engine("LR", nf, 1, 1, skip_init=True, max_batch_rows=rows, **STRESS_HP)
    st = rand_state(rng, o)
    st["lin_n"] += np.float32(0.05)
    o.set_state(st)
    e.set_state(st)
    gen = synth.Generator(39, nf, "zipf", seed=5)
    for n in (rows, rows, 1237):
        blk = gen.block(n)
        blk.field[:] = 0
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "logits, block of %d" % n)
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "LR 20 M features, sort %s" % own)
    e.close()


@pytest.mark.parametrize("mode", ["ranges", "equal_cuts", "library"])
def test_range_sort_groups_whatever_the_rows_look_like(mode, monkeypatch):
    """The grouping's sort by id RANGES (csrc/kernels_sort.h: group_sort_ranges_kernel, one workgroup per
    range of field_start; FFM_RANGE_SORT=1 without field_start: n_fields equal cuts of the id space)
    against the oracle, and the library sort on the same blocks: uneven ranges with an empty one and one
    of a single id (no radix pass at all), rows that lack fields, fields with several entries, ids that
    sit under ANOTHER field than their range's (sorted with the range they lie in), erased entries
    (out-of-range ids and fields), a ragged last block, keys that need one, two and three passes."""
    F, k = 7, 4
    widths = [1, 300, 0, 70000, 5, 2000, 40]  # ids per field: 0 and 1 wide ranges, 1 / 2 / 3 radix passes
    fs = np.concatenate([[0], np.cumsum(widths)]).astype(np.int32)
    nf = int(fs[-1])
    if mode == "library":
        monkeypatch.setenv("FFM_RANGE_SORT", "0")
    elif mode == "equal_cuts":
        monkeypatch.setenv("FFM_RANGE_SORT", "1")
    rng = np.random.default_rng(41)
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=3000, max_batch_nnz=3000 * 12, max_row_nnz=12,
                  field_start=fs if mode == "ranges" else None, **STRESS_HP)
    e.set_state(st)

    def block(n):
        rows, labels = [], []
        for _ in range(n):
            row = []
            for f in range(F):
                if widths[f] == 0 or rng.random() < 0.15:  # the row lacks the field
                    continue
                for _ in range(1 + (rng.random() < 0.1)):  # ... or holds two entries of it
                    i = int(fs[f]) + int(rng.zipf(1.3)) % widths[f]
                    if rng.random() < 0.03:  # an id of another field's range under this field
                        g = int(rng.integers(F))
                        if widths[g]:
                            i = int(fs[g]) + int(rng.integers(widths[g]))
                    if any(i == x[1] for x in row):  # (the same id twice in a row: another test's subject)
                        continue
                    row.append((f, i, float(np.float32(rng.uniform(0.2, 1.5)))))
            if rng.random() < 0.05:
                row.append((int(rng.integers(F)), nf + int(rng.integers(5)), 1.0))  # erased: id out of range
            if rng.random() < 0.05:
                row.append((F + 1, int(rng.integers(nf)), 1.0))  # erased: field out of range
            rows.append(row)
            labels.append(int(rng.integers(2)))
        return Csr.from_rows(rows, labels)

    for n in (3000, 3000, 517, 1):
        blk = block(n)
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "logits, block of %d, %s" % (n, mode))
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "range sort, " + mode)
    e.close()


@pytest.mark.parametrize("k,near_zero", [(16, False), (16, True), (8, False), (4, True), (32, False)])
def test_range_sort_short_cut_for_regular_blocks(k, near_zero):
    """Blocks whose every row is one entry per field in field order, ids inside their fields' ranges,
    take the range sort's short cut (range f = the entries f, f + F, ...: group_keys_kernel finds no
    irregular entry) AND the regular-block fold (kernels_fold.h: the touches' flags are the lane's) in
    the few-occurrence, hot and giant ranges; a block with ONE misplaced id in between takes the general
    paths.  All bitwise against the oracle: k = 4 / 8 / 16 / 32 (the tile pipeline's fact-record shapes),
    Zipf ids over 40 per field (once-only, few, hot and giant features in 1024-row blocks), warm state
    and n near 0 (the ffm.cpp:118 NaNs through the regular fold)."""
    F, per = 9, 40
    nf = F * per
    fs = (np.arange(F + 1) * per).astype(np.int32)
    rng = np.random.default_rng(43)
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o, n_hi=0.02) if near_zero else rand_state(rng, o)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=1024, field_start=fs, **STRESS_HP)
    e.set_state(st)
    gen = synth.Generator(F, nf, "zipf", seed=3)
    for step, n in enumerate((1024, 700, 1024, 1)):
        blk = gen.block(n)
        if step == 1:  # one id of field 0's range under field 3: irregular
            blk.feat[3] = 5
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, "logits, block %d" % step)
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "regular blocks through the range sort")
    e.close()


def test_empty_block_and_capacity_errors():
    e = fa.Engine("FFM", 100, 4, 4, max_batch_rows=8, max_batch_nnz=64, max_row_nnz=16)
    empty = Csr(np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                np.zeros(0, np.float32), np.zeros(0, np.int32))
    before = e.get_state()
    lg, ls = e.train_batch(empty)
    assert lg.size == 0 and ls == 0.0
    assert_state_bitwise(e.get_state(), before, "empty block leaves the model alone")
    big = synth.Generator(4, 100, "uniform", seed=1).block(9)
    with pytest.raises(fa.EngineError) as ei:
        e.train_batch(big)
    assert ei.value.code == -4
    long_row = Csr.from_rows([[(0, i, 1.0) for i in range(17)]], [1])
    with pytest.raises(fa.EngineError) as ei:
        e.train_batch(long_row)
    assert ei.value.code == -4
    bad = Csr(np.array([0, 2, 1], np.int32), np.zeros(2, np.int32), np.zeros(2, np.int32),
              np.ones(2, np.float32), np.zeros(2, np.int32))
    with pytest.raises(fa.EngineError) as ei:
        e.train_batch(bad)
    assert ei.value.code == -1
    e.close()


def test_hot_feature_in_every_row():
    """The bundled data's field 7 holds one id in every row: a group as long as the block, and
    longer than the in-wave sort (c > 64) and the LDS sort when the block is large."""
    rng = np.random.default_rng(5)
    F, k, per = 4, 4, 50
    nf = F * per
    rows = [[(0, int(rng.integers(0, per)), 1.0), (1, per + int(rng.integers(0, 3)), 1.0),
             (2, 2 * per + int(rng.integers(0, per)), 1.0),
             (3, 3 * per, float(np.float32(round(rng.random(), 4) + 0.01)))] for _ in range(3000)]
    csr = Csr.from_rows(rows, list(rng.integers(0, 2, 3000)))
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)
    st["vec_n"] += np.float32(0.05)
    o.set_state(st)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=3000, **STRESS_HP)
    e.set_state(st)
    lo, so = o.train_batch(csr)
    lg, sg = e.train_batch(csr)
    assert_bitwise(lg, lo, "hot feature logits")
    assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "hot feature")
    e.close()


def test_bundled_data_end_to_end_matches_reference_losses():
    """SURVEY.md G9 through the GPU with the host scheduler's block policy (blocks grow with the
    rows already seen: size = clamp(seen // 32, 1, 256) -- DESIGN.md "Block-size ramp"): every
    epoch's train and eval logloss stays within the north-star bound of 1e-4 of the reference's
    sequential CPU path (0.6907/0.6893, 0.6883/0.6874, 0.6867/0.6860), and the result is the
    oracle's for the same block sequence, bit for bit."""
    import os
    from util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "g9_bundled_ffm_end_to_end.npz"))
    nf, F, k = [int(x) for x in z["dims"]]
    rows, labels = bundled_rows()
    csr = Csr.from_rows(rows, labels)
    e = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=256, **DEFAULT_HP)
    o = CpuModel("oracle", "FFM", nf, F, k, **DEFAULT_HP)
    st = e.zero_state()
    rng = np.random.default_rng(int(z["init_seed"]))
    st["lin_w"][...] = rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
    st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
    e.set_state(st)
    o.set_state(st)
    seen = 0
    for ep in range(3):
        tl, r0 = 0.0, 0
        while r0 < csr.n_rows:
            r1 = min(r0 + min(256, max(1, seen // 32)), csr.n_rows)
            lg, ls = e.train_batch(csr.rows(r0, r1))
            if ep == 0:
                lo, _ = o.train_batch(csr.rows(r0, r1))
                assert_bitwise(lg, lo, "bundled logits")
            tl += ls
            seen += r1 - r0
            r0 = r1
        el = 0.0
        for r0 in range(0, csr.n_rows, 256):
            _, ls = e.predict_batch(csr.rows(r0, min(r0 + 256, csr.n_rows)))
            el += ls
        assert abs(tl / csr.n_rows - float(z["train_loss"][ep])) < 1e-4
        assert abs(el / csr.n_rows - float(z["eval_loss"][ep])) < 1e-4
        if ep == 0:
            assert_state_bitwise(e.get_state(), o.get_state(), "bundled epoch 1")
    fs = e.get_state()
    assert np.count_nonzero(fs["vec_n"]) == 0 and np.count_nonzero(fs["vec_z"]) == 0
    e.close()


def test_config1_lr_on_bundled_libsvm_sequential():
    """BASELINE.json configs[0]: LR FTRL on data/libsvm_data.txt, one epoch in file order, one row
    per call -- the reference's CPU path, reproduced bit for bit on the GPU, all 10 000 rows."""
    rows, labels = bundled_rows(libsvm=True)
    csr = Csr.from_rows(rows, labels)  # the whole 10 000-row file
    o = CpuModel("oracle", "LR", 10000, **DEFAULT_HP)
    e = fa.Engine("LR", 10000, skip_init=True, max_batch_rows=16, **DEFAULT_HP)
    lo, so = o.train_rows(csr)
    lg, sg = e.train_rows(csr)
    assert_bitwise(lg, lo, "LR logits")
    assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "LR config 1")
    assert abs(lo[0]) == 0.0  # first row of a fresh model has logit 0
    e.close()


def test_field_pair_sharding_two_shards_on_one_gpu():
    """n_shards = 2: two engines on this GPU each own half of the field pairs; summing their
    partial logits (what the RCCL all-reduce does) and updating must reproduce the unsharded
    engine.  Logits may differ in summation order only: rtol 1e-5 / atol 1e-6; every latent slot
    belongs to exactly one shard, whose (n, z, w) then agree with the unsharded run to the same
    tolerance."""
    import torch
    rng = np.random.default_rng(3)
    F, k, per, B = 6, 8, 20, 256
    nf = F * per
    o = CpuModel("oracle", "FFM", nf, F, k, **STRESS_HP)
    st = rand_state(rng, o)
    st["vec_n"] += np.float32(0.05)
    blk = synth.Generator(F, nf, "zipf", seed=9).block(B)
    ref = fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=B, **STRESS_HP)
    ref.set_state(st)
    lg_ref, _ = ref.train_batch(blk)
    s_ref = ref.get_state()
    shards = [fa.Engine("FFM", nf, F, k, skip_init=True, max_batch_rows=B, n_shards=2, shard_rank=r,
                        **STRESS_HP) for r in range(2)]
    dev = {k_: torch.from_numpy(getattr(blk, k_)).cuda() for k_ in ("row_ptr", "field", "feat", "val", "label")}
    parts = [torch.zeros(B, device="cuda") for _ in range(2)]
    for r, e in enumerate(shards):
        e.set_state(st)
        e.train_forward_device(B, blk.nnz, dev["row_ptr"].data_ptr(), dev["field"].data_ptr(),
                               dev["feat"].data_ptr(), dev["val"].data_ptr(), dev["label"].data_ptr(),
                               parts[r].data_ptr())
        e.sync()
    total = parts[0] + parts[1]
    torch.cuda.synchronize()
    np.testing.assert_allclose(total.cpu().numpy(), lg_ref, rtol=1e-5, atol=1e-6)
    for e in shards:
        e.train_update_device(total.data_ptr())
        e.sync()
    s0, s1 = shards[0].get_state(), shards[1].get_state()
    # ownership: slot (feature of field f, partner field fp) belongs to the owner of {f, fp}
    plan = fa.shard_plan(F, 2)
    fld = np.arange(nf) // per
    owner = np.repeat(plan["pair_owner"][fld], k, axis=1)
    for key in ("vec_n", "vec_z", "vec_w"):
        merged = np.where(owner == 0, s0[key], s1[key])
        np.testing.assert_allclose(merged, s_ref[key], rtol=2e-4, atol=1e-6, err_msg=key)
    lin_state = (s0, s1)[plan["bias_owner"]]  # no field map: bias and all linear terms on one shard
    for key in ("lin_n", "lin_z", "lin_w", "bias3"):
        np.testing.assert_allclose(lin_state[key], s_ref[key], rtol=2e-4, atol=1e-6, err_msg=key)
    # predict on the shards: partial logits summed, then finished by any shard
    pl_ref, ploss_ref = ref.predict_batch(blk)
    pp_ref, _ = ref.predict_batch(blk, output_prob=True)
    pparts = [torch.zeros(B, device="cuda") for _ in range(2)]
    for r, e in enumerate(shards):
        e.predict_batch_device(B, blk.nnz, dev["row_ptr"].data_ptr(), dev["field"].data_ptr(),
                               dev["feat"].data_ptr(), dev["val"].data_ptr(), None, False,
                               pparts[r].data_ptr())
        e.sync()
    ptotal = pparts[0] + pparts[1]
    out = torch.zeros(B, device="cuda")
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    shards[1].predict_finish_device(B, ptotal.data_ptr(), dev["label"].data_ptr(), True,
                                    out.data_ptr(), loss.data_ptr())
    shards[1].sync()
    np.testing.assert_allclose(ptotal.cpu().numpy(), pl_ref, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), pp_ref, rtol=2e-4, atol=1e-6)
    assert abs(float(loss.item()) - ploss_ref) <= 1e-4 * max(1.0, abs(ploss_ref))
    with pytest.raises(fa.EngineError):  # a shard cannot produce probabilities or losses itself
        shards[0].predict_batch_device(B, blk.nnz, dev["row_ptr"].data_ptr(), dev["field"].data_ptr(),
                                       dev["feat"].data_ptr(), dev["val"].data_ptr(),
                                       dev["label"].data_ptr(), True, out.data_ptr())
    for e in shards + [ref]:
        e.close()


def test_full_size_block_properties():
    """BASELINE-size block (F=39, k=16, 8192 rows): size-independent properties.
    (1) determinism: two engines fed the same blocks end bit-identical;
    (2) slots of features absent from the blocks are untouched;
    (3) splitting a block whose rows share no feature into two calls changes nothing."""
    F, k, B = 39, 16, 8192
    nf = F * 4096
    g = synth.Generator(F, nf, "zipf", seed=42)
    blocks = [g.block(B) for _ in range(2)]
    engines = [fa.Engine("FFM", nf, F, k, max_batch_rows=B, seed=1) for _ in range(2)]
    for e in engines:
        e.fill_state(seed=3)
    before = engines[0].get_state()
    outs = []
    for e in engines:
        outs.append([e.train_batch(b) for b in blocks])
    for (la, sa), (lb, sb) in zip(outs[0], outs[1]):
        assert_bitwise(la, lb, "determinism logits")
        assert sa == sb
    sa, sb = engines[0].get_state(), engines[1].get_state()
    assert_state_bitwise(sa, sb, "determinism state")
    seen = np.zeros(nf, bool)
    for b in blocks:
        seen[b.feat] = True
    for key in ("vec_w", "vec_n", "vec_z"):
        assert_bitwise(sa[key][~seen], before[key][~seen], "untouched " + key)
    assert (sa["vec_n"][seen] != before["vec_n"][seen]).any()
    for e in engines:
        e.close()
    # (3) disjoint halves
    rows = [[(f, f * 4096 + (r * 7 + f) % 4096, 1.0) for f in range(F)] for r in range(64)]
    csr = Csr.from_rows(rows, [r % 2 for r in range(64)])
    a = fa.Engine("FFM", nf, F, k, max_batch_rows=64, seed=1)
    b_ = fa.Engine("FFM", nf, F, k, max_batch_rows=64, seed=1)
    for e in (a, b_):
        e.fill_state(seed=3)
    la, _ = a.train_batch(csr)
    l1, _ = b_.train_batch(csr.rows(0, 32))
    l2, _ = b_.train_batch(csr.rows(32, 64))
    # the bias is shared by all rows, so only compare what the halves cannot influence: latents
    sa, sb = a.get_state(), b_.get_state()
    assert_bitwise(la[:32], l1, "first half logits")
    for key in ("vec_w",):
        assert_bitwise(sa[key], sb[key], "disjoint halves " + key)
    a.close()
    b_.close()


def test_device_sigmoid_is_the_c_library_sigmoid():
    """tmp_grad hinges on expf: the device restates glibc's expf (csrc/ftrl_math.h); compare
    sigmoid bit for bit with the oracle's 1/(1+expf(-x)) on 300k logits, including the tails."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 4, 200000), rng.uniform(-110, 110, 100000),
                        [0.0, -0.0, 88.7, -88.7, 103.9, -103.9, 104.5, -104.5, np.inf, -np.inf]
                        ]).astype(np.float32)
    o = CpuModel("oracle", "LR", 4)
    want = np.array([o.sigmoid(float(v)) for v in x], np.float32)
    e = fa.Engine("LR", 4)
    got = e.eval_sigmoid(x)
    assert_bitwise(got, want, "sigmoid")
    e.close()


def test_lookahead_grouping_is_transparent():
    """ffm_engine_prepare_device groups the next block on a side stream; results must be the bits
    of the inline path -- also when the prepared block is NOT the one trained next (discarded)."""
    F, k, per, B = 8, 16, 60, 256
    nf = F * per
    blocks = [synth.Generator(F, nf, "zipf", seed=5).block(B) for _ in range(1)][0:1]
    g = synth.Generator(F, nf, "zipf", seed=5)
    blocks = [g.block(B) for _ in range(5)]
    dev = [{k_: torch.from_numpy(getattr(b, k_)).cuda() for k_ in ("row_ptr", "field", "feat", "val", "label")}
           for b in blocks]
    torch.cuda.synchronize()

    def run(mode):
        e = fa.Engine("FFM", nf, F, k, max_batch_rows=B, seed=3, **STRESS_HP)
        e.fill_state(seed=9)
        out = torch.zeros(len(blocks), B, device="cuda")
        prepared_upto = [0]
        for i, d in enumerate(dev):
            e.train_batch_device(B, blocks[i].nnz, d["row_ptr"].data_ptr(), d["field"].data_ptr(),
                                 d["feat"].data_ptr(), d["val"].data_ptr(), d["label"].data_ptr(),
                                 out[i].data_ptr())
            nxt = []
            if mode == "ahead" and i + 1 < len(dev):
                nxt = [i + 1]
            if mode == "ahead2":  # two blocks ahead, as bench.py does
                nxt = [j for j in (i + 1, i + 2) if j < len(dev) and j > prepared_upto[0]]
            if mode == "wrong" and i + 2 < len(dev):
                nxt = [i + 2]  # prepares a block that is not trained next
            for j in nxt:
                dn = dev[j]
                e.prepare_device(B, blocks[j].nnz, dn["row_ptr"].data_ptr(), dn["field"].data_ptr(),
                                 dn["feat"].data_ptr(), dn["val"].data_ptr())
                prepared_upto[0] = max(prepared_upto[0], j)
        e.sync()
        st = e.get_state()
        lg = out.cpu().numpy()
        e.close()
        return lg, st

    base_l, base_s = run("inline")
    for mode in ("ahead", "ahead2", "wrong"):
        lg, st = run(mode)
        assert_bitwise(lg, base_l, mode + " logits")
        assert_state_bitwise(st, base_s, mode)
    # at most three prepared blocks may wait
    e = fa.Engine("FFM", nf, F, k, max_batch_rows=B, seed=3, **STRESS_HP)
    for j in range(3):
        dn = dev[j]
        e.prepare_device(B, blocks[j].nnz, dn["row_ptr"].data_ptr(), dn["field"].data_ptr(),
                         dn["feat"].data_ptr(), dn["val"].data_ptr())
    with pytest.raises(fa.EngineError) as err:
        dn = dev[3]
        e.prepare_device(B, blocks[3].nnz, dn["row_ptr"].data_ptr(), dn["field"].data_ptr(),
                         dn["feat"].data_ptr(), dn["val"].data_ptr())
    assert err.value.code == -4  # FFM_E_CAPACITY
    e.close()


@pytest.mark.parametrize("mt,F,k,per", [("FFM", 8, 16, 40), ("FFM", 5, 3, 20), ("FM", 1, 7, 100)])
@pytest.mark.parametrize("B", [1, 64])
def test_learning_variant_matches_its_oracle(mt, F, k, per, B):
    """FFM_FLAG_LEARN (not the reference's arithmetic): the device against the
    oracle's restatement of the same two rule changes, bit for bit, on a state where half of the
    slots have not seen a gradient yet (n = 0: the refresh must keep their weights)."""
    rng = np.random.default_rng(21)
    nf = F * per if mt == "FFM" else per
    o = CpuModel("oracle", mt, nf, F, k, learn=True, **STRESS_HP)
    st = rand_state(rng, o)
    fresh = rng.random(st["vec_n"].shape) < 0.5
    st["vec_n"][fresh] = 0.0
    st["vec_z"][fresh] = 0.0
    st["vec_n"][~fresh] += np.float32(0.05)
    o.set_state(st)
    e = fa.Engine(mt, nf, F, k, skip_init=True, max_batch_rows=256, learn=True, **STRESS_HP)
    e.set_state(st)
    blk = synth.Generator(F if mt == "FFM" else 13, nf, "zipf", seed=4).block(256)
    if mt != "FFM":
        blk.field[:] = 0
    n = 256 if B > 1 else 32
    for r0 in range(0, n, B):
        sub = blk.rows(r0, min(r0 + B, n))
        lo, so = o.train_batch(sub)
        lg, sg = e.train_batch(sub)
        assert_bitwise(lg, lo, "logits block at %d" % r0)
        assert loss_close(sg, so)
    assert_state_bitwise(e.get_state(), o.get_state(), "%s learn B=%d" % (mt, B))
    # and it is not the reference rule: the reference-mode oracle ends elsewhere
    ref = CpuModel("oracle", mt, nf, F, k, **STRESS_HP)
    ref.set_state(st)
    for r0 in range(0, n, B):
        ref.train_batch(blk.rows(r0, min(r0 + B, n)))
    assert not np.array_equal(ref.get_state()["vec_w"], o.get_state()["vec_w"])
    e.close()


def test_pipelined_host_training_equals_block_by_block():
    """ffm_engine_train_batch_async / train_flush: same blocks, same order -> the bits of
    ffm_engine_train_batch called block by block, and the same loss sum (summation order of the
    per-block sums aside)."""
    F, k, per = 8, 16, 50
    nf = F * per
    g = synth.Generator(F, nf, "zipf", seed=6)
    blocks = [g.block(n) for n in (1, 3, 64, 256, 17, 256, 256, 5)]
    blocks.insert(4, g.block(64).rows(0, 0))  # an empty block in the middle

    def make():
        e = fa.Engine("FFM", nf, F, k, max_batch_rows=256, seed=3, **STRESS_HP)
        e.fill_state(seed=9)
        return e

    a = make()
    total_a = 0.0
    for b in blocks:
        total_a += a.train_batch(b)[1]
    sa = a.get_state()
    a.close()
    b_ = make()
    for b in blocks[:5]:
        b_.train_batch_async(b)
    part1 = b_.train_flush()
    for b in blocks[5:]:
        b_.train_batch_async(b)
    part2 = b_.train_flush()
    assert b_.train_flush() == 0.0  # nothing pending: empty sum
    sb = b_.get_state()
    b_.close()
    assert_state_bitwise(sb, sa, "pipelined host training")
    assert abs((part1 + part2) - total_a) <= 1e-9 * max(1.0, abs(total_a))
