"""Pins the CPU oracle (oracle/ffm_oracle.c) to the compiled reference.

* against the committed golden vectors (tests/golden/*.npz, generated from the reference by
  tests/golden/make_golden.py) -- runs everywhere;
* against the reference itself (oracle/_ref/libftrl_ref.so) on fresh random inputs -- runs where
  that build exists (the build container, and the GPU box via the prebuilt file).
All comparisons are bit-for-bit.
"""
import hashlib

import numpy as np
import pytest

from oracle import pyoracle
from oracle.pyoracle import CpuModel, Csr
from ftrl_ffm_amd import synth
from util import (DEFAULT_HP, GOLDEN, STRESS_HP, assert_bitwise, assert_state_bitwise,
                  bundled_rows, golden_cases, load_case, make_cpu, rand_state)


def test_g1_scalars_match_reference():
    import os
    z = np.load(os.path.join(GOLDEN, "g1_scalars.npz"))
    for tag, hp in (("default", DEFAULT_HP), ("stress", STRESS_HP)):
        m = CpuModel("oracle", "LR", 4, **hp)
        got = np.array([m.maybe_zero_weight(a, b) for a, b in zip(z["n"], z["z"])], np.float32)
        assert_bitwise(got, z["w_" + tag], "W(n,z) " + tag)
    m = CpuModel("oracle", "LR", 4)
    assert_bitwise(np.array([m.sgn(v) for v in z["x"]], np.float32), z["sgn"], "sgn")
    assert_bitwise(np.array([m.sigmoid(v) for v in z["x"]], np.float32), z["sigmoid"], "sigmoid")
    assert_bitwise(np.array([m.loss(1, float(v)) for v in z["x"]]), z["loss_y1"], "loss y=1")
    assert_bitwise(np.array([m.loss(0, float(v)) for v in z["x"]]), z["loss_y0"], "loss y=0")


def test_reference_known_answers():
    """The reference's own unit tests for this path: tests/test_utils.cpp:13-24,40-43."""
    m = CpuModel("oracle", "LR", 4)
    assert m.sgn(1) == 1 and m.sgn(0) == -1 and m.sgn(-2) == -1
    assert m.sigmoid(0) == 0.5
    assert abs(m.sigmoid(1) - 0.7311) < 1e-4 and abs(m.sigmoid(-2) - 0.1192) < 1e-4
    assert abs(m.loss(1, 2) - 0.1269) < 1e-4 and abs(m.loss(0, 1) - 1.3133) < 1e-4


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_replays_golden(name):
    c = load_case(name)
    m = make_cpu("oracle", c)
    m.set_state(c["init"])
    if c["mode"] == "rows":
        for ep in range(int(c["epochs"])):
            lg, ls = m.train_rows(c["csr"])
            assert_bitwise(lg, c["logits"][ep], name + " logits")
            assert_bitwise(np.array([ls]), c["loss_sums"][ep:ep + 1], name + " loss")
        pl, pls = m.predict_batch(c["csr"])
        assert_bitwise(pl, c["post_predict"], name + " post predict")
        assert_bitwise(np.array([pls]), c["post_predict_loss"].reshape(1), name + " loss")
    else:
        pl, pls = m.predict_batch(c["csr"])
        assert_bitwise(pl, c["predict_logit"], name + " predict")
        pp, _ = m.predict_batch(c["csr"], output_prob=True)
        assert_bitwise(pp, c["predict_prob"], name + " prob")
        assert_bitwise(np.array([pls]), c["predict_loss"].reshape(1), name + " loss")
    assert_state_bitwise(m.get_state(), c["final"], name)


def test_g5_quirk_produces_nan():
    """ffm.cpp:118: sqrtf(n2 + g2*g1) with a negative argument poisons the j-side slot."""
    c = load_case("g5_quirk_nan")
    assert np.isnan(c["final"]["vec_z"][13]).any()
    assert np.isnan(c["logits"][0][1])
    f = load_case("g5_quirk_finite")
    assert np.isfinite(f["final"]["vec_z"]).all()


def test_batch_of_one_is_the_reference_step():
    """fo_train_batch with one row per call must equal sequential train()."""
    for name in ("g4_ffm_injected", "g6_fm_injected_k8", "g7_ffm_multivalued",
                 "g8_lr_out_of_range_train"):
        c = load_case(name)
        m = make_cpu("oracle", c)
        m.set_state(c["init"])
        csr = c["csr"]
        for r in range(csr.n_rows):
            lg, _ = m.train_batch(csr.rows(r, r + 1))
            assert_bitwise(lg, c["logits"][0][r:r + 1], name)
        assert_state_bitwise(m.get_state(), c["final"], name)


def test_g9_end_to_end_bundled_data():
    """Bundled data/libffm_data.txt, FFM defaults, 3 epochs in file order."""
    import os
    z = np.load(os.path.join(GOLDEN, "g9_bundled_ffm_end_to_end.npz"))
    nf, F, k = [int(x) for x in z["dims"]]
    rows, labels = bundled_rows()
    csr = Csr.from_rows(rows, labels)
    m = CpuModel("oracle", "FFM", nf, F, k, **DEFAULT_HP)
    st = m.zero_state()
    rng = np.random.default_rng(int(z["init_seed"]))
    st["lin_w"][...] = rng.normal(0, 0.02, st["lin_w"].shape).astype(np.float32)
    st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
    m.set_state(st)
    for ep in range(3):
        _, ls = m.train_rows(csr)
        _, es = m.predict_batch(csr)
        assert ls / csr.n_rows == z["train_loss"][ep]
        assert es / csr.n_rows == z["eval_loss"][ep]
    # the losses the reference CLI prints
    assert ["%.4f" % v for v in z["train_loss"]] == ["0.6907", "0.6883", "0.6867"]
    assert ["%.4f" % v for v in z["eval_loss"]] == ["0.6893", "0.6874", "0.6860"]
    fs = m.get_state()
    for k_ in ("bias3", "lin_w", "lin_n", "lin_z"):
        assert_bitwise(fs[k_], z["final_" + k_], "g9 " + k_)
    assert hashlib.sha256(fs["vec_w"].tobytes()).hexdigest() == str(z["vec_w_sha256"])
    # latent accumulators never move from a fresh model
    assert np.count_nonzero(fs["vec_n"]) == 0 == int(z["vec_n_nonzero"])
    assert np.count_nonzero(fs["vec_z"]) == 0 == int(z["vec_z_nonzero"])


needs_ref = pytest.mark.skipif(not pyoracle.have_ref(), reason="oracle/_ref not built here")


@needs_ref
@pytest.mark.parametrize("model_type", ["LR", "FM", "FFM"])
@pytest.mark.parametrize("hp", [DEFAULT_HP, STRESS_HP], ids=["default_hp", "stress_hp"])
def test_oracle_vs_compiled_reference_random(model_type, hp):
    rng = np.random.default_rng(1234)
    F, k, per = 6, 8, 10
    nf = F * per
    a = CpuModel("oracle", model_type, nf, F, k, **hp)
    b = CpuModel("ref", model_type, nf, F, k, **hp)
    st = rand_state(rng, a)
    a.set_state(st)
    b.set_state(st)
    rows, labels = [], []
    for r in range(300):
        row = [(f, f * per + int(rng.integers(0, per)), float(np.float32(rng.random() + 0.1)))
               for f in range(F) if rng.random() < 0.9]
        if r % 7 == 0:  # a second feature in field 2 (distinct id: duplicates deadlock the reference)
            extra = 2 * per + int(rng.integers(0, per))
            if extra not in [e[1] for e in row]:
                row.append((2, extra, 0.5))
        if r % 11 == 0:
            row += [(9, 5, 1.0), (1, -3, 1.0), (1, 1000, 2.0)]
        rows.append(row)
        labels.append(int(rng.integers(0, 2)))
    csr = Csr.from_rows(rows, labels)
    la, lossa = a.train_rows(csr)
    lb, lossb = b.train_rows(csr)
    assert_bitwise(la, lb, "logits")
    assert lossa == lossb
    assert_state_bitwise(a.get_state(), b.get_state(), model_type)
    pa, _ = a.predict_batch(csr)
    pb, _ = b.predict_batch(csr)
    assert_bitwise(pa, pb, "predict")


@needs_ref
@pytest.mark.parametrize("model_type", ["FM", "FFM"])
def test_sized_reference_model_behaves_as_the_constructed_one(model_type):
    """The CPU baseline's reference model is sized by the harness around the unmodified classes
    (fr_create_sized, above 1e5 weights): same bits as the oracle, i.e. as the reference's own
    constructor + fr_set_state, on a shape just above that threshold; also its threaded loop
    (the baseline's timed region) at one thread."""
    rng = np.random.default_rng(99)
    F, k, per = 13, 16, 60
    nf = F * per if model_type == "FFM" else 7000
    a = CpuModel("oracle", model_type, nf, F, k, **STRESS_HP)
    b = CpuModel("ref", model_type, nf, F, k, **STRESS_HP)
    st = rand_state(rng, a)
    st["vec_n"] += np.float32(0.05)
    a.set_state(st)
    b.set_state(st)
    rows = [[(f if model_type == "FFM" else 0, (f * per if model_type == "FFM" else f * 500) + int(rng.integers(0, per)),
              float(np.float32(rng.random() + 0.1))) for f in range(F)] for _ in range(200)]
    csr = Csr.from_rows(rows, list(rng.integers(0, 2, 200)))
    la, lossa = a.train_rows(csr.rows(0, 100))
    lb, lossb = b.train_rows(csr.rows(0, 100))
    assert_bitwise(la, lb, "logits")
    assert lossa == lossb
    _, la2 = a.train_rows_threaded(csr.rows(100, 200), 1)
    _, lb2 = b.train_rows_threaded(csr.rows(100, 200), 1)
    assert la2 == lb2
    assert_state_bitwise(a.get_state(), b.get_state(), model_type)


@needs_ref
def test_remove_out_range_rule():
    """tests/test_model.cpp:27-29 (LR keeps 1 of 3) and :46-48 (FFM drops all 3)."""
    import os
    z = np.load(os.path.join(GOLDEN, "g8_remove_out_range_counts.npz"))
    assert int(z["lr_keeps"]) == 1 and int(z["ffm_keeps"]) == 0


def test_learning_variant_is_opt_in_and_lets_factors_train():
    """SURVEY.md 8(f) rank 4.  learn=False is the reference bit for bit (all tests above); learn=True
    (1) keeps a latent slot's initial weight until its first gradient and (2) uses g2*g2 at
    ffm.cpp:118 -- so from a fresh model (n = z = 0, random w) the factors move instead of being
    zeroed by the first refresh (the reference's dead latents)."""
    rng = np.random.default_rng(11)
    F, k, per = 4, 4, 10
    nf = F * per
    blk = synth.Generator(F, nf, "zipf", seed=5).block(64)
    outs = {}
    for learn in (False, True):
        m = CpuModel("oracle", "FFM", nf, F, k, learn=learn)
        st = m.zero_state()
        st["vec_w"][...] = np.random.default_rng(1).normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
        m.set_state(st)
        w0 = st["vec_w"].copy()
        m.train_rows(blk)
        outs[learn] = (m.get_state(), w0)
    ref_state, w0 = outs[False]
    lrn_state, _ = outs[True]
    touched = ref_state["vec_w"] != w0          # slots some pair of the block refreshed
    assert touched.any()
    # the reference zeroes every touched slot at its first refresh (W(0,0) = 0); with w = 0 the
    # gradients are 0, so n and z never move: the latents are dead
    assert np.all(ref_state["vec_w"][touched] == 0.0)
    assert np.all(ref_state["vec_n"] == 0.0) and np.all(ref_state["vec_z"] == 0.0)
    # the variant keeps the initial weight for the first forward, accumulates n, z and trains w
    assert np.mean(lrn_state["vec_n"][touched] > 0.0) > 0.9
    assert np.count_nonzero(lrn_state["vec_w"][touched] != w0[touched]) > 0
    # untouched slots keep their initial weights in both
    assert_bitwise(ref_state["vec_w"][~touched], w0[~touched])
    assert_bitwise(lrn_state["vec_w"][~touched], w0[~touched])
    # and the variant never produces the :118 NaN from a fresh model
    assert np.isfinite(lrn_state["vec_z"]).all()
