"""Shared helpers for the parity tests: golden loading, seeded inputs, bitwise comparison."""
import glob
import gzip
import os

import numpy as np

from oracle.pyoracle import CpuModel, Csr

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
STATE_KEYS = ("bias3", "lin_w", "lin_n", "lin_z", "vec_w", "vec_n", "vec_z")
DEFAULT_HP = dict(w_alpha=1e-4, w_beta=1.0, w_l1=0.1, w_l2=5.0)
STRESS_HP = dict(w_alpha=0.1, w_beta=1.0, w_l1=0.01, w_l2=0.1)


def golden_cases():
    """All replayable cases (those written by make_golden.run_case)."""
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "g*.npz"))):
        with np.load(p) as z:
            if "mode" in z.files:
                out.append(os.path.basename(p)[:-4])
    return out


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["csr"] = Csr(d["row_ptr"], d["field"], d["feat"], d["val"], d["label"])
    a, b, l1, l2 = [float(x) for x in d["hp"]]
    d["hp_kw"] = dict(w_alpha=a, w_beta=b, w_l1=l1, w_l2=l2)
    d["init"] = {k: d["init_" + k] for k in STATE_KEYS}
    d["final"] = {k: d["final_" + k] for k in STATE_KEYS if "final_" + k in d}
    d["model_type"] = str(d["model_type"])
    d["mode"] = str(d["mode"])
    return d


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bitwise(a, b, what=""):
    """Bit-for-bit equality of float arrays.  NaNs must sit at the same positions; their sign and
    payload bits are not compared (IEEE-754 leaves them unspecified and x86 picks them by operand
    order, which differs between two compilations of the same expression)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.size == 0:
        return
    both_nan = (np.isnan(a) & np.isnan(b)).ravel()
    bad = np.flatnonzero((bits(a).ravel() != bits(b).ravel()) & ~both_nan)
    assert bad.size == 0, "%s: %d/%d words differ, first at %d: %r vs %r" % (
        what, bad.size, a.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]])


def assert_state_bitwise(sa, sb, what=""):
    for k in STATE_KEYS:
        if k in sa and k in sb:
            assert_bitwise(sa[k], sb[k], what + ":" + k)


def assert_close(a, b, rtol, atol, what=""):
    """NaN positions must agree exactly; finite values within rtol/atol."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), what + ": NaN positions differ"
    ok = np.isclose(a[~na], b[~nb], rtol=rtol, atol=atol)
    if not ok.all():
        i = np.flatnonzero(~ok)[0]
        raise AssertionError("%s: %d/%d outside rtol=%g atol=%g, first %r vs %r" % (
            what, (~ok).sum(), ok.size, rtol, atol, a[~na][i], b[~nb][i]))


def rand_state(rng, model, n_hi=1.0, z_sd=0.3, w_sd=0.02):
    st = model.zero_state()
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


def bundled_rows(libsvm=False):
    """The reference's bundled data/libffm_data.txt (committed gzip'd under tests/golden/data)."""
    with gzip.open(os.path.join(GOLDEN, "data", "libffm_data.txt.gz"), "rt") as f:
        lines = f.read().splitlines()
    rows, labels = [], []
    for line in lines:
        t = line.split()
        labels.append(1 if int(t[0]) > 0 else 0)
        row = []
        for tok in t[1:]:
            fld, ft, v = tok.split(":")
            if np.float32(v) != 0:
                row.append((0 if libsvm else int(fld), int(ft), float(np.float32(v))))
        rows.append(row)
    return rows, labels


def make_cpu(kind, case_or_type, dims=None, hp=None):
    if isinstance(case_or_type, dict):
        c = case_or_type
        nf, F, k = [int(x) for x in c["dims"]]
        return CpuModel(kind, c["model_type"], nf, F, k, **c["hp_kw"])
    nf, F, k = dims
    return CpuModel(kind, case_or_type, nf, F, k, **(hp or DEFAULT_HP))
