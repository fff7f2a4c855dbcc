"""bench.py's apportioning of SURVEY.md 8(d)'s per-row bytes to the kernels (CPU only): the update
launches' shares must add up to 8 B per touched slot-factor of every feature with two or more
occurrences whichever way the update is launched (one launch / three side by side), the row kernel
owns the once-only features, and a traffic figure is only attached to the workload it was captured on."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def zipf_block(rows, nnz, n_feats, seed):
    rng = np.random.default_rng(seed)
    return np.minimum(rng.zipf(1.2, rows * nnz) - 1, n_feats - 1).astype(np.int64)


def test_update_shares_add_up_either_way(monkeypatch):
    bench.MODEL = "FFM"
    nnz, k = 39, 16
    blocks = [zipf_block(8192, nnz, 1 << 22, s) for s in (1, 2)]
    per_occ = (nnz - 1) * k
    multi = float(np.mean([c[c > 1].sum() for c in (np.unique(f, return_counts=True)[1] for f in blocks)]))
    want = multi * per_occ * 8
    for split in ("0", "2"):
        monkeypatch.setenv("FFM_UPDATE_SPLIT", split)
        got = sum(bench.kernel_share_bytes(name, blocks, nnz, k, 1)
                  for name in ("ffm_update_all_kernel", "ffm_update_all_kernel<few>", "ffm_update_all_kernel<giant>"))
        if split == "0":  # one launch: the few-occurrence features are its own (no launch of theirs)
            got -= bench.kernel_share_bytes("ffm_update_all_kernel<few>", blocks, nnz, k, 1)
        assert abs(got - want) <= 1e-6 * want, (split, got, want)
    monkeypatch.delenv("FFM_UPDATE_SPLIT")
    # the rule itself: three launches for a whole model's large blocks only
    assert bench.update_split(8192 * 39, 16, 1) and not bench.update_split(4096 * 8, 16, 1)
    assert not bench.update_split(65536 * 39, 16, 8)


def test_row_kernel_owns_the_once_only_features():
    bench.MODEL = "FFM"
    nnz, k = 39, 16
    blocks = [zipf_block(4096, nnz, 1 << 22, 5)]
    once = int((np.unique(blocks[0], return_counts=True)[1] == 1).sum())
    row = bench.kernel_share_bytes("ffm_row_kernel<train>", blocks, nnz, k, 1)
    fixed = 4096 * (nnz * 12 + 12 + (nnz * 12 + 8) + 4 + 16)
    assert abs(row - (fixed + once * (nnz - 1) * k * 20)) < 1.0
    assert bench.kernel_share_bytes("ffm_update_single_kernel", blocks, nnz, k, 1) == 0.0


def test_traffic_only_for_the_captured_workload():
    key = None
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if fn.startswith("r06_pmc_hbm_summary") and fn.endswith(".json"):
            import json
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                key = json.load(f)["_capture"]["workload_key"]
            got, d = bench.matching_pmc_summary(key)
            assert got is not None and d["_capture"]["workload_key"] == key
    assert key is not None
    assert bench.matching_pmc_summary("no such workload") == (None, None)


def test_block_logloss_cost_table_is_found():
    """The sharded bench line quotes what its block size costs in logloss from the committed table
    (profiles/**/r*_logloss_vs_block*.json): 8192- and 65 536-row blocks inside the north star's 1e-4."""
    for rows in (8192, 16384, 65536):
        c = bench.block_logloss_cost(rows)
        assert c is not None and c["block_rows"] <= rows
        assert abs(c["d_train_logloss"]) < 1e-4 and abs(c["d_eval_logloss"]) < 1e-4
