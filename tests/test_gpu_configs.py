"""BASELINE.json's configurations at (or near) full size: the HIP path against the oracle on the
same seeded block, bit for bit -- configs C2 (FFM 8x16, 10k feats, 4096 rows), C3 (FFM 39x4, 1M
feats, 8192 rows), C4 (FM k=64, nnz=39, 8192 rows; 1M feats instead of 10M so that the oracle's
state fits host memory comfortably) and the C5 shape (FFM 39x16) at 200k feats."""
import numpy as np
import pytest
import torch  # noqa: F401  (first: one HIP runtime per process)

import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth
from oracle.pyoracle import CpuModel
from util import DEFAULT_HP, assert_bitwise, assert_state_bitwise

pytestmark = pytest.mark.gpu

CONFIGS = {
    "C2": ("FFM", 8, 16, 10_000, 4096, 2),
    "C3": ("FFM", 39, 4, 1_000_000 - 1_000_000 % 39, 8192, 1),
    "C4": ("FM", 39, 64, 1_000_000 - 1_000_000 % 39, 8192, 1),
    "C5-shape": ("FFM", 39, 16, 39 * 5000, 8192, 1),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_config_block_matches_oracle(name):
    mt, F, k, nf, B, n_blocks = CONFIGS[name]
    e = fa.Engine(mt, nf, F, k, max_batch_rows=B, max_batch_nnz=B * F, seed=11, **DEFAULT_HP)
    e.fill_state(seed=5, n_lo=0.05, n_hi=1.0, z_stddev=0.3)  # warm, reproducible, drawn on device
    st = e.get_state()
    o = CpuModel("oracle", mt, nf, F if mt == "FFM" else 1, k, **DEFAULT_HP)
    o.set_state(st)
    del st
    g = synth.Generator(F, nf, "zipf", seed=42)
    for _ in range(n_blocks):
        blk = g.block(B)
        if mt != "FFM":
            blk.field[:] = 0
        lo, so = o.train_batch(blk)
        lg, sg = e.train_batch(blk)
        assert_bitwise(lg, lo, name + " logits")
        assert abs(sg - so) <= 1e-9 * max(1.0, abs(so))
    assert_state_bitwise(e.get_state(), o.get_state(), name)
    pe, _ = e.predict_batch(blk)
    po, _ = o.predict_batch(blk)
    assert_bitwise(pe, po, name + " predict")
    e.close()
