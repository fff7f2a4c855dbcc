"""Several engines in one process (ffm_group_*: what the C++ trainer's --n_gpus uses) and the
pipelined evaluation entry point.

  * a group of ONE engine with FFM_GROUP_RCCL=1: the librccl binding (dlopen), ncclCommInitAll and
    ncclAllReduce on the engine's own stream run on this box -- a sum over one rank, so logits,
    loss and state are the plain engine's bit for bit;
  * a group of TWO / FOUR engines sharing this GPU (compact shard storage, per-field id ranges):
    the whole orchestration -- stage on every engine, forward, sum, update, pipelined and
    synchronous, predict -- against the unsharded engine (the cross-shard logit sum has another
    association order: rtol 2e-4 on state after three blocks, as test_gpu_sharded_procs);
  * ffm_engine_predict_batch_async + flush == the sum of ffm_engine_predict_batch's losses."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ftrl_ffm_amd as fa  # noqa: E402
from ftrl_ffm_amd import synth  # noqa: E402
from util import STRESS_HP, assert_bitwise, assert_state_bitwise  # noqa: E402

pytestmark = pytest.mark.gpu

F, K, PER, B, NBLK = 12, 8, 40, 2048, 4


def _blocks():
    g = synth.Generator(F, F * PER, "zipf", seed=31)
    return [g.block(B) for _ in range(NBLK)]


def _reference_run(blocks):
    nf = F * PER
    ref = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F, **STRESS_HP)
    ref.fill_state(seed=6)
    init = ref.get_state()
    logits, losses = zip(*[ref.train_batch(b) for b in blocks])
    pred, pred_loss = ref.predict_batch(blocks[0])
    want = ref.get_state()
    ref.close()
    return init, logits, losses, pred, pred_loss, want


def test_group_of_one_runs_the_rccl_collective(monkeypatch):
    monkeypatch.setenv("FFM_GROUP_RCCL", "1")
    blocks = _blocks()
    init, logits, losses, pred, pred_loss, want = _reference_run(blocks)
    nf = F * PER
    g = fa.Group([0], "FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F, **STRESS_HP)
    assert g.collective == "rccl" and g.size == 1
    g.engines[0].set_state(init)
    for i, b in enumerate(blocks):
        lg, ls = g.train_batch(b)
        assert_bitwise(lg, logits[i], "logits of block %d through ncclAllReduce" % i)
        assert abs(ls - losses[i]) <= 1e-9 * max(1.0, abs(losses[i]))
    assert_state_bitwise(g.engines[0].get_state(), want, "state after the RCCL-ordered steps")
    out, ls = g.predict_batch(blocks[0])
    assert_bitwise(out, pred, "predict through the group")
    assert abs(ls - pred_loss) <= 1e-9 * max(1.0, abs(pred_loss))
    g.close()


@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("pipelined", [False, True], ids=["sync", "pipelined"])
def test_group_sharing_one_gpu_matches_the_unsharded_engine(n, pipelined):
    blocks = _blocks()
    init, logits, losses, pred, pred_loss, want = _reference_run(blocks)
    nf = F * PER
    fs = (np.arange(F + 1) * PER).astype(np.int32)
    g = fa.Group([0] * n, "FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F,
                 field_start=fs, **STRESS_HP)
    assert g.collective == "device-local sum" and g.size == n
    for e in g.engines:
        e.set_state(init)
    if pipelined:
        for b in blocks:
            g.train_batch_async(b)
        total = g.train_flush()
        assert abs(total - sum(losses)) <= 2e-4 * abs(sum(losses))
    else:
        for i, b in enumerate(blocks):
            lg, ls = g.train_batch(b)
            np.testing.assert_allclose(lg, logits[i], rtol=2e-4, atol=2e-5, err_msg="block %d" % i)
            assert abs(ls - losses[i]) <= 2e-4 * abs(losses[i])
    out, ls = g.predict_batch(blocks[0])
    np.testing.assert_allclose(out, pred, rtol=2e-4, atol=2e-5)
    assert abs(ls - pred_loss) <= 2e-4 * abs(pred_loss)
    # merged state: every shard's owned slots
    plan = fa.shard_plan(F, n, field_map=True)
    states = [e.get_state() for e in g.engines]
    fld = np.arange(nf) // PER
    owner = np.repeat(plan["pair_owner"][fld], K, axis=1)
    for key in ("vec_n", "vec_z"):
        merged = np.zeros_like(want[key])
        for r in range(n):
            merged = np.where(owner == r, states[r][key], merged)
        np.testing.assert_allclose(merged, want[key], rtol=2e-4, atol=2e-5, err_msg=key)
    lin_owner = plan["lin_owner"][fld]
    for key in ("lin_n", "lin_z"):
        merged = np.zeros_like(want[key])
        for r in range(n):
            merged = np.where(lin_owner == r, states[r][key], merged)
        np.testing.assert_allclose(merged, want[key], rtol=2e-4, atol=2e-5, err_msg=key)
    g.close()


def test_group_takes_rows_without_a_field_array():
    """ffm_group_train_batch[_async] stage through ffm_engine_stage_batch, so a group, too, takes FFM rows
    that are one entry per field in field order with field == NULL (every shard's upload kernel writes
    the array; a compact shard then drops the columns it owns nothing of as usual): two groups of the
    same two shards, one fed with the array and one without -- the same bits, shard by shard."""
    import copy
    blocks = _blocks()
    nf = F * PER
    fs = (np.arange(F + 1) * PER).astype(np.int32)

    def run(bare):
        g = fa.Group([0, 0], "FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F,
                     field_start=fs, **STRESS_HP)
        for e in g.engines:
            e.fill_state(seed=6)
        logits = []
        for i, b in enumerate(blocks):
            c = b
            if bare:
                c = copy.copy(b)
                c.__dict__.pop("_ffm_csr_args", None)
                c.field = None
            if i % 2 == 0:
                logits.append(g.train_batch(c)[0].copy())
            else:
                g.train_batch_async(c)
                g.train_flush()
        states = [e.get_state() for e in g.engines]
        g.close()
        return logits, states

    lg_a, st_a = run(False)
    lg_b, st_b = run(True)
    for x, y in zip(lg_a, lg_b):
        assert_bitwise(y, x, "group logits of rows without a field array")
    for r, (a, b) in enumerate(zip(st_a, st_b)):
        assert_state_bitwise(b, a, "shard %d" % r)


def test_pipelined_evaluation_sums_the_block_losses():
    import torch
    blocks = _blocks()
    nf = F * PER
    e = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F, **STRESS_HP)
    e.fill_state(seed=6)
    want = sum(e.predict_batch(b)[1] for b in blocks)
    for b in blocks:  # the copying path
        e.predict_batch_async(b)
    got = e.train_flush()
    assert abs(got - want) <= 1e-12 * abs(want)
    keep = []
    for b in blocks:  # page-locked arrays pulled in place
        for name in ("row_ptr", "field", "feat", "val", "label"):
            t = torch.from_numpy(getattr(b, name)).pin_memory()
            keep.append(t)
            setattr(b, name, t.numpy())
    for b in blocks:
        e.predict_batch_async(b, zero_copy=True)
    got = e.train_flush()
    assert abs(got - want) <= 1e-12 * abs(want)
    assert e.blocks_pulled() == 2 * NBLK
    # a training block afterwards still trains as the plain call would
    ref = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F, **STRESS_HP)
    ref.fill_state(seed=6)
    lg_ref, _ = ref.train_batch(blocks[1])
    lg, _ = e.train_batch(blocks[1])
    assert_bitwise(lg, lg_ref, "training after pipelined evaluation")
    e.close()
    ref.close()
