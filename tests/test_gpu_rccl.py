"""The RCCL leg on hardware: a real ProcessGroupNCCL (backend "nccl" IS RCCL on
ROCm) of ONE rank, the engine on a non-default stream, blocks staged from page-locked host memory --
ShardedStep.train_staged / .predict then run the library load, the all-reduce on torch's stream and
the event ordering against the engine's stream exactly as an N-rank job does.  A sum over one rank
is the identity, so everything must be bit-identical to the plain one-engine step."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import assert_bitwise, assert_state_bitwise, STRESS_HP  # noqa: E402

pytestmark = pytest.mark.gpu

F, K, PER, B, NBLK = 12, 8, 40, 2048, 5


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _blocks():
    from ftrl_ffm_amd import synth
    g = synth.Generator(F, F * PER, "zipf", seed=21)
    return [g.block(B) for _ in range(NBLK)]


def test_sharded_step_over_a_real_nccl_group_is_the_plain_step(nccl_group):
    import ftrl_ffm_amd as fa
    from ftrl_ffm_amd import sharding
    dist = nccl_group
    assert dist.get_backend() == "nccl"
    nf = F * PER
    blocks = _blocks()
    ref = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F, **STRESS_HP)
    ref.fill_state(seed=6)
    want_logits = [ref.train_batch(b)[0] for b in blocks]
    want_pred = ref.predict_batch(blocks[0])[0]
    want = ref.get_state()
    ref.close()

    side = torch.cuda.Stream()  # the engine's stream: neither the default one nor torch's current one
    e = fa.Engine("FFM", nf, F, K, max_batch_rows=B, max_batch_nnz=B * F, seed=4, max_row_nnz=F,
                  stream=side.cuda_stream, **STRESS_HP)
    e.fill_state(seed=6)
    logit = torch.zeros(B, device="cuda")
    step = sharding.ShardedStep(e, dist, logit)
    pinned = []
    for b in blocks:  # page-locked, as the zero-copy staging wants them
        for name in ("row_ptr", "field", "feat", "val", "label"):
            t = torch.from_numpy(getattr(b, name)).pin_memory()
            pinned.append(t)
            setattr(b, name, t.numpy())
    busy = torch.randn(1024, 1024, device="cuda")
    staged = 0
    for i in range(NBLK):
        while staged < min(i + 3, NBLK):
            step.stage(blocks[staged], zero_copy=True)
            staged += 1
        for _ in range(3):  # torch's current stream runs behind the engine's
            busy = busy @ busy * 1e-3
        step.train_staged(B)
        e.sync()
        torch.cuda.synchronize()
        assert_bitwise(logit.cpu().numpy(), want_logits[i], "logits of block %d" % i)
    assert_state_bitwise(e.get_state(), want, "state after the RCCL-ordered steps")
    # predict through the same path
    d = {k_: torch.from_numpy(getattr(blocks[0], k_)).cuda() for k_ in ("row_ptr", "field", "feat", "val")}
    out = torch.zeros(B, device="cuda")
    step.predict(B, blocks[0].nnz, d["row_ptr"].data_ptr(), d["field"].data_ptr(), d["feat"].data_ptr(),
                 d["val"].data_ptr(), out=out.data_ptr())
    e.sync()
    torch.cuda.synchronize()
    assert_bitwise(out.cpu().numpy(), want_pred, "predict through ShardedStep")
    e.close()


def test_all_reduce_of_a_block_of_logits_runs_on_rccl(nccl_group):
    """The collective itself, at the message size of the path (8192 floats), on a side stream."""
    dist = nccl_group
    t = torch.arange(8192, dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(4):
            dist.all_reduce(t)
    s.synchronize()
    assert torch.equal(t.cpu(), torch.arange(8192, dtype=torch.float32))
