"""Field-pair sharding of the FFM latent tensor over the GPUs of one node (DESIGN.md "Multi-GPU").

Both latent slots of a pair -- (feature i, field of j) and (feature j, field of i) -- belong to the
rank that owns the unordered field pair {field_i, field_j}, so forward, gradient and FTRL update of
a pair are local; the only exchange per block is ONE all-reduce (sum) of n_rows partial logits
(RCCL over xGMI when the process group is "nccl"; "gloo" in the CPU tests).  Who owns which pair,
which field's linear terms and the bias is ffm_engine_shard_plan's answer (contiguous blocks of the
field x field triangle), read here through engine.shard_plan.
"""
import numpy as np

from .engine import shard_plan  # noqa: F401  (the partition itself is the library's: ffm_engine_shard_plan)


def pair_owner(field_a, field_b, n_fields, n_shards):
    """Shard owning the unordered field pair (ints or numpy arrays), per ffm_engine_shard_plan."""
    return shard_plan(n_fields, n_shards)["pair_owner"][field_a, field_b]


def owned_pair_counts(n_fields, n_shards):
    """How many of the n_fields*(n_fields-1)/2 cross-field pairs each shard owns: the load balance
    of the partition (same-field pairs, which only multi-valued fields produce, not counted)."""
    own = shard_plan(n_fields, n_shards)["pair_owner"]
    iu = np.triu_indices(n_fields, 1)
    return np.bincount(own[iu], minlength=n_shards)


class ShardedStep:
    """One training block on a sharded engine: local forward -> all-reduce -> local update.

    `engine` needs train_forward_device / train_update_device (ftrl_ffm_amd.Engine);
    `dist` is torch.distributed (initialised) or None for a single shard;
    `logit` is a torch tensor of n_rows floats on the engine's device that receives the partial
    and then the summed logits.

    Stream ordering: the engine enqueues on ITS stream (ffm_engine_stream: the one given at create
    or its own), the collective on torch's current stream.  When they differ, the collective is
    made to wait for the forward and the update for the collective (events, no host sync)."""

    def __init__(self, engine, dist, logit):
        self.engine, self.dist, self.logit = engine, dist, logit
        self._ext = None
        if getattr(logit, "is_cuda", False) and getattr(engine, "stream", 0):
            import torch
            self._torch = torch
            self._ext = torch.cuda.ExternalStream(engine.stream, device=logit.device)

    def _exchange(self, n_rows):
        if self.dist is None:
            return
        # (a process group of ONE rank still runs the collective: a sum over one rank, but RCCL, its
        # stream and the event ordering below are then the ones an N-rank job uses)
        cur = None
        if self._ext is not None:
            cur = self._torch.cuda.current_stream(self.logit.device)
            if cur.cuda_stream != self._ext.cuda_stream:
                cur.wait_stream(self._ext)      # partial logits are complete before the sum
            else:
                cur = None
        self.dist.all_reduce(self.logit[:n_rows])  # the path's one collective
        if cur is not None:
            self._ext.wait_stream(cur)          # and summed before the engine consumes them

    def _inputs_ready(self):
        """Device CSR arrays are usually produced on torch's current stream (non-blocking H2D,
        preprocessing kernels); the engine reads them on ITS stream, so that stream first waits for
        what the current stream has enqueued so far (read-before-write otherwise)."""
        if self._ext is None:
            return
        cur = self._torch.cuda.current_stream(self.logit.device)
        if cur.cuda_stream != self._ext.cuda_stream:
            self._ext.wait_stream(cur)

    def __call__(self, n_rows, nnz, row_ptr, field, feat, val, label, loss_sum_out=None):
        ptr = self.logit.data_ptr()
        self._inputs_ready()
        self.engine.train_forward_device(n_rows, nnz, row_ptr, field, feat, val, label, ptr)
        self._exchange(n_rows)
        self.engine.train_update_device(ptr, None, loss_sum_out)

    def stage(self, block, zero_copy=False):
        """Hand the NEXT host block to the engine: it is uploaded and grouped on the engine's side
        stream while the current one trains (up to two may wait)."""
        self.engine.stage_batch(block, zero_copy)

    def train_staged(self, n_rows, loss_sum_out=None):
        """One training block from the oldest staged host block: forward -> all-reduce -> update."""
        ptr = self.logit.data_ptr()
        self.engine.train_forward_staged(ptr)
        self._exchange(n_rows)
        self.engine.train_update_device(ptr, None, loss_sum_out)

    def predict(self, n_rows, nnz, row_ptr, field, feat, val, label=None, output_prob=False,
                out=None, loss_sum_out=None):
        """predict() on the sharded model: partial logits -> all-reduce -> value / logloss."""
        ptr = self.logit.data_ptr()
        self._inputs_ready()
        self.engine.predict_batch_device(n_rows, nnz, row_ptr, field, feat, val, None, False, ptr)
        self._exchange(n_rows)
        self.engine.predict_finish_device(n_rows, ptr, label, output_prob, out if out else ptr,
                                          loss_sum_out)
