"""Synthetic libffm-shaped data, the measurement inputs of bench.py and
the large-size parity tests.  Mirrors the shape of the reference's bundled data and of
python/generate_data.py's output: one feature per field, fields 0..F-1 in order, field f owning
the contiguous id range [f*per, (f+1)*per) with per = n_feats // F (so ids are distinct within a
row), id inside a field drawn Zipf(s) over a per-field random permutation (or uniform), value 1.0
except the last field, which is round(U(0,1), 4), label ~ Bernoulli(sigmoid(planted linear model)).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Block:
    """Rows in the engine's CSR wire format."""
    row_ptr: np.ndarray
    field: np.ndarray
    feat: np.ndarray
    val: np.ndarray
    label: np.ndarray

    @property
    def n_rows(self):
        return len(self.row_ptr) - 1

    @property
    def nnz(self):
        return int(self.row_ptr[-1])

    def rows(self, lo, hi):
        b, e = int(self.row_ptr[lo]), int(self.row_ptr[hi])
        return Block((self.row_ptr[lo:hi + 1] - b).astype(np.int32), self.field[b:e].copy(),
                     self.feat[b:e].copy(), self.val[b:e].copy(), self.label[lo:hi].copy())


class Generator:
    def __init__(self, n_fields, n_feats, dist="zipf", zipf_s=1.1, seed=42):
        self.F = int(n_fields)
        self.per = int(n_feats) // self.F
        assert self.per >= 1, "need at least one id per field"
        self.n_feats = int(n_feats)
        self.dist = dist
        self.rng = np.random.default_rng(seed)
        if dist == "zipf":
            w = 1.0 / np.power(np.arange(1, self.per + 1, dtype=np.float64), zipf_s)
            self.cdf = np.cumsum(w)
            self.cdf /= self.cdf[-1]
            self.perm = [self.rng.permutation(self.per).astype(np.int32) for _ in range(self.F)]
        elif dist != "uniform":
            raise ValueError("dist must be 'zipf' or 'uniform'")
        # planted sparse linear model so that the loss is learnable
        self.planted = np.zeros(self.F * self.per, np.float32)
        hot = self.rng.random(self.F * self.per) < 0.05
        self.planted[hot] = self.rng.normal(0, 1.0, int(hot.sum())).astype(np.float32)

    def block(self, n_rows):
        F, per, rng = self.F, self.per, self.rng
        if self.dist == "zipf":
            rank = np.searchsorted(self.cdf, rng.random((n_rows, F)), side="left")
            rank = np.minimum(rank, per - 1)
            local = np.empty((n_rows, F), np.int64)
            for f in range(F):
                local[:, f] = self.perm[f][rank[:, f]]
        else:
            local = rng.integers(0, per, size=(n_rows, F))
        feat = (local + np.arange(F, dtype=np.int64)[None, :] * per).astype(np.int32)
        val = np.ones((n_rows, F), np.float32)
        last = np.round(rng.random(n_rows), 4).astype(np.float32)
        val[:, F - 1] = np.maximum(last, np.float32(1e-4))  # the parsers drop zeros
        logit = (self.planted[feat] * val).sum(axis=1)
        label = (rng.random(n_rows) < 1.0 / (1.0 + np.exp(-logit))).astype(np.int32)
        field = np.broadcast_to(np.arange(F, dtype=np.int32)[None, :], (n_rows, F))
        row_ptr = (np.arange(n_rows + 1, dtype=np.int64) * F).astype(np.int32)
        return Block(row_ptr, np.ascontiguousarray(field).reshape(-1), feat.reshape(-1),
                     val.reshape(-1), label)


def to_libffm_text(block, libsvm=False):
    """Rows as libffm ("label field:feat:val ...") or libsvm ("label feat:val ...") lines."""
    lines = []
    for r in range(block.n_rows):
        b, e = int(block.row_ptr[r]), int(block.row_ptr[r + 1])
        toks = [str(int(block.label[r]))]
        for p in range(b, e):
            v = "%.6g" % float(block.val[p])
            toks.append(("%d:%s" % (block.feat[p], v)) if libsvm
                        else ("%d:%d:%s" % (block.field[p], block.feat[p], v)))
        lines.append(" ".join(toks))
    return "\n".join(lines) + "\n"
