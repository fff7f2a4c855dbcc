"""ctypes binding of the C ABI in include/ffm_engine.h (libffm_engine.so, hand-written HIP for
gfx950).  This is plumbing only: every call goes straight to the shared library, and there is no
CPU fallback -- if the library or a GPU is missing, loading / Engine() raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libffm_engine.so")

LR, FM, FFM = 0, 1, 2
MODEL_TYPES = {"LR": LR, "FM": FM, "FFM": FFM}
FLAG_SKIP_INIT = 1
FLAG_LEARN = 4

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ffm_engine error %d: %s" % (code, msg))
        self.code = code


class Config(ctypes.Structure):
    """struct ffm_engine_config (include/ffm_engine.h)."""
    _fields_ = [("model_type", ctypes.c_int32), ("n_feats", ctypes.c_int32),
                ("n_fields", ctypes.c_int32), ("n_factors", ctypes.c_int32),
                ("w_alpha", ctypes.c_float), ("w_beta", ctypes.c_float),
                ("w_l1", ctypes.c_float), ("w_l2", ctypes.c_float),
                ("init_mean", ctypes.c_float), ("init_stddev", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("max_batch_rows", ctypes.c_int32),
                ("max_batch_nnz", ctypes.c_int32), ("device_id", ctypes.c_int32),
                ("n_shards", ctypes.c_int32), ("shard_rank", ctypes.c_int32),
                ("stream", ctypes.c_void_p), ("flags", ctypes.c_int32),
                ("max_row_nnz", ctypes.c_int32), ("field_start", _i32p),
                ("reserved", ctypes.c_int32 * 4)]


# every symbol include/ffm_engine.h declares: (name, restype, argtypes)
_CSR = [_i32p, _i32p, _i32p, _f32p, _i32p]
_DCSR = [_vp, _vp, _vp, _vp, _vp]
ABI = [
    ("ffm_engine_default_config", None, [ctypes.POINTER(Config)]),
    ("ffm_engine_create", ctypes.c_int, [ctypes.POINTER(Config), ctypes.POINTER(_vp)]),
    ("ffm_engine_init_weights_host", ctypes.c_int,
     [ctypes.c_uint64, ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_int64,
      ctypes.c_int64, _f32p]),
    ("ffm_engine_destroy", None, [_vp]),
    ("ffm_engine_last_error", ctypes.c_char_p, []),
    ("ffm_engine_abi_version", ctypes.c_int, []),
    ("ffm_engine_block_segment", ctypes.c_int, []),
    ("ffm_engine_row_len", ctypes.c_int64, [_vp]),
    ("ffm_engine_default_batch_ramp", ctypes.c_int32, [ctypes.c_float]),
    ("ffm_engine_shard_plan", ctypes.c_int,
     [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i32p, _i32p, _i32p]),
    ("ffm_engine_set_weights", ctypes.c_int, [_vp, _f32p, _f32p, _f32p]),
    ("ffm_engine_get_weights", ctypes.c_int, [_vp, _f32p, _f32p, _f32p]),
    ("ffm_engine_set_state", ctypes.c_int, [_vp] + [_f32p] * 6),
    ("ffm_engine_get_state", ctypes.c_int, [_vp] + [_f32p] * 6),
    ("ffm_engine_get_rows", ctypes.c_int, [_vp, ctypes.c_int32, _i32p] + [_f32p] * 6),
    ("ffm_engine_set_rows", ctypes.c_int, [_vp, ctypes.c_int32, _i32p] + [_f32p] * 6),
    ("ffm_engine_train_batch", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR + [_f32p, _f64p]),
    ("ffm_engine_predict_batch", ctypes.c_int,
     [_vp, ctypes.c_int32] + _CSR + [ctypes.c_int32, _f32p, _f64p]),
    ("ffm_engine_train_batch_device", ctypes.c_int,
     [_vp, ctypes.c_int32, ctypes.c_int32] + _DCSR + [_vp, _vp]),
    ("ffm_engine_predict_batch_device", ctypes.c_int,
     [_vp, ctypes.c_int32, ctypes.c_int32] + _DCSR + [ctypes.c_int32, _vp, _vp]),
    ("ffm_engine_predict_finish_device", ctypes.c_int,
     [_vp, ctypes.c_int32, _vp, _vp, ctypes.c_int32, _vp, _vp]),
    ("ffm_engine_train_batch_async", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR),
    ("ffm_engine_train_batch_async_pinned", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR),
    ("ffm_engine_train_flush", ctypes.c_int, [_vp, _f64p]),
    ("ffm_engine_predict_batch_async", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR + [ctypes.c_int32]),
    ("ffm_engine_stage_batch", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR + [ctypes.c_int32]),
    ("ffm_engine_blocks_pulled", ctypes.c_int64, [_vp]),
    ("ffm_engine_train_forward_staged", ctypes.c_int, [_vp, _vp]),
    ("ffm_engine_train_staged", ctypes.c_int, [_vp, _vp, _vp]),
    ("ffm_engine_pin_host", ctypes.c_int, [_vp, ctypes.c_size_t]),
    ("ffm_engine_unpin_host", ctypes.c_int, [_vp]),
    ("ffm_engine_prepare_device", ctypes.c_int,
     [_vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    ("ffm_engine_train_forward_device", ctypes.c_int,
     [_vp, ctypes.c_int32, ctypes.c_int32] + _DCSR + [_vp]),
    ("ffm_engine_train_update_device", ctypes.c_int, [_vp, _vp, _vp, _vp]),
    ("ffm_engine_fill_state", ctypes.c_int,
     [_vp, ctypes.c_uint64, ctypes.c_float, ctypes.c_float, ctypes.c_float]),
    ("ffm_engine_eval_sigmoid", ctypes.c_int, [_vp, ctypes.c_int32, _f32p, _f32p]),
    ("ffm_engine_sync", ctypes.c_int, [_vp]),
    ("ffm_engine_check_errors", ctypes.c_int, [_vp]),
    ("ffm_engine_stream", ctypes.c_void_p, [_vp]),
    ("ffm_engine_profile_enable", ctypes.c_int, [_vp, ctypes.c_int32]),
    ("ffm_engine_profile_read", ctypes.c_int,
     [_vp, _i32p, _f64p, ctypes.c_char_p, ctypes.c_size_t]),
    ("ffm_engine_profile_focus", ctypes.c_int, [_vp]),
    ("ffm_engine_profile_dump", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    # several GPUs in one process (ffm_group_*)
    ("ffm_group_create", ctypes.c_int, [ctypes.POINTER(Config), ctypes.c_int32, _i32p, ctypes.POINTER(_vp)]),
    ("ffm_group_destroy", None, [_vp]),
    ("ffm_group_size", ctypes.c_int32, [_vp]),
    ("ffm_group_engine", ctypes.c_void_p, [_vp, ctypes.c_int32]),
    ("ffm_group_collective", ctypes.c_char_p, [_vp]),
    ("ffm_group_train_batch", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR + [_f32p, _f64p]),
    ("ffm_group_train_batch_async", ctypes.c_int, [_vp, ctypes.c_int32] + _CSR + [ctypes.c_int32]),
    ("ffm_group_train_flush", ctypes.c_int, [_vp, _f64p]),
    ("ffm_group_blocks_pulled", ctypes.c_int64, [_vp]),
    ("ffm_group_predict_batch", ctypes.c_int,
     [_vp, ctypes.c_int32] + _CSR + [ctypes.c_int32, _f32p, _f64p]),
]

_lib = None


def default_batch_ramp(w_alpha):
    """The block scheduler's default ramp for a learning rate (ffm_engine_default_batch_ramp)."""
    return int(load_library().ffm_engine_default_batch_ramp(float(w_alpha)))


def shard_plan(n_fields, n_shards, field_map=False):
    """Who owns what under field-pair sharding (ffm_engine_shard_plan): dict(pair_owner [F, F] --
    shard owning the unordered field pair, i.e. both of its latent slots --, lin_owner [F] -- shard
    that adds and updates the linear terms of a field's entries --, bias_owner)."""
    lib = load_library()
    po = np.zeros((n_fields, n_fields), np.int32)
    lo = np.zeros(n_fields, np.int32)
    bo = ctypes.c_int32(0)
    rc = lib.ffm_engine_shard_plan(int(n_fields), int(n_shards), int(bool(field_map)), _i(po), _i(lo),
                                   ctypes.byref(bo))
    if rc != 0:
        raise EngineError(rc, lib.ffm_engine_last_error().decode())
    return dict(pair_owner=po, lin_owner=lo, bias_owner=int(bo.value))


def init_weights_host(seed, mean, stddev, latent, first, count):
    """The weights a fresh engine holds, recomputed on the host (ffm_engine_init_weights_host)."""
    out = np.empty(int(count), np.float32)
    rc = load_library().ffm_engine_init_weights_host(int(seed), mean, stddev, int(bool(latent)),
                                                     int(first), int(count), _f(out))
    if rc != 0:
        raise EngineError(rc, load_library().ffm_engine_last_error().decode())
    return out


def load_library(path=None):
    """dlopen libffm_engine.so and bind every ABI symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("FFM_ENGINE_LIB") or LIB_PATH  # env: A/B experiments only
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build it with `python ftrl-ffm_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, restype, argtypes in ABI:
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def page_aligned(n, dtype):
    """A numpy array of n elements in an anonymous mapping of its own: page-aligned, whole pages,
    shared with nothing else -- what Engine.pin_block should be given.  (hipHostRegister locks whole
    pages; an array in the middle of the Python heap shares its pages with whatever lives next to
    it, and the runtime then meets half-registered ranges in later copies.)"""
    import mmap
    nbytes = max(1, int(n)) * np.dtype(dtype).itemsize
    size = (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
    return np.frombuffer(mmap.mmap(-1, size), dtype=dtype, count=int(n))


def _f(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _i(a):
    return None if a is None else a.ctypes.data_as(_i32p)


STATE_KEYS = ("bias3", "lin_w", "lin_n", "lin_z", "vec_w", "vec_n", "vec_z")


class Engine:
    """One LR / FM / FFM model resident in HBM (mirrors ftrl::FtrlModel for blocks of rows)."""

    def __init__(self, model_type="FFM", n_feats=10000, n_fields=8, n_factors=16, w_alpha=1e-4,
                 w_beta=1.0, w_l1=0.1, w_l2=5.0, init_mean=0.0, init_stddev=0.02, seed=42,
                 max_batch_rows=8192, max_batch_nnz=None, device_id=0, n_shards=1, shard_rank=0,
                 stream=None, skip_init=False, max_row_nnz=0, learn=False, field_start=None):
        self.lib = load_library()
        cfg = Config()
        self.lib.ffm_engine_default_config(ctypes.byref(cfg))
        cfg.model_type = MODEL_TYPES[model_type] if isinstance(model_type, str) else int(model_type)
        cfg.n_feats, cfg.n_fields, cfg.n_factors = int(n_feats), int(n_fields), int(n_factors)
        cfg.w_alpha, cfg.w_beta, cfg.w_l1, cfg.w_l2 = w_alpha, w_beta, w_l1, w_l2
        cfg.init_mean, cfg.init_stddev, cfg.seed = init_mean, init_stddev, int(seed)
        cfg.max_batch_rows = int(max_batch_rows)
        cfg.max_batch_nnz = int(max_batch_nnz if max_batch_nnz else max_batch_rows * 64)
        cfg.device_id, cfg.n_shards, cfg.shard_rank = int(device_id), int(n_shards), int(shard_rank)
        cfg.stream = stream
        cfg.flags = (FLAG_SKIP_INIT if skip_init else 0) | (FLAG_LEARN if learn else 0)
        cfg.max_row_nnz = int(max_row_nnz)
        self._field_start = None
        if field_start is not None:
            self._field_start = np.ascontiguousarray(field_start, np.int32)
            cfg.field_start = _i(self._field_start)
        self.cfg = cfg
        self.h = _vp()
        self._check(self.lib.ffm_engine_create(ctypes.byref(cfg), ctypes.byref(self.h)))
        self.model_type = cfg.model_type
        self.n_feats, self.n_fields, self.n_factors = cfg.n_feats, cfg.n_fields, cfg.n_factors
        self.row_len = int(self.lib.ffm_engine_row_len(self.h))

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.lib.ffm_engine_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ffm_engine_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ----
    def zero_state(self):
        nf, L = self.n_feats, self.row_len
        return dict(bias3=np.zeros(3, np.float32), lin_w=np.zeros(nf, np.float32),
                    lin_n=np.zeros(nf, np.float32), lin_z=np.zeros(nf, np.float32),
                    vec_w=np.zeros((nf, L), np.float32), vec_n=np.zeros((nf, L), np.float32),
                    vec_z=np.zeros((nf, L), np.float32))

    def get_state(self):
        st = self.zero_state()
        b = np.zeros(3, np.float32)
        vw = st["vec_w"] if self.row_len else None
        vn = st["vec_n"] if self.row_len else None
        vz = st["vec_z"] if self.row_len else None
        self._check(self.lib.ffm_engine_get_weights(self.h, _f(b[0:1]), _f(st["lin_w"]), _f(vw)))
        self._check(self.lib.ffm_engine_get_state(self.h, _f(b[1:2]), _f(b[2:3]), _f(st["lin_n"]),
                                                  _f(st["lin_z"]), _f(vn), _f(vz)))
        st["bias3"] = b
        return st

    def set_state(self, st):
        st = {k: np.ascontiguousarray(v, np.float32) for k, v in st.items()}
        b = st.get("bias3")
        g = lambda k: st[k] if k in st and st[k].size else None  # noqa: E731
        self._check(self.lib.ffm_engine_set_weights(
            self.h, _f(b[0:1].copy()) if b is not None else None, _f(g("lin_w")), _f(g("vec_w"))))
        self._check(self.lib.ffm_engine_set_state(
            self.h, _f(b[1:2].copy()) if b is not None else None,
            _f(b[2:3].copy()) if b is not None else None, _f(g("lin_n")), _f(g("lin_z")),
            _f(g("vec_n")), _f(g("vec_z"))))

    ROW_KEYS = ("lin_w", "lin_n", "lin_z", "vec_w", "vec_n", "vec_z")

    def get_rows(self, ids):
        """State of the listed features only: dict of lin_* [n] and vec_* [n, row_len]."""
        ids = np.ascontiguousarray(ids, np.int32)
        n, L = ids.size, self.row_len
        out = {k: np.zeros((n, L) if k.startswith("vec") else n, np.float32) for k in self.ROW_KEYS}
        args = [_f(out[k]) if out[k].size else None for k in self.ROW_KEYS]
        self._check(self.lib.ffm_engine_get_rows(self.h, n, _i(ids), *args))
        return out

    def set_rows(self, ids, st):
        """Overwrites the listed features' state with the given arrays (any subset of ROW_KEYS)."""
        ids = np.ascontiguousarray(ids, np.int32)
        st = {k: np.ascontiguousarray(v, np.float32) for k, v in st.items() if k in self.ROW_KEYS}
        args = [_f(st[k]) if k in st and st[k].size else None for k in self.ROW_KEYS]
        self._check(self.lib.ffm_engine_set_rows(self.h, ids.size, _i(ids), *args))

    # ---- blocks of rows in host memory ----
    def _csr(self, c):
        # numpy's .ctypes.data_as builds a fresh ctypes object per call (several microseconds each):
        # for a block that is handed over again and again (a trainer's ring of page-locked blocks)
        # the argument tuple is kept on the block, keyed by the identity of its arrays
        fld = c.field if (self.model_type == FFM or c.field is not None) else None
        key = (c.n_rows, id(c.row_ptr), id(fld), id(c.feat), id(c.val), id(c.label))
        cached = getattr(c, "_ffm_csr_args", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        args = (c.n_rows, _i(c.row_ptr), _i(fld), _i(c.feat), _f(c.val), _i(c.label))
        try:
            # (the arrays ride along: while they are referenced here their ids cannot be reused)
            c._ffm_csr_args = (key, args, (c.row_ptr, fld, c.feat, c.val, c.label))
        except AttributeError:  # (a block type without a __dict__)
            pass
        return args

    def train_batch(self, c):
        """One block with the engine's batch semantics.  Returns (logits, loss_sum)."""
        out = np.zeros(max(c.n_rows, 1), np.float32)
        loss = ctypes.c_double(0.0)
        self._check(self.lib.ffm_engine_train_batch(self.h, *self._csr(c), _f(out),
                                                    ctypes.byref(loss)))
        return out[:c.n_rows], float(loss.value)

    def train_batch_async(self, c):
        """Pipelined: stages and groups this block, trains the one passed by the previous call."""
        self._check(self.lib.ffm_engine_train_batch_async(self.h, *self._csr(c)))

    def train_batch_async_pinned(self, c):
        """The same for a block in page-locked memory (pin_block): no host copy, three blocks in
        flight; the block stays untouched until blocks_pulled() has reached its ordinal."""
        self._check(self.lib.ffm_engine_train_batch_async_pinned(self.h, *self._csr(c)))

    def stage_batch(self, c, zero_copy=False):
        """Host block -> (pinned slot ->) HBM + grouping on the side stream (returns at once).
        zero_copy: the block's arrays are page-locked (pin_block) and stay untouched until trained."""
        self._check(self.lib.ffm_engine_stage_batch(self.h, *self._csr(c), int(zero_copy)))

    def blocks_pulled(self):
        """How many staged blocks have been uploaded so far (their host arrays may be reused)."""
        return int(self.lib.ffm_engine_blocks_pulled(self.h))

    def train_staged(self, logit_out=None, loss_sum_out=None):
        """The whole step on the oldest staged block (unsharded engines; device outputs)."""
        self._check(self.lib.ffm_engine_train_staged(self.h, logit_out, loss_sum_out))

    def pin_block(self, c):
        """Page-locks the block's five arrays in place (for stage_batch(zero_copy=True)).  Give it
        arrays that own their pages (page_aligned)."""
        for a in (c.row_ptr, c.field, c.feat, c.val, c.label):
            if a is not None and a.size:
                self._check(self.lib.ffm_engine_pin_host(a.ctypes.data, a.nbytes))

    def unpin_block(self, c):
        for a in (c.row_ptr, c.field, c.feat, c.val, c.label):
            if a is not None and a.size:
                self.lib.ffm_engine_unpin_host(a.ctypes.data)

    def train_forward_staged(self, partial_logit=None):
        """Phase 1 on the oldest staged block; follow with train_update_device."""
        self._check(self.lib.ffm_engine_train_forward_staged(self.h, partial_logit))

    def train_flush(self):
        """Trains the last staged block, waits; the loss sum of all blocks since the last flush."""
        loss = ctypes.c_double(0.0)
        self._check(self.lib.ffm_engine_train_flush(self.h, ctypes.byref(loss)))
        return float(loss.value)

    def predict_batch_async(self, c, zero_copy=False):
        """Pipelined evaluation: uploads the block on the side stream and predicts it; the loss sum
        of all blocks since the last flush comes back from train_flush()."""
        self._check(self.lib.ffm_engine_predict_batch_async(self.h, *self._csr(c), int(zero_copy)))

    def train_rows(self, c):
        """Row after row (n_rows == 1 per call): the reference's sequential train() loop."""
        logits = np.zeros(c.n_rows, np.float32)
        total = 0.0
        for r in range(c.n_rows):
            lg, ls = self.train_batch(c.rows(r, r + 1))
            logits[r] = lg[0]
            total += ls
        return logits, total

    def predict_batch(self, c, output_prob=False, with_loss=True):
        out = np.zeros(max(c.n_rows, 1), np.float32)
        loss = ctypes.c_double(0.0)
        n, rp, fld, ft, v, lab = self._csr(c)
        self._check(self.lib.ffm_engine_predict_batch(self.h, n, rp, fld, ft, v,
                                                      lab if with_loss else None,
                                                      int(output_prob), _f(out),
                                                      ctypes.byref(loss)))
        return out[:c.n_rows], float(loss.value)

    # ---- blocks already in HBM (raw device addresses as ints) ----
    def train_batch_device(self, n_rows, nnz, row_ptr, field, feat, val, label, logit_out=None,
                           loss_sum_out=None):
        self._check(self.lib.ffm_engine_train_batch_device(self.h, n_rows, nnz, row_ptr, field, feat,
                                                           val, label, logit_out, loss_sum_out))

    def prepare_device(self, n_rows, nnz, row_ptr, field, feat, val):
        """Look-ahead: group the next block on a side stream (see include/ffm_engine.h)."""
        self._check(self.lib.ffm_engine_prepare_device(self.h, n_rows, nnz, row_ptr, field, feat, val))

    def train_forward_device(self, n_rows, nnz, row_ptr, field, feat, val, label, partial_logit):
        self._check(self.lib.ffm_engine_train_forward_device(self.h, n_rows, nnz, row_ptr, field,
                                                             feat, val, label, partial_logit))

    def train_update_device(self, logit, logit_out=None, loss_sum_out=None):
        self._check(self.lib.ffm_engine_train_update_device(self.h, logit, logit_out, loss_sum_out))

    def predict_batch_device(self, n_rows, nnz, row_ptr, field, feat, val, label, output_prob,
                             out, loss_sum_out=None):
        self._check(self.lib.ffm_engine_predict_batch_device(self.h, n_rows, nnz, row_ptr, field,
                                                             feat, val, label, int(output_prob),
                                                             out, loss_sum_out))

    def predict_finish_device(self, n_rows, logit, label, output_prob, out, loss_sum_out=None):
        """Second phase of a sharded predict: full logits (summed across shards) -> out / loss."""
        self._check(self.lib.ffm_engine_predict_finish_device(self.h, n_rows, logit, label,
                                                              int(output_prob), out, loss_sum_out))

    def fill_state(self, seed=7, n_lo=0.05, n_hi=1.0, z_stddev=0.3):
        """Warm random accumulators drawn on the device (measurement utility)."""
        self._check(self.lib.ffm_engine_fill_state(self.h, int(seed), n_lo, n_hi, z_stddev))

    def eval_sigmoid(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._check(self.lib.ffm_engine_eval_sigmoid(self.h, x.size, _f(x), _f(y)))
        return y

    def sync(self):
        """Waits for the stream; raises EngineError if the device flagged a block (-4: a row longer
        than max_row_nnz reached a _device entry point; that block was skipped)."""
        self._check(self.lib.ffm_engine_sync(self.h))

    check_errors = sync

    @property
    def stream(self):
        """The hipStream_t (as an int) the engine runs on."""
        return int(self.lib.ffm_engine_stream(self.h) or 0)

    # ---- kernel timing (HIP events on the engine's stream) ----
    def profile_enable(self, on=True):
        self._check(self.lib.ffm_engine_profile_enable(self.h, int(on)))

    def profile_focus(self):
        """Keep timing only the kernel that dominated so far (cheap enough for timed regions)."""
        self._check(self.lib.ffm_engine_profile_focus(self.h))

    def profile_read(self):
        n = ctypes.c_int32(0)
        ms = ctypes.c_double(0.0)
        name = ctypes.create_string_buffer(128)
        self._check(self.lib.ffm_engine_profile_read(self.h, ctypes.byref(n), ctypes.byref(ms),
                                                     name, 128))
        return name.value.decode(), int(n.value), float(ms.value)

    def profile_dump(self):
        buf = ctypes.create_string_buffer(4096)
        self._check(self.lib.ffm_engine_profile_dump(self.h, buf, 4096))
        return buf.value.decode()


class Group:
    """Several field-pair shard engines in ONE process (ffm_group_*): one engine per entry of
    `devices`, one RCCL all-reduce of the partial logits per block when the devices are distinct
    (a kernel sum when they share a device: one-GPU dry run of the orchestration)."""

    def __init__(self, devices, model_type="FFM", n_feats=10000, n_fields=8, n_factors=16, w_alpha=1e-4,
                 w_beta=1.0, w_l1=0.1, w_l2=5.0, init_mean=0.0, init_stddev=0.02, seed=42,
                 max_batch_rows=8192, max_batch_nnz=None, skip_init=False, max_row_nnz=0,
                 field_start=None):
        self.lib = load_library()
        cfg = Config()
        self.lib.ffm_engine_default_config(ctypes.byref(cfg))
        cfg.model_type = MODEL_TYPES[model_type] if isinstance(model_type, str) else int(model_type)
        cfg.n_feats, cfg.n_fields, cfg.n_factors = int(n_feats), int(n_fields), int(n_factors)
        cfg.w_alpha, cfg.w_beta, cfg.w_l1, cfg.w_l2 = w_alpha, w_beta, w_l1, w_l2
        cfg.init_mean, cfg.init_stddev, cfg.seed = init_mean, init_stddev, int(seed)
        cfg.max_batch_rows = int(max_batch_rows)
        cfg.max_batch_nnz = int(max_batch_nnz if max_batch_nnz else max_batch_rows * 64)
        cfg.flags = FLAG_SKIP_INIT if skip_init else 0
        cfg.max_row_nnz = int(max_row_nnz)
        self._field_start = None
        if field_start is not None:
            self._field_start = np.ascontiguousarray(field_start, np.int32)
            cfg.field_start = _i(self._field_start)
        dev = np.ascontiguousarray(devices, np.int32)
        self.h = _vp()
        rc = self.lib.ffm_group_create(ctypes.byref(cfg), dev.size, _i(dev), ctypes.byref(self.h))
        if rc != 0:
            raise EngineError(rc, self.lib.ffm_engine_last_error().decode())
        self.size = int(self.lib.ffm_group_size(self.h))
        self.collective = self.lib.ffm_group_collective(self.h).decode()
        # borrowed handles to the shards (for set_state / get_state in tests); the group owns them
        self.engines = []
        for r in range(self.size):
            e = Engine.__new__(Engine)
            e.lib, e.cfg, e.model_type = self.lib, cfg, cfg.model_type
            e.h = _vp(self.lib.ffm_group_engine(self.h, r))
            e._field_start = self._field_start
            e.n_feats, e.n_fields, e.n_factors = cfg.n_feats, cfg.n_fields, cfg.n_factors
            e.row_len = int(self.lib.ffm_engine_row_len(e.h))
            e.close = lambda: None  # noqa: E731  (never destroys: the group does)
            self.engines.append(e)

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.lib.ffm_engine_last_error().decode())

    def _csr(self, c):
        return self.engines[0]._csr(c)

    def train_batch(self, c):
        out = np.zeros(max(c.n_rows, 1), np.float32)
        loss = ctypes.c_double(0.0)
        self._check(self.lib.ffm_group_train_batch(self.h, *self._csr(c), _f(out), ctypes.byref(loss)))
        return out[:c.n_rows], float(loss.value)

    def train_batch_async(self, c, zero_copy=False):
        self._check(self.lib.ffm_group_train_batch_async(self.h, *self._csr(c), int(zero_copy)))

    def train_flush(self):
        loss = ctypes.c_double(0.0)
        self._check(self.lib.ffm_group_train_flush(self.h, ctypes.byref(loss)))
        return float(loss.value)

    def blocks_pulled(self):
        return int(self.lib.ffm_group_blocks_pulled(self.h))

    def predict_batch(self, c, output_prob=False, with_loss=True):
        out = np.zeros(max(c.n_rows, 1), np.float32)
        loss = ctypes.c_double(0.0)
        n, rp, fld, ft, v, lab = self._csr(c)
        self._check(self.lib.ffm_group_predict_batch(self.h, n, rp, fld, ft, v, lab if with_loss else None,
                                                     int(output_prob), _f(out), ctypes.byref(loss)))
        return out[:c.n_rows], float(loss.value)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ffm_group_destroy(self.h)
            self.h = _vp()
            self.engines = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
