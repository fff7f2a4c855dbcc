"""CPU-side checks of the drop-in boundary: libffm_engine.so builds for gfx950, loads, and
exports every symbol include/ffm_engine.h declares; argument errors are reported through the C
error convention; the product path never falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import ftrl_ffm_amd as fa
from ftrl_ffm_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ffm_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffm_(?:engine|group)_\w+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    fa.build()
    lib = fa.load_library()
    syms = header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), "libffm_engine.so lacks " + name
    bound = {n for n, _, _ in fa.ABI}
    assert set(syms) == bound, set(syms) ^ bound
    assert lib.ffm_engine_abi_version() == 4


def test_config_struct_matches_header_defaults():
    """ffm_engine_default_config carries the reference defaults (cmd_option.h:49-63)."""
    lib = fa.load_library()
    cfg = fa.Config()
    lib.ffm_engine_default_config(ctypes.byref(cfg))
    assert (cfg.model_type, cfg.n_feats, cfg.n_fields, cfg.n_factors) == (2, 10000, 8, 16)
    assert abs(cfg.w_alpha - 1e-4) < 1e-10 and cfg.w_beta == 1.0
    assert abs(cfg.w_l1 - 0.1) < 1e-7 and cfg.w_l2 == 5.0
    assert cfg.init_mean == 0.0 and abs(cfg.init_stddev - 0.02) < 1e-8
    assert cfg.n_shards == 1 and cfg.shard_rank == 0


def test_invalid_arguments_use_the_c_error_convention():
    lib = fa.load_library()
    h = ctypes.c_void_p()
    cfg = fa.Config()
    lib.ffm_engine_default_config(ctypes.byref(cfg))
    cfg.model_type = 7  # the reference throws std::invalid_argument("invalid model_type")
    rc = lib.ffm_engine_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -1 and b"model_type" in lib.ffm_engine_last_error()
    cfg.model_type = 2
    cfg.n_feats = 0
    assert lib.ffm_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.n_feats = 10
    cfg.n_shards, cfg.shard_rank = 2, 2
    assert lib.ffm_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert lib.ffm_engine_sync(None) == -1
    assert lib.ffm_engine_train_batch(None, 0, None, None, None, None, None, None, None) == -1


def test_no_cpu_fallback_without_a_gpu():
    """Without a HIP device the engine refuses to exist (FFM_E_DEVICE) instead of computing on
    the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fa.EngineError) as ei:
        fa.Engine("FFM", 100, 4, 4)
    assert ei.value.code == -2


def test_product_does_not_reference_the_oracle():
    """Nothing under ftrl-ffm_amd/ may import, link or call the checker."""
    pkg = os.path.join(ROOT, "ftrl-ffm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in text and "ffm_oracle" not in text and "fo_train" not in text, f
    out = os.popen("ldd '%s' 2>/dev/null" % fa.LIB_PATH).read()
    assert "oracle" not in out


def test_synthetic_generator_shape():
    """one feature per field, per-field disjoint id ranges, last field numeric."""
    F, nf = 8, 10000
    for dist in ("zipf", "uniform"):
        g = synth.Generator(F, nf, dist, seed=42)
        b = g.block(1000)
        assert b.n_rows == 1000 and b.nnz == 1000 * F
        per = nf // F
        feat = b.feat.reshape(1000, F)
        assert ((feat // per) == np.arange(F)[None, :]).all()
        assert (b.field.reshape(1000, F) == np.arange(F)[None, :]).all()
        val = b.val.reshape(1000, F)
        assert (val[:, :F - 1] == 1.0).all() and (val[:, F - 1] > 0).all() and (val[:, F - 1] <= 1).all()
        assert set(np.unique(b.label)) <= {0, 1}
        b2 = synth.Generator(F, nf, dist, seed=42).block(1000)
        assert np.array_equal(b.feat, b2.feat) and np.array_equal(b.label, b2.label)
    text = synth.to_libffm_text(b.rows(0, 2))
    assert text.count("\n") == 2 and text.split()[1].count(":") == 2


def test_page_aligned_arrays_own_their_pages():
    """What Engine.pin_block should be given (hipHostRegister locks whole pages): page-aligned,
    writable, and not sharing its mapping with another array."""
    import mmap

    import numpy as np

    import ftrl_ffm_amd as fa
    a = fa.page_aligned(1000, np.int32)
    b = fa.page_aligned(1, np.float32)
    for x in (a, b):
        assert x.ctypes.data % mmap.PAGESIZE == 0
    assert a.size == 1000 and a.dtype == np.int32 and b.size == 1
    a[:] = 7
    b[0] = 1.5
    assert int(a.sum()) == 7000 and float(b[0]) == 1.5
    lo, hi = sorted((a.ctypes.data, b.ctypes.data))
    assert hi - lo >= mmap.PAGESIZE
