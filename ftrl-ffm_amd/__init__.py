"""MI355X-native FTRL LR/FM/FFM trainer: hand-written HIP kernels for gfx950 behind the C ABI of
include/ffm_engine.h (csrc/), plus the host-side mirror of the reference's model/trainer interface
(host/) and this thin ctypes binding (engine.py).  No CPU fallback: importing is cheap, but
creating an Engine without libffm_engine.so or without a GPU raises."""
from . import build as _build
from .engine import (ABI, FFM, FM, LR, Config, Engine, EngineError, Group, LIB_PATH,  # noqa: F401
                     default_batch_ramp, init_weights_host, load_library, page_aligned, shard_plan)


def build(force=False, verbose=False):
    """Compile libffm_engine.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    return _build.build(force=force, verbose=verbose)


def build_host(force=False, verbose=False):
    """Compile the C++ host mirror (host/): the trainer CLI and its test executable."""
    return _build.build_host(force=force, verbose=verbose)
