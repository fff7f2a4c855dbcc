// eval/loss.h -- the reference's include path for this header (src/include/eval/loss.h:8-12); forwards to the host mirror.
// Source-level drop-in: /root/reference/src/main.cpp and tests/test_*.cpp compile against this tree
// where they lie (tests/test_compat_compile.py).
#pragma once
#include "../../ftrl_model.h"
using ftrl::loss;  // the reference declares loss(int, double) at global scope
