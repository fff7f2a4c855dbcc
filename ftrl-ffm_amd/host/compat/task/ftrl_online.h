// task/ftrl_online.h -- the reference's include path for this header (src/include/task/ftrl_online.h:17-34); forwards to the host mirror.
// Source-level drop-in: /root/reference/src/main.cpp and tests/test_*.cpp compile against this tree
// where they lie (tests/test_compat_compile.py).
#pragma once
#include "../../trainer.h"
