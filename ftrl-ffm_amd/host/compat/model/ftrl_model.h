// model/ftrl_model.h -- the reference's include path for this header (src/include/model/ftrl_model.h:14-51); forwards to the host mirror.
// Source-level drop-in: /root/reference/src/main.cpp and tests/test_*.cpp compile against this tree
// where they lie (tests/test_compat_compile.py).
#pragma once
#include "../../ftrl_model.h"
#include "../utils/utils.h"
