// utils/utils.h -- the reference's include path for this header (src/include/utils/utils.h:15-104); forwards to the host mirror.
// Source-level drop-in: /root/reference/src/main.cpp and tests/test_*.cpp compile against this tree
// where they lie (tests/test_compat_compile.py).
#pragma once
#include "../../utils.h"
