#!/usr/bin/env python3
"""Runs a command, prints its wall time and peak resident set (ru_maxrss of the child)."""
import resource
import subprocess
import sys
import time

t0 = time.time()
rc = subprocess.call(sys.argv[1:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("wall %.2f s  max RSS %.0f MB  (exit %d)" % (time.time() - t0, ru.ru_maxrss / 1024.0, rc))
sys.exit(rc)
