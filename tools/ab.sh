#!/bin/bash
# quick A/B: serial per-kernel times + overlapped throughput of the current build
cd "$GRAFT_REPO_ROOT"
FFM_ENGINE_SERIAL=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lookahead > /tmp/s.json 2> /tmp/s.err
grep -E "row_kernel|latent" /tmp/s.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
python -c "import json;d=json.load(open('/tmp/o.json'));print('overlapped',d['value'],d['ms_per_step'])"
grep -E "latent" /tmp/o.err
