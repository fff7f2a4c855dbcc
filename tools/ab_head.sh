#!/bin/bash
# The tree's build against exp_libs/libhead.so (the last commit's), alternating, per configuration.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "" "--config c2" "--config c3"; do
  export EXP_BENCH_ARGS="$c"
  echo "== ${c:-c5}"; ./tools/ab_env.sh "head" "tree" "head" "tree"
done
