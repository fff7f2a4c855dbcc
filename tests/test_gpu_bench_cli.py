"""The driver's command shapes for bench.py, run as the driver runs them (a child process, no
launcher around it) on the one GPU there is: `--gpus 2` must start its own ranks, train the
field-pair-sharded model through the staged host path and print ONE JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_launches_its_own_ranks_and_reports_both_scalings():
    d = _run(["--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "3", "--warmup", "1",
              "--n-feats", "780000", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "samples/s"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["rows_per_step"] == 2 * 8192
    assert d["config"]["sharding"].startswith("field-pair x2")
    o = d["other_scaling"]
    assert o["scaling"] == "strong" and o["rows_per_step"] == 8192 and o["value"] > 0
    # what each block size costs in logloss against the per-sample loop rides along (north-star bound 1e-4)
    assert abs(d["block_logloss_cost"]["d_eval_logloss"]) < 1e-4
    assert abs(o["block_logloss_cost"]["d_eval_logloss"]) < 1e-4
    assert 0.5 < d["train_logloss"] < 0.9


def test_gpus_1_default_shape_line_has_roofline():
    d = _run(["--steps", "3", "--warmup", "2", "--n-feats", "780000", "--no-cpu-baseline", "--no-eval"])
    assert d["n_gpus"] == 1 and d["config"]["sharding"] == "none"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
