#!/usr/bin/env python3
"""Is the oracle a fair stand-in for the reference as a CPU *speed* baseline?  (BUILD CONTAINER
ONLY: needs /root/reference to have been compiled into oracle/_ref by `make -C oracle ref`.)

bench.py's cpu_baseline times oracle/ffm_oracle.c (kind "port"): the reference's own constructor
draws every weight from a fresh std::random_device (~32 us per weight), so the compiled reference
cannot even be constructed at the baseline's model size.  SURVEY.md 8(d) / BASELINE.md 3 therefore
ask for this check: on a shape the reference CAN construct, the port's rows/s must be within
+-15 % of the real reference's loops on the same machine --

  * 1 thread : fr_train_rows_threaded(1)  = FtrlOffline::one_epoch's loop, ftrl_offline.cpp:63-91
               vs fo_train_rows_threaded(1) / fo_train_rows
  * 8 threads: fr_train_rows_threaded(8)  = the same loop over std::threads and the model's own
               per-feature mutexes, vs fo_train_rows_threaded(8)

Prints one JSON line (committed as profiles/rNN_cpu_baseline_validation.json) and exits non-zero
when a ratio leaves [0.85, 1.15].  Best of `--repeat` runs per leg (the box is shared).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--ids-per-field", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--threads", type=int, nargs="*", default=[1, 8])
    args = ap.parse_args()
    from oracle import pyoracle
    from oracle.pyoracle import CpuModel
    from ftrl_ffm_amd import synth
    pyoracle.build(ref=True)
    if not pyoracle.have_ref():
        raise SystemExit("oracle/_ref is not built: run in the build container")
    F, k = 39, 16
    nf = F * args.ids_per_field
    blk = synth.Generator(F, nf, "zipf", seed=42).block(args.rows)
    rng = np.random.default_rng(5)
    st0 = None
    out = {"shape": "FFM F=%d k=%d n_feats=%d, %d rows Zipf(1.1), warm state" % (F, k, nf, args.rows),
           "cores": os.cpu_count(), "legs": {}}
    ok = True
    for threads in args.threads:
        best = {}
        for kind in ("ref", "oracle"):
            rates = []
            for _ in range(args.repeat):
                m = CpuModel(kind, "FFM", nf, F, k)
                if st0 is None:
                    st0 = m.zero_state()
                    st0["vec_w"][...] = rng.normal(0, 0.02, st0["vec_w"].shape).astype(np.float32)
                    st0["vec_n"][...] = rng.uniform(0.05, 1.0, st0["vec_n"].shape).astype(np.float32)
                    st0["vec_z"][...] = rng.normal(0, 0.3, st0["vec_z"].shape).astype(np.float32)
                m.set_state(st0)
                sec, _ = m.train_rows_threaded(blk, threads)
                rates.append(args.rows / sec)
                del m
            best[kind] = max(rates)
        ratio = best["oracle"] / best["ref"]
        out["legs"]["%dT" % threads] = {"reference_rows_per_s": round(best["ref"], 1),
                                        "port_rows_per_s": round(best["oracle"], 1),
                                        "port_over_reference": round(ratio, 3)}
        ok = ok and 0.85 <= ratio <= 1.15
    out["within_15_percent"] = ok
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
