#!/usr/bin/env python3
"""Turns one tools/profile_round.sh capture (gpurun_out/prof_<round>/) into the committed
evidence under profiles/: the rocprofv3 --kernel-trace --stats table as is, and a per-kernel
summary of the HBM byte counters.

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come from separate --pmc passes and are in KiB; on gfx950 FETCH_SIZE reports HALF of
the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled for the kernels whose
reads are such streams (the refresh kernel, the small-feature update kernels and -- since it
refreshes and updates the once-only features' records itself -- the training row kernel stream
whole records as float4 per lane; for the row kernel the doubled figure, 2.2 GB, matches the byte
count of its loops: 0.49 GB (n,z) + 0.98 GB re-read of (n,z,w) and partner weights + the pair
phase's misses); other access patterns (the chain kernels' 4-byte gathers) are uncalibrated and left
as reported.  FETCH_SIZE counts requests leaving the L2, Infinity-Cache hits included: the row
kernel's second read of a record a few microseconds after the first is counted like a trip to HBM.
Values are per launch (steady-state launches of the bench, averaged)."""
import csv
import json
import os
import shutil
import sys

# kernels whose reads are dominated by whole records streamed as 16 B per lane.  (The update launch
# mixes such streams -- the few-occurrence features' records -- with 64-byte gathers by four lanes,
# which the calibration counted at face value, profiles/archive/r03_fetch_calibration.json slot_read_quad16:
# it is left uncorrected, a lower bound.)
WIDE_READERS = ("ffm_refresh_kernel", "ffm_update_single_kernel", "ffm_row_kernel<true", "fm_row_wave_kernel<true")


def short(name):
    n = name.replace("void ", "").replace("ftrl_dev::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def per_kernel(path, counter):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v[-5:]) / len(v[-5:]) for k, v in out.items()}


def summarize(stats_csv, fetch_csv, write_csv, bench_json, out_path):
    """One PMC summary: per kernel the average duration (kernel trace), the corrected bytes leaving the
    L2 per launch, and -- under "_capture" -- which workload the counters were collected on (the
    workload key of the bench line printed under the profiler): bench.py attaches a traffic figure
    only to runs of that same workload."""
    stats = {}
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                           pct=float(r["Percentage"]))
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    summary = {}
    for k, st in stats.items():
        if k not in fetch:
            continue
        corr = 2.0 if k.startswith(WIDE_READERS) else 1.0
        rd, wr = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        summary[k] = dict(avg_us=round(st["avg_us"], 2), calls=st["calls"], pct=st["pct"],
                          FETCH_SIZE_KiB=round(fetch[k], 1), WRITE_SIZE_KiB=round(write.get(k, 0.0), 1),
                          fetch_correction=corr,
                          hbm_bytes_per_launch=int(rd * corr + wr),
                          hbm_GBps=round((rd * corr + wr) / (st["avg_us"] * 1e-6) / 1e9, 1))
    key = None
    try:
        with open(bench_json) as f:
            key = json.loads(f.read().strip().splitlines()[-1])["config"]["workload_key"]
    except (OSError, ValueError, KeyError, IndexError):
        pass
    summary["_capture"] = {"workload_key": key, "bench_line": os.path.basename(bench_json),
                           "counters": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join("gpurun_out", "prof_" + rnd)
    os.makedirs("profiles", exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"),
                os.path.join("profiles", rnd + "_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join("profiles", rnd + "_bench.json"))
    with open(os.path.join(src, "bench.err")) as f:
        table = [l for l in f.read().splitlines() if "launches=" in l]
    with open(os.path.join("profiles", rnd + "_bench_hip_event_table.txt"), "w") as f:
        f.write("\n".join(table) + "\n")
    summarize(os.path.join(src, "trace", "bench_kernel_stats.csv"),
              os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"),
              os.path.join(src, "pmc_write", "bench_counter_collection.csv"),
              os.path.join(src, "pmc_fetch.json"), os.path.join("profiles", rnd + "_pmc_hbm_summary.json"))
    # the other workloads' counters (uniform ids, the other BASELINE configurations)
    for tag in ("uniform", "c2", "c3", "c4"):
        st = os.path.join(src, tag + "_kernel_stats.csv")
        if os.path.exists(st) and os.path.exists(os.path.join(src, "pmc_fetch_" + tag, "bench_counter_collection.csv")):
            summarize(st, os.path.join(src, "pmc_fetch_" + tag, "bench_counter_collection.csv"),
                      os.path.join(src, "pmc_write_" + tag, "bench_counter_collection.csv"),
                      os.path.join(src, "pmc_fetch_" + tag + ".json"),
                      os.path.join("profiles", rnd + "_pmc_hbm_summary_" + tag + ".json"))
    # the other artefacts of tools/profile_round.sh, as they are
    for name in ("timeline.txt", "step_gaps.txt", "sq_summary.txt", "sq_summary_emu8.txt", "c2.json", "c3.json",
                 "c4.json", "c2_kernel_stats.csv", "c3_kernel_stats.csv", "c4_kernel_stats.csv", "uniform.json",
                 "fresh.json", "bench_driver_shape.json",
                 "emu8_strong_rank3.json") + tuple("emu8_rank%d.json" % r for r in range(8)):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join("profiles", rnd + "_" + name))
    with open(os.path.join("profiles", rnd + "_pmc_hbm_summary.json")) as f:
        summary = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"]):
        print("%-36s avg %9.1f us  hbm %8.1f MB/launch  %7.1f GB/s" % (
            k, v["avg_us"], v["hbm_bytes_per_launch"] / 1e6, v["hbm_GBps"]))


if __name__ == "__main__":
    main()
