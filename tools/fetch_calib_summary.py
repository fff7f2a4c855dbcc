#!/usr/bin/env python3
"""Joins tools/fetch_calib.sh's passes: for every access pattern the known byte count, the
bandwidth it reached, and FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) per launch
-> counter bytes / known bytes.  Output: <dir>/summary.json (copied to profiles/ by hand)."""
import csv
import glob
import json
import os
import sys


def counters(d, name):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != name:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                out.setdefault(k, []).append(float(r["Counter_Value"]))
    return out


def main():
    d = sys.argv[1]
    known, valu = {}, []
    with open(os.path.join(d, "plain.jsonl")) as f:
        for line in f:
            if not line.startswith("{"):
                continue
            j = json.loads(line)
            if "valu" in j:
                valu.append(j)
            else:
                known.setdefault(j["kernel"], []).append(j)
    fetch = counters(os.path.join(d, "fetch"), "FETCH_SIZE")
    write = counters(os.path.join(d, "write"), "WRITE_SIZE")
    hit = counters(os.path.join(d, "tcc"), "TCC_HIT_sum")
    miss = counters(os.path.join(d, "tcc"), "TCC_MISS_sum")
    # launches of one kernel name with different arguments (the 96 MB passes) are told apart by
    # their order: plain.jsonl lists them in launch order too
    order = []
    with open(os.path.join(d, "fetch.jsonl")) as f:
        for line in f:
            if line.startswith("{") and '"kernel"' in line:
                order.append(json.loads(line))
    seen = {}
    rows = []
    for j in order:
        base = j["kernel"].replace("_96MB_first", "").replace("_96MB_again", "")
        i = seen.get(base, 0)
        seen[base] = i + 1
        row = dict(kernel=j["kernel"], kind=j["kind"], known_bytes=j["known_bytes"])
        best = max(known.get(j["kernel"], [j]), key=lambda x: x["GBps"])
        row["GBps_unprofiled_best"] = best["GBps"]
        for nm, tab in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write), ("TCC_HIT", hit), ("TCC_MISS", miss)):
            v = tab.get(base, [])
            if i < len(v):
                row[nm] = v[i]
        if "FETCH_SIZE" in row:
            row["fetch_bytes_over_known"] = round(row["FETCH_SIZE"] * 1024.0 / j["known_bytes"], 4)
        if "WRITE_SIZE" in row:
            row["write_bytes_over_known"] = round(row["WRITE_SIZE"] * 1024.0 / j["known_bytes"], 4)
        rows.append(row)
    with open(os.path.join(d, "summary.json"), "w") as f:
        json.dump(dict(patterns=rows, valu=valu), f, indent=1)
    for r in rows:
        print("%-28s %-10s known %8.1f MB  %7.1f GB/s  fetch/known %s  write/known %s" % (
            r["kernel"], r["kind"], r["known_bytes"] / 1e6, r["GBps_unprofiled_best"],
            r.get("fetch_bytes_over_known"), r.get("write_bytes_over_known")))
    for v in valu:
        print("%-36s waves/SIMD %d  %.2f cycles per wave-instruction" % (v["valu"], v["waves_per_simd"], v["cycles_at_2.4GHz"]))


if __name__ == "__main__":
    main()
