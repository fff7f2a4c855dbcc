#!/usr/bin/env python3
"""Instruction mix per kernel of a device-only assembly listing.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o engine.s ftrl-ffm_amd/csrc/engine.hip
       python tools/isa_mix.py engine.s [substring filter ...]"""
import collections
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    filt = sys.argv[2:]
    cur, stats = None, collections.OrderedDict()
    for l in txt:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            stats[cur] = collections.Counter()
            continue
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            cur = None
        f = l.split()
        if cur and f and l.startswith("\t") and not f[0].startswith((".", ";")):
            op, c = f[0], stats[cur]
            c["n"] += 1
            if op.startswith("v_pk_"):
                c["pk"] += 1
            elif op.startswith("v_"):
                c["v"] += 1
            elif op.startswith("s_"):
                c["s"] += 1
            elif op.startswith("ds_"):
                c["ds"] += 1
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                c["mem"] += 1
            if op.startswith("scratch_"):
                c["scr"] += 1
            if "readlane" in op or "writelane" in op:
                c["lane"] += 1
    names = subprocess.run(["c++filt"], input="\n".join(stats), capture_output=True, text=True).stdout.split("\n")
    print(f"{'total':>7} {'valu':>6} {'v_pk':>5} {'salu':>6} {'lds':>4} {'vmem':>4} {'scr':>4} {'lane':>4}  kernel")
    for (k, c), n in zip(stats.items(), names):
        n = re.sub(r"\(.*", "", n.replace("ftrl_dev::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        if filt and not any(x in n for x in filt):
            continue
        print(f"{c['n']:7d} {c['v']:6d} {c['pk']:5d} {c['s']:6d} {c['ds']:4d} {c['mem']:4d} {c['scr']:4d} {c['lane']:4d}  {n}")


if __name__ == "__main__":
    main()
