#!/usr/bin/env python3
"""Condenses hipcc's -Rpass-analysis=kernel-resource-usage remarks into one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage engine.hip -o /dev/null 2> remarks.txt
       python tools/resource_usage.py remarks.txt [substring filter ...]"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def main():
    txt = open(sys.argv[1]).read()
    filt = sys.argv[2:]
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Dynamic Stack): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([r["name"] for r in rows])
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'sSpill':>6} {'vSpill':>6} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("ftrl_dev::", "").replace("void ", "")
        if n.startswith("rocprim::"):  # (the library sort's kernels: FM, LR, ranks)
            continue
        if filt and not any(f in n for f in filt):
            continue
        print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} {r.get('SGPRs Spill','?'):>6} {r.get('VGPRs Spill','?'):>6} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('LDS Size [bytes/block]','?'):>6}  {n}")


if __name__ == "__main__":
    main()
