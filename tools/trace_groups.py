#!/usr/bin/env python3
"""Kernel dispatches of a rocprofv3 --kernel-trace run grouped by (kernel, grid size): count, average and total time — the shapes a kernel
family is launched on.    python tools/trace_groups.py <trace dir> [name regex] [last N dispatches]"""
import csv, glob, os, re, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if n: rows = rows[-n:]
    g = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if pat and not pat.search(r["Kernel_Name"]): continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        kind = re.search(r"Li\dELi(\d)ELb(\d)ELb(\d)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:40]) + (f"<kind {kind.group(1)}>" if kind else "")
        e = g[(name, int(r["Grid_Size_X"]) // max(1, int(r.get("Workgroup_Size_X", 1))))]
        e[0] += 1; e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in g.values())
    for (name, grid), (c, us) in sorted(g.items(), key=lambda kv: -kv[1][1]):
        print(f"{us / 1e3:9.3f} ms {100 * us / tot:5.1f} %  {c:5d} x {us / c:8.1f} us  blocks {grid:>7}  {name}")


if __name__ == "__main__":
    main()
