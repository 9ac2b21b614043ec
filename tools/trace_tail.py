#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 --kernel-trace run, in launch order, with their durations (one forward of an encoder, one
step): python tools/trace_tail.py <trace dir> [N]"""
import csv, glob, os, re, sys


def main():
    d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = 0.0
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += us
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        kind = re.search(r"Li\dELi(\d)ELb(\d)ELb(\d)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:40]) + (f"<kind {kind.group(1)}{' fuse' if kind.group(2) == '1' else ''}{' p3' if kind.group(3) == '1' else ''}>" if kind else "")
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {us:9.1f} us  grid {r.get('Grid_Size_X', '?'):>9}  {name}")
    print(f"sum of kernel time {tot / 1e3:.3f} ms, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
