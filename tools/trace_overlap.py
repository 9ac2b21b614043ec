#!/usr/bin/env python3
"""How much of the kernels matching a regex ran beside other kernels, from a rocprofv3 --kernel-trace run:
python tools/trace_overlap.py <trace dir> <regex> [last N dispatches]"""
import csv, glob, os, re, sys


def main():
    d, pat = sys.argv[1], re.compile(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), bool(pat.search(r["Kernel_Name"])), r.get("Queue_Id", "?")) for r in rows]
    mine = [(s, e) for s, e, m, _ in iv if m]
    others = [(s, e) for s, e, m, _ in iv if not m]
    tot = sum(e - s for s, e in mine)
    ov = 0
    j = 0
    for s, e in mine:
        for s2, e2 in others:
            if e2 <= s: continue
            if s2 >= e: break
            ov += min(e, e2) - max(s, s2)
    span = iv[-1][1] - iv[0][0]
    busy = sum(e - s for s, e, _, _ in iv)
    print(f"{len(mine)} matching dispatches, {tot / 1e3:.1f} us of kernel time, {ov / 1e3:.1f} us of it beside another kernel ({100.0 * ov / max(tot, 1):.0f} %); "
          f"queues used: {sorted(set(q for _, _, _, q in iv))}; window span {span / 1e6:.3f} ms, sum of kernel time {busy / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
