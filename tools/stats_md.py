#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats --output-format csv` summary -> profiles/<tag>_kernel_stats.{csv,md}.

    python tools/stats_md.py gpurun_out/prof_r03_step r03_step "title" "command" [units] [unit name] [marker kernel substring] [skip]

With a marker (a kernel launched exactly once per unit, e.g. `ddim_step_kernel` for a denoising step, `adamw_ema_kernel` for a training
step) the per-unit columns come from the dispatch trace, restricted to the window spanned by the LAST `units` occurrences of the marker:
whole replayed units only, so one-time work (weight packing, graph capture, warm-up) does not appear as "us per step". Without a marker
the per-unit columns divide the whole-process totals (setup included) and the file says so. `skip`: units of another kind at the END of the
trace that the window must not include (bench.py ends with 3 eager steps: one warm-up and the two event-bracketed ones of its roofline pass,
where the consumer GroupNorm of stedm_conv_args.gn_* runs as its own launch).
"""
import csv, glob, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def window_stats(trace_csv, marker, units, skip=0):
    rows = [r for r in csv.DictReader(open(trace_csv)) if r.get("Kind", "KERNEL_DISPATCH") == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    n = int(units)
    if skip:
        marks = marks[:-int(skip)]                       # the trace ends with `skip` units of another kind (bench.py: its eager, event-bracketed pass)
    if len(marks) < n + 1:
        raise SystemExit(f"stats_md: {len(marks)} launches of {marker!r} in the trace, need {n + 1} to delimit {n} whole units")
    lo, hi = marks[-(n + 1)] + 1, marks[-1] + 1          # dispatches after the (n+1)-th-from-last marker, up to and including the last one
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in rows[lo:hi]:
        tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[r["Kernel_Name"]] += 1
    span = int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
    return tot, cnt, span


def main():
    d, tag, title, cmd = sys.argv[1:5]
    units = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    uname = sys.argv[6] if len(sys.argv) > 6 else "unit"
    marker = sys.argv[7] if len(sys.argv) > 7 else None
    skip = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    win = None
    if marker and units:
        tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if tr:
            win = window_stats(tr[0], marker, units, skip)
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md"), "w") as fh:
        fh.write(f"# {title}\n\nCommand: `{cmd}`\n\nSum of kernel time in the whole trace (setup, warm-up and one-time packing included): {tot / 1e6:.2f} ms.\n\n")
        if win:
            wt, wc, span = win
            wsum = sum(wt.values())
            fh.write(f"## Per {uname}: the last {units:g} whole {uname}s of the trace (delimited by `{marker}`)\n\n"
                     f"Kernel time {wsum / 1e6 / units:.3f} ms per {uname}; wall span of the window {span / 1e6 / units:.3f} ms per {uname} "
                     f"(GPU idle between kernels {100 * (1 - wsum / span):.1f} %).\n\n")
            fh.write(f"| kernel | launches per {uname} | avg us | us per {uname} | % of the {uname} |\n|---|---|---|---|---|\n")
            for k in sorted(wt, key=lambda k: -wt[k]):
                if wt[k] / wsum < 0.0025:
                    continue
                fh.write(f"| `{k[:110]}` | {wc[k] / units:g} | {wt[k] / wc[k] / 1e3:.1f} | {wt[k] / 1e3 / units:.1f} | {100 * wt[k] / wsum:.2f} |\n")
            fh.write("\n## Whole process\n\n")
        elif units:
            fh.write(f"(no marker given: the per-{uname} column divides whole-process totals by {units:g}, one-time kernels included)\n\n")
        per = bool(units) and not win
        fh.write("| kernel | calls | total ms | avg us | % |" + (f" us per {uname} |" if per else "") + "\n|---|---|---|---|---|" + ("---|" if per else "") + "\n")
        for r in rows:
            if float(r["Percentage"]) < 0.25:
                continue
            t = float(r["TotalDurationNs"])
            fh.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {t / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |"
                     + (f" {t / 1e3 / units:.0f} |" if per else "") + "\n")
    print(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md")).read())


if __name__ == "__main__":
    main()
