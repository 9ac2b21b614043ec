#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats --output-format csv` summary -> profiles/<tag>_kernel_stats.{csv,md}.

    python tools/stats_md.py gpurun_out/prof_r02_step r02_step "title" "command" [units_in_trace] [unit name]
"""
import csv, glob, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, tag, title, cmd = sys.argv[1:5]
    units = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    uname = sys.argv[6] if len(sys.argv) > 6 else "unit"
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md"), "w") as fh:
        fh.write(f"# {title}\n\nCommand: `{cmd}`\n\nSum of kernel time in the trace: {tot / 1e6:.2f} ms"
                 + (f" = {tot / 1e6 / units:.3f} ms per {uname} ({units:g} in the trace)" if units else "") + ".\n\n")
        fh.write("| kernel | calls | total ms | avg us | % |" + (f" us per {uname} |" if units else "") + "\n|---|---|---|---|---|" + ("---|" if units else "") + "\n")
        for r in rows:
            if float(r["Percentage"]) < 0.25:
                continue
            t = float(r["TotalDurationNs"])
            fh.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {t / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |"
                     + (f" {t / 1e3 / units:.0f} |" if units else "") + "\n")
    print(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md")).read())


if __name__ == "__main__":
    main()
