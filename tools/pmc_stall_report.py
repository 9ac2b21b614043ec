#!/usr/bin/env python3
"""Stall attribution per kernel from the SQ counter passes of tools/pmc_stall.sh / tools/pmc_lsa_stall.sh (each pass its own rocprofv3 run):

    python tools/pmc_stall_report.py gpurun_out/prof_r04_pmc_stall 'conv_rs|gn_apply' > profiles/r04_pmc_stall.md
        (prefix: <prefix>1, <prefix>2, <prefix>3 are the three pass directories; second argument: regex of kernel names to keep)

Reading (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES = wave-resident time (quad-cycles, summed over waves);
SQ_WAIT_ANY = parked at s_waitcnt / s_barrier; SQ_WAIT_INST_ANY = ready but not issued (pipe busy, MFMA read-after-write, arbitration);
SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to ~WAVE_CYCLES. SQ_VALU_MFMA_BUSY_CYCLES counts cycles (not quad-cycles)
over all SIMDs: MFMA-busy % = it / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import csv, glob, os, re, sys
from collections import OrderedDict, defaultdict


def dispatches(d):
    out = OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = out.setdefault((f, r["Dispatch_Id"]), {"name": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(out.values())


def per_kernel(disp):
    agg = defaultdict(lambda: defaultdict(float))
    for e in disp:
        a = agg[e["name"]]
        a["n"] += 1
        for k, v in e.items():
            if k != "name":
                a[k] += v
    return agg


def short(name):
    kinds = {"Li0ELb0": "conv_rs 3x3 (32x32x16)", "Li0ELb1": "conv_rs 3x3+skip (32x32x16)", "Li1ELb0": "conv_rs 1x1", "Li2ELb0": "conv_rs 2x2-tap (up / s2d)",
             "Li4ELb0": "conv_rs 3x3 (16x16x32)", "Li4ELb1": "conv_rs 3x3+skip (16x16x32)"}
    if "conv_rs_kernel" in name:
        for k, v in kinds.items():
            if k in name:
                return v + (" P3" if name.rstrip("E").endswith("Lb1E") and "Lb0ELb1" in name else "")
    m = re.search(r"(\w+_kernel)", name)
    return (m.group(1) if m else name)[:40]


def main():
    prefix = sys.argv[1]
    keep = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    p = [per_kernel(dispatches(prefix + str(i))) for i in (1, 2, 3)]
    names = [n for n in p[0] if keep is None or keep.search(n)]
    names.sort(key=lambda n: -p[0][n]["us"])
    print("| kernel | launches | avg us | MFMA busy | wave-cycles: parked (s_waitcnt / barrier) | ready, not issued | issuing | of issuing: VALU | LDS | VMEM | SALU | MISC | "
          "VALU / MFMA insts | LDS / MFMA | VMEM / MFMA | SALU / MFMA | LDS bank-conflict share |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n in names:
        a, b, c = p[0][n], p[1].get(n, {}), p[2].get(n, {})
        wc = a["SQ_WAVE_CYCLES"] or 1.0
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * a["GRBM_GUI_ACTIVE"] / 8.0) if a["GRBM_GUI_ACTIVE"] else 0.0
        act = a["SQ_ACTIVE_INST_ANY"] or 1.0
        mf = c.get("SQ_INSTS_MFMA", 0.0) or float("nan")
        g = lambda d, k: d.get(k, 0.0)
        # pass 2's counters are sums over the same launches (another run): shares are taken against pass 1's issuing cycles scaled by launch count
        sc = a["n"] / b["n"] if b and b.get("n") else 1.0
        print(f"| {short(n)} | {int(a['n'])} | {a['us'] / a['n']:.1f} | {100 * busy:.1f} % | {100 * a['SQ_WAIT_ANY'] / wc:.1f} % | {100 * a['SQ_WAIT_INST_ANY'] / wc:.1f} % | "
              f"{100 * a['SQ_ACTIVE_INST_ANY'] / wc:.1f} % | {100 * sc * g(b, 'SQ_ACTIVE_INST_VALU') / act:.0f} % | {100 * sc * g(b, 'SQ_ACTIVE_INST_LDS') / act:.0f} % | "
              f"{100 * sc * (g(b, 'SQ_ACTIVE_INST_VMEM') + g(b, 'SQ_ACTIVE_INST_FLAT')) / act:.0f} % | {100 * sc * g(b, 'SQ_ACTIVE_INST_SCA') / act:.0f} % | "
              f"{100 * sc * g(b, 'SQ_ACTIVE_INST_MISC') / act:.0f} % | {g(c, 'SQ_INSTS_VALU') / mf:.2f} | {g(c, 'SQ_INSTS_LDS') / mf:.2f} | {g(c, 'SQ_INSTS_VMEM') / mf:.2f} | "
              f"{g(c, 'SQ_INSTS_SALU') / mf:.2f} | {100 * g(c, 'SQ_LDS_BANK_CONFLICT') / (g(c, 'SQ_LDS_IDX_ACTIVE') or 1.0):.1f} % |")
    print()
    print("Passes: (1) SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; "
          "(2) SQ_ACTIVE_INST_{VALU,LDS,VMEM,SCA,MISC,FLAT} SQ_INST_CYCLES_VMEM SQ_WAVES; (3) SQ_INSTS_{VALU,MFMA,LDS,VMEM,SALU,SMEM} SQ_LDS_BANK_CONFLICT "
          "SQ_LDS_IDX_ACTIVE — three separate rocprofv3 --kernel-trace --pmc runs of the same command. SQ_INSTS_VALU includes the MFMA instructions "
          "on this part if the ratio reads >= 1 for a pure-MFMA loop (check against the ISA).")


if __name__ == "__main__":
    main()
