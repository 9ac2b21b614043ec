#!/usr/bin/env python3
"""profiles/<tag>_pmc_attention.md from the two counter passes of tools/profile_round.sh over tools/bench_lsa.py (bf16 and MX-fp8 at B = 64):
    python tools/pmc_attention.py gpurun_out/prof_r03_pmc_lsa gpurun_out/prof_r03_pmc_lsa_fp8 r03
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); effective clock = GRBM_GUI_ACTIVE / 8 / kernel time."""
import csv, glob, os, sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, H = 64, 4098, 12
FLOP = 4.0 * T * T * 64 * B * H


def rows(d, label):
    disp = OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    agg = OrderedDict()
    for e in disp.values():
        if not any(k in e["name"] for k in ("lsa_flash", "qkv_pack")):
            continue
        a = agg.setdefault(e["name"], {"n": 0, "us": 0.0, "busy": 0.0, "act": 0.0, "valu": 0.0})
        a["n"] += 1; a["us"] += e["us"]; a["busy"] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a["act"] += e.get("GRBM_GUI_ACTIVE", 0.0)
        a["valu"] += e.get("SQ_INSTS_VALU", 0.0)
    out = []
    for name, a in agg.items():
        us = a["us"] / a["n"]
        pf = FLOP / (us * 1e-6) / 1e15 if "lsa_flash" in name else 0.0
        busy = 100.0 * a["busy"] / (1024 * a["act"] / 8) if a["act"] else 0.0
        clk = a["act"] / 8 / (a["us"] * 1e3) if a["us"] else 0.0
        out.append(f"| `{name[:90]}` ({label}) | {a['n']} | {us:.1f} | {pf:.3f} | {busy:.1f} % | {clk:.2f} | {a['valu'] / a['n'] / 1e6:.0f} M |")
    return out


if __name__ == "__main__":
    d_bf, d_f8, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = [f"# {tag}: counters of the LSA flash attention kernels alone (`tools/bench_lsa.py <mode> 64`: B = {B}, T = {T}, {H} heads of 64; random operands)", "",
             "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU` (one pass per mode; `tools/pmc_attention.py`). MFMA busy = "
             "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); effective clock = GRBM_GUI_ACTIVE / 8 / kernel time.", "",
             "| kernel | launches | avg us | PFLOP/s (4 T^2 64 per head) | MFMA busy | eff. clock (GHz) | vector instructions per call |", "|---|---|---|---|---|---|---|"]
    lines += rows(d_bf, "bf16") + rows(d_f8, "MX-fp8")
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_attention.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-8:]))
