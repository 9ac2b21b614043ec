#!/usr/bin/env python3
"""Roofline evidence from rocprofv3 PMC passes of `bench.py` (each counter set collected in its OWN run with --kernel-trace only, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes):

    pass 1  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE          MFMA-busy % and effective clock per kernel
    pass 2  --pmc FETCH_SIZE                                        fabric-side read bytes  (KiB; gfx950 reports 1/2 of a wide coalesced stream: x2)
    pass 3  --pmc WRITE_SIZE                                        fabric-side write bytes (KiB)

    python tools/pmc_report.py gpurun_out/pmc_mfma gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 [tag]

Writes profiles/traffic.json (read by bench.py for roofline.traffic, only while the kernel sources' fingerprint still matches) and
profiles/<tag>_pmc_report.md. MFMA-busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the counter adds the busy
cycles of every SIMD (32 per v_mfma_f32_32x32x16, 16 per 16x16x32 — checked against the MFMA count of known shapes), GRBM_GUI_ACTIVE is
summed over the 8 XCDs. Effective clock = GRBM_GUI_ACTIVE / 8 / kernel time (reads high on dispatches well below 0.3 ms)."""
import csv, glob, json, os, sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dispatches(d):
    out = OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            e = out.setdefault((f, r["Dispatch_Id"]), {"name": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(out.values())


def per_kernel(disp, keys):
    agg = defaultdict(lambda: defaultdict(float))
    for e in disp:
        a = agg[e["name"]]
        a["n"] += 1; a["us"] += e["us"]
        for k in keys:
            a[k] += e.get(k, 0.0)
    return agg


def short(name):
    kinds = {"Li0ELb0": "conv_rs 3x3 (32x32x16)", "Li0ELb1": "conv_rs 3x3 + fused skip (32x32x16)", "Li1ELb0": "conv_rs 1x1", "Li2ELb0": "conv_rs 2x2-tap (sub-pixel up / s2d down)",
             "Li4ELb0": "conv_rs 3x3 (16x16x32)", "Li4ELb1": "conv_rs 3x3 + fused skip (16x16x32)"}
    if "conv_rs_kernel" in name:
        for k, v in kinds.items():
            if k in name:
                return v
    for key in ("gn_apply16c", "conv_splitk_reduce", "conv_out_kernel", "conv_in_fast", "ddim_step", "attn64_mfma", "space_to_depth16", "linear_rows", "lsa_flash",
                "wgrad3x3", "gn_bwd", "adamw_ema", "copyBuffer"):
        if key in name:
            return key
    return name[:48]


def main():
    dm, df, dw = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r02"
    import bench
    m = per_kernel(dispatches(dm), ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    f = per_kernel(dispatches(df), ["FETCH_SIZE"])
    w = per_kernel(dispatches(dw), ["WRITE_SIZE"])
    rows = []
    for name, a in sorted(m.items(), key=lambda kv: -kv[1]["us"]):
        if a["us"] < 0.002 * sum(x["us"] for x in m.values()):
            continue
        gui = a["GRBM_GUI_ACTIVE"]
        rows.append({"kernel": short(name), "launches": int(a["n"]), "avg_us": a["us"] / a["n"],
                     "mfma_busy_pct": 100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * gui) if gui else 0.0,
                     "eff_clock_ghz": gui / 8.0 / (a["us"] * 1e-6) / 1e9 if a["us"] else 0.0,
                     "fetch_x2_MB": 2 * f[name]["FETCH_SIZE"] * 1024 / max(1, f[name]["n"]) / 1e6 if name in f else None,
                     "write_MB": w[name]["WRITE_SIZE"] * 1024 / max(1, w[name]["n"]) / 1e6 if name in w else None,
                     "share_pct": 100.0 * a["us"] / sum(x["us"] for x in m.values())})
    conv = [n for n in f if "conv_rs_kernel" in n or "conv_dma" in n or "conv_igemm_kernel" in n or "conv_splitk_reduce" in n]
    # a "launch" of the roofline = one stedm_conv_igemm call (its split-K reduce pass included): count the main kernels only
    nmain = sum(f[n]["n"] for n in conv if "splitk_reduce" not in n)
    fb = sum(f[n]["FETCH_SIZE"] for n in conv) * 1024 / nmain
    wb = sum(w[n]["WRITE_SIZE"] for n in conv if n in w) * 1024 / nmain
    out = {"csrc_fingerprint": bench.csrc_fingerprint(), "conv_igemm_launches": int(nmain), "fetch_bytes_per_launch_raw": fb,
           "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb, "conv_igemm_hbm_bytes_per_launch": 2 * fb + wb,
           "per_kernel": rows,
           "note": "rocprofv3 --pmc passes of `bench.py --steps 5 --warmup 2` (MFMA busy + GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE: three separate "
                   "runs); KiB units x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream); the fabric-side "
                   "counters include Infinity-Cache hits: an upper bound on HBM bytes"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_report.md"), "w") as fh:
        fh.write(f"# {tag}: per-kernel PMC figures of the denoising bench (the bench's default headline mode, B=64 CFG step; kernel sources {out['csrc_fingerprint']})\n\n")
        fh.write("| kernel | launches | avg us | share of GPU time | MFMA busy | GRBM_GUI_ACTIVE / 8 / time (GHz; NOT the clock: reads high on dispatches under 0.3 ms - the in-loop clock is in the conv_loop_attribution profile) | fetch x2 (MB/launch) | write (MB/launch) |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            fmt = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"
            fh.write(f"| {r['kernel']} | {r['launches']} | {r['avg_us']:.1f} | {r['share_pct']:.1f} % | {r['mfma_busy_pct']:.1f} % | {r['eff_clock_ghz']:.2f} | "
                     f"{fmt(r['fetch_x2_MB'])} | {fmt(r['write_MB'])} |\n")
        fh.write(f"\nconv launches (stedm_conv_igemm calls): fabric-side bytes per launch {out['conv_igemm_hbm_bytes_per_launch'] / 1e6:.1f} MB "
                 f"(fetch x2 {2 * fb / 1e6:.1f} + write {wb / 1e6:.1f}).\n\n{out['note']}\n")
    print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_report.md")).read())


if __name__ == "__main__":
    main()
