#!/usr/bin/env python3
"""Fabric-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes): units are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream
and is doubled. Writes profiles/traffic.json (read by bench.py for roofline.traffic).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write
"""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
                cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def main():
    fetch, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
    conv = [k for k in fetch if "conv_rs_kernel" in k or "conv_dma" in k or "conv_igemm_kernel" in k]
    n = sum(fc[k] for k in conv)
    fb = sum(fetch[k] for k in conv) / n
    wb = sum(write[k] for k in conv) / max(1, sum(wc[k] for k in conv))
    out = {"conv_igemm_launches": n, "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb,
           "conv_igemm_hbm_bytes_per_launch": 2 * fb + wb,
           "per_kernel": {k[:80]: {"launches": fc[k], "fetch_x2_MB": round(2 * fetch[k] / fc[k] / 1e6, 2), "write_MB": round(write.get(k, 0.0) / max(1, wc.get(k, 0)) / 1e6, 2)}
                          for k in sorted(fetch, key=lambda k: -fetch[k])[:12]},
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate passes (bench.py --steps 3 --warmup 1); KiB units x1024; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream); Infinity-Cache hits are included in these fabric-side counters, "
                   "so this is an upper bound on HBM bytes"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
