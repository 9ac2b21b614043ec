#!/usr/bin/env python3
"""GroupNorm-apply launches of the last replayed denoising step of a rocprofv3 kernel trace, with their shapes (tools/gn_sites.py) and rates.
    python tools/gn_trace.py gpurun_out/prof_r03_step/run_kernel_trace.csv gpurun_out/gn_sites.json"""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "ddim_step_kernel" in r["Kernel_Name"]]
w = rows[marks[-2] + 1:marks[-1] + 1]
sites = json.load(open(sys.argv[2]))
gn = [r for r in w if "gn_apply16c" in r["Kernel_Name"]]
if len(gn) != len(sites):
    # (GroupNorm passes that ride on a convolution call — stedm_conv_args.gn_* — are launched by the library or fused away: no shapes for those)
    print(f"{len(gn)} gn_apply16c launches in the trace, {len(sites)} recorded shapes: durations only")
    for r in gn:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // 256} x {r['Grid_Size_Y']}")
    print(f"total {sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in gn):.1f} us")
    sys.exit(0)
tot = totb = 0.0
for r, (B, HW, c1, c2) in zip(gn, sites):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by = B * HW * (c1 + c2) * 6
    tot += d; totb += by
    print(f"B={B:3d} HW={HW:4d} C={c1:4d}+{c2:4d} {by / 1e6:6.1f} MB {d:6.1f} us {by / d / 1e6:5.2f} TB/s  grid {int(r['Grid_Size_X']) // 256} x {r['Grid_Size_Y']}")
print(f"total {tot:.1f} us, {totb / 1e9:.2f} GB, {totb / tot / 1e6:.2f} TB/s")
