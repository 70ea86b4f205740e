#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from a rocprofv3 PMC pass

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python tools/mfma_busy.py DIR out.csv

busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of SIMD-cycles in which the
matrix pipe was executing (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles, 64 per v_mfma_f32_32x32x2_f32)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name[: name.index(">") + 1] if "<" in name.split("(")[0] else name.split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
rows = []
for name, c in acc.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    n = max(cnt[name], 1)
    busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n, c["GRBM_GUI_ACTIVE"] / n
    rows.append((name, n, busy, act, busy / (act / 8.0 * 1024.0)))
rows.sort(key=lambda r: -r[2])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "launches", "mfma_busy_cycles_per_launch", "grbm_gui_active_per_launch(sum of 8 XCDs)", "mfma_busy_fraction"])
for r in rows:
    w.writerow([r[0], r[1], f"{r[2]:.0f}", f"{r[3]:.0f}", f"{r[4]:.4f}"])
