#!/usr/bin/env python3
"""Where the wavefront cycles of each kernel go, from one rocprofv3 PMC pass over the SQ block

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \\
              --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --cpu-rows -1 --no-hipgraph --no-variants
    python tools/sq_breakdown.py DIR out.csv

MI355X_MICROARCH.md ("rocprofv3 PMC slots"): SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue
stall) + SQ_ACTIVE_INST_ANY (issuing), all in quad-cycles summed over wavefronts.  Per kernel:
  valu_share   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of wavefront time spent issuing vector ALU instructions
  wait_share   = SQ_WAIT_ANY / SQ_WAVE_CYCLES              ... parked on memory / barriers
  stall_share  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES         ... stalled at issue
  valu_busy    = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)   share of SIMD-cycles with a VALU instruction issuing
A kernel whose valu_busy approaches 1 is bound by vector-instruction issue, whatever its memory traffic."""
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
    n = max(cnt[name], 1)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((gui / n, name, n, c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                 c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * gui / 8.0) if gui else 0.0))
rows.sort(reverse=True)
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["kernel", "launches", "gui_active_per_launch(sum of 8 XCDs)", "valu_share_of_wave_cycles", "wait_share", "issue_stall_share",
            "active_inst_share", "valu_busy_fraction_of_simd_cycles"])
for gui, name, n, vs, ws, ss, as_, vb in rows[:24]:
    w.writerow([name, n, round(gui), round(vs, 3), round(ws, 3), round(ss, 3), round(as_, 3), round(vb, 3)])
