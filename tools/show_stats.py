#!/usr/bin/env python3
"""print a kernel-stats table (tools/kernel_stats.py) with demangled, shortened names: python tools/show_stats.py results.db [skip]"""
import csv
import io
import subprocess
import sys

skip = sys.argv[2] if len(sys.argv) > 2 else "10"
out = subprocess.run([sys.executable, __file__.replace("show_stats", "kernel_stats"), sys.argv[1], "--skip-first", skip],
                     capture_output=True, text=True).stdout
rows = list(csv.reader(io.StringIO(out)))
names = subprocess.run(["c++filt"], input="\n".join(r[0].replace(".kd", "") for r in rows[1:]), capture_output=True, text=True).stdout.split("\n")
tot = 0.0
for r, n in zip(rows[1:], names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0] if "<" not in n.split("(")[0] else n[: n.index(">") + 1] if ">" in n else n
    per_step = float(r[2]) / max(int(rows[1][1]), 1)
    print(f"{n[:60]:60s} calls {r[1]:>5s} avg {float(r[3]):8.2f} us  {r[6]:>6s}%")
    tot += float(r[2])
print("total device time (us):", round(tot, 1))
