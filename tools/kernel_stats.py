#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 `--kernel-trace` results
database (the .db that `rocprofv3 --kernel-trace --stats -d DIR -o NAME` leaves under DIR), written as the CSV that is
committed under profiles/.  Usage: python tools/kernel_stats.py results.db [out.csv] [--skip-first N]

--skip-first N drops the first N dispatches of every kernel (warm-up steps) before averaging."""
import csv
import re
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0
    for i, a in enumerate(sys.argv):
        if a == "--skip-first":
            skip = int(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    db = sqlite3.connect(args[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    per = {}
    for name, st, en in rows:
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\((anonymous namespace)\)::", "", name)
        name = re.sub(r"\(.*\)$", "", name)          # drop the argument list, keep template arguments
        per.setdefault(name, []).append((en - st) / 1e3)
    out = []
    tot_all = sum(sum(v[skip:]) for v in per.values())
    for name, v in per.items():
        v = v[skip:] if len(v) > skip else v
        out.append((name, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot_all))
    out.sort(key=lambda r: -r[2])
    w = csv.writer(open(args[1], "w", newline="") if len(args) > 1 else sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
    for r in out:
        w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}", f"{r[5]:.2f}", f"{r[6]:.2f}"])


if __name__ == "__main__":
    main()
