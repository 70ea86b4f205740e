#!/usr/bin/env python3
"""timeline of the LAST step of a rocprofv3 kernel trace: start offset, duration, end offset, queue/stream, kernel
   python tools/step_timeline.py results.db [first-kernel-substring]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{q} from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
first = sys.argv[2] if len(sys.argv) > 2 else "linear_fwd"
idx = [i for i, r in enumerate(rows) if first in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
for name, st, en, qq in rows[a - 3:b]:
    name = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name)); name = re.sub(r"_ZN12_GLOBAL__N_1\d+|_ZN4dggk\d+", "", name)
    print(f"{(st - t0) / 1e3:9.1f} +{(en - st) / 1e3:7.1f} = {(en - t0) / 1e3:9.1f}  q{qq}  {name[:60]}")
