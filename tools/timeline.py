#!/usr/bin/env python3
"""Development tool: the kernel timeline of ONE replay of the proof plan from a `rocprofv3 --kernel-trace` database.
    python tools/timeline.py RESULTS.db            (the last proof in the trace: from the kernel after the previous k_emit_multi)
Prints start / end relative to the first kernel, the queue, and the idle time between the end of the latest-ending earlier kernel and
each start — the critical path's gaps."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute("select name, start, end, %s from kernels order by start" % q))
emits = [i for i, r in enumerate(rows) if "k_emit_multi" in r[0]]
if len(emits) < 3: sys.exit("need two proofs in the trace")
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1           # 1 = last proof, 2 = the one before, ...
lo, hi = emits[-w - 1] + 1, emits[-w]
seg = rows[lo:hi + 1]
t0 = seg[0][1]
print("%-28s %6s %9s %9s %8s %8s" % ("kernel", "queue", "start us", "end us", "dur us", "gap us"))
latest = t0
for n, s, e, qq in seg:
    name = n.split("(")[0].replace("vp::", "")
    print("%-28s %6s %9.1f %9.1f %8.1f %8.1f" % (name[:28], qq, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - latest) / 1e3))
    latest = max(latest, e)
print("proof span %.1f us, sum of kernel durations %.1f us" % ((max(r[2] for r in seg) - t0) / 1e3, sum(r[2] - r[1] for r in seg) / 1e3))
