#!/usr/bin/env python3
"""Development tool: where is the GPU idle inside the protocol step?  From a `rocprofv3 --kernel-trace` database of the headline bench:
    python tools/step_gaps.py RESULTS.db [min_gap_us]
takes the last complete step (from one commit_private's first transform to the next one's), merges the busy intervals of all queues and lists
every idle gap of at least min_gap_us with the kernels on either side; prints the step's span, busy time and idle time."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows = [(n.split("(")[0].replace("vp::", "").replace("void ", ""), s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
# a step starts at the inverse transform of commit_private: k_ntt8_cols<true> whose predecessor chain contains k_leaf_hash_multi (end of the previous step's FRI)
starts = []
seen_multi = True
for i, (n, s, e) in enumerate(rows):
    if "k_leaf_hash_multi" in n: seen_multi = True
    if n.startswith("k_ntt8_cols<true>") and seen_multi: starts.append(i); seen_multi = False
if len(starts) < 3: sys.exit("need three steps in the trace (found %d starts)" % len(starts))
for w in (3, 2):
    lo, hi = starts[-w], starts[-w + 1]
    seg = rows[lo:hi]
    t0 = seg[0][1]; latest = seg[0][2]; busy = 0; cur_s = seg[0][1]; gaps = []
    last_name = seg[0][0]
    for n, s, e in seg[1:]:
        if s > latest:
            busy += latest - cur_s; gaps.append(((s - latest) / 1e3, (latest - t0) / 1e3, last_name, n)); cur_s = s
        if e > latest: latest = e; last_name = n
    busy += latest - cur_s
    nxt = rows[hi][1]
    span = (nxt - t0) / 1e3
    print("step %d from the end: span to the next step's first kernel %.1f us, busy %.1f us, idle inside %.1f us in %d gaps, idle before the next step %.1f us"
          % (w - 1, span, busy / 1e3, sum(g[0] for g in gaps), len(gaps), (nxt - latest) / 1e3))
    big = [g for g in gaps if g[0] >= ming]
    print("  gaps >= %.0f us: %d, %.1f us in all" % (ming, len(big), sum(g[0] for g in big)))
    for g, at, a, b in big:
        print("    %8.1f us at %9.1f us   after %-28s before %s" % (g, at, a[:28], b[:40]))
