#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of `bench.py` into profiles/<name>.json.

    python tools/pmc_summary.py OUT.json FETCH_DIR WRITE_DIR [SQ_DIR]

Each DIR holds the *_counter_collection.csv of one pass (FETCH_SIZE | WRITE_SIZE | SQ_* counters: separate passes, as
MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of a
wide coalesced read, so the corrected figure is 2x (same guide, HBM section).  Per kernel: launches, mean and max per launch.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d):
    """Per kernel, per counter: one value per dispatch.  Reads rocprofv3's CSV output (--output-format csv) or its default
    rocpd SQLite database (view counters_collection: one row per dispatch, counter and hardware instance -> summed)."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        import sqlite3
        db = sqlite3.connect(f)
        q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id order by dispatch_id"
        for name, counter, _, v in db.execute(q):
            out[name][counter].append(float(v))
    return out


def kernel_stats_csv(db_path, out_csv):
    """`rocprofv3 --kernel-trace --stats` summary from the rocpd database (view top_kernels), in the layout of *_kernel_stats.csv."""
    import sqlite3
    db = sqlite3.connect(db_path)
    raw = list(db.execute("select name, duration from kernels"))
    agg = collections.OrderedDict()
    for n, d in raw:                                   # library kernels (rocPRIM sorts of the circuit upload) under a short name, arguments dropped
        k = stats_name(n)
        a = agg.setdefault(k, [0, 0, None, 0])
        a[0] += 1; a[1] += d; a[2] = d if a[2] is None else min(a[2], d); a[3] = max(a[3], d)
    rows = sorted(((k, a[0], a[1], a[1] / a[0], a[2], a[3]) for k, a in agg.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for n, c, t, a, mn, mx in rows:
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n, c, t, a, 100.0 * t / tot, mn, mx))


def stats_name(n):
    if "rocprim" in n:
        m = re.search(r"wrapped_(\w+?)_config", n) or re.search(r"detail::(\w+?)(_kernel|<)", n)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return n.split("(")[0]


def short(n):
    m = re.search(r"(vp::)?k_\w+(<[\w, ]+>)?", n)          # template arguments kept: k_ntt8_rows<true> and <false> are different kernels
    return m.group(0) if m else n[:48]


def add_floor(summary_json, stats_csv):
    """pmc_summary.py --add-floor SUMMARY.json KERNEL_STATS_SINGLE_STREAM.csv: per kernel, the average launch duration of the SAME command's kernel-trace run
    with one stream (no overlap stretching the durations) and the VALU issue-floor fraction of that launch mix: SQ_INSTS_VALU per launch / 1024 SIMDs x 2 cycles
    (MI355X_MICROARCH.md) / 2.4 GHz over that duration — both sides from the same plan, whatever plan a later bench run's tuner picks."""
    d = json.load(open(summary_json))
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        dur[short(r["Name"]).replace("vp::", "")] = (int(r["Calls"]), float(r["AverageNs"]))
    for k in d["kernels"]:
        name = k["kernel"].replace("vp::", "")
        if name in dur and k.get("SQ_INSTS_VALU_per_launch"):
            calls, avg_ns = dur[name]
            k["single_stream_launches"] = calls
            k["single_stream_avg_launch_us"] = avg_ns / 1e3
            # totals over the run on both sides (the two profiled processes replay the same passes; where a switch of the tracing run changes how a proof's work is
            # cut into launches — VP_GKR_SERIAL=1 skips the tuner: 3 fused fold launches per proof instead of 4 — the totals still cover the same work)
            k["valu_issue_floor_frac"] = (k["launches"] * k["SQ_INSTS_VALU_per_launch"] / 1024.0 * 2.0 / 2.4e9) / (calls * avg_ns * 1e-9)
    json.dump(d, open(summary_json, "w"), indent=1)


def main():
    if sys.argv[1] == "--stats":          # pmc_summary.py --stats RESULTS.db OUT.csv
        kernel_stats_csv(sys.argv[2], sys.argv[3])
        return
    if sys.argv[1] == "--add-floor":
        add_floor(sys.argv[2], sys.argv[3])
        return
    argv = list(sys.argv)
    command = None
    if "--command" in argv:
        i = argv.index("--command"); command = argv[i + 1]; del argv[i:i + 2]
    out_path, fetch_d, write_d = argv[1:4]
    sq_d = argv[4] if len(argv) > 4 else None
    fe, wr = load(fetch_d), load(write_d)
    sq = load(sq_d) if sq_d else {}
    kernels = []
    for name in sorted(set(fe) | set(wr)):
        f = fe.get(name, {}).get("FETCH_SIZE", [])
        w = wr.get(name, {}).get("WRITE_SIZE", [])
        k = {"kernel": short(name), "launches": max(len(f), len(w))}
        if f:
            k["fetch_MB_per_launch_gfx950_corrected_x2"] = 2 * sum(f) / len(f) / 1024
            k["fetch_MB_largest_launch_corrected_x2"] = 2 * max(f) / 1024
        if w:
            k["write_MB_per_launch"] = sum(w) / len(w) / 1024
            k["write_MB_largest_launch"] = max(w) / 1024
        if f and w:
            k["hbm_bytes_per_launch"] = (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024
        for c, v in sq.get(name, {}).items():
            k[c + "_per_launch"] = sum(v) / len(v)
        kernels.append(k)
    summary = {"note": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of `python3 %s`; FETCH_SIZE corrected x2 for gfx950 (MI355X_MICROARCH.md, HBM section)"
                       % (command or "bench.py --steps 3 --warmup 2 --no-cpu-baseline"),
               "kernels": kernels}
    for k in kernels:
        if "sumfold3b_multi" in k["kernel"] and "hbm_bytes_per_launch" in k:
            summary["sumfold_avg_hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
    json.dump(summary, open(out_path, "w"), indent=1)
    print(json.dumps({k["kernel"]: round(k.get("hbm_bytes_per_launch", 0) / 1e6, 2) for k in kernels}, indent=1))


if __name__ == "__main__":
    main()
