#!/usr/bin/env python3
"""Measured cost of every sumcheck chain against the estimate the chain deal uses (vpgpu_batched.inc: chain_costs / assign_chains).
    python tools/chain_costs.py BLOCKS            (SHA-256 x BLOCKS)
With as many ranks as chains the longest-processing-time deal hands every rank ONE chain (zero-cost chains — layers without a phase 2 — stay on rank 0),
so prove_gkr() under set_shard(r, n_chains) is that chain alone on the device: its device time is the measured cost.  Printed: one row per chain
(kind, layer, estimate in units, measured ms, ms per million units), the fit, and what the deal at W = 2 / 4 / 8 looks like when the ranks' loads
are added up from the MEASURED costs instead of the estimates (the imbalance the estimate's error causes; a chain's latency floor overlaps with the
other chains of its rank, so the sums are upper bounds of a rank's time).  The assembled transcript is compared with the unsharded proof."""
import gzip
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lpt(costs, world):
    """assign_chains without the index split: chains by falling estimate to the least loaded rank."""
    load = [0.0] * world
    owner = [0] * len(costs)
    for c in sorted(range(len(costs)), key=lambda c: -costs[c]):
        if costs[c] == 0.0:
            continue
        b = min(range(world), key=lambda r: load[r])
        owner[c] = b
        load[b] += costs[c]
    return owner


def main():
    blocks = int(sys.argv[1])
    import numpy as np
    import vp_loader
    vp = vp_loader.load()
    vp.lib_host()
    with tempfile.TemporaryDirectory() as tmp:
        pws = os.path.join(tmp, "s.pws")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
            o.write(f.read())
        c = vp.Circuit.from_pws(pws, blocks, seed=1)
    s = vp.Session(c)
    s.draw_tape()
    tr, res = s.prove_gkr()
    whole_ms = min(s.prove_gkr()[1]["gkr_device_ms"] for _ in range(5))
    n_layers = c.layers
    n_lanes = 2 * (n_layers - 1)
    _, est = s.shard_chains()
    n_chains = len(est)
    world = n_chains

    def kind(ch):
        if ch == n_chains - 1:
            return "Vres", 0
        if ch < n_lanes:
            return ("phase 1" if ch % 2 == 0 else "Liu"), ch // 2 + 1
        return "phase 2", ch - n_lanes + 1

    measured = [0.0] * n_chains
    parts = []
    s.set_shard(0, world)
    owner, est = s.shard_chains()
    for r in range(world):
        s.set_shard(r, world)
        mine = [ch for ch in range(n_chains) if owner[ch] == r and est[ch] > 0.0]
        for _ in range(2):
            s.prove_gkr()
        ms = min(s.prove_gkr()[1]["gkr_device_ms"] for _ in range(4))
        parts.append(s.prove_gkr()[0])
        real = [ch for ch in mine if kind(ch)[0] != "Vres"]
        if len(real) == 1:
            measured[real[0]] = ms
        elif len(real) > 1:                       # does not happen with world = n_chains; kept honest
            for ch in real:
                measured[ch] = ms * est[ch] / sum(est[x] for x in real)
    s.set_shard(0, 1)
    same = vp.sum_transcripts(parts) == tr
    rows = []
    for ch in range(n_chains):
        k, layer = kind(ch)
        if est[ch] <= 1.0:
            continue
        rows.append({"chain": ch, "kind": k, "layer": layer, "estimate_units": float(est[ch]), "measured_ms": round(measured[ch], 4),
                     "us_per_million_units": round(1e3 * measured[ch] / (est[ch] / 1e6), 3)})
    big = [r for r in rows if r["estimate_units"] > 2.0e7]
    fit = sum(r["measured_ms"] for r in big) / sum(r["estimate_units"] for r in big) * 1e9 if big else None      # us per million units
    deals = {}
    for w in (2, 4, 8):
        own = lpt(list(est), w)
        by_est = [sum(est[ch] for ch in range(n_chains) if own[ch] == r and est[ch] > 1.0) for r in range(w)]
        by_meas = [sum(measured[ch] for ch in range(n_chains) if own[ch] == r) for r in range(w)]
        own_m = lpt(measured, w)
        ideal = [sum(measured[ch] for ch in range(n_chains) if own_m[ch] == r) for r in range(w)]
        deals["W%d" % w] = {"estimate_share_max_over_mean": round(max(by_est) / (sum(by_est) / w), 4),
                            "measured_ms_per_rank_of_the_estimate_deal": [round(x, 3) for x in by_meas],
                            "measured_max_over_mean": round(max(by_meas) / (sum(by_meas) / w), 4),
                            "measured_ms_max_if_dealt_by_measured_costs": round(max(ideal), 3)}
    out = {"blocks": blocks, "chains": n_chains, "unsharded_proof_device_ms": round(whole_ms, 4), "sum_of_solo_chain_ms": round(sum(measured), 3),
           "assembled_equals_unsharded": bool(same), "fit_us_per_million_units_chains_over_2e7": None if fit is None else round(fit, 3),
           "rows": rows, "deals_without_index_split": deals,
           "note": "solo chain times include each chain's latency floor (init -> folds -> k_seg -> k_emit), which overlaps with the other chains of a rank in a real deal"}
    print(json.dumps(out))
    s.close(); c.close()


if __name__ == "__main__":
    main()
