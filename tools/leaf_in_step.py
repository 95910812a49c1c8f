#!/usr/bin/env python3
"""Where does k_leaf_hash lose time inside the protocol step?  (MI355X_MICROARCH.md, DVFS item 6.)

Needs the diagnostic build of the library (stamps in the leaf-hash kernels only, written to a buffer of their own):
    hipcc ... -DVP_LEAF_STAMPS -o tools/_build/stamps/libvpgpu.so virgo-plus_amd/csrc/vpgpu.hip
    VP_LIBGPU=tools/_build/stamps/libvpgpu.so python tools/leaf_in_step.py [BLOCKS]
Every workgroup of k_leaf_hash / k_leaf_hash_multi stamps {s_memtime, s_memrealtime} before and after its 65-step chains.  Per launch this prints the
launch's duration by the 100 MHz real-time counter (first workgroup in -> last workgroup out), the median duration of one workgroup in shader
cycles (work: it does not depend on the clock) and the median effective shader clock d(memtime) / d(memrealtime) x 100 MHz of its workgroups —
(a) inside the protocol step (commit_private -> GKR -> commit_public -> fft_gkr -> FRI), after >= 2 s of back-to-back steps;
(b) the same launch alone, back to back, after >= 2 s of itself;
(c) the step again with idle gaps in front of it (the clock a cold chip gives the first kernels).
If (a) and (b) differ in clock at equal cycles per workgroup, the step loses to the clock; if the cycles differ, to the memory side."""
import ctypes
import gzip
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader

vp = vp_loader.load()
vp.lib_host()
G = vp.lib_gpu()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if not hasattr(G, "vp_debug_leaf_stamps"):
    sys.exit("this library has no stamps: build with -DVP_LEAF_STAMPS and point VP_LIBGPU at it")
G.vp_debug_leaf_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)]
G.vp_debug_leaf_only.argtypes = [ctypes.c_void_p, ctypes.c_int]

with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
s = vp.Session(c)
s.draw_protocol_tape()
ctx = s.gpu_ctx()
buf = np.zeros((65536, 4), dtype=np.uint64)


def stamps():
    n = ctypes.c_uint(0)
    rc = G.vp_debug_leaf_stamps(ctx, buf.ctypes.data, 65536, ctypes.byref(n))
    assert rc == 0, rc
    a = buf[: n.value].copy()
    t0, r0, t1, r1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), (a[:, 3] >> np.uint64(1)).astype(np.int64)
    kind = (a[:, 3] & np.uint64(1)).astype(np.int64)
    order = np.argsort(r0)
    return t0[order], r0[order], t1[order], r1[order], kind[order]


def launches(st, min_wg=64):
    """Workgroups sorted by start.  Launches of one stream do not overlap: a workgroup that starts after every earlier one has ended (+ 0.5 us: a
    kernel boundary is >= 1.5 us) belongs to the next launch."""
    t0, r0, t1, r1, kind = st
    out, i, n = [], 0, len(r0)
    while i < n:
        j, end = i + 1, r1[i]
        while j < n and r0[j] <= end + 50 and kind[j] == kind[i]:
            end = max(end, r1[j]); j += 1
        if j - i >= min_wg:
            dt, dr = (t1[i:j] - t0[i:j]).astype(np.float64), (r1[i:j] - r0[i:j]).astype(np.float64)
            out.append({"kernel": "k_leaf_hash_multi" if kind[i] else "k_leaf_hash", "workgroups": j - i, "launch_ms": (r1[i:j].max() - r0[i]) / 1e5,
                        "wg_cycles_median": float(np.median(dt)), "wg_ms_median": float(np.median(dr)) / 1e5,
                        "clock_GHz_median": float(np.median(dt / dr)) * 0.1, "clock_GHz_p10": float(np.percentile(dt / dr, 10)) * 0.1,
                        "start_tick": int(r0[i])})
        i = j
    return out


def show(tag, ls):
    for k in ls:
        print("%-34s %-18s wgs %5d  launch %7.3f ms  workgroup: %8.0f kcycles, %6.3f ms, clock %.3f GHz (p10 %.3f)"
              % (tag, k["kernel"], k["workgroups"], k["launch_ms"], k["wg_cycles_median"] / 1e3, k["wg_ms_median"], k["clock_GHz_median"], k["clock_GHz_p10"]), flush=True)


def octiles(tag, st, which):
    """Clock along one launch: its workgroups in start order, eight groups."""
    t0, r0, t1, r1, kind = st
    ls = launches(st)
    for idx in which:
        if idx >= len(ls): continue
        k = ls[idx]
        sel = (r0 >= k["start_tick"]) & (kind == (1 if k["kernel"].endswith("multi") else 0))
        i0 = int(np.argmax(sel)); n = k["workgroups"]
        clk = (t1[i0:i0 + n] - t0[i0:i0 + n]) / (r1[i0:i0 + n] - r0[i0:i0 + n]).astype(np.float64) * 0.1
        print("   %-28s launch %d (%s, %.3f ms): clock by eighth of the launch  %s" % (tag, idx, k["kernel"], k["launch_ms"],
              "  ".join("%.3f" % float(np.median(c)) for c in np.array_split(clk, 8))), flush=True)


def summary(tag, ls):
    by = {}
    for k in ls:
        if k["workgroups"] >= 1024:
            by.setdefault(k["kernel"], []).append(k)
    for name, v in by.items():
        print("== %-30s %-18s n %3d  launch %.3f ms (min %.3f max %.3f)  workgroup %.0f kcycles  clock %.3f GHz" % (
            tag, name, len(v), np.mean([k["launch_ms"] for k in v]), min(k["launch_ms"] for k in v), max(k["launch_ms"] for k in v),
            np.mean([k["wg_cycles_median"] for k in v]) / 1e3, np.mean([k["clock_GHz_median"] for k in v])), flush=True)


# (a) in the step
t = time.time()
while time.time() - t < 2.5:
    s.prove_protocol()
stamps()
for _ in range(5):
    s.prove_protocol()
sa = stamps()
a = launches(sa)
show("(a) in step", a[:6])
octiles("(a) in step", sa, (0, 1, 2, 3, 4, 5))
summary("(a) in step", a)

# (b) alone, back to back
t = time.time()
while time.time() - t < 2.5:
    assert G.vp_debug_leaf_only(ctx, 10) == 0
stamps()
assert G.vp_debug_leaf_only(ctx, 15) == 0
sb = stamps()
b = launches(sb)
show("(b) alone, back to back", b[:3])
octiles("(b) alone", sb, (0, 7, 14))
summary("(b) alone, back to back", b)
# (b') alone, one launch per call (a host synchronisation after each) with the host idling in between: does a short idle gap cost the next launch its clock?
for gap in (0.0, 0.0005, 0.005, 0.05):
    for _ in range(12):
        assert G.vp_debug_leaf_only(ctx, 1) == 0
        if gap: time.sleep(gap)
    sg = stamps()
    summary("(b') alone, sync + %.1f ms idle" % (gap * 1e3), launches(sg)[2:])
    octiles("(b') %.1f ms idle" % (gap * 1e3), sg, (6,))

# (a') in the step again (order effects of the probe itself)
t = time.time()
while time.time() - t < 2.5:
    s.prove_protocol()
stamps()
for _ in range(5):
    s.prove_protocol()
a2 = launches(stamps())
summary("(a') in step, again", a2)

# (a2) the step with no idle device at all: every call queued without a host wait, the next pass's commit_private queued behind this pass's folds
t = time.time()
while time.time() - t < 2.5:
    s.prove_protocol(queue_next=True)
stamps()
for _ in range(5):
    s.prove_protocol(queue_next=True)
sp = stamps()
a3 = launches(sp)
show("(a2) in step, nothing idle", a3[:6])
octiles("(a2) nothing idle", sp, (0, 1, 2, 3, 4, 5))
summary("(a2) in step, nothing idle", a3)
s.prove_protocol()
# (a3) synchronous calls (round 4: a host wait after every call)
t = time.time()
while time.time() - t < 2.5:
    s.prove_protocol(deferred=False)
stamps()
for _ in range(5):
    s.prove_protocol(deferred=False)
sq = stamps()
show("(a3) in step, synchronous calls", launches(sq)[:3])
summary("(a3) in step, synchronous calls", launches(sq))

# (c) steps separated by idle gaps
for gap in (0.02, 0.3):
    stamps()
    for _ in range(4):
        time.sleep(gap)
        s.prove_protocol()
    summary("(c) step after %.2f s idle" % gap, launches(stamps()))
