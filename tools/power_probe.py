#!/usr/bin/env python3
"""Is k_leaf_hash's duration a function of the chip's power / clock state?  commit_private profiled launch by launch, back to back and with idle
gaps, for both transform paths; sclk / power sampled from sysfs in a side thread.   python tools/power_probe.py [BLOCKS]"""
import glob
import gzip
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vp_loader

vp = vp_loader.load()
vp.lib_host()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def sysfs_candidates():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out[p] = None
        for p in glob.glob(os.path.join(card, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(card, "hwmon/hwmon*/power1_input")) + \
                glob.glob(os.path.join(card, "hwmon/hwmon*/freq1_input")) + glob.glob(os.path.join(card, "hwmon/hwmon*/temp1_input")) + \
                glob.glob(os.path.join(card, "hwmon/hwmon*/power1_cap")):
            out[p] = None
    return list(out)


files = sysfs_candidates()
print("sysfs:", files, flush=True)
samples = []
stop = False


def sampler():
    while not stop:
        row = [time.perf_counter()]
        for f in files:
            try:
                txt = open(f).read()
                if "pp_dpm" in f:
                    cur = [l for l in txt.splitlines() if l.strip().endswith("*")]
                    row.append(cur[0].split(":")[1].strip().rstrip("*").strip() if cur else "?")
                else:
                    row.append(txt.strip())
            except OSError:
                row.append("-")
        samples.append(row)
        time.sleep(0.004)


with tempfile.TemporaryDirectory() as tmp:
    pws = os.path.join(tmp, "s.pws")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "SHA256_64.pws.gz"), "rb") as f, open(pws, "wb") as o:
        o.write(f.read())
    c = vp.Circuit.from_pws(pws, blocks, seed=1)
sess = {v: vp.Session(c, options=vp.Options(ntt_r8=v)) for v in (0, 1)}
for v in (0, 1):
    sess[v].commit_private()
th = threading.Thread(target=sampler, daemon=True)
th.start()


def run(v, gap, n=6):
    s = sess[v]
    rows = []
    for _ in range(n):
        if gap:
            time.sleep(gap)
        t0 = time.perf_counter()
        s.set_profiling(1)
        s.commit_private()
        st = s.launch_stats()
        s.set_profiling(0)
        t1 = time.perf_counter()
        near = [r for r in samples if t0 <= r[0] <= t1]
        leaf = [e["us"] / 1e3 for e in st if e["kernel"] == "k_leaf_hash"]
        ntt = sum(e["us"] for e in st if "ntt" in e["kernel"]) / 1e3
        rows.append((leaf[0], ntt, near[len(near) // 2][1:] if near else None, near[-1][1:] if near else None))
    print("ntt_r8=%d gap %.2fs:" % (v, gap))
    for r in rows:
        print("   leaf %.2f ms  ntt %.2f ms  sysfs(mid) %s  sysfs(end) %s" % r, flush=True)


for gap in (0.0, 0.3):
    for v in (0, 1):
        run(v, gap)
stop = True
