#!/bin/bash
# The REAL reference (oracle/_ref/ref_run: /root/reference compiled in place, test infrastructure) at the HEADLINE size on the GPU box's host: SHA-256 x1024 with the
# commitment, one core — the CPU side of BASELINE configs[2] on the same box the device numbers come from (63 GB, ~10 minutes: too long for bench.py, whose cpu_baseline is
# the x256 sample).  tools/ref_x1024_on_box.sh [BLOCKS] -> gpurun_out/ref_cpu_x$B.txt (the reference's own report, wall / CPU seconds and peak RSS of the process, transcript and FRI record compared with
# tests/golden/).  A heartbeat line every 30 s keeps the call alive.
B=${1:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
zcat tests/golden/SHA256_64.pws.gz > /tmp/s.pws
O=gpurun_out/ref_cpu_x$B.txt
{ echo "# host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) CPUs visible, pinned to CPU $(( $(nproc) - 1 ))"; } > $O
( python3 - $B >> $O 2>&1 <<'PY'
import os, resource, subprocess, sys, time
b = sys.argv[1]
cpu = os.cpu_count() - 1
t0 = time.time()
r = subprocess.run(["taskset", "-c", str(cpu), "timeout", "-k", "10", "1080", "oracle/_ref/ref_run", "--pws", "/tmp/s.pws", "--blocks", b, "--pc", "1", "--dump", "/tmp/ref_t.bin", "--dump-fri", "/tmp/ref_f.bin"])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("rc=%d  process wall %.1f s  user %.1f s  sys %.1f s  max RSS %.1f GB" % (r.returncode, time.time() - t0, ru.ru_utime, ru.ru_stime, ru.ru_maxrss / 1e6), flush=True)
PY
) &
P=$!
T0=$(date +%s)
while kill -0 $P 2>/dev/null; do sleep 30; echo "$(( $(date +%s) - T0 )) s: reference running, rss_kb $(ps -o rss= -C ref_run | sort -n | tail -1)"; done
wait $P
cmp /tmp/ref_t.bin tests/golden/transcript_sha256_x$B.bin && echo "TRANSCRIPT_EQUAL (tests/golden/transcript_sha256_x$B.bin)" >> $O
cmp /tmp/ref_f.bin tests/golden/fri_sha256_x$B.bin && echo "FRI_EQUAL (tests/golden/fri_sha256_x$B.bin)" >> $O
tail -30 $O
