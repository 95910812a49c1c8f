#!/bin/bash
# The reference binary with the device prover at B blocks (default 1024 = BASELINE configs[2]) on the GPU box:
#   tools/seam_x1024.sh [BLOCKS]    -> gpurun_out/ref_blocks$B.txt (the reference's own report + the forwarding files' call counts), rss in gpurun_out/hb.txt
# oracle/_ref/ref_run_vpgpu_blocks = the unmodified reference (verifier, circuit code, lib/virgo) + INTEGRATION.md's forwarding files + libvpgpu.so;
# the messages handed over are compared with the CPU reference's golden transcript and FRI record of the same circuit.
B=${1:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
zcat tests/golden/SHA256_64.pws.gz > /tmp/s.pws
VPI_TRACE=1 VPI_DUMP=/tmp/m.bin VPI_DUMP_FRI=/tmp/f.bin timeout -k 10 1000 oracle/_ref/ref_run_vpgpu_blocks /tmp/s.pws $B > gpurun_out/ref_blocks$B.txt 2>&1 &
P=$!
while kill -0 $P 2>/dev/null; do
  sleep 20
  C=$(pgrep -P $P | head -1)
  echo "$(date +%T) rss_kb $(awk '/VmRSS/{print $2}' /proc/${C:-$P}/status 2>/dev/null)" >> gpurun_out/hb.txt
done
wait $P; echo "rc=$?" >> gpurun_out/ref_blocks$B.txt
cmp /tmp/m.bin tests/golden/transcript_sha256_x$B.bin && echo "TRANSCRIPT_EQUAL (forwarded messages == tests/golden/transcript_sha256_x$B.bin)" >> gpurun_out/ref_blocks$B.txt
cmp /tmp/f.bin tests/golden/fri_sha256_x$B.bin && echo "FRI_EQUAL (FRI challenges, roots, final codeword == tests/golden/fri_sha256_x$B.bin)" >> gpurun_out/ref_blocks$B.txt
tail -14 gpurun_out/ref_blocks$B.txt; tail -3 gpurun_out/hb.txt
