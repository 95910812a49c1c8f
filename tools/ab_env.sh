#!/bin/bash
# A/B of run-time switches: tools/ab_env.sh BLOCKS "VAR=val ..." "VAR=val ..." ...   (one bench run per setting; "-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$1; shift
for S in "$@"; do
  if [ "$S" = "-" ]; then E=""; else E="$S"; fi
  env $E python3 $R/bench.py --blocks $B --no-pc --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
ip=d.get('interactive_path') or {}
print('%-40s device ms %.4f wall %.4f | fold launches %d avg us %.1f | interactive %.2f ms (rounds %.2f) | ok %s bit-exact %s' % ('$S', d['prover_sec_device']*1e3, d['ms_per_step'], r['launches'], r['avg_launch_us'], 1e3*(ip.get('prover_sec') or 0), 1e3*(ip.get('round_calls_sec') or 0), d['host_verifier_accepts'], d.get('bit_exact')))"
done
