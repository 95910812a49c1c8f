#!/bin/bash
# A/B of compile-time switches on the commitment: tools/ab_pc.sh "<hipcc -D flags>" TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; F="$1"; T="$2"
cd "$R/virgo-plus_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $F -o libvpgpu.so vpgpu.hip 2> /dev/null || exit 1
cd "$R" && python3 bench.py --no-cpu-baseline --with-pc --steps 5 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['polynomial_commitment']
print('$T', {k: round(v,3) for k,v in pc.items() if 'device_ms' in k}, pc.get('full_transcript_bit_exact'), pc.get('fri_roots_bit_exact'))"
