#!/bin/bash
# A/B of compile-time switches of the device library on the GPU box:  tools/ab_build.sh "<hipcc -D flags>" TAG [bench args...]
# rebuilds virgo-plus_amd/csrc/libvpgpu.so with the extra flags and runs bench.py (x64 and x1024) against it.
R=${GRAFT_REPO_ROOT:-$(pwd)}; F="$1"; T="$2"; shift 2
cd "$R/virgo-plus_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $F -o libvpgpu.so vpgpu.hip 2> /dev/null || exit 1
cd "$R" && python3 bench.py --no-cpu-baseline --steps 30 > gpurun_out/ab_${T}_b64.json 2> /dev/null || exit 1
python3 bench.py --blocks 1024 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${T}_b1024.json 2> /dev/null || exit 1
python3 - "$T" <<'PY'
import json,sys
t=sys.argv[1]
for b in ("b64","b1024"):
    d=json.load(open("gpurun_out/ab_%s_%s.json"%(t,b))); r=d["roofline"]
    print(t,b,"device ms %.4f wall %.4f fold avg us %.1f GB/s %.0f ok %s %s"%(d["prover_sec_device"]*1e3,d["ms_per_step"],r["avg_launch_us"],r["achieved"],d["bit_exact_vs_reference_golden"],d["host_verifier_accepts"]))
PY
