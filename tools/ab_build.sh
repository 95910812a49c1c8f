#!/bin/bash
# A/B of compile-time switches of the device library on the GPU box, without touching the product library:
#   tools/ab_build.sh "<hipcc -D flags>" TAG [bench args...]
# builds tools/_build/TAG/libvpgpu.so with the extra flags and alternates it with the product build in tools/ab_bench.sh
# (same call, same box: clocks differ by several per cent from box to box).
R=${GRAFT_REPO_ROOT:-$(pwd)}; F="$1"; T="$2"; shift 2
mkdir -p "$R/tools/_build/$T"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $F -o "$R/tools/_build/$T/libvpgpu.so" "$R/virgo-plus_amd/csrc/vpgpu.hip" 2> /dev/null || exit 1
cd "$R" && tools/ab_bench.sh "ab_$T" "$R/tools/_build/$T/libvpgpu.so" --no-cpu-baseline "$@"
