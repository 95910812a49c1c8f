#!/bin/bash
# Builds tools/_build/micro_keccak (the leaf chain by the compiler's Keccak-f against the generator's variants) and tools/_build/micro_leaf (the product's two
# kernels at the x1024 geometry).  Run from the repo root; run the binaries through gpurun.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p tools/_build
{ python3 tools/gen_keccak_asm.py --rot1 fast 2> /dev/null | sed 's/vp_leaf_chain_asm/vp_leaf_chain_asm_rot1_alignbit/; s/#define VP_LEAF_ASM_THREADS 1024//; s/#pragma once//'
  python3 tools/gen_keccak_asm.py --no-barriers 2> /dev/null | sed 's/vp_leaf_chain_asm/vp_leaf_chain_asm_nobar/; s/#define VP_LEAF_ASM_THREADS 1024//; s/#pragma once//'
  python3 tools/gen_keccak_asm.py --barrier-at rho_after 2> /dev/null | sed 's/vp_leaf_chain_asm/vp_leaf_chain_asm_msgbefore/; s/#define VP_LEAF_ASM_THREADS 1024//; s/#pragma once//'
  python3 tools/gen_keccak_asm.py --barrier-at before 2> /dev/null | sed 's/vp_leaf_chain_asm/vp_leaf_chain_asm_add/; s/#define VP_LEAF_ASM_THREADS 1024//; s/#pragma once//'
  python3 tools/gen_keccak_asm.py --barrier-at both 2> /dev/null | sed 's/vp_leaf_chain_asm/vp_leaf_chain_asm_add_before/; s/#define VP_LEAF_ASM_THREADS 1024//; s/#pragma once//'; } > tools/_build/vp_keccak_asm_variants.h
cd tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../include -o _build/micro_keccak micro_keccak.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../include -o _build/micro_leaf micro_leaf.hip
