#!/usr/bin/env python3
"""Does separating the half-rate instructions (v_alignbit_b32) from the full-rate ones (v_bitop3_b32, v_xor_b32) in TIME pay, when the waves of a SIMD
are kept in phase by s_barrier?  Loop body = K x [122 full-rate][58 alignbit], with / without a workgroup barrier at every switch; workgroups of
512 threads (two waves per SIMD), one per CU.   python3 tools/micro_phase.py > tools/_build/mph.hip"""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_keccak_asm as g
c2, _ = g.build(rounds=2, debug_state=True)
c4, _ = g.build(rounds=4, debug_state=True)
FOLD = 96
def fold(t):
    return re.sub(r"\bv(\d+)\b", lambda m: "v%d" % (g.BASE + (int(m.group(1)) - g.BASE) % FOLD), t)
body = [fold(i.text) for i in c4[len(c2):]]
fast = [t for t in body if not t.startswith("v_alignbit")][:244]
slow = [t for t in body if t.startswith("v_alignbit")][:116]
variants = []
for K in (1, 2, 4):           # phase length: K rounds' worth per phase
    for bar in (0, 1):
        ins = []
        reps = 4 // K
        for r in range(reps):
            ins += fast[:61 * K] + (["s_barrier"] if bar else []) + slow[:29 * K] + (["s_barrier"] if bar else [])
        variants.append(("phases of %3d full-rate / %3d alignbit, %s" % (61 * K, 29 * K, "barrier at every switch" if bar else "no barrier"), ins))
variants.append(("interleaved as generated (2 fast : 1 slow)", [x for i in range(116) for x in (fast[2 * i], fast[2 * i + 1], slow[i])]))
variants.append(("full-rate only", fast))
variants.append(("alignbit only", slow))
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>")
print("#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf(\"HIP error %s at %d\\n\", hipGetErrorString(e_), __LINE__); return 1; } } while (0)")
clob = ", ".join('"v%d"' % r for r in range(g.BASE, g.BASE + FOLD))
for k, (name, ins) in enumerate(variants):
    print("__global__ void __launch_bounds__(1024) k%d(uint32_t *out, int iters) {" % k)
    print("    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;")
    for r in range(g.BASE, g.BASE + FOLD):
        print("    asm volatile(\"v_mov_b32 v%d, %%0\" :: \"v\"(seed + %du) : \"v%d\");" % (r, r * 40503, r))
    print("    for (int it = 0; it < iters; ++it) asm volatile(\"%s\" ::: %s, \"memory\");" % ("\\n\\t".join(ins), clob))
    print("    uint32_t acc = 0, t;")
    for r in range(g.BASE, g.BASE + FOLD, 5):
        print("    asm volatile(\"v_mov_b32 %%0, v%d\" : \"=v\"(t) :: \"v%d\"); acc ^= t;" % (r, r))
    print("    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;\n}")
print("typedef void (*kfn)(uint32_t *, int);")
print("int main() {")
print("    kfn ks[] = {%s};" % ", ".join("k%d" % k for k in range(len(variants))))
print("    const char *names[] = {%s};" % ", ".join('"%s"' % n for n, _ in variants))
print("    const int ninstr[] = {%s};" % ", ".join(str(len([t for t in ins if not t.startswith('s_')])) for _, ins in variants))
print(r"""    uint32_t *out; CK(hipMalloc(&out, 256 * 1024 * 4 * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 3000;
    printf("%-72s instr", "cycles / wave-instruction / SIMD @2.4 GHz; one workgroup per CU of"); for (int t = 256; t <= 1024; t *= 2) printf("   %4d thr (%d w/SIMD)", t, t / 256); printf("\n");
    for (unsigned k = 0; k < sizeof(ks) / sizeof(ks[0]); ++k) {
        printf("%-72s %5d", names[k], ninstr[k]);
        for (int t = 256; t <= 1024; t *= 2) {
            hipLaunchKernelGGL(ks[k], dim3(256), dim3(t), 0, 0, out, 100); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(ks[k], dim3(256), dim3(t), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("   %18.2f", ms * 1e-3 * 2.4e9 / ((double) iters * ninstr[k] * (t / 256)));
        }
        printf("\n");
    }
    return 0;
}""")
