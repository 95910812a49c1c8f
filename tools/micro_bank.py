#!/usr/bin/env python3
"""Issue cost of the three Keccak instructions by register-bank pattern and waves per SIMD (explicit registers, inline asm).
   python3 tools/micro_bank.py gen > tools/_build/micro_bank.hip ; hipcc --offload-arch=gfx950 -O3 -o tools/_build/micro_bank tools/_build/micro_bank.hip"""
import sys
PAT = []
def pat(name, f): PAT.append((name, f))
# 64 instructions per block; sources v64..v95, destinations v96..v127
def gen(fmt, pick):
    out = []
    for i in range(64):
        d, a, b, c = pick(i)
        out.append(fmt.format(d=d, a=a, b=b, c=c))
    return out
pat("xor2 distinct banks",      lambda: gen("v_xor_b32 v{d}, v{a}, v{b}",                lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("xor2 same bank",           lambda: gen("v_xor_b32 v{d}, v{a}, v{b}",                lambda i: (96 + i % 32, 64 + (i * 4) % 32, 68 + (i * 4) % 28, 0)))
pat("bitop3 3 banks",           lambda: gen("v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("bitop3 2 in one bank",     lambda: gen("v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (96 + i % 32, 64 + (i * 4) % 32, 68 + (i * 4) % 28, 66 + (i * 4) % 32)))
pat("bitop3 3 in one bank",     lambda: gen("v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (96 + i % 32, 64 + (i * 4) % 24, 68 + (i * 4) % 24, 72 + (i * 4) % 24)))
pat("bitop3 dest = src bank",   lambda: gen("v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (96 + (i * 4) % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("bitop3 in place (d = a)",  lambda: gen("v_bitop3_b32 v{a}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (0, 96 + i % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("alignbit 2 banks, const",  lambda: gen("v_alignbit_b32 v{d}, v{a}, v{b}, 7",        lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("alignbit same bank, const",lambda: gen("v_alignbit_b32 v{d}, v{a}, v{b}, 7",        lambda i: (96 + i % 32, 64 + (i * 4) % 32, 68 + (i * 4) % 28, 0)))
pat("alignbit a = b (rotate)",  lambda: gen("v_alignbit_b32 v{d}, v{a}, v{a}, 7",        lambda i: (96 + i % 32, 64 + i % 32, 0, 0)))
pat("alignbit shift in sgpr",   lambda: gen("v_alignbit_b32 v{d}, v{a}, v{b}, s20",      lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("alignbit shift in vgpr",   lambda: gen("v_alignbit_b32 v{d}, v{a}, v{b}, v{c}",     lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("alignbyte 2 banks, const", lambda: gen("v_alignbyte_b32 v{d}, v{a}, v{b}, 1",       lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("perm_b32 (sel in sgpr)",   lambda: gen("v_perm_b32 v{d}, v{a}, v{b}, s20",          lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("lshlrev_b64 const",        lambda: gen("v_lshlrev_b64 v[{d}:{c}], 7, v[{a}:{b}]",   lambda i: (96 + 2 * (i % 16), 64 + 2 * (i % 16), 65 + 2 * (i % 16), 97 + 2 * (i % 16))))
pat("lshl_or_b32",              lambda: gen("v_lshl_or_b32 v{d}, v{a}, 7, v{b}",         lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 0)))
pat("and_or_b32",               lambda: gen("v_and_or_b32 v{d}, v{a}, v{b}, v{c}",       lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("xor3_b32? (v_xad_u32)",    lambda: gen("v_xad_u32 v{d}, v{a}, v{b}, v{c}",          lambda i: (96 + i % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("mix 2 bitop3 : 1 alignbit",lambda: [x for i in range(21) for x in (
        "v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % (96 + (3 * i) % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32),
        "v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % (96 + (3 * i + 1) % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32, 67 + (i * 4) % 32),
        "v_alignbit_b32 v%d, v%d, v%d, 7" % (96 + (3 * i + 2) % 32, 64 + (i * 4) % 32, 65 + (i * 4) % 32))] + ["s_nop 0"])
pat("dependent bitop3 chain",   lambda: gen("v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x96", lambda i: (96 + (i + 1) % 32, 96 + i % 32, 65 + (i * 4) % 32, 66 + (i * 4) % 32)))
pat("dependent alignbit chain", lambda: gen("v_alignbit_b32 v{d}, v{a}, v{b}, 7",        lambda i: (96 + (i + 1) % 32, 96 + i % 32, 65 + (i * 4) % 32, 0)))

def emit():
    clob = ", ".join('"v%d"' % r for r in range(64, 128))
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>")
    print("#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf(\"HIP error %s at %d\\n\", hipGetErrorString(e_), __LINE__); return 1; } } while (0)")
    for k, (name, f) in enumerate(PAT):
        body = "\\n\\t".join(f())
        print("__global__ void __launch_bounds__(256) k%d(uint32_t *out, int iters, unsigned long long *cyc) {" % k)
        print("    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;")
        print("    asm volatile(\"s_mov_b32 s20, 7\" ::: \"s20\");")
        for r in range(64, 128):
            print("    asm volatile(\"v_mov_b32 v%d, %%0\" :: \"v\"(seed + %du) : \"v%d\");" % (r, r * 40503, r))
        print("    unsigned long long t0 = __builtin_amdgcn_s_memtime();")
        print("    for (int it = 0; it < iters; ++it) asm volatile(\"%s\" ::: %s, \"s20\");" % (body, clob))
        print("    unsigned long long t1 = __builtin_amdgcn_s_memtime();")
        print("    uint32_t acc = 0, t;")
        for r in range(96, 128):
            print("    asm volatile(\"v_mov_b32 %%0, v%d\" : \"=v\"(t) :: \"v%d\"); acc ^= t;" % (r, r))
        print("    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;")
        print("    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;")
        print("}")
    print("typedef void (*kfn)(uint32_t *, int, unsigned long long *);")
    print("int main() {")
    print("    kfn ks[] = {%s};" % ", ".join("k%d" % k for k in range(len(PAT))))
    print("    const char *names[] = {%s};" % ", ".join('"%s"' % n for n, _ in PAT))
    print("    const int ninstr[] = {%s};" % ", ".join(str(len([x for x in f() if not x.startswith('s_')])) for _, f in PAT))
    print(r"""    uint32_t *out; unsigned long long *cyc; CK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    printf("%-30s", "cycles / wave-instruction / SIMD"); for (int w = 1; w <= 8; w *= 2) printf("   %d w/SIMD (memtime | events@2.4)", w); printf("\n");
    for (unsigned k = 0; k < sizeof(ks) / sizeof(ks[0]); ++k) {
        printf("%-30s", names[k]);
        for (int w = 1; w <= 8; w *= 2) {                      // w waves per SIMD: 256 CUs x (4 w) waves = 256 x w workgroups of 256 threads
            const int blocks = 256 * w;
            hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(256), 0, 0, out, 200, cyc); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(blocks); CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            double s = 0; for (auto v : h) s += (double) v; s /= blocks;
            const double per = (double) iters * ninstr[k] * w;
            printf("   %6.2f | %6.2f               ", s / per, ms * 1e-3 * 2.4e9 / per);
        }
        printf("\n");
    }
    return 0;
}""")
if __name__ == "__main__":
    emit()
