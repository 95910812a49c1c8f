#!/usr/bin/env python3
"""Issue cost of the generated Keccak round code by instruction class (tools/gen_keccak_asm.py): two middle rounds as the loop body.
   python3 tools/micro_keccak_parts.py > tools/_build/mkp.hip && hipcc --offload-arch=gfx950 -O3 -o tools/_build/mkp tools/_build/mkp.hip"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_keccak_asm as g
c2, _ = g.build(rounds=2, debug_state=True)
c4, _ = g.build(rounds=4, debug_state=True)
import re
FOLD = 56                     # registers folded into v[BASE, BASE + FOLD) (a multiple of 4: the bank of every operand is kept) -> 6 waves per SIMD fit
def fold(t):
    return re.sub(r"\bv(\d+)\b", lambda m: "v%d" % (g.BASE + (int(m.group(1)) - g.BASE) % FOLD), t)
body = [fold(i.text) for i in c4[len(c2):]]            # rounds 2 and 3: no constants left, both buffer directions
half = len(body) // 2
def blocked(b):
    """each round: theta as generated, then all of rho's v_alignbit_b32, then all of chi (timing only: the staging registers are shared by the rows)"""
    out = []
    for r in (b[:half], b[half:]):
        chi = [t for t in r if "0xd2" in t]
        first_chi = r.index(chi[0])
        k = first_chi
        while k > 0 and r[k - 1].startswith("v_alignbit"):
            k -= 1
        theta, rest = r[:k], r[k:]
        out += theta + [t for t in rest if t.startswith("v_alignbit")] + [t for t in rest if not t.startswith("v_alignbit")]
    return out
def sorted_all(b):
    return [t for t in b if t.startswith("v_xor")] + [t for t in b if t.startswith("v_bitop3")] + [t for t in b if t.startswith("v_alignbit")]
def spread(b):
    """alignbits spread evenly between the other instructions"""
    al = [t for t in b if t.startswith("v_alignbit")]; ot = [t for t in b if not t.startswith("v_alignbit")]
    out = []; ia = 0
    for k, t in enumerate(ot):
        out.append(t)
        while ia < len(al) and ia * len(ot) < (k + 1) * len(al):
            out.append(al[ia]); ia += 1
    return out + al[ia:]
def xor_as_bitop3(b):
    out = []
    for t in b:
        if t.startswith("v_xor_b32"):
            d, a, c = [x.strip() for x in t[len("v_xor_b32"):].split(",")]
            if a.startswith("0x"):
                a = c
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (d, a, c, c))
        else:
            out.append(t)
    return out
def force_pairs(b):
    """every three-source v_bitop3_b32 gets its second source moved into the bank of the first"""
    out = []
    for t in b:
        if t.startswith("v_bitop3"):
            args, imm = t[len("v_bitop3_b32"):].split(" bitop3:")
            d, a, bb, c = [x.strip() for x in args.split(",")]
            ra, rb = int(a[1:]), int(bb[1:])
            nb = g.BASE + ((rb - g.BASE) // 4 * 4 + (ra - g.BASE) % 4) % FOLD
            if nb == ra:
                nb = g.BASE + (nb - g.BASE + 4) % FOLD
            out.append("v_bitop3_b32 %s, %s, v%d, %s bitop3:%s" % (d, a, nb, c, imm))
        else:
            out.append(t)
    return out
def align_as_bitop3(b):
    out = []
    for t in b:
        if t.startswith("v_alignbit"):
            d, a, c, sh = [x.strip() for x in t[len("v_alignbit_b32"):].split(",")]
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (d, a, c, c))
        else:
            out.append(t)
    return out
def align_as_two_shifts_or(b):
    """a rotation half as three full-rate VOP2 instructions: shift, shift, or (timing only)"""
    out = []
    for t in b:
        if t.startswith("v_alignbit"):
            d, a, c, sh = [x.strip() for x in t[len("v_alignbit_b32"):].split(",")]
            out += ["v_lshlrev_b32 %s, %d, %s" % (d, 32 - int(sh), a), "v_lshrrev_b32 v%d, %s, %s" % (g.BASE + FOLD, sh, c), "v_or_b32 %s, %s, v%d" % (d, d, g.BASE + FOLD)]
        else:
            out.append(t)
    return out
def fraction(b, keep_every):
    """the stream without its v_alignbit_b32 except every keep_every-th (the others become v_bitop3_b32)"""
    out = []; k = 0
    for t in b:
        if t.startswith("v_alignbit"):
            k += 1
            if k % keep_every == 0:
                out.append(t); continue
            d, a, c, sh = [x.strip() for x in t[len("v_alignbit_b32"):].split(",")]
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (d, a, c, c))
        else:
            out.append(t)
    return out
def inject(b, fmt, every=8):
    """the all-fast stream (alignbits as bitop3) with every `every`-th former alignbit written as the candidate instruction"""
    out = []; k = 0
    for t in b:
        if t.startswith("v_alignbit"):
            k += 1
            d, a, c, sh = [x.strip() for x in t[len("v_alignbit_b32"):].split(",")]
            if k % every == 0:
                out.append(fmt.format(d=d, a=a, c=c, sh=sh, d2=d[1:], a2=a[1:])); continue
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (d, a, c, c))
        else:
            out.append(t)
    return out
cands = {
    "v_lshlrev_b32": "v_lshlrev_b32 {d}, 7, {a}", "v_lshrrev_b32": "v_lshrrev_b32 {d}, 7, {a}", "v_ashrrev_i32": "v_ashrrev_i32 {d}, 7, {a}",
    "v_or_b32": "v_or_b32 {d}, {a}, {c}", "v_and_b32": "v_and_b32 {d}, {a}, {c}", "v_add_u32": "v_add_u32 {d}, {a}, {c}", "v_sub_u32": "v_sub_u32 {d}, {a}, {c}",
    "v_mov_b32": "v_mov_b32 {d}, {a}", "v_not_b32": "v_not_b32 {d}, {a}", "v_bfrev_b32": "v_bfrev_b32 {d}, {a}",
    "v_lshl_add_u32": "v_lshl_add_u32 {d}, {a}, 3, {c}", "v_lshl_or_b32": "v_lshl_or_b32 {d}, {a}, 7, {c}", "v_and_or_b32": "v_and_or_b32 {d}, {a}, {c}, {c}",
    "v_add3_u32": "v_add3_u32 {d}, {a}, {c}, {c}", "v_xad_u32": "v_xad_u32 {d}, {a}, {c}, {c}", "v_bfe_u32": "v_bfe_u32 {d}, {a}, 7, 9", "v_bfi_b32": "v_bfi_b32 {d}, {a}, {c}, {c}",
    "v_perm_b32": "v_perm_b32 {d}, {a}, {c}, {c}", "v_alignbyte_b32": "v_alignbyte_b32 {d}, {a}, {c}, 1", "v_alignbit_b32 (vgpr shift)": "v_alignbit_b32 {d}, {a}, {c}, {c}",
    "v_mul_u32_u24": "v_mul_u32_u24 {d}, {a}, {c}", "v_mad_u32_u24": "v_mad_u32_u24 {d}, {a}, {c}, {c}", "v_mul_lo_u32": "v_mul_lo_u32 {d}, {a}, {c}", "v_mul_hi_u32": "v_mul_hi_u32 {d}, {a}, {c}",
    "v_cndmask_b32": "v_cndmask_b32 {d}, {a}, {c}, vcc", "v_min_u32": "v_min_u32 {d}, {a}, {c}", "v_max3_u32": "v_max3_u32 {d}, {a}, {c}, {c}",
    "v_add_f32": "v_add_f32 {d}, {a}, {c}", "v_fma_f32": "v_fma_f32 {d}, {a}, {c}, {c}", "v_pk_add_u16": "v_pk_add_u16 {d}, {a}, {c}", "v_pk_lshlrev_b16": "v_pk_lshlrev_b16 {d}, {a}, {c}",
    "v_pk_mul_lo_u16": "v_pk_mul_lo_u16 {d}, {a}, {c}", "v_cvt_f32_u32": "v_cvt_f32_u32 {d}, {a}", "v_ffbh_u32": "v_ffbh_u32 {d}, {a}", "v_bcnt_u32_b32": "v_bcnt_u32_b32 {d}, {a}, {c}",
    "v_mbcnt_lo_u32_b32": "v_mbcnt_lo_u32_b32 {d}, {a}, {c}", "v_sad_u32": "v_sad_u32 {d}, {a}, {c}, {c}", "v_lerp_u8": "v_lerp_u8 {d}, {a}, {c}, {c}", "v_mov_b32 dpp row_ror:1": "v_mov_b32_dpp {d}, {a} row_ror:1 row_mask:0xf bank_mask:0xf",
    "v_mov_b32 dpp quad_perm": "v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "s_nop 0": "s_nop 0", "v_nop": "v_nop",
}
injected = {"1/8 of rotations as %s" % k: inject(body, v) for k, v in cands.items()}
fracs = {"alignbit: 1 of %d kept (%.1f %% of the stream)" % (k, 100.0 * (116 // k) / 380): fraction(body, k) for k in (2, 4, 8, 16, 32, 116)}
variants = {"all, xor written as bitop3": xor_as_bitop3(body), "all, every bitop3 with a bank pair": force_pairs(body), "all, alignbit replaced by bitop3": align_as_bitop3(body),
            "all, alignbit as lshl + lshr + or": align_as_two_shifts_or(body),
"blocked per round (theta | rho | chi)": blocked(body), "sorted by class over both rounds": sorted_all(body), "alignbits spread evenly": spread(body)}
sel = {
    "all": lambda t: True,
    "bitop3 only": lambda t: t.startswith("v_bitop3"),
    "bitop3 0x96 only (theta sums)": lambda t: "0x96" in t,
    "bitop3 0xd2 only (chi)": lambda t: "0xd2" in t,
    "alignbit only": lambda t: t.startswith("v_alignbit"),
    "xor only": lambda t: t.startswith("v_xor"),
    "all but alignbit": lambda t: not t.startswith("v_alignbit"),
    "all but xor": lambda t: not t.startswith("v_xor"),
    "all but bitop3": lambda t: not t.startswith("v_bitop3"),
}
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>")
print("#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf(\"HIP error %s at %d\\n\", hipGetErrorString(e_), __LINE__); return 1; } } while (0)")
clob = ", ".join('"v%d"' % r for r in range(g.BASE, g.BASE + FOLD + 1))
names, counts = [], []
items = [(name, [t for t in body if f(t)]) for name, f in sel.items()] + list(variants.items()) + list(fracs.items()) + list(injected.items())
for k, (name, ins) in enumerate(items):
    names.append(name); counts.append(len(ins))
    print("__global__ void __launch_bounds__(256) k%d(uint32_t *out, int iters) {" % k)
    print("    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;")
    for r in range(g.BASE, g.BASE + FOLD):
        print("    asm volatile(\"v_mov_b32 v%d, %%0\" :: \"v\"(seed + %du) : \"v%d\");" % (r, r * 40503, r))
    print("    for (int it = 0; it < iters; ++it) asm volatile(\"%s\" ::: %s);" % ("\\n\\t".join(ins), clob))
    print("    uint32_t acc = 0, t;")
    for r in range(g.BASE, g.BASE + FOLD, 5):
        print("    asm volatile(\"v_mov_b32 %%0, v%d\" : \"=v\"(t) :: \"v%d\"); acc ^= t;" % (r, r))
    print("    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;\n}")
print("typedef void (*kfn)(uint32_t *, int);")
print("int main() {")
print("    kfn ks[] = {%s};" % ", ".join("k%d" % k for k in range(len(names))))
print("    const char *names[] = {%s};" % ", ".join('"%s"' % n for n in names))
print("    const int ninstr[] = {%s};" % ", ".join(str(c) for c in counts))
print(r"""    uint32_t *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("%-34s instr", "cycles / wave-instruction / SIMD @2.4 GHz"); for (int w = 2; w <= 6; w += 2) printf("   %d w/SIMD", w); printf("\n");
    for (unsigned k = 0; k < sizeof(ks) / sizeof(ks[0]); ++k) {
        printf("%-40s %5d", names[k], ninstr[k]);
        for (int w = 2; w <= 6; w += 2) {
            const int blocks = 256 * w;
            hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(256), 0, 0, out, 100); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("   %7.2f", ms * 1e-3 * 2.4e9 / ((double) iters * ninstr[k] * w));
        }
        printf("\n");
    }
    return 0;
}""")
