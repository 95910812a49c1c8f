#!/usr/bin/env python3
"""Generator of virgo-plus_amd/csrc/vp_keccak_asm.h: the leaf-hash chain of the Virgo commitment (fri.cpp:96-124 of the reference: per leaf 65 chained
SHA3-256 of 64-byte blocks, my_hhash.h:27-33) as ONE inline-asm block per kernel — loop, loads, Keccak-f[1600] and the digest store on hand-placed
registers — for workgroups of 1024 threads whose sixteen waves move through the permutation IN PHASE.

Why (measured, tools/micro_bank.py, tools/micro_keccak_parts.py, tools/micro_phase.py; profiles/r04_micro_*):
  * a gfx950 SIMD issues the plain logic instructions (v_bitop3_b32, v_xor_b32, v_mov_b32, v_add_u32, v_lshrrev_b32 ...) every ~2.4 cycles when two or
    more waves alternate, but v_alignbit_b32 — the only way to rotate — every ~4.3, AND while any wave of the SIMD has such an instruction in flight
    every other wave's instructions issue at that rate too: the compiler's Keccak (one v_alignbit_b32 in three, waves drifting) runs 3.7 cycles per
    instruction where the instruction counts alone would give 3.0.  With the waves of a SIMD kept in phase by s_barrier, so that all of them rotate
    at the same time and all of them do logic at the same time, the sum of the parts is what is measured (2.9-3.0);
  * a v_bitop3_b32 whose three DISTINCT sources have two registers in one bank (index mod 4) costs ~4.2 instead of ~2.5 in the logic phases: the state
    lanes and chi's staging registers are placed so that theta's sums have none and chi has the one pair per row that five lanes in four banks force;
  * the thirteen constant lanes of the padded block are folded at generation time and the last round keeps only what reaches the digest.
Round structure (default: --rot1 alignbit --barrier-at after):  [chi of the previous round + column sums C] | [rot(C, 1): 10 v_alignbit_b32] | s_barrier |
[D = C ^ rot C and A ^= D as v_xor_b32] | [rho/pi: 46 v_alignbit_b32 into the staging registers] | s_barrier | ...   A workgroup barrier at the END of every
rotation phase is what measures best: the waves must START a logic phase together; a wave that reaches its rotations early only slows the others' last
logic instructions.  Isolated, 2^20 leaves, same run (tools/micro_keccak.hip): this form 5.12-5.14 ms; rot(C, 1) from v_add_u32 / v_lshrrev_b32 /
v_bitop3_b32 inside the logic phase (--rot1 fast, one barrier per round) 5.26-5.29; barriers at both ends of the rotation phases 5.59-5.65; a barrier only
BEFORE the rotation phase 8.30; after rho only 5.81; every second round 6.22; none 8.06-8.26; the compiler's kernel 6.72.

    python3 tools/gen_keccak_asm.py [--rot1 alignbit|fast] [--no-barriers] > virgo-plus_amd/csrc/vp_keccak_asm.h
"""
import sys

BASE = 8                      # first fixed register; the block owns a subset of v[BASE, BASE + SPAN)
SPAN = 112
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001, 0x8000000080008081,
      0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a, 0x000000008000808b, 0x800000000000008b,
      0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800a, 0x800000008000000a, 0x8000000080008081,
      0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]      # ROT[x][y]
# bank (register index mod 4) of state lane (x, y): rows 0-2 of a column in three banks, rows 3, 4 in two (theta's sums read (0,1,2) then (t,3,4))
FA = [(0, 1, 2, 3, 0), (1, 2, 3, 0, 1), (2, 3, 0, 1, 2), (3, 0, 1, 2, 3), (0, 1, 2, 3, 2)]       # FA[x][y]
# bank of chi's staging register B[X][Y]: per row four banks, the fifth lane shares with the lane two places away (one pair per row is forced)
GB = [(0, 1, 2, 3, 1), (1, 2, 3, 0, 2), (2, 3, 0, 1, 3), (3, 0, 1, 2, 0), (0, 3, 2, 3, 1)]       # GB[Y][X]


class Pool:
    def __init__(self, base, n):
        self.free = {b: [r for r in range(base, base + n) if r % 4 == b] for b in range(4)}

    def take(self, bank):
        return self.free[bank].pop(0)

    def take_run(self, n, align):
        """n consecutive free registers starting at a multiple of `align`"""
        allf = sorted(sum(self.free.values(), []))
        for r in allf:
            if r % align == 0 and all((r + k) in allf for k in range(n)):
                for k in range(n):
                    self.free[(r + k) % 4].remove(r + k)
                return list(range(r, r + n))
        raise RuntimeError("no run")


class Val:
    def __init__(self, const=None, reg=None):
        self.const, self.reg = const, reg

    def is_const(self):
        return self.reg is None


class Ins:
    def __init__(self, text, dst=None, srcs=(), keep=False, cls="fast"):
        self.text, self.dst, self.srcs, self.keep, self.cls = text, dst, list(srcs), keep, cls


def build_body(rot1="alignbit", barriers=True, rounds=24, dce=True, msg_after_barrier=True, bar_mode="after", bar_every=1):
    """-> (instructions of ONE block of the chain, register map).  In: the message in M[0..8) (lanes 0-3), the previous digest in the state registers
    of lanes (0..3, 0); out: the digest in the same registers.  The marker instruction 'MSG_DEAD' sits where M is free for the next block's loads."""
    pool = Pool(BASE, SPAN)
    M = ["v%d" % r for r in pool.take_run(8, 4)]
    A = [[["v%d" % pool.take(FA[x][y]) for _ in range(2)] for y in range(5)] for x in range(5)]
    B = [[[None, None] for _ in range(5)] for _ in range(5)]                          # B[X][Y]
    for Y in range(5):
        for X in range(5):
            if (X, Y) == (0, 0):
                assert GB[0][0] == FA[0][0]
                continue                                                                # rot by 0: the state register itself
            B[X][Y] = ["v%d" % pool.take(GB[Y][X]) for _ in range(2)]
    # theta's temporaries live in staging registers (dead between chi and rho)
    stage = [B[X][Y][h] for Y in range(5) for X in range(5) if B[X][Y][0] for h in range(2)]
    by_bank = {b: [r for r in stage if int(r[1:]) % 4 == b] for b in range(4)}
    C = [[by_bank[(x + h) % 4].pop(0) for h in range(2)] for x in range(5)]
    D = [[by_bank[(x + h + 2) % 4].pop(0) for h in range(2)] for x in range(5)]
    R = [[by_bank[(x + h + 1) % 4].pop(0) for h in range(2)] for x in range(5)] if rot1 == "alignbit" else None
    T = {b: [by_bank[b].pop(0) for _ in range(2)] for b in range(4)}                   # an intermediate per bank: the one outside the banks of rows 3, 4 is used
    U = {b: [by_bank[b].pop(0) for _ in range(2)] for b in range(4)} if rot1 == "fast" else None
    code = []

    def emit(text, dst, srcs, cls="fast"):
        code.append(Ins(text, dst, [s for s in srcs if s is not None and s.startswith("v")], cls=cls))

    cur_round = [0]

    def barrier(kind="before", phase="rho"):           # the barrier stands before / after a rotation phase (rho's, or the one of the ten rotations by 1);
        on = bar_mode in ("both", kind) or ("%s_%s" % (phase, kind)) in bar_mode.split(",")      # bar_mode: both | before | after | a list like rho_after,rot1_after
        if barriers and on and cur_round[0] % bar_every == bar_every - 1:
            code.append(Ins("s_barrier", keep=True, cls="sync"))

    def xor_into(dst, vals, force=False):
        """dst <- xor of vals (constants folded); -> Val (a constant, an alias of a source unless force, or dst)"""
        k = 0
        regs = []
        for v in vals:
            if v.is_const():
                k ^= v.const
            elif v.reg in regs:
                regs.remove(v.reg)
            else:
                regs.append(v.reg)
        if not regs:
            if force:
                emit("v_mov_b32 %s, 0x%x" % (dst, k), dst, [])
                return Val(reg=dst)
            return Val(const=k)
        if len(regs) == 1 and k == 0:
            if force and regs[0] != dst:
                emit("v_mov_b32 %s, %s" % (dst, regs[0]), dst, [regs[0]])
                return Val(reg=dst)
            return Val(reg=regs[0])
        while len(regs) >= 3:
            a, b, c = regs[:3]
            emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (dst, a, b, c), dst, [a, b, c])
            regs = [dst] + regs[3:]
        if len(regs) == 2:
            emit("v_xor_b32 %s, %s, %s" % (dst, regs[0], regs[1]), dst, regs)
            regs = [dst]
        if k:
            emit("v_xor_b32 %s, 0x%x, %s" % (dst, k, regs[0]), dst, [regs[0]])
            regs = [dst]
        return Val(reg=regs[0])

    def rot64(lo, hi, n, dlo, dhi):
        if n == 0:
            return lo, hi
        assert not lo.is_const() and not hi.is_const() and n != 32
        if n < 32:
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dhi, hi.reg, lo.reg, 32 - n), dhi, [hi.reg, lo.reg], "slow")
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dlo, lo.reg, hi.reg, 32 - n), dlo, [lo.reg, hi.reg], "slow")
        else:
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dhi, lo.reg, hi.reg, 64 - n), dhi, [lo.reg, hi.reg], "slow")
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dlo, hi.reg, lo.reg, 64 - n), dlo, [hi.reg, lo.reg], "slow")
        return Val(reg=dlo), Val(reg=dhi)

    # round 0 input
    S = [[[Val(const=0), Val(const=0)] for _ in range(5)] for _ in range(5)]           # S[x][y][h]
    for i in range(4):
        S[i % 5][i // 5] = [Val(reg=M[2 * i]), Val(reg=M[2 * i + 1])]
    for i in range(4, 8):                                                                 # the previous digest: still in the registers of lanes (i - 4, 0)
        S[i % 5][i // 5] = [Val(reg=A[i - 4][0][0]), Val(reg=A[i - 4][0][1])]
    S[8 % 5][8 // 5] = [Val(const=0x06), Val(const=0)]
    S[16 % 5][16 // 5] = [Val(const=0), Val(const=0x80000000)]
    for rnd in range(rounds):
        cur_round[0] = rnd
        # ---- logic: column sums
        Cv = []
        for x in range(5):
            pair = []
            for h in range(2):
                col = [S[x][y][h] for y in range(5)]
                if sum(1 for v in col if not v.is_const()) <= 3:
                    pair.append(xor_into(C[x][h], col, force=True))
                else:
                    banks34 = {int(v.reg[1:]) % 4 for v in (col[3], col[4]) if not v.is_const()}
                    tb = next(b for b in (1, 2, 3, 0) if b not in banks34)
                    t = xor_into(T[tb][h], col[:3])
                    pair.append(xor_into(C[x][h], [t, col[3], col[4]], force=True))
            Cv.append(pair)
        # ---- rot(C, 1) and D
        Dv = []
        if rot1 == "alignbit":
            barrier("before", "rot1")
            Rv = [rot64(Cv[x][0], Cv[x][1], 1, R[x][0], R[x][1]) for x in range(5)]
            barrier("after", "rot1")
            for x in range(5):
                Dv.append([xor_into(D[x][h], [Cv[(x + 4) % 5][h], Rv[(x + 1) % 5][h]], force=True) for h in range(2)])
        else:
            for x in range(5):
                lo, hi = Cv[(x + 1) % 5]
                pair = []
                for h in range(2):
                    a, b = (lo, hi) if h == 0 else (hi, lo)                               # out half h = (a << 1) | (b >> 31)
                    prev = Cv[(x + 4) % 5][h].reg
                    pb = int(prev[1:]) % 4
                    tb, ub = [bk for bk in range(4) if bk != pb][:2]
                    t, u = T[tb][h], U[ub][h]
                    emit("v_add_u32 %s, %s, %s" % (t, a.reg, a.reg), t, [a.reg])
                    emit("v_lshrrev_b32 %s, 31, %s" % (u, b.reg), u, [b.reg])
                    emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0x1e" % (D[x][h], prev, t, u), D[x][h], [prev, t, u])       # prev ^ (t | u)
                    pair.append(Val(reg=D[x][h]))
                Dv.append(pair)
        # ---- A ^= D.  Round 0: the lanes that hold the previous digest first (their sources are the registers lanes 0-3 are about to take)
        order = [(x, y) for y in range(5) for x in range(5)]
        if rnd == 0:
            order = [(i % 5, i // 5) for i in (4, 5, 6, 7, 0, 1, 2, 3)] + [(i % 5, i // 5) for i in range(8, 25)]
        for (x, y) in order:
            for h in range(2):
                S[x][y][h] = xor_into(A[x][y][h], [S[x][y][h], Dv[x][h]], force=True)
        # ---- rho + pi: every lane into its staging register
        if rnd == 0 and not msg_after_barrier:
            code.append(Ins("MSG_DEAD", keep=True, cls="sync"))
        barrier("before")
        if rnd == 0 and msg_after_barrier: # the message registers are free from here on; the address arithmetic and the loads of the next block sit at the
            code.append(Ins("MSG_DEAD", keep=True, cls="sync"))      # start of a rotation phase (whatever their issue class, they do not slow a logic phase)
        Bv = [[None] * 5 for _ in range(5)]
        for Y in range(5):
            for X in range(5):
                y = X
                x = next(xx for xx in range(5) if (2 * xx + 3 * y) % 5 == Y)
                d = B[X][Y] if B[X][Y][0] else [None, None]
                Bv[X][Y] = rot64(S[x][y][0], S[x][y][1], ROT[x][y], d[0], d[1])
        barrier("after")
        # ---- chi (+ iota), in place in the state registers.  B[0][0] IS the state register of lane (0,0) (rotation by 0): in row 0 the lane X = 0
        # is written last, after the lanes X = 3, 4 that read it
        for Y in range(5):
            for X in ((1, 2, 3, 4, 0) if Y == 0 else range(5)):
                for h in range(2):
                    a, b, c = Bv[X][Y][h].reg, Bv[(X + 1) % 5][Y][h].reg, Bv[(X + 2) % 5][Y][h].reg
                    d = A[X][Y][h]
                    emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd2" % (d, a, b, c), d, [a, b, c])      # a ^ (~b & c)
                    S[X][Y][h] = Val(reg=d)
        for h in range(2):
            k = (RC[rnd] >> (32 * h)) & 0xffffffff
            if k:
                d = A[0][0][h]
                emit("v_xor_b32 %s, 0x%x, %s" % (d, k, d), d, [d])
    digest = [A[i][0][h] for i in range(4) for h in range(2)]
    keep = code
    if dce:                               # dead-code elimination, backwards (the last round keeps only what reaches the digest)
        live = set(digest)
        keep = []
        for ins in reversed(code):
            if ins.keep:
                keep.append(ins)
            elif ins.dst in live:
                live.discard(ins.dst)
                live.update(ins.srcs)
                keep.append(ins)
        keep.reverse()
    AD = [pool.take_run(2, 2), pool.take_run(2, 2)]      # the two load addresses of the thread (pairs, even-aligned), among the registers nothing else took
    used = set(int(r[1:]) for r in M) | set(AD[0]) | set(AD[1])
    for ins in keep:
        for r in [ins.dst] + ins.srcs:
            if r and r.startswith("v"):
                used.add(int(r[1:]))
    return keep, {"M": M, "A": A, "digest": digest, "used": sorted(used), "AD": AD}


def stats(code):
    n = {"bitop3": 0, "alignbit": 0, "xor": 0, "mov": 0, "add": 0, "lshr": 0, "barrier": 0}
    pairs = 0
    for i in code:
        for k, p in (("bitop3", "v_bitop3"), ("alignbit", "v_alignbit"), ("xor", "v_xor"), ("mov", "v_mov"), ("add", "v_add"), ("lshr", "v_lshrrev"), ("barrier", "s_barrier")):
            if i.text.startswith(p):
                n[k] += 1
        if i.text.startswith("v_bitop3"):
            banks = [int(r[1:]) % 4 for r in set(i.srcs)]
            if len(set(banks)) < len(banks):
                pairs += 1
    n["bitop3_bank_pairs"] = pairs
    return n


def emit_header(out, rot1, barriers, msg_after=True, addr="mad", bar_mode="after", bar_every=1):
    code, regs = build_body(rot1, barriers, msg_after_barrier=msg_after, bar_mode=bar_mode, bar_every=bar_every)
    n = stats(code)
    M, dig = regs["M"], regs["digest"]
    mlo, mhi = int(M[0][1:]), int(M[7][1:])
    (a0, a0h), (a1, a1h) = regs["AD"]
    L = []                                       # asm lines
    L.append("v_mov_b64 v[%d:%d], %%[addr0]" % (a0, a0h))
    L.append("v_mov_b64 v[%d:%d], %%[addr1]" % (a1, a1h))
    L.append("s_mov_b32 s45, 0")
    L.append("s_add_u32 s44, %[count], 1")       # blocks of the chain: the slices' pairs and the mask slice's (all zero, src/prover.cpp:526)
    L.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off" % (mlo, mlo + 3, a0, a0h))
    L.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off" % (mlo + 4, mhi, a1, a1h))
    for r in dig:
        L.append("v_mov_b32 %s, 0" % r)
    L.append("1:")
    L.append("s_waitcnt vmcnt(0)")
    for ins in code:
        if ins.text == "MSG_DEAD":
            # the message registers are free: the next block's pair on its way (or zeros for the mask slice's block)
            L.append("s_cmp_le_u32 s44, 2")
            L.append("s_cbranch_scc1 2f")
            if addr == "add":
                L.append("v_add_co_u32 v%d, vcc, %%[stride], v%d" % (a0, a0))
                L.append("v_addc_co_u32 v%d, vcc, 0, v%d, vcc" % (a0h, a0h))
                L.append("v_add_co_u32 v%d, vcc, %%[stride], v%d" % (a1, a1))
                L.append("v_addc_co_u32 v%d, vcc, 0, v%d, vcc" % (a1h, a1h))
            else:                                    # address of block k = address of block 0 + k * stride: one multiply-add per address
                L.append("s_add_u32 s45, s45, 1")
                L.append("v_mad_u64_u32 v[%d:%d], vcc, s45, %%[stride], %%[addr0]" % (a0, a0h))
                L.append("v_mad_u64_u32 v[%d:%d], vcc, s45, %%[stride], %%[addr1]" % (a1, a1h))
            L.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off" % (mlo, mlo + 3, a0, a0h))
            L.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off" % (mlo + 4, mhi, a1, a1h))
            L.append("s_branch 3f")
            L.append("2:")
            for r in M:
                L.append("v_mov_b32 %s, 0" % r)
            L.append("3:")
        else:
            L.append(ins.text)
    L.append("s_sub_u32 s44, s44, 1")
    L.append("s_cmp_lg_u32 s44, 0")
    L.append("s_cbranch_scc1 1b")
    # digest -> eight consecutive registers -> two 16-byte stores, lanes past the end masked off
    for k, r in enumerate(dig):
        L.append("v_mov_b32 v%d, %s" % (mlo + k, r))
    L.append("v_cmp_ne_u32 vcc, 0, %[active]")
    L.append("s_and_saveexec_b64 s[46:47], vcc")
    L.append("global_store_dwordx4 %%[out], v[%d:%d], off" % (mlo, mlo + 3))
    L.append("global_store_dwordx4 %%[out], v[%d:%d], off offset:16" % (mlo + 4, mhi))
    L.append("s_mov_b64 exec, s[46:47]")
    flags = ("" if rot1 == "alignbit" else "--rot1 " + rot1) + ("" if barriers else " --no-barriers") + ("" if msg_after else " --msg-before-barrier") + ("" if addr == "mad" else " --addr add") + ("" if bar_mode == "after" else " --barrier-at " + bar_mode) + ("" if bar_every == 1 else " --barrier-every %d" % bar_every)
    out.write("// GENERATED by tools/gen_keccak_asm.py%s — do not edit.\n" % ((" " + flags.strip()) if flags.strip() else ""))
    out.write("// The leaf-hash chain (fri.cpp:96-124: SHA3-256 over (pair of slice s || previous digest), s = 0 .. count-1, then the all-zero pair of the mask slice) for\n")
    out.write("// workgroups of up to VP_LEAF_ASM_THREADS threads, ONE per CU, waves in phase.  Per block of the chain: %d v_bitop3_b32 (%d with two sources in one bank),\n"
              % (n["bitop3"], n["bitop3_bank_pairs"]))
    out.write("// %d v_alignbit_b32, %d v_xor_b32, %d v_mov_b32, %d v_add_u32 + %d v_lshrrev_b32, %d s_barrier.  Fixed registers: %d inside v[%d, %d).\n"
              % (n["alignbit"], n["xor"], n["mov"], n["add"], n["lshr"], n["barrier"], len(regs["used"]), BASE, BASE + SPAN))
    out.write("#pragma once\n#define VP_LEAF_ASM_THREADS 1024\n")
    out.write("// addr0 / addr1: this thread's two field elements of slice 0 (any thread, any leaf of any tree: nothing here is uniform but `count`); stride: bytes from\n")
    out.write("// one slice to the next; out: where the 32-byte digest goes; active == 0: the thread runs the chain (it must stay in step with its workgroup) and\n")
    out.write("// stores nothing.\n")
    out.write("__device__ __forceinline__ void vp_leaf_chain_asm(const void *addr0, const void *addr1, unsigned stride, unsigned count, void *out, unsigned active) {\n")
    out.write("    asm volatile(\n")
    for t in L:
        out.write('        "%s\\n\\t"\n' % t)
    out.write('        :\n')
    out.write('        : [addr0] "v"(addr0), [addr1] "v"(addr1), [stride] "v"(stride), [count] "s"(count), [out] "v"(out), [active] "v"(active)\n')
    out.write('        : "memory", "vcc", "scc", "s44", "s45", "s46", "s47", %s);\n' % ", ".join('"v%d"' % r for r in regs["used"]))
    out.write("}\n")
    sys.stderr.write("per block: %s; fixed registers %d\n" % (n, len(regs["used"])))


if __name__ == "__main__":
    rot1 = "alignbit"
    barriers = True
    args = sys.argv[1:]
    if "--rot1" in args:
        rot1 = args[args.index("--rot1") + 1]
    if "--no-barriers" in args:
        barriers = False
    addr = args[args.index("--addr") + 1] if "--addr" in args else "mad"
    bar_mode = args[args.index("--barrier-at") + 1] if "--barrier-at" in args else "after"
    bar_every = int(args[args.index("--barrier-every") + 1]) if "--barrier-every" in args else 1
    emit_header(sys.stdout, rot1, barriers, "--msg-before-barrier" not in args, addr, bar_mode, bar_every)
