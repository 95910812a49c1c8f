#!/usr/bin/env python3
"""Generator of virgo-plus_amd/csrc/vp_keccak_asm.h: SHA3-256 of one 64-byte block (my_hhash.h:27-33 of the reference: the leaf chains and the
Merkle nodes) as ONE inline-asm block with hand-placed registers.

Why: on gfx950 a VOP3 instruction with three DISTINCT VGPR sources of which two sit in the same register bank (index mod 4) issues in ~4.2
cycles instead of ~2.9 (tools/micro_bank.py, profiles/r04_micro_bank_keccak_instructions.txt); the compiler's allocation leaves 59 % of the
v_bitop3_b32 of Keccak-f[1600] with such a pair, which is exactly the gap between the kernel and its instruction count.  Here
  * theta's column sums take v_bitop3_b32 on operands placed in three banks (state lane (x, y) lives in bank f(y));
  * theta's update A ^= D is a two-operand v_xor_b32 (VOP2 reads two sources: no bank rule, and cheaper than a v_bitop3_b32);
  * chi's a ^ (~b & c) takes its row from five staging registers in banks (0,1,2,3,1): one unavoidable pair per row (five lanes, four banks);
  * the lanes of the padded message that are constants are folded at generation time, and everything that does not reach the four output
    lanes is dropped from the last round.
The state is double-buffered (round r reads set r mod 2, chi writes the other set): 136 fixed registers.

    python3 tools/gen_keccak_asm.py > virgo-plus_amd/csrc/vp_keccak_asm.h
"""
import sys

BASE = 16                     # first fixed register; the block owns v[BASE, BASE + 136)
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001, 0x8000000080008081,
      0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a, 0x000000008000808b, 0x800000000000008b,
      0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800a, 0x800000008000000a, 0x8000000080008081,
      0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]      # ROT[x][y]


class Pool:
    def __init__(self, base, n):
        self.free = {b: [r for r in range(base, base + n) if r % 4 == b] for b in range(4)}

    def take(self, bank=None):
        if bank is None:
            bank = max(range(4), key=lambda b: len(self.free[b]))
        return self.free[bank].pop(0)


class Val:
    """a 32-bit value: a constant or a register (physical 'vN' or an asm operand '%N')"""

    def __init__(self, const=None, reg=None):
        self.const, self.reg = const, reg

    def is_const(self):
        return self.reg is None

    def __repr__(self):
        return self.reg if self.reg is not None else hex(self.const)


class Ins:
    def __init__(self, text, dst, srcs):
        self.text, self.dst, self.srcs = text, dst, srcs


def build(rounds=24, debug_state=False):
    pool = Pool(BASE, 136)
    f = [(0, 1, 2, 3, 0), (0, 2, 3, 1, 2)]                   # bank of state lane (x, y) in set s: f[s][y]
    g = (0, 1, 2, 3, 1)                                      # bank of chi's staging register for column x
    S = [[[[None, None] for _ in range(5)] for _ in range(5)] for _ in range(2)]
    for s in range(2):
        for y in range(5):
            for x in range(5):
                for h in range(2):
                    S[s][x][y][h] = "v%d" % pool.take(f[s][y])
    B = [["v%d" % pool.take(g[x]) for _ in range(2)] for x in range(5)]
    T = [["v%d" % pool.take(1) for _ in range(2)], ["v%d" % pool.take(3) for _ in range(2)]]      # theta's intermediate, outside the banks of rows 3, 4
    C = [["v%d" % pool.take() for _ in range(2)] for _ in range(5)]
    D = [["v%d" % pool.take() for _ in range(2)] for _ in range(5)]
    R = ["v%d" % pool.take() for _ in range(2)]
    used = sorted(int(r[1:]) for r in sum([sum(sum(sum(S, []), []), []), sum(B, []), sum(T, []), sum(C, []), sum(D, []), R], []))
    assert len(set(used)) == 136 and used[0] == BASE and used[-1] == BASE + 135
    code = []

    def emit(text, dst, srcs):
        code.append(Ins(text, dst, [s for s in srcs if s is not None]))

    def xor_into(dst, vals):
        """dst <- xor of vals (folded); returns the Val holding the result (a constant, an alias of a source, or dst)"""
        k = 0
        regs = []
        for v in vals:
            if v.is_const():
                k ^= v.const
            else:
                regs.append(v.reg)
        # a register xor-ed with itself cancels
        out = []
        for r in regs:
            if r in out:
                out.remove(r)
            else:
                out.append(r)
        regs = out
        if not regs:
            return Val(const=k)
        if len(regs) == 1 and k == 0:
            return Val(reg=regs[0])
        cur = None
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
        """(lo, hi) rotated left by n -> Vals (possibly aliases or constants); dlo / dhi are the registers to use when an instruction is needed"""
        if lo.is_const() and hi.is_const():
            v = (hi.const << 32) | lo.const
            v = ((v << n) | (v >> (64 - n))) & 0xffffffffffffffff if n else v
            return Val(const=v & 0xffffffff), Val(const=v >> 32)
        if n == 0:
            return lo, hi
        if n == 32:
            return hi, lo
        assert not lo.is_const() and not hi.is_const()          # after theta no lane is constant
        if n < 32:                                               # ohi = alignbit(hi, lo, 32 - n), olo = alignbit(lo, hi, 32 - n)
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dhi, hi.reg, lo.reg, 32 - n), dhi, [hi.reg, lo.reg])
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dlo, lo.reg, hi.reg, 32 - n), dlo, [lo.reg, hi.reg])
        else:                                                    # ohi = alignbit(lo, hi, 64 - n), olo = alignbit(hi, lo, 64 - n)
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dhi, lo.reg, hi.reg, 64 - n), dhi, [lo.reg, hi.reg])
            emit("v_alignbit_b32 %s, %s, %s, %d" % (dlo, hi.reg, lo.reg, 64 - n), dlo, [hi.reg, lo.reg])
        return Val(reg=dlo), Val(reg=dhi)

    # round 0 input: lanes 0-3 message (operands %8..%15), 4-7 the previous digest (operands %0..%7), 8 = 0x06, 16 = 0x8000000000000000, rest 0
    A = [[[Val(const=0), Val(const=0)] for _ in range(5)] for _ in range(5)]
    for i in range(4):
        A[i % 5][i // 5] = [Val(reg="%%%d" % (8 + 2 * i)), Val(reg="%%%d" % (9 + 2 * i))]
    for i in range(4, 8):
        A[i % 5][i // 5] = [Val(reg="%%%d" % (2 * (i - 4))), Val(reg="%%%d" % (2 * (i - 4) + 1))]
    A[8 % 5][8 // 5] = [Val(const=0x06), Val(const=0)]
    A[16 % 5][16 // 5] = [Val(const=0), Val(const=0x80000000)]
    for rnd in range(rounds):
        s = rnd & 1
        cur, nxt = S[s], S[s ^ 1]
        # theta: column sums (three-bank operands: rows 0, 1, 2 first, then the intermediate with rows 3, 4)
        Cv = []
        for x in range(5):
            pair = []
            for h in range(2):
                col = [A[x][y][h] for y in range(5)]
                if sum(1 for v in col if not v.is_const()) <= 3:          # round 0: most lanes of the padded block are constants
                    pair.append(xor_into(C[x][h], col))
                else:
                    t = xor_into(T[s][h], col[:3])
                    pair.append(xor_into(C[x][h], [t, col[3], col[4]]))
            Cv.append(pair)
        Dv = []
        for x in range(5):
            rl, rh = rot64(Cv[(x + 1) % 5][0], Cv[(x + 1) % 5][1], 1, R[0], R[1])
            Dv.append([xor_into(D[x][0], [Cv[(x + 4) % 5][0], rl]), xor_into(D[x][1], [Cv[(x + 4) % 5][1], rh])])
        for x in range(5):
            for y in range(5):
                for h in range(2):
                    A[x][y][h] = xor_into(cur[x][y][h], [A[x][y][h], Dv[x][h]])
        # rho + pi + chi, row by row of the output: B[X][Y] = rot(A[x][y]) with X = y, Y = 2x + 3y
        N = [[[None, None] for _ in range(5)] for _ in range(5)]
        for Y in range(5):
            row = []
            for X in range(5):
                y = X
                x = next(xx for xx in range(5) if (2 * xx + 3 * y) % 5 == Y)
                row.append(rot64(A[x][y][0], A[x][y][1], ROT[x][y], B[X][0], B[X][1]))
            for X in range(5):
                for h in range(2):
                    a, b, c = row[X][h], row[(X + 1) % 5][h], row[(X + 2) % 5][h]
                    assert not (a.is_const() or b.is_const() or c.is_const())
                    d = nxt[X][Y][h]
                    emit("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd2" % (d, a.reg, b.reg, c.reg), d, [a.reg, b.reg, c.reg])      # a ^ (~b & c)
                    N[X][Y][h] = Val(reg=d)
        for h in range(2):
            k = (RC[rnd] >> (32 * h)) & 0xffffffff
            if k:
                d = nxt[0][0][h]
                emit("v_xor_b32 %s, 0x%x, %s" % (d, k, d), d, [d])
        A = N
    if debug_state:
        return code, A
    # digest = lanes 0..3 -> operands %0..%7
    outs = []
    for i in range(4):
        for h in range(2):
            src = A[i][0][h].reg
            emit("v_mov_b32 %%%d, %s" % (2 * i + h, src), "%%%d" % (2 * i + h), [src])
            outs.append("%%%d" % (2 * i + h))
    # dead-code elimination, backwards (registers are re-used: liveness by name)
    live = set(outs)
    keep = []
    for ins in reversed(code):
        if ins.dst in live:
            live.discard(ins.dst)
            live.update(ins.srcs)
            keep.append(ins)
    keep.reverse()
    return keep, used


def main():
    code, used = build()
    n_bitop = sum(1 for i in code if i.text.startswith("v_bitop3"))
    n_align = sum(1 for i in code if i.text.startswith("v_alignbit"))
    n_xor = sum(1 for i in code if i.text.startswith("v_xor"))
    conflicts = 0
    for i in code:
        if i.text.startswith("v_bitop3"):
            regs = [int(r[1:]) for r in set(i.srcs) if r.startswith("v")]
            banks = [r % 4 for r in regs]
            if len(set(banks)) < len(banks):
                conflicts += 1
    out = sys.stdout
    out.write("// GENERATED by tools/gen_keccak_asm.py — do not edit.  SHA3-256 of a 64-byte block (message words m[0..8), previous digest h[0..8) in, digest out in h)\n")
    out.write("// as one inline-asm block on fixed registers v[%d, %d): %d v_bitop3_b32 (%d with two sources in one bank), %d v_alignbit_b32, %d v_xor_b32, 8 v_mov_b32.\n"
              % (BASE, BASE + 136, n_bitop, conflicts, n_align, n_xor))
    out.write("#pragma once\n")
    out.write("#define VP_KECCAK_ASM_INSTRUCTIONS %d\n" % (len(code)))
    out.write("__device__ __forceinline__ void vp_hhash64_asm(unsigned (&h)[8], const unsigned (&m)[8]) {\n")
    out.write("    asm volatile(\n")
    for i in code:
        out.write('        "%s\\n\\t"\n' % i.text)
    out.write('        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7])\n')
    out.write('        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7])\n')
    out.write("        : %s);\n" % ", ".join('"v%d"' % r for r in used))
    out.write("}\n")
    sys.stderr.write("instructions %d: bitop3 %d (bank pairs %d), alignbit %d, xor %d\n" % (len(code), n_bitop, conflicts, n_align, n_xor))


if __name__ == "__main__":
    main()
