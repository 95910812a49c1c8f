#!/usr/bin/env python3
"""CPU check of tools/gen_keccak_asm.py: interprets the generated instruction list (the five VALU opcodes it uses) and compares a chain of blocks with
hashlib's SHA3-256 — every variant, with and without dead-code elimination.  Run by tests/test_keccak_asm_gen.py."""
import hashlib, os, random, struct, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_keccak_asm as g


def run(code, R):
    def val(t):
        t = t.strip()
        if t.startswith("0x"):
            return int(t, 16)
        if t.lstrip("-").isdigit():
            return int(t)
        return R[t]
    for ins in code:
        t = ins.text
        if t in ("s_barrier", "MSG_DEAD"):
            continue
        op, rest = t.split(" ", 1)
        if op == "v_bitop3_b32":
            args, imm = rest.split(" bitop3:")
            d, a, b, c = [x.strip() for x in args.split(",")]
            imm = int(imm, 16)
            A, B, C = val(a), val(b), val(c)
            r = 0
            for bit in range(32):
                r |= ((imm >> ((((A >> bit) & 1) << 2) | (((B >> bit) & 1) << 1) | ((C >> bit) & 1))) & 1) << bit
            R[d] = r
        elif op == "v_xor_b32":
            d, a, b = [x.strip() for x in rest.split(",")]
            R[d] = val(a) ^ val(b)
        elif op == "v_alignbit_b32":
            d, a, b, sh = [x.strip() for x in rest.split(",")]
            R[d] = (((val(a) << 32) | val(b)) >> int(sh)) & 0xffffffff
        elif op == "v_mov_b32":
            d, a = [x.strip() for x in rest.split(",")]
            R[d] = val(a)
        elif op == "v_add_u32":
            d, a, b = [x.strip() for x in rest.split(",")]
            R[d] = (val(a) + val(b)) & 0xffffffff
        elif op == "v_lshrrev_b32":
            d, sh, a = [x.strip() for x in rest.split(",")]
            R[d] = val(a) >> int(sh)
        else:
            raise RuntimeError("unknown instruction " + t)


def check(rot1, dce, blocks=3, seed=1):
    code, regs = g.build_body(rot1, True, dce=dce)
    rnd = random.Random(seed)
    R = {"v%d" % r: rnd.getrandbits(32) for r in range(g.BASE, g.BASE + g.SPAN)}      # whatever the registers held before
    h = bytes(32)
    for r in regs["digest"]:
        R[r] = 0
    for _ in range(blocks):
        m = [rnd.getrandbits(32) for _ in range(8)]
        for r, v in zip(regs["M"], m):
            R[r] = v
        run(code, R)
        h = hashlib.sha3_256(struct.pack("<8I", *m) + h).digest()
        got = struct.pack("<8I", *[R[r] for r in regs["digest"]])
        if got != h:
            return False
    return True


if __name__ == "__main__":
    ok = True
    for rot1 in ("alignbit", "fast"):
        for dce in (True, False):
            r = check(rot1, dce)
            code, regs = g.build_body(rot1, True, dce=dce)
            print("rot1 %-8s dce %-5s: %s   %s, fixed registers %d" % (rot1, dce, "ok" if r else "MISMATCH", g.stats(code), len(regs["used"])))
            ok = ok and r
    sys.exit(0 if ok else 1)
