"""Random layered circuits that use EVERY gate type of the reference's enum gateType (src/inputCircuit.hpp:13-15),
including the ones its .pws loader never produces (Addc, Mulc, Copy, AntiNaab, AntiSub) and assert gates
(src/prover.cpp:18-21,209-212), for parity tests between the oracle and the device."""
import numpy as np

MUL, ADD, SUB, ANTISUB, NAAB, ANTINAAB, INPUT, MULC, ADDC, XOR, NOT, COPY = range(12)
P = (1 << 61) - 1


def make(seed, layer_sizes, with_asserts=True):
    rng = np.random.default_rng(seed)
    ty, l, u, v, c, a = [], [], [], [], [], []
    for i, n in enumerate(layer_sizes):
        for g in range(n):
            if i == 0:
                ty.append(INPUT); l.append(-1); u.append(int(rng.integers(0, P))); v.append(0); c.append((0, 0)); a.append(0)
                continue
            if with_asserts and g == n - 1:
                # x - x == 0: a legal assert gate (both operands the same wire of layer i-1)
                w = int(rng.integers(0, layer_sizes[i - 1]))
                ty.append(SUB); l.append(i - 1); u.append(w); v.append(w); c.append((0, 0)); a.append(1)
                continue
            t = int(rng.choice([MUL, ADD, SUB, ANTISUB, NAAB, ANTINAAB, MULC, ADDC, XOR, NOT, COPY]))
            ty.append(t)
            u.append(int(rng.integers(0, layer_sizes[i - 1])))
            if t in (MULC, ADDC, NOT, COPY):
                l.append(-1); v.append(0)
            else:
                ll = int(rng.integers(0, i))
                l.append(ll); v.append(int(rng.integers(0, layer_sizes[ll])))
            c.append((int(rng.integers(0, P)), int(rng.integers(0, P))) if t in (MULC, ADDC) else (0, 0))
            a.append(0)
    return (np.array(layer_sizes, np.uint64), np.array(ty, np.int32), np.array(l, np.int32), np.array(u, np.uint64),
            np.array(v, np.uint64), np.array(c, np.uint64).reshape(-1, 2), np.array(a, np.uint8))
