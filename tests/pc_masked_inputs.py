"""Inputs of the masked-commitment cases (tests/golden/make_pc_masked.py writes the goldens from the REAL reference with them; the GPU test hands the same
arrays to the device): seeded, canonical limbs, (count, 2) uint64."""
import numpy as np

P61 = (1 << 61) - 1
# name -> (input bit length n, mask length m, seed).  Masks that pad to fewer than 8 elements are not cases: the reference's own transforms of fewer than 8 points
# read stale scratch (RS_polynomial.cpp:104-133, and its packed leaf loop runs zero times below 4 coefficients), its commitment is not a function of such a mask.
CASES = {"n13_zero": (13, 1, 10), "n13_m5": (13, 5, 11), "n13_m64": (13, 64, 13), "n16_m100": (16, 100, 14), "n19_m3000": (19, 3000, 15),
         # masks longer than a slice's message (2^(n-6) elements): 4 and 16 blocks of it, the longest the reference takes (mask_position_gap = 2)
         "n13_m300": (13, 300, 16), "n13_m2000": (13, 2000, 17)}


def inputs(name):
    n, m, seed = CASES[name]
    rng = np.random.default_rng(seed)
    f = lambda cnt: rng.integers(0, P61, size=(cnt, 2), dtype=np.uint64)
    x = {"n": n, "m": m, "values": f(1 << n), "pub": f(1 << n), "pri_mask": f(m), "pub_mask": f(m)}
    if name.endswith("_zero"):               # the protocol's own case (one zero each): what the unmasked entry points must reproduce from the same record layout
        x["pri_mask"][:] = 0; x["pub_mask"][:] = 0
    return x
