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


def write_case_file(name, path):
    """The IN file of `ref_run --pc-masked` / `ref_run_vpgpu_masked` (oracle/ref_driver.cpp): i32 n, i32 m, values[2^n], pub[2^n], pri_mask[m], pub_mask[m]."""
    import struct
    x = inputs(name)
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", x["n"], x["m"]))
        for k in ("values", "pub", "pri_mask", "pub_mask"):
            f.write(x[k].tobytes())
    return x


# what the reference's OWN verifier (poly_commit_verifier::verify_poly_commitment, vpd_verifier.cpp:76-328) says about the reference's own commitment of each
# case (oracle/_ref/ref_run --pc-masked IN --verify, measured in the build container): a mask that pads to MORE than a slice's message (2^(n-6) elements) is
# not reduced to a constant by the n - 6 folds, and the verifier's last check ("Fri msk rs code check fail", :318-324) rejects it — prover and verifier of
# lib/virgo agree only up to that length.  The device prover must get the same verdicts.
REFERENCE_VERIFIER_ACCEPTS = {"n13_zero": True, "n13_m5": True, "n13_m64": True, "n16_m100": True, "n19_m3000": True, "n13_m300": False, "n13_m2000": False}
