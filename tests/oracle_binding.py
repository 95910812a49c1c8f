"""ctypes binding of the oracle (oracle/_build/libvp_oracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libvp_oracle.so")


class Stats(ctypes.Structure):
    _fields_ = [("prove_sec", ctypes.c_double), ("evaluate_sec", ctypes.c_double), ("verify_sec", ctypes.c_double),
                ("mult_count", ctypes.c_uint64), ("add_count", ctypes.c_uint64), ("rounds", ctypes.c_uint64),
                ("pairs", ctypes.c_uint64), ("proof_kb", ctypes.c_double), ("verified", ctypes.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(ROOT, "oracle", f) for f in ("vp_oracle.cpp", "vp_oracle.h")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(LIB)
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.orc_circuit_from_pws.restype = vp
        L.orc_circuit_from_pws.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_long]
        L.orc_circuit_randomize.restype = vp
        L.orc_circuit_randomize.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
        L.orc_circuit_custom.restype = vp
        L.orc_circuit_custom.argtypes = [ctypes.c_int] + [vp] * 7
        L.orc_circuit_free.argtypes = [vp]
        L.orc_circuit_layers.argtypes = [vp]
        L.orc_circuit_layer_size.restype = u64
        L.orc_circuit_layer_size.argtypes = [vp, ctypes.c_int]
        L.orc_circuit_layer_bitlen.argtypes = [vp, ctypes.c_int]
        L.orc_circuit_gates.restype = u64
        L.orc_circuit_gates.argtypes = [vp]
        L.orc_circuit_hash.argtypes = [vp, ctypes.POINTER(u64)]
        L.orc_circuit_inputs.argtypes = [vp, vp]
        L.orc_prove_gkr.restype = ctypes.c_int64
        L.orc_prove_gkr.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(Stats)]
        L.orc_prove_fs.restype = ctypes.c_int64
        L.orc_prove_fs.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(Stats)]
        for f in ("orc_f_add", "orc_f_sub", "orc_f_mul"):
            getattr(L, f).argtypes = [vp, vp, vp]
        L.orc_f_neg.argtypes = [vp, vp]
        L.orc_f_inv.argtypes = [vp, vp]
        L.orc_f_root_of_unity.argtypes = [ctypes.c_int, vp]
        L.orc_f_random_seq.argtypes = [ctypes.c_uint, ctypes.c_int, vp]
        L.orc_beta_table.argtypes = [vp, ctypes.c_int, vp, vp]
        L.orc_update_each.argtypes = [vp, vp, vp, u64, u64, vp, vp]
        _lib = L
    return _lib


class Circuit:
    def __init__(self, h):
        if not h:
            raise RuntimeError("oracle circuit construction failed")
        self.h = h

    @classmethod
    def from_pws(cls, path, blocks=1, seed=-1):
        return cls(lib().orc_circuit_from_pws(os.fsencode(path), blocks, seed))

    @classmethod
    def randomize(cls, layers, log_size, seed=-1):
        return cls(lib().orc_circuit_randomize(layers, log_size, seed))

    @classmethod
    def custom(cls, layer_sizes, ty, l, u, v, c_pairs, is_assert):
        import numpy as np
        a = [np.ascontiguousarray(layer_sizes, np.uint64), np.ascontiguousarray(ty, np.int32), np.ascontiguousarray(l, np.int32),
             np.ascontiguousarray(u, np.uint64), np.ascontiguousarray(v, np.uint64), np.ascontiguousarray(c_pairs, np.uint64),
             np.ascontiguousarray(is_assert, np.uint8)]
        return cls(lib().orc_circuit_custom(len(a[0]), *[x.ctypes.data for x in a]))

    @property
    def layers(self):
        return lib().orc_circuit_layers(self.h)

    @property
    def gates(self):
        return lib().orc_circuit_gates(self.h)

    def layer_size(self, i):
        return lib().orc_circuit_layer_size(self.h, i)

    def hash(self):
        out = (ctypes.c_uint64 * 2)()
        lib().orc_circuit_hash(self.h, out)
        return "%016x%016x" % (out[0], out[1])

    def prove_gkr(self, capacity=1 << 20):
        """F::init() + the reference's GKR protocol on the CPU: (transcript bytes, stats dict)."""
        buf = ctypes.create_string_buffer(capacity)
        st = Stats()
        n = lib().orc_prove_gkr(self.h, ctypes.cast(buf, ctypes.c_void_p), capacity, ctypes.byref(st))
        if n < 0:
            raise RuntimeError("oracle prove failed")
        return buf.raw[:n], st.as_dict()

    def prove_fs(self, capacity=1 << 20):
        """The same proof in Fiat-Shamir mode (orc_prove_fs): (proof bytes, stats dict)."""
        buf = ctypes.create_string_buffer(capacity)
        st = Stats()
        n = lib().orc_prove_fs(self.h, ctypes.cast(buf, ctypes.c_void_p), capacity, ctypes.byref(st))
        if n < 0:
            raise RuntimeError("oracle FS prove failed")
        return buf.raw[:n], st.as_dict()

    def close(self):
        if self.h:
            lib().orc_circuit_free(self.h)
            self.h = None
