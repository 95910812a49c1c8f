"""virgo-plus_amd — MI355X-native GKR sumcheck prover (Virgo++), Python harness side.

The product is native: ``csrc/`` (HIP kernels + the C ABI of include/vpgpu.h -> libvpgpu.so) and
``host/`` (C++ mirror of the reference's prover / verifier / circuit-loader API -> libvphost.so and the
``virgo_plus_run`` CLI).  This module only builds those in-tree and exposes them through ctypes for
tests/ and bench.py.  There is no CPU fallback: creating a Session without a working gfx950 device and
the built libraries raises.

The directory name contains a hyphen; load it with ``load()`` from the repo root's ``vp_loader.py`` or
importlib (see tests/conftest.py).
"""
import ctypes
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
HOST = os.path.join(PKG_DIR, "host")
# VP_LIBGPU: development override (A/B builds of the kernels under tools/_build/<variant>/libvpgpu.so are measured without touching the
# product library; build() never writes to an overridden path)
LIB_GPU = os.environ.get("VP_LIBGPU") or os.path.join(CSRC, "libvpgpu.so")
LIB_GPU_CHECKED = os.path.join(ROOT, "tools", "_build", "checked", "libvpgpu.so")      # -DVP_CHECKED flavour (csrc/vp_check.h), loaded through VP_LIBGPU by its test
LIB_GPU_TESTDRV = os.path.join(ROOT, "tools", "_build", "testdrv", "libvpgpu.so")      # -DVP_TEST_DRIVERS flavour: the launch plan's two cross-check drivers (VP_GKR_PATH=lanes|simple), loaded through VP_LIBGPU by their test
LIB_HOST = os.path.join(HOST, "libvphost.so")
CLI = os.path.join(HOST, "virgo_plus_run")

GPU_SRC = [os.path.join(CSRC, f) for f in ("vpgpu.hip", "vpgpu_batched.inc", "vpgpu_pc.inc", "vpgpu_pc_shard.inc", "vpgpu_fftgkr.inc", "vpgpu_upload.inc", "vp_kernels_fftgkr.h", "vp_kernels.h", "vp_kernels_round.h", "vp_kernels_persist.h", "vp_kernels_batch.h",
                                              "vp_kernels_plan.h", "vp_kernels_pc.h", "vp_kernels_ntt8.h", "vp_keccak_asm.h", "vp_check.h", "vp_field.h")] + [
    os.path.join(ROOT, "include", "vpgpu.h")]
HOST_SRC = [os.path.join(HOST, f) for f in ("circuit.cpp", "prover.cpp", "verifier.cpp", "vphost.cpp")]
HOST_HDR = [os.path.join(HOST, f) for f in ("circuit.hpp", "prover.hpp", "verifier.hpp", "vphost.h", "field.hpp",
                                              "polynomial.hpp")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def build(force=False, verbose=False):
    """Compile libvpgpu.so (hipcc, gfx950), libvphost.so and virgo_plus_run (g++) in-tree."""
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if os.environ.get("VP_LIBGPU"):
        pass                                            # development override: never rebuilt from here
    elif force or _stale(LIB_GPU, GPU_SRC):
        hipcc = _hipcc()
        if hipcc is None:
            if os.path.exists(LIB_GPU):
                return  # prebuilt library travelled with the tree (GPU box without a toolchain)
            raise RuntimeError("hipcc not found and no prebuilt libvpgpu.so")
        run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
             "-o", LIB_GPU, os.path.join(CSRC, "vpgpu.hip")])
    if (force or _stale(LIB_GPU_CHECKED, GPU_SRC)) and _hipcc() is not None:
        os.makedirs(os.path.dirname(LIB_GPU_CHECKED), exist_ok=True)
        run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DVP_CHECKED",
             "-o", LIB_GPU_CHECKED, os.path.join(CSRC, "vpgpu.hip")])
    if (force or _stale(LIB_GPU_TESTDRV, GPU_SRC)) and _hipcc() is not None:
        os.makedirs(os.path.dirname(LIB_GPU_TESTDRV), exist_ok=True)
        run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DVP_TEST_DRIVERS",
             "-o", LIB_GPU_TESTDRV, os.path.join(CSRC, "vpgpu.hip")])
    if force or _stale(LIB_HOST, HOST_SRC + HOST_HDR + [LIB_GPU]):
        run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-pthread", "-o", LIB_HOST] + HOST_SRC +
            ["-L" + CSRC, "-lvpgpu", "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath,/opt/rocm/lib"])
    if force or _stale(CLI, [os.path.join(HOST, "virgo_plus_run.cpp"), LIB_HOST]):
        run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-o", CLI, os.path.join(HOST, "virgo_plus_run.cpp"),
             "-L" + HOST, "-lvphost", "-L" + CSRC, "-lvpgpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../csrc",
             "-Wl,-rpath,/opt/rocm/lib"])


class VphResult(ctypes.Structure):
    _fields_ = [("prove_sec", ctypes.c_double), ("gkr_device_ms", ctypes.c_double), ("evaluate_ms", ctypes.c_double),
                ("verify_sec", ctypes.c_double), ("fold_ms", ctypes.c_double), ("fold_launches", ctypes.c_uint64),
                ("fold_bytes", ctypes.c_uint64), ("rounds", ctypes.c_uint64), ("launches", ctypes.c_uint64),
                ("proof_kb", ctypes.c_double), ("verified", ctypes.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RoundStat(ctypes.Structure):           # vp_round_stat (include/vpgpu.h)
    _fields_ = [("phase", ctypes.c_int32), ("layer", ctypes.c_int32), ("round", ctypes.c_int32), ("how", ctypes.c_int32), ("tables", ctypes.c_int32),
                ("bytes", ctypes.c_uint64), ("us", ctypes.c_double)]


class LaunchStat(ctypes.Structure):          # vp_launch_stat (include/vpgpu.h)
    _fields_ = [("kind", ctypes.c_int32), ("step", ctypes.c_int32), ("workgroups", ctypes.c_uint32), ("jobs", ctypes.c_uint32),
                ("rounds", ctypes.c_uint32), ("first_round", ctypes.c_uint32), ("bytes", ctypes.c_uint64), ("work", ctypes.c_uint64),
                ("us", ctypes.c_double)]


_gpu = None
_host = None


def lib_gpu():
    """ctypes handle of libvpgpu.so (raises if it is not built)."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(LIB_GPU):
            raise RuntimeError("libvpgpu.so is not built: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_GPU, mode=ctypes.RTLD_GLOBAL)
        vp = ctypes.c_void_p
        L.vp_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.vp_create_with_options.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
        L.vp_get_options.argtypes = [vp, vp]
        L.vp_get_resident_resumes.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.vp_set_shard_split.argtypes = [vp, ctypes.c_int]
        L.vp_shard_finish.argtypes = [vp, vp, ctypes.c_uint64, vp]
        L.vp_gkr_sizes.argtypes = [vp, vp, vp]
        L.vp_options_default.argtypes = [vp]
        L.vp_tuning_get.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32)]
        u64_ = ctypes.c_uint64
        L.vp_set_deferred.argtypes = [vp, ctypes.c_int]
        L.vp_warm.argtypes = [vp, ctypes.c_uint32]
        L.vp_flush.argtypes = [vp, ctypes.c_int]
        L.vp_pending.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.vp_phase_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.vp_commit_private.argtypes = [vp, vp]
        L.vp_fft_gkr_sizes.argtypes = [ctypes.c_int, vp, vp]
        L.vp_fft_gkr.argtypes = [vp, ctypes.c_int, vp, u64_, vp, u64_, vp]
        L.vp_fft_gkr_begin.argtypes = [vp, ctypes.c_int, vp, u64_]
        L.vp_fft_gkr_end.argtypes = [vp, vp, u64_, vp]
        L.vp_fft_gkr_cancel.argtypes = [vp]
        L.vp_options_default.restype = None
        L.vp_destroy.argtypes = [vp]
        L.vp_last_error.argtypes = [vp]
        L.vp_last_error.restype = ctypes.c_char_p
        L.vp_version.restype = ctypes.c_char_p
        L.vp_test_field.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_uint64]
        L.vp_test_beta.argtypes = [vp, vp, ctypes.c_int, vp, vp]
        L.vp_test_sha3.argtypes = [vp, vp, vp, ctypes.c_uint64]
        L.vp_test_fft.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        L.vp_pc_load_input.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_int]
        L.vp_pc_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.vp_shard_exchange_local.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
        L.vp_commit_private.argtypes = [vp, vp]
        L.vp_commit_public.argtypes = [vp, vp, ctypes.c_uint64, vp, vp, vp]
        L.vp_commit_public_eq.argtypes = [vp, vp, ctypes.c_int, vp, vp, vp]
        L.vp_fri_commit.argtypes = [vp, vp, ctypes.c_int, vp]
        L.vp_fri_final.argtypes = [vp, vp]
        L.vp_fri_open.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vp_commit_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.vp_comm_unique_id.argtypes = [vp]
        L.vp_comm_init.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
        L.vp_comm_destroy.argtypes = [vp]
        L.vp_comm_count.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.vp_get_round_stats.argtypes = [vp, ctypes.POINTER(RoundStat), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vp_allreduce_u64.argtypes = [vp, vp, ctypes.c_uint64]
        L.vp_set_profiling.argtypes = [vp, ctypes.c_int]
        L.vp_get_launch_stats.argtypes = [vp, ctypes.POINTER(LaunchStat), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vp_kernel_name.argtypes = [ctypes.c_int]
        L.vp_kernel_name.restype = ctypes.c_char_p
        _gpu = L
    return _gpu


def lib_host():
    """ctypes handle of libvphost.so (raises if it is not built)."""
    global _host
    if _host is None:
        lib_gpu()
        if not os.path.exists(LIB_HOST):
            raise RuntimeError("libvphost.so is not built: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_HOST)
        vp = ctypes.c_void_p
        u64 = ctypes.c_uint64
        L.vph_circuit_from_pws.restype = vp
        L.vph_circuit_from_pws.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_long, ctypes.c_char_p, ctypes.c_int]
        L.vph_circuit_randomize.restype = vp
        L.vph_circuit_randomize.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
        L.vph_circuit_custom.restype = vp
        L.vph_circuit_custom.argtypes = [ctypes.c_int] + [vp] * 7
        L.vph_circuit_free.argtypes = [vp]
        L.vph_circuit_layers.argtypes = [vp]
        L.vph_circuit_gates.restype = u64
        L.vph_circuit_gates.argtypes = [vp]
        L.vph_circuit_layer_size.restype = u64
        L.vph_circuit_layer_size.argtypes = [vp, ctypes.c_int]
        L.vph_circuit_layer_bitlen.argtypes = [vp, ctypes.c_int]
        L.vph_circuit_hash.argtypes = [vp, ctypes.POINTER(u64)]
        L.vph_session_create.restype = vp
        L.vph_session_create.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.vph_session_create_opts.restype = vp
        L.vph_session_create_opts.argtypes = [vp, ctypes.c_int, vp, ctypes.c_char_p, ctypes.c_int]
        L.vph_session_free.argtypes = [vp]
        L.vph_set_profiling.argtypes = [vp, ctypes.c_int]
        L.vph_session_ctx.restype = vp
        L.vph_session_ctx.argtypes = [vp]
        L.vph_layer_values.argtypes = [vp, ctypes.c_int, vp, u64]
        L.vph_prove_interactive.argtypes = [vp, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(VphResult), ctypes.c_char_p,
                                            ctypes.c_int]
        L.vph_draw_tape.argtypes = [vp]
        L.vph_prove_gkr.argtypes = [vp, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(VphResult), ctypes.c_char_p, ctypes.c_int]
        L.vph_check.argtypes = [vp, vp, u64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.vph_verify_transcript.argtypes = [vp, vp, u64, ctypes.c_int]
        L.vph_commit_public.argtypes = [vp, vp, u64, vp, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
        L.vph_prove_full.argtypes = [vp, vp, u64, ctypes.POINTER(u64), ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.vph_prove_and_verify_full.argtypes = [vp, ctypes.c_int, vp, u64, ctypes.POINTER(u64)] + [ctypes.POINTER(ctypes.c_double)] * 3 + [ctypes.c_char_p, ctypes.c_int]
        L.vph_last_fri.argtypes = [vp, vp, u64, vp, vp]
        L.vph_draw_protocol_tape.argtypes = [vp]
        L.vph_prove_protocol.argtypes = [vp, vp, u64, ctypes.POINTER(u64), vp, u64, vp, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
        L.vph_prove_protocol_ex.argtypes = [vp, vp, u64, ctypes.POINTER(u64), vp, u64, vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.vph_last_point.argtypes = [vp, vp, ctypes.c_int]
        L.vph_last_fft_gkr.restype = ctypes.c_int64
        L.vph_last_fft_gkr.argtypes = [vp, vp, u64]
        L.vph_last_pc_times.restype = None
        L.vph_last_pc_times.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.vph_interactive_breakdown.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.vph_test_sha3.argtypes = [vp, vp, u64]
        L.vph_fri_commit.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_char_p, ctypes.c_int]
        L.vph_fri_commit_batched.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_char_p, ctypes.c_int]
        L.vph_commit_private.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_int]
        L.vph_prove_fs.argtypes = [vp, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(VphResult), ctypes.c_char_p, ctypes.c_int]
        L.vph_verify_fs.argtypes = [vp, vp, u64]
        L.vph_commit_device_ms.restype = ctypes.c_double
        L.vph_commit_device_ms.argtypes = [vp]
        L.vph_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.vph_shard_chains.argtypes = [vp, vp, vp, ctypes.c_int]
        L.vph_shard_vu_partials.argtypes = [vp, vp, ctypes.c_int]
        L.vph_shard_vu_set.argtypes = [vp, vp, ctypes.c_int]
        L.vph_transcript_bytes.restype = u64
        L.vph_transcript_bytes.argtypes = [vp]
        _host = L
    return _host


def sum_transcripts(parts):
    """Element-wise u64 sum (mod 2^64) of the transcripts of a sharded proof = what the all-reduce over RCCL computes.  The
    ranks' slices are disjoint and everything else is zero, so the sum is the unsharded transcript."""
    import numpy as np
    acc = np.zeros(len(parts[0]) // 8, dtype=np.uint64)
    for p in parts:
        acc += np.frombuffer(p, dtype=np.uint64)
    return acc.tobytes()


def allreduce_transcript(tr, device=None):
    """The one data-path collective of a sharded proof: all-reduce (sum, int64 lanes) of the transcript over the default
    torch.distributed group — RCCL over xGMI when the group's backend is "nccl" (the transcript goes through a device tensor),
    gloo on the CPU in the tests.  Returns the assembled transcript bytes on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.frombuffer(tr, dtype=np.int64).copy())
    if "nccl" in dist.get_backend():
        t = t.cuda(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().tobytes()


class ShardedCommitment:
    """The Virgo commitment sharded over `world` ranks (include/vpgpu.h: vp_pc_set_shard), rehearsed as `world` light contexts of ONE
    process on one GPU: every rank runs the same entry points a multi-GPU run would, the collectives between them (one all-to-all per
    committed oracle, one all-gather of the level-5 tree nodes) are performed by vp_shard_exchange_local instead of RCCL."""
    VP_EXCHANGE = 1

    def __init__(self, inputs, bit_length, world, device=0):
        import numpy as np
        L = lib_gpu()
        self.world, self.n = world, bit_length
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        self.ctx = []
        for r in range(world):
            c = ctypes.c_void_p()
            if L.vp_create(device, ctypes.byref(c)):
                raise RuntimeError("vp_create failed")
            self.ctx.append(c)
            self._chk(L.vp_pc_load_input(c, inputs.ctypes.data, inputs.shape[0], bit_length), c, "vp_pc_load_input")
            self._chk(L.vp_pc_set_shard(c, r, world), c, "vp_pc_set_shard")
        self.arr = (ctypes.c_void_p * world)(*[c.value for c in self.ctx])

    @staticmethod
    def _chk(rc, c, what):
        if rc < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, lib_gpu().vp_last_error(c).decode()))

    def _run(self, call, what):
        """call(rank) on every rank until all report VP_OK, exchanging whenever all report VP_EXCHANGE."""
        L = lib_gpu()
        for _ in range(8):
            rcs = [call(r) for r in range(self.world)]
            for r, rc in enumerate(rcs):
                self._chk(rc, self.ctx[r], what)
            if all(rc == 0 for rc in rcs):
                return
            if not all(rc == self.VP_EXCHANGE for rc in rcs):
                raise RuntimeError("%s: ranks disagree on the protocol stage %r" % (what, rcs))
            self._chk(L.vp_shard_exchange_local(self.arr, self.world), self.ctx[0], "vp_shard_exchange_local")
        raise RuntimeError("%s did not finish" % what)

    def device_ms(self):
        out = []
        for c in self.ctx:
            ms = ctypes.c_double(0)
            lib_gpu().vp_commit_stats(c, ctypes.byref(ms))
            out.append(ms.value)
        return out

    def commit_private(self):
        roots = [ctypes.create_string_buffer(32) for _ in range(self.world)]
        self._run(lambda r: lib_gpu().vp_commit_private(self.ctx[r], ctypes.cast(roots[r], ctypes.c_void_p)), "vp_commit_private")
        assert all(x.raw == roots[0].raw for x in roots), "ranks disagree on the root"
        return roots[0].raw

    def commit_public(self, pub):
        import numpy as np
        pub = np.ascontiguousarray(pub, dtype=np.uint64)
        roots = [ctypes.create_string_buffer(32) for _ in range(self.world)]
        inner = [np.zeros(2, np.uint64) for _ in range(self.world)]
        alls = [np.zeros((65, 2), np.uint64) for _ in range(self.world)]
        self._run(lambda r: lib_gpu().vp_commit_public(self.ctx[r], pub.ctypes.data, pub.shape[0], inner[r].ctypes.data, alls[r].ctypes.data,
                                                       ctypes.cast(roots[r], ctypes.c_void_p)), "vp_commit_public")
        for r in range(1, self.world):
            assert roots[r].raw == roots[0].raw and inner[r].tobytes() == inner[0].tobytes() and alls[r].tobytes() == alls[0].tobytes()
        return roots[0].raw, inner[0].tobytes(), alls[0].tobytes()

    def fri_commit(self, r):
        import numpy as np
        r = np.ascontiguousarray(r, dtype=np.uint64)
        st = r.shape[0]
        roots = [ctypes.create_string_buffer(32 * st) for _ in range(self.world)]
        self._run(lambda k: lib_gpu().vp_fri_commit(self.ctx[k], r.ctypes.data, st, ctypes.cast(roots[k], ctypes.c_void_p)), "vp_fri_commit")
        fins = []
        for k in range(self.world):
            fin = np.zeros((2048, 2), dtype=np.uint64)
            self._chk(lib_gpu().vp_fri_final(self.ctx[k], fin.ctypes.data), self.ctx[k], "vp_fri_final")
            fins.append(fin)
        for k in range(1, self.world):
            assert roots[k].raw == roots[0].raw and np.array_equal(fins[k], fins[0])
        return roots[0].raw, fins[0]

    def open(self, oracle, leaf, rank=None):
        """(values (130, 2), path digests) of one leaf, asked of its owner (or of `rank`)."""
        import numpy as np
        owner = ((leaf >> 5) % self.world) if rank is None else rank
        vals = np.zeros((130, 2), dtype=np.uint64)
        path = ctypes.create_string_buffer(32 * 40)
        n = ctypes.c_int(0)
        rc = lib_gpu().vp_fri_open(self.ctx[owner], oracle, leaf, vals.ctypes.data, ctypes.cast(path, ctypes.c_void_p), len(path), ctypes.byref(n))
        if rc:
            return None
        return vals, [path.raw[32 * i:32 * i + 32] for i in range(n.value)]

    def close(self):
        for c in self.ctx:
            lib_gpu().vp_destroy(c)
        self.ctx = []


class ShardedCommitmentRank:
    """ONE rank of the Virgo commitment sharded over the ranks of a torch.distributed job (include/vpgpu.h: vp_pc_set_shard): a light
    context holding the input layer.  transport="rccl": the collectives (one all-to-all per committed oracle, one all-gather of tree nodes)
    run inside the entry points over an RCCL communicator attached to the context (needs one GPU per rank).  transport="host": no
    communicator — every call that returns VP_EXCHANGE has its pending collectives moved by torch.distributed on host buffers (gloo; the
    all-to-all as an all-gather of every rank's send side), which is how more ranks than GPUs are rehearsed and how the CPU suite's
    world-size-2 test drives it."""
    VP_EXCHANGE = 1

    def __init__(self, inputs, bit_length, rank, world, device=0, transport="rccl"):
        import numpy as np
        L = lib_gpu()
        for f in ("vp_shard_pending", "vp_shard_exchange_done"):
            getattr(L, f).argtypes = [ctypes.c_void_p] + ([ctypes.POINTER(ctypes.c_int)] if f == "vp_shard_pending" else [])
        L.vp_shard_exchange_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
        L.vp_shard_exchange_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.vp_shard_exchange_put.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.rank, self.world, self.n, self.transport = rank, world, bit_length, transport
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        self.ctx = ctypes.c_void_p()
        if L.vp_create(device, ctypes.byref(self.ctx)):
            raise RuntimeError("vp_create failed")
        self._chk(L.vp_pc_load_input(self.ctx, inputs.ctypes.data, inputs.shape[0], bit_length), "vp_pc_load_input")
        if transport == "rccl":
            import torch
            import torch.distributed as dist
            uid = ctypes.create_string_buffer(128)
            if rank == 0 and L.vp_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)):
                raise RuntimeError("vp_comm_unique_id failed (librccl.so.1 not found?)")
            t = torch.from_numpy(np.frombuffer(uid.raw, dtype=np.uint8).copy())
            if "nccl" in dist.get_backend():
                t = t.cuda()
            dist.broadcast(t, src=0)
            idb = ctypes.create_string_buffer(t.cpu().numpy().tobytes(), 128)
            self._chk(L.vp_comm_init(self.ctx, ctypes.cast(idb, ctypes.c_void_p), rank, world), "vp_comm_init")
        self._chk(L.vp_pc_set_shard(self.ctx, rank, world), "vp_pc_set_shard")

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, lib_gpu().vp_last_error(self.ctx).decode()))

    def comm_count(self):
        n = ctypes.c_int(0)
        self._chk(lib_gpu().vp_comm_count(self.ctx, ctypes.byref(n)), "vp_comm_count")
        return n.value

    def _exchange_host(self):
        import numpy as np
        import torch
        import torch.distributed as dist
        L = lib_gpu()
        n = ctypes.c_int(0)
        L.vp_shard_pending(self.ctx, ctypes.byref(n))
        for i in range(n.value):
            kind, nb = ctypes.c_int(0), ctypes.c_uint64(0)
            self._chk(L.vp_shard_exchange_info(self.ctx, i, ctypes.byref(kind), ctypes.byref(nb)), "vp_shard_exchange_info")
            b = int(nb.value)
            send = np.zeros(self.world * b if kind.value == 1 else b, dtype=np.uint8)
            self._chk(L.vp_shard_exchange_get(self.ctx, i, send.ctypes.data), "vp_shard_exchange_get")
            parts = [torch.zeros(send.shape[0], dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, torch.from_numpy(send))
            if kind.value == 1:        # all-to-all: from every peer p, the block it addressed to this rank
                recv = np.concatenate([parts[p].numpy()[self.rank * b:(self.rank + 1) * b] for p in range(self.world)])
            else:
                recv = np.concatenate([parts[p].numpy() for p in range(self.world)])
            recv = np.ascontiguousarray(recv)
            self._chk(L.vp_shard_exchange_put(self.ctx, i, recv.ctypes.data), "vp_shard_exchange_put")
        L.vp_shard_exchange_done(self.ctx)

    def _run(self, call, what):
        for _ in range(8):
            rc = call()
            self._chk(rc, what)
            if rc == 0:
                return
            if rc != self.VP_EXCHANGE or self.transport != "host":
                raise RuntimeError("%s: unexpected return %d" % (what, rc))
            self._exchange_host()
        raise RuntimeError("%s did not finish" % what)

    def device_ms(self):
        ms = ctypes.c_double(0)
        lib_gpu().vp_commit_stats(self.ctx, ctypes.byref(ms))
        return ms.value

    def commit_private(self):
        root = ctypes.create_string_buffer(32)
        self._run(lambda: lib_gpu().vp_commit_private(self.ctx, ctypes.cast(root, ctypes.c_void_p)), "vp_commit_private")
        return root.raw

    def commit_public(self, pub):
        import numpy as np
        pub = np.ascontiguousarray(pub, dtype=np.uint64)
        root = ctypes.create_string_buffer(32)
        inner, alls = np.zeros(2, np.uint64), np.zeros((65, 2), np.uint64)
        self._run(lambda: lib_gpu().vp_commit_public(self.ctx, pub.ctypes.data, pub.shape[0], inner.ctypes.data, alls.ctypes.data,
                                                     ctypes.cast(root, ctypes.c_void_p)), "vp_commit_public")
        return root.raw, inner.tobytes(), alls.tobytes()

    def fri_commit(self, r):
        import numpy as np
        r = np.ascontiguousarray(r, dtype=np.uint64)
        roots = ctypes.create_string_buffer(32 * r.shape[0])
        self._run(lambda: lib_gpu().vp_fri_commit(self.ctx, r.ctypes.data, r.shape[0], ctypes.cast(roots, ctypes.c_void_p)), "vp_fri_commit")
        fin = np.zeros((2048, 2), dtype=np.uint64)
        self._chk(lib_gpu().vp_fri_final(self.ctx, fin.ctypes.data), "vp_fri_final")
        return roots.raw, fin

    def close(self):
        if self.ctx:
            lib_gpu().vp_destroy(self.ctx)
            self.ctx = None


class Circuit:
    """layeredCircuit after subsetInit (host/circuit.hpp)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("circuit construction failed")
        self.h = handle

    @classmethod
    def from_pws(cls, path, blocks=1, seed=-1):
        err = ctypes.create_string_buffer(256)
        h = lib_host().vph_circuit_from_pws(os.fsencode(path), blocks, seed, err, len(err))
        if not h:
            raise RuntimeError("from_pws: " + err.value.decode())
        return cls(h)

    @classmethod
    def randomize(cls, layers, log_size, seed=-1):
        return cls(lib_host().vph_circuit_randomize(layers, log_size, seed))

    @classmethod
    def custom(cls, layer_sizes, ty, l, u, v, c_pairs, is_assert):
        """Arbitrary layered circuit from flat numpy arrays (see vphost.h)."""
        import numpy as np
        a = [np.ascontiguousarray(layer_sizes, np.uint64), np.ascontiguousarray(ty, np.int32), np.ascontiguousarray(l, np.int32),
             np.ascontiguousarray(u, np.uint64), np.ascontiguousarray(v, np.uint64), np.ascontiguousarray(c_pairs, np.uint64),
             np.ascontiguousarray(is_assert, np.uint8)]
        return cls(lib_host().vph_circuit_custom(len(a[0]), *[x.ctypes.data for x in a]))

    @property
    def layers(self):
        return lib_host().vph_circuit_layers(self.h)

    @property
    def gates(self):
        return lib_host().vph_circuit_gates(self.h)

    def layer_size(self, i):
        return lib_host().vph_circuit_layer_size(self.h, i)

    def layer_bitlen(self, i):
        return lib_host().vph_circuit_layer_bitlen(self.h, i)

    def hash(self):
        out = (ctypes.c_uint64 * 2)()
        lib_host().vph_circuit_hash(self.h, out)
        return "%016x%016x" % (out[0], out[1])

    def verify_transcript(self, transcript, skip_predicates=False):
        """Host verifier replay (no GPU): F::init(), draw the tape, check the GKR transcript."""
        buf = ctypes.create_string_buffer(bytes(transcript), len(transcript))
        return lib_host().vph_verify_transcript(self.h, ctypes.cast(buf, ctypes.c_void_p), len(transcript),
                                                1 if skip_predicates else 0) == 0

    def verify_fs(self, proof):
        """Verify a Fiat-Shamir proof (Session.prove_fs) on the host: no GPU, no tape."""
        buf = ctypes.create_string_buffer(bytes(proof), len(proof))
        return lib_host().vph_verify_fs(self.h, ctypes.cast(buf, ctypes.c_void_p), len(proof)) == 0

    def close(self):
        if self.h:
            lib_host().vph_circuit_free(self.h)
            self.h = None


_PUBLIC_OPTIONS = ("use_graph", "plan_autotune", "real_values", "real_pairs", "pc_tensor_pub", "persistent_rounds", "persistent_timeout_ms", "poll", "prefetch_round1",
                   "interactive_fast_init", "split_cost_percent", "debug")
# internal tuning knobs (csrc/vpgpu.hip, VpOpt): name -> the VP_* environment variable the library reads once per vp_create (tests / benches / A-B only)
_TUNING_ENV = {"gkr_path": "VP_GKR_PATH", "serial": "VP_GKR_SERIAL", "fuse_init": "VP_FUSE_INIT", "fuse_min_log": "VP_FUSE_MIN_LOG", "fuse_dot": "VP_FUSE_DOT",
               "drop_y": "VP_DROP_Y", "drop_y_round1": "VP_DROP_Y1", "seg_tiny": "VP_SEG_TINY", "sf_big_log": "VP_SF_BIG_LOG", "sf3b_grid": "VP_SF3B_GRID",
               "dot_blocks": "VP_DOT_BLOCKS", "plan_align": "VP_PLAN_ALIGN", "xcd_map": "VP_XCD_MAP", "round_fused_max": "VP_ROUND_FUSED_MAX",
               "kernel_copies": "VP_KERNEL_COPIES", "fold_branches": "VP_FOLD_BRANCHES", "ntt_scatter": "VP_NTT_SCATTER", "fuse_combine": "VP_FUSE_COMBINE",
               "graph_explicit": "VP_GRAPH_EXPLICIT", "ntt_r8": "VP_NTT_R8", "fri_vo_fused": "VP_FRI_VO_FUSED", "fuse_p2": "VP_FUSE_P2", "leaf_asm": "VP_LEAF_ASM",
               "fft_gkr_batched": "VP_FFT_GKR_BATCHED", "split_vu": "VP_SPLIT_VU", "fri_fold3": "VP_FRI_FOLD3", "sf3c": "VP_SF3C"}
VP_OPTIONS_ABI = 0x76700005


class Options(ctypes.Structure):
    """vp_options of include/vpgpu.h — what a caller can sensibly choose about how the library computes (never what).  Options() holds the shipped defaults.
    Keywords that are not fields of the struct but INTERNAL tuning knobs (the names of csrc/vpgpu.hip's VpOpt: fuse_combine, sf3b_grid, ntt_r8, gkr_path ...)
    are accepted too, for the tests: Session() sets the VP_* environment variable of each around its vp_create, which is the only way the library takes them."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("abi", ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in _PUBLIC_OPTIONS] + [("reserved", ctypes.c_int32 * 4)]

    def __init__(self, **kw):
        super().__init__()
        lib_gpu().vp_options_default(ctypes.byref(self))
        self.tuning = {}
        for k, v in kw.items():
            if k in _PUBLIC_OPTIONS:
                setattr(self, k, v)
            elif k in _TUNING_ENV:
                self.tuning[k] = v
            else:
                raise TypeError("unknown option " + k)

    def tuning_env(self):
        env = {}
        for k, v in getattr(self, "tuning", {}).items():
            if k == "gkr_path":
                v = {PATH_PLAN: "plan", PATH_LANES: "lanes", PATH_SIMPLE: "simple"}[v]
            elif k == "plan_align":
                v = {0: "", 1: "left", 2: "right"}[v]
            env[_TUNING_ENV[k]] = str(v)
        return env


PATH_PLAN, PATH_LANES, PATH_SIMPLE = 0, 1, 3


class Session:
    """One prover on one GPU: circuit resident in HBM, witness evaluated on the device."""

    def __init__(self, circuit, device=0, options=None):
        err = ctypes.create_string_buffer(512)
        self.circuit = circuit
        keep = {}
        for k, v in (options.tuning_env() if options is not None else {}).items():          # internal knobs: the library reads them from the environment at vp_create
            keep[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            self.h = lib_host().vph_session_create_opts(circuit.h, device, ctypes.byref(options) if options is not None else None, err, len(err))
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if not self.h:
            raise RuntimeError("Session: " + err.value.decode())
        self._cap = int(lib_host().vph_transcript_bytes(self.h)) + 4096
        self._buf = ctypes.create_string_buffer(self._cap)

    def gpu_ctx(self):
        """The vp_ctx* of this session (for direct C-ABI calls in tests)."""
        return ctypes.c_void_p(lib_host().vph_session_ctx(self.h))

    def set_profiling(self, level):
        lib_host().vph_set_profiling(self.h, level)

    def launch_stats(self):
        """Per-launch table of the last profiled call (set_profiling(1), then prove_gkr / commit_private / commit_public /
        fri_commit): list of dicts {kernel, step, workgroups, jobs, rounds, first_round, bytes, work, us}."""
        ctx = lib_host().vph_session_ctx(self.h)
        n = ctypes.c_int(0)
        lib_gpu().vp_get_launch_stats(ctx, None, 0, ctypes.byref(n))
        arr = (LaunchStat * max(1, n.value))()
        lib_gpu().vp_get_launch_stats(ctx, arr, n.value, ctypes.byref(n))
        out = []
        for i in range(n.value):
            e = arr[i]
            out.append({"kernel": lib_gpu().vp_kernel_name(e.kind).decode(), "step": e.step, "workgroups": e.workgroups, "jobs": e.jobs,
                        "rounds": e.rounds, "first_round": e.first_round, "bytes": e.bytes, "work": e.work, "us": e.us})
        return out

    def round_stats(self):
        """Interactive path, one dict per vp_round since the last prove_interactive() started: {phase, layer, round, how, tables, bytes, us}
        (include/vpgpu.h: vp_round_stat — algorithmic bytes of the round by SURVEY §8d and the wall time of the call)."""
        ctx = lib_host().vph_session_ctx(self.h)
        n = ctypes.c_int(0)
        lib_gpu().vp_get_round_stats(ctx, None, 0, ctypes.byref(n))
        arr = (RoundStat * max(1, n.value))()
        lib_gpu().vp_get_round_stats(ctx, arr, n.value, ctypes.byref(n))
        return [{"phase": e.phase, "layer": e.layer, "round": e.round, "how": e.how, "tables": e.tables, "bytes": e.bytes, "us": e.us}
                for e in (arr[i] for i in range(n.value))]

    def comm_count(self):
        """ncclCommCount of the communicator attached with attach_comm()."""
        n = ctypes.c_int(0)
        if lib_gpu().vp_comm_count(lib_host().vph_session_ctx(self.h), ctypes.byref(n)):
            raise RuntimeError("vp_comm_count failed")
        return n.value

    def layer_values(self, layer):
        import numpy as np
        n = self.circuit.layer_size(layer)
        out = np.zeros((n, 2), dtype=np.uint64)
        rc = lib_host().vph_layer_values(self.h, layer, out.ctypes.data, n)
        if rc:
            raise RuntimeError("layer_values failed")
        return out

    def _call(self, fn):
        # per-session scratch objects: the batched proof is sub-millisecond, so ctypes allocations per call would show up
        if not hasattr(self, "_n"):
            self._n = ctypes.c_uint64(0)
            self._res = VphResult()
            self._err = ctypes.create_string_buffer(512)
            self._bufp = ctypes.cast(self._buf, ctypes.c_void_p)
        n, res, err = self._n, self._res, self._err
        rc = fn(self.h, self._bufp, self._cap, ctypes.byref(n), ctypes.byref(res), err, 512)
        if rc < 0:
            raise RuntimeError("prover failed: " + err.value.decode())
        return ctypes.string_at(self._buf, n.value), res.as_dict(), rc

    def prove_interactive(self):
        """F::init() + verifier::verify(): returns (transcript bytes, stats, verified)."""
        tr, res, rc = self._call(lib_host().vph_prove_interactive)
        b = (ctypes.c_double * 3)()
        lib_host().vph_interactive_breakdown(self.h, b)
        res["init_sec"], res["round_sec"], res["finalize_sec"] = b[0], b[1], b[2]
        return tr, res, rc == 0

    def prove_fs(self):
        """Non-interactive GKR proof (Fiat-Shamir over SHA3-256): returns (proof bytes, stats, accepted by the built-in verifier)."""
        tr, res, rc = self._call(lib_host().vph_prove_fs)
        return tr, res, rc == 0

    def draw_tape(self):
        lib_host().vph_draw_tape(self.h)

    def prove_gkr(self):
        """One batched device pass from the attached tape: returns (transcript bytes, stats)."""
        tr, res, _ = self._call(lib_host().vph_prove_gkr)
        return tr, res

    def commit_device_ms(self):
        """Device milliseconds (HIP events) of the last commit_private / commit_public / fri_commit call."""
        return float(lib_host().vph_commit_device_ms(self.h))

    def tail_resumes(self):
        """vp_get_resident_resumes: relaunches of this session's resident round kernel on a saved phase."""
        n = ctypes.c_uint64(0)
        if lib_gpu().vp_get_resident_resumes(lib_host().vph_session_ctx(self.h), ctypes.byref(n)):
            raise RuntimeError("vp_get_resident_resumes failed")
        return int(n.value)

    def options_in_effect(self):
        """vp_get_options: the configuration this session runs with (after the plan tuner, once a proof has run)."""
        import types
        o = Options()
        ctx = lib_host().vph_session_ctx(self.h)
        if lib_gpu().vp_get_options(ctx, ctypes.byref(o)):
            raise RuntimeError("vp_get_options failed")
        eff = types.SimpleNamespace(**{k: getattr(o, k) for k in _PUBLIC_OPTIONS})
        v = ctypes.c_int32(0)
        for name in _TUNING_ENV:                               # the internal knobs, by name (vp_tuning_get)
            if lib_gpu().vp_tuning_get(ctx, name.encode(), ctypes.byref(v)) == 0:
                setattr(eff, name, v.value)
        return eff

    def set_shard(self, rank, world):
        """One proof over `world` GPUs: prove_gkr() then runs only the sumcheck chains dealt to `rank` and leaves the rest of the
        transcript zero; sum_transcripts() of all ranks' outputs (one all-reduce) is the proof.  world=1 undoes it."""
        if lib_host().vph_set_shard(self.h, rank, world):
            raise RuntimeError("set_shard(%d, %d) refused" % (rank, world))

    def attach_comm(self, rank, world):
        """RCCL communicator inside the library (include/vpgpu.h: vp_comm_unique_id / vp_comm_init): rank 0 makes the id, the default
        torch.distributed group only carries those 128 bytes (control plane).  From then on a sharded prove_gkr() all-reduces the
        transcript (and the export area of an index-split proof) on the device, inside the call, and returns the finished transcript on
        every rank — no torch tensor and no host bounce on the data path."""
        import numpy as np
        import torch
        import torch.distributed as dist
        L = lib_gpu()
        uid = ctypes.create_string_buffer(128)
        if rank == 0 and L.vp_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)):
            raise RuntimeError("vp_comm_unique_id failed (librccl.so.1 not found?)")
        t = torch.from_numpy(np.frombuffer(uid.raw, dtype=np.uint8).copy())
        if "nccl" in dist.get_backend():
            t = t.cuda()
        dist.broadcast(t, src=0)
        idb = ctypes.create_string_buffer(t.cpu().numpy().tobytes(), 128)
        ctx = lib_host().vph_session_ctx(self.h)
        if L.vp_comm_init(ctx, ctypes.cast(idb, ctypes.c_void_p), rank, world):
            raise RuntimeError("vp_comm_init failed: " + (L.vp_last_error(ctx) or b"").decode())

    def set_shard_split(self, min_log):
        """On top of set_shard: tables of at least 2^(log2 W + min_log) entries are also split by index over the ranks (include/vpgpu.h:
        vp_set_shard_split); prove_gkr() then returns the rank's partial transcript followed by its export area, and
        shard_finish(sum_transcripts(parts)) is the proof.  min_log=0 switches it off."""
        ctx = lib_host().vph_session_ctx(self.h)
        L = lib_gpu()
        if L.vp_set_shard_split(ctx, int(min_log)):
            raise RuntimeError("set_shard_split(%d) refused" % min_log)
        n = ctypes.c_uint64(0)
        L.vp_gkr_sizes(ctx, None, ctypes.byref(n))
        if int(n.value) + 4096 > self._cap:
            self._cap = int(n.value) + 4096
            self._buf = ctypes.create_string_buffer(self._cap)
            if hasattr(self, "_n"):
                self._bufp = ctypes.cast(self._buf, ctypes.c_void_p)

    def shard_vu_partials(self):
        """Index-split proof without a communicator: this rank's partial inner products for the V_u of the split phase-2 chains (include/vpgpu.h:
        vp_shard_vu_partials) as a numpy array of shape (n, 2), uint64; n may be 0."""
        import numpy as np
        cap = max(1, int(self.circuit.layers))
        out = np.zeros((cap, 2), dtype=np.uint64)
        n = lib_host().vph_shard_vu_partials(self.h, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("shard_vu_partials refused")
        return out[:n].copy()

    def shard_vu_set(self, sums):
        """... and their u64 sum over the ranks, for this rank's next prove_gkr() (vp_shard_vu_set)."""
        import numpy as np
        a = np.ascontiguousarray(sums, dtype=np.uint64).reshape(-1, 2)
        if lib_host().vph_shard_vu_set(self.h, a.ctypes.data, int(a.shape[0])):
            raise RuntimeError("shard_vu_set refused")

    def shard_finish(self, summed):
        """The transcript of an index-split proof from the u64 sum of the ranks' prove_gkr() outputs (vp_shard_finish)."""
        ctx = lib_host().vph_session_ctx(self.h)
        buf = ctypes.create_string_buffer(bytes(summed), len(summed))
        n = ctypes.c_uint64(0)
        if lib_gpu().vp_shard_finish(ctx, buf, len(summed), ctypes.byref(n)):
            raise RuntimeError("shard_finish refused")
        return buf.raw[:int(n.value)]

    def shard_chains(self):
        """(owner rank, cost estimate) per sumcheck chain of the proof, in plan order (include/vpgpu.h: vp_shard_chains)."""
        import numpy as np
        cap = 3 * self.circuit.layers + 1
        owner = np.zeros(cap, dtype=np.int32)
        cost = np.zeros(cap, dtype=np.float64)
        n = lib_host().vph_shard_chains(self.h, owner.ctypes.data, cost.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("shard_chains failed")
        return owner[:n].copy(), cost[:n].copy()

    def commit_private(self):
        """prover::commit_private(): (32-byte Merkle root, device milliseconds)."""
        root = ctypes.create_string_buffer(32)
        ms = ctypes.c_double(0)
        err = ctypes.create_string_buffer(512)
        rc = lib_host().vph_commit_private(self.h, ctypes.cast(root, ctypes.c_void_p), ctypes.byref(ms), err, len(err))
        if rc:
            raise RuntimeError("commit_private failed: " + err.value.decode())
        return root.raw, ms.value

    def commit_public(self, pub):
        """prover::commit_public on a (2^n, 2) uint64 array: (root_h, input_0 bytes, all_sum bytes, device ms)."""
        import numpy as np
        pub = np.ascontiguousarray(pub, dtype=np.uint64)
        out = ctypes.create_string_buffer(32 + 16 + 65 * 16)
        ms = ctypes.c_double(0)
        err = ctypes.create_string_buffer(512)
        rc = lib_host().vph_commit_public(self.h, pub.ctypes.data, pub.shape[0], ctypes.cast(out, ctypes.c_void_p), ctypes.byref(ms),
                                          err, len(err))
        if rc:
            raise RuntimeError("commit_public failed: " + err.value.decode())
        return out.raw[:32], out.raw[32:48], out.raw[48:], ms.value

    def prove_full(self, batched=False):
        """commit_private + GKR + commit_public: the full golden transcript layout; returns (bytes, verified)."""
        cap = self._cap + 32 + 32 + 16 + 65 * 16
        buf = ctypes.create_string_buffer(cap)
        n = ctypes.c_uint64(0)
        err = ctypes.create_string_buffer(512)
        rc = lib_host().vph_prove_full(self.h, ctypes.cast(buf, ctypes.c_void_p), cap, ctypes.byref(n), 1 if batched else 0, err, len(err))
        if rc < 0:
            raise RuntimeError("prove_full failed: " + err.value.decode())
        return buf.raw[: n.value], rc == 0

    def commit_public_eq(self, point):
        """prover::commit_public on pub = eq(point, .) built on the device (vp_commit_public_eq): (root_h, input_0 bytes, all_sum bytes, device ms)."""
        import numpy as np
        point = np.ascontiguousarray(point, dtype=np.uint64)
        out = ctypes.create_string_buffer(32 + 16 + 65 * 16)
        ctx = lib_host().vph_session_ctx(self.h)
        base = ctypes.addressof(out)
        rc = lib_gpu().vp_commit_public_eq(ctx, point.ctypes.data, point.shape[0], base + 32, base + 48, base)
        if rc:
            raise RuntimeError("vp_commit_public_eq failed: %d %s" % (rc, (lib_gpu().vp_last_error(ctx) or b"").decode()))
        return out.raw[:32], out.raw[32:48], out.raw[48:], self.commit_device_ms()

    def warm(self):
        """vp_warm(VP_WARM_COMMITMENT | VP_WARM_FFT_GKR): the commitment's tables, buffers and pinned staging (and fft_gkr's arrays) exist before the first prover call (include/vpgpu.h)."""
        ctx = lib_host().vph_session_ctx(self.h)
        if lib_gpu().vp_warm(ctx, 3):
            raise RuntimeError("vp_warm failed: " + (lib_gpu().vp_last_error(ctx) or b"").decode())

    def draw_protocol_tape(self):
        """F::init() + every draw of verifier::verify() up front, in its order: GKR, fft_gkr, FRI fold challenges."""
        if lib_host().vph_draw_protocol_tape(self.h):
            raise RuntimeError("draw_protocol_tape: the commitment needs an input layer of at least 2^7 wires")

    def prove_protocol(self, deferred=False, queue_next=False):
        """The prover side of the complete protocol in one pass (no verifier work): commit_private -> batched GKR -> commit_public on
        eq(r_liu, .) -> fft_gkr -> FRI commit phase.  Returns (transcript in the golden layout, FRI roots bytes, final codeword (2048, 2),
        seconds dict {total, commit_private, gkr, commit_public, fft_gkr, fri_commit}).
        deferred: the calls are queued back to back and collected at the end (vp_set_deferred: no idle device between them); the per-call seconds are then
        device times.  queue_next: the pass also queues the next pass's commit_private behind its own folds (vphost.h, VPH_PASS_QUEUE_NEXT): for a
        session that proves back to back; this pass's commitment cannot be opened afterwards."""
        import numpy as np
        if not hasattr(self, "_pp"):
            cap = self._cap + 32 + 32 + 16 + 65 * 16
            self._pp = (ctypes.create_string_buffer(cap), cap, ctypes.create_string_buffer(32 * 32), np.zeros((2048, 2), dtype=np.uint64),
                        (ctypes.c_double * 6)(), ctypes.c_uint64(0), ctypes.create_string_buffer(512))
        buf, cap, roots, fin, sec, n, err = self._pp
        rc = lib_host().vph_prove_protocol_ex(self.h, ctypes.cast(buf, ctypes.c_void_p), cap, ctypes.byref(n), ctypes.cast(roots, ctypes.c_void_p), len(roots),
                                              fin.ctypes.data, sec, (1 if deferred or queue_next else 0) | (2 if queue_next else 0), err, len(err))
        if rc:
            raise RuntimeError("prove_protocol failed: " + err.value.decode())
        st = self.circuit.layer_bitlen(0) - 6
        return (buf.raw[: n.value], roots.raw[:32 * st], fin.copy(),
                {"total": sec[0], "commit_private": sec[1], "gkr": sec[2], "commit_public": sec[3], "fft_gkr": sec[4], "fri_commit": sec[5]})

    def prove_and_verify_full(self, reps=33):
        """The complete protocol incl. commitment verification: (transcript bytes, accepted, times dict)."""
        cap = self._cap + 32 + 32 + 16 + 65 * 16
        buf = ctypes.create_string_buffer(cap)
        n = ctypes.c_uint64(0)
        t = [ctypes.c_double(0) for _ in range(3)]
        err = ctypes.create_string_buffer(512)
        rc = lib_host().vph_prove_and_verify_full(self.h, reps, ctypes.cast(buf, ctypes.c_void_p), cap, ctypes.byref(n),
                                                  ctypes.byref(t[0]), ctypes.byref(t[1]), ctypes.byref(t[2]), err, len(err))
        if rc < 0:
            raise RuntimeError("prove_and_verify_full failed: " + err.value.decode())
        pt = (ctypes.c_double * 3)()
        lib_host().vph_last_pc_times(self.h, pt)
        return buf.raw[: n.value], rc == 0, {"gkr_prove_sec": t[0].value, "pc_prove_sec": t[1].value, "verify_sec": t[2].value,
                                            "pc_fft_gkr_sec": pt[1], "pc_query_answer_sec": pt[2]}

    def last_fft_gkr(self):
        """Messages of fft_gkr (vp_fft_gkr layout) of the last prove_and_verify_full(), as bytes."""
        import numpy as np
        out = np.zeros((4096, 2), dtype=np.uint64)
        n = lib_host().vph_last_fft_gkr(self.h, out.ctypes.data, out.shape[0])
        if n < 0:
            raise RuntimeError("no complete-protocol run on this session yet")
        return out[:n].tobytes()

    def last_fri(self):
        """FRI commit phase of the last prove_and_verify_full(): (roots bytes, final codeword (2048, 2), challenges (steps, 2))."""
        import numpy as np
        roots = ctypes.create_string_buffer(32 * 32)
        fin = np.zeros((2048, 2), dtype=np.uint64)
        r = np.zeros((32, 2), dtype=np.uint64)
        st = lib_host().vph_last_fri(self.h, ctypes.cast(roots, ctypes.c_void_p), len(roots), fin.ctypes.data, r.ctypes.data)
        if st < 0:
            raise RuntimeError("no complete-protocol run on this session yet")
        return roots.raw[:32 * st], fin, r[:st].copy()

    def last_point(self):
        """r_liu after the last Liu sumcheck of the last prove_full / prove_and_verify_full, (n, 2) uint64."""
        import numpy as np
        out = np.zeros((64, 2), dtype=np.uint64)
        n = lib_host().vph_last_point(self.h, out.ctypes.data, 64)
        if n < 0:
            raise RuntimeError("no complete-protocol run on this session yet")
        return out[:n].copy()

    def eq_table(self, point):
        """eq(point, .) over 2^n entries on the device (vp_test_beta): the protocol's public vector for point = last_point()."""
        import numpy as np
        point = np.ascontiguousarray(point, dtype=np.uint64)
        n = point.shape[0]
        one = np.array([1, 0], dtype=np.uint64)
        out = np.zeros((1 << n, 2), dtype=np.uint64)
        rc = lib_gpu().vp_test_beta(lib_host().vph_session_ctx(self.h), point.ctypes.data, n, one.ctypes.data, out.ctypes.data)
        if rc:
            raise RuntimeError("vp_test_beta failed")
        return out

    def fri_commit(self, r, batched=True):
        """FRI commit phase with the given fold challenges ((steps, 2) uint64): (roots bytes, final codeword (2048, 2)).
        batched: all steps in one device pass (vp_fri_commit); otherwise one vp_fri_step per challenge."""
        import numpy as np
        r = np.ascontiguousarray(r, dtype=np.uint64)
        roots = ctypes.create_string_buffer(32 * r.shape[0])
        fin = np.zeros((2048, 2), dtype=np.uint64)
        err = ctypes.create_string_buffer(512)
        fn = lib_host().vph_fri_commit_batched if batched else lib_host().vph_fri_commit
        rc = fn(self.h, r.ctypes.data, r.shape[0], ctypes.cast(roots, ctypes.c_void_p), fin.ctypes.data, err, len(err))
        if rc:
            raise RuntimeError("fri_commit failed: " + err.value.decode())
        return roots.raw, fin

    def check(self, transcript, skip_predicates=False, device_predicates=False):
        """Replay the transcript through the host verifier: (accepted, seconds).  device_predicates: the O(|C|) wiring-predicate
        loops (verifier.cpp:50-113) run on the GPU (vp_predicates) instead of the host."""
        sec = ctypes.c_double(0)
        buf = ctypes.create_string_buffer(bytes(transcript), len(transcript))
        rc = lib_host().vph_check(self.h, ctypes.cast(buf, ctypes.c_void_p), len(transcript),
                                  (1 if skip_predicates else 0) | (2 if device_predicates else 0), ctypes.byref(sec))
        return rc == 0, sec.value

    def close(self):
        if self.h:
            lib_host().vph_session_free(self.h)
            self.h = None
