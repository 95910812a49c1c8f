"""Sanitizer leg of the CPU build ("ASan/UBSan CPU targets").  GPU AddressSanitizer is not available on this pool, so the
host-side code is what can be — and is — run under -fsanitize=address,undefined: the oracle (make -C oracle asan) and the host library's
sources (VP_HOST_SANITIZE build below), each as a stand-alone program that walks the code paths the parity tests use.  A sanitizer
report makes the program exit non-zero (-fno-sanitize-recover / ASan's default abort)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_oracle_under_asan_ubsan(pws_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "oracle_asan"), pws_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "oracle_asan ok" in r.stdout, (r.returncode, r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]


def test_host_library_under_asan_ubsan(vp, pws_path):
    """virgo-plus_amd/host/*.cpp (loader, levelisation, subsetInit, replicated builder, verifier replay, Fiat-Shamir verifier) built with the
    sanitizers and run without a GPU.  libvpgpu.so is linked (the prover class forwards to it) but never initialised."""
    out_dir = os.path.join(ROOT, "tests", "sanitize", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "host_asan")
    host = os.path.join(ROOT, "virgo-plus_amd", "host")
    csrc = os.path.join(ROOT, "virgo-plus_amd", "csrc")
    src = [os.path.join(host, f) for f in ("circuit.cpp", "prover.cpp", "verifier.cpp", "vphost.cpp")] + [os.path.join(ROOT, "tests", "sanitize", "host_main.cpp")]
    deps = src + [os.path.join(host, f) for f in os.listdir(host) if f.endswith((".hpp", ".h"))]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-pthread"] + SAN + ["-o", exe] + src +
                       ["-L" + csrc, "-lvpgpu", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    args = [exe, pws_path, os.path.join(GOLDEN, "transcript_sha256_x1.bin"), os.path.join(GOLDEN, "transcript_randomize_8_12.bin"),
            os.path.join(GOLDEN, "fs_proof_randomize_6_8_seed5.bin")]
    # the ROCm runtime libraries that libvpgpu.so pulls in are not instrumented and keep process-lifetime allocations: leaks are checked
    # for the oracle above (pure C++), here ASan's memory-error and UBSan's checks are what counts
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:exitcode=99")
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "host_asan ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
