"""INTEGRATION.md's claim, compiled: the reference's UNMODIFIED main.cpp / verifier.cpp / circuit.cpp / polynomial.cpp / utils.cpp and
lib/virgo (every file but fri.cpp) link against oracle/integration/prover_vpgpu.cpp (the forwarding bodies for src/prover.cpp, written
against the reference's own src/prover.h), oracle/integration/fri_vpgpu.cpp (the forwarding bodies for lib/virgo/src/fri.cpp, written
against the reference's own fri.h) and libvpgpu.so.  Test infrastructure only (oracle/Makefile target `integration`); the GPU run of the
same binary is tests/test_gpu_parity.py::test_reference_binary_drives_the_device_prover."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_run_vpgpu")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_unmodified_reference_links_against_the_forwarding_prover():
    import vp_loader
    vp_loader.load().build()                                        # libvpgpu.so first: the binary links against it
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "integration"], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN)
    # every prover method verifier.cpp calls is defined by the forwarding file, none by the reference's own prover.cpp
    syms = subprocess.run(["nm", "-C", "--defined-only", os.path.join(ROOT, "oracle", "_ref", "integration_prover.o")],
                          check=True, stdout=subprocess.PIPE, text=True).stdout
    for m in ("prover::prover(", "prover::evaluate()", "prover::Vres(", "prover::sumcheckInitAll(", "prover::sumcheckInit()",
              "prover::sumcheckInitPhase1(", "prover::sumcheckInitPhase2()", "prover::sumcheckInitLiu(", "prover::sumcheckUpdatePhase1(",
              "prover::sumcheckUpdatePhase2(", "prover::sumcheckLiuUpdate(", "prover::sumcheckFinalize1(", "prover::sumcheckFinalize2(",
              "prover::sumcheckLiuFinalize(", "prover::proveTime()", "prover::proofSize()", "prover::commit_private()", "prover::commit_public("):
        assert m in syms, m
    und = subprocess.run(["nm", "-C", "--undefined-only", os.path.join(ROOT, "oracle", "_ref", "plain_verifier.o")],
                         check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "prover::sumcheckUpdatePhase1(" in und and "prover::Vres(" in und          # the unmodified verifier asks for exactly these
    # the commitment seam: every fri:: function lib/virgo's UNMODIFIED verifier (vpd_verifier.cpp) calls is defined by the forwarding file ...
    fsyms = subprocess.run(["nm", "-C", "--defined-only", os.path.join(ROOT, "oracle", "_ref", "integration_fri.o")],
                           check=True, stdout=subprocess.PIPE, text=True).stdout
    for m in ("virgo::fri::request_init_commit(", "virgo::fri::commit_phase_step(", "virgo::fri::commit_phase_final()",
              "virgo::fri::request_init_value_with_merkle(", "virgo::fri::request_step_commit(", "virgo::fri::cpd",
              "virgo::fri::log_current_witness_size_per_slice"):
        assert m in fsyms, m
    vund = subprocess.run(["nm", "-C", "--undefined-only", os.path.join(ROOT, "oracle", "_ref", "virgo_vpd_verifier.o")],
                          check=True, stdout=subprocess.PIPE, text=True).stdout
    for m in ("virgo::fri::commit_phase_step(", "virgo::fri::commit_phase_final()", "virgo::fri::request_init_value_with_merkle(",
              "virgo::fri::request_step_commit(", "virgo::fri::cpd"):
        assert m in vund, m                                                            # ... which is exactly what that verifier asks for
    gsyms = subprocess.run(["nm", "-C", "--defined-only", os.path.join(ROOT, "oracle", "_ref", "integration_fftgkr.o")],
                           check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "virgo::fft_circuit_gkr::fft_gkr(int, double&, int&, double&)" in gsyms and "virgo::fft_circuit_gkr::fft_gkr(" in vund
    # ... and the reference's own fri.cpp is not in the binary (its file-static helper has no other definition)
    allsyms = subprocess.run(["nm", "-C", BIN], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "merkle_tree_consistency_check" not in allsyms
    assert "virgo::fft_circuit_gkr::engage_gkr" not in allsyms and "virgo::fft_circuit_gkr::build_circuit" not in allsyms   # nor its fft_circuit_GKR.cpp
    dyn = subprocess.run(["nm", "-D", "--undefined-only", BIN], check=True, stdout=subprocess.PIPE, text=True).stdout
    for f in ("vp_create", "vp_circuit_upload", "vp_evaluate", "vp_vres", "vp_phase1_init", "vp_phase2_init", "vp_liu_init", "vp_round", "vp_finalize",
              "vp_commit_private", "vp_commit_public", "vp_fri_step", "vp_fri_final", "vp_fri_open", "vp_fft_gkr"):
        assert f in dyn, f                                                             # ... and they reach the C ABI of libvpgpu.so
    # the block-count variant (oracle/integration/blocks_main.cpp replaces main() only): same seam, same absence of the reference's prover side
    blk = os.path.join(ROOT, "oracle", "_ref", "ref_run_vpgpu_blocks")
    assert os.path.exists(blk)
    bsyms = subprocess.run(["nm", "-C", blk], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "merkle_tree_consistency_check" not in bsyms and "virgo::fft_circuit_gkr::engage_gkr" not in bsyms and "DAG_to_layered()" in bsyms


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ref_run_vpgpu not built")
def test_reference_binary_without_a_gpu_fails_loudly(pws_path):
    """No CPU fallback: on a box without a device the binary stops at vp_create."""
    from conftest import gpu_count
    if gpu_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([BIN, str(pws_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and "vp_create failed" in r.stderr
