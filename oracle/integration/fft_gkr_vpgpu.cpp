// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// The third forwarding file of INTEGRATION.md: what a reference maintainer puts in place of lib/virgo/src/fft_circuit_GKR.cpp so that
// fft_circuit_gkr::fft_gkr — the self-contained GKR the UNMODIFIED verify_poly_commitment runs (vpd_verifier.cpp:92) and whose prover
// time it adds to the reported commitment prove time (:94) — has its PROVER on the MI355X.  Same signature and outputs (v_time, proof
// size, p_time; "Error, fft gkr failed" on stderr), same F::random() draws in the same order (fft_circuit_GKR.cpp:840, :84, :789-790,
// :275-276, :394-395, :563-564, :763-764), so the FRI challenges verify_poly_commitment draws next are the CPU reference's.  The draws are
// transcript-independent, so the tape is drawn first and handed to vp_fft_gkr in one piece; the checks of the function's embedded
// verifier run here on the returned messages, in the reference's own field type (virgo-plus_amd/host/fft_gkr_verify.hpp).
#include "fft_circuit_GKR.h"        // the reference's, -I$(REF)/lib/virgo/src
#include "fieldElement.hpp"
#include "vpgpu_glue.h"
#include <fft_gkr_verify.hpp>       // -I<repo>/virgo-plus_amd/host
#include <chrono>
#include <cstdlib>
#include <vector>

namespace virgo { namespace fft_circuit_gkr {

int fft_gkr(int lg_size, double &vt, int &ps, double &pt) {
    static_assert(sizeof(fieldElement) == sizeof(vp_F), "virgo::fieldElement is two u64 limbs");
    uint64_t n_tape = 0, n_msgs = 0;
    vpi_must(vp_fft_gkr_sizes(lg_size, &n_tape, &n_msgs), "vp_fft_gkr_sizes");
    std::vector<fieldElement> tape(n_tape), msgs(n_msgs);
    for (auto &x : tape) x = fieldElement::random();
    const auto t0 = std::chrono::high_resolution_clock::now();
    {
        vpi_rand_guard guard("vp_fft_gkr");
        vpi_stopwatch sw(&g_vpi_sec.fft_gkr);
        uint64_t written = 0;
        vpi_must(vp_fft_gkr(vpi_ctx(), lg_size, reinterpret_cast<const vp_F *>(tape.data()), n_tape, reinterpret_cast<vp_F *>(msgs.data()), n_msgs, &written),
                 "vp_fft_gkr");
        if (written != n_msgs) { fprintf(stderr, "vpgpu glue: fft_gkr message count\n"); exit(EXIT_FAILURE); }
    }
    ++g_vpi_count.fft_gkr;
    const auto t1 = std::chrono::high_resolution_clock::now();
    const bool ok = vph::fft_gkr_check<fieldElement>(lg_size, tape.data(), tape.size(), msgs.data(), msgs.size(), fieldElement::getRootOfUnity(lg_size).inv());
    const auto t2 = std::chrono::high_resolution_clock::now();
    if (!ok) fprintf(stderr, "Error, fft gkr failed\n");                       // fft_circuit_GKR.cpp:843-844
    if (FILE *f = vpi_dump_fft_file()) { fwrite(msgs.data(), 16, msgs.size(), f); fflush(f); }
    pt = std::chrono::duration<double>(t1 - t0).count();
    vt = std::chrono::duration<double>(t2 - t1).count();
    // proof-size accounting of the reference: one quadratic_poly (3 F) per round (:260, :399, :560, :616) + extension_gkr's lg (lg + 1) / 2 (:770-779)
    const int polys = (lg_size + 6) + lg_size + 2 * lg_size * lg_size + lg_size * (lg_size + 1) / 2;
    ps = polys * 3 * (int) sizeof(fieldElement);
    return 0;
}

} }
