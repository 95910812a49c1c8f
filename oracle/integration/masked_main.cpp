// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// lib/virgo's commitment with NON-ZERO masks, prover side on the device, decided by the reference's OWN verifier (round 6): the reference's
// prover / verifier pair never passes a mask (src/prover.cpp:526, src/verifier.cpp:375-377), so — exactly as oracle/ref_driver.cpp's --pc-masked
// mode does for the CPU reference — the commitment is driven directly:
//     vp_commit_private_masked / vp_commit_public_masked          in place of poly_commit_prover::commit_private_array / commit_public_array
//                                                                 (lib/virgo/src/poly_commit.h:41-124, 126-349)
//     poly_commit_verifier::verify_poly_commitment(all_sum, n, processed, pub_mask, ...)       (vpd_verifier.cpp:76-328, UNMODIFIED object)
// with namespace virgo::fri and fft_gkr coming from INTEGRATION.md's forwarding files: 33 random queries — Merkle paths of l, h and every FRI
// level, the first-round consistency of all 65 slices (the mask slice against the public mask's polynomial and all_sum[64]), the later rounds,
// both final codewords — all served from HBM.
//   ref_run_vpgpu_masked IN          IN as for ref_run --pc-masked: i32 n, i32 m, values[2^n], pub[2^n], pri_mask[m], pub_mask[m]
// prints `pc-masked (device) verify_poly_commitment ACCEPT|REJECT`; exit code 0 = accepted.
#include "poly_commit.h"            // the reference's, -I$(REF)/lib/virgo/src
#include "RS_polynomial.h"
#include "vpgpu_glue.h"
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    using namespace virgo;
    if (argc != 2) { fprintf(stderr, "usage: %s IN\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int n = 0, m = 0;
    if (fread(&n, 4, 1, f) != 1 || fread(&m, 4, 1, f) != 1 || n < 7 || n > 24 || m < 1) return 2;
    auto rd = [&](fieldElement *dst, size_t cnt) { for (size_t i = 0; i < cnt; ++i) { unsigned long long w[2]; if (fread(w, 8, 2, f) != 2) return false; dst[i].real = w[0]; dst[i].img = w[1]; } return true; };
    std::vector<fieldElement> values((size_t) 1 << n), pub((size_t) 1 << n), pri(m), pubm(m);
    if (!rd(values.data(), values.size()) || !rd(pub.data(), pub.size()) || !rd(pri.data(), m) || !rd(pubm.data(), m)) return 2;
    fclose(f);
    fieldElement::init();
    const int slice_size = 1 << (n + rs_code_rate - log_slice_number);
    init_scratch_pad(slice_size);                                   // the host transforms of the verifier side (RS_polynomial.cpp:9-16); commit_private_array does this at :72
    vp_ctx *ctx = vpi_ctx_standalone();
    __hhash_digest root_l, root_h;
    std::vector<fieldElement> all_sum(slice_number + 1);
    fieldElement inner;
    {
        vpi_rand_guard guard("masked commitment");
        vpi_must(vp_pc_load_input(ctx, reinterpret_cast<const vp_F *>(values.data()), (uint64_t) values.size(), n), "vp_pc_load_input");
        vpi_must(vp_commit_private_masked(ctx, reinterpret_cast<const vp_F *>(pri.data()), (uint64_t) m, reinterpret_cast<uint8_t *>(&root_l)), "vp_commit_private_masked");
        vpi_oracle_committed(0, n, reinterpret_cast<const unsigned char *>(&root_l));
        vpi_must(vp_commit_public_masked(ctx, reinterpret_cast<const vp_F *>(pub.data()), (uint64_t) pub.size(), reinterpret_cast<const vp_F *>(pubm.data()), (uint64_t) m,
                                         reinterpret_cast<vp_F *>(&inner), reinterpret_cast<vp_F *>(all_sum.data()), reinterpret_cast<uint8_t *>(&root_h)), "vp_commit_public_masked");
        vpi_oracle_committed(1, n, reinterpret_cast<const unsigned char *>(&root_h));
    }
    // what commit_public_array leaves in the caller's vector (poly_commit.h:55-63,138-141): the public mask padded with zeros to slice_size / mask_position_gap
    long long gap = slice_size / m;
    for (int j = 0; j < 64; ++j) if ((1LL << j) <= gap && (1LL << (j + 1)) > gap) { gap = 1LL << j; break; }
    while ((long long) pubm.size() < slice_size / gap) pubm.push_back(fieldElement(0));
    // `processed` as src/verifier.cpp:346-358,371: the public vector's slices in coefficient form
    std::vector<fieldElement> processed((size_t) 1 << n);
    const int cs = 1 << (n - log_slice_number);
    for (int i = 0; i < slice_number; ++i)
        inverse_fast_fourier_transform(pub.data() + (size_t) i * cs, cs, cs, fieldElement::getRootOfUnity(n - log_slice_number), processed.data() + (size_t) i * cs);
    poly_commit::poly_commit_prover pp;                              // only its commit_phase (vpd_verifier.cpp:44-74: the loop over fri::commit_phase_step) is used
    poly_commit::poly_commit_verifier pv;
    pv.p = &pp;
    double vt = 0, pt = 0; int ps = 0;
    const bool ok = pv.verify_poly_commitment(all_sum.data(), n, processed.data(), pubm, vt, ps, pt, root_l, root_h);
    fprintf(stdout, "pc-masked (device) n %d mask %d mask_position_gap %lld verify_poly_commitment %s proof_bytes %d\n", n, m, gap, ok ? "ACCEPT" : "REJECT", ps);
    return ok ? 0 : 1;
}
