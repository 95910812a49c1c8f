// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// INTEGRATION.md made complete and compilable: the bodies a reference maintainer would put in src/prover.cpp so that the UNMODIFIED
// reference (src/main.cpp, src/verifier.cpp, src/circuit.cpp, src/polynomial.cpp, src/utils.cpp, lib/virgo/src/*.cpp) drives the
// MI355X library through include/vpgpu.h.  oracle/Makefile (target `integration`) compiles this file against the reference's own
// src/prover.h (the class declaration is used as it is: the device context lives in a file-static, the reference has one prover per
// process) and links everything with -lvpgpu into oracle/_ref/ref_run_vpgpu:
//   * on the CPU box that link is the proof that `class prover`'s surface as verifier.cpp uses it is served by the C ABI
//     (tests/test_integration_link.py);
//   * on the GPU box the binary runs the reference's own main() on data/SHA256_64.pws: every sumcheck message comes from the
//     device, the reference verifier checks it and prints "Verification pass" (tests/test_gpu_parity.py).
// The polynomial commitment of this binary stays on lib/virgo's CPU prover (poly_prover, a public member the verifier hands to
// lib/virgo's verifier, vpd_verifier.cpp): forwarding it needs the hooks inside lib/virgo that INTEGRATION.md lists, not only prover.cpp.
#include "prover.h"                 // the reference's, -I$(REF)/src
#include <vpgpu.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

static inline const vp_F *cF(const F *p) { return reinterpret_cast<const vp_F *>(p); }
static inline vp_F *mF(F *p) { return reinterpret_cast<vp_F *>(p); }
static vp_ctx *g_ctx = nullptr;
static void must(int rc, const char *what) {
    if (rc == VP_OK) return;
    fprintf(stderr, "vpgpu: %s failed (%d): %s\n", what, rc, g_ctx ? vp_last_error(g_ctx) : "no context");
    exit(EXIT_FAILURE);
}
static_assert(sizeof(F) == sizeof(vp_F), "virgo::fieldElement is two u64 limbs (fieldElement.hpp:96-97)");

prover::prover(const layeredCircuit &cir) : C(cir) {              // src/prover.cpp:14
    proof_size = 0;
    must(vp_create(0, &g_ctx), "vp_create");
    const int n = C.size;
    struct Flat {
        std::vector<uint8_t> ty, as; std::vector<int32_t> l; std::vector<uint32_t> u, v, lv; std::vector<vp_F> c;
        std::vector<uint64_t> dsz; std::vector<int32_t> dbl; std::vector<std::vector<uint32_t>> did; std::vector<const uint32_t *> dptr;
    };
    std::vector<Flat> flat(n);
    std::vector<vp_layer_desc> desc(n);
    for (int i = 0; i < n; ++i) {
        const layer &L = C.circuit[i];
        Flat &f = flat[i];
        const u64 m = L.size;
        f.ty.resize(m); f.as.resize(m); f.l.resize(m); f.u.resize(m); f.v.resize(m); f.lv.resize(m); f.c.resize(m);
        bool any_c = false, any_as = false;
        for (u64 g = 0; g < m; ++g) {
            const gate &G = L.gates[g];
            f.ty[g] = (uint8_t) G.ty; f.l[g] = G.l;
            f.u[g] = i == 0 ? 0u : (uint32_t) G.u; f.v[g] = (uint32_t) G.v; f.lv[g] = (uint32_t) G.lv;
            f.as[g] = G.is_assert ? 1 : 0; any_as |= G.is_assert;
            f.c[g].real = G.c.real; f.c[g].img = G.c.img;
            any_c |= (G.ty == gateType::Addc || G.ty == gateType::Mulc);
        }
        f.dsz.assign(L.dadSize.begin(), L.dadSize.end());
        f.dbl.assign(L.dadBitLength.begin(), L.dadBitLength.end());
        f.did.resize(L.dadId.size()); f.dptr.resize(L.dadId.size());
        for (size_t j = 0; j < L.dadId.size(); ++j) { f.did[j].assign(L.dadId[j].begin(), L.dadId[j].end()); f.dptr[j] = f.did[j].data(); }
        vp_layer_desc &d = desc[i];
        d.size = m; d.bit_length = L.bitLength;
        d.ty = f.ty.data(); d.l = f.l.data(); d.u = f.u.data(); d.v = f.v.data(); d.lv = f.lv.data();
        d.c = any_c ? f.c.data() : nullptr; d.is_assert = any_as ? f.as.data() : nullptr;
        d.dad_size = f.dsz.data(); d.dad_bitlen = f.dbl.data(); d.dad_id = f.dptr.data();
    }
    must(vp_circuit_upload(g_ctx, n, desc.data()), "vp_circuit_upload");
    evaluate();
}

void prover::evaluate() {                                           // src/prover.cpp:27-91: the layers are evaluated in HBM
    const layer &L0 = C.circuit[0];
    circuitValue.resize(1);                                         // the commitment of this binary (lib/virgo on the CPU) reads the padded input layer only
    circuitValue[0].assign(1ULL << L0.bitLength, F_ZERO);
    for (u64 g = 0; g < L0.size; ++g) circuitValue[0][g] = F((long long) L0.gates[g].u);
    must(vp_evaluate(g_ctx, cF(circuitValue[0].data()), L0.size), "vp_evaluate");
}

void prover::init() {                                               // src/prover.cpp:132-160 without the bookkeeping tables (they live in HBM)
    int max_bl = 0;
    for (auto &c : C.circuit) max_bl = std::max(max_bl, c.bitLength);
    r_u.assign(max_bl, F_ZERO);
    r_liu.assign(max_bl, F_ZERO);
    r_v.assign(C.size, std::vector<F>());
    for (int i = 1; i < C.size; ++i)
        if (~C.circuit[i].maxDadBitLength) r_v[i].assign(C.circuit[i].maxDadBitLength, F_ZERO);
}

F prover::Vres(const vector<F>::const_iterator &r_0, int r_0_size) {
    prove_timer.start();
    F out;
    must(vp_vres(g_ctx, r_0_size ? cF(&*r_0) : nullptr, r_0_size, mF(&out)), "vp_vres");
    prove_timer.stop();
    return out;
}

void prover::sumcheckInitAll(const vector<F>::const_iterator &r_last) {
    prove_timer.start();
    if (r_liu.empty()) init();                                     // verifier.cpp never calls init() itself (the reference's ctor chain does)
    sumcheckLayerId = C.size;
    for (int i = 0; i < C.circuit[C.size - 1].bitLength; ++i) r_liu[i] = r_last[i];
    prove_timer.stop();
}
void prover::sumcheckInit() { --sumcheckLayerId; }

void prover::sumcheckInitPhase1(const F &assert_random) {
    prove_timer.start();
    must(vp_phase1_init(g_ctx, sumcheckLayerId, cF(r_liu.data()), cF(&assert_random)), "vp_phase1_init");
    round = 0;
    prove_timer.stop();
}
void prover::sumcheckInitPhase2() {
    prove_timer.start();
    must(vp_phase2_init(g_ctx, sumcheckLayerId, cF(r_u.data())), "vp_phase2_init");
    round = 0;
    prove_timer.stop();
}
void prover::sumcheckInitLiu(vector<F>::const_iterator s) {
    prove_timer.start();
    std::vector<const vp_F *> rv(C.size, nullptr);
    for (int k = sumcheckLayerId; k < C.size; ++k) if (!r_v[k].empty()) rv[k] = cF(r_v[k].data());
    must(vp_liu_init(g_ctx, sumcheckLayerId, cF(r_u.data()), rv.data(), cF(&*s)), "vp_liu_init");
    round = 0;
    prove_timer.stop();
}

quadratic_poly prover::sumcheckUpdate(const F &previous_random, vector<F> &r_arr, int) {
    prove_timer.start();
    if (round) r_arr.at(round - 1) = previous_random;
    ++round;
    F p[3];
    must(vp_round(g_ctx, cF(&previous_random), mF(p)), "vp_round");
    prove_timer.stop();
    proof_size += sizeof(F) * 3;
    return quadratic_poly(p[0], p[1], p[2]);
}
quadratic_poly prover::sumcheckUpdateEach(const F &, int) { return quadratic_poly(); }      // private helper of the CPU loops: unused here
quadratic_poly prover::sumcheckUpdatePhase1(const F &r) { return sumcheckUpdate(r, r_u, 0); }
quadratic_poly prover::sumcheckUpdatePhase2(const F &r) { return sumcheckUpdate(r, r_v[sumcheckLayerId], 0); }
quadratic_poly prover::sumcheckLiuUpdate(const F &r) { return sumcheckUpdate(r, r_liu, 0); }

void prover::sumcheckFinalize1(const F &previousRandom, F &claim) {
    prove_timer.start();
    if (round) r_u[round - 1] = previousRandom;
    must(vp_finalize(g_ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
    prove_timer.stop();
    proof_size += sizeof(F);
}
void prover::sumcheckFinalize2(const F &previousRandom, vector<F>::iterator claims) {
    prove_timer.start();
    if (round) r_v[sumcheckLayerId][round - 1] = previousRandom;
    std::vector<F> tmp(sumcheckLayerId);
    must(vp_finalize(g_ctx, cF(&previousRandom), mF(tmp.data()), sumcheckLayerId), "vp_finalize");
    for (int i = 0; i < sumcheckLayerId; ++i) claims[i] = tmp[i];
    proof_size += sizeof(F) * sumcheckLayerId;
    prove_timer.stop();
}
void prover::sumcheckLiuFinalize(const F &previousRandom, F &claim) {
    if (round) r_liu[round - 1] = previousRandom;
    must(vp_finalize(g_ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
}

double prover::proveTime() const { return prove_timer.elapse_sec(); }
double prover::proofSize() const { return (double) proof_size / 1024.0; }

#ifdef USE_VIRGO
// The commitment of this binary: lib/virgo's own CPU prover on the padded input layer (see the header of this file).
virgo::__hhash_digest prover::commit_private() {
    std::vector<F> mask(1, F_ZERO);
    return poly_prover.commit_private_array(circuitValue[0].data(), C.circuit[0].bitLength, mask);
}
F prover::inner_prod(const vector<F> &a, const vector<F> &b, u64 l) {
    F s = F_ZERO;
    for (u64 i = 0; i < l; ++i) s = s + a[i] * b[i];
    return s;
}
virgo::__hhash_digest prover::commit_public(vector<F> &pub, F &inner_product_sum, std::vector<F> &mask, vector<F> &all_sum) {
    prove_timer.start();
    inner_product_sum = inner_prod(circuitValue[0], pub, C.circuit[0].size);
    prove_timer.stop();
    return poly_prover.commit_public_array(mask, pub.data(), C.circuit[0].bitLength, inner_product_sum, all_sum.data());
}
#endif
