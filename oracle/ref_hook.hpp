// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// Force-included (-include) in front of the *unmodified* reference translation
// units src/verifier.cpp and src/main.cpp when oracle/Makefile builds
// oracle/_ref/ref_run.  The reference's `class prover` (src/prover.h:12-66) is
// compiled under the name `ref_prover` (prover.cpp gets -Dprover=ref_prover);
// this header then defines a thin `class prover` with the same public surface
// that forwards every call to the real reference prover and appends each
// returned protocol message to a transcript file.  The reference verifier
// (src/verifier.cpp:134-189) therefore drives the real reference prover with
// its own, untouched randomness schedule, and we get the prover messages in
// call order without editing a single reference line.
//
// Transcript layout (little-endian u64 pairs {real,img}), SURVEY.md §8c:
//   merkle_root_l[32] | Vres[16] | per layer i = size-1..1:
//     phase-1 polys (48 B each) | final_claim_u[16] |
//     phase-2 polys | final_claims_v[i][0..i) (16 B each) |
//     Liu polys | vr[16]
//   | merkle_root_h[32] | input_0[16] | all_sum[0..65) (16 B each)
// With PC switched off (VP_REF_PC=0) only the GKR slice is written.
#pragma once
#define prover ref_prover
#include "prover.h"
#undef prover
#include <cstdio>
#include <cstring>

struct vp_ref_pc_off_stop {};   // thrown after the last GKR message when PC is off

struct vp_ref_state {
    FILE *dump = nullptr;
    int pc_on = 1;
    unsigned long rounds = 0;
};
extern vp_ref_state g_vp_ref;

static inline void vp_ref_dumpF(const F &x) {
    if (!g_vp_ref.dump) return;
    unsigned long long w[2] = {x.real, x.img};
    fwrite(w, 8, 2, g_vp_ref.dump);
}
static inline void vp_ref_dumpQ(const quadratic_poly &q) {
    vp_ref_dumpF(q.a); vp_ref_dumpF(q.b); vp_ref_dumpF(q.c);
}
static inline void vp_ref_dumpH(const virgo::__hhash_digest &h) {
    if (!g_vp_ref.dump) return;
    fwrite(&h, 32, 1, g_vp_ref.dump);
}

class prover {
public:
    ref_prover *rp;
    virgo::poly_commit::poly_commit_prover &poly_prover;
    int layer_id = 0;   // mirrors ref_prover::sumcheckLayerId (private there)

    explicit prover(const layeredCircuit &cir)
        : rp(new ref_prover(cir)), poly_prover(rp->poly_prover), C_(cir) {}

    void init() { rp->init(); }
    void sumcheckInitAll(const vector<F>::const_iterator &r_last) {
        layer_id = C_.size;
        rp->sumcheckInitAll(r_last);
    }
    void sumcheckInit() { --layer_id; rp->sumcheckInit(); }
    void sumcheckInitPhase1(const F &assert_random) { rp->sumcheckInitPhase1(assert_random); }
    void sumcheckInitPhase2() { rp->sumcheckInitPhase2(); }
    void sumcheckInitLiu(vector<F>::const_iterator s) { rp->sumcheckInitLiu(s); }

    quadratic_poly sumcheckUpdatePhase1(const F &r) {
        auto q = rp->sumcheckUpdatePhase1(r); vp_ref_dumpQ(q); ++g_vp_ref.rounds; return q;
    }
    quadratic_poly sumcheckUpdatePhase2(const F &r) {
        auto q = rp->sumcheckUpdatePhase2(r); vp_ref_dumpQ(q); ++g_vp_ref.rounds; return q;
    }
    quadratic_poly sumcheckLiuUpdate(const F &r) {
        auto q = rp->sumcheckLiuUpdate(r); vp_ref_dumpQ(q); ++g_vp_ref.rounds; return q;
    }
    void sumcheckFinalize1(const F &r, F &claim) { rp->sumcheckFinalize1(r, claim); vp_ref_dumpF(claim); }
    void sumcheckFinalize2(const F &r, vector<F>::iterator claims) {
        rp->sumcheckFinalize2(r, claims);
        for (int j = 0; j < layer_id; ++j) vp_ref_dumpF(claims[j]);
    }
    void sumcheckLiuFinalize(const F &r, F &claim) {
        rp->sumcheckLiuFinalize(r, claim); vp_ref_dumpF(claim);
        // PC off: verifyPoly (verifier.cpp:363) would touch FFT scratch that only commit_private
        // allocates, so stop right after the last GKR message (layer 1's Liu claim).
        if (!g_vp_ref.pc_on && layer_id == 1) throw vp_ref_pc_off_stop();
    }

    F Vres(const vector<F>::const_iterator &r_0, int r_0_size) {
        F v = rp->Vres(r_0, r_0_size); vp_ref_dumpF(v); return v;
    }
    double proveTime() const { return rp->proveTime(); }
    double proofSize() const { return rp->proofSize(); }

    virgo::__hhash_digest commit_private() {
        virgo::__hhash_digest d;
        memset(&d, 0, sizeof d);
        if (!g_vp_ref.pc_on) return d;
        d = rp->commit_private();
        vp_ref_dumpH(d);
        return d;
    }
    virgo::__hhash_digest commit_public(vector<F> &pub, F &inner_product_sum, std::vector<F> &mask,
                                        vector<F> &all_sum) {
        if (!g_vp_ref.pc_on) throw vp_ref_pc_off_stop();
        auto d = rp->commit_public(pub, inner_product_sum, mask, all_sum);
        vp_ref_dumpH(d);
        vp_ref_dumpF(inner_product_sum);
        for (auto &x : all_sum) vp_ref_dumpF(x);
        return d;
    }

private:
    const layeredCircuit &C_;
};
