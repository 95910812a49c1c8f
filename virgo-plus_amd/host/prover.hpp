// Host mirror of the reference's `class prover` (src/prover.h:12-66): same method names, argument
// meaning and call-order contract, so the verifier below (and the reference's own verifier.cpp, see
// INTEGRATION.md) can drive it unchanged.  Every sumcheck method forwards to the C ABI of
// libvpgpu.so (include/vpgpu.h); no arithmetic on tables happens on the host and there is no CPU
// fallback: construction throws if the device library cannot run.
#pragma once
#include <chrono>
#include <stdexcept>
#include <vector>

#include "../../include/vpgpu.h"
#include "circuit.hpp"
#include "polynomial.hpp"

class timer {          // accumulate semantics of lib/virgo/src/timer.hpp:11-25
public:
    void start() { t0 = std::chrono::high_resolution_clock::now(); }
    void stop() { total += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); }
    void clear() { total = 0; }
    void add(double sec) { total += sec; }       // a span measured elsewhere (the GKR part of a protocol pass, in device time when its completion is deferred)
    double elapse_sec() const { return total; }
private:
    std::chrono::high_resolution_clock::time_point t0;
    double total = 0;
};

class prover {
public:
    explicit prover(const layeredCircuit &cir, int device = 0, const vp_options *options = nullptr);   // options: include/vpgpu.h (NULL = shipped defaults)
    ~prover();
    prover(const prover &) = delete;
    prover &operator=(const prover &) = delete;

    void evaluate();
    void init();
    void sumcheckInitAll(const std::vector<F>::const_iterator &r_last);
    void sumcheckInit();
    void sumcheckInitPhase1(const F &assert_random);
    void sumcheckInitPhase2();
    void sumcheckInitLiu(std::vector<F>::const_iterator s);

    quadratic_poly sumcheckUpdatePhase1(const F &previousRandom);
    quadratic_poly sumcheckUpdatePhase2(const F &previousRandom);
    quadratic_poly sumcheckLiuUpdate(const F &previousRandom);

    void sumcheckFinalize1(const F &previousRandom, F &claim);
    void sumcheckFinalize2(const F &previousRandom, std::vector<F>::iterator claims);
    void sumcheckLiuFinalize(const F &previousRandom, F &claim);

    F Vres(const std::vector<F>::const_iterator &r_0, int r_0_size);

    // Virgo polynomial commitment (reference: `#ifdef USE_VIRGO` block of src/prover.h:38-43)
    struct hhash_digest { unsigned char b[32]; };
    hhash_digest commit_private();                       // src/prover.cpp:524-530
    // src/prover.cpp:542-546 (the mask argument of the reference is the one-element zero vector and is implied)
    hhash_digest commit_public(std::vector<F> &pub, F &inner_product_sum, std::vector<F> &all_sum);
    // the same two with the mask vectors lib/virgo's commit_private_array / commit_public_array take (poly_commit.h:41-42,126-128; the reference's own
    // prover passes one zero, src/prover.cpp:526, and its commit_public has this signature, src/prover.cpp:542): a non-zero private mask puts content into the
    // 65th slice of every oracle (vp_commit_private_masked / vp_commit_public_masked); an all-zero one is the call above
    hhash_digest commit_private(const std::vector<F> &mask);
    hhash_digest commit_public(std::vector<F> &pub, F &inner_product_sum, std::vector<F> &mask, std::vector<F> &all_sum);
    std::vector<F> friFinalMask();                       // fri::cpd.rs_codeword_msk[last] (vpd_verifier.cpp:321-325)
    // extension: the protocol's own public vector, pub = eq(point, .) (src/verifier.cpp:368-369), built on the device from the point
    // (vp_commit_public_eq) — same outputs as commit_public on that table, nothing of it crosses PCIe
    hhash_digest commit_public_eq(const std::vector<F> &point, F &inner_product_sum, std::vector<F> &all_sum);
    // poly_commit_prover::commit_phase pieces (vpd_verifier.cpp:44-74 -> fri::commit_phase_step / commit_phase_final)
    // verifier-side wiring predicates of one layer on the device (vp_predicates): 5 + 7*layer sums, see include/vpgpu.h
    std::vector<F> predicates(int layer, const std::vector<F> &r_g, const F &assert_random, const std::vector<F> &r_u,
                              const std::vector<F> &r_v, int n_v);
    // the verifier's other O(|C|) loops on the device: gr of verifyLiu (vp_liu_gr) and a layer's MLE at r (vp_layer_mle)
    F liuGr(int layer, const std::vector<F> &r_u, const std::vector<std::vector<F>> &r_v, const std::vector<F> &sig, const std::vector<F> &r_liu);
    F layerMle(int layer, const std::vector<F> &r, int n);
    hhash_digest friStep(const F &r);
    std::vector<hhash_digest> friCommit(const std::vector<F> &r);   // every step in one device pass (challenges are transcript-independent)
    std::vector<F> friFinal();                           // 2048 elements, reference layout [i << 7 | slice << 1 | hi]
    // fri::request_init_value_with_merkle (oracle 0 = l, 1 = h) / fri::request_step_commit (oracle 2 + level)
    void friOpen(int oracle, u64 leaf, std::vector<F> &values /* 130 */, std::vector<hhash_digest> &path);
    // fft_circuit_gkr::fft_gkr (lib/virgo/src/fft_circuit_GKR.cpp:833-849), prover side on the device (vp_fft_gkr): tape = the verifier's
    // draws in the reference's order; returns every prover message (layouts: include/vpgpu.h)
    std::vector<F> fftGkr(int lg, const std::vector<F> &tape);
    // the same in two halves (vp_fft_gkr_begin / _end): queued on a stream of its own, collected later; other prover calls may run in between
    void fftGkrBegin(int lg, const std::vector<F> &tape);
    std::vector<F> fftGkrEnd(int lg);
    void fftGkrCancel() noexcept;                   // drop a begun run (a pass that failed between begin and end)
    double commitDeviceMs();

    double proveTime() const { return prove_timer.elapse_sec(); }
    void addProveTime(double sec) { prove_timer.add(sec); }     // vph_prove_protocol_ex calls vp_prove_gkr itself: its GKR span belongs to Prove Time (src/verifier.cpp:177-184)
    // where Prove Time goes on the interactive path: phase inits | round messages | finalize calls | Vres (all inside prove_timer's spans,
    // except sumcheckLiuFinalize, which the reference leaves out of its timer as well)
    double initTime() const { return init_timer.elapse_sec(); }
    double roundTime() const { return round_timer.elapse_sec(); }
    double finalizeTime() const { return fin_timer.elapse_sec(); }
    double proofSize() const { return (double) proof_size / 1024.0; }

    // ---- extensions (not in the reference) ----
    // Whole GKR part in one device pass; tape = the verifier's draws in its own order
    // (layout in include/vpgpu.h).  Appends the messages to `transcript`.
    void proveGKR(const std::vector<F> &tape, std::vector<uint8_t> &transcript);
    void gkrSizes(u64 &n_tape, u64 &n_transcript_bytes);
    std::vector<F> layerValues(int layer);
    vp_ctx *context() { return ctx; }
    vp_stats stats();

private:
    quadratic_poly sumcheckUpdate(const F &previous_random, std::vector<F> &r_arr);
    void check(int rc, const char *what);

    const layeredCircuit &C;
    vp_ctx *ctx = nullptr;
    std::vector<F> r_u, r_liu;
    std::vector<std::vector<F>> r_v;
    int round = 0;
    int sumcheckLayerId = 0;
    timer prove_timer, init_timer, round_timer, fin_timer;
    bool masked = false;                                 // the standing private commitment carries a non-zero mask slice
    u64 proof_size = 0;
};
