// Host verifier: the caller the drop-in prover must satisfy (reference: src/verifier.h, src/verifier.cpp).
// It stays plain host C++ ("OUT OF SCOPE as a GPU target").  Two ways to run it:
//   verify()            interactive, exactly the reference's call sequence into `prover` (verifier.cpp:134-169):
//                       challenges are drawn with F::random() as the prover's messages arrive;
//   drawTape()+check()  the same checks replayed over a recorded transcript produced by
//                       prover::proveGKR from a pre-drawn tape (the challenges do not depend on the
//                       transcript: lib/virgo/src/fieldElement.cpp:119-124).
// Both record the prover's messages in the golden transcript layout (GKR slice).
#pragma once
#include <cstdint>
#include <vector>

#include "circuit.hpp"
#include "polynomial.hpp"
#include "prover.hpp"
#include "sha3.hpp"

class verifier {
public:
    verifier(prover *pr, const layeredCircuit &cir);

    bool verify();                                              // interactive GKR (needs pr != nullptr)
    // The complete protocol of the reference's verifier::verify() (src/verifier.cpp:134-189): commit_private, GKR,
    // verifyPoly = commit_public + FRI commit phase + `reps` query repetitions (vpd_verifier.cpp:76-328; the
    // reference hard-codes 33).  fullTranscript() is then merkle_root_l | GKR | merkle_root_h | input_0 | all_sum[65].
    bool verifyFull(int reps = 33);
    prover *pred_dev = nullptr;     // when set, the O(|C|) predicate loops run on the device (prover::predicates) instead of the host
    bool fri_batched = true;        // FRI commit phase as one device pass (prover::friCommit) instead of one friStep per challenge
    const std::vector<uint8_t> &fullTranscript() const { return full_tr; }
    // FRI commit phase of the last verifyFull(): Merkle root per fold step (32 bytes each), final codeword (2048), fold challenges
    const std::vector<uint8_t> &friRoots() const { return fri_roots_; }
    const std::vector<F> &friFinalCode() const { return fri_final_; }
    const std::vector<F> &friChallenges() const { return fri_r_; }
    static int fftGkrDraws(int lg);
    double polyVerifyTime() const { return poly_timer.elapse_sec(); }
    // "Polynomial commitment: prove time" by the REFERENCE's definition (src/verifier.cpp:183): poly_prover.total_time = commit_private +
    // commit_public + commit_phase (lib/virgo/src/poly_commit.h:43,121,336,345, vpd_verifier.cpp:70) + fft_gkr's prover time (:92-94).
    // Answering the verifier's queries (fri::request_*) is outside that number in the reference and is reported separately here.
    double polyProveTime() const { return poly_prove_timer.elapse_sec(); }
    double polyOpenTime() const { return open_timer.elapse_sec(); }
    // fft_gkr of the last verifyFull(): its share of polyProveTime() (the reference's p_time_fft, vpd_verifier.cpp:92-94) and its messages
    double fftGkrProveTime() const { return fft_gkr_timer.elapse_sec(); }
    const std::vector<F> &fftGkrMessages() const { return fft_gkr_msgs_; }
    // Fiat-Shamir mode (the reference's GKRProof.hpp / transcriptCache.hpp are dead code, so this is a separate
    // mode, not part of the bit-exact parity): every challenge is SHA3-256-derived from the circuit hash and ALL prover messages
    // sent before it, so the transcript is a non-interactive proof.  Differences from the interactive schedule: the challenges of
    // a sumcheck are drawn one per round (after that round's polynomial), not all up front.
    //   proveFS()   runs the prover (interactive entry points, one vp_round per derived challenge); transcript() is the proof;
    //   checkFS()   needs no prover and no tape: re-derives every challenge from the proof and runs the same checks.
    bool proveFS();
    bool checkFS(const std::vector<uint8_t> &proof);
    std::vector<F> drawTape();                                  // the verifier's draws, in its own order
    bool check(const std::vector<F> &tape, const std::vector<uint8_t> &transcript);   // replay

    const std::vector<uint8_t> &transcript() const { return tr; }
    const std::vector<F> &tape() const { return tape_; }
    const std::vector<F> &finalPoint() const { return r_liu; }     // r_liu after the last Liu sumcheck: where the input MLE is opened
    double verifyTime() const { return verify_timer.elapse_sec(); }
    bool skip_predicates = false;      // replay only: skip the O(|C|) wiring-predicate check (getFinalValue)

private:
    bool run();
    bool verifyPhase1(int layer_id, F &previousSum);
    bool verifyPhase2(int layer_id, F &previousSum);
    bool verifyLiu(int layer_id, F &previousSum);
    void predicatePhase1(int layer_id);
    void predicatePhase2(int layer_id);
    void predicatesOnDevice(int layer_id, bool with_phase2);
    F getFinalValue(int layer_id, const F &claim_u, const std::vector<F> &claim_v);
    bool checkInput(const F &claim);
    bool verifyPoly(const prover::hhash_digest &root_l, const F &claim, int reps);
    bool checkOpening(const vph::hhash_digest &root, u64 leaf, const std::vector<F> &vals, const std::vector<prover::hhash_digest> &path);

    F draw();
    quadratic_poly nextPoly(int phase, const F &prev);
    F nextF();
    void putF(const F &x);

    prover *p;
    const layeredCircuit &C;
    bool replay = false;
    bool fs = false;                                   // challenges derived from the transcript
    vph::hhash_digest fs_state{};
    uint64_t fs_ctr = 0;
    void fsInit();
    const std::vector<F> *rtape = nullptr;
    const std::vector<uint8_t> *rtr = nullptr;
    size_t tape_pos = 0, tr_pos = 0;
    std::vector<F> tape_;
    std::vector<uint8_t> tr;

    int max_bl = 0;
    std::vector<F> beta_g, beta_u, beta_v, r_u, r_liu, sig;
    std::vector<std::vector<F>> r_v;
    F coeff_l[(int) gateType::SIZE];
    std::vector<F> coeff_r[(int) gateType::SIZE];
    F bias, final_claim_u, assert_random;
    std::vector<std::vector<F>> final_claims_v;
    timer verify_timer, poly_timer, poly_prove_timer, fft_gkr_timer, open_timer;
    std::vector<F> fft_gkr_msgs_;
    std::vector<uint8_t> full_tr;
    std::vector<uint8_t> fri_roots_; std::vector<F> fri_final_, fri_r_;
    bool input_check_by_commitment = false;
    F last_claim;
};

// eq table (reference: src/utils.cpp:29-45), host version used by the verifier only
void initBetaTable(std::vector<F> &beta_g, int gLength, const std::vector<F>::const_iterator &r, const F &init);
