/* TEST INFRASTRUCTURE ONLY — the oracle.  Never linked into, imported by or called from the product
 * path (virgo-plus_amd/, include/vpgpu.h).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * CPU restatement of the reference GKR prover path (TAMUCrypto/virgo-plus @ /root/reference):
 *   field F_p^2, p = 2^61-1 ........ lib/virgo/src/fieldElement.cpp:34-104,322-367
 *   linear / quadratic polys ....... src/polynomial.cpp:64-131
 *   eq / beta table ................ src/utils.cpp:8-45
 *   .pws loader, levelisation ...... src/main.cpp:15-137,176-231
 *   subsetInit, randomize .......... src/circuit.cpp:17-80
 *   prover ......................... src/prover.cpp:27-521
 *   verifier schedule + checks ..... src/verifier.cpp:12-337
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against transcripts
 * produced by the real reference compiled here (oracle/_ref/ref_run, see oracle/Makefile):
 * tests/golden/ (the .bin files), whose SHA-256 digests equal the ones recorded in SURVEY.md §8c, plus the
 * field / root-of-unity / F::random known answers listed there.
 */
#ifndef VP_ORACLE_H
#define VP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_circuit orc_circuit;

/* F = {real, img}, canonical (each limb in [0, p)). */
typedef struct { uint64_t real, img; } orc_F;

typedef struct {
    double prove_sec;        /* reference "Prove Time" definition (src/prover.cpp:549)   */
    double evaluate_sec;     /* prover::evaluate (outside prove_timer in the reference)            */
    double verify_sec;       /* verifier-side work incl. the "slow" predicate loops                 */
    uint64_t mult_count;     /* F::multCounter semantics (fieldElement.cpp:50-53), 64-bit           */
    uint64_t add_count;      /* F::addCounter semantics (fieldElement.cpp:35-38,81-84,99-103)       */
    uint64_t rounds;         /* number of sumcheckUpdate calls                                      */
    uint64_t pairs;          /* active fold pairs (src/prover.cpp:483)                              */
    double   proof_kb;       /* prover::proofSize()                                                 */
    int      verified;       /* 1 = every verifier check passed                                     */
} orc_stats;

/* ---- circuits ------------------------------------------------------------------------------------ */
/* Parse a .pws file (grammar of src/main.cpp:161-168), replicate the DAG `blocks` times with all inputs
 * numbered first, draw the witness with glibc random()%p in file order
 * (src/main.cpp:188; `seed` < 0 keeps glibc's default seed, else srandom(seed) first), levelise.      */
orc_circuit *orc_circuit_from_pws(const char *path, int blocks, long seed);
/* layeredCircuit::randomize(layerNum, eachLayer) (src/circuit.cpp:17-41), g++ argument order.         */
orc_circuit *orc_circuit_randomize(int layers, int log_size, long seed);
/* Arbitrary layered circuit (tests of gate types the .pws loader never emits: Addc, Mulc, Copy, AntiNaab, assert
 * gates).  Flat arrays over all gates in layer order; layer_sizes[n_layers]; for layer 0 `u` carries the input value.
 * c_pairs: {real,img} per gate (used by Addc/Mulc).                                                       */
orc_circuit *orc_circuit_custom(int n_layers, const uint64_t *layer_sizes, const int32_t *ty, const int32_t *l, const uint64_t *u,
                                const uint64_t *v, const uint64_t *c_pairs, const uint8_t *is_assert);
void orc_circuit_free(orc_circuit *);
int  orc_circuit_layers(const orc_circuit *);
uint64_t orc_circuit_layer_size(const orc_circuit *, int layer);
int  orc_circuit_layer_bitlen(const orc_circuit *, int layer);
uint64_t orc_circuit_gates(const orc_circuit *);
/* 128-bit structural hash, same serialisation as oracle/ref_driver.cpp::circuit_hash (after subsetInit). */
void orc_circuit_hash(orc_circuit *, uint64_t out[2]);
/* Flat export of one layer's gate table (after subsetInit): arrays of length layer size.              */
void orc_circuit_export_layer(orc_circuit *, int layer, int32_t *ty, int32_t *l, uint64_t *u, uint64_t *v,
                              uint64_t *lv);
/* Input-layer witness (layer 0 values as drawn), length = layer 0 size.                               */
void orc_circuit_inputs(const orc_circuit *, orc_F *out);

/* ---- whole GKR proof ----------------------------------------------------------------------------- */
/* F::init() (srand(3396)), subsetInit, prover(evaluate), verifier::verify with the polynomial
 * commitment switched off.  Writes the GKR transcript slice (SURVEY §8c layout without merkle_root_l
 * and without the trailing PC fields) into `transcript`; returns its length in bytes, or <0 on error
 * (-1: capacity too small).                                                                           */
int64_t orc_prove_gkr(orc_circuit *, uint8_t *transcript, int64_t capacity, orc_stats *stats);

/* The same proof in Fiat-Shamir mode (not a reference mode — the reference draws glibc random()): every challenge
 * is derived by a SHA3-256 chain from the statement (serialised circuit, subset tables, input values) and all prover messages before
 * it, one challenge per sumcheck round AFTER that round's polynomial; definition in virgo-plus_amd/host/verifier.cpp (proveFS).  This
 * is an independent second implementation (own sponge, own serialiser, own prover): tests compare its proof bytes with the product's. */
int64_t orc_prove_fs(orc_circuit *, uint8_t *proof, int64_t capacity, orc_stats *stats);

/* ---- primitives, for unit parity tests against the HIP kernels ------------------------------------ */
void orc_f_add(const orc_F *a, const orc_F *b, orc_F *out);
void orc_f_sub(const orc_F *a, const orc_F *b, orc_F *out);
void orc_f_mul(const orc_F *a, const orc_F *b, orc_F *out);
void orc_f_neg(const orc_F *a, orc_F *out);
void orc_f_inv(const orc_F *a, orc_F *out);
void orc_f_root_of_unity(int log_order, orc_F *out);
/* srand(seed) then n draws of F::random() (fieldElement.cpp:119-124,362-367).                         */
void orc_f_random_seq(unsigned seed, int n, orc_F *out);
/* n further draws from the current glibc state (no reseed), e.g. right after orc_prove_full: the draws the
 * reference's verifier makes next.                                                                     */
void orc_f_random_next(int n, orc_F *out);
/* F::random() draws consumed by fft_circuit_gkr::fft_gkr(lg) (lib/virgo/src/fft_circuit_GKR.cpp:84,106,763-764,840),
 * which verify_poly_commitment runs (vpd_verifier.cpp:92) before commit_phase draws the FRI challenges (:56).
 * Pinned by tests/test_oracle_golden.py against the challenges the real reference recorded.            */
int orc_fft_gkr_draws(int lg);
/* fft_circuit_gkr::fft_gkr(lg) (lib/virgo/src/fft_circuit_GKR.cpp:833-849), prover and verifier: seed >= 0 -> srand(seed) first, else the
 * current glibc state.  msgs receives the record the real reference is hooked for (oracle/ref_driver.cpp --dump-fft): the circuit's 64
 * outputs, then per sumcheck its round polynomials (3 F each) and the table value(s) claimed at its end — addition layer (lg + 6 rounds,
 * v_u), multiplication layer (lg rounds, v_u), per inverse-FFT depth: phase 1 (lg rounds, v_u), phase 2 (lg rounds, v_v).  Returns the
 * byte count (16 * (64 + 3 * (2 lg^2 + 2 lg + 6) + 2 + 2 lg)), -1 if capacity is too small.  *verified = every check of the
 * reference's embedded verifier held.                                                                 */
int64_t orc_fft_gkr(int lg, long seed, uint8_t *msgs, int64_t capacity, double *prove_sec, int *verified);
/* initBetaTable(beta, n, r, init): out has 2^n entries.                                               */
void orc_beta_table(const orc_F *r, int n, const orc_F *init, orc_F *out);
/* One call of prover::sumcheckUpdateEach on value tables (the .a parts implied zero / carried in
 * `a_*`): literal AoS restatement.  Tables V, add, mult are arrays of `total` linear polys stored as
 * {a,b} pairs (2*total orc_F each), folded in place; returns the 3-coefficient round polynomial.      */
void orc_update_each(orc_F *V, orc_F *add, orc_F *mult, uint64_t total, uint64_t total_size,
                     const orc_F *prev, orc_F out_poly[3]);

/* ---- Virgo polynomial commitment, commit side (lib/virgo/src) ------------------------------------- */
/* my_hhash (my_hhash.h:27-33): SHA3-256 of exactly 64 bytes (FIPS 202; the reference calls the prebuilt,
 * source-less libXKCP.a, so the pin is the standard + Python hashlib + the golden Merkle roots).         */
void orc_sha3_256_64(const uint8_t in[64], uint8_t out[32]);
/* fast_fourier_transform (RS_polynomial.cpp:26-157): out[k] = sum_j coefs[j] * w^(jk), w = root of order `order`. */
void orc_fft(const orc_F *coefs, int coef_len, int order, orc_F *out);
/* inverse_fast_fourier_transform (RS_polynomial.cpp:159-220) with coef_len == order == n.                */
void orc_ifft(const orc_F *evals, int n, orc_F *out);
/* prover::commit_private (src/prover.cpp:524-530 -> poly_commit.h:41-124 -> fri.cpp:36-139 ->
 * merkle_tree.cpp:7-51): Merkle root over the RS-encoded input layer.                                    */
int orc_commit_private(orc_circuit *, uint8_t root[32]);
/* The whole protocol up to and including prover::commit_public (src/verifier.cpp:134-169,363-379 ->
 * src/prover.cpp:542-546 -> poly_commit.h:126-349): writes the FULL golden layout
 * merkle_root_l | GKR slice | merkle_root_h | input_0 | all_sum[65].  Returns the length or <0.           */
int64_t orc_prove_full(orc_circuit *, uint8_t *transcript, int64_t capacity, orc_stats *stats);
/* commit_public_array on caller-supplied arrays (unit parity for the device kernels): input and pub have
 * 2^n_bits entries; outputs inner product, all_sum[65], root_h.                                          */
int orc_commit_public(const orc_F *input, const orc_F *pub, int n_bits, uint64_t n_used, orc_F *inner, orc_F *all_sum,
                      uint8_t root_h[32]);

/* FRI commit phase on the virtual oracle (poly_commit.h:294-318 builds it; poly_commit_prover::commit_phase,
 * vpd_verifier.cpp:44-74, loops fri::commit_phase_step, fri.cpp:289-424).  `r` are the n_bits-6 fold challenges
 * (the reference draws them with F::random() inside commit_phase; tests feed the recorded ones).  roots gets one
 * 32-byte Merkle root per step; final_code the last codeword in the reference's interleaved layout
 * [i << 7 | slice << 1 | hi], i < 16 (2048 elements).                                                     */
/* The point the input layer is opened at (r_liu after the last Liu sumcheck) of the last orc_prove_full run.  */
int orc_last_point(const orc_circuit *, orc_F *out, int n);
int orc_fri_commit(const orc_F *input, const orc_F *pub, int n_bits, const orc_F *r, uint8_t *roots, orc_F *final_code);

#ifdef __cplusplus
}
#endif
#endif
