/* vphost.h — C entry points of libvphost.so for the Python harness (tests/, bench.py) and other FFI
 * users: circuit loading, one prover session per GPU, interactive and batched GKR proofs.  Everything
 * here is host orchestration; the arithmetic runs in libvpgpu.so (include/vpgpu.h).                  */
#ifndef VPHOST_H
#define VPHOST_H
#include <stdint.h>
#include "../../include/vpgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vph_circuit vph_circuit;
typedef struct vph_session vph_session;

typedef struct {
    double prove_sec;        /* reference "Prove Time" definition: host wall clock over the prover-method spans */
    double gkr_device_ms;    /* hipEvent time of the batched device pass (0 for interactive)                   */
    double evaluate_ms;      /* device time of circuit evaluation                                              */
    double verify_sec;       /* host verifier time                                                             */
    double fold_ms;          /* profiled dominant-kernel launches: total duration ...                          */
    uint64_t fold_launches;  /* ... count ...                                                                  */
    uint64_t fold_bytes;     /* ... and algorithmic bytes                                      */
    uint64_t rounds;
    uint64_t launches;
    double proof_kb;
    int verified;
} vph_result;

/* seed < 0: keep the process's current glibc random() state (default seed 1 in a fresh process);
 * else srandom(seed) before the witness is drawn (src/main.cpp:188).                                 */
vph_circuit *vph_circuit_from_pws(const char *path, int blocks, long seed, char *err, int errlen);
vph_circuit *vph_circuit_randomize(int layers, int log_size, long seed);
/* Arbitrary layered circuit from flat arrays (all gate types of enum gateType, constants, assert gates); subsetInit runs. */
vph_circuit *vph_circuit_custom(int n_layers, const uint64_t *layer_sizes, const int32_t *ty, const int32_t *l, const uint64_t *u,
                                const uint64_t *v, const uint64_t *c_pairs, const uint8_t *is_assert);
void vph_circuit_free(vph_circuit *);
int vph_circuit_layers(const vph_circuit *);
uint64_t vph_circuit_gates(const vph_circuit *);
uint64_t vph_circuit_layer_size(const vph_circuit *, int layer);
int vph_circuit_layer_bitlen(const vph_circuit *, int layer);
void vph_circuit_hash(const vph_circuit *, uint64_t out[2]);

/* Uploads the circuit to `device` and evaluates it there.  NULL + message on failure (no CPU fallback). */
vph_session *vph_session_create(vph_circuit *, int device, char *err, int errlen);
vph_session *vph_session_create_opts(vph_circuit *, int device, const vp_options *opt /* include/vpgpu.h, NULL = defaults */, char *err, int errlen);
void vph_session_free(vph_session *);
int vph_set_profiling(vph_session *, int level);
/* the vp_ctx behind the session's prover (for the measurement calls of include/vpgpu.h: vp_get_launch_stats, ...) */
void *vph_session_ctx(vph_session *);
/* circuitValue[layer] copied back (tests).                                                             */
int vph_layer_values(vph_session *, int layer, uint64_t *out_pairs, uint64_t n);

/* F::init() (srand(3396)) + verifier::verify(): one device round trip per sumcheck round.              */
int vph_prove_interactive(vph_session *, uint8_t *transcript, uint64_t capacity, uint64_t *n_written,
                          vph_result *res, char *err, int errlen);
/* F::init() + draw the verifier tape once; it stays attached to the session.                           */
int vph_draw_tape(vph_session *);
/* One batched GKR proof from the attached tape (a bench "step").                                       */
int vph_prove_gkr(vph_session *, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, vph_result *res,
                  char *err, int errlen);
/* Host verifier replay over a transcript made from the attached tape.  skip_predicates=1 skips the
 * O(|C|) wiring-predicate sums (the sumcheck and Liu checks still run).                                */
int vph_check(vph_session *, const uint8_t *transcript, uint64_t n, int skip_predicates, double *verify_sec);
uint64_t vph_transcript_bytes(vph_session *);
/* prover::commit_private(): Merkle root of the RS-encoded input layer (merkle_root_l); device ms via *ms. */
int vph_commit_private(vph_session *, uint8_t root[32], double *ms, char *err, int errlen);
/* prover::commit_public on a caller-supplied public vector of 2^bit_length(layer 0) elements ({real,img} pairs).
 * out = root_h[32] | input_0[16] | all_sum[65*16]  (the tail of the golden transcript layout).         */
int vph_commit_public(vph_session *, const uint64_t *pub_pairs, uint64_t n_pub, uint8_t out[32 + 16 + 65 * 16], double *ms,
                      char *err, int errlen);
/* The COMPLETE protocol with polynomial-commitment verification (verifier::verify, src/verifier.cpp:134-189):
 * commit_private, interactive GKR, commit_public, FRI commit phase with fresh challenges, `reps` query repetitions
 * answered through vp_fri_open and checked on the host.  Returns 0 = accepted, 1 = rejected.  transcript = the golden
 * layout (identical to the reference's up to all_sum); times in seconds.                                  */
int vph_prove_and_verify_full(vph_session *, int reps, uint8_t *transcript, uint64_t capacity, uint64_t *n_written,
                              double *gkr_prove_sec, double *pc_prove_sec, double *verify_sec, char *err, int errlen);
/* FRI commit phase of the last vph_prove_and_verify_full: Merkle root per fold step (32 bytes each), final codeword (2048
 * elements), fold challenges (one element per step); any pointer may be NULL.  Returns the number of steps or -1. */
int vph_last_fri(vph_session *, uint8_t *roots, uint64_t roots_cap, uint64_t *final_pairs, uint64_t *r_pairs);
/* fft_gkr of the last vph_prove_and_verify_full (vp_fft_gkr's message layout); returns the element count or -1.  vph_last_pc_times:
 * [0] "Polynomial commitment: prove time" by the reference's definition (src/verifier.cpp:183: commit_private + commit_public +
 * commit_phase + fft_gkr's prover time), [1] the fft_gkr share of it, [2] answering the queries (not part of [0], as in the reference). */
int64_t vph_last_fft_gkr(vph_session *, uint64_t *pairs, uint64_t cap_elems);
void vph_last_pc_times(vph_session *, double out[3]);
/* r_liu after the last Liu sumcheck of the last vph_prove_full / vph_prove_and_verify_full (the protocol's public vector is its eq
 * table, src/verifier.cpp:368-369); returns the number of coordinates or -1. */
int vph_last_point(vph_session *, uint64_t *pairs, int cap);
/* seconds spent in phase inits | round messages | finalize calls by the last vph_prove_interactive */
void vph_interactive_breakdown(vph_session *, double out[3]);
/* my_hhash on the host (verifier side): SHA3-256 of n 64-byte messages.                                   */
void vph_test_sha3(const uint8_t *in, uint8_t *out, uint64_t n);
/* poly_commit_prover::commit_phase (vpd_verifier.cpp:44-74) with caller-supplied fold challenges: n_steps calls of
 * fri::commit_phase_step, then commit_phase_final.  roots: 32 bytes per step; final_code: 2048 {real,img} pairs.
 * commit_public (or vph_prove_full) must have run on this session.                                       */
int vph_fri_commit(vph_session *, const uint64_t *r_pairs, int n_steps, uint8_t *roots, uint64_t *final_pairs, char *err, int errlen);
/* same through vp_fri_commit: every step in one device pass */
int vph_fri_commit_batched(vph_session *, const uint64_t *r_pairs, int n_steps, uint8_t *roots, uint64_t *final_pairs, char *err, int errlen);
/* The whole protocol of verifier::verify() up to commit_public (src/verifier.cpp:134-169,363-379), interactive
 * GKR: writes merkle_root_l | GKR slice | merkle_root_h | input_0 | all_sum[65] — the golden layout.   */
int vph_prove_full(vph_session *, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, int batched, char *err,
                   int errlen);
/* The prover side of the COMPLETE protocol in one pass (bench step of BASELINE configs[2]).  vph_draw_protocol_tape: F::init() and every
 * draw of verifier::verify() in its order — GKR (src/verifier.cpp:144-279), fft_gkr (fft_circuit_GKR.cpp), FRI fold challenges
 * (vpd_verifier.cpp:57) — valid up front because the reference's challenges are glibc random() (fieldElement.cpp:119-124).
 * vph_prove_protocol: commit_private -> batched GKR -> commit_public on eq(r_liu, .) built on the device (vp_commit_public_eq) -> fft_gkr
 * -> FRI commit phase + final codeword; no verifier work inside.  transcript: the golden layout; fri_roots: 32 bytes per step;
 * final_pairs: 2048 elements; sec[6] = whole pass | commit_private | GKR | commit_public | fft_gkr | FRI commit (host wall clock).   */
int vph_draw_protocol_tape(vph_session *);
int vph_prove_protocol(vph_session *, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, uint8_t *fri_roots, uint64_t roots_cap,
                       uint64_t *final_pairs, double sec[6], char *err, int errlen);            /* = _ex with flags 0 */
/* flags: 0 = every call waits for its result before the next is made (sec[1..5]: host wall clock per call);
 * VPH_PASS_DEFERRED = the calls are queued back to back and collected at the end (vp_set_deferred, include/vpgpu.h: a device that idles between two calls
 *   runs the following milliseconds at a lower clock); sec[1..5]: device time per call;
 * | VPH_PASS_QUEUE_NEXT = before it waits, the pass queues the next pass's head (commit_private of the same witness) behind its own FRI folds, and
 *   the next vph_prove_protocol_ex of the session starts at its GKR part: no idle device between two proofs of a session that proves back to back.  The
 *   openings of THIS pass's commitment are gone once that head runs (it overwrites the codeword); a pass that will be opened is made without the flag. */
enum { VPH_PASS_DEFERRED = 1, VPH_PASS_QUEUE_NEXT = 2 };
int vph_prove_protocol_ex(vph_session *, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, uint8_t *fri_roots, uint64_t roots_cap,
                          uint64_t *final_pairs, double sec[6], int flags, char *err, int errlen);
/* No GPU needed: F::init(), draw the tape for `circuit`, replay the host verifier over `transcript`
 * (GKR slice).  0 = accepted, 1 = rejected.                                                            */
int vph_verify_transcript(vph_circuit *, const uint8_t *transcript, uint64_t n, int skip_predicates);
/* Fiat-Shamir mode: a non-interactive GKR proof.  Every verifier challenge is derived with SHA3-256 from the
 * circuit's structural hash and all prover messages before it (one challenge per sumcheck round, after that round's
 * polynomial); the prover runs through its interactive entry points.  The proof has the transcript layout of vph_prove_gkr.
 * Not comparable with the reference's transcripts (its verifier draws random()).  0 = the built-in verifier accepted.  */
int vph_prove_fs(vph_session *, uint8_t *proof, uint64_t capacity, uint64_t *n_written, vph_result *res, char *err, int errlen);
/* No GPU, no tape: re-derive the challenges from `proof` and run the verifier's checks.  0 = accepted, 1 = rejected.   */
int vph_verify_fs(vph_circuit *, const uint8_t *proof, uint64_t n);
/* vp_commit_stats: device milliseconds of the last commit_private / commit_public / FRI call on this session. */
double vph_commit_device_ms(vph_session *);
/* One proof over `world` GPUs (vp_set_shard): vph_prove_gkr on this session then proves only the sumchecks dealt to `rank`
 * and leaves the rest of the transcript zero; the u64 sum of all ranks' transcripts is the proof.      */
int vph_set_shard(vph_session *, int rank, int world);
/* Index-split proof without a communicator: V_u of the split phase-2 chains by a caller-side exchange (include/vpgpu.h: vp_shard_vu_partials /
 * vp_shard_vu_set) on the session's tape.  partials: 2 u64 per entry; returns the number of entries (0: nothing to exchange), < 0: refused.     */
int vph_shard_vu_partials(vph_session *, uint64_t *partials, int capacity);
int vph_shard_vu_set(vph_session *, const uint64_t *sums, int n);
/* vp_shard_chains: owner and cost estimate per sumcheck chain; returns the number of chains (< 0 on error). */
int vph_shard_chains(vph_session *, int32_t *owner, double *cost, int capacity);

#ifdef __cplusplus
}
#endif
#endif
