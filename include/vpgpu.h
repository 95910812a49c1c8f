/* vpgpu.h — C ABI of libvpgpu.so, the MI355X (gfx950) device side of the Virgo++ GKR prover.
 *
 * The reference (TAMUCrypto/virgo-plus) has no plugin/FFI layer: its seam is the C++ class `prover`
 * (src/prover.h:12-66) whose only caller is `verifier` (src/verifier.cpp:24,137,151-152,156,203,206,
 * 220,242,247,261,289,293,308,379).  This header is the FFI that class would bind if its hot loops
 * lived on a GPU: one entry point per prover method on the sumcheck path.  The host
 * mirror of the class that calls these functions is virgo-plus_amd/host/prover.{hpp,cpp}; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - C linkage, opaque context, plain pointers and sizes; no C++/torch types.
 *   - Every function returns VP_OK (0) or a negative VP_E* code; the library never exits the process
 *     (the reference calls exit(EXIT_FAILURE) on a violated assert gate, src/prover.cpp:18-21,211).
 *   - A field element F of F_p^2, p = 2^61-1, is two little-endian uint64 {real, img}, canonical
 *     (each < p), exactly the in-memory layout of virgo::fieldElement (lib/virgo/src/fieldElement.hpp:96-97).
 *   - All buffers passed in are caller-owned host memory and may be freed after the call returns.
 *   - One context = one GPU = one proof at a time (the reference prover is not re-entrant either).
 *     Call order is the reference's state machine (SURVEY.md §8b "Threading").
 *   - The library never consumes the caller's glibc random() / rand() stream: the reference verifier draws its challenges and query
 *     positions from it (lib/virgo/src/fieldElement.cpp:119-124, vpd_verifier.cpp:121), and the ROCm runtime takes draws of the same
 *     process-wide generator while it initialises, so vp_create*, vp_circuit_upload and vp_comm_* run on a private generator state
 *     (initstate / setstate) and restore the caller's.  Pinned by the reference binary of oracle/integration reproducing the CPU
 *     reference's transcript byte for byte.
 */
#ifndef VPGPU_H
#define VPGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp_ctx vp_ctx;
typedef struct { uint64_t real, img; } vp_F;

enum {
    VP_OK = 0,
    VP_EINVAL = -1,      /* bad argument / call out of order                       */
    VP_EHIP = -2,        /* HIP runtime error (vp_last_error has the string)        */
    VP_ENOGPU = -3,      /* no usable gfx950 device                                  */
    VP_EASSERT = -4,     /* an assert gate evaluated to non-zero (prover.cpp:18-21)  */
    VP_ELIMIT = -5,      /* circuit exceeds a compiled-in limit                      */
    VP_EXCHANGE = 1      /* sharded commitment without a communicator: the call stopped at a collective; run
                            vp_shard_exchange_local on all ranks' contexts, then call the same function again */
};

/* Gate types, same numbering as enum gateType (src/inputCircuit.hpp:13-15). */
enum { VP_MUL = 0, VP_ADD, VP_SUB, VP_ANTISUB, VP_NAAB, VP_ANTINAAB, VP_INPUT, VP_MULC, VP_ADDC, VP_XOR, VP_NOT, VP_COPY };

/* One circuit layer, structure-of-arrays view of `class layer` / `class gate` (src/circuit.h:11-35)
 * after layeredCircuit::subsetInit() (src/circuit.cpp:43-80).  Arrays have `size` entries; `c` and
 * `is_assert` may be NULL when no gate of the layer uses them.  For layer i the dad* arrays have i
 * entries; dad_id[j] has dad_size[j] entries.  An empty subset has dad_size[j] == 0 and dad_bitlen[j]
 * is then ignored (the reference stores (int)log2(0) there).                                           */
typedef struct {
    uint64_t size;
    int32_t bit_length;
    const uint8_t *ty;          /* gate::ty                                  */
    const int32_t *l;           /* gate::l  (-1 for unary gates)             */
    const uint32_t *u;          /* gate::u  (for VP_INPUT gates: ignored)    */
    const uint32_t *v;          /* gate::v                                   */
    const uint32_t *lv;         /* gate::lv                                  */
    const vp_F *c;              /* gate::c  (Addc / Mulc constants) or NULL  */
    const uint8_t *is_assert;   /* gate::is_assert or NULL                   */
    const uint64_t *dad_size;   /* layer::dadSize[j],      j < layer index   */
    const int32_t *dad_bitlen;  /* layer::dadBitLength[j]                    */
    const uint32_t *const *dad_id; /* layer::dadId[j][k]                     */
} vp_layer_desc;

/* ---- options -------------------------------------------------------------------------------------
 * What a caller can sensibly choose about HOW the library computes (never WHAT: every combination gives the same field elements; the parity tests run
 * the alternatives against the same golden transcripts).  vp_options_default fills the shipped configuration; vp_create_with_options copies the struct
 * into the context.  Layout rule (round 5): `abi` names the layout — a struct with another value, or a struct_size that is not this header's, is refused
 * with VP_EINVAL instead of being read field by field at the wrong offsets; new fields take the reserved slots, a field that goes away bumps VP_OPTIONS_ABI.
 * Everything else that steers the kernels — launch-plan layout, fold-kernel grids, which transform / hash / init kernel runs, the tests' cross-check
 * drivers — is INTERNAL tuning: the VP_* environment variables listed in DESIGN.md section 4 (read once per vp_create, test / bench / A-B use only) and
 * vp_tuning_get to read one back (what plan_autotune kept, say).                                                                                      */
#define VP_OPTIONS_ABI 0x76700005u
typedef struct {
    uint32_t struct_size;           /* sizeof(vp_options)                                                                               */
    uint32_t abi;                   /* VP_OPTIONS_ABI                                                                                   */
    int32_t use_graph;              /* VP_GKR_GRAPH: vp_prove_gkr replays its launch plan as a hipGraph                                              [1] */
    int32_t plan_autotune;          /* VP_PLAN_AUTOTUNE: the first vp_prove_gkr of a circuit replays the plan in the few layouts it can take and keeps the
                                       fastest (one-off: ~0.9 s at x1024; shared between contexts of one process with the same plan shape)          [1] */
    int32_t real_values;            /* VP_REAL_V: real x complex products wherever vp_evaluate found every circuit value real                        [1] */
    int32_t real_pairs;             /* VP_REAL_PAIRS: vp_commit_private of a real witness sends two slices through each transform as one complex sequence [1] */
    int32_t pc_tensor_pub;          /* VP_PC_TENSOR: vp_commit_public checks (exactly, on the device) whether the public vector is a tensor — the protocol's
                                       own eq(r, .) always is — and then encodes ONE slice instead of 64                                             [1] */
    int32_t persistent_rounds;      /* VP_PERSIST: interactive rounds that fit one CU are answered by a resident kernel through a pinned mailbox      [1] */
    int32_t persistent_timeout_ms;  /* VP_PERSIST_TIMEOUT_MS: how long that kernel waits for the next message before it saves its phase and leaves    [10000] */
    int32_t poll;                   /* VP_POLL: spin on pinned replies instead of hipStreamSynchronize (a core per waiting call)                      [1] */
    int32_t prefetch_round1;        /* VP_PREFETCH_R1: an interactive init call queues round 1 of its sumcheck and returns without waiting            [1] */
    int32_t interactive_fast_init;  /* VP_FAST_INIT: the interactive init calls run the batched path's init kernels                                   [1] */
    int32_t split_cost_percent;     /* VP_SPLIT_COST_PERCENT: vp_set_shard_split cuts a chain by index only above this share of a rank's fair load    [50] */
    int32_t debug;                  /* bit 0 (VP_DEBUG): diagnostics on stderr; bit 1 (VP_DEBUG_UPLOAD): phase times of vp_circuit_upload             [0] */
    int32_t reserved[4];            /* zero                                                                                                            */
} vp_options;
void vp_options_default(vp_options *opt);

/* ---- life cycle ---------------------------------------------------------------------------------- */
int vp_create(int device, vp_ctx **out);                                 /* = vp_create_with_options(device, NULL, out) */
int vp_create_with_options(int device, const vp_options *opt, vp_ctx **out);
int vp_get_options(const vp_ctx *, vp_options *out);                   /* the configuration in effect: defaults <- struct <- environment */
/* one internal tuning knob by the name DESIGN.md lists it under ("fuse_combine", "sf3b_grid", "gkr_path", ...), e.g. what plan_autotune kept after the
 * first vp_prove_gkr of a circuit.  VP_EINVAL for a name the library does not have.  Tests / benches only.                                           */
int vp_tuning_get(const vp_ctx *, const char *name, int32_t *value);
/* 1 when the library was built with -DVP_TEST_DRIVERS: the two extra drivers of vp_prove_gkr the tests cross-check the launch plan against (VP_GKR_PATH=lanes:
 * the plan's recorder run live, one stream per sumcheck chain; =simple: one launch per round through the interactive path's kernels).  The product library
 * (0) ships the launch plan alone and refuses VP_GKR_PATH.                                                                                            */
int vp_test_drivers(void);
/* The launch-plan layout plan_autotune kept for this circuit — v = { fuse_combine, fold_branches, plan_align, fuse_min_log, sf3b_grid, graph_explicit } — readable
 * after the first vp_prove_gkr, and the way to carry it to another process: vp_plan_tuning_set on a fresh context after vp_circuit_upload and BEFORE its first
 * proof; the tuner then has nothing left to try (first proof of a x1024 circuit 0.9 s -> the plan recording and one graph capture).  A layout tuned on another
 * circuit or device is still correct (every layout gives the same transcript), only not necessarily the fastest.  VP_EINVAL for values out of range or a
 * context that has already recorded its plan.                                                                                                         */
int vp_plan_tuning_get(const vp_ctx *, int32_t v[6]);
int vp_plan_tuning_set(vp_ctx *, const int32_t v[6]);
/* The same across processes without a line of caller code: with VP_PLAN_CACHE=<file> in the environment of vp_create, the tuner's choice of every plan shape
 * it meets is appended to that text file (one line: shape key, the six values above) and looked up there before tuning — the first proof of a circuit whose shape
 * is on file records one plan and captures one graph.  Lines the library does not accept are ignored; a missing or unwritable file means "no cache".        */
/* One-shot latency (round 6): the reference allocates the commitment's arrays and its FFT scratch at namespace scope / in init code, outside every prove timer
 * (lib/virgo/src/poly_commit.cpp:4-13, fri.cpp:13-34, RS_polynomial.cpp:9-16).  vp_warm(ctx, VP_WARM_COMMITMENT), after vp_evaluate and before the first
 * prover call, does the same here: the order-M root table, the transforms' circles and twist table, their scratch, every codeword / tree / FRI buffer, the fold
 * constants and a pinned staging area for vp_commit_public's vector (which is then uploaded in pieces at the bus's speed) exist when vp_commit_private starts.
 * Optional: every call still sets up what it finds missing.  VP_EINVAL before vp_evaluate.                                                              */
#define VP_WARM_COMMITMENT 1u
#define VP_WARM_FFT_GKR 2u          /* the ~20 device arrays of vp_fft_gkr at lg = input bit length - 6 (the size lib/virgo's verifier runs it at) */
int vp_warm(vp_ctx *, uint32_t what);
void vp_destroy(vp_ctx *);
const char *vp_last_error(const vp_ctx *);     /* static/ctx-owned string, never NULL */
const char *vp_version(void);
/* 1 when the library was built with -DVP_CHECKED: device-side index checks in the gather / scatter kernels (csrc/vp_check.h: operand and slot
 * gathers against the layer sizes, row pointers of the target-sorted lists, stores of folded entries against the table capacity, LDS slots of a
 * transform tile, codeword positions of a leaf).  A violated check never traps: the access is skipped, the FIRST violation is recorded on the device,
 * and the entry point during which it happened returns VP_EHIP with "device check failed: site N (...)".  A checked process holds one circuit at a
 * time.  The product library (0) compiles every check away.                                                                                    */
int vp_checked_build(void);

/* prover::prover(const layeredCircuit&) (src/prover.cpp:14) — the circuit is copied to HBM together
 * with the per-layer target-sorted gate indices the init kernels use.  May be called again to replace
 * the circuit.                                                                                         */
int vp_circuit_upload(vp_ctx *, int n_layers, const vp_layer_desc *layers);

/* prover::evaluate() (src/prover.cpp:27-91).  `inputs` are the layer-0 values (size of layer 0).
 * Returns VP_EASSERT if an assert gate is non-zero.                                                    */
int vp_evaluate(vp_ctx *, const vp_F *inputs, uint64_t n_inputs);
/* Copy circuitValue[layer] back (tests / commit_private hand-off).  n = layer size.                    */
int vp_layer_values(vp_ctx *, int layer, vp_F *out, uint64_t n);

/* prover::Vres(r_0, r_0_size) (src/prover.cpp:99-129): MLE of the output layer at r_0.                 */
int vp_vres(vp_ctx *, const vp_F *r_0, int r_0_size, vp_F *out);

/* ---- the drop-in (interactive) path: one call per prover method, one vp_round per verifier message -------------------
 * How it is served (vp_kernels_persist.h), none of which changes the call sequence or the values:
 *   * an init call returns as soon as its kernels are queued, together with round 1 of the sumcheck (which takes no challenge:
 *     `prefetch_round1`); the arrays passed in are copied before the call returns.  A failure of that queued work is reported by the
 *     first vp_round of the phase;
 *   * once every live table of the phase fits one CU's LDS, ONE resident kernel answers all remaining messages of the phase
 *     through a mailbox in pinned host memory (`persistent_rounds`): vp_round then costs a 3 us round trip plus the arithmetic
 *     instead of a kernel launch.  vp_finalize ends it.  Any other entry point called ON THE SAME CONTEXT in the middle of a sumcheck
 *     first tells the resident kernel to leave (the sumcheck is abandoned, the context stays usable).  It never stands in anybody's way
 *     for long (round 3): an entry point called on ANOTHER context of the process suspends it — the phase (LDS tables, lengths, round
 *     counter, add_term) is saved to device memory, the CU is released, device-wide synchronising HIP calls of that other context
 *     (hipMalloc / hipFree of an upload or a destroy) proceed at once — and left alone for persistent_timeout_ms it suspends itself the
 *     same way; this context's next vp_round / vp_finalize relaunches it on the saved phase and the sumcheck continues (bit-identical
 *     transcript).
 *     Threads: every entry point holds its context's lock for the duration of the call — calls on one context are serialised, calls on
 *     different contexts run concurrently (two proofs in flight from two threads).  An entry point that suspends another context's resident
 *     kernel waits for that context's current call to return first; the set-up entry points (vp_create*, vp_circuit_upload, vp_comm_*) take
 *     turns among themselves (they park the process-wide random() state).  While a thread is inside one of those set-up entry points NO other
 *     thread of the process may call random() / rand() / srandom(): glibc's generator state is process-wide, a draw made meanwhile comes from the
 *     library's private seed-1 state and is missing from the caller's stream afterwards (the reference's verifier is single-threaded and draws
 *     between prover calls, never during one).  vp_destroy of a context another thread is still calling into is the caller's bug, as is
 *     vp_shard_exchange_local on contexts other threads are using.
 * prover::sumcheckInitPhase1(assert_random) for layer `layer` (src/prover.cpp:189-280).  r_liu is the
 * point the layer's claim is at (bit_length(layer) entries; prover::r_liu in the reference).           */
int vp_phase1_init(vp_ctx *, int layer, const vp_F *r_liu, const vp_F *assert_random);
/* prover::sumcheckInitPhase2() (src/prover.cpp:282-367).  r_u = the bit_length(layer-1) challenges of
 * phase 1; V_u (set by sumcheckFinalize1) is taken from the device.                                    */
int vp_phase2_init(vp_ctx *, int layer, const vp_F *r_u);
/* prover::sumcheckInitLiu(s) (src/prover.cpp:369-420).  r_v[k] (k = layer..n_layers-1) points to the
 * phase-2 challenges of layer k (at least dad_bitlen[k][layer-1] entries, may be NULL if that subset
 * is empty); s has n_layers-layer+1 used entries.                                                      */
int vp_liu_init(vp_ctx *, int layer, const vp_F *r_u, const vp_F *const *r_v, const vp_F *s);

/* prover::sumcheckUpdatePhase1 / Phase2 / sumcheckLiuUpdate (src/prover.cpp:422-492): one sumcheck
 * round over all live tables of the current phase.  out_poly = {a, b, c} of a*x^2 + b*x + c.
 * Blocks until the round polynomial is on the host.  previous_random must be canonical (both limbs < 2^61 - 1): VP_EINVAL
 * otherwise, for vp_finalize too (the sumcheck in progress is not disturbed).                           */
int vp_round(vp_ctx *, const vp_F *previous_random, vp_F out_poly[3]);

/* prover::sumcheckFinalize1 / Finalize2 / sumcheckLiuFinalize (src/prover.cpp:494-521).  n_claims is
 * 1 for phase 1 and Liu, `layer` for phase 2.                                                          */
int vp_finalize(vp_ctx *, const vp_F *previous_random, vp_F *claims, int n_claims);

/* ---- batched mode -------------------------------------------------------------- */
/* The reference verifier's randomness does not depend on the transcript (plain random(),
 * lib/virgo/src/fieldElement.cpp:119-124), so the whole GKR part of verifier::verify()
 * (src/verifier.cpp:144-169) can run on the device without a host round trip per round.  `tape` holds
 * the verifier's draws in its own order:
 *   r_0[bl(out)] | for layer i = n-1..1: r_u[max_bl] assert_random[1] r_v[i][maxDadBl(i)] (if any)
 *                                        sig[n_layers] r_liu[max_bl]
 * `transcript` receives the GKR slice of the golden layout: Vres | per layer: phase-1
 * polys, claim_u, phase-2 polys, claims_v[0..i), Liu polys, vr.  Evaluate must have been called.      */
int vp_prove_gkr(vp_ctx *, const vp_F *tape, uint64_t n_tape, uint8_t *transcript, uint64_t capacity,
                 uint64_t *n_written);
/* Number of tape entries / transcript bytes vp_prove_gkr needs for the uploaded circuit.               */
int vp_gkr_sizes(vp_ctx *, uint64_t *n_tape, uint64_t *n_transcript_bytes);

/* One proof sharded over the GPUs of a node ("independent sumcheck instances shard across the GPUs").  Given
 * the tape, every sumcheck of the proof — phase 1, phase 2 and the Liu sumcheck of each layer (src/verifier.cpp:191-337) and
 * Vres (:151) — is independent of the others and writes its own slice of the transcript.  After vp_set_shard(rank, world),
 * vp_prove_gkr on this context runs only the sumchecks dealt to `rank` (longest-processing-time greedy on table sizes,
 * the same assignment on every rank) and leaves all other transcript bytes ZERO: the element-wise u64 sum of the `world`
 * transcripts (one all-reduce over RCCL; the slices are disjoint, so the sum is exact) is the transcript of the unsharded
 * proof, byte for byte.  Every rank holds the whole circuit and witness.  world = 1 restores the unsharded proof.  Only
 * the batched entry point shards; the interactive entry points are unaffected.                                         */
int vp_set_shard(vp_ctx *, int rank, int world);
/* On top of vp_set_shard: tables of at least 2^(log2 W' + min_log) entries (W' = the largest power of two <= world) are cut into W' slices by
 * index and slice s is folded by rank s through the rounds that stay inside a slice; the entries the slices end in travel in an export area
 * behind the transcript (the same u64-sum all-reduce gathers them) and the last log2 W' rounds of those tables are finished on the host.
 * With a communicator attached vp_prove_gkr returns the finished transcript as before.  Without one it returns the rank's partial sums AND its
 * export area (vp_gkr_sizes gives the size): add the ranks' buffers as u64 and call vp_shard_finish.  min_log = 0 switches the split off;
 * call after vp_set_shard.  Chains without a long enough table are dealt out whole, as before.
 * LIMIT: W' <= 8 (world < 16), VP_ELIMIT otherwise — the W' ranks of a split chain add partial sums < 2^61 into the same u64 slots before
 * anything is reduced mod p; eight addends fit in 64 bits, sixteen can wrap (and 2^64 = 8 mod p would be a silently wrong transcript).  The
 * same bound holds for the caller-side u64 sum handed to vp_shard_finish.  vp_set_shard alone (disjoint slices) has no such bound.           */
int vp_set_shard_split(vp_ctx *, int min_log);
int vp_shard_finish(vp_ctx *, uint8_t *summed, uint64_t n_bytes, uint64_t *n_transcript_bytes);
/* V_u of the index-split phase-2 chains (src/prover.cpp:494-500: what phase 1's last fold leaves in V, = <eq(r_u, .), V_{i-1}>).  It depends on
 * the tape and the witness only, so it is taken ahead of the proof: every rank adds up its share of the previous layer (1 / W' of it), ONE exchange
 * of 16 bytes per split phase-2 chain completes the sums, and no rank reads a whole layer for a V_u of its own.
 *   with a communicator attached: vp_prove_gkr does all of it (one extra all-reduce of *n * 16 bytes before its graph);
 *   without one:  vp_shard_vu_partials(ctx, tape, n_tape, partials, capacity, &n)   this rank's n partial sums (n may be 0: nothing to exchange),
 *                 [the caller adds the ranks' arrays up as u64 — at most 8 addends < 2^61 each, as for vp_shard_finish]
 *                 vp_shard_vu_set(ctx, sums, n)                                    on every rank, then vp_prove_gkr with the SAME tape;
 *   a vp_prove_gkr without either adds up the whole layer on every rank of the chain (correct, slower: the round-4 behaviour).
 * vp_stats' gkr_device_ms of the proof that follows includes the device time of vp_shard_vu_partials.                                          */
int vp_shard_vu_partials(vp_ctx *, const vp_F *tape, uint64_t n_tape, vp_F *partials, uint64_t capacity, uint64_t *n);
int vp_shard_vu_set(vp_ctx *, const vp_F *sums, uint64_t n);
/* The assignment: owner rank and cost estimate of every chain, in the order phase-1(layer 1), Liu(layer 1), phase-1(layer 2),
 * Liu(layer 2), ... then phase-2(layer 1..n-1), then Vres; *n_chains = 3*(n_layers-1) + 1.  Chains that do not exist (layers
 * without a phase 2) have cost 0.                                                                                      */
int vp_shard_chains(vp_ctx *, int32_t *owner, double *cost, int capacity, int *n_chains);

/* ---- Virgo polynomial commitment, commit side ------------------------------------------------------ */
/* prover::commit_private() (src/prover.cpp:524-530 -> poly_commit_prover::commit_private_array,
 * lib/virgo/src/poly_commit.h:41-124 -> fri::request_init_commit, fri.cpp:36-139 -> create_tree,
 * merkle_tree.cpp:7-51): Reed-Solomon encode (rate 1/32) the 64 slices of the input layer, hash the leaf
 * chains with SHA3-256 and build the Merkle tree; returns the root (merkle_root_l).  The codeword and the
 * tree stay in HBM for the later openings.  Needs bit_length(layer 0) >= 7 (vpd_verifier.cpp:115);
 * VP_ELIMIT if a slice is longer than 2^17 elements (input layer of more than 2^23 wires): transforms up to 2^13 run
 * in LDS, longer ones through a register split in front of it.                                         */
int vp_commit_private(vp_ctx *, uint8_t root[32]);
/* prover::commit_public(pub, inner_product_sum, mask, all_sum) (src/prover.cpp:542-546 ->
 * poly_commit_prover::commit_public_array, poly_commit.h:126-349 -> fri::request_init_commit(.., 1)):
 * `pub` has 2^bit_length(layer 0) entries (the verifier's eq table, src/verifier.cpp:368-369).  Outputs
 * the inner product <circuitValue[0], pub> (input_0), all_sum[0..65) and the Merkle root of the quotient
 * codewords h (merkle_root_h).  vp_commit_private must have run.                                       */
int vp_commit_public(vp_ctx *, const vp_F *pub, uint64_t n_pub, vp_F *inner_product_sum, vp_F all_sum[65],
                     uint8_t root_h[32]);
/* The same call for the public vector the PROTOCOL passes (src/verifier.cpp:368-369: `initBetaTable(pub, bit_length, r_liu, 1)` right before
 * `p->commit_public(pub, ...)`): the caller hands over the opening point (bit_length(layer 0) canonical coordinates) instead of its 2^n-entry
 * eq table, and the device builds eq(point, .) in HBM itself (src/utils.cpp:29-45) — nothing of the public vector crosses PCIe (134 MB at
 * n = 23).  Same outputs, same field elements as vp_commit_public(eq table of the point); an eq table is a tensor by construction, so the one-
 * slice encoding (pc_tensor_pub) applies without its check.  VP_EINVAL on a sharded commitment.                                       */
int vp_commit_public_eq(vp_ctx *, const vp_F *point, int n_point, vp_F *inner_product_sum, vp_F all_sum[65], uint8_t root_h[32]);
/* fri::commit_phase_step(r) (lib/virgo/src/fri.cpp:289-424), called n-6 times by poly_commit_prover::commit_phase
 * (vpd_verifier.cpp:44-74): fold the current codewords of all slices by r, hash the new leaves, build the Merkle
 * tree, return its root.  The first call builds the virtual oracle (poly_commit.h:294-318) from the data
 * vp_commit_public left in HBM.  VP_EINVAL once the codeword is down to 32 values per slice.            */
int vp_fri_step(vp_ctx *, const vp_F *r, uint8_t root[32]);
/* The same n_steps calls of fri::commit_phase_step in ONE device pass (extension, like vp_prove_gkr: valid because the
 * reference verifier's challenges do not depend on the transcript, fieldElement.cpp:119-124): r[0..n_steps) in, the
 * n_steps Merkle roots out (32 bytes each, in step order).  Folds run back to back; the leaves of all levels are hashed
 * by one launch and the trees are built level by level across all of them.  Must start from a fresh vp_commit_public
 * (not after vp_fri_step); afterwards vp_fri_final / vp_fri_open behave as after n_steps vp_fri_step calls.          */
int vp_fri_commit(vp_ctx *, const vp_F *r, int n_steps, uint8_t *roots);
/* fri::commit_phase_final() (fri.cpp:426-431): the last codeword, 2048 elements in the reference's interleaved
 * layout [i << 7 | slice << 1 | hi], i < 16.                                                            */
int vp_fri_final(vp_ctx *, vp_F *final_code);
/* fri::request_init_value_with_merkle (fri.cpp:148-205; oracle 0 = l, 1 = h) and fri::request_step_commit
 * (fri.cpp:229-287; oracle 2 + level): open leaf `leaf` of a committed oracle.  values: 65 pairs (64 slices, then the
 * mask pair) = the two codeword entries of the leaf per slice; path: depth+1 digests, path[k] = sibling at height k,
 * path[depth] = the leaf digest (the reference's com_hhash layout).  *path_len receives depth + 1.           */
int vp_fri_open(vp_ctx *, int oracle, uint64_t leaf, vp_F values[130], uint8_t *path, int path_capacity, int *path_len);
/* fft_circuit_gkr::fft_gkr(lg) (lib/virgo/src/fft_circuit_GKR.cpp:833-849), prover side: the self-contained GKR over the inverse-FFT +
 * polynomial-evaluation circuit that verify_poly_commitment runs between commit_public and the FRI commit phase (vpd_verifier.cpp:92,
 * lg = bit_length(layer 0) - 6) and whose prover time the reference adds to "Polynomial commitment: prove time" (:94, src/verifier.cpp:183).
 * Like vp_prove_gkr it takes the verifier's whole tape up front (its draws are glibc random(), transcript-independent) — in the
 * reference's draw order:  r[lg] (:840) | eval_points[64] (:84) | r_0[lg+10] | r_1[lg+10] (:789-790) | addition layer r_u[lg+6], r_v[lg+6]
 * (:275-276) | multiplication layer r_u[lg], r_v[lg] (:394-395) | per inverse-FFT depth (lg of them): r_u[lg], r_v[lg] (:563-564), alpha, beta
 * (:763-764) — and returns every prover message:  the circuit's 64 outputs | addition layer: lg+6 round polynomials (a, b, c), v_u |
 * multiplication layer: lg polynomials, v_u | per depth: lg polynomials, v_u, lg polynomials, v_v.  The checks of the reference's embedded
 * verifier (:261-266, :285-308, :405-446, :639-752) stay on the host (virgo-plus_amd/host/fft_gkr_verify.hpp).  No circuit needs to be
 * uploaded.  vp_fft_gkr_sizes gives the two element counts (2 lg^2 + 9 lg + 96 and 64 + 3 (2 lg^2 + 2 lg + 6) + 2 + 2 lg); lg in 1..20.        */
int vp_fft_gkr_sizes(int lg, uint64_t *n_tape, uint64_t *n_msgs);
int vp_fft_gkr(vp_ctx *, int lg, const vp_F *tape, uint64_t n_tape, vp_F *msgs, uint64_t capacity, uint64_t *n_written);
/* The same call in two halves: vp_fft_gkr_begin copies the tape and queues the whole pass on a stream of its own, vp_fft_gkr_end waits for it and
 * hands out the messages.  fft_gkr reads nothing of the circuit or the commitment (its circuit is fixed by lg, its inputs are the verifier's
 * draws), so a prover that has the tape up front starts it first and lets its ~250 small launches fill the gaps of the commitment's and the
 * proof's kernels (vph_prove_protocol does).  Between the two calls every other entry point of the context may be used; a second begin, or the
 * one-call form, before the end is VP_EINVAL.                                                                                             */
int vp_fft_gkr_begin(vp_ctx *, int lg, const vp_F *tape, uint64_t n_tape);
int vp_fft_gkr_end(vp_ctx *, vp_F *msgs, uint64_t capacity, uint64_t *n_written);
/* Drop a run begun with vp_fft_gkr_begin without reading its messages (the caller's pass failed between begin and end); VP_OK when none is pending.
 * vp_fft_gkr_begin itself drains and drops a run that was never collected, so a failed pass cannot wedge the context. */
int vp_fft_gkr_cancel(vp_ctx *);
/* ---- the mask slice with CONTENT (round 6) ------------------------------------------------------------------------------------------------
 * lib/virgo's commit_private_array / commit_public_array take a mask vector that fills the 65th slice (lib/virgo/src/poly_commit.h:42,55-86,138-161,187-247).
 * The reference's own prover and verifier only ever pass one zero (src/prover.cpp:526, src/verifier.cpp:375-377): vp_commit_private / vp_commit_public.  These
 * two take the vectors: the private mask (n_mask elements) is padded with zeros to ms = slice_size / gap elements, gap = the largest power of two <=
 * slice_size / n_mask (poly_commit.h:55-66); the public mask pads to the same ms.  From then on the commitment carries the slice everywhere the reference does —
 * the 65th block of every leaf chain of both oracles, all_sum[64], its own virtual oracle and fold on every FRI level (vp_fri_step / vp_fri_commit), the last pair
 * of every opening (vp_fri_open), its final codeword (vp_fri_final_mask) — bit-exact against the reference called directly (tests/golden/pc_masked_*.bin).
 * Limits: ms >= 8 (below that the reference's own transforms read stale scratch, RS_polynomial.cpp:104-133: VP_EINVAL), gap >= 2 (the reference asserts it,
 * poly_commit.h:195: VP_EINVAL) and ms <= 2^16 (the quotient's 2 ms-point transform must be one of this library's: VP_ELIMIT); not on a sharded commitment;
 * vp_commit_public_eq refuses a masked commitment.  vp_commit_private (or a new witness) returns the context to the zero mask.
 * What the reference's VERIFIER makes of it (vpd_verifier.cpp:76-328 with the public mask, oracle/integration/masked_main.cpp): a commitment whose ms is at most a
 * slice's message length 2^(n-6) is accepted; a longer mask is committed exactly as the reference's prover commits it and rejected exactly as the reference's verifier
 * rejects that prover's (its last check, :318-324: n - 6 folds do not reduce the mask slice to a constant).                                                        */
int vp_commit_private_masked(vp_ctx *, const vp_F *mask, uint64_t n_mask, uint8_t root[32]);
int vp_commit_public_masked(vp_ctx *, const vp_F *pub, uint64_t n_pub, const vp_F *pub_mask, uint64_t n_pub_mask, vp_F *inner, vp_F all_sum[65], uint8_t root_h[32]);
/* fri::cpd.rs_codeword_msk[last] (vpd_verifier.cpp:321-325): the mask slice's last codeword, 32 values, out[2 i + hi] = value at position i + 16 hi (zeros
 * for the zero mask).  After the last FRI step.                                                                                                              */
int vp_fri_final_mask(vp_ctx *, vp_F out[32]);
/* Device time of the last vp_commit_private / vp_commit_public / vp_fri_step / vp_fft_gkr in milliseconds (hipEvents). */
int vp_commit_stats(vp_ctx *, double *commit_ms);

/* ---- deferred completion: a prover pass without a host wait between its calls (round 5) -------------------------------------------------
 * Measured on MI355X (tools/leaf_in_step.py, profiles/r05_leaf_hash_in_step.txt): a GPU that idles for a fraction of a millisecond runs the next ~10 ms at a
 * lower shader clock (the leaf-hash launch, same cycles per workgroup: 10.1 ms back to back, 10.5 ms behind 0.5 ms of idling, 11.7 ms behind 5 ms).  Each
 * synchronous entry point ends in such a gap: the host wakes up, copies the result, makes the next call.  With vp_set_deferred(ctx, 1) the entry points of the
 * reference's prover pass whose results the HOST does not need before the next call —
 *     vp_commit_private, vp_prove_gkr (launch plan, unsharded, not profiled), vp_commit_public_eq, vp_fri_commit, vp_fri_final
 * — queue their launches, stage their results in pinned memory and return VP_OK at once; the output pointers are written by vp_flush, which waits for
 * the first `count` pending calls (count < 0: all) in the order they were made and returns the first error (the calls behind a failed one are dropped).
 * Output buffers and *n_written must stay valid until then.  Every other entry point of the context finishes what is pending before it runs, so a caller
 * that never calls vp_flush sees the synchronous behaviour one call late at worst (vp_set_deferred(ctx, 0) itself leaves what is pending pending).  Stream order is the order of the calls: a vp_commit_private queued
 * BEHIND a proof's vp_fri_commit (the head of the next proof, so that the device never idles between two proofs) starts when the folds are done and
 * overwrites the committed codeword — vp_fri_open answers for a proof only until the next vp_commit_private is queued.                               */
int vp_set_deferred(vp_ctx *, int on);
int vp_flush(vp_ctx *, int count);
int vp_pending(vp_ctx *, int *n);
/* device time in ms of the last finished vp_commit_private | vp_prove_gkr | vp_commit_public(_eq) | vp_fri_commit | vp_fri_final of the context */
int vp_phase_ms(vp_ctx *, double out[5]);
/* how many vp_commit_private calls the context has queued so far, and whether the latest one still stands (no upload / evaluate / vp_pc_load_input since) */
int vp_commit_private_state(vp_ctx *, uint64_t *epoch, int *valid);

/* ---- commitment sharded over the GPUs of a node (SURVEY.md §8e "PC sharding"; north_star "FFT subtrees shard") ---------------- */
/* After vp_pc_set_shard(rank, world) (world a power of two <= 64 with 2^(n-6) >= 2 world) the SAME entry points vp_commit_private /
 * vp_commit_public / vp_fri_commit / vp_fri_final / vp_fri_open work on this rank's share of the commitment: rank r transforms
 * slices [64 r / world, 64 (r+1) / world) (the 64 slices are independent transforms, lib/virgo/src/poly_commit.h:89-107), ONE
 * all-to-all per committed oracle hands every rank the positions a = rank (mod world) of ALL slices (a leaf chains all 64 slices at a
 * position pair, lib/virgo/src/fri.cpp:81-124), the rank hashes its leaves and five tree levels, the level-5 nodes are all-gathered and
 * every rank builds the top of the tree (merkle_tree.cpp:7-51): all ranks return the same root, equal to the unsharded one.  FRI folds
 * stay local (fold partners a, a + N_k/2 share their low bits) until one position per rank is left.  Every rank passes the same `pub`
 * / `r`; vp_fri_commit must be given all n-6 challenges.  vp_fri_open is answered by the owner of the leaf, rank (leaf >> 5) mod world
 * (VP_EINVAL elsewhere); levels of the last log2(world) folds are replicated.
 * Collectives: over RCCL when vp_comm_init attached a communicator; otherwise a call returns VP_EXCHANGE at each collective and
 * vp_shard_exchange_local(ctxs, world) performs the pending ones among contexts of ONE process (parity tests: W ranks on one GPU).
 * vp_pc_load_input makes a context that holds only an input layer (no circuit): enough for the commitment calls.                  */
int vp_pc_load_input(vp_ctx *, const vp_F *inputs, uint64_t n_inputs, int bit_length);
int vp_pc_set_shard(vp_ctx *, int rank, int world);
int vp_shard_exchange_local(vp_ctx **ctxs, int world);
/* The same pending collectives through a CALLER-SUPPLIED transport — ranks in different processes without RCCL (more ranks than GPUs in a
 * rehearsal, CPU tests over gloo): after VP_EXCHANGE, for i < *n: vp_shard_exchange_info gives kind (1 all-to-all, 2 all-gather) and bytes (per
 * peer / per rank); vp_shard_exchange_get copies the send side to the host (kind 1: world x bytes, peer-major; kind 2: bytes), the caller moves
 * the data, vp_shard_exchange_put copies the world x bytes received back; vp_shard_exchange_done, then call the interrupted function again.   */
int vp_shard_pending(vp_ctx *, int *n);
int vp_shard_exchange_info(vp_ctx *, int i, int *kind, uint64_t *bytes);
int vp_shard_exchange_get(vp_ctx *, int i, void *host_send);
int vp_shard_exchange_put(vp_ctx *, int i, const void *host_recv);
int vp_shard_exchange_done(vp_ctx *);
/* RCCL over xGMI: rank 0 makes the id (128 bytes, ncclUniqueId), every rank calls vp_comm_init with it.  With a communicator attached
 * a chain-sharded vp_prove_gkr (vp_set_shard) all-reduces the transcript on the device (u64 sum of disjoint slices) before returning
 * it, and the sharded commitment calls run their collectives inside the call.  librccl.so.1 is resolved at run time.
 * The communicator's (rank, world) must equal vp_set_shard's / vp_pc_set_shard's (VP_EINVAL from whichever call comes second): with a
 * duplicate shard rank a slice would be summed twice.  Every fallible set-up step of a sharded vp_prove_gkr (plan build, graph capture,
 * capacity check) happens BEFORE the collective; a rank that returns an error there never enters it while its peers wait in
 * ncclAllReduce — after an error on any rank the communicator must be torn down (vp_comm_destroy) on all of them.                   */
int vp_comm_unique_id(uint8_t id[128]);
int vp_comm_init(vp_ctx *, const uint8_t id[128], int rank, int world);
int vp_comm_destroy(vp_ctx *);
/* ncclCommCount of the attached communicator: the number of ranks RCCL itself reports (a benchmark prints it beside its own world size). */
int vp_comm_count(vp_ctx *, int *n_ranks);
int vp_allreduce_u64(vp_ctx *, void *device_buffer, uint64_t count);

/* ---- verifier side ------------------------------------------------------------------------------- */
/* The verifier's O(|C|) wiring-predicate loops for one layer (verifier::betaInitPhase1/2, predicatePhase1/2,
 * src/verifier.cpp:50-113) on the uploaded circuit: r_g = r_liu (bit_length(layer) entries, beta_g; assert gates scaled by
 * assert_random), r_u (bit_length(layer-1) entries), r_v (n_v = maxDadBitLength(layer) entries, 0 when the layer has no
 * phase 2).  out[0..5) = coeff_l[Copy], coeff_l[Not], coeff_l[Addc], coeff_l[Mulc], bias — WITHOUT the factor beta_v[0]
 * that predicatePhase2 applies (verifier.cpp:95-96); out[5 + t*layer + l] = coeff_r[type t][l], t in the order Add, Sub,
 * AntiSub, Mul, Naab, AntiNaab, Xor, l < layer.  n_out must be 5 + 7*layer.                                            */
int vp_predicates(vp_ctx *, int layer, const vp_F *r_g, const vp_F *assert_random, const vp_F *r_u, const vp_F *r_v, int n_v,
                  vp_F *out, uint64_t n_out);

/* The verifier's two other O(|C|) loops.  vp_liu_gr: the `gr` of verifier::verifyLiu (src/verifier.cpp:311-323) =
 * sum_u mult_Liu[u] * eq(r_liu, u), with mult_Liu the table prover::sumcheckInitLiu builds from (r_u, r_v[], s) — same
 * arguments as vp_liu_init — and r_liu the bit_length(layer-1) challenges of that Liu sumcheck.  vp_layer_mle: the
 * multilinear extension of a layer's values at r (layer 0: the input check of verifier.cpp:363-389 when the polynomial
 * commitment is off).  Both leave no sumcheck in progress.                                                          */
int vp_liu_gr(vp_ctx *, int layer, const vp_F *r_u, const vp_F *const *r_v, const vp_F *s, const vp_F *r_liu, vp_F *out);
int vp_layer_mle(vp_ctx *, int layer, const vp_F *r, int n, vp_F *out);

/* ---- measurement --------------------------------------------------------------------------------- */
typedef struct {
    double gkr_ms;            /* device time of the last vp_prove_gkr (hipEvents on the library stream)     */
    double evaluate_ms;       /* device time of the last vp_evaluate                                        */
    double fold_ms;           /* sum of durations of the event-bracketed dominant-kernel launches           */
    uint64_t fold_launches;   /* how many launches were bracketed                                           */
    uint64_t fold_bytes;      /* their algorithmic bytes (48*(L_in+L_out) per round)         */
    uint64_t rounds;          /* sumcheck rounds executed                                                   */
    uint64_t launches;        /* kernel launches of the last proof                                          */
} vp_stats;
int vp_get_stats(vp_ctx *, vp_stats *out);
/* The interactive path, per sumcheck round (north_star: "achieved HBM-bandwidth fraction reported per sumcheck round"): one entry per
 * vp_round since the last vp_vres.  bytes = the round's ALGORITHMIC bytes by SURVEY.md §8d: 48 B x (L_in + L_out) summed over the live
 * table families of the phase (32 B in the Liu phase, which has no add table); round 1 reads L_in = valid entries and writes nothing, round
 * k >= 2 reads ceil(valid / 2^(k-2)) and writes ceil(valid / 2^(k-1)) (fold by r_(k-1) fused with the sums of round k).  us = wall time of
 * the vp_round call as the caller sees it (launch or mailbox round trip + arithmetic + reply).  how: 0 one launch for the round, 1 answered
 * by the resident kernel through the pinned mailbox, 2 round 1 computed behind the init call (the call only collects it).            */
typedef struct {
    int32_t phase, layer, round, how, tables;
    uint64_t bytes;
    double us;
} vp_round_stat;
int vp_get_round_stats(vp_ctx *, vp_round_stat *out, int capacity, int *n);
/* How many times this context's resident round kernel has been relaunched on a saved phase (suspended by another context's call, or by its
 * own time-out) since vp_create. */
int vp_get_resident_resumes(const vp_ctx *, uint64_t *n);
/* 0: no per-kernel events (default); 1: the next vp_prove_gkr replays its launch plan on ONE stream and brackets EVERY
 * launch with hipEvents (in the default run the launches of independent sumchecks overlap on several streams, so
 * per-kernel times would be meaningless there); vp_commit_private / vp_commit_public / vp_fri_commit bracket their
 * launches as well.  The table of the last profiled call is read with vp_get_launch_stats.                          */
int vp_set_profiling(vp_ctx *, int level);

/* One kernel launch of the last profiled call: which kernel, how much it covered, its ALGORITHMIC bytes (SURVEY.md §8d:
 * fold launch = 48 B (32 B in the Liu phase) x (valid entries in + entries out); init = 26 B per gate contribution (10 B
 * record + 16 B gathered operand; 5 B in the Liu gather — stricter than SURVEY's 48 B: the eq value is a product of two
 * L2-resident half-table entries, not HBM traffic) + 4 B row pointer + the tables written; NTT = 16 B x (in + out) per pass,
 * FRI fold = 48 B per output element, leaf hash = 32 B per chained block) and its duration.  `work` counts the kernel's
 * own unit: F_p^2 multiplications for the transforms, Keccak-f[1600] permutations for the hash kernels, active pairs for
 * the fold kernels, gate contributions for the init kernels.  `rounds`: sumcheck rounds the launch performs (fold: 3 per
 * launch on every table it holds; closing kernels: the remaining ones), `first_round` the earliest of them (1-based).   */
enum { VP_K_BETA = 0, VP_K_LIGHT, VP_K_CHUNKS, VP_K_COMBINE, VP_K_DOT, VP_K_DOTFIN, VP_K_SFGEN, VP_K_SF, VP_K_SEG, VP_K_EMIT,
       VP_K_FIXUP, VP_K_NTT_SPLIT, VP_K_NTT_LDS, VP_K_NTT_UNSPLIT, VP_K_LEAF_HASH, VP_K_MERKLE, VP_K_PC_POINTWISE, VP_K_FRI_FOLD,
       VP_K_ROUND, VP_K_NTT8_COLS, VP_K_NTT8_ROWS, VP_K_COUNT };
typedef struct {
    int32_t kind;             /* VP_K_*                                                          */
    int32_t step;             /* position in the launch order of the call                        */
    uint32_t workgroups;      /* grid size                                                       */
    uint32_t jobs;            /* independent sumchecks / transforms / trees the launch batches   */
    uint32_t rounds, first_round;
    uint64_t bytes;           /* algorithmic bytes                                               */
    uint64_t work;            /* kind-specific unit count (see above)                            */
    double us;                /* hipEvent duration in microseconds                               */
} vp_launch_stat;
/* Copies up to `capacity` entries; *n receives the number available.                                                */
int vp_get_launch_stats(vp_ctx *, vp_launch_stat *out, int capacity, int *n);
const char *vp_kernel_name(int kind);

/* ---- primitives exported for parity tests (thin wrappers over the device functions) -------------- */
/* out[i] = a[i] op b[i], op: 0 add, 1 sub, 2 mul (device arithmetic of fieldElement.cpp:34-104); op 3: out[i] = a[i] b[i] + a[i+1] b[i+1] (cyclic) through the
 * one-reduction two-product form the FRI fold kernel uses (canonical inputs).                           */
int vp_test_field(vp_ctx *, int op, const vp_F *a, const vp_F *b, vp_F *out, uint64_t n);
/* initBetaTable(out, n, r, init) (src/utils.cpp:29-45): out has 2^n entries.                           */
int vp_test_beta(vp_ctx *, const vp_F *r, int n, const vp_F *init, vp_F *out);
/* out[i] = SHA3-256(in[i]) for n 64-byte messages (my_hhash, lib/virgo/src/my_hhash.h:27-33).          */
int vp_test_sha3(vp_ctx *, const uint8_t *in, uint8_t *out, uint64_t n);
/* fast_fourier_transform(coefs, coef_len, order) / inverse_fast_fourier_transform(evals, n, n)
 * (RS_polynomial.cpp:26-220), natural order in and out.  order/coef_len must be 1 or 32 for the forward
 * transform (the two shapes the commitment uses); sizes up to 2^17 per transform.                      */
int vp_test_fft(vp_ctx *, const vp_F *coefs, int coef_len, int order, int inverse, vp_F *out);

#ifdef __cplusplus
}
#endif
#endif
