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
// The polynomial commitment runs on the device as well: commit_private / commit_public below forward to vp_commit_private /
// vp_commit_public, and the `fri::` functions lib/virgo's verifier calls back into (vpd_verifier.cpp:44-97,152-167,282,309-324) are served
// by oracle/integration/fri_vpgpu.cpp, which replaces lib/virgo/src/fri.cpp in the link.  `poly_prover` (the public member verifier.cpp
// hands to lib/virgo's verifier) is the reference's own object; only its total_time is written here.
#include "prover.h"                 // the reference's, -I$(REF)/src
#include "vpgpu_glue.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

static inline const vp_F *cF(const F *p) { return reinterpret_cast<const vp_F *>(p); }
static inline vp_F *mF(F *p) { return reinterpret_cast<vp_F *>(p); }
static vp_ctx *g_ctx = nullptr;
vpi_counters g_vpi_count;
vpi_seconds g_vpi_sec;
void vpi_must(int rc, const char *what) {
    if (rc == VP_OK) return;
    fprintf(stderr, "vpgpu: %s failed (%d): %s\n", what, rc, g_ctx ? vp_last_error(g_ctx) : "no context");
    exit(EXIT_FAILURE);
}
static void must(int rc, const char *what) { vpi_must(rc, what); }
// see vpgpu_glue.h: the caller's random()/rand() stream is not ours to consume
vpi_rand_guard::vpi_rand_guard(const char *w) : what(w) {
    memset(priv, 0, sizeof priv);
    caller_state = initstate(1u, priv, sizeof priv);              // switches to the private state, returns the caller's
}
vpi_rand_guard::~vpi_rand_guard() {
    char fresh[128], *mine;
    memset(fresh, 0, sizeof fresh);
    mine = initstate(1u, fresh, sizeof fresh);                     // `mine` == priv as the call left it; `fresh` = the same seed untouched
    if (memcmp(mine, fresh, sizeof fresh) != 0) {
        ++g_vpi_count.rand_consumers;
        const char *t = getenv("VPI_TRACE");
        if (t && *t && *t != '0') fprintf(stderr, "vpgpu glue: %s consumed draws of the process's random() generator (kept away from the caller's stream)\n", what);
    }
    setstate(caller_state);
}
#define GUARDED(call, what) do { vpi_rand_guard guard_(what); must((call), what); } while (0)
vp_ctx *vpi_ctx() {
    if (!g_ctx) { fprintf(stderr, "vpgpu: no device context (prover not constructed)\n"); exit(EXIT_FAILURE); }
    return g_ctx;
}
// evidence for the parity tests (vpgpu_glue.h): message dump in the golden layout, FRI dump, call counters
static FILE *env_file(const char *name) {
    const char *p = getenv(name);
    if (!p || !*p) return nullptr;
    FILE *f = fopen(p, "wb");
    if (!f) { perror(p); exit(EXIT_FAILURE); }
    return f;
}
FILE *vpi_dump_file() { static FILE *f = env_file("VPI_DUMP"); return f; }
FILE *vpi_dump_fri_file() { static FILE *f = env_file("VPI_DUMP_FRI"); return f; }
FILE *vpi_dump_fft_file() { static FILE *f = env_file("VPI_DUMP_FFT"); return f; }
static void dumpF(const F &x) { if (FILE *f = vpi_dump_file()) { unsigned long long w[2] = {x.real, x.img}; fwrite(w, 8, 2, f); fflush(f); } }
static void dumpH(const void *d) { if (FILE *f = vpi_dump_file()) { fwrite(d, 32, 1, f); fflush(f); } }
static void trace_at_exit() {
    const char *t = getenv("VPI_TRACE");
    if (!t || !*t || *t == '0') return;
    fprintf(stderr, "vpgpu calls: commit_private %lu commit_public %lu fri_step %lu fri_final %lu open_init %lu open_step %lu round %lu finalize %lu rand_consumers %lu fft_gkr %lu\n",
            g_vpi_count.commit_private, g_vpi_count.commit_public, g_vpi_count.fri_step, g_vpi_count.fri_final, g_vpi_count.open_init,
            g_vpi_count.open_step, g_vpi_count.round, g_vpi_count.finalize, g_vpi_count.rand_consumers, g_vpi_count.fft_gkr);
    fprintf(stderr, "vpgpu seconds inside device calls: commit_private %.4f commit_public %.4f fri_step (all) %.4f (first %.4f) fri_final %.4f fft_gkr %.4f\n",
            g_vpi_sec.commit_private, g_vpi_sec.commit_public, g_vpi_sec.fri_step, g_vpi_sec.first_fri_step, g_vpi_sec.fri_final, g_vpi_sec.fft_gkr);
}
static_assert(sizeof(F) == sizeof(vp_F), "virgo::fieldElement is two u64 limbs (fieldElement.hpp:96-97)");
// a context without a circuit, for a test main that drives the commitment alone (integration/masked_main.cpp)
vp_ctx *vpi_ctx_standalone() {
    if (!g_ctx) { GUARDED(vp_create(0, &g_ctx), "vp_create"); atexit(trace_at_exit); }
    return g_ctx;
}

prover::prover(const layeredCircuit &cir) : C(cir) {              // src/prover.cpp:14
    proof_size = 0;
    GUARDED(vp_create(0, &g_ctx), "vp_create");
    atexit(trace_at_exit);
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
    GUARDED(vp_circuit_upload(g_ctx, n, desc.data()), "vp_circuit_upload");
    evaluate();
#ifdef USE_VIRGO
    // the commitment's tables and buffers exist before the first prover call, as the reference's namespace-scope arrays do (poly_commit.cpp:4-13, fri.cpp:13-34)
    GUARDED(vp_warm(g_ctx, VP_WARM_COMMITMENT | VP_WARM_FFT_GKR), "vp_warm");
    // the FFT scratch pad of lib/virgo's HOST transforms (init_scratch_pad, RS_polynomial.cpp:9-16: six arrays of slice_size elements, 384 MB at x1024, every
    // element constructed) is what the VERIFIER side reads later (verifier.cpp:348-361, vpd_verifier.cpp:84-86); this prover never touches it, so it is set up
    // here with the other one-off allocations instead of inside commit_private's timed span (where the reference's CPU prover, which does use it, allocates it)
    {
        const int n0 = C.circuit[0].bitLength;
        if (n0 >= 7) virgo::init_scratch_pad(1 << (n0 + virgo::rs_code_rate - virgo::log_slice_number));
    }
#endif
}

void prover::evaluate() {                                           // src/prover.cpp:27-91: the layers are evaluated in HBM
    const layer &L0 = C.circuit[0];
    circuitValue.resize(1);                                         // only the input values are kept on the host (handed to vp_evaluate)
    circuitValue[0].assign(1ULL << L0.bitLength, F_ZERO);
    for (u64 g = 0; g < L0.size; ++g) circuitValue[0][g] = F((long long) L0.gates[g].u);
    GUARDED(vp_evaluate(g_ctx, cF(circuitValue[0].data()), L0.size), "vp_evaluate");
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
    GUARDED(vp_vres(g_ctx, r_0_size ? cF(&*r_0) : nullptr, r_0_size, mF(&out)), "vp_vres");
    prove_timer.stop();
    dumpF(out);
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
    GUARDED(vp_phase1_init(g_ctx, sumcheckLayerId, cF(r_liu.data()), cF(&assert_random)), "vp_phase1_init");
    round = 0;
    prove_timer.stop();
}
void prover::sumcheckInitPhase2() {
    prove_timer.start();
    GUARDED(vp_phase2_init(g_ctx, sumcheckLayerId, cF(r_u.data())), "vp_phase2_init");
    round = 0;
    prove_timer.stop();
}
void prover::sumcheckInitLiu(vector<F>::const_iterator s) {
    prove_timer.start();
    std::vector<const vp_F *> rv(C.size, nullptr);
    for (int k = sumcheckLayerId; k < C.size; ++k) if (!r_v[k].empty()) rv[k] = cF(r_v[k].data());
    GUARDED(vp_liu_init(g_ctx, sumcheckLayerId, cF(r_u.data()), rv.data(), cF(&*s)), "vp_liu_init");
    round = 0;
    prove_timer.stop();
}

quadratic_poly prover::sumcheckUpdate(const F &previous_random, vector<F> &r_arr, int) {
    prove_timer.start();
    if (round) r_arr.at(round - 1) = previous_random;
    ++round;
    F p[3];
    GUARDED(vp_round(g_ctx, cF(&previous_random), mF(p)), "vp_round");
    prove_timer.stop();
    ++g_vpi_count.round;
    dumpF(p[0]); dumpF(p[1]); dumpF(p[2]);
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
    GUARDED(vp_finalize(g_ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
    prove_timer.stop();
    ++g_vpi_count.finalize;
    dumpF(claim);
    proof_size += sizeof(F);
}
void prover::sumcheckFinalize2(const F &previousRandom, vector<F>::iterator claims) {
    prove_timer.start();
    if (round) r_v[sumcheckLayerId][round - 1] = previousRandom;
    std::vector<F> tmp(sumcheckLayerId);
    GUARDED(vp_finalize(g_ctx, cF(&previousRandom), mF(tmp.data()), sumcheckLayerId), "vp_finalize");
    for (int i = 0; i < sumcheckLayerId; ++i) { claims[i] = tmp[i]; dumpF(tmp[i]); }
    ++g_vpi_count.finalize;
    proof_size += sizeof(F) * sumcheckLayerId;
    prove_timer.stop();
}
void prover::sumcheckLiuFinalize(const F &previousRandom, F &claim) {
    if (round) r_liu[round - 1] = previousRandom;
    GUARDED(vp_finalize(g_ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
    ++g_vpi_count.finalize;
    dumpF(claim);
}

double prover::proveTime() const { return prove_timer.elapse_sec(); }
double prover::proofSize() const { return (double) proof_size / 1024.0; }

#ifdef USE_VIRGO
// src/prover.cpp:524-530 -> poly_commit_prover::commit_private_array (lib/virgo/src/poly_commit.h:41-124) -> vpd_prover_init ->
// fri::request_init_commit(.., 0) (fri.cpp:36-139): all of it is vp_commit_private.  What stays on the host is what the VERIFIER side of
// lib/virgo reads afterwards: the FFT scratch pad its own host transforms use (init_scratch_pad, RS_polynomial.cpp:9-16; verifier.cpp:348-361
// and vpd_verifier.cpp:84-86 call inverse_fast_fourier_transform), the slice geometry globals, and poly_prover.total_time
// (printed as "Polynomial commitment: prove time", verifier.cpp:183).
virgo::__hhash_digest prover::commit_private() {
    using namespace virgo;
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int n = C.circuit[0].bitLength;
    poly_prover.total_time = 0;
    poly_commit::pre_prepare_executed = true;
    poly_commit::slice_count = (1 << log_slice_number) + 1;
    poly_commit::slice_size = 1 << (n + rs_code_rate - log_slice_number);
    poly_commit::slice_real_ele_cnt = poly_commit::slice_size >> rs_code_rate;
    poly_commit::l_eval_len = poly_commit::slice_count * poly_commit::slice_size;
    poly_commit::mask_position_gap = poly_commit::slice_size;       // one mask element (prover.cpp:526): gap = slice_size (poly_commit.h:56-62)
    poly_prover.all_pri_mask.assign(1, fieldElement(0));
    __hhash_digest d;
    { vpi_stopwatch sw(&g_vpi_sec.commit_private); GUARDED(vp_commit_private(g_ctx, reinterpret_cast<uint8_t *>(&d)), "vp_commit_private"); }
    ++g_vpi_count.commit_private;
    vpi_oracle_committed(0, n, reinterpret_cast<const unsigned char *>(&d));
    dumpH(&d);
    poly_prover.total_time += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    return d;
}
F prover::inner_prod(const vector<F> &a, const vector<F> &b, u64 l) {      // prover.cpp:532-540; commit_public takes the device's value instead
    F s = F_ZERO;
    for (u64 i = 0; i < l; ++i) s = s + a[i] * b[i];
    return s;
}
// src/prover.cpp:542-546 -> inner_prod + poly_commit_prover::commit_public_array (poly_commit.h:126-349) -> fri::request_init_commit(.., 1):
// vp_commit_public returns the inner product <circuitValue[0], pub>, all_sum[0..65) and the root of the quotient oracle.
virgo::__hhash_digest prover::commit_public(vector<F> &pub, F &inner_product_sum, std::vector<F> &mask, vector<F> &all_sum) {
    using namespace virgo;
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int n = C.circuit[0].bitLength;
    // `mask` (the PUBLIC mask, one zero from verifier.cpp:375-377) needs no forwarding whatever it holds: commit_private above committed to the private mask
    // {0} (src/prover.cpp:526), so the mask slice of l is identically zero and every place the public mask's slice enters — l q of the quotient, all_sum[64], the
    // slice's virtual oracle (poly_commit.h:196-247) — multiplies it by that zero.  A prover that commits to a NON-ZERO private mask calls
    // vp_commit_private_masked / vp_commit_public_masked (include/vpgpu.h) instead.
    (void) mask;
    if (all_sum.size() < (size_t) slice_number + 1) all_sum.resize(slice_number + 1);
    __hhash_digest d;
    { vpi_stopwatch sw(&g_vpi_sec.commit_public);
      GUARDED(vp_commit_public(g_ctx, cF(pub.data()), pub.size(), mF(&inner_product_sum), mF(all_sum.data()), reinterpret_cast<uint8_t *>(&d)),
              "vp_commit_public"); }
    ++g_vpi_count.commit_public;
    vpi_oracle_committed(1, n, reinterpret_cast<const unsigned char *>(&d));
    dumpH(&d); dumpF(inner_product_sum);
    for (int i = 0; i <= slice_number; ++i) dumpF(all_sum[i]);
    poly_prover.total_time += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    return d;
}
#endif
