#include "prover.hpp"
#include <atomic>
#include <functional>
#include <memory>
#include <chrono>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <cstdio>
#include <cstdlib>
#include <string>

static_assert(sizeof(F) == sizeof(vp_F), "F must be two u64 limbs");
static inline const vp_F *cF(const F *p) { return reinterpret_cast<const vp_F *>(p); }
static inline vp_F *mF(F *p) { return reinterpret_cast<vp_F *>(p); }

void prover::check(int rc, const char *what) {
    if (rc == VP_OK) return;
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + vp_last_error(ctx));
}

// src/prover.cpp:14-25.  The circuit tables are flattened to structure-of-arrays and copied to HBM once;
// the constructor then evaluates the circuit on the device like the reference's constructor does on the
// CPU.  A violated assert gate surfaces as an exception instead of the reference's exit(EXIT_FAILURE).
prover::prover(const layeredCircuit &cir, int device, const vp_options *options) : C(cir) {
    int rc = vp_create_with_options(device, options, &ctx);
    if (rc != VP_OK) throw std::runtime_error("vp_create failed (" + std::to_string(rc) + "): no usable MI355X / HIP device");
    const int n = C.size;
    const bool dbg = getenv("VP_DEBUG_UPLOAD") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<vp_layer_desc> desc(n);
    // uninitialised arrays (no zero fill, the pages are first touched by the threads that write them)
    struct Flat {
        std::unique_ptr<uint8_t[]> ty, as; std::unique_ptr<int32_t[]> l; std::unique_ptr<uint32_t[]> u, v, lv; std::unique_ptr<vp_F[]> c;
        std::vector<uint64_t> dsz; std::vector<int32_t> dbl; std::vector<std::vector<uint32_t>> did; std::vector<const uint32_t *> dptr;
    };
    std::vector<Flat> flat(n);
    // gate (array of structs, src/circuit.h:11-22) -> the structure-of-arrays view of vp_layer_desc.  Plain copies, split over the host's
    // cores in ranges of 2^20 gates (x1024: 1e8 gates, 0.5 s on one core): pass 1 finds the layers with constants / assert gates, pass 2 copies.
    struct Range { int layer; u64 b, e; };
    std::vector<Range> ranges;
    for (int i = 0; i < n; ++i) {
        const u64 m = C.circuit[i].size;
        for (u64 b = 0; b < m; b += (1u << 20)) ranges.push_back({i, b, std::min<u64>(m, b + (1u << 20))});
    }
    std::vector<std::atomic<char>> c_seen(n), as_seen(n);
    for (int i = 0; i < n; ++i) { c_seen[i] = 0; as_seen[i] = 0; }
    auto run_pool = [&](const std::function<void(const Range &)> &body) {
        std::atomic<size_t> next{0};
        auto work = [&]() { for (;;) { const size_t q = next.fetch_add(1); if (q >= ranges.size()) return; body(ranges[q]); } };
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const unsigned nt = (unsigned) std::min<size_t>(hw, std::max<size_t>(1, ranges.size()));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    run_pool([&](const Range &r) {
        const layer &L = C.circuit[r.layer];
        bool c = false, as = false;
        for (u64 g = r.b; g < r.e; ++g) { const gate &G = L.gates[g]; c |= (G.ty == Addc || G.ty == Mulc); as |= G.is_assert; }
        if (c) c_seen[r.layer] = 1;
        if (as) as_seen[r.layer] = 1;
    });
    std::vector<char> any_c(n, 0), any_as(n, 0);
    for (int i = 0; i < n; ++i) {
        Flat &f = flat[i];
        const u64 m = C.circuit[i].size;
        any_c[i] = c_seen[i]; any_as[i] = as_seen[i];
        f.ty.reset(new uint8_t[m]); f.as.reset(new uint8_t[m]); f.l.reset(new int32_t[m]);
        f.u.reset(new uint32_t[m]); f.v.reset(new uint32_t[m]); f.lv.reset(new uint32_t[m]);
        if (any_c[i]) f.c.reset(new vp_F[m]);
    }
    run_pool([&](const Range &r) {
        const layer &L = C.circuit[r.layer];
        Flat &f = flat[r.layer];
        const bool has_c = any_c[r.layer];
        for (u64 g = r.b; g < r.e; ++g) {
            const gate &G = L.gates[g];
            f.ty[g] = (uint8_t) G.ty; f.l[g] = G.l;
            f.u[g] = r.layer == 0 ? 0u : (uint32_t) G.u;      // layer 0: u carries the input value, not an index
            f.v[g] = (uint32_t) G.v; f.lv[g] = (uint32_t) G.lv;
            f.as[g] = G.is_assert ? 1 : 0;
            if (has_c) { f.c[g].real = G.c.real; f.c[g].img = G.c.img; }
        }
    });
    {   // the subset lists (u64 in the reference's layout) -> u32, one layer per task
        std::atomic<int> next{0};
        auto work = [&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) return;
                const layer &L = C.circuit[i];
                Flat &f = flat[i];
                f.did.resize(L.dadId.size());
                for (size_t j = 0; j < L.dadId.size(); ++j) f.did[j].assign(L.dadId[j].begin(), L.dadId[j].end());
            }
        };
        const unsigned nt = std::max(1u, std::min<unsigned>(std::min(16u, std::thread::hardware_concurrency()), (unsigned) n));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    for (int i = 0; i < n; ++i) {
        const layer &L = C.circuit[i];
        Flat &f = flat[i];
        f.dsz.assign(L.dadSize.begin(), L.dadSize.end());
        f.dbl.assign(L.dadBitLength.begin(), L.dadBitLength.end());
        f.dptr.resize(L.dadId.size());
        for (size_t j = 0; j < L.dadId.size(); ++j) f.dptr[j] = f.did[j].data();
        vp_layer_desc &d = desc[i];
        d.size = L.size; d.bit_length = L.bitLength;
        d.ty = f.ty.get(); d.l = f.l.get(); d.u = f.u.get(); d.v = f.v.get(); d.lv = f.lv.get();
        d.c = any_c[i] ? f.c.get() : nullptr;
        d.is_assert = any_as[i] ? f.as.get() : nullptr;
        d.dad_size = f.dsz.data(); d.dad_bitlen = f.dbl.data(); d.dad_id = f.dptr.data();
    }
    // the destructor of a partially constructed object never runs: release the context (streams, pinned buffers, the circuit and
    // witness in HBM) before the exception leaves, e.g. when evaluate() reports a violated assert gate (VP_EASSERT)
    try {
        const double t0 = since();
        check(vp_circuit_upload(ctx, n, desc.data()), "vp_circuit_upload");
        const double t1 = since();
        evaluate();
        if (dbg) fprintf(stderr, "[vp upload] flatten %.3f s  vp_circuit_upload %.3f s  evaluate %.3f s\n", t0, t1 - t0, since() - t1);
    } catch (...) {
        vp_destroy(ctx);
        ctx = nullptr;
        throw;
    }
}

prover::~prover() { vp_destroy(ctx); }

void prover::evaluate() {      // src/prover.cpp:27-91
    const layer &L0 = C.circuit[0];
    std::vector<vp_F> in(L0.size);
    for (u64 g = 0; g < L0.size; ++g) { F x((long long) L0.gates[g].u); in[g].real = x.real; in[g].img = x.img; }
    check(vp_evaluate(ctx, in.data(), in.size()), "vp_evaluate");
}

void prover::init() {          // src/prover.cpp:131-155
    int max_bl = 0;
    for (auto &c : C.circuit) max_bl = std::max(max_bl, c.bitLength);
    r_u.assign(max_bl, F_ZERO);
    r_liu.assign(max_bl, F_ZERO);
    r_v.assign(C.size, std::vector<F>());
    for (int i = 1; i < C.size; ++i)
        if (C.circuit[i].maxDadBitLength != -1) r_v[i].assign(C.circuit[i].maxDadBitLength, F_ZERO);
}

F prover::Vres(const std::vector<F>::const_iterator &r_0, int r_0_size) {       // src/prover.cpp:99-129
    prove_timer.start();
    F out;
    check(vp_vres(ctx, r_0_size ? cF(&*r_0) : nullptr, r_0_size, mF(&out)), "vp_vres");
    prove_timer.stop();
    return out;
}

void prover::sumcheckInitAll(const std::vector<F>::const_iterator &r_last) {     // src/prover.cpp:162-170
    prove_timer.start();
    const int last_bl = C.circuit[C.size - 1].bitLength;
    sumcheckLayerId = C.size;
    for (int i = 0; i < last_bl; ++i) r_liu[i] = r_last[i];
    prove_timer.stop();
}

void prover::sumcheckInit() { --sumcheckLayerId; }                               // src/prover.cpp:177-184

void prover::sumcheckInitPhase1(const F &assert_random) {                        // src/prover.cpp:189-280
    prove_timer.start();
    init_timer.start();
    check(vp_phase1_init(ctx, sumcheckLayerId, cF(r_liu.data()), cF(&assert_random)), "vp_phase1_init");
    init_timer.stop();
    round = 0;
    prove_timer.stop();
}

void prover::sumcheckInitPhase2() {                                              // src/prover.cpp:282-367
    prove_timer.start();
    init_timer.start();
    check(vp_phase2_init(ctx, sumcheckLayerId, cF(r_u.data())), "vp_phase2_init");
    init_timer.stop();
    round = 0;
    prove_timer.stop();
}

void prover::sumcheckInitLiu(std::vector<F>::const_iterator s) {                 // src/prover.cpp:369-420
    prove_timer.start();
    std::vector<const vp_F *> rv(C.size, nullptr);
    for (int k = sumcheckLayerId; k < C.size; ++k) if (!r_v[k].empty()) rv[k] = cF(r_v[k].data());
    init_timer.start();
    check(vp_liu_init(ctx, sumcheckLayerId, cF(r_u.data()), rv.data(), cF(&*s)), "vp_liu_init");
    init_timer.stop();
    round = 0;
    prove_timer.stop();
}

quadratic_poly prover::sumcheckUpdate(const F &previous_random, std::vector<F> &r_arr) {   // src/prover.cpp:436-455
    prove_timer.start();
    if (round) r_arr.at(round - 1) = previous_random;
    ++round;
    F p[3];
    static const bool dbg_rounds = getenv("VP_DEBUG_ROUNDS") != nullptr;      // development aid: per-message latency on stderr
    const auto t_dbg = std::chrono::steady_clock::now();
    round_timer.start();
    check(vp_round(ctx, cF(&previous_random), mF(p)), "vp_round");
    round_timer.stop();
    if (dbg_rounds) fprintf(stderr, "[round] layer %d round %d: %.2f us\n", sumcheckLayerId, round, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_dbg).count());
    prove_timer.stop();
    proof_size += sizeof(F) * 3;
    return quadratic_poly(p[0], p[1], p[2]);
}
quadratic_poly prover::sumcheckUpdatePhase1(const F &r) { return sumcheckUpdate(r, r_u); }
quadratic_poly prover::sumcheckUpdatePhase2(const F &r) { return sumcheckUpdate(r, r_v[sumcheckLayerId]); }
quadratic_poly prover::sumcheckLiuUpdate(const F &r) { return sumcheckUpdate(r, r_liu); }

void prover::sumcheckFinalize1(const F &previousRandom, F &claim) {              // src/prover.cpp:494-501
    prove_timer.start();
    if (round) r_u[round - 1] = previousRandom;
    fin_timer.start();
    check(vp_finalize(ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
    fin_timer.stop();
    prove_timer.stop();
    proof_size += sizeof(F);
}

void prover::sumcheckFinalize2(const F &previousRandom, std::vector<F>::iterator claims) {   // src/prover.cpp:504-516
    prove_timer.start();
    if (round) r_v[sumcheckLayerId][round - 1] = previousRandom;
    std::vector<F> tmp(sumcheckLayerId);
    fin_timer.start();
    check(vp_finalize(ctx, cF(&previousRandom), mF(tmp.data()), sumcheckLayerId), "vp_finalize");
    fin_timer.stop();
    for (int i = 0; i < sumcheckLayerId; ++i) claims[i] = tmp[i];
    proof_size += sizeof(F) * sumcheckLayerId;
    prove_timer.stop();
}

void prover::sumcheckLiuFinalize(const F &previousRandom, F &claim) {            // src/prover.cpp:518-521
    if (round) r_liu[round - 1] = previousRandom;
    fin_timer.start();
    check(vp_finalize(ctx, cF(&previousRandom), mF(&claim), 1), "vp_finalize");
    fin_timer.stop();
}

prover::hhash_digest prover::commit_private() {      // src/prover.cpp:524-530 (mask = one zero element)
    masked = false;
    hhash_digest d;
    check(vp_commit_private(ctx, d.b), "vp_commit_private");
    return d;
}
prover::hhash_digest prover::commit_public(std::vector<F> &pub, F &inner_product_sum, std::vector<F> &all_sum) {
    hhash_digest d;
    all_sum.resize(65);
    check(vp_commit_public(ctx, cF(pub.data()), pub.size(), mF(&inner_product_sum), mF(all_sum.data()), d.b), "vp_commit_public");
    return d;
}
static bool all_zero(const std::vector<F> &v) { for (auto &x : v) if (x.real | x.img) return false; return true; }
prover::hhash_digest prover::commit_private(const std::vector<F> &mask) {      // poly_commit.h:41-124 with its mask argument
    if (mask.empty() || all_zero(mask)) { masked = false; return commit_private(); }
    hhash_digest d;
    check(vp_commit_private_masked(ctx, cF(mask.data()), mask.size(), d.b), "vp_commit_private_masked");
    masked = true;
    return d;
}
prover::hhash_digest prover::commit_public(std::vector<F> &pub, F &inner_product_sum, std::vector<F> &mask, std::vector<F> &all_sum) {      // src/prover.cpp:542-546
    if (!masked) return commit_public(pub, inner_product_sum, all_sum);      // behind a zero private mask the public mask's slice is multiplied by zero everywhere
    hhash_digest d;
    all_sum.resize(65);
    std::vector<F> one_zero(1, F_ZERO);
    const std::vector<F> &m = mask.empty() ? one_zero : mask;
    check(vp_commit_public_masked(ctx, cF(pub.data()), pub.size(), cF(m.data()), m.size(), mF(&inner_product_sum), mF(all_sum.data()), d.b), "vp_commit_public_masked");
    return d;
}
std::vector<F> prover::friFinalMask() {
    std::vector<F> out(32);
    check(vp_fri_final_mask(ctx, mF(out.data())), "vp_fri_final_mask");
    return out;
}
prover::hhash_digest prover::commit_public_eq(const std::vector<F> &point, F &inner_product_sum, std::vector<F> &all_sum) {
    hhash_digest d;
    all_sum.resize(65);
    check(vp_commit_public_eq(ctx, cF(point.data()), (int) point.size(), mF(&inner_product_sum), mF(all_sum.data()), d.b), "vp_commit_public_eq");
    return d;
}
std::vector<F> prover::predicates(int layer, const std::vector<F> &r_g, const F &assert_random, const std::vector<F> &r_u,
                                  const std::vector<F> &r_v, int n_v) {
    std::vector<F> out(5 + 7 * (size_t) layer);
    check(vp_predicates(ctx, layer, cF(r_g.data()), cF(&assert_random), cF(r_u.data()), n_v ? cF(r_v.data()) : nullptr, n_v,
                        mF(out.data()), out.size()), "vp_predicates");
    return out;
}
F prover::liuGr(int layer, const std::vector<F> &ru, const std::vector<std::vector<F>> &rv_all, const std::vector<F> &sig, const std::vector<F> &rliu) {
    std::vector<const vp_F *> rv(C.size, nullptr);
    for (int k = layer; k < C.size; ++k) if (!rv_all[k].empty()) rv[k] = cF(rv_all[k].data());
    F out;
    check(vp_liu_gr(ctx, layer, cF(ru.data()), rv.data(), cF(sig.data()), cF(rliu.data()), mF(&out)), "vp_liu_gr");
    return out;
}
F prover::layerMle(int layer, const std::vector<F> &r, int n) {
    F out;
    check(vp_layer_mle(ctx, layer, n ? cF(r.data()) : nullptr, n, mF(&out)), "vp_layer_mle");
    return out;
}
prover::hhash_digest prover::friStep(const F &r) {
    hhash_digest d;
    check(vp_fri_step(ctx, cF(&r), d.b), "vp_fri_step");
    return d;
}
std::vector<prover::hhash_digest> prover::friCommit(const std::vector<F> &r) {
    std::vector<hhash_digest> d(r.size());
    check(vp_fri_commit(ctx, cF(r.data()), (int) r.size(), d[0].b), "vp_fri_commit");
    return d;
}
std::vector<F> prover::friFinal() {
    std::vector<F> out(2048);
    check(vp_fri_final(ctx, mF(out.data())), "vp_fri_final");
    return out;
}
void prover::friOpen(int oracle, u64 leaf, std::vector<F> &values, std::vector<hhash_digest> &path) {
    values.resize(130);
    path.resize(40);
    int len = 0;
    check(vp_fri_open(ctx, oracle, leaf, mF(values.data()), path[0].b, 40 * 32, &len), "vp_fri_open");
    path.resize(len);
}
std::vector<F> prover::fftGkr(int lg, const std::vector<F> &tape) {
    uint64_t nt = 0, nm = 0;
    check(vp_fft_gkr_sizes(lg, &nt, &nm), "vp_fft_gkr_sizes");
    if (tape.size() != nt) throw std::runtime_error("fftGkr: tape has the wrong length");
    std::vector<F> msgs(nm);
    uint64_t written = 0;
    check(vp_fft_gkr(ctx, lg, cF(tape.data()), nt, mF(msgs.data()), nm, &written), "vp_fft_gkr");
    msgs.resize(written);
    return msgs;
}
void prover::fftGkrBegin(int lg, const std::vector<F> &tape) {
    check(vp_fft_gkr_begin(ctx, lg, cF(tape.data()), tape.size()), "vp_fft_gkr_begin");
}
std::vector<F> prover::fftGkrEnd(int lg) {
    uint64_t nt = 0, nm = 0;
    check(vp_fft_gkr_sizes(lg, &nt, &nm), "vp_fft_gkr_sizes");
    std::vector<F> msgs(nm);
    uint64_t written = 0;
    check(vp_fft_gkr_end(ctx, mF(msgs.data()), nm, &written), "vp_fft_gkr_end");
    msgs.resize(written);
    return msgs;
}
void prover::fftGkrCancel() noexcept { (void) vp_fft_gkr_cancel(ctx); }
double prover::commitDeviceMs() { double ms = 0; check(vp_commit_stats(ctx, &ms), "vp_commit_stats"); return ms; }

void prover::gkrSizes(u64 &n_tape, u64 &n_bytes) {
    uint64_t a = 0, b = 0;
    check(vp_gkr_sizes(ctx, &a, &b), "vp_gkr_sizes");
    n_tape = a; n_bytes = b;
}

void prover::proveGKR(const std::vector<F> &tape, std::vector<uint8_t> &transcript) {
    u64 nt, nb;
    gkrSizes(nt, nb);
    if (tape.size() != nt) throw std::runtime_error("proveGKR: tape has the wrong length");
    const size_t at = transcript.size();
    transcript.resize(at + nb);
    uint64_t written = 0;
    prove_timer.start();
    check(vp_prove_gkr(ctx, cF(tape.data()), nt, transcript.data() + at, nb, &written), "vp_prove_gkr");
    prove_timer.stop();
    transcript.resize(at + written);
}

std::vector<F> prover::layerValues(int layer) {
    std::vector<F> out(C.circuit[layer].size);
    check(vp_layer_values(ctx, layer, mF(out.data()), out.size()), "vp_layer_values");
    return out;
}

vp_stats prover::stats() {
    vp_stats s{};
    check(vp_get_stats(ctx, &s), "vp_get_stats");
    return s;
}
