#include "vphost.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "circuit.hpp"
#include "prover.hpp"
#include "verifier.hpp"

struct vph_circuit { layeredCircuit c; };
struct vph_session {
    vph_circuit *circ;
    std::unique_ptr<prover> p;
    std::vector<F> tape;
    std::vector<uint8_t> fri_roots; std::vector<F> fri_final, fri_r;      // FRI commit phase of the last complete-protocol run
    std::vector<F> fft_gkr_msgs; double pc_times[3] = {0, 0, 0};          // fft_gkr messages; {PC prove (reference definition), of which fft_gkr, query answering}
    double t_init = 0, t_round = 0, t_fin = 0;
    std::vector<F> last_point;                                           // r_liu after the last Liu sumcheck of the last complete-protocol run
    std::vector<F> ptape_fft, ptape_fri;                                 // vph_draw_protocol_tape: the draws after the GKR tape (fft_gkr, FRI folds)
    // vph_prove_protocol_ex: where the device's results land (the calls of a deferred pass write them when the pass is collected)
    std::vector<uint8_t> pp_gkr, pp_roots; std::vector<F> pp_all, pp_final; F pp_inner; uint64_t pp_written = 0;
    uint8_t pp_root_l[32], pp_root_h[32], pp_root_next[32];
    bool head_queued = false; uint64_t head_epoch = 0;                   // the previous pass queued this one's commit_private (VPH_PASS_QUEUE_NEXT)
};

static void set_err(char *err, int errlen, const std::string &m) {
    if (err && errlen > 0) { strncpy(err, m.c_str(), errlen - 1); err[errlen - 1] = 0; }
}

extern "C" {

vph_circuit *vph_circuit_from_pws(const char *path, int blocks, long seed, char *err, int errlen) {
    if (!path || blocks < 1) { set_err(err, errlen, "bad arguments"); return nullptr; }
    if (seed >= 0) srandom((unsigned) seed);
    std::string e;
    vph_circuit *vc = new vph_circuit();
    const char *route = getenv("VPH_BUILD");                 // "dag": the loader's own route (DAG of all blocks) for any block count
    if (blocks > 1 && !(route && !strcmp(route, "dag"))) {
        if (!vph::build_replicated(path, blocks, vc->c, &e)) { delete vc; set_err(err, errlen, e); return nullptr; }
        return vc;
    }
    std::vector<DAG_gate> dag;
    if (!vph::parse_pws(path, blocks, dag, &e)) { delete vc; set_err(err, errlen, e); return nullptr; }
    vc->c = vph::DAG_to_layered(dag);
    vc->c.subsetInit();
    return vc;
}
vph_circuit *vph_circuit_randomize(int layers, int log_size, long seed) {
    if (layers < 2 || log_size < 0 || log_size > 28) return nullptr;
    if (seed >= 0) srandom((unsigned) seed);
    vph_circuit *vc = new vph_circuit();
    vc->c = layeredCircuit::randomize(layers, log_size);
    vc->c.subsetInit();
    return vc;
}
vph_circuit *vph_circuit_custom(int n_layers, const uint64_t *layer_sizes, const int32_t *ty, const int32_t *l, const uint64_t *u,
                                const uint64_t *v, const uint64_t *c_pairs, const uint8_t *is_assert) {
    vph_circuit *vc = new vph_circuit();
    layeredCircuit &c = vc->c;
    c.size = n_layers;
    c.circuit.resize(n_layers);
    u64 at = 0;
    for (int i = 0; i < n_layers; ++i) {
        layer &L = c.circuit[i];
        L.size = layer_sizes[i];
        L.bitLength = 0;
        while ((1ull << L.bitLength) < L.size) ++L.bitLength;
        L.gates.resize(L.size);
        for (u64 g = 0; g < L.size; ++g, ++at) {
            F cc; cc.real = c_pairs[2 * at]; cc.img = c_pairs[2 * at + 1];
            L.gates[g] = gate((gateType) ty[at], l[at], u[at], v[at], cc, is_assert[at] != 0);
        }
    }
    c.subsetInit();
    return vc;
}
void vph_circuit_free(vph_circuit *c) { delete c; }
int vph_circuit_layers(const vph_circuit *c) { return c->c.size; }
uint64_t vph_circuit_gates(const vph_circuit *c) { u64 n = 0; for (auto &l : c->c.circuit) n += l.size; return n; }
uint64_t vph_circuit_layer_size(const vph_circuit *c, int layer) { return c->c.circuit[layer].size; }
int vph_circuit_layer_bitlen(const vph_circuit *c, int layer) { return c->c.circuit[layer].bitLength; }
void vph_circuit_hash(const vph_circuit *c, uint64_t out[2]) { u64 h[2]; c->c.structuralHash(h); out[0] = h[0]; out[1] = h[1]; }

vph_session *vph_session_create(vph_circuit *c, int device, char *err, int errlen) { return vph_session_create_opts(c, device, nullptr, err, errlen); }
vph_session *vph_session_create_opts(vph_circuit *c, int device, const vp_options *opt, char *err, int errlen) {
    if (!c) { set_err(err, errlen, "null circuit"); return nullptr; }
    try {
        std::unique_ptr<vph_session> s(new vph_session());
        s->circ = c;
        s->p.reset(new prover(c->c, device, opt));
        return s.release();
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return nullptr;
    }
}
void vph_session_free(vph_session *s) { delete s; }
int vph_set_profiling(vph_session *s, int level) { return vp_set_profiling(s->p->context(), level); }
void *vph_session_ctx(vph_session *s) { return s ? (void *) s->p->context() : nullptr; }

int vph_layer_values(vph_session *s, int layer, uint64_t *out, uint64_t n) {
    try {
        std::vector<F> v = s->p->layerValues(layer);
        if (n > v.size()) return -1;
        for (u64 i = 0; i < n; ++i) { out[2 * i] = v[i].real; out[2 * i + 1] = v[i].img; }
        return 0;
    } catch (const std::exception &) { return -2; }
}

static void fill(vph_result *res, prover &p, double prove_sec, double verify_sec, bool ok) {
    if (!res) return;
    vp_stats st = p.stats();
    res->prove_sec = prove_sec; res->gkr_device_ms = st.gkr_ms; res->evaluate_ms = st.evaluate_ms;
    res->verify_sec = verify_sec; res->fold_ms = st.fold_ms; res->fold_launches = st.fold_launches;
    res->fold_bytes = st.fold_bytes; res->rounds = st.rounds; res->launches = st.launches;
    res->proof_kb = p.proofSize(); res->verified = ok ? 1 : 0;
}

// seconds the interactive entry points spent in phase inits / round messages / finalize calls during the last vph_prove_interactive
void vph_interactive_breakdown(vph_session *s, double out[3]) { out[0] = s->t_init; out[1] = s->t_round; out[2] = s->t_fin; }

int vph_prove_interactive(vph_session *s, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, vph_result *res,
                          char *err, int errlen) {
    try {
        F::init();
        verifier v(s->p.get(), s->circ->c);
        const double t0 = s->p->proveTime();
        const double i0 = s->p->initTime(), r0 = s->p->roundTime(), f0 = s->p->finalizeTime();
        struct Keep { vph_session *s; double i0, r0, f0; ~Keep() { s->t_init = s->p->initTime() - i0; s->t_round = s->p->roundTime() - r0; s->t_fin = s->p->finalizeTime() - f0; } } keep{s, i0, r0, f0};
        const bool ok = v.verify();
        const auto &tr = v.transcript();
        if (tr.size() > capacity) { set_err(err, errlen, "transcript buffer too small"); return -1; }
        memcpy(transcript, tr.data(), tr.size());
        if (n_written) *n_written = tr.size();
        fill(res, *s->p, s->p->proveTime() - t0, v.verifyTime(), ok);
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

int vph_prove_fs(vph_session *s, uint8_t *proof, uint64_t capacity, uint64_t *n_written, vph_result *res, char *err, int errlen) {
    try {
        verifier v(s->p.get(), s->circ->c);
        const double t0 = s->p->proveTime();
        const bool ok = v.proveFS();
        const auto &tr = v.transcript();
        if (tr.size() > capacity) { set_err(err, errlen, "proof buffer too small"); return -1; }
        memcpy(proof, tr.data(), tr.size());
        if (n_written) *n_written = tr.size();
        fill(res, *s->p, s->p->proveTime() - t0, v.verifyTime(), ok);
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

int vph_verify_fs(vph_circuit *c, const uint8_t *proof, uint64_t n) {
    try {
        verifier v(nullptr, c->c);
        std::vector<uint8_t> tr(proof, proof + n);
        return v.checkFS(tr) ? 0 : 1;
    } catch (const std::exception &) { return 1; }
}

int vph_draw_tape(vph_session *s) {
    F::init();
    verifier v(nullptr, s->circ->c);
    s->tape = v.drawTape();
    return 0;
}

int vph_prove_gkr(vph_session *s, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, vph_result *res, char *err,
                  int errlen) {
    try {
        if (s->tape.empty()) vph_draw_tape(s);
        std::vector<uint8_t> tr;
        const double t0 = s->p->proveTime();
        s->p->proveGKR(s->tape, tr);
        if (tr.size() > capacity) { set_err(err, errlen, "transcript buffer too small"); return -1; }
        memcpy(transcript, tr.data(), tr.size());
        if (n_written) *n_written = tr.size();
        fill(res, *s->p, s->p->proveTime() - t0, 0, false);
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

double vph_commit_device_ms(vph_session *s) { double ms = -1; vp_commit_stats(s->p->context(), &ms); return ms; }

int vph_set_shard(vph_session *s, int rank, int world) { return vp_set_shard(s->p->context(), rank, world) == VP_OK ? 0 : -1; }

// index-split proof, caller-side exchange of V_u (include/vpgpu.h: vp_shard_vu_partials / vp_shard_vu_set) on the session's tape; -> n, < 0: refused
int vph_shard_vu_partials(vph_session *s, uint64_t *partials /* 2 per entry */, int capacity) {
    if (s->tape.empty()) vph_draw_tape(s);
    uint64_t n = 0;
    static_assert(sizeof(F) == sizeof(vp_F), "field element layout");
    const int rc = vp_shard_vu_partials(s->p->context(), reinterpret_cast<const vp_F *>(s->tape.data()), s->tape.size(), reinterpret_cast<vp_F *>(partials), (uint64_t) capacity, &n);
    return rc == VP_OK ? (int) n : -1;
}
int vph_shard_vu_set(vph_session *s, const uint64_t *sums, int n) {
    return vp_shard_vu_set(s->p->context(), reinterpret_cast<const vp_F *>(sums), (uint64_t) n) == VP_OK ? 0 : -1;
}

int vph_shard_chains(vph_session *s, int32_t *owner, double *cost, int capacity) {
    int n = 0;
    return vp_shard_chains(s->p->context(), owner, cost, capacity, &n) == VP_OK ? n : -1;
}

int vph_check(vph_session *s, const uint8_t *transcript, uint64_t n, int skip_predicates, double *verify_sec) {
    try {
        verifier v(nullptr, s->circ->c);
        v.skip_predicates = (skip_predicates & 1) != 0;
        if (skip_predicates & 2) v.pred_dev = s->p.get();          // bit 1: predicate loops on the device
        std::vector<uint8_t> tr(transcript, transcript + n);
        timer t; t.start();
        const bool ok = v.check(s->tape, tr);
        t.stop();
        if (verify_sec) *verify_sec = t.elapse_sec();
        return ok ? 0 : 1;
    } catch (const std::exception &) { return -2; }
}

int vph_commit_private(vph_session *s, uint8_t root[32], double *ms, char *err, int errlen) {
    try {
        prover::hhash_digest d = s->p->commit_private();
        memcpy(root, d.b, 32);
        if (ms) *ms = s->p->commitDeviceMs();
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

int vph_commit_public(vph_session *s, const uint64_t *pub_pairs, uint64_t n_pub, uint8_t *out, double *ms, char *err, int errlen) {
    try {
        std::vector<F> pub(n_pub), all_sum;
        for (u64 i = 0; i < n_pub; ++i) { pub[i].real = pub_pairs[2 * i]; pub[i].img = pub_pairs[2 * i + 1]; }
        F inner;
        prover::hhash_digest d = s->p->commit_public(pub, inner, all_sum);
        memcpy(out, d.b, 32);
        memcpy(out + 32, &inner, 16);
        memcpy(out + 48, all_sum.data(), 65 * 16);
        if (ms) *ms = s->p->commitDeviceMs();
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

int vph_prove_and_verify_full(vph_session *s, int reps, uint8_t *transcript, uint64_t capacity, uint64_t *n_written,
                              double *gkr_prove_sec, double *pc_prove_sec, double *verify_sec, char *err, int errlen) {
    try {
        F::init();
        verifier v(s->p.get(), s->circ->c);
        const double t0 = s->p->proveTime();
        const bool ok = v.verifyFull(reps);
        s->fri_roots = v.friRoots(); s->fri_final = v.friFinalCode(); s->fri_r = v.friChallenges();
        s->fft_gkr_msgs = v.fftGkrMessages();
        s->pc_times[0] = v.polyProveTime(); s->pc_times[1] = v.fftGkrProveTime(); s->pc_times[2] = v.polyOpenTime();
        s->last_point.assign(v.finalPoint().begin(), v.finalPoint().begin() + s->circ->c.circuit[0].bitLength);
        const auto &tr = v.fullTranscript();
        if (tr.size() > capacity) { set_err(err, errlen, "transcript buffer too small"); return -1; }
        memcpy(transcript, tr.data(), tr.size());
        if (n_written) *n_written = tr.size();
        if (gkr_prove_sec) *gkr_prove_sec = s->p->proveTime() - t0;
        if (pc_prove_sec) *pc_prove_sec = v.polyProveTime();
        if (verify_sec) *verify_sec = v.verifyTime() + v.polyVerifyTime();
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

// fft_gkr of the last vph_prove_and_verify_full: its messages ({real,img} pairs, layout of vp_fft_gkr); returns their number or -1
int64_t vph_last_fft_gkr(vph_session *s, uint64_t *pairs, uint64_t cap_elems) {
    if (!s || s->fft_gkr_msgs.empty() || cap_elems < s->fft_gkr_msgs.size()) return -1;
    memcpy(pairs, s->fft_gkr_msgs.data(), s->fft_gkr_msgs.size() * sizeof(F));
    return (int64_t) s->fft_gkr_msgs.size();
}
// seconds of the last vph_prove_and_verify_full: [0] "Polynomial commitment: prove time" by the reference's definition (commit_private +
// commit_public + fft_gkr + FRI commit phase), [1] its fft_gkr share, [2] answering the verifier's queries (outside [0], as in the reference)
void vph_last_pc_times(vph_session *s, double out[3]) { for (int i = 0; i < 3; ++i) out[i] = s ? s->pc_times[i] : 0; }

// FRI data of the last vph_prove_and_verify_full on this session: roots (32 bytes per step), final codeword (2048 elements),
// challenges (16 bytes per step); returns the number of fold steps or -1.
int vph_last_fri(vph_session *s, uint8_t *roots, uint64_t roots_cap, uint64_t *final_pairs, uint64_t *r_pairs) {
    if (!s || s->fri_roots.empty()) return -1;
    const int steps = (int) (s->fri_roots.size() / 32);
    if (roots) { if (roots_cap < s->fri_roots.size()) return -1; memcpy(roots, s->fri_roots.data(), s->fri_roots.size()); }
    if (final_pairs) memcpy(final_pairs, s->fri_final.data(), s->fri_final.size() * sizeof(F));
    if (r_pairs) memcpy(r_pairs, s->fri_r.data(), s->fri_r.size() * sizeof(F));
    return steps;
}

// The point the input layer is opened at (r_liu after the last Liu sumcheck) in the last vph_prove_full / vph_prove_and_verify_full:
// the protocol's public vector is its eq table (src/verifier.cpp:368-369).  Returns the number of coordinates or -1.
int vph_last_point(vph_session *s, uint64_t *pairs, int cap) {
    if (!s || s->last_point.empty() || cap < (int) s->last_point.size()) return -1;
    memcpy(pairs, s->last_point.data(), s->last_point.size() * sizeof(F));
    return (int) s->last_point.size();
}

void vph_test_sha3(const uint8_t *in, uint8_t *out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t m[8];
        memcpy(m, in + 64 * i, 64);
        vph::hhash_digest prev; memcpy(prev.w, m + 4, 32);
        vph::hhash_digest d = vph::hhash(m, prev);
        memcpy(out + 32 * i, d.w, 32);
    }
}

int vph_fri_commit(vph_session *s, const uint64_t *r_pairs, int n_steps, uint8_t *roots, uint64_t *final_pairs, char *err, int errlen) {
    vp_ctx *ctx = s->p->context();
    for (int k = 0; k < n_steps; ++k) {
        vp_F r; r.real = r_pairs[2 * k]; r.img = r_pairs[2 * k + 1];
        int rc = vp_fri_step(ctx, &r, roots + 32 * k);
        if (rc != VP_OK) { set_err(err, errlen, std::string("vp_fri_step: ") + vp_last_error(ctx)); return rc; }
    }
    int rc = vp_fri_final(ctx, reinterpret_cast<vp_F *>(final_pairs));
    if (rc != VP_OK) { set_err(err, errlen, std::string("vp_fri_final: ") + vp_last_error(ctx)); return rc; }
    return 0;
}

int vph_fri_commit_batched(vph_session *s, const uint64_t *r_pairs, int n_steps, uint8_t *roots, uint64_t *final_pairs, char *err, int errlen) {
    vp_ctx *ctx = s->p->context();
    int rc = vp_fri_commit(ctx, reinterpret_cast<const vp_F *>(r_pairs), n_steps, roots);
    if (rc != VP_OK) { set_err(err, errlen, std::string("vp_fri_commit: ") + vp_last_error(ctx)); return rc; }
    rc = vp_fri_final(ctx, reinterpret_cast<vp_F *>(final_pairs));
    if (rc != VP_OK) { set_err(err, errlen, std::string("vp_fri_final: ") + vp_last_error(ctx)); return rc; }
    return 0;
}

int vph_prove_full(vph_session *s, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, int batched, char *err, int errlen) {
    try {
        F::init();
        std::vector<uint8_t> out;
        prover::hhash_digest rl = s->p->commit_private();                      // verifier.cpp:137
        out.insert(out.end(), rl.b, rl.b + 32);
        verifier v(batched ? nullptr : s->p.get(), s->circ->c);
        bool ok;
        if (batched) {
            std::vector<F> tape = v.drawTape();
            std::vector<uint8_t> tr;
            s->p->proveGKR(tape, tr);
            v.pred_dev = s->p.get();                 // the verifier's O(|C|) loops on the device (the prover is idle during the replay)
            ok = v.check(tape, tr);
        } else {
            ok = v.verify();
        }
        out.insert(out.end(), v.transcript().begin(), v.transcript().end());
        // verifyPoly (verifier.cpp:363-379): the public vector is eq(r_liu, .) over the input layer
        s->last_point.assign(v.finalPoint().begin(), v.finalPoint().begin() + s->circ->c.circuit[0].bitLength);
        std::vector<F> pub;
        initBetaTable(pub, s->circ->c.circuit[0].bitLength, v.finalPoint().begin(), F_ONE);
        pub.resize(1ull << s->circ->c.circuit[0].bitLength);
        F inner; std::vector<F> all_sum;
        prover::hhash_digest rh = s->p->commit_public(pub, inner, all_sum);
        out.insert(out.end(), rh.b, rh.b + 32);
        const uint8_t *ib = reinterpret_cast<const uint8_t *>(&inner);
        out.insert(out.end(), ib, ib + 16);
        const uint8_t *ab = reinterpret_cast<const uint8_t *>(all_sum.data());
        out.insert(out.end(), ab, ab + 65 * 16);
        if (out.size() > capacity) { set_err(err, errlen, "transcript buffer too small"); return -1; }
        memcpy(transcript, out.data(), out.size());
        if (n_written) *n_written = out.size();
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

// ---- the prover side of the COMPLETE protocol as one pass (the bench's configs[2] step) -----------------------------------------------
// The reference verifier's draws do not depend on the prover's messages (glibc random(), lib/virgo/src/fieldElement.cpp:119-124), so the
// whole tape of verifier::verify() can be drawn first, in its order: F::init(), the GKR draws (src/verifier.cpp:144-279), fft_gkr's
// (lib/virgo/src/fft_circuit_GKR.cpp, count verifier::fftGkrDraws), the FRI fold challenges (vpd_verifier.cpp:57).
int vph_draw_protocol_tape(vph_session *s) {
    const int n = s->circ->c.circuit[0].bitLength;
    if (n < 7) return -1;
    F::init();
    verifier v(nullptr, s->circ->c);
    s->tape = v.drawTape();
    const int ln = n - 6;
    s->ptape_fft.resize((size_t) verifier::fftGkrDraws(ln));
    for (auto &x : s->ptape_fft) x = F::random();
    s->ptape_fri.resize(ln);
    for (auto &x : s->ptape_fri) x = F::random();
    return 0;
}
// commit_private -> GKR (one batched device pass) -> commit_public on eq(r_liu, .) built on the device -> fft_gkr -> FRI commit phase +
// final codeword: every prover call of verifier::verify() except answering the queries, nothing of the verifier.  transcript = the golden
// layout (merkle_root_l | GKR | merkle_root_h | input_0 | all_sum[65]); fri_roots: 32 bytes per fold step; final_pairs: 2048 elements.
// flags (vphost.h): VPH_PASS_DEFERRED — the calls are queued back to back without a host wait between them (vp_set_deferred) and collected at the end;
// VPH_PASS_QUEUE_NEXT — before it waits, the pass queues the HEAD of the next one (commit_private of the same witness) behind its own folds, so
// that the device goes from this proof's last kernel straight into the next proof's first (the next pass finds it and starts at the GKR part).
// sec[6] = whole pass (host wall clock) | commit_private | GKR | commit_public | fft_gkr (host time of its begin + end) | FRI commit phase + final;
// synchronous: host wall clock of each call, deferred: device time of each call (vp_phase_ms).  The fft_gkr messages stay with the session
// (vph_last_fft_gkr), the FRI data too (vph_last_fri).  0 = done, < 0 = error.
int vph_prove_protocol_ex(vph_session *s, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, uint8_t *fri_roots, uint64_t roots_cap,
                          uint64_t *final_pairs, double sec[6], int flags, char *err, int errlen) {
    vp_ctx *ctx = s->p->context();
    const bool defer = (flags & VPH_PASS_DEFERRED) != 0, queue_next = defer && (flags & VPH_PASS_QUEUE_NEXT) != 0;
    // a run of fft_gkr that is begun and not collected when this function leaves, for whatever reason, is dropped (it would make every later pass of the
    // session fail with "not collected")
    struct FftGuard { prover *p; bool armed; ~FftGuard() { if (armed) p->fftGkrCancel(); } } fft_guard{s->p.get(), false};
    auto chk = [&](int rc, const char *what) { if (rc != VP_OK) throw std::runtime_error(std::string(what) + ": " + vp_last_error(ctx)); };
    try {
        const layeredCircuit &C = s->circ->c;
        const int n = C.circuit[0].bitLength, ln = n - 6;
        if (s->tape.empty() || (int) s->ptape_fri.size() != ln) { set_err(err, errlen, "vph_draw_protocol_tape first"); return -1; }
        int max_bl = 0;
        for (auto &l : C.circuit) max_bl = std::max(max_bl, l.bitLength);
        using clk = std::chrono::high_resolution_clock;
        auto since = [](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
        const auto t0 = clk::now();
        static const bool fft_sync = getenv("VPH_FFT_GKR_SYNC") != nullptr;
        u64 nt = 0, nb = 0;
        s->p->gkrSizes(nt, nb);
        if (s->tape.size() != nt) throw std::runtime_error("the tape has the wrong length");
        s->pp_gkr.resize(nb); s->pp_all.resize(65); s->pp_roots.resize((size_t) 32 * ln); s->pp_final.resize(2048);
        chk(vp_set_deferred(ctx, defer ? 1 : 0), "vp_set_deferred");
        // ---- head: fft_gkr (depends on the verifier's draws only: queued on its own stream, its small launches run in the gaps of everything below;
        // VPH_FFT_GKR_SYNC=1: in the reference's place, between commit_public and the FRI folds) and commit_private (src/verifier.cpp:137) —
        // unless the previous pass queued them for this one and that commitment still stands
        uint64_t epoch = 0; int valid = 0;
        chk(vp_commit_private_state(ctx, &epoch, &valid), "vp_commit_private_state");
        const bool have_head = s->head_queued && valid && epoch == s->head_epoch;
        s->head_queued = false;
        auto t = clk::now();
        double s_fft = 0, s_priv = 0, s_gkr = 0, s_pub = 0, s_fri = 0;
        if (!have_head) {
            chk(vp_commit_private(ctx, s->pp_root_l), "vp_commit_private");
            s_priv = since(t);
        }
        // ---- GKR (src/verifier.cpp:144-169)
        t = clk::now();
        chk(vp_prove_gkr(ctx, reinterpret_cast<const vp_F *>(s->tape.data()), nt, s->pp_gkr.data(), nb, &s->pp_written), "vp_prove_gkr");
        s_gkr = since(t);
        // ---- fft_gkr (vpd_verifier.cpp:92) depends on the verifier's draws only.  It is queued on a stream of its own BEHIND the proof (vp_fft_gkr_begin starts
        // behind what the context has queued so far) and collected last: its ~100 few-workgroup launches then run beside commit_public's transforms, which do not
        // notice them.  Beside a leaf-hash launch they cost that launch a ninth round of workgroups (11.4 instead of 10.2 ms), beside the proof's graph 0.9 ms
        // of its critical path (measured, tools/pass_modes.py / tools/leaf_in_step.py).  VPH_FFT_GKR_SYNC=1: in the reference's place, on the main stream
        t = clk::now();
        if (!fft_sync) { s->p->fftGkrBegin(ln, s->ptape_fft); fft_guard.armed = true; }
        s_fft = since(t);
        // r_liu after the last Liu sumcheck = the last max_bl draws of the GKR tape (verifier::drawTape), its first n coordinates
        s->last_point.assign(s->tape.end() - max_bl, s->tape.end() - max_bl + n);
        // ---- commit_public on eq(r_liu, .) (:368-379)
        t = clk::now();
        chk(vp_commit_public_eq(ctx, reinterpret_cast<const vp_F *>(s->last_point.data()), n, reinterpret_cast<vp_F *>(&s->pp_inner),
                                reinterpret_cast<vp_F *>(s->pp_all.data()), s->pp_root_h), "vp_commit_public_eq");
        s_pub = since(t);
        if (fft_sync) { t = clk::now(); s->fft_gkr_msgs = s->p->fftGkr(ln, s->ptape_fft); s_fft = since(t); }
        // ---- FRI commit phase (vpd_verifier.cpp:44-74) + final codeword
        t = clk::now();
        chk(vp_fri_commit(ctx, reinterpret_cast<const vp_F *>(s->ptape_fri.data()), ln, s->pp_roots.data()), "vp_fri_commit");
        chk(vp_fri_final(ctx, reinterpret_cast<vp_F *>(s->pp_final.data())), "vp_fri_final");
        s_fri = since(t);
        int n_mine = 0;
        chk(vp_pending(ctx, &n_mine), "vp_pending");
        if (queue_next) {
            // the next pass's commit_private, behind this pass's folds in stream order; this pass's results are collected below while the device runs it
            chk(vp_commit_private(ctx, s->pp_root_next), "vp_commit_private (next pass)");
            uint64_t e2 = 0; int v2 = 0;
            chk(vp_commit_private_state(ctx, &e2, &v2), "vp_commit_private_state");
            s->head_queued = true; s->head_epoch = e2;
        }
        if (defer) {
            chk(vp_flush(ctx, n_mine), "vp_flush");
            double ms[5] = {0, 0, 0, 0, 0};
            chk(vp_phase_ms(ctx, ms), "vp_phase_ms");
            s_priv = ms[0] * 1e-3; s_gkr = ms[1] * 1e-3; s_pub = ms[2] * 1e-3; s_fri = (ms[3] + ms[4]) * 1e-3;
        }
        chk(vp_set_deferred(ctx, 0), "vp_set_deferred");      // the mode belongs to this pass: whatever the session calls next waits for its results as usual
        // fft_gkr is collected last: its launches had the whole pass to run beside the main stream's
        if (!fft_sync) { t = clk::now(); fft_guard.armed = false; s->fft_gkr_msgs = s->p->fftGkrEnd(ln); s_fft += since(t); }
        const uint64_t total = 32 + s->pp_written + 32 + 16 + 65 * 16;
        if (total > capacity || (fri_roots && roots_cap < s->pp_roots.size())) { set_err(err, errlen, "output buffer too small"); return -1; }
        uint8_t *o = transcript;
        memcpy(o, have_head ? s->pp_root_next : s->pp_root_l, 32); o += 32;      // (a head queued by the previous pass left its root in pp_root_next)
        memcpy(o, s->pp_gkr.data(), s->pp_written); o += s->pp_written;
        memcpy(o, s->pp_root_h, 32); o += 32;
        memcpy(o, &s->pp_inner, 16); o += 16;
        memcpy(o, s->pp_all.data(), 65 * 16);
        if (n_written) *n_written = total;
        s->fri_roots = s->pp_roots;
        s->fri_final = s->pp_final;
        s->fri_r = s->ptape_fri;
        if (fri_roots) memcpy(fri_roots, s->fri_roots.data(), s->fri_roots.size());
        if (final_pairs) memcpy(final_pairs, s->fri_final.data(), s->fri_final.size() * sizeof(F));
        s->p->addProveTime(s_gkr);
        if (sec) { sec[0] = since(t0); sec[1] = s_priv; sec[2] = s_gkr; sec[3] = s_pub; sec[4] = s_fft; sec[5] = s_fri; }
        return 0;
    } catch (const std::exception &e) {
        (void) vp_flush(ctx, -1);                        // nothing of a failed pass stays queued
        (void) vp_set_deferred(ctx, 0);
        s->head_queued = false;
        set_err(err, errlen, e.what());
        return -2;
    }
}
int vph_prove_protocol(vph_session *s, uint8_t *transcript, uint64_t capacity, uint64_t *n_written, uint8_t *fri_roots, uint64_t roots_cap,
                       uint64_t *final_pairs, double sec[6], char *err, int errlen) {
    return vph_prove_protocol_ex(s, transcript, capacity, n_written, fri_roots, roots_cap, final_pairs, sec, 0, err, errlen);
}

int vph_verify_transcript(vph_circuit *c, const uint8_t *transcript, uint64_t n, int skip_predicates) {
    try {
        F::init();
        verifier v(nullptr, c->c);
        const std::vector<F> tape = v.drawTape();
        v.skip_predicates = skip_predicates != 0;
        std::vector<uint8_t> tr(transcript, transcript + n);
        return v.check(tape, tr) ? 0 : 1;
    } catch (const std::exception &) { return 1; }
}

uint64_t vph_transcript_bytes(vph_session *s) {
    u64 a = 0, b = 0;
    s->p->gkrSizes(a, b);
    return b;
}

}  // extern "C"
