// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// Driver for the real reference (sources compiled where they lie under
// /root/reference by oracle/Makefile; nothing is copied).  It replaces only
// the reference's `main` (src/main.cpp:145-159, renamed ref_main at compile
// time) so that
//   * the transcript of prover messages can be written (see ref_hook.hpp),
//   * B-fold block-replicated SHA-256 circuits can be fed to the reference's
//     own DAG_to_layered() without a 224 MB .pws text file and its regex
//     parser, and
//   * layeredCircuit::randomize() (src/circuit.cpp:17-41) is reachable
//     (no CLI reaches it in the reference).
// Everything downstream of circuit construction is the reference's own code:
// DAG_to_layered, F::init, subsetInit, prover, verifier::verify.
//
//   * the messages of lib/virgo's self-contained fft_gkr (lib/virgo/src/fft_circuit_GKR.cpp:833-849, run by verify_poly_commitment,
//     vpd_verifier.cpp:92) can be recorded: that function sends nothing to its caller, so the record is taken at LINK time —
//     `ld --wrap` (oracle/Makefile) routes fft_circuit_GKR.o's calls to virgo::quadratic_poly::eval / virgo::linear_poly::eval (out-of-line
//     in lib/virgo/src/polynomial.cpp) and vpd_verifier.o's call to fft_gkr through the wrappers below, which forward to the real
//     functions unchanged.  Every round polynomial is evaluated three times in a row (at 0, at 1, at the challenge: :261,263,265 and
//     their copies), every phase ends with one linear_poly::eval (v_u / v_v, :268): the third evaluation of each triple and every linear
//     evaluation, in call order, preceded by the circuit's 64 outputs, is the record (--dump-fft).  No reference line is edited.
// Usage:
//   ref_run --pws FILE [--blocks B] [--ref-parser] [--pc 0|1] [--dump OUT] [--dump-fri OUT] [--dump-fft OUT] [--seed S]
//   ref_run --randomize LAYERS LOG_SIZE      [--pc 0|1] [--dump OUT] [--seed S]
//   ref_run --fft-gkr LG [--dump-fft OUT]    F::init(), then fft_gkr(LG) alone (its own prover + verifier)
//   ref_run --pc-masked IN [--dump OUT] [--dump-fri OUT]      lib/virgo's commitment ALONE with NON-ZERO masks (round 6): the reference's prover / verifier
//       never pass one (src/prover.cpp:526, src/verifier.cpp:375-377), so poly_commit_prover::commit_private_array / commit_public_array
//       (lib/virgo/src/poly_commit.h:41-349) and commit_phase (vpd_verifier.cpp:44-74) are called directly.  IN: i32 n, i32 m, then values[2^n], pub[2^n],
//       pri_mask[m], pub_mask[m] as (u64 real, u64 img) pairs.  --dump: root_l | root_h | all_sum[65] | openings (130 values each: 65 pairs, the mask
//       slice's last) of oracle 0 and 1 at leaves 0, 5, M/2 - 1 and of FRI levels 0, 2 at position 3; --dump-fri: as for a protocol run.
//       --verify: then poly_commit_verifier::verify_poly_commitment (vpd_verifier.cpp:76-328) with the public mask decides on that commitment; exit code 0 = accepted.
#include "verifier.h"
#include "inputCircuit.hpp"
#include "virgo/src/polynomial.h"          // lib/virgo's own polynomial classes (namespace virgo), the ones fft_circuit_GKR.cpp uses
#include "virgo/src/fft_circuit_GKR.h"
#include "virgo/src/poly_commit.h"
#include <fstream>
#include <string>
#include <chrono>

vp_ref_state g_vp_ref;
static FILE *g_fri_dump = nullptr;

// fri::commit_phase_step (lib/virgo/src/fri.cpp:289-424) is compiled as ref_commit_phase_step (oracle/Makefile);
// this wrapper keeps its name and records (challenge, Merkle root) of every FRI commit step.
namespace virgo { namespace fri {
__hhash_digest ref_commit_phase_step(fieldElement r);
__hhash_digest commit_phase_step(fieldElement r) {
    __hhash_digest d = ref_commit_phase_step(r);
    if (g_fri_dump) {
        unsigned long long w[2] = {r.real, r.img};
        fwrite(w, 8, 2, g_fri_dump);
        fwrite(&d, 32, 1, g_fri_dump);
    }
    return d;
}
} }

// ---- fft_gkr record (see the header comment).  The reference's globals of that translation unit are namespace-scope with external
// linkage; `C` holds the layer values (fft_circuit_GKR.cpp:9-13) and is declared here with the same definition to read the outputs.
namespace virgo { namespace fft_circuit_gkr {
class circuit { public: std::vector<fieldElement *> circuit_val; std::vector<int> size; };
extern circuit C;
} }
static FILE *g_fft_dump = nullptr;
static bool g_fft_rec = false;
static unsigned long g_fft_qcalls = 0;
static std::vector<unsigned long long> g_fft_msgs;
static inline void fft_put(const virgo::fieldElement &x) { g_fft_msgs.push_back(x.real); g_fft_msgs.push_back(x.img); }
extern "C" {
virgo::fieldElement __real__ZNK5virgo14quadratic_poly4evalERKNS_12fieldElementE(const virgo::quadratic_poly *, const virgo::fieldElement &);
virgo::fieldElement __real__ZNK5virgo11linear_poly4evalERKNS_12fieldElementE(const virgo::linear_poly *, const virgo::fieldElement &);
int __real__ZN5virgo15fft_circuit_gkr7fft_gkrEiRdRiS1_(int, double &, int &, double &);
virgo::fieldElement __wrap__ZNK5virgo14quadratic_poly4evalERKNS_12fieldElementE(const virgo::quadratic_poly *self, const virgo::fieldElement &x) {
    if (g_fft_rec && (g_fft_qcalls++ % 3) == 2) { fft_put(self->a); fft_put(self->b); fft_put(self->c); }
    return __real__ZNK5virgo14quadratic_poly4evalERKNS_12fieldElementE(self, x);
}
virgo::fieldElement __wrap__ZNK5virgo11linear_poly4evalERKNS_12fieldElementE(const virgo::linear_poly *self, const virgo::fieldElement &x) {
    virgo::fieldElement v = __real__ZNK5virgo11linear_poly4evalERKNS_12fieldElementE(self, x);
    if (g_fft_rec) fft_put(v);
    return v;
}
int __wrap__ZN5virgo15fft_circuit_gkr7fft_gkrEiRdRiS1_(int lg, double &vt, int &ps, double &pt) {
    g_fft_rec = true; g_fft_qcalls = 0; g_fft_msgs.clear();
    const int rc = __real__ZN5virgo15fft_circuit_gkr7fft_gkrEiRdRiS1_(lg, vt, ps, pt);
    g_fft_rec = false;
    if (g_fft_dump) {
        using namespace virgo::fft_circuit_gkr;
        const virgo::fieldElement *out = C.circuit_val.back();                 // the 64 evaluations (fft_circuit_GKR.cpp:91-100)
        for (int i = 0; i < 64; ++i) { unsigned long long w[2] = {out[i].real, out[i].img}; fwrite(w, 8, 2, g_fft_dump); }
        fwrite(g_fft_msgs.data(), 8, g_fft_msgs.size(), g_fft_dump);
        fflush(g_fft_dump);
    }
    fprintf(stdout, "fft_gkr lg %d p_time %lf v_time %lf proof_bytes %d recorded_F %lu\n", lg, pt, vt, ps, (unsigned long) (g_fft_msgs.size() / 2));
    return rc;
}
}

#include "ref_replicate.hpp"

// Streaming 2x64-bit hash of the layered circuit in a canonical serialisation
// (shared with oracle/vp_oracle.cpp and the product loader's tests).
struct chash {
    unsigned long long a = 1469598103934665603ull, b = 0x9e3779b97f4a7c15ull;
    void u64v(unsigned long long x) {
        for (int i = 0; i < 8; ++i) { a ^= (x >> (8 * i)) & 0xff; a *= 1099511628211ull; }
        b = (b ^ x) * 0xff51afd7ed558ccdull; b ^= b >> 32;
    }
};
static void circuit_hash(const layeredCircuit &C, unsigned long long out[2]) {
    chash h;
    h.u64v(C.size);
    for (int i = 0; i < C.size; ++i) {
        auto &L = C.circuit[i];
        h.u64v(L.size); h.u64v((long long) L.bitLength);
        for (u64 g = 0; g < L.size; ++g) {
            auto &G = L.gates[g];
            h.u64v((long long) G.ty); h.u64v((long long) G.l); h.u64v(G.u); h.u64v(G.v); h.u64v(G.lv);
            h.u64v(G.c.real); h.u64v(G.c.img); h.u64v(G.is_assert ? 1 : 0);
        }
        h.u64v((long long) L.maxDadBitLength); h.u64v(L.maxDadSize);
        for (int j = 0; j < i; ++j) {
            h.u64v(L.dadSize[j]);
            h.u64v(L.dadSize[j] ? (long long) L.dadBitLength[j] : -1ll);  // (int)log2(0) is UB in the reference
            for (u64 k = 0; k < L.dadSize[j]; ++k) h.u64v(L.dadId[j][k]);
        }
    }
    out[0] = h.a; out[1] = h.b;
}

int main(int argc, char **argv) {
    const char *pws = nullptr, *dump = nullptr, *custom = nullptr;
    int blocks = 1, ref_parser = 0, rnd_layers = 0, rnd_log = 0, fft_lg = 0;
    const char *pc_masked = nullptr;
    bool pc_masked_verify = false;
    long seed = -1;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--pws" && i + 1 < argc) pws = argv[++i];
        else if (a == "--blocks" && i + 1 < argc) blocks = atoi(argv[++i]);
        else if (a == "--ref-parser") ref_parser = 1;
        else if (a == "--pc" && i + 1 < argc) g_vp_ref.pc_on = atoi(argv[++i]);
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (a == "--seed" && i + 1 < argc) seed = atol(argv[++i]);
        else if (a == "--dump-fri" && i + 1 < argc) { g_fri_dump = fopen(argv[++i], "wb"); if (!g_fri_dump) { perror("dump-fri"); return 2; } }
        else if (a == "--dump-fft" && i + 1 < argc) { g_fft_dump = fopen(argv[++i], "wb"); if (!g_fft_dump) { perror("dump-fft"); return 2; } }
        else if (a == "--fft-gkr" && i + 1 < argc) fft_lg = atoi(argv[++i]);
        else if (a == "--randomize" && i + 2 < argc) { rnd_layers = atoi(argv[++i]); rnd_log = atoi(argv[++i]); }
        else if (a == "--custom" && i + 1 < argc) custom = argv[++i];
        else if (a == "--pc-masked" && i + 1 < argc) pc_masked = argv[++i];
        else if (a == "--verify") pc_masked_verify = true;
        else { fprintf(stderr, "bad arg %s\n", argv[i]); return 2; }
    }
    if (fft_lg > 0) {                          // fft_gkr alone, from the generator state F::init() leaves (srand(3396))
        F::init();
        double vt = 0, pt = 0; int ps = 0;
        virgo::fft_circuit_gkr::fft_gkr(fft_lg, vt, ps, pt);
        if (g_fft_dump) fclose(g_fft_dump);
        return 0;
    }
    if (pc_masked) {
        using namespace virgo;
        FILE *f = fopen(pc_masked, "rb");
        if (!f) { perror(pc_masked); return 2; }
        int n = 0, m = 0;
        if (fread(&n, 4, 1, f) != 1 || fread(&m, 4, 1, f) != 1 || n < 7 || n > 24 || m < 1) return 2;
        auto rd = [&](fieldElement *dst, size_t cnt) { for (size_t i = 0; i < cnt; ++i) { unsigned long long w[2]; if (fread(w, 8, 2, f) != 2) return false; dst[i].real = w[0]; dst[i].img = w[1]; } return true; };
        std::vector<fieldElement> values((size_t) 1 << n), pub((size_t) 1 << n), pri(m), pubm(m);
        if (!rd(values.data(), values.size()) || !rd(pub.data(), pub.size()) || !rd(pri.data(), m) || !rd(pubm.data(), m)) return 2;
        fclose(f);
        fieldElement::init();
        poly_commit::poly_commit_prover pp;
        __hhash_digest root_l = pp.commit_private_array(values.data(), n, pri);
        std::vector<fieldElement> all_sum(slice_number + 1);
        __hhash_digest root_h = pp.commit_public_array(pubm, pub.data(), n, fieldElement(0), all_sum.data());
        FILE *o = dump ? fopen(dump, "wb") : nullptr;
        auto wF = [&](const fieldElement &x) { if (o) { unsigned long long w[2] = {x.real, x.img}; fwrite(w, 8, 2, o); } };
        if (o) { fwrite(&root_l, 32, 1, o); fwrite(&root_h, 32, 1, o); }
        for (auto &x : all_sum) wF(x);
        const long long half = (1LL << (n + rs_code_rate - log_slice_number)) / 2;
        for (int oracle = 0; oracle < 2; ++oracle)
            for (long long leaf : {0LL, 5LL, half - 1}) {
                int ns = 0;
                auto r = fri::request_init_value_with_merkle(leaf, leaf + half, ns, oracle);
                for (auto &pr : r.first) { wF(pr.first); wF(pr.second); }
            }
        if (pc_masked_verify) {
            // the reference's OWN verifier on that commitment INSTEAD of the recorded commit phase (commit_phase runs once per commitment: fri.cpp's state is not re-entrant) (vpd_verifier.cpp:76-328 with the public mask; `processed` as src/verifier.cpp:346-358,371):
            // 33 random queries — Merkle paths of l, h and every FRI level, the first-round consistency of all 65 slices (the mask slice against the public
            // mask's polynomial and all_sum[64]), the later rounds, both final codewords.
            std::vector<fieldElement> processed((size_t) 1 << n);
            const int cs = 1 << (n - log_slice_number);
            for (int i = 0; i < slice_number; ++i)
                inverse_fast_fourier_transform(pub.data() + (size_t) i * cs, cs, cs, fieldElement::getRootOfUnity(n - log_slice_number), processed.data() + (size_t) i * cs);
            poly_commit::poly_commit_verifier pv;
            pv.p = &pp;
            double vt = 0, pt = 0; int ps = 0;
            const bool ok = pv.verify_poly_commitment(all_sum.data(), n, processed.data(), pubm, vt, ps, pt, root_l, root_h);
            fprintf(stdout, "pc-masked verify_poly_commitment %s proof_bytes %d\n", ok ? "ACCEPT" : "REJECT", ps);
            return ok ? 0 : 1;
        }
        poly_commit::ldt_commitment com = pp.commit_phase(n);          // the wrapper above records (challenge, root) of every step
        for (int lvl : {0, 2}) {
            if (lvl >= com.mx_depth) continue;
            int ns = 0;
            auto r = fri::request_step_commit(lvl, 3, ns);
            for (auto &pr : r.first) { wF(pr.first); wF(pr.second); }
        }
        if (o) fclose(o);
        if (g_fri_dump) {
            const int last = fri::current_step_no - 1;
            for (int k = 0; k < 16 * 128; ++k) { unsigned long long w[2] = {fri::cpd.rs_codeword[last][k].real, fri::cpd.rs_codeword[last][k].img}; fwrite(w, 8, 2, g_fri_dump); }
            for (int k = 0; k < 32; ++k) { unsigned long long w[2] = {fri::cpd.rs_codeword_msk[last][k].real, fri::cpd.rs_codeword_msk[last][k].img}; fwrite(w, 8, 2, g_fri_dump); }
            fclose(g_fri_dump);
        }
        fprintf(stdout, "pc-masked n %d mask %d steps %d mask_position_gap %d\n", n, m, com.mx_depth, poly_commit::mask_position_gap);
        return 0;
    }
    if (!pws && !rnd_layers && !custom) { fprintf(stderr, "need --pws, --randomize, --custom or --fft-gkr\n"); return 2; }
    if (seed >= 0) srandom((unsigned) seed);   // per-proof witness seed

    auto t0 = std::chrono::high_resolution_clock::now();
    if (custom) {
        // flat circuit file written by tests/golden/make_golden.py (tests/custom_circuits.py): int32 n_layers,
        // u64 sizes[n], then per gate: i32 ty, i32 l, u64 u, u64 v, u64 c.real, u64 c.img, u8 is_assert
        FILE *f = fopen(custom, "rb");
        if (!f) { perror(custom); return 2; }
        int nl = 0;
        if (fread(&nl, 4, 1, f) != 1) return 2;
        std::vector<u64> sz(nl);
        if (fread(sz.data(), 8, nl, f) != (size_t) nl) return 2;
        c.size = nl; c.circuit.resize(nl);
        for (int i = 0; i < nl; ++i) {
            auto &L = c.circuit[i];
            L.size = sz[i]; L.gates.resize(sz[i]);
            L.bitLength = 0; while ((1ULL << L.bitLength) < L.size) ++L.bitLength;
            for (u64 g = 0; g < sz[i]; ++g) {
                int ty, l; u64 u, v, cr, ci; unsigned char as;
                if (fread(&ty, 4, 1, f) != 1 || fread(&l, 4, 1, f) != 1 || fread(&u, 8, 1, f) != 1 || fread(&v, 8, 1, f) != 1 ||
                    fread(&cr, 8, 1, f) != 1 || fread(&ci, 8, 1, f) != 1 || fread(&as, 1, 1, f) != 1) return 2;
                F cc; cc.real = cr; cc.img = ci;
                L.gates[g] = gate((gateType) ty, l, u, v, cc, as != 0);
            }
        }
        fclose(f);
    } else if (rnd_layers) {
        c = layeredCircuit::randomize(rnd_layers, rnd_log);    // src/circuit.cpp:17
    } else {
        in_circuit_dag.clear();
        if (ref_parser) {
            if (blocks != 1) { fprintf(stderr, "--ref-parser reads the file as is\n"); return 2; }
            std::ifstream in(pws);
            parse(in);                                         // src/main.cpp:176
        } else {
            populate_replicated(pws, blocks);
        }
        DAG_to_layered();                                      // src/main.cpp:15
    }
    F::init();                                                 // src/main.cpp:152
    c.subsetInit();                                            // src/main.cpp:153
    auto t1 = std::chrono::high_resolution_clock::now();
    unsigned long long ch[2];
    circuit_hash(c, ch);
    u64 ngates = 0;
    for (int i = 0; i < c.size; ++i) ngates += c.circuit[i].size;
    fprintf(stdout, "circuit layers %d gates %llu hash %016llx%016llx build_sec %.3f\n", c.size, ngates, ch[0], ch[1],
            std::chrono::duration<double>(t1 - t0).count());

    if (dump) { g_vp_ref.dump = fopen(dump, "wb"); if (!g_vp_ref.dump) { perror(dump); return 2; } }
    prover p(c);                                               // src/main.cpp:154 (evaluates the circuit)
    verifier v(&p, c);                                         // src/main.cpp:155
    bool ok = false;
    auto t2 = std::chrono::high_resolution_clock::now();
    try {
        ok = v.verify();                                       // src/main.cpp:156
    } catch (vp_ref_pc_off_stop &) {
        // PC off: the GKR part is complete and was checked round by round.
        ok = true;
        fprintf(stdout, "Prove Time %lf\n", p.proveTime());
        fprintf(stdout, "proof size = %lf kb\n", p.proofSize());
    }
    auto t3 = std::chrono::high_resolution_clock::now();
    if (g_vp_ref.dump) fclose(g_vp_ref.dump);
    if (g_fft_dump) fclose(g_fft_dump);
    if (g_fri_dump) {
        // final codeword of the commit phase (fri::commit_phase_final, fri.cpp:426-431): 32 values per slice,
        // interleaved [i << 7 | slice << 1 | hi] for i < 16, followed by the mask codeword (32 values)
        using namespace virgo;
        const int last = fri::current_step_no - 1;
        if (last >= 0) {
            for (int k = 0; k < 16 * 128; ++k) { unsigned long long w[2] = {fri::cpd.rs_codeword[last][k].real, fri::cpd.rs_codeword[last][k].img}; fwrite(w, 8, 2, g_fri_dump); }
            for (int k = 0; k < 32; ++k) { unsigned long long w[2] = {fri::cpd.rs_codeword_msk[last][k].real, fri::cpd.rs_codeword_msk[last][k].img}; fwrite(w, 8, 2, g_fri_dump); }
        }
        fclose(g_fri_dump);
    }
    fprintf(stdout, "mult counter %d, add counter %d\n", F::multCounter, F::addCounter);   // src/main.cpp:157
    fprintf(stdout, "rounds %lu verify_wall_sec %.3f ok %d\n", g_vp_ref.rounds,
            std::chrono::duration<double>(t3 - t2).count(), ok ? 1 : 0);
    return ok ? 0 : 1;
}
