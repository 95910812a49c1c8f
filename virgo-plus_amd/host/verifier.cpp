#include "verifier.hpp"

#include <cstdio>
#include <cstring>

void initBetaTable(std::vector<F> &beta, int n, const std::vector<F>::const_iterator &r, const F &init) {
    if (n < 0) return;
    if (beta.size() < (1ull << n)) beta.resize(1ull << n);
    // direct doubling: after step k the first 2^k entries hold init * eq(r[0..k), .)
    beta[0] = init;
    for (int k = 0; k < n; ++k) {
        const u64 half = 1ull << k;
        for (u64 j = 0; j < half; ++j) {
            const F t = beta[j] * r[k];
            beta[j | half] = t;
            beta[j] = beta[j] - t;
        }
    }
}

verifier::verifier(prover *pr, const layeredCircuit &cir) : p(pr), C(cir) {       // verifier.cpp:12-48
    final_claims_v.resize(C.size);
    for (int i = 1; i < C.size; ++i) final_claims_v[i].assign(i, F_ZERO);
    for (auto &v : coeff_r) v.assign(C.size, F_ZERO);
    r_v.resize(C.size + 2);
    if (p) p->init();
    int max_dad_bl = 0;
    for (auto &l : C.circuit) { max_bl = std::max(max_bl, l.bitLength); max_dad_bl = std::max(max_dad_bl, l.maxDadBitLength); }
    beta_g.resize(1ull << std::max(max_bl, max_dad_bl));
    beta_u.resize(1ull << max_bl);
    beta_v.resize(1ull << max_bl);
    r_u.assign(max_bl, F_ZERO);
    r_liu.assign(max_bl, F_ZERO);
    for (int i = 1; i < C.size; ++i)
        if (C.circuit[i].maxDadBitLength != -1) r_v[i].assign(C.circuit[i].maxDadBitLength, F_ZERO);
    sig.assign(C.size, F_ZERO);
}

F verifier::draw() {
    if (replay) return (*rtape)[tape_pos++];
    F x = F::random();
    tape_.push_back(x);
    return x;
}
void verifier::putF(const F &x) {
    unsigned long long w[2] = {x.real, x.img};
    const uint8_t *b = reinterpret_cast<const uint8_t *>(w);
    tr.insert(tr.end(), b, b + 16);
}
F verifier::nextF() {
    F x;
    if (tr_pos + 16 > rtr->size()) throw std::runtime_error("transcript too short");
    unsigned long long w[2];
    memcpy(w, rtr->data() + tr_pos, 16);
    tr_pos += 16;
    x.real = w[0]; x.img = w[1];
    return x;
}
quadratic_poly verifier::nextPoly(int phase, const F &prev) {
    quadratic_poly q;
    if (replay) { q.a = nextF(); q.b = nextF(); q.c = nextF(); }
    else q = phase == 1 ? p->sumcheckUpdatePhase1(prev) : phase == 2 ? p->sumcheckUpdatePhase2(prev) : p->sumcheckLiuUpdate(prev);
    putF(q.a); putF(q.b); putF(q.c);
    return q;
}

std::vector<F> verifier::drawTape() {           // same draws as run(), without a prover
    std::vector<F> t;
    auto take = [&](size_t n) { for (size_t i = 0; i < n; ++i) t.push_back(F::random()); };
    take(C.circuit[C.size - 1].bitLength);                       // verifier.cpp:144
    for (int i = C.size - 1; i; --i) {
        take(max_bl);                                            // r_u            :196
        take(1);                                                 // assert_random  :202
        if (C.circuit[i].maxDadBitLength != -1) take(C.circuit[i].maxDadBitLength);   // r_v[i] :236
        take(C.size);                                            // sig            :278
        take(max_bl);                                            // r_liu          :279
    }
    return t;
}

bool verifier::verify() {
    if (!p) throw std::runtime_error("verify(): no prover attached");
    replay = false; tape_.clear(); tr.clear();
    return run();
}
bool verifier::check(const std::vector<F> &tape, const std::vector<uint8_t> &transcript) {
    replay = true; rtape = &tape; rtr = &transcript; tape_pos = 0; tr_pos = 0; tr.clear();
    bool ok = run();
    return ok && tr_pos == transcript.size() && tape_pos == tape.size();
}

bool verifier::run() {                          // verifier.cpp:134-169 (GKR part)
    for (int i = 0; i < C.circuit[C.size - 1].bitLength; ++i) r_liu[i] = draw();
    F previousSum;
    if (replay) previousSum = nextF();
    else {
        previousSum = p->Vres(r_liu.begin(), C.circuit[C.size - 1].bitLength);
        p->sumcheckInitAll(r_liu.begin());
    }
    putF(previousSum);
    for (int i = C.size - 1; i; --i) {
        if (!replay) p->sumcheckInit();
        if (!verifyPhase1(i, previousSum)) return false;
        if (C.circuit[i].maxDadBitLength != -1 && !verifyPhase2(i, previousSum)) return false;
        if (!(replay && skip_predicates)) {
            verify_timer.start();
            const F test_value = getFinalValue(i, final_claim_u, final_claims_v[i]);
            verify_timer.stop();
            if (previousSum != test_value) {
                fprintf(stderr, "Verification fail, semi final, circuit level %d\n", i);
                return false;
            }
        }
        if (!verifyLiu(i, previousSum)) return false;
    }
    return checkInput(previousSum);
}

bool verifier::verifyPhase1(int layer_id, F &previousSum) {      // verifier.cpp:191-229
    const layer &pre = C.circuit[layer_id - 1];
    for (auto &x : r_u) x = draw();
    F previousRandom = F_ZERO;
    assert_random = draw();
    if (!replay) p->sumcheckInitPhase1(assert_random);
    for (int j = 0; j < pre.bitLength; ++j) {
        const quadratic_poly poly = nextPoly(1, previousRandom);
        if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
            fprintf(stderr, "Verification fail, phase1, circuit %d, current bit %d\n", layer_id, j);
            return false;
        }
        previousRandom = r_u[j];
        previousSum = poly.eval(r_u[j]);
    }
    if (replay) final_claim_u = nextF(); else p->sumcheckFinalize1(previousRandom, final_claim_u);
    putF(final_claim_u);
    if (!(replay && skip_predicates)) { verify_timer.start(); predicatePhase1(layer_id); verify_timer.stop(); }
    return true;
}

bool verifier::verifyPhase2(int layer_id, F &previousSum) {      // verifier.cpp:231-270
    for (auto &x : r_v[layer_id]) x = draw();
    F previousRandom = F_ZERO;
    if (!replay) p->sumcheckInitPhase2();
    for (int j = 0; j < C.circuit[layer_id].maxDadBitLength; ++j) {
        const quadratic_poly poly = nextPoly(2, previousRandom);
        if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
            fprintf(stderr, "Verification fail, phase2, circuit level %d, current bit %d\n", layer_id, j);
            return false;
        }
        previousRandom = r_v[layer_id][j];
        previousSum = poly.eval(previousRandom);
    }
    if (replay) for (int j = 0; j < layer_id; ++j) final_claims_v[layer_id][j] = nextF();
    else p->sumcheckFinalize2(previousRandom, final_claims_v[layer_id].begin());
    for (int j = 0; j < layer_id; ++j) putF(final_claims_v[layer_id][j]);
    if (!(replay && skip_predicates)) { verify_timer.start(); predicatePhase2(layer_id); verify_timer.stop(); }
    return true;
}

bool verifier::verifyLiu(int layer_id, F &previousSum) {         // verifier.cpp:272-337
    const int pre_layer_id = layer_id - 1;
    const layer &pre = C.circuit[pre_layer_id];
    for (auto &x : sig) x = draw();
    for (auto &x : r_liu) x = draw();
    previousSum = sig[0] * final_claim_u;
    for (int j = layer_id; j < C.size; ++j)
        if (C.circuit[j].dadSize[pre_layer_id])                  // an empty subset's claim is zero
            previousSum += sig[j - pre_layer_id] * final_claims_v[j][pre_layer_id];
    if (!replay) p->sumcheckInitLiu(sig.begin());
    F previousRandom = F_ZERO;
    for (int j = 0; j < pre.bitLength; ++j) {
        const quadratic_poly poly = nextPoly(3, previousRandom);
        if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
            fprintf(stderr, "Liu fail, circuit %d, current bit %d\n", layer_id, j);
            return false;
        }
        previousRandom = r_liu[j];
        previousSum = poly.eval(previousRandom);
    }
    F vr;
    if (replay) vr = nextF(); else p->sumcheckLiuFinalize(previousRandom, vr);
    putF(vr);
    verify_timer.start();
    F gr = F_ZERO;
    initBetaTable(beta_u, pre.bitLength, r_liu.begin(), F_ONE);
    initBetaTable(beta_g, pre.bitLength, r_u.begin(), sig[0]);
    for (u64 g = 0; g < pre.size; ++g) gr = gr + beta_g[g] * beta_u[g];
    for (int j = layer_id; j < C.size; ++j) {
        const layer &Lj = C.circuit[j];
        if (!Lj.dadSize[pre_layer_id]) continue;
        initBetaTable(beta_g, Lj.dadBitLength[pre_layer_id], r_v[j].begin(), sig[j - pre_layer_id]);
        for (u64 g = 0; g < Lj.dadSize[pre_layer_id]; ++g) gr = gr + beta_g[g] * beta_u[Lj.dadId[pre_layer_id][g]];
    }
    const bool ok = (vr * gr == previousSum);
    verify_timer.stop();
    if (!ok) { fprintf(stderr, "Liu fail, semi final, circuit %d\n", layer_id); return false; }
    previousSum = vr;
    return true;
}

void verifier::predicatePhase1(int layer_id) {                   // verifier.cpp:50-56,63-90
    const layer &cur = C.circuit[layer_id];
    initBetaTable(beta_g, cur.bitLength, r_liu.begin(), F_ONE);
    for (u64 g = 0; g < cur.size; ++g) if (cur.gates[g].is_assert) beta_g[g] *= assert_random;        // verifier.cpp:53-54
    initBetaTable(beta_u, C.circuit[layer_id - 1].bitLength, r_u.begin(), F_ONE);
    for (int t : {(int) Copy, (int) Not, (int) Addc, (int) Mulc}) coeff_l[t] = F_ZERO;
    bias = F_ZERO;
    for (u64 g = 0; g < cur.size; ++g) {
        const gate &G = cur.gates[g];
        switch (G.ty) {
            case Addc:
                bias += beta_g[g] * beta_u[G.u] * G.c;
                coeff_l[G.ty] += beta_g[g] * beta_u[G.u];
                break;
            case Not: case Copy: coeff_l[G.ty] += beta_g[g] * beta_u[G.u]; break;
            case Mulc: coeff_l[G.ty] += beta_g[g] * beta_u[G.u] * G.c; break;
            default: break;
        }
    }
    for (int t : {(int) Add, (int) Sub, (int) AntiSub, (int) Mul, (int) Naab, (int) AntiNaab, (int) Xor})
        std::fill(coeff_r[t].begin(), coeff_r[t].end(), F_ZERO);
}

void verifier::predicatePhase2(int layer_id) {                   // verifier.cpp:58-61,92-113
    const layer &cur = C.circuit[layer_id];
    initBetaTable(beta_v, cur.maxDadBitLength, r_v[layer_id].begin(), F_ONE);
    for (int t : {(int) Copy, (int) Not, (int) Addc, (int) Mulc}) coeff_l[t] *= beta_v[0];
    bias *= beta_v[0];
    for (u64 g = 0; g < cur.size; ++g) {
        const gate &G = cur.gates[g];
        switch (G.ty) {
            case Add: case Sub: case AntiSub: case Mul: case Naab: case AntiNaab: case Xor:
                coeff_r[G.ty][G.l] += beta_g[g] * beta_u[G.u] * beta_v[G.lv];
                break;
            default: break;
        }
    }
}

F verifier::getFinalValue(int layer_id, const F &cu, const std::vector<F> &cv) {   // verifier.cpp:115-132
    F res = coeff_l[Not] * (F_ONE - cu) + coeff_l[Copy] * cu + coeff_l[Addc] * cu + bias + coeff_l[Mulc] * cu;
    for (int j = 0; j < layer_id; ++j) {
        const F uv = cu * cv[j];
        res = res + coeff_r[Add][j] * (cu + cv[j]) + coeff_r[Sub][j] * (cu - cv[j]) + coeff_r[AntiSub][j] * (cv[j] - cu)
              + coeff_r[Mul][j] * uv + coeff_r[Naab][j] * (cv[j] - uv) + coeff_r[AntiNaab][j] * (cu - uv)
              + coeff_r[Xor][j] * (cu + cv[j] - F(2ll) * uv);
    }
    return res;
}

// With the polynomial commitment off, the last claim is checked directly against the input layer's MLE
// at r_liu (the role of verifyPoly, verifier.cpp:363-389).
bool verifier::checkInput(const F &claim) {
    const layer &L0 = C.circuit[0];
    std::vector<F> beta;
    initBetaTable(beta, L0.bitLength, r_liu.begin(), F_ONE);
    F acc = F_ZERO;
    for (u64 g = 0; g < L0.size; ++g) acc = acc + beta[g] * F((long long) L0.gates[g].u);
    if (acc != claim) { fprintf(stderr, "Verification fail, final input check fail.\n"); return false; }
    return true;
}
