// The VERIFIER half of lib/virgo's fft_circuit_gkr::fft_gkr (lib/virgo/src/fft_circuit_GKR.cpp): the reference interleaves prover and
// verifier in one function; with the prover on the device (vp_fft_gkr, include/vpgpu.h) what remains on the host is every check that
// function makes on the prover's messages — the per-round identity p(0) + p(1) = claim (:261-266), the closed forms of the wiring
// predicates at the end of the addition layer (:285-308), the multiplication layer (:405-446) and each inverse-FFT depth (:639-752), and
// the claim hand-over between layers (:307, :445, :449-456, :763-765).  Field-type generic (Fe needs Fe(long long), + - *, == / !=):
// virgo-plus_amd/host/verifier.cpp instantiates it with vph::fieldElement, the forwarding file for the reference's fft_circuit_GKR.cpp
//  with the reference's own virgo::fieldElement.  `tape` is the draw sequence in the reference's order, `msgs` what vp_fft_gkr returned (layouts: include/vpgpu.h).
#pragma once
#include <cstddef>
#include <vector>

namespace vph {

struct FftGkrLayout {
    int lg;
    size_t r, x, r0, r1, ru_a, rv_a, ru_m, rv_m, dep0, n_tape, n_msgs;
    explicit FftGkrLayout(int lg_) : lg(lg_) {
        r = 0; x = lg; r0 = x + 64; r1 = r0 + lg + 10; ru_a = r1 + lg + 10; rv_a = ru_a + lg + 6; ru_m = rv_a + lg + 6; rv_m = ru_m + lg;
        dep0 = rv_m + lg; n_tape = dep0 + (size_t) lg * (2 * lg + 2);
        n_msgs = 64 + 3 * ((size_t) 2 * lg * lg + 2 * lg + 6) + 2 + 2 * (size_t) lg;
    }
    size_t ru_d(int d) const { return dep0 + (size_t) d * (2 * lg + 2); }
    size_t rv_d(int d) const { return ru_d(d) + lg; }
    size_t alpha_d(int d) const { return rv_d(d) + lg; }
    size_t beta_d(int d) const { return alpha_d(d) + 1; }
};

// inv_rou = getRootOfUnity(lg)^-1 in the caller's field type.  Returns true iff every check of the reference's embedded verifier holds.
template <class Fe>
bool fft_gkr_check(int lg, const Fe *tape, size_t n_tape, const Fe *msgs, size_t n_msgs, const Fe &inv_rou) {
    const FftGkrLayout L(lg);
    if (lg < 1 || n_tape != L.n_tape || n_msgs != L.n_msgs) return false;
    const Fe one(1ll), zero(0ll);
    size_t pos = 0;
    // claim on the 64 outputs at r_0[0..6)  (V_output, :121-136)
    std::vector<Fe> o(msgs, msgs + 64);
    pos = 64;
    for (int i = 0; i < 6; ++i) {
        const Fe ri = tape[L.r0 + i];
        for (size_t j = 0; j < o.size() / 2; ++j) o[j] = o[2 * j] * (one - ri) + o[2 * j + 1] * ri;
        o.resize(o.size() / 2);
    }
    Fe claim = o[0];
    // rounds of one sumcheck: p(0) + p(1) == claim, claim <- p(r)
    auto rounds = [&](int n, const Fe *ch) -> bool {
        for (int k = 0; k < n; ++k) {
            const Fe a = msgs[pos], b = msgs[pos + 1], c = msgs[pos + 2];
            pos += 3;
            if (c + (a + b + c) != claim) return false;
            claim = (a * ch[k] + b) * ch[k] + c;
        }
        return true;
    };
    Fe alpha = one, beta = zero;
    std::vector<Fe> r0(tape + L.r0, tape + L.r0 + lg + 10), r1(tape + L.r1, tape + L.r1 + lg + 10);
    {   // addition layer
        const int n = lg + 6;
        const Fe *ru = tape + L.ru_a, *rv = tape + L.rv_a;
        if (!rounds(n, ru)) return false;
        const Fe vu = msgs[pos++];
        Fe s = zero;
        for (int i = 0; i < 64; ++i) {
            Fe g0 = alpha, g1 = beta, u = one;
            for (int j = 0; j < 6; ++j) {
                if ((i >> j) & 1) { g0 = g0 * r0[j]; g1 = g1 * r1[j]; u = u * ru[lg + j]; }
                else { g0 = g0 * (one - r0[j]); g1 = g1 * (one - r1[j]); u = u * (one - ru[lg + j]); }
            }
            s = s + (g0 + g1) * u;
        }
        if (claim != s * vu) return false;
        for (int i = 0; i < n; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
        claim = alpha * vu;
    }
    {   // multiplication layer
        const Fe *ru = tape + L.ru_m, *rv = tape + L.rv_m;
        if (!rounds(lg, ru)) return false;
        const Fe vu = msgs[pos++];
        Fe s = zero;
        for (int i = 0; i < 64; ++i) {
            Fe g0 = alpha, g1 = beta;
            for (int j = 0; j < 6; ++j) {
                if ((i >> j) & 1) { g0 = g0 * r0[lg + j]; g1 = g1 * r1[lg + j]; }
                else { g0 = g0 * (one - r0[lg + j]); g1 = g1 * (one - r1[lg + j]); }
            }
            Fe u0 = one, u1 = one, x = tape[L.x + i];
            for (int j = 0; j < lg; ++j) {
                u0 = u0 * (r0[j] * ru[j] * x + (one - r0[j]) * (one - ru[j]));
                u1 = u1 * (r1[j] * ru[j] * x + (one - r1[j]) * (one - ru[j]));
                x = x * x;
            }
            s = s + g0 * u0 + g1 * u1;
        }
        if (claim != s * vu) return false;
        for (int i = 0; i < lg; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
        claim = alpha * vu;
    }
    claim = claim * Fe((long long) 1 << lg);                                    // the scaling layer (:449-456)
    for (int dep = 0; dep < lg; ++dep) {
        const Fe *ru = tape + L.ru_d(dep), *rv = tape + L.rv_d(dep);
        if (!rounds(lg, ru)) return false;
        const Fe vu = msgs[pos++];
        if (!rounds(lg, rv)) return false;
        const Fe vv = msgs[pos++];
        Fe w = inv_rou;
        for (int q = 0; q < dep; ++q) w = w * w;
        const int lj = dep, lk = lg - dep - 1;
        const Fe hu = (one - ru[lj]) * rv[lj];
        Fe uA0 = (one - r0[lg - 1]) * hu * alpha, uA1 = (one - r1[lg - 1]) * hu * beta, vA0 = uA0, vA1 = uA1;
        Fe uB0 = r0[lg - 1] * hu * alpha, uB1 = r1[lg - 1] * hu * beta, vB0 = uB0, vB1 = uB1;
        Fe xx = w;
        for (int i = 0; i < lk; ++i) {
            const Fe p0 = r0[lj + i] * ru[lj + 1 + i] * rv[lj + 1 + i], q0 = (one - r0[lj + i]) * (one - ru[lj + 1 + i]) * (one - rv[lj + 1 + i]);
            const Fe p1 = r1[lj + i] * ru[lj + 1 + i] * rv[lj + 1 + i], q1 = (one - r1[lj + i]) * (one - ru[lj + 1 + i]) * (one - rv[lj + 1 + i]);
            uA0 = uA0 * (p0 + q0); uA1 = uA1 * (p1 + q1); uB0 = uB0 * (p0 + q0); uB1 = uB1 * (p1 + q1);
            vA0 = vA0 * (p0 * xx + q0); vA1 = vA1 * (p1 * xx + q1); vB0 = vB0 * (q0 + p0 * xx); vB1 = vB1 * (q1 + p1 * xx);
            xx = xx * xx;
        }
        for (int i = 0; i < lj; ++i) {
            const Fe e0 = r0[i] * ru[i] * rv[i] + (one - r0[i]) * (one - ru[i]) * (one - rv[i]);
            const Fe e1 = r1[i] * ru[i] * rv[i] + (one - r1[i]) * (one - ru[i]) * (one - rv[i]);
            uA0 = uA0 * e0; vA0 = vA0 * e0; uB0 = uB0 * e0; vB0 = vB0 * e0;
            uA1 = uA1 * e1; vA1 = vA1 * e1; uB1 = uB1 * e1; vB1 = vB1 * e1;
        }
        if (claim != (uA0 + uA1 + uB0 + uB1) * vu + (vA0 + vA1 - vB0 - vB1) * vv) return false;
        for (int i = 0; i < lg; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
        alpha = tape[L.alpha_d(dep)]; beta = tape[L.beta_d(dep)];
        claim = alpha * vu + beta * vv;
    }
    return pos == n_msgs;
}

}  // namespace vph
