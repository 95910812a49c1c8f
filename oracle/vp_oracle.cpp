// TEST INFRASTRUCTURE ONLY — see vp_oracle.h.  CPU restatement of the reference's GKR prover path.
// Plain, single-threaded, portable C++ (unsigned __int128 instead of the reference's mulx asm / AVX2).
// Every function cites the reference lines it follows; nothing here is shared with the product.
#include "vp_oracle.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

namespace {

typedef unsigned long long u64;
typedef unsigned __int128 u128;
using std::vector;

const u64 P = 2305843009213693951ull;   // 2^61 - 1, lib/virgo/src/fieldElement.cpp:7

// ----------------------------------------------------------------------------------------------------
// Field: F_p[i]/(i^2+1).  lib/virgo/src/fieldElement.cpp:34-104 (ops), :336-360 (Mersenne fold).
// Results are canonical, exactly like the reference's (every reference op ends in a conditional
// subtraction), so values are mathematically determined and any correct implementation is bit-equal.
// The op counters follow fieldElement.cpp:35-38,50-53,81-84,99-103 (unary minus counts twice: once
// in operator-() and once in the zero()-x it performs) but are 64-bit.
// ----------------------------------------------------------------------------------------------------
struct Counter { bool on = false; u64 mul = 0, add = 0; };
Counter g_cnt;

inline u64 red128(u128 x) {            // x < 2^125  ->  [0, p)
    u64 lo = (u64) x & P;
    u64 hi = (u64) (x >> 61);          // < 2^64
    u64 s = lo + (hi & P) + (hi >> 61);
    s = (s & P) + (s >> 61);
    if (s >= P) s -= P;
    return s;
}
inline u64 addm(u64 a, u64 b) { u64 s = a + b; return s >= P ? s - P : s; }
inline u64 subm(u64 a, u64 b) { return a >= b ? a - b : a + P - b; }
inline u64 mulm(u64 a, u64 b) { return red128((u128) a * b); }

struct F {
    u64 re, im;
    F() : re(0), im(0) {}
    F(long long x) : re(x >= 0 ? (u64) x : P + x), im(0) {}          // fieldElement.cpp:24-27
    F(u64 r, u64 i) : re(r), im(i) {}
    bool operator==(const F &o) const { return re == o.re && im == o.im; }
    bool operator!=(const F &o) const { return !(*this == o); }
};
const F F_ZERO = F(0ll), F_ONE = F(1ll);

inline F operator+(const F &a, const F &b) {
    if (g_cnt.on) ++g_cnt.add;
    return F(addm(a.re, b.re), addm(a.im, b.im));
}
inline F operator-(const F &a, const F &b) {
    if (g_cnt.on) ++g_cnt.add;
    return F(subm(a.re, b.re), subm(a.im, b.im));
}
inline F operator-(const F &a) {
    if (g_cnt.on) ++g_cnt.add;
    return F_ZERO - a;
}
inline F operator*(const F &a, const F &b) {
    if (g_cnt.on) ++g_cnt.mul;
    u64 ac = mulm(a.re, b.re), bd = mulm(a.im, b.im);
    u64 cross = red128((u128) (a.re + a.im) * (b.re + b.im));
    return F(subm(ac, bd), subm(subm(cross, ac), bd));
}
inline F &operator+=(F &a, const F &b) { a = a + b; return a; }
inline F &operator*=(F &a, const F &b) { a = a * b; return a; }

F fpow(F x, u128 e) {                   // fieldElement.cpp:322-334
    F ret = F_ONE, t = x;
    while (e) { if (e & 1) ret = ret * t; t = t * t; e >>= 1; }
    return ret;
}
F finv(const F &x) { return fpow(x, (u128) P * P - 2); }            // fieldElement.cpp:206-209
F root_of_unity(int log_order) {                                     // fieldElement.cpp:237-249
    F rou(2147483648ull, 1033321771269002680ull);
    for (int i = 0; i < 62 - log_order; ++i) rou = rou * rou;
    return rou;
}
u64 random_number() {                                                // fieldElement.cpp:362-367
    u64 ret = ::random() % 10;
    for (int i = 1; i < 20; ++i) ret = (ret * 10ull + (u64) (::random() % 10)) % P;
    return ret;
}
F frandom() {                                                        // fieldElement.cpp:119-124
    F r;
    r.re = random_number() % P;
    r.im = random_number() % P;
    return r;
}

// ----------------------------------------------------------------------------------------------------
// Polynomials.  src/polynomial.cpp:64-131.  The implicit F -> linear_poly conversion (a = 0) and the
// component-wise additions are kept because they determine the reference's op counts.
// ----------------------------------------------------------------------------------------------------
struct Quad;
struct Lin {
    F a, b;
    Lin() {}
    Lin(const F &x) : a(F_ZERO), b(x) {}
    Lin(const F &aa, const F &bb) : a(aa), b(bb) {}
    Lin operator+(const Lin &x) const { return Lin(a + x.a, b + x.b); }
    F eval(const F &x) const { return a * x + b; }
    Quad operator*(const Lin &x) const;
};
struct Quad {
    F a, b, c;
    Quad() {}
    Quad(const F &aa, const F &bb, const F &cc) : a(aa), b(bb), c(cc) {}
    Quad operator+(const Quad &x) const { return Quad(a + x.a, b + x.b, c + x.c); }
    F eval(const F &x) const { return ((a * x) + b) * x + c; }
};
Quad Lin::operator*(const Lin &x) const { return Quad(a * x.a, a * x.b + b * x.a, b * x.b); }
inline Lin interpolate(const F &zero_v, const F &one_v) { return Lin(one_v - zero_v, zero_v); }   // prover.cpp:10-12

// ----------------------------------------------------------------------------------------------------
// eq table.  src/utils.cpp:8-45.
// ----------------------------------------------------------------------------------------------------
void init_half_tables(vector<F> &bf, vector<F> &bs, const F *r, const F &init, int first_half, int second_half) {
    bf[0] = init;
    bs[0] = F_ONE;
    for (int i = 0; i < first_half; ++i)
        for (u64 j = 0; j < (1ull << i); ++j) {
            F t = bf[j] * r[i];
            bf[j | (1ull << i)] = t;
            bf[j] = bf[j] - t;
        }
    for (int i = 0; i < second_half; ++i)
        for (u64 j = 0; j < (1ull << i); ++j) {
            F t = bs[j] * r[i + first_half];
            bs[j | (1ull << i)] = t;
            bs[j] = bs[j] - t;
        }
}
void init_beta_table(vector<F> &beta, int n, const F *r, const F &init) {
    int first_half = n >> 1, second_half = n - first_half;
    u64 mask = (1ull << first_half) - 1;
    if (beta.size() < (1ull << n)) beta.resize(1ull << n);
    vector<F> bf(1ull << first_half), bs(1ull << second_half);
    if (init != F_ZERO) {
        init_half_tables(bf, bs, r, init, first_half, second_half);
        for (u64 i = 0; i < (1ull << n); ++i) beta[i] = bf[i & mask] * bs[i >> first_half];
    } else {
        for (u64 i = 0; i < (1ull << n); ++i) beta[i] = F_ZERO;
    }
}

// ----------------------------------------------------------------------------------------------------
// Circuit model.  src/circuit.h:11-47, src/inputCircuit.hpp:13-15.
// ----------------------------------------------------------------------------------------------------
enum GateType { Mul, Add, Sub, AntiSub, Naab, AntiNaab, Input, Mulc, Addc, Xor, Not, Copy };

struct Gate {
    int ty; int l; u64 u, v, lv; F c; bool is_assert;
    Gate() : ty(0), l(0), u(0), v(0), lv(0), is_assert(false) {}
    Gate(int t, int ll, u64 uu, u64 vv, const F &cc, bool as) : ty(t), l(ll), u(uu), v(vv), lv(0), c(cc), is_assert(as) {}
};
struct Layer {
    vector<Gate> gates;
    int bitLength = 0;
    u64 size = 0;
    vector<vector<u64>> dadId;
    vector<int> dadBitLength;    // see subset_init for the empty-subset convention
    vector<u64> dadSize;
    u64 maxDadSize = 0;
    int maxDadBitLength = -1;
};
struct Circuit {
    vector<Layer> circuit;
    int size = 0;
    bool subset_done = false;
};

int ceil_log2(u64 x) {                  // src/main.cpp:133-136 / circuit.cpp:73-75 for x >= 1
    int b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

struct DagGate { int ty; int k0; u64 i0; int k1; u64 i1; bool is_assert; };   // inputCircuit.hpp:17-23

// src/main.cpp:15-137 (repeat == 1, so :114-131 is a no-op).
void dag_to_layered(const vector<DagGate> &dag, Circuit &c) {
    u64 n = dag.size();
    vector<u64> in_deg(n);
    vector<int> lyr(n);
    vector<u64> idl(n);
    vector<vector<u64>> edges(n);
    std::queue<u64> q;
    for (u64 i = 0; i < n; ++i) {
        const DagGate &g = dag[i];
        if (g.k0 == 'V') { ++in_deg[i]; edges[g.i0].push_back(i); }
        if (g.k1 == 'V') { ++in_deg[i]; edges[g.i1].push_back(i); }
        if (g.ty == Input) { lyr[i] = 0; q.push(i); }
    }
    int max_lyr = 0;
    while (!q.empty()) {
        u64 u = q.front(); q.pop();
        max_lyr = std::max(lyr[u], max_lyr);
        for (u64 v : edges[u])
            if (!(--in_deg[v])) { q.push(v); lyr[v] = std::max(lyr[v], lyr[u] + 1); }
    }
    c.circuit.assign(max_lyr + 1, Layer());
    c.size = max_lyr + 1;
    for (u64 i = 0; i < n; ++i) idl[i] = c.circuit[lyr[i]].size++;
    for (int i = 0; i < c.size; ++i) c.circuit[i].gates.resize(c.circuit[i].size);
    for (u64 i = 0; i < n; ++i) {
        int lg = lyr[i];
        u64 gid = idl[i];
        const DagGate &g = dag[i];
        int ty = g.ty, nty = ty;
        u64 in0 = g.i0, in1 = g.i1, u, v;
        switch (ty) {
            case Mul: case Add: case Xor:
                u = idl[in0]; v = idl[in1];
                if (lyr[in0] < lg - 1) { std::swap(u, v); std::swap(in0, in1); }
                c.circuit[lg].gates[gid] = Gate(ty, lyr[in1], u, v, F_ZERO, g.is_assert);
                break;
            case Sub: case Naab:
                u = idl[in0]; v = idl[in1];
                if (lyr[in0] < lg - 1) { nty = (ty == Sub ? AntiSub : AntiNaab); std::swap(u, v); std::swap(in0, in1); }
                c.circuit[lg].gates[gid] = Gate(nty, lyr[in1], u, v, F_ZERO, g.is_assert);
                break;
            case Mulc: case Addc:
                u = idl[in0];
                c.circuit[lg].gates[gid] = Gate(ty, -1, u, 0, F((long long) in1), g.is_assert);
                break;
            case Not: case Copy:      // main.cpp:104-110: falls through into `case Input`, so u ends up
            case Input:               // as the RAW DAG id of the operand and c as zero.
                u = in0;
                c.circuit[lg].gates[gid] = Gate(ty, -1, u, 0, F_ZERO, g.is_assert);
                break;
        }
    }
    for (int i = 0; i <= max_lyr; ++i) c.circuit[i].bitLength = ceil_log2(c.circuit[i].size);
}

// src/circuit.cpp:43-80.  For an EMPTY subset the reference evaluates (int) log2(0), which is
// undefined; compiled with g++ on x86-64 it yields INT_MIN, and every later use behaves as
// "present, bit length 0, size 0" (~INT_MIN != 0; 1ULL << INT_MIN == 1 because x86 masks the
// count; (u8) INT_MIN == 0).  We store bit length 0 with dadSize 0 and reproduce exactly that.
void subset_init(Circuit &C) {
    if (C.subset_done) return;
    int size = C.size;
    for (int i = 0; i < size; ++i) {
        C.circuit[i].dadBitLength.assign(i, -1);
        C.circuit[i].dadSize.assign(i, 0);
        C.circuit[i].dadId.assign(i, vector<u64>());
        C.circuit[i].maxDadBitLength = -1;
        C.circuit[i].maxDadSize = 0;
    }
    vector<vector<int>> visited(size);
    vector<vector<u64>> subset(size);
    for (int i = 0; i < size; ++i) { visited[i].assign(C.circuit[i].size, 0); subset[i].assign(C.circuit[i].size, 0); }
    for (int i = size - 1; i > 0; --i) {
        Layer &L = C.circuit[i];
        for (u64 j = L.size - 1; j < L.size; --j) {
            Gate &g = L.gates[j];
            int l = g.l;
            u64 v = g.v;
            if (l == -1) continue;
            if (visited[l][v] != i) {
                visited[l][v] = i;
                subset[l][v] = L.dadSize[l]++;
                L.dadId[l].push_back(v);
            }
            g.lv = subset[l][v];
        }
        for (int j = 0; j < i; ++j) {
            if (L.dadSize[j] == 0) { L.dadBitLength[j] = 0; continue; }   // INT_MIN in the reference: no effect on the max
            L.dadBitLength[j] = ceil_log2(L.dadSize[j]);
            L.maxDadSize = std::max(L.dadSize[j], L.maxDadSize);
            L.maxDadBitLength = std::max(L.dadBitLength[j], L.maxDadBitLength);
        }
    }
    C.subset_done = true;
}

// ----------------------------------------------------------------------------------------------------
// Timer with the reference's accumulate semantics (lib/virgo/src/timer.cpp:7-20).
// ----------------------------------------------------------------------------------------------------
struct Timer {
    double total = 0;
    std::chrono::high_resolution_clock::time_point t0;
    void start() { t0 = std::chrono::high_resolution_clock::now(); }
    void stop() { total += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); }
};

// ----------------------------------------------------------------------------------------------------
// Prover.  src/prover.cpp.
// ----------------------------------------------------------------------------------------------------
struct Prover {
    const Circuit &C;
    vector<vector<F>> circuitValue;
    vector<F> r_u, r_liu;
    vector<vector<F>> r_v;
    vector<F> beta_g, beta_u;
    vector<u64> total, totalSize;
    int round = 0, sumcheckLayerId = 0;
    vector<vector<Lin>> multArray, addVArray, Vmult;
    F add_term, V_u;
    Timer prove_timer;
    u64 proof_size = 0, pairs = 0, rounds = 0;
    double evaluate_sec = 0;

    explicit Prover(const Circuit &c) : C(c) {
        Timer t; t.start(); evaluate(); t.stop(); evaluate_sec = t.total;
    }

    void evaluate() {                                                // prover.cpp:27-91
        circuitValue.resize(C.size + 1);
        circuitValue[0].assign(1ull << C.circuit[0].bitLength, F_ZERO);
        for (u64 g = 0; g < C.circuit[0].size; ++g) circuitValue[0][g] = F((long long) C.circuit[0].gates[g].u);
        for (int i = 1; i < C.size; ++i) {
            circuitValue[i].resize(C.circuit[i].size);
            for (u64 g = 0; g < C.circuit[i].size; ++g) {
                const Gate &info = C.circuit[i].gates[g];
                int l = info.l;
                u64 u = info.u, v = info.v;
                const vector<F> &pre = circuitValue[i - 1];
                F &out = circuitValue[i][g];
                switch (info.ty) {
                    case Add: out = pre[u] + circuitValue[l][v]; break;
                    case Sub: out = pre[u] - circuitValue[l][v]; break;
                    case AntiSub: out = -pre[u] + circuitValue[l][v]; break;
                    case Mul: out = pre[u] * circuitValue[l][v]; break;
                    case Naab: out = circuitValue[l][v] - pre[u] * circuitValue[l][v]; break;
                    case AntiNaab: out = pre[u] - pre[u] * circuitValue[l][v]; break;
                    case Addc: out = pre[u] + info.c; break;
                    case Mulc: out = pre[u] * info.c; break;
                    case Copy: out = pre[u]; break;
                    case Not: out = F_ONE - pre[u]; break;
                    case Xor: out = pre[u] + circuitValue[l][v] - F(2ll) * pre[u] * circuitValue[l][v]; break;
                    default: fprintf(stderr, "oracle: bad gate type\n"); abort();
                }
            }
        }
    }

    F Vres(const F *r_0, int r_0_size) {                             // prover.cpp:99-129
        g_cnt.on = true;
        prove_timer.start();
        vector<F> output = circuitValue[C.size - 1];
        u64 output_size = output.size();
        u64 whole = 1ull << r_0_size;
        output.resize(whole);
        for (int i = 0; i < r_0_size; ++i) {
            for (u64 j = 0; j < (whole >> 1); ++j) {
                if (j > 0) output[j] = F_ZERO;
                if ((j << 1) < output_size) output[j] = output[j << 1] * (F_ONE - r_0[i]);
                if ((j << 1 | 1) < output_size) output[j] = output[j] + output[j << 1 | 1] * r_0[i];
            }
            whole >>= 1;
        }
        F res = output[0];
        prove_timer.stop();
        g_cnt.on = false;
        return res;
    }

    void init() {                                                    // prover.cpp:131-155
        r_v.resize(C.size);
        total.resize(C.size - 1);
        totalSize.resize(C.size - 1);
        int max_bl = 0;
        for (auto &c : C.circuit) max_bl = std::max(max_bl, c.bitLength);
        beta_g.resize(1ull << max_bl);
        beta_u.resize(1ull << max_bl);
        r_u.resize(max_bl);
        r_liu.resize(max_bl);
        for (int i = 1; i < C.size; ++i)
            if (C.circuit[i].maxDadBitLength != -1) r_v[i].resize(C.circuit[i].maxDadBitLength);
        Vmult.resize(C.size + 1);
        addVArray.resize(C.size + 1);
        multArray.resize(C.size + 1);
    }

    void sumcheckInitAll(const F *r_last) {                          // prover.cpp:162-170
        prove_timer.start();
        int last_bl = C.circuit[C.size - 1].bitLength;
        sumcheckLayerId = C.size;
        for (int i = 0; i < last_bl; ++i) r_liu[i] = r_last[i];
        prove_timer.stop();
    }
    void sumcheckInit() {                                            // prover.cpp:177-184
        prove_timer.start();
        --sumcheckLayerId;
        Vmult.pop_back(); addVArray.pop_back(); multArray.pop_back();
        prove_timer.stop();
    }

    static void grow(vector<Lin> &v, u64 n) { if (v.size() < n) v.resize(n); }

    void sumcheckInitPhase1(const F &assert_random) {                // prover.cpp:189-280
        g_cnt.on = true;
        prove_timer.start();
        const Layer &cur = C.circuit[sumcheckLayerId], &pre = C.circuit[sumcheckLayerId - 1];
        total[0] = 1ull << pre.bitLength;
        totalSize[0] = pre.size;
        grow(Vmult[0], total[0]); grow(addVArray[0], total[0]); grow(multArray[0], total[0]);
        vector<Lin> &tmp_mult = multArray[0], &tmp_add = addVArray[0], &tmp_v = Vmult[0];
        init_beta_table(beta_g, cur.bitLength, r_liu.data(), F_ONE);
        for (u64 g = 0; g < cur.size; ++g)
            if (cur.gates[g].is_assert) beta_g[g] *= assert_random;
        for (u64 g = 0; g < total[0]; ++g) {
            tmp_mult[g] = Lin(F_ZERO, F_ZERO);
            tmp_add[g] = Lin(F_ZERO, F_ZERO);
            tmp_v[g] = (g < totalSize[0]) ? Lin(circuitValue[sumcheckLayerId - 1][g]) : Lin(F_ZERO);
        }
        for (u64 g = 0; g < cur.size; ++g) {
            const Gate &info = cur.gates[g];
            int l = info.l;
            u64 u = info.u, v = info.v;
            const F &tmp = beta_g[g];
            switch (info.ty) {                                       // Appendix A of SURVEY.md, prover.cpp:229-272
                case Add:
                    tmp_add[u] = tmp_add[u] + Lin(circuitValue[l][v] * tmp);
                    tmp_mult[u] = tmp_mult[u] + Lin(tmp);
                    break;
                case Sub:
                    tmp_add[u] = Lin(tmp_add[u].b - circuitValue[l][v] * tmp);
                    tmp_mult[u] = tmp_mult[u] + Lin(tmp);
                    break;
                case AntiSub:
                    tmp_add[u] = tmp_add[u] + Lin(circuitValue[l][v] * tmp);
                    tmp_mult[u] = Lin(tmp_mult[u].b - tmp);
                    break;
                case Mul:
                    tmp_mult[u] = tmp_mult[u] + Lin(circuitValue[l][v] * tmp);
                    break;
                case Naab:
                    tmp_add[u] = tmp_add[u] + Lin(tmp * circuitValue[l][v]);
                    tmp_mult[u].b = tmp_mult[u].b - (circuitValue[l][v] * tmp);
                    break;
                case AntiNaab:
                    tmp_mult[u] = tmp_mult[u] + Lin(tmp - (circuitValue[l][v] * tmp));
                    break;
                case Addc:
                    tmp_add[u] = tmp_add[u] + Lin(info.c * tmp);
                    tmp_mult[u] = tmp_mult[u] + Lin(tmp);
                    break;
                case Mulc:
                    tmp_mult[u] = tmp_mult[u] + Lin(info.c * tmp);
                    break;
                case Copy:
                    tmp_mult[u].b += tmp;
                    break;
                case Not:
                    tmp_add[u] = tmp_add[u] + Lin(tmp);
                    tmp_mult[u].b = tmp_mult[u].b - tmp;
                    break;
                case Xor:
                    tmp_add[u] = tmp_add[u] + Lin(tmp * circuitValue[l][v]);
                    tmp_mult[u] = tmp_mult[u] + Lin(tmp * (F_ONE - (circuitValue[l][v] + circuitValue[l][v])));
                    break;
                default: fprintf(stderr, "oracle: bad gate type\n"); abort();
            }
        }
        round = 0;
        prove_timer.stop();
        g_cnt.on = false;
    }

    void sumcheckInitPhase2() {                                      // prover.cpp:282-367
        g_cnt.on = true;
        prove_timer.start();
        const Layer &cur = C.circuit[sumcheckLayerId], &pre = C.circuit[sumcheckLayerId - 1];
        for (int i = 0; i < sumcheckLayerId; ++i) {
            total[i] = 1ull << cur.dadBitLength[i];                  // empty subset: 1 (see subset_init)
            totalSize[i] = cur.dadSize[i];
            grow(Vmult[i], total[i]); grow(addVArray[i], total[i]); grow(multArray[i], total[i]);
        }
        add_term = F_ZERO;
        for (int i = 0; i < sumcheckLayerId; ++i)
            for (u64 v = 0; v < total[i]; ++v) {
                Vmult[i][v] = v < totalSize[i] ? Lin(circuitValue[i][cur.dadId[i][v]]) : Lin(F_ZERO);
                addVArray[i][v] = Lin(F_ZERO);
                multArray[i][v] = Lin(F_ZERO);
            }
        init_beta_table(beta_u, pre.bitLength, r_u.data(), F_ONE);
        for (u64 g = 0; g < cur.size; ++g) {
            const Gate &info = cur.gates[g];
            int l = info.l == -1 ? sumcheckLayerId - 1 : info.l;
            u64 u = info.u, v = info.lv;
            F tmp = beta_g[g] * beta_u[u];
            vector<Lin> &M = multArray[l], &A = addVArray[l];
            switch (info.ty) {                                       // prover.cpp:319-360
                case Add:
                    M[v] = M[v] + Lin(tmp);
                    A[v] = A[v] + Lin(tmp * V_u);
                    break;
                case Sub:
                    A[v] = A[v] + Lin(tmp * V_u);
                    M[v] = Lin(M[v].b - tmp);
                    break;
                case AntiSub:
                    A[v].b = A[v].b - tmp * V_u;
                    M[v].b = M[v].b + tmp;
                    break;
                case Mul:
                    M[v] = M[v] + Lin(tmp * V_u);
                    break;
                case Naab:
                    M[v] = M[v] + Lin(tmp - V_u * tmp);
                    break;
                case AntiNaab:
                    A[v] = A[v] + Lin(tmp * V_u);
                    M[v].b = M[v].b - V_u * tmp;
                    break;
                case Addc:
                    A[0] = A[0] + Lin(tmp * (info.c + V_u));
                    break;
                case Mulc:
                    A[0] = A[0] + Lin(tmp * info.c * V_u);
                    break;
                case Copy:
                    A[0] = A[0] + Lin(tmp * V_u);
                    break;
                case Not:
                    A[0] = A[0] + Lin(tmp * (F_ONE - V_u));
                    break;
                case Xor:
                    A[v] = A[v] + Lin(tmp * V_u);
                    M[v] = M[v] + Lin(tmp * (F_ONE - (V_u + V_u)));
                    break;
                default: fprintf(stderr, "oracle: bad gate type\n"); abort();
            }
        }
        round = 0;
        prove_timer.stop();
        g_cnt.on = false;
    }

    void sumcheckInitLiu(const F *s) {                               // prover.cpp:369-420
        g_cnt.on = true;
        prove_timer.start();
        int pre_layer_id = sumcheckLayerId - 1;
        const Layer &pre = C.circuit[pre_layer_id];
        total[0] = 1ull << pre.bitLength;
        totalSize[0] = pre.size;
        grow(Vmult[0], total[0]); grow(addVArray[0], total[0]); grow(multArray[0], total[0]);
        add_term = F_ZERO;
        for (u64 u = 0; u < total[0]; ++u) {
            addVArray[0][u] = Lin(F_ZERO);
            multArray[0][u] = Lin(F_ZERO);
            Vmult[0][u] = (u < totalSize[0]) ? Lin(circuitValue[pre_layer_id][u]) : Lin(F_ZERO);
        }
        init_beta_table(beta_g, pre.bitLength, r_u.data(), s[0]);
        for (u64 u = 0; u < totalSize[0]; ++u) multArray[0][u] = multArray[0][u] + Lin(beta_g[u]);
        for (int i = sumcheckLayerId; i < C.size; ++i) {
            int bl = C.circuit[i].dadBitLength[pre_layer_id];
            u64 size_i = C.circuit[i].dadSize[pre_layer_id];
            // `if (~bit_length_i)` is true for every i, empty subsets included (see subset_init)
            init_beta_table(beta_g, bl, r_v[i].data(), s[i - sumcheckLayerId + 1]);
            for (u64 g = 0; g < size_i; ++g) {
                u64 u = C.circuit[i].dadId[pre_layer_id][g];
                multArray[0][u] = multArray[0][u] + Lin(beta_g[g]);
            }
        }
        round = 0;
        prove_timer.stop();
        g_cnt.on = false;
    }

    Quad sumcheckUpdateEach(const F &prev, int idx) {                // prover.cpp:457-492
        vector<Lin> &tmp_v = Vmult[idx], &tmp_add = addVArray[idx], &tmp_mult = multArray[idx];
        if (total[idx] == 1) {
            tmp_v[0] = Lin(tmp_v[0].eval(prev));
            tmp_add[0] = Lin(tmp_add[0].eval(prev));
            tmp_mult[0] = Lin(tmp_mult[0].eval(prev));
            add_term = add_term + tmp_v[0].b * tmp_mult[0].b + tmp_add[0].b;
        }
        Quad ret(F_ZERO, F_ZERO, F_ZERO);
        for (u64 i = 0; i < (total[idx] >> 1); ++i) {
            u64 g0 = i << 1, g1 = i << 1 | 1;
            if (g0 >= totalSize[idx]) {
                tmp_v[i] = Lin(F_ZERO); tmp_add[i] = Lin(F_ZERO); tmp_mult[i] = Lin(F_ZERO);
                continue;
            }
            if (g1 >= totalSize[idx]) { tmp_v[g1] = Lin(F_ZERO); tmp_add[g1] = Lin(F_ZERO); tmp_mult[g1] = Lin(F_ZERO); }
            tmp_v[i] = interpolate(tmp_v[g0].eval(prev), tmp_v[g1].eval(prev));
            tmp_add[i] = interpolate(tmp_add[g0].eval(prev), tmp_add[g1].eval(prev));
            tmp_mult[i] = interpolate(tmp_mult[g0].eval(prev), tmp_mult[g1].eval(prev));
            ret = ret + tmp_mult[i] * tmp_v[i] + Quad(F_ZERO, tmp_add[i].a, tmp_add[i].b);
            ++pairs;
        }
        total[idx] >>= 1;
        totalSize[idx] = (totalSize[idx] + 1) >> 1;
        return ret;
    }

    Quad sumcheckUpdate(const F &prev, vector<F> &r_arr, int n_pre_layer) {   // prover.cpp:436-455
        g_cnt.on = true;
        prove_timer.start();
        if (round) r_arr.at(round - 1) = prev;
        ++round;
        ++rounds;
        Quad ret(F_ZERO, F_ZERO, F_ZERO);
        add_term = add_term == F_ZERO ? F_ZERO : add_term * (F_ONE - prev);
        for (int i = 0; i < n_pre_layer; ++i) ret = ret + sumcheckUpdateEach(prev, i);
        ret = ret + Quad(F_ZERO, -add_term, add_term);
        prove_timer.stop();
        proof_size += 16 * 3;
        g_cnt.on = false;
        return ret;
    }
    Quad sumcheckUpdatePhase1(const F &prev) { return sumcheckUpdate(prev, r_u, 1); }
    Quad sumcheckUpdatePhase2(const F &prev) { return sumcheckUpdate(prev, r_v[sumcheckLayerId], sumcheckLayerId); }
    Quad sumcheckLiuUpdate(const F &prev) { return sumcheckUpdate(prev, r_liu, 1); }

    void sumcheckFinalize1(const F &prev, F &claim) {                // prover.cpp:494-501
        prove_timer.start();
        if (round) r_u[round - 1] = prev;   // the reference writes r_u[-1] here when layer i-1 has one gate (UB); skipped
        V_u = claim = total[0] ? Vmult[0][0].eval(prev) : Vmult[0][0].b;
        prove_timer.stop();
        proof_size += 16;
    }
    void sumcheckFinalize2(const F &prev, F *claims) {               // prover.cpp:504-516
        prove_timer.start();
        if (round) r_v[sumcheckLayerId][round - 1] = prev;
        for (int i = 0; i < sumcheckLayerId; ++i) {
            claims[i] = total[i] ? Vmult[i][0].eval(prev) : Vmult[i][0].b;
            proof_size += 16;                                        // ~dadBitLength is never 0 (see subset_init)
        }
        prove_timer.stop();
    }
    void sumcheckLiuFinalize(const F &prev, F &claim) {              // prover.cpp:518-521
        if (round) r_liu[round - 1] = prev;
        claim = total[0] ? Vmult[0][0].eval(prev) : Vmult[0][0].b;
    }
};

// ----------------------------------------------------------------------------------------------------
// Verifier (GKR part).  src/verifier.cpp:12-337.  Randomness schedule and checks are literal.
// ----------------------------------------------------------------------------------------------------
// SHA3 pieces defined with the commitment further down (same unnamed namespace)
struct Digest { u64 w[4]; };
Digest hhash(const u64 in[8]);
void keccak_f1600(u64 A[25]);

// Streaming SHA3-256 (FIPS 202: Keccak[512], rate 136 bytes, domain bits 01, pad10*1) — the oracle's own sponge around its own
// permutation, for the statement digest of the Fiat-Shamir mode.
struct Sha3Stream {
    u64 A[25] = {0};
    unsigned char buf[136];
    size_t fill = 0;
    void block() {
        for (int i = 0; i < 17; ++i) { u64 w = 0; for (int b = 0; b < 8; ++b) w |= (u64) buf[8 * i + b] << (8 * b); A[i] ^= w; }
        keccak_f1600(A);
        fill = 0;
    }
    void bytes(const unsigned char *p, size_t n) { for (size_t i = 0; i < n; ++i) { buf[fill++] = p[i]; if (fill == 136) block(); } }
    void word(u64 x) { unsigned char b[8]; for (int i = 0; i < 8; ++i) b[i] = (unsigned char) (x >> (8 * i)); bytes(b, 8); }
    Digest done() {
        for (size_t i = fill; i < 136; ++i) buf[i] = 0;
        buf[fill] ^= 0x06; buf[135] ^= 0x80;
        block();
        Digest d; for (int i = 0; i < 4; ++i) d.w[i] = A[i];
        return d;
    }
};

struct Verifier {
    Prover *p;
    const Circuit &C;
    vector<F> beta_g, beta_u, beta_v, r_u, r_liu, sig;
    vector<vector<F>> r_v;
    F coeff_l[12];
    vector<F> coeff_r[12];
    F bias, final_claim_u;
    vector<vector<F>> final_claims_v;
    vector<unsigned char> *out;
    Timer vt;

    Verifier(Prover *pr, const Circuit &c, vector<unsigned char> *o) : p(pr), C(c), out(o) {   // verifier.cpp:12-48
        final_claims_v.resize(C.size);
        for (int i = 1; i < C.size; ++i) final_claims_v[i].resize(i);
        for (auto &v : coeff_r) v.resize(C.size);
        r_v.resize(C.size + 2);
        p->init();
        int max_bl = 0, max_dad_bl = 0;
        for (auto &l : C.circuit) max_bl = std::max(max_bl, l.bitLength);
        for (auto &l : C.circuit) max_dad_bl = std::max(max_dad_bl, l.maxDadBitLength);
        beta_g.resize(1ull << std::max(max_bl, max_dad_bl));
        beta_u.resize(1ull << max_bl);
        beta_v.resize(1ull << max_bl);
        r_u.resize(max_bl);
        r_liu.resize(max_bl);
        for (int i = 1; i < C.size; ++i)
            if (C.circuit[i].maxDadBitLength != -1) r_v[i].resize(C.circuit[i].maxDadBitLength);
        sig.resize(C.size);
    }

    // ---- Fiat-Shamir mode (NOT in the reference, whose verifier draws glibc random()).  The definition restated
    // here is the one virgo-plus_amd/host/verifier.cpp documents for verifier::proveFS(): a SHA3-256 chain state' = H(block || state)
    // over 64-byte blocks; the statement (serialised circuit + subset tables + input values, domain tag "virg") is absorbed first as
    // H(SHA3-256(serialisation) || 0) and then {n_layers, 0, 0, 'S'}; every prover message x is absorbed as {x.re, x.im, 0, 'M'}; the
    // c-th challenge is squeezed by absorbing {c, 0, 0, 'C'} and taking the first two state words mod 2^61 (p -> 0).  Challenges that
    // belong to a sumcheck round are drawn AFTER that round's polynomial; the others where the interactive verifier draws them.
    bool fs = false;
    Digest fs_state{};
    u64 fs_ctr = 0;
    void fs_absorb(u64 a, u64 b, u64 c, u64 tag) { const u64 in[8] = {a, b, c, tag, fs_state.w[0], fs_state.w[1], fs_state.w[2], fs_state.w[3]}; fs_state = hhash(in); }
    void fs_init() {
        Sha3Stream h;
        h.word(0x76697267ull);
        h.word((u64) C.size);
        for (int i = 0; i < C.size; ++i) {
            const Layer &L = C.circuit[i];
            h.word(L.size); h.word((u64) (long long) L.bitLength);
            for (u64 g = 0; g < L.size; ++g) {
                const Gate &G = L.gates[g];
                h.word((u64) (long long) G.ty | ((u64) (G.is_assert ? 1 : 0) << 32)); h.word((u64) (long long) G.l);
                h.word(G.u); h.word(G.v); h.word(G.lv); h.word(G.c.re); h.word(G.c.im);
            }
            h.word((u64) (long long) L.maxDadBitLength); h.word(L.maxDadSize);
            for (int j = 0; j < i; ++j) {
                h.word(L.dadSize[j]);
                h.word(L.dadSize[j] ? (u64) (long long) L.dadBitLength[j] : ~0ull);
                for (u64 k = 0; k < L.dadSize[j]; ++k) h.word(L.dadId[j][k]);
            }
        }
        const Digest st = h.done();
        fs_state = Digest{}; fs_ctr = 0;
        fs_absorb(st.w[0], st.w[1], st.w[2], st.w[3]);
        fs_absorb((u64) C.size, 0, 0, 0x53);
    }
    F draw() {
        if (!fs) return frandom();
        fs_absorb(fs_ctr++, 0, 0, 0x43);
        u64 a = fs_state.w[0] & P, b = fs_state.w[1] & P;
        if (a == P) a = 0;
        if (b == P) b = 0;
        return F(a, b);
    }
    void putF(const F &x) {
        u64 w[2] = {x.re, x.im}; const unsigned char *b = (const unsigned char *) w; out->insert(out->end(), b, b + 16);
        if (fs) fs_absorb(x.re, x.im, 0, 0x4d);
    }
    void putQ(const Quad &q) { putF(q.a); putF(q.b); putF(q.c); }

    F assert_random;
    void predicatePhase1(int layer_id) {                             // verifier.cpp:50-90
        const Layer &cur = C.circuit[layer_id];
        init_beta_table(beta_g, cur.bitLength, r_liu.data(), F_ONE);
        for (u64 g = 0; g < cur.size; ++g) if (cur.gates[g].is_assert) beta_g[g] *= assert_random;     // verifier.cpp:53-54
        init_beta_table(beta_u, C.circuit[layer_id - 1].bitLength, r_u.data(), F_ONE);
        coeff_l[Copy] = coeff_l[Not] = coeff_l[Addc] = coeff_l[Mulc] = F_ZERO;
        bias = F_ZERO;
        for (u64 g = 0; g < cur.size; ++g) {
            const Gate &gt = cur.gates[g];
            switch (gt.ty) {
                case Addc: bias += beta_g[g] * beta_u[gt.u] * gt.c;   // falls through
                case Not: case Copy: coeff_l[gt.ty] += beta_g[g] * beta_u[gt.u]; break;
                case Mulc: coeff_l[gt.ty] += beta_g[g] * beta_u[gt.u] * gt.c; break;
                default: break;
            }
        }
        for (int t : {Add, Sub, AntiSub, Mul, Naab, AntiNaab, Xor}) std::fill(coeff_r[t].begin(), coeff_r[t].end(), F_ZERO);
    }
    void predicatePhase2(int layer_id) {                             // verifier.cpp:58-61,92-113
        const Layer &cur = C.circuit[layer_id];
        init_beta_table(beta_v, cur.maxDadBitLength, r_v[layer_id].data(), F_ONE);
        coeff_l[Copy] *= beta_v[0]; coeff_l[Not] *= beta_v[0]; coeff_l[Addc] *= beta_v[0]; coeff_l[Mulc] *= beta_v[0];
        bias *= beta_v[0];
        for (u64 g = 0; g < cur.size; ++g) {
            const Gate &gt = cur.gates[g];
            switch (gt.ty) {
                case Add: case Sub: case AntiSub: case Mul: case Naab: case AntiNaab: case Xor:
                    coeff_r[gt.ty][gt.l] += beta_g[g] * beta_u[gt.u] * beta_v[gt.lv];
                default: break;
            }
        }
    }
    F getFinalValue(int layer_id, const F &cu, const F *cv) {        // verifier.cpp:115-132
        F res = coeff_l[Not] * (F_ONE - cu) + coeff_l[Copy] * cu + coeff_l[Addc] * cu + bias + coeff_l[Mulc] * cu;
        for (int j = 0; j < layer_id; ++j) {
            F tmp = coeff_r[Add][j] * (cu + cv[j]) + coeff_r[Sub][j] * (cu - cv[j]) + coeff_r[AntiSub][j] * (cv[j] - cu)
                    + coeff_r[Mul][j] * (cu * cv[j]) + coeff_r[Naab][j] * (cv[j] - cu * cv[j])
                    + coeff_r[AntiNaab][j] * (cu - cu * cv[j]) + coeff_r[Xor][j] * (cu + cv[j] - F(2ll) * cu * cv[j]);
            res = res + tmp;
        }
        return res;
    }

    bool verifyPhase1(int layer_id, F &previousSum) {                // verifier.cpp:191-229
        const Layer &pre = C.circuit[layer_id - 1];
        if (!fs) for (auto &x : r_u) x = draw();
        F previousRandom = F_ZERO;
        assert_random = draw();
        p->sumcheckInitPhase1(assert_random);
        for (int j = 0; j < pre.bitLength; ++j) {
            Quad poly = p->sumcheckUpdatePhase1(previousRandom);
            putQ(poly);
            vt.start();
            if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
                fprintf(stderr, "oracle: verification fail, phase1, circuit %d, bit %d\n", layer_id, j);
                return false;
            }
            if (fs) { vt.stop(); r_u[j] = draw(); vt.start(); }
            previousRandom = r_u[j];
            previousSum = poly.eval(r_u[j]);
            vt.stop();
        }
        p->sumcheckFinalize1(previousRandom, final_claim_u);
        putF(final_claim_u);
        vt.start();
        predicatePhase1(layer_id);
        vt.stop();
        return true;
    }
    bool verifyPhase2(int layer_id, F &previousSum) {                // verifier.cpp:231-270
        if (!fs) for (auto &x : r_v[layer_id]) x = draw();
        F previousRandom = F_ZERO;
        p->sumcheckInitPhase2();
        for (int j = 0; j < C.circuit[layer_id].maxDadBitLength; ++j) {
            Quad poly = p->sumcheckUpdatePhase2(previousRandom);
            putQ(poly);
            vt.start();
            if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
                fprintf(stderr, "oracle: verification fail, phase2, circuit %d, bit %d\n", layer_id, j);
                return false;
            }
            if (fs) r_v[layer_id][j] = draw();
            previousRandom = r_v[layer_id][j];
            previousSum = poly.eval(previousRandom);
            vt.stop();
        }
        p->sumcheckFinalize2(previousRandom, final_claims_v[layer_id].data());
        for (int j = 0; j < layer_id; ++j) putF(final_claims_v[layer_id][j]);
        vt.start();
        predicatePhase2(layer_id);
        vt.stop();
        return true;
    }
    bool verifyLiu(int layer_id, F &previousSum) {                   // verifier.cpp:272-337
        int pre_layer_id = layer_id - 1;
        const Layer &pre = C.circuit[pre_layer_id];
        for (auto &x : sig) x = draw();
        if (!fs) for (auto &x : r_liu) x = draw();
        previousSum = sig[0] * final_claim_u;
        for (int j = layer_id; j < C.size; ++j)                      // `~dadBitLength` is never 0
            previousSum += sig[j - pre_layer_id] * final_claims_v[j][pre_layer_id];
        p->sumcheckInitLiu(sig.data());
        F previousRandom = F_ZERO;
        for (int j = 0; j < pre.bitLength; ++j) {
            Quad poly = p->sumcheckLiuUpdate(previousRandom);
            putQ(poly);
            vt.start();
            if (poly.eval(F_ZERO) + poly.eval(F_ONE) != previousSum) {
                fprintf(stderr, "oracle: Liu fail, circuit %d, bit %d\n", layer_id, j);
                return false;
            }
            if (fs) r_liu[j] = draw();
            previousRandom = r_liu[j];
            previousSum = poly.eval(previousRandom);
            vt.stop();
        }
        F gr = F_ZERO, vr;
        p->sumcheckLiuFinalize(previousRandom, vr);
        putF(vr);
        vt.start();
        init_beta_table(beta_u, pre.bitLength, r_liu.data(), F_ONE);
        init_beta_table(beta_g, pre.bitLength, r_u.data(), sig[0]);
        for (u64 g = 0; g < pre.size; ++g) gr = gr + beta_g[g] * beta_u[g];
        for (int j = layer_id; j < C.size; ++j) {
            init_beta_table(beta_g, C.circuit[j].dadBitLength[pre_layer_id], r_v[j].data(), sig[j - pre_layer_id]);
            for (u64 g = 0; g < C.circuit[j].dadSize[pre_layer_id]; ++g)
                gr = gr + beta_g[g] * beta_u[C.circuit[j].dadId[pre_layer_id][g]];
        }
        bool ok = (vr * gr == previousSum);
        if (!ok) fprintf(stderr, "oracle: Liu fail, semi final, circuit %d\n", layer_id);
        previousSum = vr;
        vt.stop();
        return ok;
    }

    bool verify_gkr() {                                              // verifier.cpp:134-169 without the PC calls
        for (int i = 0; i < C.circuit[C.size - 1].bitLength; ++i) r_liu[i] = draw();
        F previousSum = p->Vres(r_liu.data(), C.circuit[C.size - 1].bitLength);
        putF(previousSum);
        p->sumcheckInitAll(r_liu.data());
        for (int i = C.size - 1; i; --i) {
            p->sumcheckInit();
            if (!verifyPhase1(i, previousSum)) return false;
            if (C.circuit[i].maxDadBitLength != -1 && !verifyPhase2(i, previousSum)) return false;
            vt.start();
            F test_value = getFinalValue(i, final_claim_u, final_claims_v[i].data());
            vt.stop();
            if (previousSum != test_value) {
                fprintf(stderr, "oracle: verification fail, semi final, circuit level %d\n", i);
                return false;
            }
            if (!verifyLiu(i, previousSum)) return false;
        }
        // verifyPoly (verifier.cpp:363-389) would now check previousSum against the committed input.
        // With PC off we check it directly against the input layer's MLE at r_liu.
        {
            vector<F> beta;
            init_beta_table(beta, C.circuit[0].bitLength, r_liu.data(), F_ONE);
            F acc = F_ZERO;
            for (u64 g = 0; g < C.circuit[0].size; ++g) acc = acc + beta[g] * p->circuitValue[0][g];
            if (acc != previousSum) { fprintf(stderr, "oracle: final input check fail\n"); return false; }
        }
        return true;
    }
};

// ----------------------------------------------------------------------------------------------------
// .pws reading + replication (grammar src/main.cpp:161-207).
// ----------------------------------------------------------------------------------------------------
struct PwsGate { int ty; u64 tgt, s0, s1; };
bool read_pws(const char *path, vector<u64> &inputs, vector<PwsGate> &gates) {
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char line[256];
    bool ok = true;
    while (ok && fgets(line, sizeof line, f)) {
        long long t, a, b; char op[16];
        if (sscanf(line, "P V%lld = I%lld E", &t, &a) == 2) { inputs.push_back(t); continue; }
        if (sscanf(line, "P O%lld = V%lld E", &t, &a) == 2) continue;
        if (sscanf(line, "P V%lld = V%lld %15s V%lld E", &t, &a, op, &b) == 4) {
            int ty;
            if (!strcmp(op, "+")) ty = Add; else if (!strcmp(op, "*")) ty = Mul; else if (!strcmp(op, "XOR")) ty = Xor;
            else if (!strcmp(op, "minus")) ty = Sub; else if (!strcmp(op, "NAAB")) ty = Naab; else if (!strcmp(op, "NOT")) ty = Not;
            else { ok = false; break; }
            gates.push_back({ty, (u64) t, (u64) a, (u64) b});
            continue;
        }
        ok = false;
    }
    fclose(f);
    return ok;
}

}  // namespace

struct orc_circuit { Circuit c; std::vector<F> last_point; };

extern "C" {

orc_circuit *orc_circuit_from_pws(const char *path, int blocks, long seed) {
    vector<u64> inputs; vector<PwsGate> gates;
    if (!read_pws(path, inputs, gates)) return nullptr;
    u64 nin = inputs.size(), ng = gates.size();
    for (u64 k = 0; k < nin; ++k) if (inputs[k] != k) return nullptr;
    for (u64 g = 0; g < ng; ++g) if (gates[g].tgt != nin + g) return nullptr;
    if (seed >= 0) srandom((unsigned) seed);
    u64 B = blocks;
    vector<DagGate> dag(B * (nin + ng));
    for (u64 b = 0; b < B; ++b)
        for (u64 k = 0; k < nin; ++k)                                // buildInput, main.cpp:221-231; value main.cpp:188
            dag[b * nin + k] = {Input, 'S', (u64) (::random() % P), 'N', 0, false};
    auto map = [&](u64 b, u64 id) { return id < nin ? b * nin + id : B * nin + b * ng + (id - nin); };
    for (u64 b = 0; b < B; ++b)
        for (u64 g = 0; g < ng; ++g) {
            const PwsGate &x = gates[g];
            if (x.ty == Not) dag[map(b, x.tgt)] = {Not, 'V', map(b, x.s0), 'S', 0, false};     // main.cpp:202
            else dag[map(b, x.tgt)] = {x.ty, 'V', map(b, x.s0), 'V', map(b, x.s1), false};
        }
    orc_circuit *oc = new orc_circuit();
    dag_to_layered(dag, oc->c);
    return oc;
}

orc_circuit *orc_circuit_randomize(int layers, int log_size, long seed) {   // circuit.cpp:17-41
    if (seed >= 0) srandom((unsigned) seed);
    orc_circuit *oc = new orc_circuit();
    Circuit &c = oc->c;
    u64 gs = 1ull << log_size;
    c.circuit.assign(layers, Layer());
    c.size = layers;
    for (int i = 0; i < layers; ++i) {
        c.circuit[i].bitLength = log_size;
        c.circuit[i].size = gs;
        c.circuit[i].gates.resize(gs);
    }
    for (u64 j = 0; j < gs; ++j) c.circuit[0].gates[j] = Gate(Input, 0, (u64) ::random(), 0, F_ZERO, false);
    for (int i = 1; i < layers; ++i)
        for (u64 j = 0; j < gs; ++j) {
            // g++ evaluates the constructor arguments of circuit.cpp:38 right to left: v, u, l, then the type bit.
            u64 v = ::random() % gs;
            u64 u = ::random() % gs;
            int l = ::random() % i;
            int ty = (::random() & 1) == 0 ? Add : Mul;
            c.circuit[i].gates[j] = Gate(ty, l, u, v, F_ZERO, false);
        }
    return oc;
}

orc_circuit *orc_circuit_custom(int n_layers, const uint64_t *layer_sizes, const int32_t *ty, const int32_t *l, const uint64_t *u,
                                const uint64_t *v, const uint64_t *c_pairs, const uint8_t *is_assert) {
    orc_circuit *oc = new orc_circuit();
    Circuit &c = oc->c;
    c.size = n_layers;
    c.circuit.assign(n_layers, Layer());
    u64 at = 0;
    for (int i = 0; i < n_layers; ++i) {
        Layer &L = c.circuit[i];
        L.size = layer_sizes[i];
        L.bitLength = ceil_log2(L.size);
        L.gates.resize(L.size);
        for (u64 g = 0; g < L.size; ++g, ++at)
            L.gates[g] = Gate(ty[at], l[at], u[at], v[at], F(c_pairs[2 * at], c_pairs[2 * at + 1]), is_assert[at] != 0);
    }
    return oc;
}
void orc_circuit_free(orc_circuit *c) { delete c; }
int orc_circuit_layers(const orc_circuit *c) { return c->c.size; }
uint64_t orc_circuit_layer_size(const orc_circuit *c, int layer) { return c->c.circuit[layer].size; }
int orc_circuit_layer_bitlen(const orc_circuit *c, int layer) { return c->c.circuit[layer].bitLength; }
uint64_t orc_circuit_gates(const orc_circuit *c) { u64 n = 0; for (auto &l : c->c.circuit) n += l.size; return n; }

void orc_circuit_hash(orc_circuit *oc, uint64_t out[2]) {
    subset_init(oc->c);
    const Circuit &C = oc->c;
    u64 a = 1469598103934665603ull, b = 0x9e3779b97f4a7c15ull;
    auto put = [&](u64 x) {
        for (int i = 0; i < 8; ++i) { a ^= (x >> (8 * i)) & 0xff; a *= 1099511628211ull; }
        b = (b ^ x) * 0xff51afd7ed558ccdull; b ^= b >> 32;
    };
    put(C.size);
    for (int i = 0; i < C.size; ++i) {
        const Layer &L = C.circuit[i];
        put(L.size); put((long long) L.bitLength);
        for (u64 g = 0; g < L.size; ++g) {
            const Gate &G = L.gates[g];
            put((long long) G.ty); put((long long) G.l); put(G.u); put(G.v); put(G.lv);
            put(G.c.re); put(G.c.im); put(G.is_assert ? 1 : 0);
        }
        put((long long) L.maxDadBitLength); put(L.maxDadSize);
        for (int j = 0; j < i; ++j) {
            put(L.dadSize[j]);
            put(L.dadSize[j] ? (long long) L.dadBitLength[j] : -1ll);
            for (u64 k = 0; k < L.dadSize[j]; ++k) put(L.dadId[j][k]);
        }
    }
    out[0] = a; out[1] = b;
}

void orc_circuit_export_layer(orc_circuit *oc, int layer, int32_t *ty, int32_t *l, uint64_t *u, uint64_t *v, uint64_t *lv) {
    subset_init(oc->c);
    const Layer &L = oc->c.circuit[layer];
    for (u64 g = 0; g < L.size; ++g) {
        ty[g] = L.gates[g].ty; l[g] = L.gates[g].l; u[g] = L.gates[g].u; v[g] = L.gates[g].v; lv[g] = L.gates[g].lv;
    }
}

void orc_circuit_inputs(const orc_circuit *oc, orc_F *out) {
    const Layer &L = oc->c.circuit[0];
    for (u64 g = 0; g < L.size; ++g) { out[g].real = F((long long) L.gates[g].u).re; out[g].img = 0; }
}

int64_t orc_prove_gkr(orc_circuit *oc, uint8_t *transcript, int64_t capacity, orc_stats *st) {
    srand(3396);                                                     // F::init(), fieldElement.cpp:106-111
    g_cnt = Counter();
    subset_init(oc->c);
    Prover p(oc->c);
    vector<unsigned char> out;
    Verifier v(&p, oc->c, &out);
    bool ok = v.verify_gkr();
    if (st) {
        st->prove_sec = p.prove_timer.total;
        st->evaluate_sec = p.evaluate_sec;
        st->verify_sec = v.vt.total;
        st->mult_count = g_cnt.mul;
        st->add_count = g_cnt.add;
        st->rounds = p.rounds;
        st->pairs = p.pairs;
        st->proof_kb = (double) p.proof_size / 1024.0;
        st->verified = ok ? 1 : 0;
    }
    if ((int64_t) out.size() > capacity) return -1;
    memcpy(transcript, out.data(), out.size());
    return (int64_t) out.size();
}

// Fiat-Shamir mode: same prover, challenges from the SHA3 chain over the statement and the messages (Verifier::fs_*).  The proof is the
// message stream in the GKR transcript layout.
int64_t orc_prove_fs(orc_circuit *oc, uint8_t *proof, int64_t capacity, orc_stats *st) {
    g_cnt = Counter();
    subset_init(oc->c);
    Prover p(oc->c);
    vector<unsigned char> out;
    Verifier v(&p, oc->c, &out);
    v.fs = true;
    v.fs_init();
    bool ok = v.verify_gkr();
    if (st) {
        st->prove_sec = p.prove_timer.total; st->evaluate_sec = p.evaluate_sec; st->verify_sec = v.vt.total;
        st->mult_count = g_cnt.mul; st->add_count = g_cnt.add; st->rounds = p.rounds; st->pairs = p.pairs;
        st->proof_kb = (double) p.proof_size / 1024.0; st->verified = ok ? 1 : 0;
    }
    if ((int64_t) out.size() > capacity) return -1;
    memcpy(proof, out.data(), out.size());
    return (int64_t) out.size();
}

static inline F toF(const orc_F *x) { return F(x->real, x->img); }
static inline void fromF(const F &x, orc_F *o) { o->real = x.re; o->img = x.im; }

void orc_f_add(const orc_F *a, const orc_F *b, orc_F *out) { fromF(toF(a) + toF(b), out); }
void orc_f_sub(const orc_F *a, const orc_F *b, orc_F *out) { fromF(toF(a) - toF(b), out); }
void orc_f_mul(const orc_F *a, const orc_F *b, orc_F *out) { fromF(toF(a) * toF(b), out); }
void orc_f_neg(const orc_F *a, orc_F *out) { fromF(-toF(a), out); }
void orc_f_inv(const orc_F *a, orc_F *out) { fromF(finv(toF(a)), out); }
void orc_f_root_of_unity(int log_order, orc_F *out) { fromF(root_of_unity(log_order), out); }
void orc_f_random_seq(unsigned seed, int n, orc_F *out) {
    srand(seed);
    for (int i = 0; i < n; ++i) fromF(frandom(), &out[i]);
}
// n further draws of F::random() from the CURRENT glibc state (no reseed): what the reference's verifier draws next.
void orc_f_random_next(int n, orc_F *out) { for (int i = 0; i < n; ++i) fromF(frandom(), &out[i]); }
// Number of F::random() draws fft_circuit_gkr::fft_gkr(lg) consumes (lib/virgo/src/fft_circuit_GKR.cpp), called from
// verify_poly_commitment (vpd_verifier.cpp:92) BEFORE commit_phase draws the FRI fold challenges (:56):
//   fft_gkr                 r[lg]                                              (:840)
//   build_circuit           eval_points[64]                                    (:84)
//   engage_gkr              refresh(r_0, lg+10), refresh(r_1, lg+10)           (:106 via :789-790)
//   addition_layer          refresh(r_u), refresh(r_v), log_uv = lg + 6 each   (:275-276)
//   mult_layer              refresh(r_u), refresh(r_v), lg each                (:394-395)
//   ifft_gkr, per depth     refresh(r_u), refresh(r_v), lg each + alpha, beta  (:563-564, :763-764), lg depths
int orc_fft_gkr_draws(int lg) { return lg + 64 + 2 * (lg + 10) + 2 * (lg + 6) + 2 * lg + lg * (2 * lg + 2); }
void orc_beta_table(const orc_F *r, int n, const orc_F *init, orc_F *out) {
    vector<F> rr(n), beta;
    for (int i = 0; i < n; ++i) rr[i] = toF(&r[i]);
    init_beta_table(beta, n, rr.data(), toF(init));
    for (u64 i = 0; i < (1ull << n); ++i) fromF(beta[i], &out[i]);
}
void orc_update_each(orc_F *V, orc_F *add, orc_F *mult, uint64_t total, uint64_t total_size, const orc_F *prev,
                     orc_F out_poly[3]) {
    // Standalone copy of the loop body of Prover::sumcheckUpdateEach on caller-owned AoS tables.
    auto L = [](orc_F *t, u64 i) { return Lin(F(t[2 * i].real, t[2 * i].img), F(t[2 * i + 1].real, t[2 * i + 1].img)); };
    auto S = [](orc_F *t, u64 i, const Lin &x) { t[2 * i].real = x.a.re; t[2 * i].img = x.a.im; t[2 * i + 1].real = x.b.re; t[2 * i + 1].img = x.b.im; };
    F pr = toF(prev);
    Quad ret(F_ZERO, F_ZERO, F_ZERO);
    for (u64 i = 0; i < (total >> 1); ++i) {
        u64 g0 = i << 1, g1 = i << 1 | 1;
        if (g0 >= total_size) { S(V, i, Lin(F_ZERO)); S(add, i, Lin(F_ZERO)); S(mult, i, Lin(F_ZERO)); continue; }
        if (g1 >= total_size) { S(V, g1, Lin(F_ZERO)); S(add, g1, Lin(F_ZERO)); S(mult, g1, Lin(F_ZERO)); }
        Lin nv = interpolate(L(V, g0).eval(pr), L(V, g1).eval(pr));
        Lin na = interpolate(L(add, g0).eval(pr), L(add, g1).eval(pr));
        Lin nm = interpolate(L(mult, g0).eval(pr), L(mult, g1).eval(pr));
        S(V, i, nv); S(add, i, na); S(mult, i, nm);
        ret = ret + nm * nv + Quad(F_ZERO, na.a, na.b);
    }
    fromF(ret.a, &out_poly[0]); fromF(ret.b, &out_poly[1]); fromF(ret.c, &out_poly[2]);
}

}  // extern "C"

// ====================================================================================================
// Virgo polynomial commitment — commit side.  lib/virgo/src/{RS_polynomial.cpp, poly_commit.h, fri.cpp,
// merkle_tree.cpp, my_hhash.h}.
// ====================================================================================================
namespace {

// ---- SHA3-256 on a 64-byte message (FIPS 202).  The reference links the prebuilt libXKCP.a
// (my_hhash.h:29); this is the published algorithm, pinned by hashlib and the golden Merkle roots.
const u64 KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
const int KECCAK_ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
inline u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
void keccak_f1600(u64 A[25]) {
    for (int rnd = 0; rnd < 24; ++rnd) {
        u64 C[5], D[5], B[25];
        for (int x = 0; x < 5; ++x) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
        for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) A[i] ^= D[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(A[x + 5 * y], KECCAK_ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        A[0] ^= KECCAK_RC[rnd];
    }
}
Digest hhash(const u64 in[8]) {           // my_hhash: 64 bytes in, 32 bytes out
    u64 A[25] = {0};
    for (int i = 0; i < 8; ++i) A[i] = in[i];
    A[8] ^= 0x06;                          // SHA3 domain bits + first pad bit at byte 64
    A[16] ^= 0x8000000000000000ull;        // last pad bit at byte 135 (rate = 136)
    keccak_f1600(A);
    Digest d; for (int i = 0; i < 4; ++i) d.w[i] = A[i];
    return d;
}

// ---- FFT over F_p^2.  Any correct DFT gives the same canonical values as RS_polynomial.cpp:26-157.
void fft_pow2(vector<F> &a, const F &root) {       // in-place, natural in / natural out, a.size() = power of two
    const u64 n = a.size();
    for (u64 i = 1, j = 0; i < n; ++i) {
        u64 bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (u64 len = 2; len <= n; len <<= 1) {
        F wl = root;
        for (u64 m = n; m > len; m >>= 1) wl = wl * wl;
        vector<F> tw(len / 2);
        tw[0] = F_ONE;
        for (u64 k = 1; k < len / 2; ++k) tw[k] = tw[k - 1] * wl;
        for (u64 i = 0; i < n; i += len)
            for (u64 k = 0; k < len / 2; ++k) {
                F u = a[i + k], v = a[i + k + len / 2] * tw[k];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}
int log2_exact(u64 x) { int b = 0; while ((1ull << b) < x) ++b; return b; }
void fft_eval(const F *coefs, u64 coef_len, u64 order, F *out) {           // fast_fourier_transform
    vector<F> a(order, F_ZERO);
    for (u64 i = 0; i < coef_len; ++i) a[i] = coefs[i];
    fft_pow2(a, root_of_unity(log2_exact(order)));
    for (u64 i = 0; i < order; ++i) out[i] = a[i];
}
void ifft_eval(const F *evals, u64 n, F *out) {                             // inverse_fast_fourier_transform, coef_len == order
    vector<F> a(evals, evals + n);
    fft_pow2(a, finv(root_of_unity(log2_exact(n))));
    const F inv_n = fpow(F((long long) n), (u128) P - 2);                    // RS_polynomial.cpp:214
    for (u64 i = 0; i < n; ++i) out[i] = a[i] * inv_n;
}

// ---- Merkle tree (merkle_tree.cpp:7-51): root of 2^k leaf digests.
Digest merkle_root(vector<Digest> level) {
    while (level.size() > 1) {
        vector<Digest> up(level.size() / 2);
        for (size_t i = 0; i < up.size(); ++i) {
            u64 in[8];
            memcpy(in, level[2 * i].w, 32); memcpy(in + 4, level[2 * i + 1].w, 32);
            up[i] = hhash(in);
        }
        level.swap(up);
    }
    return level[0];
}

// leaf hashes of fri::request_init_commit (fri.cpp:95-124): chain over the 64 slices then the mask slice
vector<Digest> leaf_hashes(const vector<F> &evals /* [65][slice_size] */, u64 slice_size) {
    const u64 half = slice_size / 2;
    vector<Digest> leaves(half);
    for (u64 i = 0; i < half; ++i) {
        Digest h; memset(&h, 0, sizeof h);
        for (int s = 0; s < 65; ++s) {
            const F &x = evals[(u64) s * slice_size + i], &y = evals[(u64) s * slice_size + i + half];
            u64 in[8] = {x.re, x.im, y.re, y.im, h.w[0], h.w[1], h.w[2], h.w[3]};
            h = hhash(in);
        }
        leaves[i] = h;
    }
    return leaves;
}

// commit_private_array (poly_commit.h:41-124) with the one-element zero mask prover::commit_private passes
// (src/prover.cpp:526): l_eval[i] = RS encoding (rate 1/32) of slice i of circuitValue[0], mask slice all zero.
void commit_private_evals(const vector<F> &input, int n_bits, vector<F> &l_eval, u64 &slice_size) {
    const u64 slice_real = 1ull << (n_bits - 6);
    slice_size = 1ull << (n_bits - 1);
    l_eval.assign(65 * slice_size, F_ZERO);
    vector<F> coef(slice_real);
    for (int i = 0; i < 64; ++i) {
        bool all_zero = true;
        for (u64 j = 0; j < slice_real; ++j) if (input[i * slice_real + j] != F_ZERO) { all_zero = false; break; }
        if (all_zero) continue;                                              // poly_commit.h:89-99
        ifft_eval(&input[i * slice_real], slice_real, coef.data());
        fft_eval(coef.data(), slice_real, slice_size, &l_eval[(u64) i * slice_size]);
    }
}

}  // namespace

extern "C" {

void orc_sha3_256_64(const uint8_t in[64], uint8_t out[32]) {
    u64 w[8];
    memcpy(w, in, 64);
    Digest d = hhash(w);
    memcpy(out, d.w, 32);
}
void orc_fft(const orc_F *coefs, int coef_len, int order, orc_F *out) {
    vector<F> c(coef_len), o(order);
    for (int i = 0; i < coef_len; ++i) c[i] = F(coefs[i].real, coefs[i].img);
    fft_eval(c.data(), coef_len, order, o.data());
    for (int i = 0; i < order; ++i) { out[i].real = o[i].re; out[i].img = o[i].im; }
}
void orc_ifft(const orc_F *evals, int n, orc_F *out) {
    vector<F> e(n), o(n);
    for (int i = 0; i < n; ++i) e[i] = F(evals[i].real, evals[i].img);
    ifft_eval(e.data(), n, o.data());
    for (int i = 0; i < n; ++i) { out[i].real = o[i].re; out[i].img = o[i].im; }
}
int orc_commit_private(orc_circuit *oc, uint8_t root[32]) {
    const Circuit &C = oc->c;
    const int n_bits = C.circuit[0].bitLength;
    if (n_bits < 7) return -1;                                               // vpd_verifier.cpp:115
    vector<F> input(1ull << n_bits, F_ZERO);                                  // circuitValue[0], padded (prover.cpp:30)
    for (u64 g = 0; g < C.circuit[0].size; ++g) input[g] = F((long long) C.circuit[0].gates[g].u);
    vector<F> l_eval; u64 slice_size;
    commit_private_evals(input, n_bits, l_eval, slice_size);
    Digest r = merkle_root(leaf_hashes(l_eval, slice_size));
    memcpy(root, r.w, 32);
    return 0;
}

}  // extern "C"

// ---- commit_public_array (poly_commit.h:126-349) with zero masks ---------------------------------------
namespace {
struct PublicOut { F inner; F all_sum[65]; Digest root_h; };
void commit_public_core(const vector<F> &input, const vector<F> &pub, int n_bits, u64 n_used, PublicOut &out, vector<F> *vo = nullptr) {
    const u64 N = 1ull << (n_bits - 6);
    vector<F> l_eval, q_eval; u64 M;
    commit_private_evals(input, n_bits, l_eval, M);                       // the prover still holds l_eval (poly_commit.cpp:4-13)
    commit_private_evals(pub, n_bits, q_eval, M);                         // poly_commit.h:163-176 (no all-zero shortcut there; same values)
    out.inner = F_ZERO;                                                    // prover::inner_prod, src/prover.cpp:532-540
    for (u64 i = 0; i < n_used; ++i) out.inner = out.inner + input[i] * pub[i];
    vector<F> h_arr(65 * M, F_ZERO), lq_eval(2 * N), lq_coef(2 * N), h_coef(N);
    if (vo) vo->assign(64 * M, F_ZERO);
    const F rou = root_of_unity(log2_exact(M)), inv_rou = finv(rou), rou_n = fpow(rou, N);
    for (int i = 0; i < 64; ++i) {
        bool all_zero = true;
        const u64 step = M / (2 * N);
        for (u64 j = 0; j < 2 * N; ++j) {
            lq_eval[j] = l_eval[(u64) i * M + j * step] * q_eval[(u64) i * M + j * step];
            if (lq_eval[j] != F_ZERO) all_zero = false;
        }
        if (all_zero) { out.all_sum[i] = F_ZERO; continue; }
        ifft_eval(lq_eval.data(), 2 * N, lq_coef.data());
        for (u64 j = 0; j < N; ++j) h_coef[j] = lq_coef[j + N];
        fft_eval(h_coef.data(), N, M, &h_arr[(u64) i * M]);
        out.all_sum[i] = (lq_coef[0] + h_coef[0]) * F((long long) N);      // poly_commit.h:323
        if (vo) {                                                          // virtual oracle, poly_commit.h:294-318
            F x_n = F_ONE, inv_x = F((long long) N);
            const F const_sum = F_ZERO - (lq_coef[0] + h_coef[0]);
            for (u64 j = 0; j < M; ++j) {
                const F g = l_eval[(u64) i * M + j] * q_eval[(u64) i * M + j] - (x_n - F_ONE) * h_arr[(u64) i * M + j];
                (*vo)[(u64) i * M + j] = (g + const_sum) * inv_x;
                inv_x = inv_x * inv_rou;
                x_n = x_n * rou_n;
            }
        }
    }
    out.all_sum[64] = F_ZERO;                                              // mask slice: zero polynomial (:262)
    out.root_h = merkle_root(leaf_hashes(h_arr, M));
}
}  // namespace

extern "C" {

int orc_commit_public(const orc_F *input, const orc_F *pub, int n_bits, uint64_t n_used, orc_F *inner, orc_F *all_sum,
                      uint8_t root_h[32]) {
    if (n_bits < 7) return -1;
    const u64 n = 1ull << n_bits;
    vector<F> in(n), pb(n);
    for (u64 i = 0; i < n; ++i) { in[i] = F(input[i].real, input[i].img); pb[i] = F(pub[i].real, pub[i].img); }
    PublicOut o;
    commit_public_core(in, pb, n_bits, n_used, o);
    inner->real = o.inner.re; inner->img = o.inner.im;
    for (int i = 0; i < 65; ++i) { all_sum[i].real = o.all_sum[i].re; all_sum[i].img = o.all_sum[i].im; }
    memcpy(root_h, o.root_h.w, 32);
    return 0;
}

int64_t orc_prove_full(orc_circuit *oc, uint8_t *transcript, int64_t capacity, orc_stats *st) {
    srand(3396);
    g_cnt = Counter();
    subset_init(oc->c);
    const Circuit &C = oc->c;
    const int n_bits = C.circuit[0].bitLength;
    if (n_bits < 7) return -2;
    Prover p(C);
    vector<unsigned char> out;
    {   // merkle_root_l (verifier.cpp:137)
        uint8_t root[32];
        orc_commit_private(oc, root);
        out.insert(out.end(), root, root + 32);
    }
    Verifier v(&p, C, &out);
    bool ok = v.verify_gkr();
    // verifyPoly (verifier.cpp:363-379): the public vector is eq(r_liu, .) over the input layer
    vector<F> pub;
    init_beta_table(pub, n_bits, v.r_liu.data(), F_ONE);
    oc->last_point.assign(v.r_liu.begin(), v.r_liu.begin() + n_bits);
    PublicOut po;
    commit_public_core(p.circuitValue[0], pub, n_bits, C.circuit[0].size, po);
    out.insert(out.end(), (unsigned char *) po.root_h.w, (unsigned char *) po.root_h.w + 32);
    v.putF(po.inner);
    for (int i = 0; i < 65; ++i) v.putF(po.all_sum[i]);
    if (st) {
        st->prove_sec = p.prove_timer.total; st->evaluate_sec = p.evaluate_sec; st->verify_sec = v.vt.total;
        st->mult_count = g_cnt.mul; st->add_count = g_cnt.add; st->rounds = p.rounds; st->pairs = p.pairs;
        st->proof_kb = (double) p.proof_size / 1024.0; st->verified = ok ? 1 : 0;
    }
    if ((int64_t) out.size() > capacity) return -1;
    memcpy(transcript, out.data(), out.size());
    return (int64_t) out.size();
}

}  // extern "C"

extern "C" int orc_fri_commit(const orc_F *input, const orc_F *pub, int n_bits, const orc_F *r, uint8_t *roots, orc_F *final_code) {
    if (n_bits < 7) return -1;
    const u64 n = 1ull << n_bits;
    vector<F> in(n), pb(n), vo;
    for (u64 i = 0; i < n; ++i) { in[i] = F(input[i].real, input[i].img); pb[i] = F(pub[i].real, pub[i].img); }
    PublicOut po;
    commit_public_core(in, pb, n_bits, n, po, &vo);
    u64 M = 1ull << (n_bits - 1);
    const int steps = n_bits - 6;
    const F inv2 = finv(F(2ll));
    vector<F> cur = vo;                                                   // [64][M]
    F w = root_of_unity(log2_exact(M));
    for (int k = 0; k < steps; ++k) {
        const F rk(r[k].real, r[k].img);
        const u64 half = M / 2;
        const F inv_w = finv(w);
        vector<F> nxt(64 * half);
        F inv_mu = F_ONE;                                                 // w^-i  (L_group[(M - i) & (M - 1)], fri.cpp:322)
        for (u64 i = 0; i < half; ++i) {
            const F c = inv_mu * rk;
            for (int s = 0; s < 64; ++s) {
                const F a = cur[(u64) s * M + i], b = cur[(u64) s * M + i + half];
                nxt[(u64) s * half + i] = inv2 * ((a + b) + c * (a - b));  // fri.cpp:327-329
            }
            inv_mu = inv_mu * inv_w;
        }
        // leaves of the new codeword: pairs (i, i + half/2) chained over the slices, then the (zero) mask pair
        vector<F> padded(65 * half, F_ZERO);
        std::copy(nxt.begin(), nxt.end(), padded.begin());
        Digest root = merkle_root(leaf_hashes(padded, half));
        memcpy(roots + 32 * k, root.w, 32);
        cur.swap(nxt);
        M = half;
        w = w * w;
    }
    for (u64 i = 0; i < 16; ++i)
        for (int s = 0; s < 64; ++s)
            for (int hi = 0; hi < 2; ++hi) {
                const F x = cur[(u64) s * 32 + i + 16 * hi];
                orc_F &o = final_code[(i << 7) | (s << 1) | hi];
                o.real = x.re; o.img = x.im;
            }
    return 0;
}

extern "C" int orc_last_point(const orc_circuit *oc, orc_F *out, int n) {
    if ((int) oc->last_point.size() < n) return -1;
    for (int i = 0; i < n; ++i) { out[i].real = oc->last_point[i].re; out[i].img = oc->last_point[i].im; }
    return 0;
}

// ====================================================================================================
// fft_gkr — lib/virgo/src/fft_circuit_GKR.cpp:22-849.  verify_poly_commitment runs it between the commitment and
// the FRI commit phase (vpd_verifier.cpp:92): a self-contained GKR (prover AND verifier in one function, verifier randomness from
// the same glibc stream) over the circuit  r -> eq-expansion E(r) -> inverse FFT (lg butterfly layers) -> scaling by 1/n ->
// 64 polynomial evaluations at random points (a multiplication layer of 64 * 2^lg products and an addition layer of 64 sums).
// Restated here on VALUE tables (fold, then sum the pair products — the same field elements as the reference's linear_poly tables,
// :156-188); pinned against the real reference's record (oracle/ref_driver.cpp --dump-fft: the 64 outputs, every round polynomial,
// every v_u / v_v) by tests/test_oracle_golden.py.  Draw order = the reference's: r[lg] (:840), eval_points[64] (:84), r_0, r_1 of
// lg + 10 (:789-790), per sumcheck r_u then r_v (:275-276, :394-395, :563-564), alpha, beta after every inverse-FFT depth (:763-764).
// ====================================================================================================
namespace {

struct FftGkr {
    int lg;
    vector<F> msgs;                                  // outputs[64] | per sumcheck: round polynomials, then the claimed table value(s)
    vector<vector<F>> B;                             // B[0] = E(r), B[t] = after the butterflies of depth lg - t (t = 1..lg)
    vector<F> S, Pm, O, xs;                          // scaled coefficients, the 64 * 2^lg products, the 64 sums, the evaluation points
    F alpha, beta;
    vector<F> r0, r1, ru, rv;
    bool ok = true;
    double p_sec = 0, v_sec = 0;
    u64 rounds = 0, pairs = 0;

    static void draws(vector<F> &v, int n) { for (int i = 0; i < n; ++i) v[i] = frandom(); }
    // alpha * eq(r0[0..n), g) + beta * eq(r1[0..n), g), bit b of g set <-> r[b] (the beta_g half tables, :193-216)
    vector<F> g_table(int n) const {
        vector<F> t0, t1;
        init_beta_table(t0, n, r0.data(), alpha);
        init_beta_table(t1, n, r1.data(), beta);
        vector<F> g(1ull << n);
        for (u64 i = 0; i < g.size(); ++i) g[i] = t0[i] + t1[i];
        return g;
    }
    // one sumcheck over (V, M, A): `n` rounds with challenges ch[0..n); appends the polynomials; returns V's last value
    F sumcheck(vector<F> V, vector<F> M, vector<F> A, int n, const F *ch, F &claim) {
        Timer t; t.start();
        for (int k = 0; k < n; ++k) {
            const u64 half = V.size() >> 1;
            Quad q(F_ZERO, F_ZERO, F_ZERO);
            for (u64 i = 0; i < half; ++i) {
                const F v0 = V[2 * i], dv = V[2 * i + 1] - v0, m0 = M[2 * i], dm = M[2 * i + 1] - m0, a0 = A[2 * i], da = A[2 * i + 1] - a0;
                q.a = q.a + dm * dv;
                q.b = q.b + dm * v0 + m0 * dv + da;
                q.c = q.c + m0 * v0 + a0;
            }
            pairs += half; ++rounds;
            msgs.push_back(q.a); msgs.push_back(q.b); msgs.push_back(q.c);
            t.stop();
            Timer tv; tv.start();
            if (q.eval(F_ZERO) + q.eval(F_ONE) != claim) ok = false;              // :264-266
            claim = q.eval(ch[k]);
            tv.stop(); v_sec += tv.total;
            t.start();
            for (u64 i = 0; i < half; ++i) {
                V[i] = V[2 * i] + ch[k] * (V[2 * i + 1] - V[2 * i]);
                M[i] = M[2 * i] + ch[k] * (M[2 * i + 1] - M[2 * i]);
                A[i] = A[2 * i] + ch[k] * (A[2 * i + 1] - A[2 * i]);
            }
            V.resize(half); M.resize(half); A.resize(half);
        }
        t.stop(); p_sec += t.total;
        msgs.push_back(V[0]);
        return V[0];
    }

    void build(const vector<F> &r) {                                              // build_circuit, :22-101
        const u64 N = 1ull << lg;
        B.assign(lg + 1, vector<F>());
        vector<F> e(1, F_ONE);
        for (int i = 0; i < lg; ++i) {
            vector<F> nx(e.size() * 2);
            for (u64 j = 0; j < e.size(); ++j) { nx[2 * j] = e[j] * r[i]; nx[2 * j + 1] = e[j] * (F_ONE - r[i]); }
            e.swap(nx);
        }
        B[0] = e;
        const F inv_rou = finv(root_of_unity(lg));
        for (int dep = lg - 1; dep >= 0; --dep) {
            const int t = lg - dep;
            const u64 half = 1ull << (lg - dep - 1), J = 1ull << dep;
            F w = inv_rou;
            for (int q = 0; q < dep; ++q) w = w * w;                                // rot_mul[dep] = inv_rou^(2^dep)
            B[t].assign(N, F_ZERO);
            F x = F_ONE;
            for (u64 k = 0; k < half; ++k) {
                for (u64 j = 0; j < J; ++j) {
                    const F l = B[t - 1][k << (dep + 1) | j], rr = x * B[t - 1][k << (dep + 1) | J | j];
                    B[t][k << dep | j] = l + rr;
                    B[t][(k + half) << dep | j] = l - rr;
                }
                x = x * w;
            }
        }
        const F inv_n = fpow(F((long long) N), (u128) P - 2);
        S.resize(N);
        for (u64 i = 0; i < N; ++i) S[i] = B[lg][i] * inv_n;
        xs.resize(64); Pm.resize(64 * N); O.assign(64, F_ZERO);
        for (int i = 0; i < 64; ++i) {
            xs[i] = frandom();
            F x = F_ONE;
            for (u64 j = 0; j < N; ++j) { Pm[j + ((u64) i << lg)] = S[j] * x; x = x * xs[i]; }
        }
        for (int i = 0; i < 64; ++i) for (u64 j = 0; j < N; ++j) O[i] = O[i] + Pm[j + (u64) i * N];
    }

    void run() {
        const u64 N = 1ull << lg;
        vector<F> r(lg);
        draws(r, lg);
        build(r);
        for (int i = 0; i < 64; ++i) msgs.push_back(O[i]);
        alpha = F_ONE; beta = F_ZERO;                                              // engage_gkr, :774-831
        r0.assign(lg + 10, F_ZERO); r1 = r0; ru = r0; rv = r0;
        draws(r0, lg + 10); draws(r1, lg + 10);
        vector<F> o = O;                                                           // V_output, :121-136
        for (int i = 0; i < 6; ++i) { for (u64 j = 0; j < o.size() / 2; ++j) o[j] = o[2 * j] * (F_ONE - r0[i]) + o[2 * j + 1] * r0[i]; o.resize(o.size() / 2); }
        F claim = o[0];
        {   // addition layer (:190-309): the 64 sums; g ranges over the 64 outputs, u over the 64 * 2^lg products
            const int n = lg + 6;
            const vector<F> g = g_table(6);
            vector<F> M(64 * N), A(64 * N, F_ZERO);
            for (u64 j = 0; j < 64 * N; ++j) M[j] = g[j >> lg];
            draws(ru, n); draws(rv, n);
            const F vu = sumcheck(Pm, M, A, n, ru.data(), claim);
            Timer tv; tv.start();
            vector<F> eu;
            init_beta_table(eu, 6, ru.data() + lg, F_ONE);
            F s = F_ZERO;
            for (int i = 0; i < 64; ++i) s = s + g[i] * eu[i];
            if (claim != s * vu) ok = false;
            tv.stop(); v_sec += tv.total;
            for (int i = 0; i < n; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
            claim = alpha * vu;
        }
        {   // multiplication layer (:311-447): product j * 2^lg + i = S[i] * x_j^i
            const vector<F> g = g_table(lg + 6);
            vector<F> M(N, F_ZERO), A(N, F_ZERO), xp(64, F_ONE);
            for (u64 i = 0; i < N; ++i)
                for (int j = 0; j < 64; ++j) { M[i] = M[i] + g[(u64) j * N + i] * xp[j]; xp[j] = xp[j] * xs[j]; }
            draws(ru, lg); draws(rv, lg);
            const F vu = sumcheck(S, M, A, lg, ru.data(), claim);
            Timer tv; tv.start();
            F s = F_ZERO;
            for (int i = 0; i < 64; ++i) {
                F g0 = alpha, g1 = beta;
                for (int j = 0; j < 6; ++j) {
                    if ((i >> j) & 1) { g0 = g0 * r0[lg + j]; g1 = g1 * r1[lg + j]; }
                    else { g0 = g0 * (F_ONE - r0[lg + j]); g1 = g1 * (F_ONE - r1[lg + j]); }
                }
                F u0 = F_ONE, u1 = F_ONE, x = xs[i];
                for (int j = 0; j < lg; ++j) {
                    u0 = u0 * (r0[j] * ru[j] * x + (F_ONE - r0[j]) * (F_ONE - ru[j]));
                    u1 = u1 * (r1[j] * ru[j] * x + (F_ONE - r1[j]) * (F_ONE - ru[j]));
                    x = x * x;
                }
                s = s + g0 * u0 + g1 * u1;
            }
            if (claim != s * vu) ok = false;
            tv.stop(); v_sec += tv.total;
            for (int i = 0; i < lg; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
            claim = alpha * vu;
        }
        claim = claim * F((long long) N);                                          // intermediate_layer, :449-456
        const F inv_rou = finv(root_of_unity(lg));
        for (int dep = 0; dep < lg; ++dep) {                                       // ifft_gkr, :458-768
            const vector<F> &pre = B[lg - dep - 1];
            const vector<F> g = g_table(lg);
            const u64 half = 1ull << (lg - dep - 1), J = 1ull << dep;
            F w = inv_rou;
            for (int q = 0; q < dep; ++q) w = w * w;
            vector<F> M(N, F_ZERO), A(N, F_ZERO);
            F x = F_ONE;
            for (u64 k = 0; k < half; ++k) {
                for (u64 j = 0; j < J; ++j) {
                    const u64 u = k << (dep + 1) | j, v = u | J;
                    const F t1 = g[k << dep | j], t2 = g[(k + half) << dep | j];
                    M[u] = t1 + t2;
                    A[u] = t1 * x * pre[v] - t2 * x * pre[v];
                }
                x = x * w;
            }
            draws(ru, lg); draws(rv, lg);
            const F vu = sumcheck(pre, M, A, lg, ru.data(), claim);
            vector<F> eu;
            init_beta_table(eu, lg, ru.data(), F_ONE);
            M.assign(N, F_ZERO); A.assign(N, F_ZERO);
            x = F_ONE;
            for (u64 k = 0; k < half; ++k) {
                for (u64 j = 0; j < J; ++j) {
                    const u64 u = k << (dep + 1) | j, v = u | J;
                    const F a1 = g[k << dep | j] * eu[u], a2 = g[(k + half) << dep | j] * eu[u];
                    M[v] = a1 * x - a2 * x;
                    A[v] = a1 * vu + a2 * vu;
                }
                x = x * w;
            }
            const F vv = sumcheck(pre, M, A, lg, rv.data(), claim);
            Timer tv; tv.start();
            {   // the verifier's closed form of the wiring predicate at (r_0 | r_1, r_u, r_v), :639-752
                const int lj = dep, lk = lg - dep - 1;
                const F hu = (F_ONE - ru[lj]) * rv[lj];
                F uA0 = (F_ONE - r0[lg - 1]) * hu * alpha, uA1 = (F_ONE - r1[lg - 1]) * hu * beta, vA0 = uA0, vA1 = uA1;
                F uB0 = r0[lg - 1] * hu * alpha, uB1 = r1[lg - 1] * hu * beta, vB0 = uB0, vB1 = uB1;
                F xx = w;
                for (int i = 0; i < lk; ++i) {
                    const F p0 = r0[lj + i] * ru[lj + 1 + i] * rv[lj + 1 + i], q0 = (F_ONE - r0[lj + i]) * (F_ONE - ru[lj + 1 + i]) * (F_ONE - rv[lj + 1 + i]);
                    const F p1 = r1[lj + i] * ru[lj + 1 + i] * rv[lj + 1 + i], q1 = (F_ONE - r1[lj + i]) * (F_ONE - ru[lj + 1 + i]) * (F_ONE - rv[lj + 1 + i]);
                    uA0 = uA0 * (p0 + q0); uA1 = uA1 * (p1 + q1); uB0 = uB0 * (p0 + q0); uB1 = uB1 * (p1 + q1);
                    vA0 = vA0 * (p0 * xx + q0); vA1 = vA1 * (p1 * xx + q1); vB0 = vB0 * (q0 + p0 * xx); vB1 = vB1 * (q1 + p1 * xx);
                    xx = xx * xx;
                }
                for (int i = 0; i < lj; ++i) {
                    const F e0 = r0[i] * ru[i] * rv[i] + (F_ONE - r0[i]) * (F_ONE - ru[i]) * (F_ONE - rv[i]);
                    const F e1 = r1[i] * ru[i] * rv[i] + (F_ONE - r1[i]) * (F_ONE - ru[i]) * (F_ONE - rv[i]);
                    uA0 = uA0 * e0; vA0 = vA0 * e0; uB0 = uB0 * e0; vB0 = vB0 * e0;
                    uA1 = uA1 * e1; vA1 = vA1 * e1; uB1 = uB1 * e1; vB1 = vB1 * e1;
                }
                if (claim != (uA0 + uA1 + uB0 + uB1) * vu + (vA0 + vA1 - vB0 - vB1) * vv) ok = false;
            }
            tv.stop(); v_sec += tv.total;
            for (int i = 0; i < lg; ++i) { r0[i] = ru[i]; r1[i] = rv[i]; }
            alpha = frandom(); beta = frandom();
            claim = alpha * vu + beta * vv;
        }
    }
};

}  // namespace

extern "C" int64_t orc_fft_gkr(int lg, long seed, uint8_t *msgs, int64_t capacity, double *prove_sec, int *verified) {
    if (lg < 1 || lg > 24) return -2;
    if (seed >= 0) srand((unsigned) seed);
    FftGkr f; f.lg = lg;
    f.run();
    if (prove_sec) *prove_sec = f.p_sec;
    if (verified) *verified = f.ok ? 1 : 0;
    const int64_t n = (int64_t) f.msgs.size() * 16;
    if (n > capacity) return -1;
    for (size_t i = 0; i < f.msgs.size(); ++i) { u64 w[2] = {f.msgs[i].re, f.msgs[i].im}; memcpy(msgs + 16 * i, w, 16); }
    return n;
}
