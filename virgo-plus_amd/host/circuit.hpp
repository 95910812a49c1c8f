// Circuit model and loader: the host-side API surface of the reference's src/circuit.h:11-47,
// src/inputCircuit.hpp:13-23 and the loader in src/main.cpp:15-137,161-231, kept on the host (north
// star: "keeps the existing ... circuit-loader API surface").  The device only ever sees the flat
// structure-of-arrays view produced by prover (vp_layer_desc in include/vpgpu.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "field.hpp"

typedef long long i64;
typedef unsigned long long u64;

enum gateType { Mul, Add, Sub, AntiSub, Naab, AntiNaab, Input, Mulc, Addc, Xor, Not, Copy, SIZE };

class gate {
public:
    gateType ty;
    int l;
    u64 u, v, lv;
    F c;
    bool is_assert;
    gate() : ty(Input), l(-1), u(0), v(0), lv(0), is_assert(false) {}
    gate(gateType t, int ll, u64 uu, u64 vv, const F &cc, bool is_assert_zero)
        : ty(t), l(ll), u(uu), v(vv), lv(0), c(cc), is_assert(is_assert_zero) {}
};

class layer {
public:
    std::vector<gate> gates;
    int bitLength = 0;
    u64 size = 0;
    std::vector<std::vector<u64>> dadId;   // subset id -> real id, per source layer
    std::vector<int> dadBitLength;         // -1: subset is empty (the reference leaves (int)log2(0) here)
    std::vector<u64> dadSize;
    u64 maxDadSize = 0;
    int maxDadBitLength = -1;
};

// DAG node as produced by the .pws parser (src/inputCircuit.hpp:17-23)
struct DAG_gate {
    std::pair<int, u64> input0, input1;    // ('V', id) wire | ('S', value) constant | ('N', 0) none
    bool is_assert = false;
    gateType ty = Input;
};

class layeredCircuit {
public:
    std::vector<layer> circuit;
    int size = 0;

    // src/circuit.cpp:17-41 (the draws are consumed in the order g++ evaluates the constructor
    // arguments there: v, u, l, type bit)
    static layeredCircuit randomize(int layerNum, int eachLayer);
    // src/circuit.cpp:43-80
    void subsetInit();
    // 128-bit structural hash, same serialisation as oracle/ref_driver.cpp (tests)
    void structuralHash(u64 out[2]) const;
    // SHA3-256 of the same serialisation (every gate tuple, the subset tables, the layer sizes and — through gate::u of
    // layer 0 — the input values): the statement digest the Fiat-Shamir mode binds its challenges to.  structuralHash is
    // only a loader-parity fingerprint (FNV + xor-multiply, invertible); this one is collision resistant.  Not cached (the gate
    // tables are public and may be edited between proofs).
    void statementDigest(u64 out[4]) const;
};

namespace vph {

// .pws text -> DAG (grammar of src/main.cpp:161-168; hand-written scanner instead of std::regex),
// replicated `blocks` times with all inputs numbered first.  Input values are
// drawn with glibc random() % p in file order, as src/main.cpp:188 does.  Returns false on a syntax error.
bool parse_pws(const std::string &path, int blocks, std::vector<DAG_gate> &dag, std::string *err);
// src/main.cpp:15-137
layeredCircuit DAG_to_layered(const std::vector<DAG_gate> &dag);
// `blocks` independent copies of the circuit in `path`, layered and with subsets initialised, built from the layered form of one
// block (same result as parse_pws(path, blocks) -> DAG_to_layered -> subsetInit, in linear sequential time)
bool build_replicated(const std::string &path, int blocks, layeredCircuit &out, std::string *err);

}  // namespace vph
