// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// B-fold block-replicated SHA-256 circuits for the reference's own DAG_to_layered() (src/main.cpp:15), without a 224 MB .pws text file and
// its regex parser.  Shared by oracle/ref_driver.cpp (the CPU reference) and
// oracle/integration/blocks_main.cpp (the reference driven through libvpgpu.so).  Include after the reference's verifier.h / inputCircuit.hpp.
#pragma once
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

// symbols defined in the reference's src/main.cpp
extern layeredCircuit c;
extern std::vector<DAG_gate *> in_circuit_dag;
void DAG_to_layered();
void parse(std::ifstream &circuit_in);
DAG_gate *buildGate(gateType ty, u64 tgt, u64 src0, u64 src1, bool has_constant);
DAG_gate *buildInput(u64 tgt, u64 src0);

struct pws_gate { gateType ty; u64 tgt, s0, s1; };

// Read a .pws file without regexes.  Same grammar as src/main.cpp:161-168.
static void read_pws(const char *path, std::vector<u64> &inputs, std::vector<pws_gate> &gates) {
    FILE *f = fopen(path, "r");
    if (!f) { perror(path); exit(2); }
    char line[256];
    while (fgets(line, sizeof line, f)) {
        long long t, a, b; char op[16];
        if (sscanf(line, "P V%lld = I%lld E", &t, &a) == 2) { inputs.push_back(t); continue; }
        if (sscanf(line, "P O%lld = V%lld E", &t, &a) == 2) continue;
        if (sscanf(line, "P V%lld = V%lld %15s V%lld E", &t, &a, op, &b) == 4) {
            gateType ty;
            if (!strcmp(op, "+")) ty = Add;
            else if (!strcmp(op, "*")) ty = Mul;
            else if (!strcmp(op, "XOR")) ty = Xor;
            else if (!strcmp(op, "minus")) ty = Sub;
            else if (!strcmp(op, "NAAB")) ty = Naab;
            else if (!strcmp(op, "NOT")) ty = Not;
            else { fprintf(stderr, "bad op %s\n", op); exit(2); }
            gates.push_back({ty, (u64) t, (u64) a, (u64) b});
            continue;
        }
        fprintf(stderr, "unparsed line: %s", line); exit(2);
    }
    fclose(f);
}

// Populate in_circuit_dag with B copies of the DAG: all inputs first (block b
// input k -> b*nin+k, witness drawn in that order exactly as parse() would on
// the replicated file), then gates block-major.
static void populate_replicated(const char *path, int B) {
    std::vector<u64> inputs; std::vector<pws_gate> gates;
    read_pws(path, inputs, gates);
    u64 nin = inputs.size(), ng = gates.size();
    for (u64 k = 0; k < nin; ++k) if (inputs[k] != k) { fprintf(stderr, "inputs not V0..\n"); exit(2); }
    for (u64 g = 0; g < ng; ++g) if (gates[g].tgt != nin + g) { fprintf(stderr, "gates not dense\n"); exit(2); }
    for (int b = 0; b < B; ++b)
        for (u64 k = 0; k < nin; ++k)
            buildInput(b * nin + k, random() % virgo::fieldElement::mod);   // src/main.cpp:188
    auto map = [&](int b, u64 id) { return id < nin ? b * nin + id : (u64) B * nin + b * ng + (id - nin); };
    for (int b = 0; b < B; ++b)
        for (u64 g = 0; g < ng; ++g) {
            auto &x = gates[g];
            if (x.ty == Not) buildGate(Not, map(b, x.tgt), map(b, x.s0), 0, true);        // src/main.cpp:202
            else buildGate(x.ty, map(b, x.tgt), map(b, x.s0), map(b, x.s1), false);
        }
}

