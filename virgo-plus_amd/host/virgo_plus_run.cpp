// Command-line driver with the reference's interface (src/main.cpp:145-159):
//     virgo_plus_run <file.pws> [--blocks B] [--batched | --fs] [--seed S] [--device D] [--dump transcript.bin]
//     --fs: non-interactive GKR proof (Fiat-Shamir over SHA3-256), then verified from the proof bytes alone
// Loads the circuit, runs the GKR proof on the GPU against the host verifier and prints the
// reference's result lines (interactive mode runs the whole protocol incl. the Virgo commitment and its verification).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "circuit.hpp"
#include "prover.hpp"
#include "verifier.hpp"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <file.pws> [--blocks B] [--batched | --fs] [--device D] [--dump out.bin]\n", argv[0]); return 2; }
    int blocks = 1, device = 0; bool batched = false, fsmode = false; const char *dump = nullptr;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--blocks" && i + 1 < argc) blocks = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--batched") batched = true;
        else if (a == "--fs") fsmode = true;
        else if (a == "--seed" && i + 1 < argc) srandom((unsigned) atol(argv[++i]));      // witness draw (default: glibc's initial state, as the reference)
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else { fprintf(stderr, "bad argument %s\n", argv[i]); return 2; }
    }
    std::vector<DAG_gate> dag;
    std::string err;
    if (!vph::parse_pws(argv[1], blocks, dag, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    layeredCircuit c = vph::DAG_to_layered(dag);
    F::init();
    c.subsetInit();
    try {
        prover p(c, device);
        bool ok;
        std::vector<uint8_t> tr;
        double vt = 0, pc_pt = -1;
        if (fsmode) {
            verifier v(&p, c);
            ok = v.proveFS();
            tr = v.transcript();
            verifier w(nullptr, c);                  // a verifier that has only the proof
            ok = ok && w.checkFS(tr);
            vt = w.verifyTime();
            batched = true;                          // report the proof's own size
        } else if (batched) {
            verifier v(nullptr, c);
            std::vector<F> tape = v.drawTape();
            p.proveGKR(tape, tr);
            ok = v.check(tape, tr);
            vt = v.verifyTime();
        } else if (c.circuit[0].bitLength >= 7) {
            verifier v(&p, c);                       // the reference's flow: commitment on (src/verifier.cpp:134-189)
            ok = v.verifyFull();
            tr = v.fullTranscript();
            vt = v.verifyTime() + v.polyVerifyTime();
            pc_pt = v.polyProveTime();
        } else {
            verifier v(&p, c);
            ok = v.verify();
            tr = v.transcript();
            vt = v.verifyTime();
        }
        if (!ok) { fprintf(stderr, "Verification fail\n"); return 1; }
        fprintf(stderr, "Verification pass\n");
        fprintf(stdout, "Input size %d\n", (int) c.circuit[0].size);
        fprintf(stdout, "Prove Time %lf\n", p.proveTime());
        fprintf(stdout, "verify time %lf\n", vt);
        fprintf(stdout, "proof size = %lf kb\n", batched ? tr.size() / 1024.0 : p.proofSize());
        if (pc_pt >= 0) fprintf(stdout, "Polynomial commitment: prove time %lf\n", pc_pt);
        if (dump) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(tr.data(), 1, tr.size(), f); fclose(f); } }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
    return 0;
}
