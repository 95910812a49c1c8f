// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// The reference driven through libvpgpu.so at the block counts its own main() cannot reach (src/main.cpp:8 fixes `repeat = 1`): this file
// replaces ONLY main() (src/main.cpp:145-159, compiled as ref_main) — the B-fold SHA-256 DAG goes through the reference's own
// DAG_to_layered(), F::init(), subsetInit(), and then, exactly as src/main.cpp:154-157 does,
//     prover p(c); verifier v(&p, c); v.verify();
// with `class prover`, namespace virgo::fri and fft_gkr coming from INTEGRATION.md's three forwarding files.  The reference's verifier
// (src/verifier.cpp, lib/virgo/src/vpd_verifier.cpp, unmodified, no hooks) decides; VPI_DUMP / VPI_DUMP_FRI / VPI_DUMP_FFT (vpgpu_glue.h)
// write what the forwarding files handed over, for comparison with the CPU reference's records of the same circuit and seed.
//   ref_run_vpgpu_blocks FILE.pws BLOCKS
#include "verifier.h"
#include "inputCircuit.hpp"
#include "../ref_replicate.hpp"
#include <chrono>

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s FILE.pws BLOCKS\n", argv[0]); return 2; }
    const int blocks = atoi(argv[2]);
    if (blocks < 1) return 2;
    in_circuit_dag.clear();
    populate_replicated(argv[1], blocks);
    DAG_to_layered();                                          // src/main.cpp:15
    F::init();                                                 // src/main.cpp:152
    c.subsetInit();                                            // src/main.cpp:153
    prover p(c);                                               // src/main.cpp:154
    verifier v(&p, c);                                         // src/main.cpp:155
    const auto t0 = std::chrono::high_resolution_clock::now();
    const bool ok = v.verify();                                // src/main.cpp:156
    const auto t1 = std::chrono::high_resolution_clock::now();
    fprintf(stdout, "blocks %d verify_wall_sec %.3f ok %d\n", blocks, std::chrono::duration<double>(t1 - t0).count(), ok ? 1 : 0);
    return ok ? 0 : 1;
}
