// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// Shared between the three forwarding files a reference maintainer would add:
//   prover_vpgpu.cpp   — bodies of src/prover.cpp        (class prover, src/prover.h:12-66)
//   fri_vpgpu.cpp      — bodies of lib/virgo/src/fri.cpp (namespace virgo::fri, lib/virgo/src/fri.h:56-104)
//   fft_gkr_vpgpu.cpp  — body of lib/virgo/src/fft_circuit_GKR.cpp's fft_gkr (lib/virgo/src/fft_circuit_GKR.h:4)
// The reference has one prover per process (not re-entrant), so the device context is one
// process-wide object owned by prover_vpgpu.cpp.
#pragma once
#include <vpgpu.h>
#include <chrono>
#include <cstdio>

vp_ctx *vpi_ctx();                                   // the context prover::prover() created (exits if there is none)
vp_ctx *vpi_ctx_standalone();                        // ... or creates one without a circuit (masked_main.cpp: the commitment alone)
void vpi_must(int rc, const char *what);             // VP_OK or: print vp_last_error, exit(EXIT_FAILURE) — the reference's own error style
// commit_private / commit_public tell the FRI side which oracle now exists (its root and the input bit length):
// fri::request_init_commit(bit_len, oracle) returns exactly this (lib/virgo/src/fri.cpp:36-139 computed it there).
void vpi_oracle_committed(int oracle_indicator, int bit_len, const unsigned char root[32]);

// Optional evidence for the parity tests (environment, read once):
//   VPI_DUMP=<file>      every prover message in the golden transcript layout of SURVEY.md §8c
//   VPI_DUMP_FRI=<file>  per FRI step challenge[16] | root[32], then the final codeword (2048 F) and the mask codeword (32 F)
//                        — the layout of tests/golden/fri_*.bin (oracle/ref_driver.cpp)
//   VPI_DUMP_FFT=<file>  the messages of fft_gkr (vp_fft_gkr's layout = the layout of tests/golden/fftgkr_*.bin)
//   VPI_TRACE=1          at exit: how many vp_* calls of each kind served the reference's verifier (stderr)

// The reference's verifier draws every challenge from glibc random() (lib/virgo/src/fieldElement.cpp:119-124,362-367) and its query positions
// from rand() (vpd_verifier.cpp:121) — one process-wide generator.  The ROCm runtime consumes draws of that same generator while the library
// sets itself up (measured: with the device calls unguarded the reference binary still verifies, but its challenges — hence every message
// after merkle_root_l — differ from the CPU reference's).  A drop-in must leave the caller's stream alone, so every vp_* call of the two
// forwarding files runs under this guard: the generator is switched to a private state for the duration of the call and switched back after.
struct vpi_rand_guard {
    char priv[128];
    char *caller_state;
    const char *what;
    vpi_rand_guard(const char *w);
    ~vpi_rand_guard();
};
FILE *vpi_dump_file();
FILE *vpi_dump_fri_file();
FILE *vpi_dump_fft_file();
struct vpi_counters { unsigned long commit_private, commit_public, fri_step, fri_final, open_init, open_step, round, finalize, rand_consumers, fft_gkr; };
extern vpi_counters g_vpi_count;
// VPI_TRACE: wall seconds spent inside the device calls of each kind (what the reference adds up into its prove times)
struct vpi_seconds { double commit_private, commit_public, fri_step, fri_final, fft_gkr, first_fri_step; };
extern vpi_seconds g_vpi_sec;
struct vpi_stopwatch { double *acc; std::chrono::high_resolution_clock::time_point t0; explicit vpi_stopwatch(double *a) : acc(a), t0(std::chrono::high_resolution_clock::now()) {}
                       ~vpi_stopwatch() { *acc += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); } };
