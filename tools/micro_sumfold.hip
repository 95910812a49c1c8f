// Development harness (not part of the product): k_sumfold<3> vs k_sumfold3b on synthetic tables — same SfArgs, outputs
// compared element by element, each timed with HIP events.   tools/_build/micro_sumfold [log2 len] [n_tab] [valid]
#include "../virgo-plus_amd/csrc/vp_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
using namespace vp;
template <bool HAS_A> __global__ void __launch_bounds__(VP_BLOCK, 4) k_sumfold3b_nr(SfArgs a) {      // second instance of the same body (slot for A/B variants)
    __shared__ Sf3bLds sm;
    sumfold3b_body<HAS_A>(a, blockIdx.x, gridDim.x, sm, GenLoad());
}
static u64 rng_state = 88172645463325252ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static F rndF() { return f_make(rnd() % P61, rnd() % P61); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int n_tab = argc > 2 ? atoi(argv[2]) : 1;
    const u32 len = 1u << lg;
    const u32 valid = argc > 3 ? (u32) atoi(argv[3]) : len;
    const size_t tot = (size_t) len * n_tab;
    std::vector<F> hV(tot), hM(tot), hA(tot);
    for (size_t i = 0; i < tot; ++i) { hV[i] = rndF(); hM[i] = rndF(); hA[i] = rndF(); }
    hV[5] = f_make(P61 - 1, 0); hM[5] = f_make(P61 - 1, P61 - 1); hV[4] = f_zero(); hM[4] = f_zero();   // extremes
    F hr[3] = {rndF(), f_make(P61 - 1, P61 - 1), rndF()};
    F *dV, *dM, *dA, *oV[3], *oM[3], *oA[3], *dr, *part[3];
    CK(hipMalloc(&dV, tot * 16)); CK(hipMalloc(&dM, tot * 16)); CK(hipMalloc(&dA, tot * 16)); CK(hipMalloc(&dr, 48));
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&oV[k], tot * 16)); CK(hipMalloc(&oM[k], tot * 16)); CK(hipMalloc(&oA[k], tot * 16)); CK(hipMalloc(&part[k], 3 * 2048 * 3 * 16)); }
    CK(hipMemcpy(dV, hV.data(), tot * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, hM.data(), tot * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), tot * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr, 48, hipMemcpyHostToDevice));
    SfArgs a{};
    a.inV = dV; a.inM = dM; a.inA = dA; a.r = dr; a.has_a = 1; a.n_tab = n_tab;
    u32 chunks = 0;
    for (int j = 0; j < n_tab; ++j) { a.t[j].off = j * len; a.t[j].len = len; a.t[j].valid = valid; a.t[j].chunk_start = chunks; chunks += (valid + 511) / 512; }
    a.total_chunks = chunks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double) n_tab * (valid + ((valid + 7) >> 3)) * 48.0;
    u32 grids[3];
    for (int k = 0; k < 3; ++k) {
        const u32 grid = k == 0 ? std::max<u32>(1, std::min<u32>((chunks + 3) / 4, 2048)) : std::max<u32>(1, std::min<u32>(chunks, argc > 4 ? atoi(argv[4]) : 2048));
        grids[k] = grid;
        a.outV = oV[k]; a.outM = oM[k]; a.outA = oA[k]; a.part = part[k]; a.part_stride = grid * 3;
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL((k_sumfold<3, 1>), dim3(grid), dim3(256), 0, 0, a);
            else if (k == 1) hipLaunchKernelGGL(k_sumfold3b<true>, dim3(grid), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL(k_sumfold3b_nr<true>, dim3(grid), dim3(256), 0, 0, a);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = std::min(best, ms);
        }
        printf("%-14s len 2^%d x%d valid %u grid %u: %.1f us  %.0f GB/s algorithmic\n", k == 1 ? "k_sumfold3b" : k == 2 ? "k_sumfold3b_nr" : "k_sumfold<3>", lg, n_tab, valid, grid, best * 1e3, bytes / (best * 1e-3) * 1e-9);
    }
    // compare
    const size_t no = tot;
    std::vector<F> x(no), y(no);
    int bad = 0;
    F **outs[3] = {oV, oM, oA};
    for (int other = 1; other < 3; ++other)
    for (int tb = 0; tb < 3; ++tb) {
        CK(hipMemcpy(x.data(), outs[tb][0], no * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), outs[tb][other], no * 16, hipMemcpyDeviceToHost));
        for (int j = 0; j < n_tab; ++j) for (u32 i = 0; i < ((valid + 7) >> 3); ++i) { size_t q = (size_t) j * len + i; if (!f_eq(x[q], y[q])) { if (bad < 5) printf("table %d mismatch at %zu\n", tb, q); ++bad; } }
    }
    for (int s = 0; s < 3; ++s) for (int c = 0; c < 3; ++c) {
        F sum[3];
        for (int k = 0; k < 3; ++k) {
            std::vector<F> p(grids[k] * 3);
            CK(hipMemcpy(p.data(), part[k] + (size_t) s * grids[k] * 3, p.size() * 16, hipMemcpyDeviceToHost));
            F t = f_zero(); for (u32 b = 0; b < grids[k]; ++b) t = f_add(t, p[b * 3 + c]);
            sum[k] = t;
        }
        if (!f_eq(sum[0], sum[1]) || !f_eq(sum[0], sum[2])) { printf("round %d coef %d mismatch\n", s, c); ++bad; }
    }
    printf(bad ? "MISMATCH (%d)\n" : "outputs identical\n", bad);
    return bad != 0;
}
