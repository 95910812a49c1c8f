// Development harness (not part of the product): k_sumfold<3> vs k_sumfold3b on synthetic tables — same SfArgs, outputs
// compared element by element, each timed with HIP events.   tools/_build/micro_sumfold [log2 len] [n_tab] [valid]
#include "../virgo-plus_amd/csrc/vp_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
namespace vp {
// ---------------------------------------------------------------------------------------------------
// k_sumfoldr: the same three rounds per launch, ROLE-SPLIT for occupancy.
//
// Measured (rocprofv3 SQ counters, profiles/r01_l_pmc_summary_b*.json): k_sumfold3b and its generating variant retire one
// wave-instruction per ~5.9 cycles per SIMD, while the same multiply code runs at ~3.0 cycles per instruction with 8 waves per
// SIMD (tools/micro_rates.hip): at 116-140 VGPRs only 3-4 waves fit, and a wave that does a whole pair (six multiply-adds on six
// table entries) cannot be squeezed much further.  Here a pair is split over three waves with wave-uniform roles, two
// multiply-adds each, as in k_seg:
//     role 0:  X += (m1-m0)(v1-v0)          V' = v0 + r (v1-v0)
//     role 1:  Z += m0 v0 + a0              M' = m0 + r (m1-m0)
//     role 2:  Y += m1 v1 + a1              A' = a0 + r (a1-a0)
// A 768-thread workgroup (4 groups x 3 roles x 64 lanes) stages a 512-entry chunk of the three tables in LDS with coalesced
// 16-byte loads, runs round k on all 256 pair slots, round k+1 on slots 0..127, round k+2 on slots 0..63 (LDS in between, one
// barrier per round), and keeps ONE lazy accumulator per round in registers.  ~45 KB of LDS and <= 84 VGPRs: two workgroups = 24
// waves per CU.  Same SfArgs, same block-partial layout (X, Y-X-Z, Z per round) and bit-identical outputs as k_sumfold3b.
// ---------------------------------------------------------------------------------------------------
#define VP_SFR_THREADS 768          // experiment kept out of the product: see the note above sumfold3b_body in vp_kernels_batch.h
struct SfrLds { F in[3][512]; F s1[3][256]; F s2[3][128]; F red[12][3]; };
template <bool HAS_A>
__device__ __forceinline__ void sumfoldr_body(const SfArgs &a, u32 bid, u32 nb, SfrLds &sm) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int role = __builtin_amdgcn_readfirstlane(w % 3);
    const u32 slot = (u32) ((w / 3) * 64 + lane);                   // pair slot 0..255
    Lz acc0{0, 0}, acc1{0, 0}, acc2{0, 0};                         // this role's sum of rounds k, k+1, k+2
    const F r0 = a.r[0], r1 = a.r[1], r2 = a.r[2];
    const F *gin[3] = {a.inV, a.inM, a.inA};
    // one pair of this role from three table rows in LDS (row stride `str`): q = product (+ addend), o = folded entry
    auto task = [&](const F *src, u32 str, u32 p, const F &r, Lz &acc, F &o) {
        if (role == 0) {
            const F v0 = src[2 * p], v1 = src[2 * p + 1], m0 = src[str + 2 * p], m1 = src[str + 2 * p + 1];
            const F dv = f_sub_lazy(v1, v0);
            lz_add(acc, f_mad_lazy<true>(f_sub_lazy(m1, m0), dv, f_zero()));
            o = f_mad_c(r, dv, v0);
        } else if (role == 1) {
            const F v0 = src[2 * p], m0 = src[str + 2 * p], m1 = src[str + 2 * p + 1];
            const F a0 = HAS_A ? src[2 * str + 2 * p] : f_zero();
            lz_add(acc, f_mad_c<true>(m0, v0, a0));
            o = f_mad_c(r, f_sub_lazy(m1, m0), m0);
        } else {
            const F v1 = src[2 * p + 1], m1 = src[str + 2 * p + 1];
            F a0 = f_zero(), a1 = f_zero();
            if (HAS_A) { a0 = src[2 * str + 2 * p]; a1 = src[2 * str + 2 * p + 1]; }
            lz_add(acc, f_mad_c<true>(m1, v1, a1));
            o = HAS_A ? f_mad_c(r, f_sub_lazy(a1, a0), a0) : f_zero();
        }
        lz_fold(acc);
    };
    for (u32 c = bid; c < a.total_chunks; c += nb) {
        int j = 0;
        while (j + 1 < a.n_tab && c >= a.t[j + 1].chunk_start) ++j;
        const SfTab td = a.t[j];
        const u32 cl = c - td.chunk_start;
        const u32 base = td.off + cl * 512, vend = td.off + td.valid;
        // stage: 1536 entries, two per thread, 1 KiB per wave instruction
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32 e = (u32) t + 768u * q;                       // 0..1535
            const u32 tab = e >> 9, i = e & 511;
            if (HAS_A || tab < 2) (&sm.in[0][0])[e] = ld_or_zero(gin[tab], base + i, vend);
        }
        __syncthreads();
        F o;
        task(&sm.in[0][0], 512, slot, r0, acc0, o);               // round k: 256 pairs
        if (HAS_A || role < 2) sm.s1[role][slot] = o;
        __syncthreads();
        if (slot < 128) {                                          // round k+1 (wave-uniform: groups 0-1)
            task(&sm.s1[0][0], 256, slot, r1, acc1, o);
            if (HAS_A || role < 2) sm.s2[role][slot] = o;
        }
        __syncthreads();
        if (slot < 64) {                                           // round k+2 (group 0): the folded table
            task(&sm.s2[0][0], 128, slot, r2, acc2, o);
            const u32 oi = cl * 64 + slot;
            if (oi < ((td.valid + 7) >> 3)) {
                if (role == 0) a.outV[td.off + oi] = o;
                else if (role == 1) a.outM[td.off + oi] = o;
                else if (HAS_A) a.outA[td.off + oi] = o;
            }
        }
    }
    // block partials: red[wave][round]; role of wave w is w % 3
    {
        const F x0 = wave_sum63(lz_canon(acc0)), x1 = wave_sum63(lz_canon(acc1)), x2 = wave_sum63(lz_canon(acc2));
        if (lane == 63) { sm.red[w][0] = x0; sm.red[w][1] = x1; sm.red[w][2] = x2; }
    }
    __syncthreads();
    if (t < 3) {                                                   // thread t: round k + t
        F S[3];
        for (int ro = 0; ro < 3; ++ro) { F q = f_zero(); for (int g = 0; g < 4; ++g) q = f_add(q, sm.red[g * 3 + ro][t]); S[ro] = q; }
        const F X = S[0], Z = S[1], Y = S[2];
        F *o = a.part + (size_t) t * a.part_stride + bid * 3;
        o[0] = X; o[1] = f_sub(f_sub(Y, X), Z); o[2] = Z;
    }
}
template <bool HAS_A>
__global__ void __launch_bounds__(VP_SFR_THREADS, 2) k_sumfoldr(SfArgs a) {
    __shared__ SfrLds sm;
    sumfoldr_body<HAS_A>(a, blockIdx.x, gridDim.x, sm);
}

}  // namespace vp
using namespace vp;
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
    const bool real_v = getenv("REAL_V") && atoi(getenv("REAL_V"));      // REAL_V=1: V entries real, a fourth variant takes the real-V pair step in round k
    for (size_t i = 0; i < tot; ++i) { hV[i] = rndF(); hM[i] = rndF(); hA[i] = rndF(); if (real_v) hV[i].im = 0; }
    hV[5] = f_make(P61 - 1, 0); hM[5] = f_make(P61 - 1, P61 - 1); hV[4] = f_zero(); hM[4] = f_zero();   // extremes
    F hr[3] = {rndF(), f_make(P61 - 1, P61 - 1), rndF()};
    const int NV = real_v ? 4 : 3;
    F *dV, *dM, *dA, *oV[4], *oM[4], *oA[4], *dr, *part[4];
    CK(hipMalloc(&dV, tot * 16)); CK(hipMalloc(&dM, tot * 16)); CK(hipMalloc(&dA, tot * 16)); CK(hipMalloc(&dr, 48));
    for (int k = 0; k < NV; ++k) { CK(hipMalloc(&oV[k], tot * 16)); CK(hipMalloc(&oM[k], tot * 16)); CK(hipMalloc(&oA[k], tot * 16)); CK(hipMalloc(&part[k], 3 * 2048 * 3 * 16)); }
    CK(hipMemcpy(dV, hV.data(), tot * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, hM.data(), tot * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), tot * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr, 48, hipMemcpyHostToDevice));
    SfArgs a{};
    a.inV = dV; a.inM = dM; a.inA = dA; a.r = dr; a.has_a = 1; a.n_tab = n_tab; a.keep_y0 = getenv("KEEP_Y") ? atoi(getenv("KEEP_Y")) : 3;
    u32 chunks = 0;
    for (int j = 0; j < n_tab; ++j) { a.t[j].off = j * len; a.t[j].len = len; a.t[j].valid = valid; a.t[j].chunk_start = chunks; chunks += (valid + 511) / 512; }
    a.total_chunks = chunks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double) n_tab * (valid + ((valid + 7) >> 3)) * 48.0;
    u32 grids[4];
    const int ky = a.keep_y0;
    for (int k = 0; k < NV; ++k) {
        a.keep_y0 = ky | (k == 3 ? 4 : 0);
        const u32 grid = k == 0 ? std::max<u32>(1, std::min<u32>((chunks + 3) / 4, 2048)) : std::max<u32>(1, std::min<u32>(chunks, argc > 4 ? atoi(argv[4]) : 2048));
        grids[k] = grid;
        a.outV = oV[k]; a.outM = oM[k]; a.outA = oA[k]; a.part = part[k]; a.part_stride = grid * 3;
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL((k_sumfold<3, 1>), dim3(grid), dim3(256), 0, 0, a);
            else if (k == 1 || k == 3) hipLaunchKernelGGL(k_sumfold3b<true>, dim3(grid), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL(k_sumfoldr<true>, dim3(grid), dim3(VP_SFR_THREADS), 0, 0, a);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = std::min(best, ms);
        }
        printf("%-14s len 2^%d x%d valid %u grid %u: %.1f us  %.0f GB/s algorithmic\n", k == 1 ? "k_sumfold3b" : k == 2 ? "k_sumfoldr" : k == 3 ? "k_sumfold3b realV" : "k_sumfold<3>", lg, n_tab, valid, grid, best * 1e3, bytes / (best * 1e-3) * 1e-9);
    }
    // compare
    const size_t no = tot;
    std::vector<F> x(no), y(no);
    int bad = 0;
    F **outs[3] = {oV, oM, oA};
    for (int other = 1; other < NV; ++other)
    for (int tb = 0; tb < 3; ++tb) {
        CK(hipMemcpy(x.data(), outs[tb][0], no * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), outs[tb][other], no * 16, hipMemcpyDeviceToHost));
        for (int j = 0; j < n_tab; ++j) for (u32 i = 0; i < ((valid + 7) >> 3); ++i) { size_t q = (size_t) j * len + i; if (!f_eq(x[q], y[q])) { if (bad < 5) printf("table %d mismatch at %zu\n", tb, q); ++bad; } }
    }
    for (int s = 0; s < 3; ++s) for (int c = 0; c < 3; ++c) {
        F sum[4];
        for (int k = 0; k < NV; ++k) {
            std::vector<F> p(grids[k] * 3);
            CK(hipMemcpy(p.data(), part[k] + (size_t) s * grids[k] * 3, p.size() * 16, hipMemcpyDeviceToHost));
            F t = f_zero(); for (u32 b = 0; b < grids[k]; ++b) t = f_add(t, p[b * 3 + c]);
            sum[k] = t;
        }
        if (!f_eq(sum[0], sum[1]) || !f_eq(sum[0], sum[2]) || (NV == 4 && !f_eq(sum[0], sum[3]))) { printf("round %d coef %d mismatch\n", s, c); ++bad; }
    }
    printf(bad ? "MISMATCH (%d)\n" : "outputs identical\n", bad);
    return bad != 0;
}
