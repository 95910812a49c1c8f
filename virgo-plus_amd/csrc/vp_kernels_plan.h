// Batched path, part 2: segment kernel k_seg, closing kernel k_emit, and the batched ("plan") launches that run one kernel body for many independent sumchecks.
// Part of the single translation unit vpgpu.hip (see vp_kernels.h for the overall layout rules).
#pragma once
#include "vp_kernels_batch.h"

// ===================================================================================================
// Segment kernels (default batched path).
//
// The cost of this path is integer ALU, not bytes: one F-multiply is ~75 VALU instructions (12 of them
// v_mad_u64_u32), a lone wave issues one instruction every ~4 cycles, so the dependent chain of a round
// — not the 288 B per pair — sets the time of every table that does not fill the chip.  Hence:
//   * k_seg: a workgroup stages a SEGMENT of <= 1024 consecutive entries of V/mult/add in LDS (coalesced
//     1 KiB wave loads) and runs log2(segment) rounds on it without leaving the CU; ten rounds cost
//     48 B/entry of HBM reads and 48 B per 1024 entries of writes.  The challenges are on the tape, so only
//     the fold x0 + r (x1 - x0) is serial: the chain of folds runs first, one multiplication per lane and
//     round with every level kept in LDS, then the three products of all pairs of all levels in one pass
//     (seg_body below).
//   * k_emit: one workgroup finishes the sumcheck: it owns every table that is down to <= 2^e entries
//     (LDS resident, one wave per table, same two passes), adds the block partials of the fold / k_seg
//     launches, retires finished tables into add_term and writes all round polynomials and the claims.
// ===================================================================================================
namespace vp {

#ifndef VP_SEG_LOG
#define VP_SEG_LOG 10           // 1024-entry segments (120 KB of LDS, one workgroup per CU); 9 = 512 entries, two per CU, was measured: no gain
#endif
#define VP_SEG (1 << VP_SEG_LOG)
#define VP_SEG_THREADS 768          // 12 waves

struct SegTab {
    u32 off;          // table offset (same in input and output buffers)
    u32 valid;        // valid entries of the input table
    u32 seg_start;    // first global segment index of this table
    int seg_log;      // log2(segment length) = rounds performed on this table by the launch
};
struct SegArgs {
    const F *inV, *inM, *inA;
    F *outV, *outM, *outA;
    const F *r;               // r[s] = challenge of the s-th round of this launch
    F *part;                  // part[s * part_stride + block * 3 + c]
    u32 part_stride;
    u32 total_segs;
    int n_tab, n_rounds;      // n_rounds = max seg_log
    int has_a; u32 nblk;      // batched launches: table family has an add array; blocks given to this job
    SegTab t[VP_MAX_TAB];
};

// A segment of S = 2^R entries is closed in two passes (the challenges are on the tape):
//   (1) the fold chain, the only serial part: level s+1 = level s folded with r_s — ONE multiplication per lane and round (the V, mult
//       and add folds of a pair are three tasks), a workgroup barrier per round while a round has more than 64 tasks, then wave 0 alone;
//       every level is kept: table t lives in a pyramid of 2 S slots, level s at offset 2 S - (2 S >> s);
//   (2) the three products of ALL S - 1 pairs of the pyramid, pair i (level R-1-floor(log2 i), index i - 2^floor(log2 i)) on lane i:
//       blocks of 64 pairs from i = 64 on lie inside one level (a wave sum), the first block holds levels of 1..32 pairs (butterfly
//       inside each aligned lane block).  Every wave keeps its round sums (wsum[round][wave]) across all the segments it works on.
// Round by round with wave-uniform roles and a barrier per round (the previous form) a 1024-entry segment took ~20 us, most of it the
// latency of rounds with fewer pairs than lanes; the chain costs ~0.4 us per round.
struct SegLds { F pyr[3][2 * VP_SEG]; F wsum[VP_SEG_LOG][VP_SEG_THREADS / 64][3]; F rr[VP_SEG_LOG + 1]; };
template <bool HAS_A>
__device__ __forceinline__ void seg_body(const SegArgs &a, u32 bid, u32 nb, SegLds &sm) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    constexpr int NW = VP_SEG_THREADS / 64;
    constexpr u32 NARR = HAS_A ? 3u : 2u;
    for (int i = tid; i < VP_SEG_LOG * NW * 3; i += VP_SEG_THREADS) (&sm.wsum[0][0][0])[i] = f_zero();
    if (tid < a.n_rounds) sm.rr[tid] = a.r[tid];          // the challenges of this launch: one LDS read per round instead of a global load on the round's critical path
    __syncthreads();

    for (u32 seg = bid; seg < a.total_segs; seg += nb) {
        int j = 0;
        while (j + 1 < a.n_tab && seg >= a.t[j + 1].seg_start) ++j;
        const SegTab td = a.t[j];
        const u32 q = seg - td.seg_start;
        const int R = td.seg_log;
        const u32 S = 1u << R;
        const u32 base = td.off + q * S;
        const u32 vseg = td.valid > q * S ? min(td.valid - q * S, S) : 0;     // valid entries of this segment
        for (u32 i = tid; i < S; i += VP_SEG_THREADS) {
            const bool ok = i < vseg;
            sm.pyr[0][i] = ok ? a.inV[base + i] : f_zero();
            sm.pyr[1][i] = ok ? a.inM[base + i] : f_zero();
            if (HAS_A) sm.pyr[2][i] = ok ? a.inA[base + i] : f_zero();
        }
        __syncthreads();
        // ---- (1) the chain ----
#pragma unroll 1
        for (int s = 0; s < R; ++s) {
            const u32 n = S >> (s + 1);                                   // pairs of this round
            const int lg = R - s - 1;
            const u32 so = 2 * S - ((2 * S) >> s), dof = 2 * S - ((2 * S) >> (s + 1));
            const bool wide = NARR * n > 64;
            if (wide || wu == 0) {
                const F rs = sm.rr[s];
                for (u32 t = (u32) tid; t < NARR * n; t += VP_SEG_THREADS) {
                    F *tb = sm.pyr[t >> lg];
                    const u32 p = t & (n - 1);
                    const F x0 = tb[so + 2 * p], x1 = tb[so + 2 * p + 1];
                    tb[dof + p] = f_add(x0, f_mul(rs, f_sub(x1, x0)));
                }
            }
            if (wide) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }      // wave 0 finishes the chain alone
        }
        __syncthreads();
        // the segment is down to one entry per table
        if (tid < 3 && q * S < td.valid) {
            const u32 lo = 2 * S - 2;
            if (tid == 0) a.outV[td.off + q] = sm.pyr[0][lo];
            else if (tid == 1) a.outM[td.off + q] = sm.pyr[1][lo];
            else if (HAS_A) a.outA[td.off + q] = sm.pyr[2][lo];
        }
        // ---- (2) the products of every level ----
        for (u32 i0 = 64 * (u32) wu; i0 < S; i0 += VP_SEG_THREADS) {       // wave-uniform: blocks of 64 pair slots
            const u32 i = i0 + lane;
            const int e = 31 - __builtin_clz(i | 1u);
            const int s = R - 1 - e;
            F ca = f_zero(), cbv = f_zero(), cc = f_zero();
            if (i >= 1 && i < S) {
                const u32 p = i - (1u << e);
                const u32 so = 2 * S - ((2 * S) >> s);
                const F m0 = sm.pyr[1][so + 2 * p], m1 = sm.pyr[1][so + 2 * p + 1], v0 = sm.pyr[0][so + 2 * p], v1 = sm.pyr[0][so + 2 * p + 1];
                const F qa = f_mul(f_sub(m1, m0), f_sub(v1, v0)), qc = f_mul(m0, v0), y = f_mul(m1, v1);
                ca = qa; cc = qc; cbv = f_sub(f_sub(y, qa), qc);
                if (HAS_A) { const F a0 = sm.pyr[2][so + 2 * p], a1 = sm.pyr[2][so + 2 * p + 1]; cc = f_add(cc, a0); cbv = f_add(cbv, f_sub(a1, a0)); }
            }
            if (i0 >= 64) {                                               // one level: pairs [i0, i0 + 64) of level floor(log2 i0)
                ca = wave_sum63(ca); cbv = wave_sum63(cbv); cc = wave_sum63(cc);
                if (lane == 63) {
                    F *o = sm.wsum[s][wu];
                    o[0] = f_add(o[0], ca); o[1] = f_add(o[1], cbv); o[2] = f_add(o[2], cc);
                }
            } else {                                                      // levels of 1, 2, .. 32 pairs: lane block [2^e, 2^(e+1))
                u64 acc[6] = {ca.re, ca.im, cbv.re, cbv.im, cc.re, cc.im};
                const int steps = min(R - 1, 5);
#pragma unroll 1
                for (int st = 0; st < steps; ++st) {                      // at most 8 canonical limbs are added before a fold
                    const bool take = (1 << st) < (1 << e);
#pragma unroll
                    for (int c = 0; c < 6; ++c) { const u64 o = __shfl_xor(acc[c], 1 << st, 64); acc[c] += take ? o : 0ull; }
                    if (st == 2) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) acc[c] = m_fold(acc[c]);
                    }
                }
                if (i >= 1 && i < S && i == (1u << e)) {
                    F *o = sm.wsum[s][wu];
                    o[0] = f_add(o[0], f_make(m_fold(acc[0]), m_fold(acc[1])));
                    o[1] = f_add(o[1], f_make(m_fold(acc[2]), m_fold(acc[3])));
                    o[2] = f_add(o[2], f_make(m_fold(acc[4]), m_fold(acc[5])));
                }
            }
        }
        __syncthreads();
    }
    // per-round block partials, as the reference orders the coefficients: a = sum dm dv, b = sum (m1 v1 + da) - a - sum m0 v0, c = sum (m0 v0 + a0)
    if (tid < 3 * a.n_rounds) {
        const int s = tid / 3, c = tid % 3;
        F x = f_zero();
        for (int q = 0; q < NW; ++q) x = f_add(x, sm.wsum[s][q][c]);
        a.part[(size_t) s * a.part_stride + bid * 3 + c] = x;
    }
}
template <bool HAS_A>
__global__ void __launch_bounds__(VP_SEG_THREADS) k_seg(SegArgs a) {
    __shared__ SegLds sm;
    seg_body<HAS_A>(a, blockIdx.x, gridDim.x, sm);
}

// ---------------------------------------------------------------------------------------------------
// k_emit: one workgroup closes a sumcheck.
//   phase 0/1  one wave per table fetches it into LDS while the same wave adds up its share of the block partials the fold and
//              k_seg launches wrote (one round per wave, contiguous reads);
//   phase 2    a table per wave, no workgroup barrier: fold chain first, then the products of all its levels in one pass;
//   phase 3    totals per (round, coefficient) in parallel; the add_term recurrence (src/prover.cpp:445, 462-467) and the
//              recurrence that restores the b's the fold launches left out, both as affine wave scans; polynomials and claims
//              written out by parallel lanes.
// ---------------------------------------------------------------------------------------------------
#define VP_EMIT_THREADS 768         // 12 waves
#define VP_EMIT_WAVES (VP_EMIT_THREADS / 64)
#define VP_MAX_PD 24
#define VP_EMIT_CAP 1280            // LDS entries per buffer per table family (2 x 3 x 1280 x 16 B = 120 KiB)
struct EmitTab {
    u32 off;          // offset in the global buffers
    int enter;        // first round (1-based) handled here
    u32 len_enter;    // logical length at `enter` (<= 2^emit_log)
    u32 valid_enter;  // valid entries at `enter`
    int src;          // global buffer holding the table at `enter` (tab[src]); V from V0 if v_from_v0
    int v_from_v0;
    int bl;           // log2 of the table's length at round 1
    int exp;          // index-split proof (vp_set_shard_split): 1 + export slot; the table is a SLICE of a longer one, and when it is down to one entry
                      // its (v, m, a) go to exp_out[3 * slot ..] instead of retiring into add_term — the rounds that pair it with the other
                      // ranks' slices are finished on the host from the gathered entries (vpgpu_batched.inc, split_finish).  0: ordinary table
};
struct EmitArgs {
    const F *V0;
    const F *buf[2][3];
    const F *r;                 // r[k-1] = challenge of round k
    const F *part; u32 part_stride;
    int n_tab, rounds, has_a, emit_log;
    u32 work_mask;              // bit k-1: some table of this kernel has pairs or retires in round k
    u32 enter_mask;             // bit k-1: some table is loaded from global memory in round k
    u32 pair_mask;              // bit k-1: some table has PAIRS in round k (work_mask minus the retire-only rounds)
    u32 derive_mask;            // bit k-1: a fold launch left out the sum of m1 v1 + a1 in round k; b_k = S_{k-1}(r_{k-1}) - a_k - 2 c_k
    int derive_later;           // 1: k_fixup derives those b's after all sumchecks are closed (round 1 included: its claim comes from another sumcheck)
    F *poly_out, *claims_out, *Vu;
    F *exp_out;                 // export area of the index-split proof (EmitTab::exp)
    int n_pd;                   // launches that left block partials
    struct { int k0, nr; u32 nblk, off; } pd[VP_MAX_PD];   // rounds k0..k0+nr-1: part[off + s*nblk*3 + b*3 + c]
    EmitTab t[VP_MAX_TAB];
};

// x_k = A_k x_{k-1} + B_k over lanes 0..31 of a wave (x_{-1} = 0): on return B holds x_k.  Five steps of two independent multiplies.
__device__ __forceinline__ void affine_scan32(F &A, F &B, int lane) {
#pragma unroll 1
    for (int d = 1; d < 32; d <<= 1) {
        F Ap, Bp;
        Ap.re = __shfl_up(A.re, d, 64); Ap.im = __shfl_up(A.im, d, 64); Bp.re = __shfl_up(B.re, d, 64); Bp.im = __shfl_up(B.im, d, 64);
        if (lane >= d) { B = f_add(f_mul(A, Bp), B); A = f_mul(A, Ap); }
    }
}
// dynamic LDS: tables [2][3][cap] | psum[32][3] | wred[32][12][3] | claim[64] | retv[64] | atv[32] | r[32] | retk[64] (int)
#define VP_EMIT_LDS_EXTRA_F (32 * 3 + 32 * VP_EMIT_WAVES * 3 + VP_MAX_TAB + VP_MAX_TAB + 32 + 32)
__device__ __forceinline__ void emit_body(const EmitArgs &a, unsigned char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nth = blockDim.x;
    const u32 E = 1u << a.emit_log, cap = (u32) a.n_tab * E;
    F *lbuf = reinterpret_cast<F *>(smem_raw);
    F *psum = lbuf + (size_t) 6 * cap;
    F *wred = psum + 32 * 3;
    F *s_claim = wred + 32 * VP_EMIT_WAVES * 3;
    F *s_retv = s_claim + VP_MAX_TAB;
    F *s_at = s_retv + VP_MAX_TAB;
    F *s_r = s_at + 32;                                   // the challenges r[0..rounds), read once
    int *s_retk = reinterpret_cast<int *>(s_r + 32);
    // table j, array t (0: V, 1: mult, 2: add): a pyramid of 2E slots — the table as it enters at offset 0, its fold after s rounds at
    // offset 2 len - (2 len >> s), the last single entry at 2 len - 2
    auto T = [&](int t, int j) { return lbuf + ((size_t) (t * a.n_tab + j)) * 2 * E; };
    if (tid < VP_MAX_TAB) { s_claim[tid] = f_zero(); s_retv[tid] = f_zero(); s_retk[tid] = 0; }
    for (int i = tid; i < 32 * VP_EMIT_WAVES * 3; i += nth) wred[i] = f_zero();      // per-round, per-wave sums of phase 2
    if (tid >= 64 && tid - 64 < a.rounds) s_r[tid - 64] = a.r[tid - 64];
    // ---- phase 0: every table this kernel owns is fetched into its LDS slot NOW (buffer = parity of its first round here).
    // The tables are complete when the kernel starts and their slots are untouched until that round.  One table per wave, one entry
    // per lane (emit_log <= 6); the loads of a wave's first table stay in flight while it adds up its share of phase 1, so that the
    // closing launch starts with ONE memory round trip instead of one per table and one per 64 block partials. ----
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int nrounds = a.rounds > 0 ? a.rounds : 1;
    auto tab_fetch = [&](int j, F &v, F &m, F &ad) -> int {        // -> LDS buffer of the table's first round, -1: not folded here
        const EmitTab td = a.t[j];
        v = f_zero(); m = f_zero(); ad = f_zero();
        if (td.enter > nrounds) return -1;
        const F *gV = td.v_from_v0 ? a.V0 + td.off : a.buf[td.src][0] + td.off;
        const F *gM = a.buf[td.src][1] + td.off, *gA = a.buf[td.src][2] + td.off;
        const bool single = td.bl == 0 && td.valid_enter > 0;        // always-initialised single entry (an EMPTY one-entry table is all zero: empty subsets, and the placeholders of an index-split proof)
        if ((u32) lane < td.len_enter && (single || (u32) lane < td.valid_enter)) { v = gV[lane]; m = gM[lane]; if (a.has_a) ad = gA[lane]; }
        return td.enter & 1;
    };
    auto tab_store = [&](int j, int cbj, const F &v, const F &m, const F &ad) {
        if (cbj >= 0 && (u32) lane < E) { T(0, j)[lane] = v; T(1, j)[lane] = m; T(2, j)[lane] = ad; }
    };
    F t0v = f_zero(), t0m = f_zero(), t0a = f_zero();
    const int t0cb = wu < a.n_tab ? tab_fetch(wu, t0v, t0m, t0a) : -1;
    // ---- phase 1: block partials of the fold / k_seg launches, one round per wave, 256 blocks per memory round trip.  The launch
    // descriptors come with ONE vector load (lane d holds pd[d]): read as scalars in the loop they cost a scalar-cache miss each. ----
    int pd_k0 = 0, pd_nr = 0; u32 pd_nb = 0, pd_off = 0;
    if (lane < a.n_pd) { pd_k0 = a.pd[lane].k0; pd_nr = a.pd[lane].nr; pd_nb = a.pd[lane].nblk; pd_off = a.pd[lane].off; }
    const int n_pd = a.n_pd;
    const F *part = a.part;
    for (int k = wu + 1; k <= a.rounds; k += VP_EMIT_WAVES) {
        F ca = f_zero(), cbv = f_zero(), cc = f_zero();
        for (int d = 0; d < n_pd; ++d) {
            const int k0 = __builtin_amdgcn_readlane(pd_k0, d), nr = __builtin_amdgcn_readlane(pd_nr, d);
            if (k < k0 || k >= k0 + nr) continue;
            const u32 nb = (u32) __builtin_amdgcn_readlane((int) pd_nb, d);
            const F *pp = part + (u32) __builtin_amdgcn_readlane((int) pd_off, d) + (size_t) (k - k0) * nb * 3;
            // the region of a round is nb x (a, b, c) = 3 nb consecutive elements: read as such (a lane taking element 3 i + c of block
            // i fetches every cache line three times — measured: 5-9 us for the first round of a 512-block launch).  64 = 1 mod 3, so
            // the t-th load of a lane always sees component (lane + t) mod 3: three accumulators, rotated back at the end.
            const u32 nF = 3 * nb;
            F A0 = f_zero(), A1 = f_zero(), A2 = f_zero();
            for (u32 e0 = 0; e0 < nF; e0 += 768) {
                F x[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) { const u32 e = e0 + 192 * u + 64 * t + lane; x[u][t] = e < nF ? pp[e] : f_zero(); }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { A0 = f_add(A0, x[u][0]); A1 = f_add(A1, x[u][1]); A2 = f_add(A2, x[u][2]); }
            }
            const int rot = lane % 3;
            ca = f_add(ca, rot == 0 ? A0 : rot == 1 ? A2 : A1);
            cbv = f_add(cbv, rot == 0 ? A1 : rot == 1 ? A0 : A2);
            cc = f_add(cc, rot == 0 ? A2 : rot == 1 ? A1 : A0);
        }
        ca = wave_sum63(ca); cbv = wave_sum63(cbv); cc = wave_sum63(cc);
        if (lane == 63) { psum[3 * (k - 1)] = ca; psum[3 * (k - 1) + 1] = cbv; psum[3 * (k - 1) + 2] = cc; }
    }
    tab_store(wu, t0cb, t0v, t0m, t0a);
    for (int j = wu + VP_EMIT_WAVES; j < a.n_tab; j += VP_EMIT_WAVES) {
        F v, m, ad;
        const int cbj = tab_fetch(j, v, m, ad);
        tab_store(j, cbj, v, m, ad);
    }
    __syncthreads();
    // ---- phase 2: the closing rounds.  A table is folded by ONE wave from the round it enters down to its last entry, with no
    // workgroup barrier in between: the challenges are on the tape, so tables meet only in the per-round sums, which every wave keeps
    // to itself (wred[round][wave]) until phase 3.  Per table:
    //   (1) the fold chain — the only serial part: level s+1 = level s folded with r_k, one multiplication per lane and round (the
    //       V, mult and add folds of a pair go to three lanes), every level kept (the pyramid above);
    //   (2) the products of ALL rounds at once — the len - 1 pairs of the pyramid, one per lane, three multiplications each — and a
    //       butterfly sum inside each level's aligned lane block (lane i: level S-1-floor(log2 i), pair i - 2^floor(log2 i)).
    // (Round by round behind a workgroup barrier this cost ~2 us per round, 40 us on a 23-round sumcheck; round by round inside one wave
    // 1.7 us per round, 10 us for a 64-entry table: the products and their sums are most of a round and none of the chain.) ----
    for (int j = wu; j < a.n_tab; j += VP_EMIT_WAVES) {
        const EmitTab td = a.t[j];
        if (td.enter > nrounds) continue;
        const u32 len0 = td.len_enter;
        const int S = 31 - __builtin_clz(len0 | 1u);
        F *tv = T(0, j), *tm = T(1, j), *ta = T(2, j);
        const u32 narr = a.has_a ? 3u : 2u;
        int s_done = 0;
        F retire_mv = f_zero();
        for (int sl = 0; sl < S; ++sl) {
            const int k = td.enter + sl;
            if (k > a.rounds) break;
            const u32 np = len0 >> (sl + 1);
            const int lg = S - sl - 1;                          // log2 np
            const u32 so = 2 * len0 - ((2 * len0) >> sl), dof = 2 * len0 - ((2 * len0) >> (sl + 1));
            const F rk = s_r[k - 1];
            for (u32 t = (u32) lane; t < narr * np; t += 64) {
                F *base = T((int) (t >> lg), j);
                const u32 pp = t & (np - 1);
                const F x0 = base[so + 2 * pp], x1 = base[so + 2 * pp + 1];
                base[dof + pp] = f_add(x0, f_mul(rk, f_sub(x1, x0)));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next round reads what other lanes of this wave just stored
            __builtin_amdgcn_wave_barrier();
            s_done = sl + 1;
        }
        if (S > 0) {
            const int e = 31 - __builtin_clz((u32) lane | 1u);
            const int sl = S - 1 - e;
            F ca = f_zero(), cbv = f_zero(), cc = f_zero();
            F y = f_zero();
            if (lane == 0 && s_done == S) y = f_mul(tm[2 * len0 - 2], tv[2 * len0 - 2]);        // idle in the pair pass: the retiring entry's m v
            if (lane >= 1 && (u32) lane < len0 && sl < s_done) {
                const u32 pp = (u32) lane - (1u << e);
                const u32 so = 2 * len0 - ((2 * len0) >> sl);
                const F m0 = tm[so + 2 * pp], m1 = tm[so + 2 * pp + 1], v0 = tv[so + 2 * pp], v1 = tv[so + 2 * pp + 1];
                const F qa = f_mul(f_sub(m1, m0), f_sub(v1, v0)), qc = f_mul(m0, v0);
                y = f_mul(m1, v1);
                ca = qa; cc = qc; cbv = f_sub(f_sub(y, qa), qc);
                if (a.has_a) { const F a0 = ta[so + 2 * pp], a1 = ta[so + 2 * pp + 1]; cc = f_add(cc, a0); cbv = f_add(cbv, f_sub(a1, a0)); }
            }
            retire_mv = y;
            u64 acc[6] = {ca.re, ca.im, cbv.re, cbv.im, cc.re, cc.im};
#pragma unroll 1
            for (int st = 0; st < S - 1; ++st) {               // the largest level has 2^(S-1) pairs; at most 8 canonical limbs are added before a fold
                const bool take = (1 << st) < (1 << e);
#pragma unroll
                for (int q = 0; q < 6; ++q) { const u64 o = __shfl_xor(acc[q], 1 << st, 64); acc[q] += take ? o : 0ull; }
                if (st == 2) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) acc[q] = m_fold(acc[q]);
                }
            }
            if (lane >= 1 && (u32) lane == (1u << e) && (u32) lane < len0 && sl < s_done) {
                F *o = wred + ((size_t) (td.enter + sl - 1) * VP_EMIT_WAVES + wu) * 3;
                o[0] = f_add(o[0], f_make(m_fold(acc[0]), m_fold(acc[1])));
                o[1] = f_add(o[1], f_make(m_fold(acc[2]), m_fold(acc[3])));
                o[2] = f_add(o[2], f_make(m_fold(acc[4]), m_fold(acc[5])));
            }
        }
        // the single entry left: it is the claim; in a real round it retires into add_term
        if (lane == 0 && s_done == S) {
            const u32 lo = 2 * len0 - 2;
            const int k = td.enter + S;
            const F v = tv[lo], m = tm[lo], ad = ta[lo];
            if (td.exp) { F *o = a.exp_out + 3 * (size_t) (td.exp - 1); o[0] = v; o[1] = m; o[2] = ad; }       // a slice: neither claim nor add_term here
            else {
                s_claim[j] = v;
                if (k <= a.rounds) { s_retv[j] = f_add(S > 0 ? retire_mv : f_mul(v, m), ad); s_retk[j] = k; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- phase 3 ----
    if (tid < a.rounds * 3) {
        const int k = tid / 3, c = tid % 3;
        F t = psum[3 * k + c];
        if ((a.pair_mask >> k) & 1u)
            for (int q = 0; q < VP_EMIT_WAVES; ++q) t = f_add(t, wred[((size_t) k * VP_EMIT_WAVES + q) * 3 + c]);
        psum[3 * k + c] = t;
    }
    if (w == VP_EMIT_WAVES - 1 && lane < a.rounds) {           // retire sum of round lane+1
        F t = f_zero();
        for (int j = 0; j < a.n_tab; ++j) if (s_retk[j] == lane + 1) t = f_add(t, s_retv[j]);
        s_at[lane] = t;
    }
    __syncthreads();
    if (w == 0) {                                              // add_term recurrence  at_k = (1 - r_{k-1}) at_{k-1} + retired_k
        const bool in = lane < a.rounds;
        F A = (in && lane >= 1) ? f_sub(f_one(), s_r[lane - 1]) : f_zero();
        F B = in ? s_at[lane] : f_zero();
        if (__any(!f_is_zero(B))) {                            // uniform; nothing retires in a sumcheck over tables of one length
            affine_scan32(A, B, lane);
            if (in) s_at[lane] = B;
        }
    }
    __syncthreads();
    if (tid < a.rounds * 3) {
        const int k = tid / 3, c = tid % 3;
        F t = psum[3 * k + c];
        if (c == 1) t = f_sub(t, s_at[k]); else if (c == 2) t = f_add(t, s_at[k]);
        psum[3 * k + c] = t;                                   // the round polynomial (a, b, c) as the reference sends it
    }
    if (a.derive_mask && !a.derive_later) {                    // uniform
        __syncthreads();
        // Rounds whose fold launches skipped the product sum: b_k from the verifier's identity S_k(0) + S_k(1) = S_{k-1}(r_{k-1})
        // (totals over every table of the phase, add_term included):  b_k = r b_{k-1} + (a_{k-1} r^2 + c_{k-1} - a_k - 2 c_k), r = r_{k-1}
        // — an affine recurrence in b (a known b restarts it), closed by a scan over the rounds instead of two multiplies per round in order.
        if (w == 0) {
            const int k = lane;
            const bool in = k < a.rounds, der = in && k >= 1 && ((a.derive_mask >> k) & 1u);
            F A = f_zero(), B = in ? psum[3 * k + 1] : f_zero();
            if (der) {
                const F r = s_r[k - 1];
                A = r;
                B = f_sub(f_sub(f_add(f_mul(f_mul(psum[3 * (k - 1)], r), r), psum[3 * (k - 1) + 2]), psum[3 * k]), f_dbl(psum[3 * k + 2]));
            }
            affine_scan32(A, B, lane);
            if (der) psum[3 * k + 1] = B;
        }
    }
    __syncthreads();
    if (tid < a.rounds * 3) a.poly_out[tid] = psum[tid];
    if (tid < a.n_tab) {
        F c = s_claim[tid];
        if (a.rounds > 0) {
            const EmitTab td = a.t[tid];
            if (td.bl == a.rounds) {
                // as long as the sumcheck: folded to one entry by the last round — here, or already by k_seg
                if (td.enter > a.rounds) c = td.valid_enter ? (a.buf[td.src][0] + td.off)[0] : f_zero();
                // else: phase 2 left it in s_claim
            }
        }
        a.claims_out[tid] = c;
        if (a.Vu && tid == 0) *a.Vu = c;
    }
}
__global__ void __launch_bounds__(VP_EMIT_THREADS) k_emit(EmitArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    emit_body(a, smem_raw);
}

// ---------------------------------------------------------------------------------------------------
// k_fixup: the b coefficients the fold launches left out, round 1 of every sumcheck included.  The claim a sumcheck starts
// from is the verifier's running `previousSum` (src/verifier.cpp): Vres for the first phase 1 (:151), vr of the layer above for
// the others (:336), the last phase-1 polynomial at its last challenge for phase 2 (:229), sum of sig * claims for Liu
// (:281-286).  Those inputs are final values of OTHER sumchecks that their closing launches produce directly (a last round is
// never a fold-launch round), so every sumcheck is fixed up independently: one lane each, two multiplies per derived round.
// ---------------------------------------------------------------------------------------------------
struct FixJob {
    u32 poly_pos, rounds, derive_mask, r_off;      // transcript index of the first polynomial; tape index of r[0]
    int kind;                                      // 0: claim = tr[ref0]   1: claim = poly at tr[ref0..ref0+3) evaluated at tape[ref1]   2: sum of tape[sig[t]] * tr[cl[t]]
    u32 ref0, ref1, n_terms;
    u32 sig[VP_MAX_TAB + 1], cl[VP_MAX_TAB + 1];
};
__global__ void __launch_bounds__(64) k_fixup(const FixJob *__restrict__ jobs, u32 n_jobs, const F *__restrict__ tape, F *__restrict__ tr) {
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_jobs) return;
    const FixJob &j = jobs[q];
    if (!j.derive_mask) return;
    F claim = f_zero();
    if (j.kind == 0) claim = tr[j.ref0];
    else if (j.kind == 1) { const F r = tape[j.ref1]; claim = f_add(f_mul(f_add(f_mul(tr[j.ref0], r), tr[j.ref0 + 1]), r), tr[j.ref0 + 2]); }
    else for (u32 t = 0; t < j.n_terms; ++t) claim = f_add(claim, f_mul(tape[j.sig[t]], tr[j.cl[t]]));
    for (u32 k = 0; k < j.rounds; ++k) {
        F *pk = tr + j.poly_pos + 3 * k;
        if ((j.derive_mask >> k) & 1u) pk[1] = f_sub(f_sub(claim, pk[0]), f_dbl(pk[2]));
        if (!(j.derive_mask >> (k + 1))) break;                       // no derived round after this one
        const F r = tape[j.r_off + k];
        claim = f_add(f_mul(f_add(f_mul(pk[0], r), pk[1]), r), pk[2]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Batched ("plan") launches.  Every argument of every launch of a proof depends on the circuit only (the
// challenges are read from the device tape), so the job descriptors are built once per circuit, kept in
// device memory, and one launch runs the same kernel body for MANY independent sumchecks: block b looks up
// (job, block-in-job) in a map.  The hardware runs at most a handful of kernels at a time; with ~40
// independent sumchecks per proof, batching them side by side is what fills the chip.
// ---------------------------------------------------------------------------------------------------
struct BlkMap { u32 job, bid; };
struct GatherJob { const u32 *rowptr; const uint8_t *e_q; const u32 *e_g; const Half *H; F *M; u32 size; u32 u0; };     // u0: first wire of a row range (rowptr and M already point at it)
// phase 0: Liu gather (g), 1 / 2: phase inits (a).  A phase-1 job can carry the inner product V_u = sum_u eq(r_u,u) V[u]
// of its layer (same rows u): one more coalesced load and two multiplies in a kernel that waits on gathers anyway.
struct LightJob { InitArgs2 a; GatherJob g; Half dot_h; const F *dot_val; F *dot_part; int phase; u32 dot_size; };
struct ChunkJob { InitArgs2 a; const u32 *chunk_beg; const u32 *chunk_end; F *part; u32 n_chunks; int phase; ChunkFuse fuse; int fused, pad; };
struct CombineJob { const u32 *heavy_row; const u32 *heavy_cptr; const F *part; F *M; F *A; u32 n_heavy; int pad; };

__global__ void __launch_bounds__(VP_BLOCK) k_light_multi(const LightJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const LightJob &j = jobs[m.job];
    if (j.phase == 1) init2_light_body<1>(j.a, m.bid);
    else if (j.phase == 2) init2_light_body<2>(j.a, m.bid);
    else liu_gather_body(j.g.rowptr, j.g.e_q, j.g.e_g, j.g.H, j.g.size, j.g.M, m.bid, j.g.u0);
    if (j.phase == 1 && j.dot_part) {                           // uniform per workgroup
        __shared__ F lds[4];
        const u32 row = m.bid * blockDim.x + threadIdx.x;
        F acc[1] = {row < j.dot_size ? f_mul(half_at(j.dot_h, row), j.dot_val[row]) : f_zero()};
        block_sum<1>(acc, lds);
        if (threadIdx.x == 0) j.dot_part[m.bid] = acc[0];
    }
}
// Verifier-side wiring predicates (reference: verifier::betaInitPhase1/2 + predicatePhase1/2, src/verifier.cpp:50-113): for
// layer i,  coeff_l[t] = sum over unary gates g of type t of beta_g[g] beta_u[u_g] (x c_g for Mulc),  bias = the Addc sum
// x c_g,  coeff_r[t][l] = sum over binary gates of type t with second operand in layer l of beta_g[g] beta_u[u_g] beta_v[lv_g].
// The gates of a layer are listed by bucket at upload; a wave sums a piece of <= 512 gates, a second launch adds the
// pieces of each bucket.  flag bit 0: assert gate (beta_g scaled), bits 1-2: class (0 binary, 1 unary, 2 unary x c).
struct PredArgs {
    const u32 *idx; const uint8_t *flag; const u32 *chunk_beg; const u32 *chunk_end; u32 n_chunks;
    Half hg, hu, hv;
    const u32 *gu; const u32 *glv; const F *gc; const F *assert_r; F *part;
};
__global__ void __launch_bounds__(VP_BLOCK) k_pred_chunks(PredArgs a) {
    const u32 c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= a.n_chunks) return;
    const int lane = threadIdx.x & 63;
    F acc = f_zero();
    for (u32 k = a.chunk_beg[c] + lane; k < a.chunk_end[c]; k += 64) {
        const u32 g = a.idx[k];
        const int fl = a.flag[k], cls = fl >> 1;
        F t = f_mul(half_at(a.hg, g), half_at(a.hu, a.gu[g]));
        if (fl & 1) t = f_mul(t, *a.assert_r);
        if (cls == 0) t = f_mul(t, half_at(a.hv, a.glv[g]));
        else if (cls == 2) t = f_mul(t, a.gc[g]);
        acc = f_add(acc, t);
    }
    acc = wave_sum(acc);
    if (lane == 0) a.part[c] = acc;
}
__global__ void __launch_bounds__(VP_BLOCK) k_pred_combine(const u32 *__restrict__ bucket_cptr, u32 n_buckets, const F *__restrict__ part, F *__restrict__ out) {
    const u32 b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n_buckets) return;
    const int lane = threadIdx.x & 63;
    F acc = f_zero();
    for (u32 c = bucket_cptr[b] + lane; c < bucket_cptr[b + 1]; c += 64) acc = f_add(acc, part[c]);
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc;
}

// V_u = V(r_u) = sum_u eq(r_u, u) * V[u] (what phase 1's last fold leaves in the V table, src/prover.cpp:494-500) as an inner
// product: with it phase 2 of a layer no longer waits for phase 1's sumcheck, every sumcheck of the proof is independent.
// beg: first entry of the range [beg, size) the job adds up (a multiple of 2^h1; 0 everywhere but in the per-rank partial inner products of an index-split proof)
struct DotJob { Half h; const F *val; F *part; F *out; u32 size, nblk; int vreal; u32 beg; const unsigned long long *valr; };     // vreal: val[] are real circuit values; valr: their real parts as a dense array
__global__ void __launch_bounds__(VP_BLOCK) k_dot_multi(const DotJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ F lds[4];
    const BlkMap m = map[blockIdx.x];
    const DotJob &j = jobs[m.job];
    F acc[1] = {f_zero()};
    const u32 H = 1u << j.h.h1;
    if (H >= blockDim.x) {
        // eq(r, i) = bf[i mod H] * bs[i div H]: the second factor is common to a run of H consecutive entries, so a workgroup takes whole
        // runs — every thread adds up bf[lo] * V[hi H + lo] over its share of a run and multiplies by bs[hi] once (real values: 8 + 16/k
        // multiplier instructions per entry instead of 24; the kernel was as much multiplier- as memory-bound)
        const u32 runs = (j.size + H - 1) >> j.h.h1;
        for (u32 hb = (j.beg >> j.h.h1) + m.bid; hb < runs; hb += j.nblk) {
            const u32 base = hb << j.h.h1, lim = min(H, j.size - base);
            F in = f_zero();
            if (j.vreal && j.valr) for (u32 lo = threadIdx.x; lo < lim; lo += blockDim.x) in = f_mad31c_rb<false>(j.h.bf[lo], j.valr[base + lo], in);
            else if (j.vreal) for (u32 lo = threadIdx.x; lo < lim; lo += blockDim.x) in = f_mad31c_rb<false>(j.h.bf[lo], j.val[base + lo].re, in);
            else for (u32 lo = threadIdx.x; lo < lim; lo += blockDim.x) in = f_add(in, f_mul(j.h.bf[lo], j.val[base + lo]));
            acc[0] = f_add(acc[0], f_mul(in, j.h.bs[hb]));
        }
    } else if (j.vreal) for (u32 i = j.beg + m.bid * blockDim.x + threadIdx.x; i < j.size; i += j.nblk * blockDim.x) acc[0] = f_mad31c_rb<false>(half_at(j.h, i), j.val[i].re, acc[0]);
    else for (u32 i = j.beg + m.bid * blockDim.x + threadIdx.x; i < j.size; i += j.nblk * blockDim.x) acc[0] = f_add(acc[0], f_mul(half_at(j.h, i), j.val[i]));
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) j.part[m.bid] = acc[0];
}
__global__ void __launch_bounds__(VP_BLOCK) k_dotfin_multi(const DotJob *__restrict__ jobs) {
    __shared__ F lds[4];
    const DotJob &j = jobs[blockIdx.x];
    F acc[1] = {f_zero()};
    for (u32 i = threadIdx.x; i < j.nblk; i += blockDim.x) acc[0] = f_add(acc[0], j.part[i]);
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) *j.out = acc[0];
}
// Index-split proof: the V_u of the split phase-2 chains, complete (u64 sums of the ranks' partial inner products, < 2^64) -> canonical, each into
// the slot its phase-2 init reads.
__global__ void k_vu_place(const F *__restrict__ sums, F *const *__restrict__ dst, u32 n) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    F v = sums[c];
    v.re = m_red128((u128) v.re); v.im = m_red128((u128) v.im);
    *dst[c] = v;
}
__global__ void __launch_bounds__(VP_BLOCK) k_chunks_multi(const ChunkJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const ChunkJob &j = jobs[m.job];
    if (j.phase == 1) init2_chunks_body<1>(j.a, j.chunk_beg, j.chunk_end, j.n_chunks, j.part, m.bid, j.fused ? &j.fuse : nullptr);
    else init2_chunks_body<2>(j.a, j.chunk_beg, j.chunk_end, j.n_chunks, j.part, m.bid, j.fused ? &j.fuse : nullptr);
}
__global__ void __launch_bounds__(VP_BLOCK) k_combine_multi(const CombineJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const CombineJob &j = jobs[m.job];
    init_combine_body(j.heavy_row, j.heavy_cptr, j.n_heavy, j.part, j.M, j.A, m.bid);
}
__global__ void __launch_bounds__(VP_BLOCK, VP_SF_MINB) k_sumfold3b_multi(const SfArgs *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ Sf3bLds sm;
    const BlkMap m = map[blockIdx.x];
    const SfArgs &a = jobs[m.job];
    if (a.has_a) sumfold3b_body<true>(a, m.bid, a.nblk, sm, GenLoad()); else sumfold3b_body<false>(a, m.bid, a.nblk, sm, GenLoad());
}
// First fold launch of a phase-1 / Liu / phase-2 sumcheck with its init fused in (see GenP1 / GenLiu / GenP2).
struct SfGenJob { SfArgs sf; InitArgs2 a; GatherJob g; Half dot_h; F *dot_part; int mode; int pad; };   // mode 1: phase-1 init (GenP1), 2: Liu gather (GenLiu), 3: phase-2 init (GenP2)
__global__ void __launch_bounds__(VP_BLOCK, VP_SF_MINB) k_sumfold3b_gen_multi(const SfGenJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ Sf3bLds sm;
    const BlkMap m = map[blockIdx.x];
    const SfGenJob &j = jobs[m.job];
    if (j.mode == 1) {
        GenP1 g; g.a = &j.a; g.dot_h = j.dot_h; g.dot_part = j.dot_part;
        sumfold3b_body<true>(j.sf, m.bid, j.sf.nblk, sm, g);
    } else if (j.mode == 3) {
        GenP2 g; g.a = &j.a;
        sumfold3b_body<true>(j.sf, m.bid, j.sf.nblk, sm, g);
    } else {
        GenLiu g; g.rowptr = j.g.rowptr; g.e_q = j.g.e_q; g.e_g = j.g.e_g; g.H = j.g.H;
        sumfold3b_body<false>(j.sf, m.bid, j.sf.nblk, sm, g);
    }
}
// The same two launches with a WAVE per chunk (sumfold3c_body): block b of a job takes chunks 4 b + wave, 4 b + wave + 4 nblk, ...
#ifndef VP_SF3C_MINB
#define VP_SF3C_MINB 3
#endif
__global__ void __launch_bounds__(VP_BLOCK, VP_SF3C_MINB) k_sumfold3c_multi(const SfArgs *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ Sf3cLds sm;
    const BlkMap m = map[blockIdx.x];
    const SfArgs &a = jobs[m.job];
    if (a.has_a) sumfold3c_body<true>(a, m.bid, a.nblk, sm, GenLoad()); else sumfold3c_body<false>(a, m.bid, a.nblk, sm, GenLoad());
}
__global__ void __launch_bounds__(VP_BLOCK, VP_SF3C_MINB) k_sumfold3c_gen_multi(const SfGenJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ Sf3cLds sm;
    const BlkMap m = map[blockIdx.x];
    const SfGenJob &j = jobs[m.job];
    if (j.mode == 1) {
        GenP1 g; g.a = &j.a; g.dot_h = j.dot_h; g.dot_part = j.dot_part;
        sumfold3c_body<true>(j.sf, m.bid, j.sf.nblk, sm, g);
    } else if (j.mode == 3) {
        GenP2 g; g.a = &j.a;
        sumfold3c_body<true>(j.sf, m.bid, j.sf.nblk, sm, g);
    } else {
        GenLiu g; g.rowptr = j.g.rowptr; g.e_q = j.g.e_q; g.e_g = j.g.e_g; g.H = j.g.H;
        sumfold3c_body<false>(j.sf, m.bid, j.sf.nblk, sm, g);
    }
}
__global__ void __launch_bounds__(VP_SEG_THREADS) k_seg_multi(const SegArgs *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ SegLds sm;
    const BlkMap m = map[blockIdx.x];
    const SegArgs &a = jobs[m.job];
    if (a.has_a) seg_body<true>(a, m.bid, a.nblk, sm); else seg_body<false>(a, m.bid, a.nblk, sm);
}
__global__ void __launch_bounds__(VP_EMIT_THREADS) k_emit_multi(const EmitArgs *__restrict__ jobs) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    emit_body(jobs[blockIdx.x], smem_raw);
}

}  // namespace vp
