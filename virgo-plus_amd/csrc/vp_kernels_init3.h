// Batched path, part 3: entry-parallel phase inits (src/prover.cpp:189-420 as gathers, one LANE PER CONTRIBUTION).
// Part of the single translation unit vpgpu.hip (see vp_kernels.h for the overall layout rules).
#pragma once
#include "vp_kernels_batch.h"

// ===================================================================================================
// Why: k_light_multi gives a lane one target ROW and lets it walk that row's contribution list.  The rows of a layer
// have 0..16 contributions each (SHA-256: 0.7 on average), so a wave executes the longest list of its 64 rows while
// most lanes idle: measured 8.3 wave-instructions per contribution against ~2.5 for the arithmetic itself
// (profiles/r01_l_pmc_summary_b64.json: 52.9 M VALU wave-instructions per launch for 6.4 M contributions), and every
// list walk is a chain of dependent loads (row pointer -> record -> operands).
//
// Here a 256-thread workgroup owns a CHUNK of 512 consecutive target rows.  The contributions of a chunk are one
// contiguous range of the target-sorted lists (chunk pointers built at upload); lane i of the workgroup takes entry
// base + i: records are read fully coalesced, every lane has exactly one term to compute (no divergence in the
// multiplies), and the terms of one row — adjacent lanes — are combined by a segmented scan inside the 16-lane DPP
// rows (row_shr, no LDS traffic).  The last lane of each segment adds its partial sum to the row's accumulator in LDS
// (ds_add_u64, four words per row: no two segment ends of one instruction share a row, so no bank serialisation).
// Afterwards the workgroup walks its 512 rows once: accumulator -> canonical value, plus the per-row work of the phase
// (Liu: eq(r_u, u); phase 2: the V gather; rows with more than 16 contributions: the sums the chunk kernels left in
// global memory) and writes mult/add — or, inside the fused fold kernel, feeds round k directly.
//
// MEASURED (round 2, profiles/r02_b_ab_init3_*.json): bit-exact, but no faster.  Stand-alone the launch takes the same time as
// k_light_multi (x64: 278 vs 277 us for both init launches; x1024: 1.54 vs 1.56 ms) — the init launches move ~500 MB per
// launch at x64, 60 % of it the mult/add tables they WRITE for every wire (most rows hold zero or one contribution), and run at
// 3.5 TB/s of mixed read / write / 16-byte gathers; the divergence this kernel removes was not what they waited for.  Inside
// the fused fold launch (GenI3P1 / GenI3Liu) it is slower than the row-per-lane gather (x1024: 4.7 vs 3.5 ms for the two fused
// launches): zeroing + two more barriers per 512-entry chunk at 3 workgroups per CU.  Kept as VP_INIT3=1 (parity-tested).
//
// Exactness: a term is canonical (< 2^61); three scan steps add at most 8 of them (< 2^64), one Mersenne fold brings the
// partial below 2^61 + 8, the fourth step adds two of those.  A light row (<= 16 contributions, contiguous) touches at
// most two DPP rows, a Liu row (<= 64 later layers) at most five; partials are folded once more (< 2^61 + 2) before the
// LDS add, so an accumulator holds < 7 * 2^61 < 2^64.  Field addition is associative and commutative on residues, so the
// canonical results equal the reference's sums bit for bit.
// ===================================================================================================
namespace vp {

#define VP_I3_ROWS 512
struct Csr3 {
    const u32 *cptr;                 // [n_chunks + 1] first entry of each 512-row chunk (rows with > VP_LIGHT_MAX contributions excluded)
    const u32 *e_g, *e_x;            // per entry: gate index (eq table), operand index
    const uint16_t *e_tl, *e_r;      // per entry: (assert << 15 | type << 8 | layer) as in the row lists; row offset inside its chunk
    const u32 *hptr;                 // [n_chunks + 1] range of heavy_row[] that falls into the chunk
    const u32 *heavy_row;
    u32 n_chunks, pad;
};
// Liu gather lists in the same form: e_g = position in the subset, e_q (u8) = which later layer's subset (index into H)
struct Csr3L { const u32 *cptr; const u32 *e_g; const uint8_t *e_q; const uint16_t *e_r; u32 n_chunks, pad; };

struct I3Lds { u64 acc[4][VP_I3_ROWS]; F red[4]; };     // acc: m.re | m.im | a.re | a.im per row of the chunk

template <int CTRL>
__device__ __forceinline__ u64 dpp_shr_u64(u64 v) {       // value of the lane CTRL-0x110 places below inside the 16-lane row, 0 where there is none
    const u32 lo = __builtin_amdgcn_update_dpp(0u, (u32) v, CTRL, 0xf, 0xf, false);
    const u32 hi = __builtin_amdgcn_update_dpp(0u, (u32) (v >> 32), CTRL, 0xf, 0xf, false);
    return ((u64) hi << 32) | lo;
}
template <int CTRL, int NV>
__device__ __forceinline__ void seg_step(u32 row, u64 (&v)[NV]) {
    const u32 pr = (u32) __builtin_amdgcn_update_dpp((int) 0xffffffffu, (int) row, CTRL, 0xf, 0xf, false);     // lanes without a source keep the sentinel
    const bool same = pr == row;
#pragma unroll
    for (int q = 0; q < NV; ++q) { const u64 t = dpp_shr_u64<CTRL>(v[q]); v[q] += same ? t : 0ull; }
}
// Inclusive segmented sum (segments = runs of equal `row`) inside each 16-lane DPP row; on return the LAST lane of every
// run holds the run's sum, folded below 2^61 + 2.  `end` tells a lane whether it is such a last lane.
template <int NV>
__device__ __forceinline__ void seg_sum16(u32 row, u64 (&v)[NV], bool &end) {
    seg_step<0x111, NV>(row, v);
    seg_step<0x112, NV>(row, v);
    seg_step<0x114, NV>(row, v);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (v[q] & P61) + (v[q] >> 61);
    seg_step<0x118, NV>(row, v);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (v[q] & P61) + (v[q] >> 61);
    const u32 nx = (u32) __builtin_amdgcn_update_dpp((int) 0xffffffffu, (int) row, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
    end = nx != row;                                          // lane 15 of a DPP row keeps the sentinel: always an end
}

// term of one contribution, phase 1 (phase-1 column): mult[u] += tm, add[u] += ta
__device__ __forceinline__ void i3_term_p1(const InitArgs2 &a, u32 g, u32 x, u32 tl, F &tm, F &ta) {
    const int ty = (tl >> 8) & 0x7f, l = tl & 0xff;
    F t = half_at(a.hg, g);
    if (tl & 0x8000) t = f_mul(t, *a.assert_r);
    F ty_ = f_zero();
    if (l != 0xff) ty_ = f_mul(a.vals[l][x], t);
    tm = f_zero(); ta = f_zero();
    switch (ty) {
        case T_ADD: ta = ty_; tm = t; break;
        case T_SUB: ta = f_neg(ty_); tm = t; break;
        case T_ANTISUB: ta = ty_; tm = f_neg(t); break;
        case T_MUL: tm = ty_; break;
        case T_NAAB: ta = ty_; tm = f_neg(ty_); break;
        case T_ANTINAAB: tm = f_sub(t, ty_); break;
        case T_ADDC: ta = f_mul(a.gc[g], t); tm = t; break;
        case T_MULC: tm = f_mul(a.gc[g], t); break;
        case T_COPY: tm = t; break;
        case T_NOT: ta = t; tm = f_neg(t); break;
        case T_XOR: ta = ty_; tm = f_sub(t, f_dbl(ty_)); break;
        default: break;
    }
}
// phase 2 (Appendix A, phase-2 column): t = eq_g(g) eq_u(u), X = V_u
__device__ __forceinline__ void i3_term_p2(const InitArgs2 &a, u32 g, u32 x, u32 tl, const F &vu, F &tm, F &ta) {
    const int ty = (tl >> 8) & 0x7f;
    F t = half_at(a.hg, g);
    if (tl & 0x8000) t = f_mul(t, *a.assert_r);
    t = f_mul(t, half_at(a.hu, x));
    const F tv = f_mul(t, vu);
    tm = f_zero(); ta = f_zero();
    switch (ty) {
        case T_ADD: tm = t; ta = tv; break;
        case T_SUB: tm = f_neg(t); ta = tv; break;
        case T_ANTISUB: tm = t; ta = f_neg(tv); break;
        case T_MUL: tm = tv; break;
        case T_NAAB: tm = f_sub(t, tv); break;
        case T_ANTINAAB: tm = f_neg(tv); ta = tv; break;
        case T_XOR: ta = tv; tm = f_sub(t, f_dbl(tv)); break;
        case T_COPY: ta = tv; break;
        case T_NOT: ta = f_sub(t, tv); break;
        case T_ADDC: ta = f_mul(t, f_add(a.gc[g], vu)); break;
        case T_MULC: ta = f_mul(tv, a.gc[g]); break;
        default: break;
    }
}

// Accumulate every light contribution of chunk `ch` into sm.acc (zeroed here).  PHASE 1 / 2: InitArgs2 + Csr3; PHASE 0: Liu lists.
// LOAD_HEAVY (fused fold launch): rows with more than VP_LIGHT_MAX contributions take the sums the chunk kernels left in a.M / a.A.
template <int PHASE, bool LOAD_HEAVY = false>
__device__ __forceinline__ void i3_accumulate(const InitArgs2 &a, const Csr3 &c, const Csr3L &cl, const Half *__restrict__ H, u32 ch, I3Lds &sm) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 4 * VP_I3_ROWS; i += blockDim.x) (&sm.acc[0][0])[i] = 0;
    __syncthreads();
    if (LOAD_HEAVY && PHASE != 0) {                            // heavy rows have no light entries: nobody else touches their slots
        for (u32 h = c.hptr[ch] + tid; h < c.hptr[ch + 1]; h += blockDim.x) {
            const u32 row = c.heavy_row[h], q = row - ch * VP_I3_ROWS;
            const F m = a.M[row], ad = a.A[row];
            sm.acc[0][q] = m.re; sm.acc[1][q] = m.im; sm.acc[2][q] = ad.re; sm.acc[3][q] = ad.im;
        }
    }
    const u32 eb = PHASE == 0 ? cl.cptr[ch] : c.cptr[ch], ee = PHASE == 0 ? cl.cptr[ch + 1] : c.cptr[ch + 1];
    const F vu = PHASE == 2 ? *a.Vu : f_zero();
    for (u32 e0 = eb; e0 < ee; e0 += blockDim.x) {            // uniform trip count
        const u32 e = e0 + tid;
        const bool ok = e < ee;
        u32 row = 0xfffeu;                                     // padding lanes: a row of their own, zero terms, never written
        constexpr int NV = PHASE == 0 ? 2 : 4;
        u64 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = 0;
        if (ok) {
            if (PHASE == 0) {
                row = cl.e_r[e];
                const F t = half_at(H[cl.e_q[e]], cl.e_g[e]);
                v[0] = t.re; v[1] = t.im;
            } else {
                row = c.e_r[e];
                F tm, ta;
                if (PHASE == 1) i3_term_p1(a, c.e_g[e], c.e_x[e], c.e_tl[e], tm, ta);
                else i3_term_p2(a, c.e_g[e], c.e_x[e], c.e_tl[e], vu, tm, ta);
                v[0] = tm.re; v[1] = tm.im; v[NV - 2] = ta.re; v[NV - 1] = ta.im;
            }
        }
        bool end;
        seg_sum16<NV>(row, v, end);
        if (ok && end) {
#pragma unroll
            for (int q = 0; q < NV; ++q)
                if (v[q]) atomicAdd((unsigned long long *) &sm.acc[PHASE == 0 ? q : q][row], (unsigned long long) v[q]);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ F i3_acc_m(const I3Lds &sm, u32 r) { return f_make(m_fold(sm.acc[0][r]), m_fold(sm.acc[1][r])); }
__device__ __forceinline__ F i3_acc_a(const I3Lds &sm, u32 r) { return f_make(m_fold(sm.acc[2][r]), m_fold(sm.acc[3][r])); }

// ---- stand-alone launches: the tables are written to HBM at full length -------------------------------------------------
struct I3Job {
    InitArgs2 a; Csr3 c; Csr3L cl; const Half *H; F *liuM; u32 liu_size;
    Half dot_h; const F *dot_val; F *dot_part; u32 dot_size; int phase;      // phase 0: Liu gather, 1 / 2: phase inits; a phase-1 job can carry V_u's inner product
};
template <int PHASE>
__device__ __forceinline__ void i3_body(const I3Job &j, u32 ch, I3Lds &sm) {
    i3_accumulate<PHASE>(j.a, j.c, j.cl, j.H, ch, sm);
    const u32 r0 = ch * VP_I3_ROWS;
    const u32 n_rows = PHASE == 0 ? j.liu_size : j.a.n_rows;
    for (u32 q = threadIdx.x; q < VP_I3_ROWS; q += blockDim.x) {
        const u32 row = r0 + q;
        if (row >= n_rows) break;
        if (PHASE == 0) {
            j.liuM[row] = f_add(half_at(j.H[0], row), i3_acc_m(sm, q));
        } else {
            if (PHASE == 2) {
                const int l = j.a.s_layer[row];
                if (l != 0xfe) j.a.V[row] = (l == 0xff) ? f_zero() : j.a.vals[l][j.a.s_idx[row]];   // 0xfe: padding slot, never read
            }
            j.a.M[row] = i3_acc_m(sm, q);
            j.a.A[row] = i3_acc_a(sm, q);
        }
    }
    // rows with more than VP_LIGHT_MAX contributions: the combine launch that follows writes them (same chain, later step)
    if (PHASE == 1 && j.dot_part) {                            // uniform per workgroup
        F acc[1] = {f_zero()};
        for (u32 q = threadIdx.x; q < VP_I3_ROWS; q += blockDim.x) {
            const u32 row = r0 + q;
            if (row < j.dot_size) acc[0] = f_add(acc[0], f_mul(half_at(j.dot_h, row), j.dot_val[row]));
        }
        __syncthreads();
        block_sum<1>(acc, sm.red);
        if (threadIdx.x == 0) j.dot_part[ch] = acc[0];
    }
}

// ---- inside the fused fold launch (sumfold3b_body, Gen::MODE 3 / 4): the 512 entries of a fold chunk ARE one init chunk --------
struct GenI3P1 {
    static constexpr int MODE = 3;
    const InitArgs2 *a; const Csr3 *c; Half dot_h; F *dot_part; I3Lds *sm;
    __device__ __forceinline__ void chunk(u32 ch, u32 i0, u32 valid, F &m0, F &a0, F &m1, F &a1) const {
        i3_accumulate<1, true>(*a, *c, Csr3L{}, nullptr, ch, *sm);
        const u32 q = i0 - ch * VP_I3_ROWS;                    // one table per job at offset 0: i0 = 512 ch + 2t
        m0 = i3_acc_m(*sm, q); a0 = i3_acc_a(*sm, q); m1 = i3_acc_m(*sm, q + 1); a1 = i3_acc_a(*sm, q + 1);
        __syncthreads();                                       // the accumulators are zeroed again by the next chunk
    }
};
struct GenI3Liu {
    static constexpr int MODE = 4;
    const Csr3L *cl; const Half *H; I3Lds *sm;
    __device__ __forceinline__ void chunk(u32 ch, u32 i0, u32 valid, F &m0, F &a0, F &m1, F &a1) const {
        i3_accumulate<0>(InitArgs2{}, Csr3{}, *cl, H, ch, *sm);
        const u32 q = i0 - ch * VP_I3_ROWS;
        a0 = f_zero(); a1 = f_zero();
        m0 = i0 < valid ? f_add(half_at(H[0], i0), i3_acc_m(*sm, q)) : f_zero();
        m1 = i0 + 1 < valid ? f_add(half_at(H[0], i0 + 1), i3_acc_m(*sm, q + 1)) : f_zero();
        __syncthreads();
    }
};

}  // namespace vp
