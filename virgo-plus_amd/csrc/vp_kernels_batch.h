// Batched path, part 1: init kernels without materialised eq tables, the register/shuffle fold k_sumfold<R>, the block-cooperative fold k_sumfold3b, the single-workgroup tail k_tail.
// Part of the single translation unit vpgpu.hip (see vp_kernels.h for the overall layout rules).
#pragma once
#include "vp_kernels_round.h"

// ===================================================================================================
// Batched path (vp_prove_gkr): every challenge is on the device tape, so one launch can cover several
// rounds and whole sumcheck tails.  Same field values as the per-round kernels above, fewer bytes
// and far fewer launches:
//   * eq tables are never materialised: consumers multiply the two half tables on the fly;
//   * k_sumfold<R>: a wave takes 64*2^R CONTIGUOUS entries per table (2^R fully coalesced 1 KiB loads),
//     produces the sums of R rounds and stores the 64 folded entries of round k+R; neighbours are
//     exchanged with wavefront shuffles (lane ^ 2^s at level s), no LDS staging of table data;
//   * k_tail: one workgroup runs ALL remaining rounds once the live tables are small, adds the block
//     partials of the earlier rounds, retires tables into add_term, and emits every round polynomial of
//     the sumcheck plus the final claims.
// ===================================================================================================
namespace vp {

struct Half { const F *bf; const F *bs; int h1; int pad; };
__device__ __forceinline__ F half_at(const Half &h, u32 i) {
    return f_mul(h.bf[i & ((1u << h.h1) - 1)], h.bs[i >> h.h1]);
}
// half_at(h, i) + c with one reduction (c canonical)
__device__ __forceinline__ F half_mad(const Half &h, u32 i, const F &c) {
    return f_mad31c<false, VP_MADSHIFT != 0>(h.bf[i & ((1u << h.h1) - 1)], h.bs[i >> h.h1], c);
}

struct InitArgs2 {
    const u32 *rowptr; const u32 *e_g; const u32 *e_x; const uint16_t *e_tl;
    Half hg, hu;              // eq(r_liu, .) over layer i, eq(r_u, .) over layer i-1
    F *const *vals;
    const F *gc;
    const F *Vu;              // phase 2
    const F *assert_r;        // scales beta_g of assert gates (bit 15 of e_tl)
    F *V, *M, *A;
    const uint8_t *s_layer; const u32 *s_idx;    // phase 2: slot -> (source layer, index) for the V gather
    u32 n_rows;
    int vreal;                // 1: every circuit value is real (vp_evaluate), beta * V[v] takes the half-price product
    const unsigned long long *const *valsr;      // vreal: the same values as dense arrays of their real parts — a gathered operand is 8 bytes, eight to a cache line
};
__device__ __forceinline__ F mul_val(const InitArgs2 &a, const F &y, const F &t) {       // V[v] * t
    return a.vreal ? f_mad31c_rb<false>(t, y.re, f_zero()) : f_mul(y, t);
}
__device__ __forceinline__ F val_at(const InitArgs2 &a, int l, u32 x) {                  // circuit value x of layer l
    return (a.vreal && a.valsr) ? f_make(a.valsr[l][x], 0) : a.vals[l][x];
}

template <int PHASE>
__device__ __forceinline__ void contrib2(const InitArgs2 &a, u32 e, F &m, F &ad, const F &vu) {       // vu = V_u (phase 2), loaded once per row
    const u32 g = a.e_g[e], x = a.e_x[e], tl = a.e_tl[e];
    const int ty = (tl >> 8) & 0x7f;
    F t = half_at(a.hg, g);
    if (tl & 0x8000) t = f_mul(t, *a.assert_r);
    if (PHASE == 1) {
        const int l = tl & 0xff;
        F ty_ = f_zero();
        if (l != 0xff && VP_CHK_LAYER(l, x)) ty_ = mul_val(a, val_at(a, l, x), t);
        switch (ty) {
            case T_ADD: ad = f_add(ad, ty_); m = f_add(m, t); break;
            case T_SUB: ad = f_sub(ad, ty_); m = f_add(m, t); break;
            case T_ANTISUB: ad = f_add(ad, ty_); m = f_sub(m, t); break;
            case T_MUL: m = f_add(m, ty_); break;
            case T_NAAB: ad = f_add(ad, ty_); m = f_sub(m, ty_); break;
            case T_ANTINAAB: m = f_add(m, f_sub(t, ty_)); break;
            case T_ADDC: ad = f_add(ad, f_mul(a.gc[g], t)); m = f_add(m, t); break;
            case T_MULC: m = f_add(m, f_mul(a.gc[g], t)); break;
            case T_COPY: m = f_add(m, t); break;
            case T_NOT: ad = f_add(ad, t); m = f_sub(m, t); break;
            case T_XOR: ad = f_add(ad, ty_); m = f_add(m, f_sub(t, f_dbl(ty_))); break;
            default: break;
        }
    } else {
        t = f_mul(t, half_at(a.hu, x));
        const F tv = f_mul(t, vu);                         // t * V_u
        switch (ty) {                                      // SURVEY.md Appendix A, phase-2 column
            case T_ADD: m = f_add(m, t); ad = f_add(ad, tv); break;
            case T_SUB: m = f_sub(m, t); ad = f_add(ad, tv); break;
            case T_ANTISUB: m = f_add(m, t); ad = f_sub(ad, tv); break;
            case T_MUL: m = f_add(m, tv); break;
            case T_NAAB: m = f_add(m, f_sub(t, tv)); break;
            case T_ANTINAAB: m = f_sub(m, tv); ad = f_add(ad, tv); break;
            case T_XOR: ad = f_add(ad, tv); m = f_add(m, f_sub(t, f_dbl(tv))); break;
            case T_COPY: ad = f_add(ad, tv); break;
            case T_NOT: ad = f_add(ad, f_sub(t, tv)); break;
            case T_ADDC: ad = f_add(ad, f_mul(t, f_add(a.gc[g], vu))); break;
            case T_MULC: ad = f_add(ad, f_mul(tv, a.gc[g])); break;
            default: break;
        }
    }
}

// The same phase-1 contribution in three stages, so that a caller can issue the loads of several rows level by level (entry
// -> operands -> arithmetic) instead of walking one row's dependent chain to its end before the next row's first load.
struct P1Entry { u32 g, x, tl; };
struct P1Vals { F bf, bs, y; };
__device__ __forceinline__ P1Entry p1_load(const InitArgs2 &a, u32 e) { P1Entry c; c.g = a.e_g[e]; c.x = a.e_x[e]; c.tl = a.e_tl[e]; return c; }
__device__ __forceinline__ P1Vals p1_gather(const InitArgs2 &a, const P1Entry &c) {
    P1Vals v;
    v.bf = a.hg.bf[c.g & ((1u << a.hg.h1) - 1)]; v.bs = a.hg.bs[c.g >> a.hg.h1];
    const int l = c.tl & 0xff;
    v.y = (l != 0xff && VP_CHK_LAYER(l, c.x)) ? val_at(a, l, c.x) : f_zero();
    return v;
}
__device__ __forceinline__ void p1_apply(const InitArgs2 &a, const P1Entry &c, const P1Vals &v, F &m, F &ad) {       // == contrib2<1>
    const int ty = (c.tl >> 8) & 0x7f;
    F t = f_mul(v.bf, v.bs);
    if (c.tl & 0x8000) t = f_mul(t, *a.assert_r);
    F ty_ = f_zero();
    if ((c.tl & 0xff) != 0xff) ty_ = mul_val(a, v.y, t);
    switch (ty) {
        case T_ADD: ad = f_add(ad, ty_); m = f_add(m, t); break;
        case T_SUB: ad = f_sub(ad, ty_); m = f_add(m, t); break;
        case T_ANTISUB: ad = f_add(ad, ty_); m = f_sub(m, t); break;
        case T_MUL: m = f_add(m, ty_); break;
        case T_NAAB: ad = f_add(ad, ty_); m = f_sub(m, ty_); break;
        case T_ANTINAAB: m = f_add(m, f_sub(t, ty_)); break;
        case T_ADDC: ad = f_add(ad, f_mul(a.gc[c.g], t)); m = f_add(m, t); break;
        case T_MULC: m = f_add(m, f_mul(a.gc[c.g], t)); break;
        case T_COPY: m = f_add(m, t); break;
        case T_NOT: ad = f_add(ad, t); m = f_sub(m, t); break;
        case T_XOR: ad = f_add(ad, ty_); m = f_add(m, f_sub(t, f_dbl(ty_))); break;
        default: break;
    }
}

template <int PHASE>
__device__ __forceinline__ void init2_light_body(const InitArgs2 &a, u32 bid) {
    u32 row = bid * blockDim.x + threadIdx.x;
    if (row >= a.n_rows) return;
    if (PHASE == 2) {
        const int l = a.s_layer[row];
        if (l != 0xfe) a.V[row] = (l == 0xff || !VP_CHK((unsigned) l < g_vp_chk_layers() && a.s_idx[row] < g_vp_chk_lsize(l), 2, l, a.s_idx[row], row)) ? f_zero() : val_at(a, l, a.s_idx[row]);   // 0xfe: padding slot, never read
    }
    u32 b = a.rowptr[row], e = a.rowptr[row + 1];
    if (!VP_CHK(b <= e, 3, row, b, e)) return;
    if (e - b > VP_LIGHT_MAX) return;
    F m = f_zero(), ad = f_zero();
    const F vu = PHASE == 2 ? *a.Vu : f_zero();
    for (u32 k = b; k < e; ++k) contrib2<PHASE>(a, k, m, ad, vu);
    a.M[row] = m;
    a.A[row] = ad;
}
template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK) k_init2_light(InitArgs2 a) { init2_light_body<PHASE>(a, blockIdx.x); }

// Heavy rows finished by the chunk launch itself (ChunkFuse, plan path): a row of ONE chunk (nearly all of them) is written straight into
// the mult / add arrays; of a row cut into several chunks the wave that arrives last adds the others' partials (written through to memory
// with sc1 stores, counted with a relaxed agent-scope atomic per ROW — a few arrivals per counter, no fence) and resets the counter.
struct ChunkFuse { const u32 *chunk_h; const u32 *heavy_row; const u32 *heavy_cptr; u32 *heavy_cnt; F *M; F *A; };
template <int PHASE>
__device__ __forceinline__ void init2_chunks_body(const InitArgs2 &a, const u32 *__restrict__ chunk_beg, const u32 *__restrict__ chunk_end,
                                                  u32 n_chunks, F *__restrict__ part, u32 bid, const ChunkFuse *fuse = nullptr) {
    const u32 c = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    F m = f_zero(), ad = f_zero();
    const F vu = PHASE == 2 ? *a.Vu : f_zero();
    for (u32 k = chunk_beg[c] + lane; k < chunk_end[c]; k += 64) contrib2<PHASE>(a, k, m, ad, vu);
    m = wave_sum(m);
    ad = wave_sum(ad);
    if (!fuse) { if (lane == 0) { part[2 * c] = m; part[2 * c + 1] = ad; } return; }
    const u32 h = fuse->chunk_h[c], c0 = fuse->heavy_cptr[h], n = fuse->heavy_cptr[h + 1] - c0, row = fuse->heavy_row[h];
    if (n == 1) { if (lane == 0) { fuse->M[row] = m; fuse->A[row] = ad; } return; }
    int last = 0;
    if (lane == 0) {
        unsigned long long *w = reinterpret_cast<unsigned long long *>(part + 2 * c);
        cf_st(w, m.re); cf_st(w + 1, m.im); cf_st(w + 2, ad.re); cf_st(w + 3, ad.im);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the partial is out before it is counted
        last = __hip_atomic_fetch_add(fuse->heavy_cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    F tm = f_zero(), ta = f_zero();
    for (u32 i = lane; i < n; i += 64) {
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(part + 2 * (c0 + i));
        tm = f_add(tm, f_make(cf_ld(w), cf_ld(w + 1))); ta = f_add(ta, f_make(cf_ld(w + 2), cf_ld(w + 3)));
    }
    tm = wave_sum(tm); ta = wave_sum(ta);
    if (lane == 0) { fuse->M[row] = tm; fuse->A[row] = ta; __hip_atomic_store(fuse->heavy_cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK)
k_init2_chunks(InitArgs2 a, const u32 *__restrict__ chunk_beg, const u32 *__restrict__ chunk_end, u32 n_chunks,
               F *__restrict__ part) { init2_chunks_body<PHASE>(a, chunk_beg, chunk_end, n_chunks, part, blockIdx.x); }

// Liu init as a gather (src/prover.cpp:396-414): for every u of layer i-1 the (later layer, subset
// position) pairs that point at it were listed at upload; M[u] = s0*eq(r_u,u) + sum eq_q(g).
__device__ __forceinline__ void liu_gather_body(const u32 *__restrict__ rowptr, const uint8_t *__restrict__ e_q, const u32 *__restrict__ e_g,
                                                const Half *__restrict__ H, u32 size, F *__restrict__ M, u32 bid, u32 u0 = 0) {
    u32 u = bid * blockDim.x + threadIdx.x;
    if (u >= size) return;
    F m = half_at(H[0], u0 + u);
    if (!VP_CHK(rowptr[u] <= rowptr[u + 1], 4, u, rowptr[u], rowptr[u + 1])) return;
    for (u32 k = rowptr[u]; k < rowptr[u + 1]; ++k) if (VP_CHK(e_q[k] < g_vp_chk_liu(), 4, u, k, e_q[k])) m = f_add(m, half_at(H[e_q[k]], e_g[k]));
    M[u] = m;
}
__global__ void __launch_bounds__(VP_BLOCK)
k_liu_gather(const u32 *__restrict__ rowptr, const uint8_t *__restrict__ e_q, const u32 *__restrict__ e_g,
             const Half *__restrict__ H, u32 size, F *__restrict__ M) { liu_gather_body(rowptr, e_q, e_g, H, size, M, blockIdx.x); }

__global__ void __launch_bounds__(VP_BLOCK)
k_vres2(Half h, const F *__restrict__ val, u32 size, F *out_dev) {
    __shared__ F lds[4];
    F acc[1] = {f_zero()};
    for (u32 i = threadIdx.x; i < size; i += blockDim.x) acc[0] = f_add(acc[0], f_mul(half_at(h, i), val[i]));
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) *out_dev = acc[0];
}

// ---------------------------------------------------------------------------------------------------
// k_sumfold<R>: R rounds per launch over tables whose length is a multiple of 64*2^R.
// ---------------------------------------------------------------------------------------------------
struct SfTab { u32 off, len, valid, chunk_start; };
struct SfArgs {
    const F *inV, *inM, *inA;
    const unsigned long long *inVr;      // keep_y0 bit 2 (all-real first round): the V entries as a dense array of their real parts, or nullptr
    F *outV, *outM, *outA;
    const F *r;               // r[s] = challenge of the s-th round of this launch
    F *part;                  // part[(s * part_stride) + block*3 + c]
    u32 part_stride;
    u32 total_chunks;
    int n_tab, has_a;
    u32 nblk;                 // batched launches: blocks given to this job
    int keep_y0;              // bit 2: the V entries of the first round of this launch are circuit values and vp_evaluate found them all REAL (sf_pair_step_rv).  k_sumfold3b / 4b, bit 0: the first round of this launch sums m1 v1 + a1 (it is round 1 of its sumcheck, whose sum cannot be
                              // derived from a previous claim); bit 1: the later rounds do too (VP_DROP_Y=0).  Otherwise the product is left out (sf_pair_step)
    SfTab t[VP_MAX_TAB];
};

__device__ __forceinline__ F shfl_xor_F(const F &x, int mask) {
    F y;
    y.re = __shfl_xor(x.re, mask, 64);
    y.im = __shfl_xor(x.im, mask, 64);
    return y;
}

// One level: regs x[0..2n) -> x[0..n).  Lane keeps the pair (lo, hi) = two neighbouring table entries:
// lanes with bit s clear take theirs from the even register, the others from the odd register.
// k_sumfold3b: the same three rounds per launch, laid out for parallelism instead of per-lane work.
//
// k_sumfold<3> gives a lane 8 entries of each table (7 dependent pair steps, ~250 VGPRs): a 2^20-entry table
// is only 2048 waves, two per SIMD, and the kernel runs at the latency of its own dependency chain.  Here a
// 256-thread workgroup takes the same 512-entry chunk: round k+0 is one pair per thread (entries 2t, 2t+1
// as one 32-byte load per table), the 256 folded entries go through LDS, round k+1 runs on the first two
// waves, round k+2 on the first.  Idle waves issue nothing, so the instruction count is that of the dense
// schedule, but a chunk exposes 4x the waves, a thread holds 6 entries instead of 24 (~100 VGPRs, 5 waves
// per SIMD), and the sums are accumulated unreduced (one fold per chunk, not one canonical add per pair).
//
// Lazy arithmetic used below (values are limbs of F):
//   d = x1 + p - x0            in [0, 2p]           (x0, x1 canonical)
//   a*b + c  with a, b in [0, 2p], c in [0, p]:  f_mad31 (vp_field.h), canonical result; every stored value
//   is canonical, so results are bit-identical to the strict sequence.
// ---------------------------------------------------------------------------------------------------
// (f_sub_lazy, f_mad_lazy / f_mad_c and the unreduced sums Lz live in vp_kernels_round.h: the per-round kernels use them too)

// one pair of one table family: sums into (X, Y, Z) = (sum dm*dv, sum m1*v1 + a1, sum m0*v0 + a0), folds with r.
// The round polynomial is a = X, b = Y - X - Z, c = Z.  From round 2 of a sumcheck on, Y is redundant: the verifier's own check
// S_k(0) + S_k(1) = S_{k-1}(r_{k-1}) (src/verifier.cpp:208,249,295) gives Y = S_{k-1}(r_{k-1}) - Z for the totals over all tables
// of the phase, add_term included, and the closing kernel has both (k_emit, derive_mask).  KEEP_Y = false leaves the product
// out: five multiply-adds per pair instead of six, the same field elements in the transcript.
// (Round 3, built and measured: M and A entries travelling through LDS weakly reduced (<= p + 3) between the rounds of a launch, so that the
// conditional subtraction disappears from two of the three folds.  The lazy difference of two such values needs an offset that is a multiple
// of p AND at least p + 3, i.e. 2p, and then exceeds the 2^62 the split multiply takes; with the extra fold that brings it back the net gain
// is ~3 % of the fold arithmetic.  Not kept.  Nor the fold through f_fold31, the negated challenge limb instead of a negated difference: HISTORY.md.)
template <bool WEAK> __device__ __forceinline__ F sf_fold(const F &r, const F &d, const F &x0) { return f_mad_c<WEAK>(r, d, x0); }
template <bool HAS_A>
__device__ __forceinline__ void sf_pair_step(const F &v0, const F &v1, const F &m0, const F &m1, const F &a0, const F &a1,
                                             const F &r, Lz &X, Lz &Y, Lz &Z, F &vo, F &mo, F &ao, bool keep_y) {
    const F dv = f_sub_lazy(v1, v0), dm = f_sub_lazy(m1, m0);
    lz_add(X, f_mad_lazy<true>(dm, dv, f_zero()));                       // sums take weakly reduced products (< 2^61 + 4)
    if (keep_y) lz_add(Y, f_mad_c<true>(m1, v1, HAS_A ? a1 : f_zero()));
    lz_add(Z, f_mad_c<true>(m0, v0, HAS_A ? a0 : f_zero()));
    vo = sf_fold<false>(r, dv, v0);                                      // stored values are canonical
    mo = sf_fold<false>(r, dm, m0);
    if (HAS_A) ao = sf_fold<false>(r, f_sub_lazy(a1, a0), a0);
}

// The same pair step when v0 and v1 are REAL (round 1 of a sumcheck over a circuit with real values, f_mad31c_rb in vp_field.h):
// the four products that have a V factor cost half the multiplier instructions.
template <bool HAS_A>
__device__ __forceinline__ void sf_pair_step_rv(u64 v0, u64 v1, const F &m0, const F &m1, const F &a0, const F &a1,
                                                const F &r, Lz &X, Lz &Y, Lz &Z, F &vo, F &mo, F &ao, bool keep_y) {
    const u64 dv = v1 + P61 - v0;
    const F dm = f_sub_lazy(m1, m0);
    lz_add(X, f_mad31_rb<true, VP_MADSHIFT>(dm, dv, f_zero()));
    if (keep_y) lz_add(Y, f_mad31c_rb<true, VP_MADSHIFT>(m1, v1, HAS_A ? a1 : f_zero()));
    lz_add(Z, f_mad31c_rb<true, VP_MADSHIFT>(m0, v0, HAS_A ? a0 : f_zero()));
    vo = f_mad31c_rb<false, VP_MADSHIFT>(r, dv, f_make(v0, 0));
    mo = sf_fold<false>(r, dm, m0);
    if (HAS_A) ao = sf_fold<false>(r, f_sub_lazy(a1, a0), a0);
}

// (Measured and dropped, HISTORY.md: rotating the wave roles from chunk to chunk; requesting the next chunk's six entries into registers ahead of
// the arithmetic; staging the next chunk in LDS with global_load_lds_dwordx4 — the kernel waits for its multiplier, not for memory.)
struct Sf3bLds { F s1[3][256]; F s2[3][128]; F red[4][9]; Lz acc2[3][128]; Lz acc3[3][64]; F dred[4]; };   // acc2/acc3: per-thread sums of rounds k+1, k+2

// Where round k takes its mult / add entries from.  GenLoad: the tables in HBM.  GenP1 / GenLiu: computed on the spot from
// the target-sorted contribution lists (the phase-1 init / the Liu gather), so that these tables are never written at full
// length nor read back: the first fold launch of a sumcheck IS its init.  Rows with more than VP_LIGHT_MAX contributions
// were summed by the chunk kernels into the mult/add arrays beforehand and are read from there.  A GenP1 job can also carry
// the inner product V_u = sum eq(r_u,u) V[u] of its layer (it has V[u] in registers anyway).
struct GenLoad { static constexpr int MODE = 0; };
struct GenP1 {
    static constexpr int MODE = 1;
    const InitArgs2 *a; Half dot_h; F *dot_part;
    __device__ __forceinline__ void row(u32 row, u32 valid, F &m, F &ad) const {
        m = f_zero(); ad = f_zero();
        if (row >= valid || row >= a->n_rows) return;
        const u32 b = a->rowptr[row], e = a->rowptr[row + 1];
        if (e - b > VP_LIGHT_MAX) { m = a->M[row]; ad = a->A[row]; return; }
        for (u32 k = b; k < e; ++k) contrib2<1>(*a, k, m, ad, f_zero());
    }
    // rows r0 and r0 + 1 together: row pointers, then the first contribution of both rows, then their operands, then the
    // arithmetic — two dependent chains in flight per lane instead of one (the fused launch runs at 3 waves per SIMD and is
    // bound by exactly this gather latency)
    // row pointers of rows r0, r0 + 1 (b0 | e0 = b1 | e1); rows past the end read as empty.
    __device__ __forceinline__ void ptrs(u32 r0, u32 valid, u32 &b0, u32 &e0, u32 &e1) const {
        const u32 lim = min(valid, a->n_rows);
        b0 = e0 = e1 = 0;
        if (r0 >= lim) return;
        b0 = a->rowptr[r0]; e0 = a->rowptr[r0 + 1];
        e1 = r0 + 1 < lim ? a->rowptr[r0 + 2] : e0;
    }
    __device__ __forceinline__ void row2(u32 r0, u32 valid, u32 b0, u32 e0, u32 e1, F &m0, F &a0, F &m1, F &a1) const {
        m0 = f_zero(); a0 = f_zero(); m1 = f_zero(); a1 = f_zero();
        const u32 lim = min(valid, a->n_rows);
        if (r0 >= lim) return;
        const bool heavy0 = e0 - b0 > VP_LIGHT_MAX, heavy1 = e1 - e0 > VP_LIGHT_MAX;
        const bool f0 = !heavy0 && e0 > b0, f1 = !heavy1 && e1 > e0;
        P1Entry c0{}, c1{};
        if (f0) c0 = p1_load(*a, b0);
        if (f1) c1 = p1_load(*a, e0);
        P1Vals v0{}, v1{};
        if (f0) v0 = p1_gather(*a, c0);
        if (f1) v1 = p1_gather(*a, c1);
        if (heavy0) { m0 = a->M[r0]; a0 = a->A[r0]; }
        if (heavy1) { m1 = a->M[r0 + 1]; a1 = a->A[r0 + 1]; }
        if (f0) p1_apply(*a, c0, v0, m0, a0);
        if (f1) p1_apply(*a, c1, v1, m1, a1);
        if (!heavy0) for (u32 k = b0 + 1; k < e0; ++k) contrib2<1>(*a, k, m0, a0, f_zero());
        if (!heavy1) for (u32 k = e0 + 1; k < e1; ++k) contrib2<1>(*a, k, m1, a1, f_zero());
    }
};
struct GenLiu {
    static constexpr int MODE = 2;
    const u32 *rowptr; const uint8_t *e_q; const u32 *e_g; const Half *H;
    __device__ __forceinline__ void row(u32 u, u32 valid, F &m, F &ad) const {
        ad = f_zero(); m = f_zero();
        if (u >= valid) return;
        m = half_at(H[0], u);
        for (u32 k = rowptr[u]; k < rowptr[u + 1]; ++k) m = f_add(m, half_at(H[e_q[k]], e_g[k]));
    }
    __device__ __forceinline__ void ptrs(u32 u0, u32 valid, u32 &b0, u32 &e0, u32 &e1) const {
        b0 = e0 = e1 = 0;
        if (u0 >= valid) return;
        b0 = rowptr[u0]; e0 = rowptr[u0 + 1];
        e1 = u0 + 1 < valid ? rowptr[u0 + 2] : e0;
    }
    __device__ __forceinline__ void row2(u32 u0, u32 valid, u32 b0, u32 e0, u32 e1, F &m0, F &a0, F &m1, F &a1) const {     // see GenP1::row2
        m0 = f_zero(); a0 = f_zero(); m1 = f_zero(); a1 = f_zero();
        if (u0 >= valid) return;
        const bool ok1 = u0 + 1 < valid;
        const Half h0 = H[0];
        // base terms eq(r_u, u): independent of the lists
        const F bf0 = h0.bf[u0 & ((1u << h0.h1) - 1)], bs0 = h0.bs[u0 >> h0.h1];
        F bf1 = f_zero(), bs1 = f_zero();
        if (ok1) { bf1 = h0.bf[(u0 + 1) & ((1u << h0.h1) - 1)]; bs1 = h0.bs[(u0 + 1) >> h0.h1]; }
        const bool f0 = e0 > b0, f1 = e1 > e0;
        u32 q0 = 0, g0 = 0, q1 = 0, g1 = 0;
        if (f0) { q0 = e_q[b0]; g0 = e_g[b0]; }
        if (f1) { q1 = e_q[e0]; g1 = e_g[e0]; }
        Half hq0 = h0, hq1 = h0;
        if (f0) hq0 = H[q0];
        if (f1) hq1 = H[q1];
        F xf0 = f_zero(), xs0 = f_zero(), xf1 = f_zero(), xs1 = f_zero();
        if (f0) { xf0 = hq0.bf[g0 & ((1u << hq0.h1) - 1)]; xs0 = hq0.bs[g0 >> hq0.h1]; }
        if (f1) { xf1 = hq1.bf[g1 & ((1u << hq1.h1) - 1)]; xs1 = hq1.bs[g1 >> hq1.h1]; }
        // base term + first list term as ONE two-product sum (one reduction per limb; a row without a list adds 0 * 0), the further terms as a*b + c
        m0 = f_dot2cc<VP_MADSHIFT != 0>(bf0, bs0, xf0, xs0);
        if (ok1) m1 = f_dot2cc<VP_MADSHIFT != 0>(bf1, bs1, xf1, xs1);
        for (u32 k = b0 + 1; k < e0; ++k) m0 = half_mad(H[e_q[k]], e_g[k], m0);
        for (u32 k = e0 + 1; k < e1; ++k) m1 = half_mad(H[e_q[k]], e_g[k], m1);
    }
};

// Phase 2 (round 4): the mult / add entries of slot `row` from the v-sorted contribution list (contrib2<2>: eq(r_liu, g) eq(r_u, u) and V_u), and the
// slot's V entry gathered through the slot map, inside the first fold launch of the sumcheck — the three tables of the long subsets are never
// written nor read back (x1024: 48 B written + 48 B read per slot).  The tables of a phase 2 lie side by side in one slot space (t_off), so a
// row index is the global slot; the short subsets (< one chunk) keep their light-row init (they are folded by k_seg / k_emit from memory).
struct GenP2 {
    static constexpr int MODE = 3;
    const InitArgs2 *a;
    __device__ __forceinline__ F vrow(u32 row, u32 vend) const {
        if (row >= vend || row >= a->n_rows) return f_zero();
        const int l = a->s_layer[row];
        if (l == 0xff || l == 0xfe) return f_zero();
        const u32 x = a->s_idx[row];
        if (!VP_CHK((unsigned) l < g_vp_chk_layers() && x < g_vp_chk_lsize(l), 2, l, x, row)) return f_zero();
        return val_at(*a, l, x);
    }
    __device__ __forceinline__ void ptrs(u32 r0, u32 valid, u32 &b0, u32 &e0, u32 &e1) const {
        const u32 lim = min(valid, a->n_rows);
        b0 = e0 = e1 = 0;
        if (r0 >= lim) return;
        b0 = a->rowptr[r0]; e0 = a->rowptr[r0 + 1];
        e1 = r0 + 1 < lim ? a->rowptr[r0 + 2] : e0;
    }
    __device__ __forceinline__ void row2(u32 r0, u32 valid, u32 b0, u32 e0, u32 e1, F &m0, F &a0, F &m1, F &a1) const {
        m0 = f_zero(); a0 = f_zero(); m1 = f_zero(); a1 = f_zero();
        const u32 lim = min(valid, a->n_rows);
        if (r0 >= lim) return;
        const bool heavy0 = e0 - b0 > VP_LIGHT_MAX, heavy1 = e1 - e0 > VP_LIGHT_MAX;
        if (heavy0) { m0 = a->M[r0]; a0 = a->A[r0]; }
        if (heavy1) { m1 = a->M[r0 + 1]; a1 = a->A[r0 + 1]; }
        const F vu = *a->Vu;
        // the two rows' lists side by side: two dependent chains (record -> half-table entries -> products) in flight per lane
        u32 k0 = heavy0 ? e0 : b0, k1 = heavy1 ? e1 : e0;
        while (k0 < e0 && k1 < e1) { contrib2<2>(*a, k0++, m0, a0, vu); contrib2<2>(*a, k1++, m1, a1, vu); }
        for (; k0 < e0; ++k0) contrib2<2>(*a, k0, m0, a0, vu);
        for (; k1 < e1; ++k1) contrib2<2>(*a, k1, m1, a1, vu);
    }
};

// Measured on this kernel (tools/micro_sumfold.hip and its -DVP_EXP_* probes, 2^24 entries, profiles/r01_j_micro_*.txt): 327 us
// as is; 257 us with the global loads replaced by synthesised values (pure instruction issue: ~758 VALU instructions per wave
// pair-step, 528 of them in the six multiply-adds); 175 us with the multiply-adds replaced by three cheap ops (memory + LDS +
// barriers: 4.9 TB/s).  Built, measured and rejected because they did not beat it: rotating the wave roles per chunk so that
// every wave issues 7 steps per four chunks (+-5 %, +3 spilled VGPRs); requesting the next chunk's six entries right after
// round k (needs 140 VGPRs: 342 us at 3 waves/SIMD, 385 us spilling at 4); 3 instead of 4 workgroups per CU (same); a
// role-split variant in the manner of k_seg (768-thread workgroup, chunk staged in LDS, two multiply-adds per lane and pair, 64
// VGPRs, 6 waves per SIMD; kept in tools/micro_sumfold.hip): identical outputs, 360 us against 280 us — occupancy is not what
// holds this kernel at ~5.9 cycles per wave-instruction.
template <bool HAS_A, class Gen>
__device__ __forceinline__ void sumfold3b_body(const SfArgs &a, u32 bid, u32 nb, Sf3bLds &sm, const Gen &gen) {
    F (&s1)[3][256] = sm.s1; F (&s2)[3][128] = sm.s2; F (&red)[4][9] = sm.red;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // round k sums stay in registers; those of rounds k+1 / k+2 (first two waves / first wave only) live in LDS, one
    // private slot per thread, so that the kernel fits 128 VGPRs (4 waves per SIMD) without scratch
    Lz acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i].re = acc[i].im = 0;
    if (t < 128) { for (int i = 0; i < 3; ++i) { sm.acc2[i][t].re = 0; sm.acc2[i][t].im = 0; } }
    if (t < 64) { for (int i = 0; i < 3; ++i) { sm.acc3[i][t].re = 0; sm.acc3[i][t].im = 0; } }
    const F r0 = a.r[0], r1 = a.r[1], r2 = a.r[2];
    const bool keep_y0 = (a.keep_y0 & 1) != 0, keep_rest = (a.keep_y0 & 2) != 0;              // uniform
    const bool vreal = (a.keep_y0 & 4) != 0;             // the V entries of this launch's first round are real circuit values (uniform)
    F dacc = f_zero();                                   // GenP1: this thread's share of the V_u inner product
    for (u32 c = bid; c < a.total_chunks; c += nb) {
        int j = 0;
        while (j + 1 < a.n_tab && c >= a.t[j + 1].chunk_start) ++j;
        const SfTab td = a.t[j];
        const u32 cl = c - td.chunk_start;
        const int wl = w, tl = t;
        const u32 i0 = td.off + cl * 512 + 2 * tl, vend = td.off + td.valid;
        {   // round k: one pair per thread
            F v0, v1, m0, m1, a0 = f_zero(), a1 = f_zero();
            if constexpr (Gen::MODE == 3) { v0 = gen.vrow(i0, vend); v1 = gen.vrow(i0 + 1, vend); }      // phase 2: V through the slot map
            else if (vreal && a.inVr) {                                   // uniform: 8-byte real circuit values (i0 is even: one 16-byte load per pair)
                v0 = f_make(i0 < vend ? a.inVr[i0] : 0ull, 0); v1 = f_make(i0 + 1 < vend ? a.inVr[i0 + 1] : 0ull, 0);
            } else {
                // a chunk that lies completely inside the table (all but the last one) is loaded without the per-entry bounds selects
                if (cl * 512 + 512 <= td.valid) { v0 = a.inV[i0]; v1 = a.inV[i0 + 1]; }
                else { v0 = ld_or_zero(a.inV, i0, vend); v1 = ld_or_zero(a.inV, i0 + 1, vend); }
            }
            if constexpr (Gen::MODE == 0) {
                if (cl * 512 + 512 <= td.valid) {                            // uniform
                    m0 = a.inM[i0]; m1 = a.inM[i0 + 1];
                    if (HAS_A) { a0 = a.inA[i0]; a1 = a.inA[i0 + 1]; }
                } else {
                    m0 = ld_or_zero(a.inM, i0, vend); m1 = ld_or_zero(a.inM, i0 + 1, vend);
                    if (HAS_A) { a0 = ld_or_zero(a.inA, i0, vend); a1 = ld_or_zero(a.inA, i0 + 1, vend); }
                }
            } else {                      // generated tables (GenP1 / GenLiu: one table per job, offset 0; GenP2: rows = global slots)
                u32 cb0, ce0, ce1;                       // (requesting them one chunk ahead was measured: no gain)
                gen.ptrs(i0, vend, cb0, ce0, ce1);
                gen.row2(i0, vend, cb0, ce0, ce1, m0, a0, m1, a1);
                if constexpr (Gen::MODE == 1) {
                    if (gen.dot_part) {
                        if (vreal) {
                            if (i0 < vend) dacc = f_mad31c_rb<false>(half_at(gen.dot_h, i0), v0.re, dacc);
                            if (i0 + 1 < vend) dacc = f_mad31c_rb<false>(half_at(gen.dot_h, i0 + 1), v1.re, dacc);
                        } else {
                            if (i0 < vend) dacc = f_add(dacc, f_mul(half_at(gen.dot_h, i0), v0));
                            if (i0 + 1 < vend) dacc = f_add(dacc, f_mul(half_at(gen.dot_h, i0 + 1), v1));
                        }
                    }
                }
            }
            F vo, mo, ao = f_zero();
            if (vreal) sf_pair_step_rv<HAS_A>(v0.re, v1.re, m0, m1, a0, a1, r0, acc[0], acc[1], acc[2], vo, mo, ao, keep_y0);
            else sf_pair_step<HAS_A>(v0, v1, m0, m1, a0, a1, r0, acc[0], acc[1], acc[2], vo, mo, ao, keep_y0);
            s1[0][tl] = vo; s1[1][tl] = mo;
            if (HAS_A) s1[2][tl] = ao;
        }
        __syncthreads();
        if (wl < 2) {   // round k+1: 128 pairs
            F vo, mo, ao = f_zero();
            Lz x = sm.acc2[0][tl], y{0, 0}, z = sm.acc2[2][tl];
            if (keep_rest) y = sm.acc2[1][tl];
            sf_pair_step<HAS_A>(s1[0][2 * tl], s1[0][2 * tl + 1], s1[1][2 * tl], s1[1][2 * tl + 1],
                                HAS_A ? s1[2][2 * tl] : f_zero(), HAS_A ? s1[2][2 * tl + 1] : f_zero(), r1,
                                x, y, z, vo, mo, ao, keep_rest);
            lz_fold(x); lz_fold(z);
            sm.acc2[0][tl] = x; sm.acc2[2][tl] = z;
            if (keep_rest) { lz_fold(y); sm.acc2[1][tl] = y; }
            s2[0][tl] = vo; s2[1][tl] = mo;
            if (HAS_A) s2[2][tl] = ao;
        }
        __syncthreads();
        if (wl == 0) {  // round k+2: 64 pairs, results are the folded table
            F vo, mo, ao = f_zero();
            Lz x = sm.acc3[0][lane], y{0, 0}, z = sm.acc3[2][lane];
            if (keep_rest) y = sm.acc3[1][lane];
            sf_pair_step<HAS_A>(s2[0][2 * lane], s2[0][2 * lane + 1], s2[1][2 * lane], s2[1][2 * lane + 1],
                                HAS_A ? s2[2][2 * lane] : f_zero(), HAS_A ? s2[2][2 * lane + 1] : f_zero(), r2,
                                x, y, z, vo, mo, ao, keep_rest);
            lz_fold(x); lz_fold(z);
            sm.acc3[0][lane] = x; sm.acc3[2][lane] = z;
            if (keep_rest) { lz_fold(y); sm.acc3[1][lane] = y; }
            const u32 oi = cl * 64 + lane;
            if (oi < ((td.valid + 7) >> 3)) {
                if (VP_CHK((unsigned long long) td.off + oi < g_vp_chk_cap(), 5, td.off, oi, 0)) {     // (a failed check skips the store, never the barriers below)
                    a.outV[td.off + oi] = vo;
                    a.outM[td.off + oi] = mo;
                    if (HAS_A) a.outA[td.off + oi] = ao;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) lz_fold(acc[i]);
    }
    // block partials: the sums of rounds k+1 and k+2 sit in the LDS slots of threads 0-127 and 0-63
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i >= 3 && w >= 2) break;
        if (i >= 6 && w >= 1) break;
        const Lz v = i < 3 ? acc[i] : i < 6 ? sm.acc2[i - 3][t] : sm.acc3[i - 6][t];
        const F x = wave_sum63(lz_canon(v));
        if (lane == 63) red[w][i] = x;
    }
    __syncthreads();
    if (t < 3) {
        const int nw = t == 0 ? 4 : t == 1 ? 2 : 1;
        F X = red[0][3 * t], Y = red[0][3 * t + 1], Z = red[0][3 * t + 2];
        for (int k = 1; k < nw; ++k) { X = f_add(X, red[k][3 * t]); Y = f_add(Y, red[k][3 * t + 1]); Z = f_add(Z, red[k][3 * t + 2]); }
        F *o = a.part + (size_t) t * a.part_stride + bid * 3;
        o[0] = X; o[1] = (t == 0 ? keep_y0 : keep_rest) ? f_sub(f_sub(Y, X), Z) : f_zero(); o[2] = Z;       // b of the other rounds: derived by k_emit
    }
    if constexpr (Gen::MODE == 1) {
        if (gen.dot_part) {                              // uniform per launch
            F d[1] = {dacc};
            block_sum<1>(d, sm.dred);
            if (t == 0) gen.dot_part[bid] = d[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_sumfold3c (round 6): the same three rounds per launch with NO idle waves, no LDS traffic and no workgroup barrier inside the chunk loop.
// k_sumfold3b parks waves by construction: of a chunk's four waves two sit out round k+1 and three round k+2 at its two barriers (7 of 12
// wave-slots busy; SQ_WAIT_ANY 0.56 of the wave cycles, profiles/r05_pmc_summary_b1024.json).  Here a WAVE owns a 512-entry chunk: every
// lane folds four pairs in round k (four independent pair steps), the partner entries of round k+1 sit in lane ^ 32 and arrive with
// v_permlane32_swap (one instruction transposes a 2 x 2 block of registers between the wave's halves: afterwards BOTH lanes hold a complete
// pair, each of a different local slot), two pair steps per lane; round k+2's partners sit in lane ^ 16: v_permlane16_swap, one pair step.
// 7 pair steps per lane and 8 entries, as many instructions as the dense schedule, every wave busy in every round.
//   pair p (0..255) of the chunk = entries 2p, 2p+1;  lane l, local slot k (0..3) holds  p = l5 | l4 << 1 | (l & 15) << 2 | (k >> 1) << 6 | (k & 1) << 7:
//   round k+1 pairs p with p ^ 1 (lane ^ 32, same slot), round k+2 pairs (p >> 1) with (p >> 1) ^ 1 (lane ^ 16), and lane l ends with the folded
//   entry p >> 2 = l of the chunk (slot k = l5 | l4 << 1 is the one whose results landed in this lane): the stores are one 1 KiB run per wave.
// A load instruction still covers a contiguous 2 KiB (slot k fixed), in permuted lane order.  The sums of the three rounds stay in registers
// (lazy, folded once per chunk) and are reduced once per wave at the end of the launch.
// MEASURED (profiles/r06_ab_sumfold3c_x1024_x64_randomize.txt, same call, transcripts identical): SLOWER — x1024 proof 6.43-6.62 -> 7.14-7.16 ms, the
// fused-init fold launches 5.03 -> 5.91 ms: four pairs and nine lazy sums per lane are 168 registers (10 / 58 dwords spilled) = 3 waves per SIMD against 3b's 4,
// and at 2 waves without spills 8.1 ms.  The fold is bound by the issue of dependent v_mad_u64_u32 chains; resident waves hide that, work per lane does not.
// Kept as the A/B partner (VP_SF3C=1), off by default.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32_u64(u64 &a, u64 &b) {        // lanes 32-63 of a <-> lanes 0-31 of b
    u32 al = (u32) a, ah = (u32) (a >> 32), bl = (u32) b, bh = (u32) (b >> 32);
    auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false); al = r0[0]; bl = r0[1];
    auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false); ah = r1[0]; bh = r1[1];
    a = ((u64) ah << 32) | al; b = ((u64) bh << 32) | bl;
}
__device__ __forceinline__ void swap16_u64(u64 &a, u64 &b) {        // rows 1, 3 of a <-> rows 0, 2 of b (rows of 16 lanes)
    u32 al = (u32) a, ah = (u32) (a >> 32), bl = (u32) b, bh = (u32) (b >> 32);
    auto r0 = __builtin_amdgcn_permlane16_swap(al, bl, false, false); al = r0[0]; bl = r0[1];
    auto r1 = __builtin_amdgcn_permlane16_swap(ah, bh, false, false); ah = r1[0]; bh = r1[1];
    a = ((u64) ah << 32) | al; b = ((u64) bh << 32) | bl;
}
__device__ __forceinline__ void swap32_F(F &a, F &b) { swap32_u64(a.re, b.re); swap32_u64(a.im, b.im); }
__device__ __forceinline__ void swap16_F(F &a, F &b) { swap16_u64(a.re, b.re); swap16_u64(a.im, b.im); }

struct Sf3cLds { F red[4][9]; F dred[4]; };

template <bool HAS_A, class Gen>
__device__ __forceinline__ void sumfold3c_body(const SfArgs &a, u32 bid, u32 nb, Sf3cLds &sm, const Gen &gen) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    Lz acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i].re = acc[i].im = 0;
    const F r0 = a.r[0], r1 = a.r[1], r2 = a.r[2];
    const bool keep_y0 = (a.keep_y0 & 1) != 0, keep_rest = (a.keep_y0 & 2) != 0;              // uniform
    const bool vreal = (a.keep_y0 & 4) != 0;             // the V entries of this launch's first round are real circuit values (uniform)
    F dacc = f_zero();                                   // GenP1: this thread's share of the V_u inner product
    const u32 lp = (u32) ((lane >> 5) | (((lane >> 4) & 1) << 1) | ((lane & 15) << 2));
    for (u32 c = bid * 4 + w; c < a.total_chunks; c += nb * 4) {           // wave-uniform: the swaps below run with all lanes on
        int j = 0;
        while (j + 1 < a.n_tab && c >= a.t[j + 1].chunk_start) ++j;
        const SfTab td = a.t[j];
        const u32 cl = c - td.chunk_start;
        const u32 vend = td.off + td.valid;
        const bool full = cl * 512 + 512 <= td.valid;                        // uniform
        // round k of local slot k_: one pair of this lane
        auto round_k = [&](const int k_, F &fv, F &fm, F &fa) {
            const u32 p = lp | (u32) ((k_ >> 1) << 6) | (u32) ((k_ & 1) << 7);
            const u32 i0 = td.off + cl * 512 + 2 * p;
            F v0, v1, m0, m1, a0 = f_zero(), a1 = f_zero();
            if constexpr (Gen::MODE == 3) { v0 = gen.vrow(i0, vend); v1 = gen.vrow(i0 + 1, vend); }      // phase 2: V through the slot map
            else if (vreal && a.inVr) {                                   // uniform: 8-byte real circuit values (i0 is even: one 16-byte load per pair)
                v0 = f_make(i0 < vend ? a.inVr[i0] : 0ull, 0); v1 = f_make(i0 + 1 < vend ? a.inVr[i0 + 1] : 0ull, 0);
            } else if (full) { v0 = a.inV[i0]; v1 = a.inV[i0 + 1]; }
            else { v0 = ld_or_zero(a.inV, i0, vend); v1 = ld_or_zero(a.inV, i0 + 1, vend); }
            if constexpr (Gen::MODE == 0) {
                if (full) {
                    m0 = a.inM[i0]; m1 = a.inM[i0 + 1];
                    if (HAS_A) { a0 = a.inA[i0]; a1 = a.inA[i0 + 1]; }
                } else {
                    m0 = ld_or_zero(a.inM, i0, vend); m1 = ld_or_zero(a.inM, i0 + 1, vend);
                    if (HAS_A) { a0 = ld_or_zero(a.inA, i0, vend); a1 = ld_or_zero(a.inA, i0 + 1, vend); }
                }
            } else {                      // generated tables (GenP1 / GenLiu: one table per job, offset 0; GenP2: rows = global slots)
                u32 cb0, ce0, ce1;
                gen.ptrs(i0, vend, cb0, ce0, ce1);
                gen.row2(i0, vend, cb0, ce0, ce1, m0, a0, m1, a1);
                if constexpr (Gen::MODE == 1) {
                    if (gen.dot_part) {
                        if (vreal) {
                            if (i0 < vend) dacc = f_mad31c_rb<false>(half_at(gen.dot_h, i0), v0.re, dacc);
                            if (i0 + 1 < vend) dacc = f_mad31c_rb<false>(half_at(gen.dot_h, i0 + 1), v1.re, dacc);
                        } else {
                            if (i0 < vend) dacc = f_add(dacc, f_mul(half_at(gen.dot_h, i0), v0));
                            if (i0 + 1 < vend) dacc = f_add(dacc, f_mul(half_at(gen.dot_h, i0 + 1), v1));
                        }
                    }
                }
            }
            fa = f_zero();
            if (vreal) sf_pair_step_rv<HAS_A>(v0.re, v1.re, m0, m1, a0, a1, r0, acc[0], acc[1], acc[2], fv, fm, fa, keep_y0);
            else sf_pair_step<HAS_A>(v0, v1, m0, m1, a0, a1, r0, acc[0], acc[1], acc[2], fv, fm, fa, keep_y0);
        };
        // rounds k and k+1 of two local slots (s, s + 1): after the swap both lanes of a pair (l, l ^ 32) hold one complete pair of round k+1
        auto two_slots = [&](const int s_, F &ov, F &om, F &oa) {
            F xv, xm, xa, yv, ym, ya;
            round_k(s_, xv, xm, xa);
            round_k(s_ + 1, yv, ym, ya);
            swap32_F(xv, yv); swap32_F(xm, ym);
            if (HAS_A) swap32_F(xa, ya);
            oa = f_zero();
            sf_pair_step<HAS_A>(xv, yv, xm, ym, xa, ya, r1, acc[3], acc[4], acc[5], ov, om, oa, keep_rest);
        };
        F ev, em, ea, gv, gm, ga;
        two_slots(0, ev, em, ea);
        two_slots(2, gv, gm, ga);
        // round k+2: partners in lane ^ 16; the result is entry `lane` of the chunk's 64 folded entries
        swap16_F(ev, gv); swap16_F(em, gm);
        if (HAS_A) swap16_F(ea, ga);
        F vo, mo, ao = f_zero();
        sf_pair_step<HAS_A>(ev, gv, em, gm, ea, ga, r2, acc[6], acc[7], acc[8], vo, mo, ao, keep_rest);
        const u32 oi = cl * 64 + lane;
        if (oi < ((td.valid + 7) >> 3)) {
            if (VP_CHK((unsigned long long) td.off + oi < g_vp_chk_cap(), 5, td.off, oi, 0)) {
                a.outV[td.off + oi] = vo;
                a.outM[td.off + oi] = mo;
                if (HAS_A) a.outA[td.off + oi] = ao;
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) lz_fold(acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const F x = wave_sum63(lz_canon(acc[i]));
        if (lane == 63) sm.red[w][i] = x;
    }
    __syncthreads();
    if (t < 3) {
        F X = sm.red[0][3 * t], Y = sm.red[0][3 * t + 1], Z = sm.red[0][3 * t + 2];
        for (int k = 1; k < 4; ++k) { X = f_add(X, sm.red[k][3 * t]); Y = f_add(Y, sm.red[k][3 * t + 1]); Z = f_add(Z, sm.red[k][3 * t + 2]); }
        F *o = a.part + (size_t) t * a.part_stride + bid * 3;
        o[0] = X; o[1] = (t == 0 ? keep_y0 : keep_rest) ? f_sub(f_sub(Y, X), Z) : f_zero(); o[2] = Z;       // b of the other rounds: derived by k_emit
    }
    if constexpr (Gen::MODE == 1) {
        if (gen.dot_part) {                              // uniform per launch
            F d[1] = {dacc};
            block_sum<1>(d, sm.dred);
            if (t == 0) gen.dot_part[bid] = d[0];
        }
    }
}

#ifndef VP_SF_MINB
#define VP_SF_MINB 3          // workgroups per CU the fold kernels are compiled for (register budget 512 / MINB per lane)
#endif
template <bool HAS_A>
__global__ void __launch_bounds__(VP_BLOCK, VP_SF_MINB) k_sumfold3b(SfArgs a) {
    __shared__ Sf3bLds sm;
    sumfold3b_body<HAS_A>(a, blockIdx.x, gridDim.x, sm, GenLoad());
}

// ---------------------------------------------------------------------------------------------------
}  // namespace vp
