// Round 4: the long transforms of the commitment (N = 2^13 .. 2^17 points: RS_polynomial.cpp:26-220 at the sizes of poly_commit.h:41-349) as two
// LDS passes of radix-8 Stockham butterflies with LAZY arithmetic.  Part of the single translation unit vpgpu.hip (included by vp_kernels_pc.h).
//
//   N = N1 x N2, N2 = 512.   j = j1 N2 + j2,  k = k1 + N1 k2:   w_N^(jk) = w_N1^(j1 k1) . w_N^(j2 k1) . w_N2^(j2 k2)
//   k_ntt8_cols (pass A): a workgroup holds 4096 / N1 neighbouring COLUMNS j2 of one (row, coset) in LDS ([j1][column], columns fastest: every
//       global and LDS access of a wave is a run of >= 256 contiguous bytes) and runs the N1-point transform down each of them; the first pass
//       reads its eight inputs straight from global memory (the coset twist w_M^(j1 N2 b) of the encoder rides on that load), the last pass
//       multiplies by w_N^(j2 k1) (times the column's share w_M^(j2 b) of the twist: ONE root w_M^(j2 (32 k1 + b))) and stores [k1][j2].
//   k_ntt8_rows (pass B): a workgroup holds 8 neighbouring ROWS k1 (512 contiguous elements each), one wave per row: 64 lanes x 8 points, three
//       radix-8 passes, then the 8 x 512 tile goes out in natural order k1 + N1 k2 — eight consecutive k1 per k2: whole 128-byte lines.
//
// What is different from k_ntt_lds / k_ntt_split (kept for N <= 2^12 and as the cross-check, vp_options.ntt_r8 = 0):
//   * radix 8 with the eighth root for free.  p = 2^61 - 1 has sqrt(2) = 2^31, so w_8 = 2^30 (1 -+ i): a multiplication by it is two additions and
//     two 61-bit rotations, by w_4 = -+i a swap and a negation.  Seven twiddle multiplications per eight points and THREE stages (radix 4: three
//     per four points and two stages), none in the first pass of a transform; three LDS round trips for 512 points instead of five.
//   * lazy butterflies.  Limbs travel weakly reduced (< 2^61 + 8, congruent mod p): sums and differences of the three butterfly levels are plain
//     64-bit additions (a difference adds a multiple of p first), folded (x & p) + (x >> 61) once after the second level and once at the end —
//     no compare / subtract / select anywhere; the products are the weak form of f_mad31c (one fold, no conditional subtraction).  Only the last
//     store of a transform canonicalises.  (k_ntt_lds: eight canonical add/sub of ~20 instructions per radix-4 butterfly.)
//   * Stockham ordering: natural order in, natural order out, no bit reversal anywhere; an in-place pass is "all reads, barrier, all writes".
//   * the second pass stores whole lines by itself: no dependence on which XCD the neighbouring sub-transform ran on.
#pragma once

namespace vp {

constexpr u64 LZ_P2 = 2 * P61;              // 2^62 - 2
constexpr u64 LZ_P4 = 4 * P61;              // 2^63 - 4
__device__ __forceinline__ u64 lz_fold(u64 x) { return (x & P61) + (x >> 61); }
__device__ __forceinline__ F lz_fold(const F &x) { return f_make(lz_fold(x.re), lz_fold(x.im)); }
__device__ __forceinline__ F lz_add(const F &a, const F &b) { return f_make(a.re + b.re, a.im + b.im); }
// a - b (mod p) as a + (K - b), K a multiple of p with K >= every limb of b
template <u64 K> __device__ __forceinline__ F lz_sub(const F &a, const F &b) { return f_make(a.re + (K - b.re), a.im + (K - b.im)); }
// x * w_4, w_4 = -i (forward) / +i (inverse); K as above for the limbs of x
template <bool INV, u64 K> __device__ __forceinline__ F lz_mul_w4(const F &x) { return INV ? f_make(K - x.im, x.re) : f_make(x.im, K - x.re); }
// v * 2^30 (mod p) for any v < 2^64: v = h 2^31 + l  ->  h + l 2^30  (< 2^61 + 2^33)
__device__ __forceinline__ u64 lz_rot30(u64 v) { return (v >> 31) + ((v & 0x7fffffffull) << 30); }
// x * w_8, w_8 = 2^30 (1 - i) (forward: the reference's w_M^(M/8), fieldElement.cpp:237-249) / 2^30 (1 + i) (inverse); limbs of x < 2^61 + 8
template <bool INV> __device__ __forceinline__ F lz_mul_w8(const F &x) {
    const u64 s = x.re + x.im;
    return INV ? f_make(lz_rot30(x.re + (LZ_P2 - x.im)), lz_rot30(s)) : f_make(lz_rot30(s), lz_rot30(x.im + (LZ_P2 - x.re)));
}
__device__ __forceinline__ F lz_canon(const F &x) {                      // weakly reduced (< 2^62) -> canonical
    u64 a = lz_fold(x.re), b = lz_fold(x.im);
    return f_make(a >= P61 ? a - P61 : a, b >= P61 ? b - P61 : b);
}
// root (canonical) x data (limbs < 2^62): weak product, limbs < 2^61 + 4
__device__ __forceinline__ F lz_mul(const F &root, const F &x) { return f_mad31c<true, false>(root, x, f_make(0, 0)); }
// Round 6: the same product with a PRE-SPLIT root.  The root tables of k_ntt8_colsx / k_ntt8_rows hold each limb w = hi 2^31 + lo as the word (hi << 32) | lo
// (still 16 bytes per root: lz_presplit): the two 31-bit halves are the two dwords of the register pair, no instruction splits them; and the negated imaginary
// part the real limb needs is taken on the ROOT side — p - w.im = (2^30 - 1 - hi) 2^31 + (2^31 - 1 - lo), two 32-bit xors, no borrow — instead of a 64-bit
// subtraction and a third split on the data side.  Per product 3 v_alignbit_b32 (half rate), 3 v_and_b32 and a v_sub_co / v_subb_co pair less; the same field
// element (the weak form's value depends on the operands only through their residues and the split is exact).
#ifndef VP_NTT_PS
#define VP_NTT_PS 1
#endif
__host__ __device__ __forceinline__ F lz_presplit(const F &w) {          // canonical root -> packed halves
    return f_make(((w.re >> 31) << 32) | (w.re & 0x7fffffffull), ((w.im >> 31) << 32) | (w.im & 0x7fffffffull));
}
__device__ __forceinline__ F lz_mul_ps(const F &wp, const F &x) {
    Sp31 ar, ai, nai;
    ar.lo = (u32) wp.re; ar.hi = (u32) (wp.re >> 32);
    ai.lo = (u32) wp.im; ai.hi = (u32) (wp.im >> 32);
    nai.lo = ai.lo ^ 0x7fffffffu; nai.hi = ai.hi ^ 0x3fffffffu;          // p - w.im in [0, p]: hi < 2^30 as dot2_31c wants
    const Sp31 br = split31(x.re), bi = split31(x.im);
    return f_make(dot2_31c<true, false>(ar, br, nai, bi, 0), dot2_31c<true, false>(ar, bi, ai, br, 0));
}
// root x data through whichever form the tables of this build hold
__device__ __forceinline__ F lz_mul_t(const F &w, const F &x) {
#if VP_NTT_PS
    return lz_mul_ps(w, x);
#else
    return lz_mul(w, x);
#endif
}
__global__ void __launch_bounds__(256) k_root_presplit(const F *__restrict__ in, F *__restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lz_presplit(in[i]);
}

// Eight-point DFT in place: u[m'] <- sum_m u[m] w_8^(m m').  Inputs: limbs < 2^61 + 8.  Outputs: limbs < 2^61 + 8 (folded), congruent mod p.
template <bool INV> __device__ __forceinline__ void lz_dft8(F (&u)[8]) {
    // level 1 (pairs m, m + 4):                         sums < 2^62 + 16, differences < 3 * 2^61
    const F a0 = lz_add(u[0], u[4]), a1 = lz_sub<LZ_P2>(u[0], u[4]), a2 = lz_add(u[2], u[6]), a3 = lz_sub<LZ_P2>(u[2], u[6]);
    const F a4 = lz_add(u[1], u[5]), a5 = lz_sub<LZ_P2>(u[1], u[5]), a6 = lz_add(u[3], u[7]), a7 = lz_sub<LZ_P2>(u[3], u[7]);
    // level 2 (4-point transforms of the even and of the odd inputs):   every limb < 7 * 2^61 < 2^64, then folded to < 2^61 + 8
    const F t3 = lz_mul_w4<INV, LZ_P4>(a3), t7 = lz_mul_w4<INV, LZ_P4>(a7);          // limbs <= 2^63 - 4
    const F e0 = lz_fold(lz_add(a0, a2)), e2 = lz_fold(lz_sub<LZ_P4>(a0, a2)), e1 = lz_fold(lz_add(a1, t3)), e3 = lz_fold(lz_sub<LZ_P4>(a1, t3));
    const F o0 = lz_fold(lz_add(a4, a6)), o2 = lz_fold(lz_sub<LZ_P4>(a4, a6)), o1 = lz_fold(lz_add(a5, t7)), o3 = lz_fold(lz_sub<LZ_P4>(a5, t7));
    // level 3: X[k] = E[k] + w_8^k O[k], X[k + 4] = E[k] - w_8^k O[k];  w_8^k O[k] has limbs <= 2^62 - 2
    const F q1 = lz_mul_w8<INV>(o1), q2 = lz_mul_w4<INV, LZ_P2>(o2), q3 = lz_mul_w4<INV, LZ_P2>(lz_mul_w8<INV>(o3));
    u[0] = lz_fold(lz_add(e0, o0)); u[4] = lz_fold(lz_sub<LZ_P2>(e0, o0));
    u[1] = lz_fold(lz_add(e1, q1)); u[5] = lz_fold(lz_sub<LZ_P2>(e1, q1));
    u[2] = lz_fold(lz_add(e2, q2)); u[6] = lz_fold(lz_sub<LZ_P2>(e2, q2));
    u[3] = lz_fold(lz_add(e3, q3)); u[7] = lz_fold(lz_sub<LZ_P2>(e3, q3));
}
// Four-point DFT in place (same bounds in and out)
template <bool INV> __device__ __forceinline__ void lz_dft4(F (&u)[4]) {
    const F a0 = lz_add(u[0], u[2]), a1 = lz_sub<LZ_P2>(u[0], u[2]), a2 = lz_add(u[1], u[3]), a3 = lz_sub<LZ_P2>(u[1], u[3]);
    const F t3 = lz_mul_w4<INV, LZ_P4>(a3);
    u[0] = lz_fold(lz_add(a0, a2)); u[2] = lz_fold(lz_sub<LZ_P4>(a0, a2)); u[1] = lz_fold(lz_add(a1, t3)); u[3] = lz_fold(lz_sub<LZ_P4>(a1, t3));
}
__device__ __forceinline__ void lz_dft2(F (&u)[2]) {
    const F s = lz_fold(lz_add(u[0], u[1])), d = lz_fold(lz_sub<LZ_P2>(u[0], u[1]));
    u[0] = s; u[1] = d;
}

struct Ntt8Args {
    const F *in; F *out;            // cols: source rows -> scratch [row x coset][k1][j2];   rows: scratch -> destination (natural order)
    const F *RT; u32 half_m;        // FULL circle of order M: w_M^e, e < M = 2 half_m (no sign logic at the gathers): the coset twist and the w_N^(j2 k1)
                                    // twiddles between the passes.  Exponents are formed in 32-bit arithmetic: M divides 2^32, so wrapping is harmless
    const F *RTn;                   // FULL circle of the pass's own order (N1 entries for cols, 512 for rows): w^k, k < order
    int ln, l1;                     // N = 2^ln, N1 = 2^l1 (4 <= l1 <= 8), N2 = 2^(ln - l1) = 512
    u32 in_stride;                  // cols: elements between input rows
    u32 ncoset;                     // forward encoder: cosets per row (blockIdx.z); otherwise 1
    int twist;                      // cols: multiply input j by w_M^(j coset)
    int do_scale; F scale;          // rows: multiply by N^-1 on the way out (inverse transforms)
    // Two REAL sequences per complex transform (commit_private of a real witness, round 4): row p carries slices p and p + pair_rows as x_p + i x_(p + pair_rows).
    //   cols, inverse: in2 != nullptr — the eight inputs of the first pass are (in[..].re, in2[..].re); the output E = c + i c' (coefficients of both slices);
    //   rows, forward: pair_rows != 0 — a transform of E on coset b gives Y = E_0 + sigma_b (rho + i rho'), rho, rho' REAL (the coefficients of a real sequence are
    //     Hermitian, c_(N-j) = conj(c_j), and x^N = zeta_b = sigma_b^2 on the coset), so with Z = zeta_b conj(Y - E_0):
    //     slice p: Re(E_0) + ((Y - E_0) + Z) / 2,   slice p + pair_rows: Im(E_0) + (-i) ((Y - E_0) - Z) / 2   — the same field elements as two transforms.
    const F *in2; u32 pair_rows; const F *e0;        // e0: E_0 of row p at e0[p << ln] (the coefficient array the transform reads)
};

constexpr u32 NTT8_TILE = 4096;     // elements of a workgroup's tile (64 KiB of LDS; two workgroups per CU)
constexpr u32 NTT8_THREADS = 512;

// ---- pass A: N1-point transforms down 4096 / N1 neighbouring columns --------------------------------------------------------------------
// Stockham pass of radix R at sub-length s over a transform of length n: butterfly q < n / R takes the inputs at q + m n / R, multiplies input m by
// w_(R s)^(k m), k = q mod s, and leaves output m' at (q - k) R + k + m' s.
template <bool INV>
__global__ void __launch_bounds__(NTT8_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_ntt8_cols(Ntt8Args a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    F *L = reinterpret_cast<F *>(smem_raw);
    const u32 l1 = (u32) a.l1, N1 = 1u << l1, lc = 12 - l1, cols = 1u << lc, N2 = 1u << (a.ln - a.l1), M = 2 * a.half_m;
    // grid: rows fastest — the workgroups that run next to each other in time are the SAME tile and coset of different rows, and those gather the
    // same 4096 w_M^(j2 (32 k1 + b)) lines: they come from the XCD's L2 instead of the fabric (PMC, round 4: 9.5 GB fetched per 4.3 GB written before)
    const u32 tid = threadIdx.x, row = blockIdx.x, coset = blockIdx.z;
    const u32 c = tid & (cols - 1), j2 = blockIdx.y * cols + c;
    const F *src = a.in + (size_t) row * a.in_stride + j2;
    F *dst = a.out + (((size_t) row * a.ncoset + coset) << a.ln) + j2;
    const u32 wN = M >> a.ln;                                   // w_N = w_M^wN
    const u32 nr8 = l1 / 3, rem = l1 - 3 * nr8;                 // passes: nr8 of radix 8, then one of radix 4 (rem 2) or 2 (rem 1)
    // ---- first pass (radix 8, s = 1, no twiddles): inputs from global memory, twisted
    {
        const u32 q = tid >> lc;                                // < N1 / 8
        F u[8];
#pragma unroll
        for (u32 m = 0; m < 8; ++m) u[m] = src[(size_t) (q + m * (N1 >> 3)) * N2];
        if (a.in2) {                                            // uniform: two real slices as one complex sequence
            const F *src2 = a.in2 + (size_t) row * a.in_stride + j2;
#pragma unroll
            for (u32 m = 0; m < 8; ++m) u[m].im = src2[(size_t) (q + m * (N1 >> 3)) * N2].re;
        }
        if (a.twist && coset) {
            // the j1 share of w_M^(j coset), j1 = q + m N1 / 8 (j2's share joins the output twiddle): exponents e0 + m step (mod M)
            const u32 e0 = q * N2 * coset, step = (N1 >> 3) * N2 * coset;
#pragma unroll
            for (u32 g = 0; g < 8; g += 4) {                    // four root gathers in flight at a time (register budget: 128 VGPRs, 4 waves per SIMD)
                F w[4];
#pragma unroll
                for (u32 m = 0; m < 4; ++m) w[m] = a.RT[(e0 + (g + m) * step) & (M - 1)];
                loads_first();
#pragma unroll
                for (u32 m = 0; m < 4; ++m) u[g + m] = lz_mul(w[m], u[g + m]);
            }
        }
        lz_dft8<INV>(u);
#pragma unroll
        for (u32 m = 0; m < 8; ++m) if (VP_CHK((((q * 8 + m) << lc) + c) < NTT8_TILE, 6, q, m, c)) L[((q * 8 + m) << lc) + c] = u[m];
    }
    __syncthreads();
    u32 s = 8;
    // ---- middle passes of radix 8 (all but the last pass of the transform)
    const u32 total_passes = nr8 + (rem ? 1 : 0);
    for (u32 p = 1; p + 1 < total_passes; ++p) {
        const u32 q = tid >> lc, k = q & (s - 1);
        F u[8];
#pragma unroll
        for (u32 m = 0; m < 8; ++m) u[m] = L[((q + m * (N1 >> 3)) << lc) + c];
        {
            const u32 st = N1 / (8 * s);                        // w_(8 s) = w_N1^st
            F w[8];
#pragma unroll
            for (u32 m = 1; m < 8; ++m) { const u32 e = (k * m * st) & (N1 - 1); w[m] = a.RTn[INV ? ((N1 - e) & (N1 - 1)) : e]; }
#pragma unroll
            for (u32 m = 1; m < 8; ++m) u[m] = lz_mul(w[m], u[m]);
        }
        lz_dft8<INV>(u);
        __syncthreads();
#pragma unroll
        for (u32 m = 0; m < 8; ++m) L[(((q - k) * 8 + k + m * s) << lc) + c] = u[m];
        __syncthreads();
        s *= 8;
    }
    // ---- last pass: radix 8, 4 or 2; outputs multiplied by w_N^(j2 k1) (and the column's share of the twist) and stored [k1][j2]
    const u32 ec = (a.twist && coset) ? coset : 0u;
    // output k1 takes w_M^(j2 (k1 wN' + coset)), wN' = wN (forward) or M - wN (inverse): exponent base + k1 stepk (mod M), 32-bit
    const u32 wNs = INV ? (M - wN) & (M - 1) : wN;
    const u32 ebase = j2 * ec, stepk = j2 * wNs;
    if (rem == 0) {
        const u32 q = tid >> lc, k = q & (s - 1);               // s = N1 / 8: k = q
        F u[8];
#pragma unroll
        for (u32 m = 0; m < 8; ++m) u[m] = L[((q + m * (N1 >> 3)) << lc) + c];
        {
            const u32 st = N1 / (8 * s);
            F w[8];
#pragma unroll
            for (u32 m = 1; m < 8; ++m) { const u32 e = (k * m * st) & (N1 - 1); w[m] = a.RTn[INV ? ((N1 - e) & (N1 - 1)) : e]; }
#pragma unroll
            for (u32 m = 1; m < 8; ++m) u[m] = lz_mul(w[m], u[m]);
        }
        lz_dft8<INV>(u);
        const u32 k1_0 = (q - k) * 8 + k, eo = ebase + k1_0 * stepk, so = s * stepk;
#pragma unroll
        for (u32 g = 0; g < 8; g += 4) {
            F w[4];
#pragma unroll
            for (u32 m = 0; m < 4; ++m) w[m] = a.RT[(eo + (g + m) * so) & (M - 1)];
            loads_first();
#pragma unroll
            for (u32 m = 0; m < 4; ++m) dst[(size_t) (k1_0 + (g + m) * s) * N2] = lz_mul(w[m], u[g + m]);     // weakly reduced: pass B folds
        }
    } else if (rem == 2) {
        // radix 4 at s = N1 / 4: two butterflies per thread
#pragma unroll
        for (u32 h = 0; h < 2; ++h) {
            const u32 t = tid + h * NTT8_THREADS, q = t >> lc, k = q & (s - 1);
            F u[4];
#pragma unroll
            for (u32 m = 0; m < 4; ++m) u[m] = L[((q + m * (N1 >> 2)) << lc) + c];
            const u32 st = N1 / (4 * s);
            F w[4];
#pragma unroll
            for (u32 m = 1; m < 4; ++m) { const u32 e = (k * m * st) & (N1 - 1); w[m] = a.RTn[INV ? ((N1 - e) & (N1 - 1)) : e]; }
#pragma unroll
            for (u32 m = 1; m < 4; ++m) u[m] = lz_mul(w[m], u[m]);
            lz_dft4<INV>(u);
            const u32 k1_0 = (q - k) * 4 + k, eo = ebase + k1_0 * stepk, so = s * stepk;
            F ww[4];
#pragma unroll
            for (u32 m = 0; m < 4; ++m) ww[m] = a.RT[(eo + m * so) & (M - 1)];
            loads_first();
#pragma unroll
            for (u32 m = 0; m < 4; ++m) dst[(size_t) (k1_0 + m * s) * N2] = lz_mul(ww[m], u[m]);
        }
    } else {
        // radix 2 at s = N1 / 2: four butterflies per thread
#pragma unroll
        for (u32 h = 0; h < 4; ++h) {
            const u32 t = tid + h * NTT8_THREADS, q = t >> lc, k = q & (s - 1);
            F u[2];
            u[0] = L[(q << lc) + c]; u[1] = L[((q + (N1 >> 1)) << lc) + c];
            { const u32 e = k & (N1 - 1); u[1] = lz_mul(a.RTn[INV ? ((N1 - e) & (N1 - 1)) : e], u[1]); }      // w_(2 s)^k = w_N1^k
            lz_dft2(u);
            const u32 k1_0 = (q - k) * 2 + k, eo = ebase + k1_0 * stepk, so = s * stepk;
            F ww[2];
#pragma unroll
            for (u32 m = 0; m < 2; ++m) ww[m] = a.RT[(eo + m * so) & (M - 1)];
            loads_first();
#pragma unroll
            for (u32 m = 0; m < 2; ++m) dst[(size_t) (k1_0 + m * s) * N2] = lz_mul(ww[m], u[m]);
        }
    }
}

// ---- pass A of the ENCODER (forward, rate 1 / ncoset), round 5: one workgroup runs ALL (or a chunk of) the cosets of its tile ---------------------
// k_ntt8_cols reads its 64 KB input tile once per coset (32 times) and gathers, per coset, 4096 roots w_M^(j2 (ncoset k1 + b)) that no other tile or coset
// shares — the whole 64 MB circle of order M once per row (PMC, round 4: 2.05 bytes fetched per byte written, SQ_WAIT_ANY 0.36).  Here
//   * the eight inputs of a thread are loaded ONCE and stay in registers for every coset of the workgroup (the twist changes, the data does not);
//   * the twiddle between the passes w_N^(j2 k1) does not depend on the coset: eight roots per thread, loaded once, in registers;
//   * the column's share w_M^(j2 b) of the twist is common to every element of the column, so it commutes with the column transform: it rides on the
//     twiddles of the SECOND pass (s = 8: w_64^(k m), 64 values) as ONE table T2[b][tile][m][k][c] = w_M^(j2 b) w_64^(k m) — 64 ncoset N2 entries (16 MB at
//     x1024) read as whole 1 KB runs per wave, shared by every row — at the price of the m = 0 product (8 instead of 7 multiplications in that pass:
//     30 instead of 29 per eight points for N1 = 256);
//   * the j1 share of the twist w_M^(j1 N2 b) = w_(ncoset N1)^(j1 b) comes from the compact circle of that order (128 KB).
// Per coset a workgroup touches global memory for 8 + 8 cached root loads per thread and its stores; nothing it waits for comes from HBM.
// 256 threads on a 2048-element tile (2048 / N1 columns, 32 KB of LDS), 168 VGPRs: three workgroups per CU at different points of their loops.
// N1 = 64, 128, 256 (N = 2^15 .. 2^17): radix 8, radix 8, then nothing / radix 2 / radix 4 — the Stockham schedule of k_ntt8_cols.
struct Ntt8xArgs {
    const F *in; F *out;            // source rows -> scratch [row x coset][k1][j2] (weakly reduced; pass B folds)
    const F *TW;                    // w_(ncoset N1)^e, e < ncoset N1
    const F *T2;                    // see above
    const F *W1;                    // w_N^e, e < N
    const F *RTn;                   // w_N1^e, e < N1 (third pass)
    int ln; u32 in_stride, ncoset, cpw;     // cpw: cosets per workgroup (blockIdx.z = chunk)
};
constexpr u32 NTT8X_THREADS = 256, NTT8X_TILE = 2048;
#ifndef VP_NTT8X_WAVES
#define VP_NTT8X_WAVES 2
#endif
#ifndef VP_NTT8X_G
#define VP_NTT8X_G 4
#endif
constexpr u32 NTT8X_G = VP_NTT8X_G;          // products whose roots are loaded, and which the scheduler may interleave, at a time (register budget: inputs and
                                             // output twiddles of the thread stay in registers, 64 of the 168)
// output slot o < 8 of a thread -> k1 (the Stockham schedule's last pass): N1 = 64: radix 8 at s = 8; N1 = 128 / 256: butterfly h = o / 2 (o / 4) of the radix-2 (radix-4)
// pass at s = 64, output o % 2 (o % 4); q2 = tid >> LC of the first butterfly (the others sit 256 >> LC = N1 / 8 further)
template <int L1> __device__ __forceinline__ u32 ntt8x_k1(u32 q, u32 o) {
    constexpr u32 Q8 = (1u << L1) >> 3;
    if (L1 == 6) return (q - (q & 7)) * 8 + (q & 7) + o * 8;
    if (L1 == 7) return q + (o >> 1) * Q8 + (o & 1) * 64;
    return q + (o >> 2) * Q8 + (o & 3) * 64;
}
template <int L1>
__global__ void __launch_bounds__(NTT8X_THREADS) __attribute__((amdgpu_waves_per_eu(VP_NTT8X_WAVES, VP_NTT8X_WAVES))) k_ntt8_colsx(Ntt8xArgs a) {
    __shared__ __align__(16) F L[NTT8X_TILE];
    constexpr u32 N1 = 1u << L1, LC = 11 - L1, COLS = 1u << LC, Q8 = N1 >> 3, N2 = 512;       // N = N1 x 512 (the caller's split)
    const u32 NM = (1u << a.ln) - 1;
    const u32 tid = threadIdx.x, c = tid & (COLS - 1), q = tid >> LC;                    // q < N1 / 8
    const u32 row = blockIdx.x, tile = blockIdx.y, j2 = tile * COLS + c, b0 = blockIdx.z * a.cpw;
    const F *src = a.in + (size_t) row * a.in_stride + j2;
    F uin[8], w1[8];
#pragma unroll
    for (u32 m = 0; m < 8; ++m) uin[m] = src[(size_t) (q + m * Q8) * N2];
#pragma unroll
    for (u32 m = 0; m < 8; ++m) w1[m] = a.W1[(j2 * ntt8x_k1<L1>(q, m)) & NM];              // w_N^(j2 k1): the same for every coset
    const u32 TWM = a.ncoset * N1 - 1, ntiles = N2 >> LC, k = q & 7;
    F *dst = a.out + (((size_t) row * a.ncoset + b0) << a.ln) + j2;
    for (u32 bi = 0; bi < a.cpw; ++bi, dst += (size_t) 1 << a.ln) {
        const u32 b = b0 + bi;
        F u[8];
        // ---- first pass (radix 8, s = 1): the inputs from registers, twisted by w^(j1 b), j1 = q + m N1 / 8
        {
            const u32 e0 = q * b, step = Q8 * b;
#pragma unroll
            for (u32 g = 0; g < 8; g += NTT8X_G) {
                F w[NTT8X_G];
#pragma unroll
                for (u32 m = 0; m < NTT8X_G; ++m) w[m] = a.TW[(e0 + (g + m) * step) & TWM];
                loads_first();
#pragma unroll
                for (u32 m = 0; m < NTT8X_G; ++m) u[g + m] = lz_mul_t(w[m], uin[g + m]);
                loads_first();
            }
            lz_dft8<false>(u);
#pragma unroll
            for (u32 m = 0; m < 8; ++m) if (VP_CHK((((q * 8 + m) << LC) + c) < NTT8X_TILE, 6, q, m, c)) L[((q * 8 + m) << LC) + c] = u[m];
        }
        __syncthreads();
        // ---- second pass (radix 8, s = 8): twiddles x the column's share of the twist, one table
        {
            const F *t2 = a.T2 + ((((size_t) b * ntiles + tile) * 64 + k) << LC) + c;              // [m] at stride 8 COLS
#pragma unroll
            for (u32 m = 0; m < 8; ++m) u[m] = L[((q + m * Q8) << LC) + c];
#pragma unroll
            for (u32 g = 0; g < 8; g += NTT8X_G) {
                F w[NTT8X_G];
#pragma unroll
                for (u32 m = 0; m < NTT8X_G; ++m) w[m] = t2[(size_t) (g + m) * 8 * COLS];
                loads_first();
#pragma unroll
                for (u32 m = 0; m < NTT8X_G; ++m) u[g + m] = lz_mul_t(w[m], u[g + m]);
                loads_first();
            }
            lz_dft8<false>(u);
        }
        if (L1 == 6) {
#pragma unroll
            for (u32 m = 0; m < 8; ++m) { dst[(size_t) ntt8x_k1<L1>(q, m) * N2] = lz_mul_t(w1[m], u[m]); if ((m & (NTT8X_G - 1)) == NTT8X_G - 1) loads_first(); }
            __syncthreads();                                    // the tile is free for the next coset's first pass
            continue;
        }
        __syncthreads();
#pragma unroll
        for (u32 m = 0; m < 8; ++m) L[(((q - k) * 8 + k + m * 8) << LC) + c] = u[m];
        __syncthreads();
        // ---- third pass at s = 64: radix 2 (N1 = 128, four butterflies per thread) or radix 4 (N1 = 256, two)
        if (L1 == 7) {
#pragma unroll
            for (u32 h = 0; h < 4; ++h) {
                const u32 qq = q + h * Q8;                                                // < 64 = s: k = qq
                F v[2];
                v[0] = L[(qq << LC) + c]; v[1] = L[((qq + 64) << LC) + c];
                v[1] = lz_mul_t(a.RTn[qq], v[1]);
                lz_dft2(v);
                dst[(size_t) ntt8x_k1<L1>(q, 2 * h) * N2] = lz_mul_t(w1[2 * h], v[0]);
                dst[(size_t) ntt8x_k1<L1>(q, 2 * h + 1) * N2] = lz_mul_t(w1[2 * h + 1], v[1]);
                loads_first();
            }
        } else {
#pragma unroll
            for (u32 h = 0; h < 2; ++h) {
                const u32 qq = q + h * Q8;
                F v[4], w[4];
#pragma unroll
                for (u32 m = 0; m < 4; ++m) v[m] = L[((qq + 64 * m) << LC) + c];
#pragma unroll
                for (u32 m = 1; m < 4; ++m) w[m] = a.RTn[(qq * m) & (N1 - 1)];
                loads_first();
#pragma unroll
                for (u32 m = 1; m < 4; ++m) v[m] = lz_mul_t(w[m], v[m]);
                lz_dft4<false>(v);
#pragma unroll
                for (u32 m = 0; m < 4; ++m) { dst[(size_t) ntt8x_k1<L1>(q, 4 * h + m) * N2] = lz_mul_t(w1[4 * h + m], v[m]); if (m & 1) loads_first(); }
            }
        }
        __syncthreads();                                        // every read of the tile is done before the next coset's first pass writes it
    }
}
// T2[b][tile][m][k][c] = w_M^(j2 b + (k m mod 64) M / 64), j2 = tile 2^lc + c: the second-pass roots of k_ntt8_colsx, from the half table of order M
__global__ void __launch_bounds__(VP_BLOCK) k_ntt8x_t2(const F *__restrict__ RT, u32 half_m, u32 lc, u32 n2, u32 total, F *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u32 c = i & ((1u << lc) - 1), k = (i >> lc) & 7, m = (i >> (lc + 3)) & 7, bt = i >> (lc + 6), ntiles = n2 >> lc;
    const u32 tile = bt % ntiles, b = bt / ntiles, j2 = (tile << lc) + c, M = 2 * half_m;
    out[i] = root_pow(RT, half_m, (j2 * b + ((k * m) & 63) * (M >> 6)) & (M - 1));
}

// ---- pass B: 512-point transforms along 8 neighbouring rows k1, one wave per row; natural-order store -----------------------------------------
constexpr u32 NTT8_PITCH = 578;     // elements per row in LDS: 512 * 9 / 8 padded positions, + 2 so that the transposed read-out spreads over the banks
__device__ __forceinline__ u32 ntt8_pad(u32 i) { return i + (i >> 3); }
template <bool INV>
__global__ void __launch_bounds__(NTT8_THREADS, 2) k_ntt8_rows(Ntt8Args a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    F *L = reinterpret_cast<F *>(smem_raw);                       // 8 x NTT8_PITCH elements
    const u32 N1 = 1u << a.l1, tid = threadIdx.x, w = tid >> 6, j = tid & 63;
    const u32 k1_0 = blockIdx.x * 8, rt = blockIdx.y;             // rt = row * ncoset + coset
    const F *src = a.in + ((size_t) rt << a.ln) + (size_t) (k1_0 + w) * 512;
    F *Lw = L + w * NTT8_PITCH;
    F u[8];
    // pass 1 (s = 1): straight from global memory, no twiddles; outputs at 8 j + m
#pragma unroll
    for (u32 m = 0; m < 8; ++m) u[m] = src[j + 64 * m];
    lz_dft8<INV>(u);
#pragma unroll
    for (u32 m = 0; m < 8; ++m) Lw[9 * j + m] = u[m];
    __syncthreads();
    // pass 2 (s = 8): twiddles w_64^(k m) = w_512^(8 k m); outputs at (j - k) 8 + k + 8 m
    {
        const u32 k = j & 7;
#pragma unroll
        for (u32 m = 0; m < 8; ++m) u[m] = Lw[ntt8_pad(j) + 72 * m];
        F tw[8];
#pragma unroll
        for (u32 m = 1; m < 8; ++m) { const u32 e = 8 * k * m; tw[m] = a.RTn[INV ? ((512 - e) & 511) : e]; }
#pragma unroll
        for (u32 m = 1; m < 8; ++m) u[m] = lz_mul_t(tw[m], u[m]);
        lz_dft8<INV>(u);
        __syncthreads();
#pragma unroll
        for (u32 m = 0; m < 8; ++m) Lw[9 * (j - k) + k + 9 * m] = u[m];
        __syncthreads();
    }
    // pass 3 (s = 64): twiddles w_512^(j m); outputs at j + 64 m
    {
#pragma unroll
        for (u32 m = 0; m < 8; ++m) u[m] = Lw[ntt8_pad(j) + 72 * m];
        F tw[8];
#pragma unroll
        for (u32 m = 1; m < 8; ++m) { const u32 e = (j * m) & 511; tw[m] = a.RTn[INV ? ((512 - e) & 511) : e]; }
#pragma unroll
        for (u32 m = 1; m < 8; ++m) u[m] = lz_mul_t(tw[m], u[m]);
        lz_dft8<INV>(u);
        __syncthreads();
#pragma unroll
        for (u32 m = 0; m < 8; ++m) Lw[ntt8_pad(j) + 72 * m] = a.do_scale ? lz_mul_t(a.scale, u[m]) : u[m];
        __syncthreads();
    }
    // natural order: element k2 of row k1 goes to k1 + N1 k2 — the tile's eight k1 are consecutive: 128 contiguous bytes per k2
    if (a.pair_rows) {                                            // uniform: the two real slices of this row, see Ntt8Args
        const u32 pr = rt / a.ncoset, coset = rt - pr * a.ncoset, M = 2 * a.half_m;
        const F E0 = a.e0[(size_t) pr << a.ln];
        const F zeta = a.RT[(coset * (M / a.ncoset)) & (M - 1)];  // w_ncoset^coset = x^N on the coset
        F *d1 = a.out + (((size_t) pr * a.ncoset + coset) << a.ln) + k1_0;
        F *d2 = a.out + (((size_t) (pr + a.pair_rows) * a.ncoset + coset) << a.ln) + k1_0;
#pragma unroll
        for (u32 it = 0; it < 8; ++it) {
            const u32 e = it * NTT8_THREADS + tid, cc = e & 7, k2 = e >> 3;
            const F Y = f_sub(lz_canon(L[cc * NTT8_PITCH + ntt8_pad(k2)]), E0);
            const F Z = f_mul_plain(zeta, f_make(Y.re, Y.im ? P61 - Y.im : 0));
            const F s = f_half(f_add(Y, Z)), d = f_half(f_sub(Y, Z));
            d1[(size_t) k2 * N1 + cc] = f_make(m_add(s.re, E0.re), s.im);
            d2[(size_t) k2 * N1 + cc] = f_make(m_add(d.im, E0.im), d.re ? P61 - d.re : 0);       // -i (x + i y) = y - i x
        }
        return;
    }
    F *dst = a.out + ((size_t) rt << a.ln) + k1_0;
#pragma unroll
    for (u32 it = 0; it < 8; ++it) {
        const u32 e = it * NTT8_THREADS + tid, cc = e & 7, k2 = e >> 3;
        dst[(size_t) k2 * N1 + cc] = lz_canon(L[cc * NTT8_PITCH + ntt8_pad(k2)]);
    }
}

// out[k] = w^k for k < n = 2^lo, w of order n, from the half table of order M (full circle: the inverse transforms index it with n - e)
__global__ void __launch_bounds__(VP_BLOCK) k_root_circle(const F *__restrict__ RT, u32 half_m, u32 stride, u32 n, F *__restrict__ out) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = root_pow(RT, half_m, (u32) (((unsigned long long) k * stride) & (2ull * half_m - 1)));
}

}  // namespace vp
