// Device/host arithmetic in F = F_p[i]/(i^2+1), p = 2^61-1.
//
// Value semantics follow lib/virgo/src/fieldElement.cpp:34-104 of the reference (canonical limbs in
// [0,p), i^2 = -1), but the computation is laid out for CDNA4: there is no 64x64 multiplier on gfx950,
// a 64x64->128 product is four v_mad_u64_u32; the three Karatsuba products of an F-multiply are kept
// as unreduced 128-bit values and only the two output limbs are folded (2^61 == 1 mod p), instead of
// the reference's reduce-after-every-step sequence.  Results are bit-identical because every output
// is the canonical representative.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VP_HD __host__ __device__ __forceinline__
#else
#define VP_HD inline
#endif

namespace vp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

constexpr u64 P61 = 2305843009213693951ull;

struct alignas(16) F {
    u64 re, im;
};

VP_HD F f_make(u64 re, u64 im) { F r; r.re = re; r.im = im; return r; }
VP_HD F f_zero() { return f_make(0, 0); }
VP_HD F f_one() { return f_make(1, 0); }
VP_HD bool f_eq(const F &a, const F &b) { return a.re == b.re && a.im == b.im; }
VP_HD bool f_is_zero(const F &a) { return (a.re | a.im) == 0; }

VP_HD u64 m_add(u64 a, u64 b) { u64 s = a + b; return s >= P61 ? s - P61 : s; }
VP_HD u64 m_sub(u64 a, u64 b) { return a >= b ? a - b : a + P61 - b; }
// x < 2^125 -> canonical
VP_HD u64 m_red128(u128 x) {
    u64 lo = (u64) x & P61;
    u64 hi = (u64) (x >> 61);
    u64 s = lo + (hi & P61) + (hi >> 61);
    s = (s & P61) + (s >> 61);
    return s >= P61 ? s - P61 : s;
}
VP_HD u64 m_mul(u64 a, u64 b) { return m_red128((u128) a * b); }

VP_HD F f_add(const F &a, const F &b) { return f_make(m_add(a.re, b.re), m_add(a.im, b.im)); }
VP_HD F f_sub(const F &a, const F &b) { return f_make(m_sub(a.re, b.re), m_sub(a.im, b.im)); }
VP_HD F f_neg(const F &a) { return f_make(a.re ? P61 - a.re : 0, a.im ? P61 - a.im : 0); }
VP_HD F f_dbl(const F &a) { return f_add(a, a); }
// x / 2 for canonical limbs: (x + p) / 2 when x is odd — the same field element as x * 2^-1, without the multiplier
VP_HD u64 m_half(u64 x) { return (x + ((x & 1) ? P61 : 0ull)) >> 1; }
VP_HD F f_half(const F &x) { return f_make(m_half(x.re), m_half(x.im)); }

VP_HD F f_mul128(const F &a, const F &b) {      // Karatsuba on 128-bit products (host code, and the reference point of the tests)
    const u128 C = ((u128) P61) << 61;           // multiple of p, >= any product of two canonical limbs
    u128 ac = (u128) a.re * b.re;
    u128 bd = (u128) a.im * b.im;
    u128 cr = (u128) (a.re + a.im) * (b.re + b.im);
    return f_make(m_red128(ac + C - bd), m_red128(cr + C + C - ac - bd));
}
template <bool WEAK, bool MS> VP_HD F f_mad31c(const F &a, const F &b, const F &c);
// a*b for canonical a, b.  On the device the 31-bit split form (f_mad31c below) is ~12 % cheaper than the 128-bit Karatsuba
// form (fewer shift/mask/select instructions around the same sixteen v_mad_u64_u32); both give the canonical product.
#ifndef VP_MADSHIFT
#define VP_MADSHIFT 1         // device multiplies take C * 2^31 through the multiplier (c31_add below); f_mul_plain is the form without it
#endif
VP_HD F f_mul(const F &a, const F &b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VP_MUL128)
    return f_mad31c<false, VP_MADSHIFT != 0>(a, b, f_make(0, 0));
#else
    return f_mul128(a, b);
#endif
}
VP_HD F f_mul_plain(const F &a, const F &b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VP_MUL128)
    return f_mad31c<false, false>(a, b, f_make(0, 0));
#else
    return f_mul128(a, b);
#endif
}
// 31-bit split multiply-add (used by the throughput kernels).  With x = hi*2^31 + lo every partial sum of
//   a*b + c*d = H*2^62 + C*2^31 + L   (H, C, L sums of 32x32 products)
// fits one 64-bit register (no carries, no 128-bit values), v_mad_u64_u32 accumulates them in place, and
// 2^61 = 1 (mod p) folds the three words with shifts: ~85 instructions per F-multiply instead of ~110-200 for
// the 128-bit form, same 16 multiplier instructions.  Operands may be lazy differences in [0, 2p].
struct Sp31 { u32 lo, hi; };                     // x = hi * 2^31 + lo,  x < 2^62
VP_HD Sp31 split31(u64 x) { Sp31 s; s.lo = (u32) x & 0x7fffffffu; s.hi = (u32) (x >> 31); return s; }
// C * 2^31 + base (mod p) for the middle word C < 2^64 of a split product:  (C mod 2^30) * 2^31 + (C >> 30) + base.
// MS (device only): the shift by 31 rides on the multiplier — v_mad_u64_u32(C mod 2^30, 2^31, base) instead of v_lshlrev_b64 + two v_and +
// v_lshl_add_u64; the constant is kept opaque in an SGPR or the compiler turns the product back into the shift.  A template switch, because it
// pays in the GKR kernels (x64 proof -2.5 %, x1024 -3 %: the gain sits in the plain products of the init / light / closing kernels, the fold
// pair step alone is indifferent) and costs in the transforms of the commitment (commit side +25 %, same call): vp_kernels_pc.h multiplies
// with f_mul_plain.
template <bool MS>
VP_HD u64 c31_add(u64 C, u64 base) {             // base + c2 < 2^64 is the caller's business
#if defined(__HIP_DEVICE_COMPILE__)
    if (MS) {
        u32 k31;
        asm("s_mov_b32 %0, 0x80000000" : "=s"(k31));
        return (u64) ((u32) C & 0x3fffffffu) * k31 + base + (C >> 30);
    }
#endif
    return ((C & 0x3fffffffull) << 31) + (C >> 30) + base;
}
// x*y + z*w + addend (mod p);  x, y, z, w < 2^62, addend < 2^61 + 8.  Canonical result (WEAK: folded once only, < 2^61 + 4,
// for values that go straight into an unreduced sum).
template <bool WEAK = false, bool MS = false>
VP_HD u64 dot2_31(const Sp31 &x, const Sp31 &y, const Sp31 &z, const Sp31 &w, u64 addend) {
    const u64 L = (u64) x.lo * y.lo + (u64) z.lo * w.lo;                                            // < 2^63
    const u64 C = (u64) x.lo * y.hi + (u64) x.hi * y.lo + (u64) z.lo * w.hi + (u64) z.hi * w.lo;    // < 2^64
    const u64 H = (u64) x.hi * y.hi + (u64) z.hi * w.hi;                                            // < 2^63
    const u64 h2 = ((H & ((1ull << 60) - 1)) << 1) + (H >> 60);          // H * 2^62
    const u64 l2 = (L & P61) + (L >> 61);
    u64 s = c31_add<MS>(C, l2) + h2 + addend;                                // C * 2^31 + ...  < 2^63
    s = (s & P61) + (s >> 61);
    if (WEAK) return s;
    return s >= P61 ? s - P61 : s;
}
// a*b + c;  limbs of a, b in [0, 2p], limbs of c in [0, p].  Canonical result unless WEAK.
template <bool WEAK = false, bool MS = false>
VP_HD F f_mad31(const F &a, const F &b, const F &c) {
    const Sp31 ar = split31(a.re), ai = split31(a.im), br = split31(b.re), bi = split31(b.im);
    const Sp31 nbi = split31(2 * P61 - b.im);                // -b.im (mod p), in [0, 2p]
    return f_make(dot2_31<WEAK, MS>(ar, br, ai, nbi, c.re), dot2_31<WEAK, MS>(ar, bi, ai, br, c.im));
}
// Same with x and z CANONICAL (< 2^61, so hi < 2^30): H*2^62 = 2H (mod p) is obtained by doubling x.hi / z.hi, and then
// L + 2H < 2^64 shares ONE accumulator (four multiply-adds in a row, no separate shift-fold of H).  y, w < 2^62 as before.
// WEAK: the result is only folded once (< 2^61 + 4, congruent mod p, not canonical) — for values that go straight into an
// unreduced sum.
template <bool WEAK, bool MS = false>
VP_HD u64 dot2_31c(const Sp31 &x, const Sp31 &y, const Sp31 &z, const Sp31 &w, u64 addend) {
    const u64 LH = (u64) x.lo * y.lo + (u64) z.lo * w.lo + (u64) (2 * x.hi) * y.hi + (u64) (2 * z.hi) * w.hi;    // < 2^64
    const u64 C = (u64) x.lo * y.hi + (u64) x.hi * y.lo + (u64) z.lo * w.hi + (u64) z.hi * w.lo;              // < 2^64
    u64 s = c31_add<MS>(C, LH & P61) + (LH >> 61) + addend;              // C * 2^31 + ...  < 2^63
    s = (s & P61) + (s >> 61);
    if (WEAK) return s;
    return s >= P61 ? s - P61 : s;
}
// a*b + c with the limbs of a canonical; limbs of b in [0, 2p], limbs of c in [0, p].
template <bool WEAK, bool MS = false>
VP_HD F f_mad31c(const F &a, const F &b, const F &c) {
    const Sp31 ar = split31(a.re), ai = split31(a.im), br = split31(b.re), bi = split31(b.im);
    const Sp31 nbi = split31(2 * P61 - b.im);
    return f_make(dot2_31c<WEAK, MS>(ar, br, ai, nbi, c.re), dot2_31c<WEAK, MS>(ar, bi, ai, br, c.im));
}
// x0 + r*d for a challenge r that the whole launch shares (the fold of a bookkeeping table): the two partial products that carry -r.im take
// the negated SCALAR p - r.im (wave-uniform, computed once on the scalar unit) instead of a per-element negation of d.im — a 64-bit subtract
// and one split less per fold than f_mad31c(r, d, x0), and d.im may exceed 2p.  r canonical; limbs of d < 2^62 + 8, of x0 < 2^61 + 8.
template <bool WEAK, bool MS = false>
VP_HD F f_fold31(const F &r, const F &d, const F &x0) {
    const Sp31 rr = split31(r.re), ri = split31(r.im), nri = split31(P61 - r.im);       // p - r.im in [1, p]: hi < 2^30, as dot2_31c wants
    const Sp31 dr = split31(d.re), di = split31(d.im);
    return f_make(dot2_31c<WEAK, MS>(rr, dr, nri, di, x0.re), dot2_31c<WEAK, MS>(rr, di, ri, dr, x0.im));
}
// The same two forms for a REAL second factor (b = (y, 0)): each limb of a*b + c is ONE product, eight multiplier instructions per
// F-multiply instead of sixteen.  Circuit values of a circuit with real inputs and constants are real (every gate of
// src/prover.cpp:27-91 maps reals to reals), so the first round of every sumcheck takes this path when vp_evaluate found no
// imaginary part (vp_ctx::d_vcplx); the results are the same field elements.
template <bool WEAK = false, bool MS = false>
VP_HD u64 dot1_31(const Sp31 &x, const Sp31 &y, u64 addend) {                // x, y < 2^62
    const u64 L = (u64) x.lo * y.lo;                                       // < 2^62
    const u64 C = (u64) x.lo * y.hi + (u64) x.hi * y.lo;                   // < 2^63
    const u64 H = (u64) x.hi * y.hi;                                       // < 2^62
    const u64 h2 = ((H & ((1ull << 60) - 1)) << 1) + (H >> 60);
    u64 s = c31_add<MS>(C, L) + h2 + addend;                                   // < 2^61 + 2^61 + 2^62 + 2^61 + 8 < 2^64
    s = (s & P61) + (s >> 61);
    if (WEAK) return s;
    return s >= P61 ? s - P61 : s;
}
template <bool WEAK, bool MS = false>
VP_HD u64 dot1_31c(const Sp31 &x, const Sp31 &y, u64 addend) {               // x canonical, y < 2^62
    const u64 LH = (u64) x.lo * y.lo + (u64) (2 * x.hi) * y.hi;            // < 2^62 + 2^62
    const u64 C = (u64) x.lo * y.hi + (u64) x.hi * y.lo;                   // < 2^63
    u64 s = c31_add<MS>(C, LH & P61) + (LH >> 61) + addend;
    s = (s & P61) + (s >> 61);
    if (WEAK) return s;
    return s >= P61 ? s - P61 : s;
}
// a * (y, 0) + c: limbs of a in [0, 2p], y in [0, 2p], limbs of c in [0, p]
template <bool WEAK = false, bool MS = false>
VP_HD F f_mad31_rb(const F &a, u64 y, const F &c) {
    const Sp31 ar = split31(a.re), ai = split31(a.im), b = split31(y);
    return f_make(dot1_31<WEAK, MS>(ar, b, c.re), dot1_31<WEAK, MS>(ai, b, c.im));
}
// the same with the limbs of a canonical
template <bool WEAK, bool MS = false>
VP_HD F f_mad31c_rb(const F &a, u64 y, const F &c) {
    const Sp31 ar = split31(a.re), ai = split31(a.im), b = split31(y);
    return f_make(dot1_31c<WEAK, MS>(ar, b, c.re), dot1_31c<WEAK, MS>(ai, b, c.im));
}
// a*b + c*d for CANONICAL a, b, c, d (every limb < 2^61, so every high half of a split is < 2^30): each limb of the result is ONE sum of four split
// products — L < 2^64 (four lo*lo, each < 2^62), C < 2^64 (eight lo*hi, each < 2^61), H < 2^62 (four hi*hi, each < 2^60) — folded once.  The same thirty-two
// multiplier instructions as two F-multiplications, ONE reduction per limb instead of two and no addition between them; the negated imaginary parts are
// p - x (in [1, p]: high half still < 2^30).  The result is the canonical field element a*b + c*d.
template <bool MS = false>
VP_HD u64 dot4_31cc(const Sp31 &x0, const Sp31 &y0, const Sp31 &x1, const Sp31 &y1, const Sp31 &x2, const Sp31 &y2, const Sp31 &x3, const Sp31 &y3) {
    const u64 L = (u64) x0.lo * y0.lo + (u64) x1.lo * y1.lo + (u64) x2.lo * y2.lo + (u64) x3.lo * y3.lo;
    const u64 C = (u64) x0.lo * y0.hi + (u64) x0.hi * y0.lo + (u64) x1.lo * y1.hi + (u64) x1.hi * y1.lo
                + (u64) x2.lo * y2.hi + (u64) x2.hi * y2.lo + (u64) x3.lo * y3.hi + (u64) x3.hi * y3.lo;
    const u64 H = (u64) x0.hi * y0.hi + (u64) x1.hi * y1.hi + (u64) x2.hi * y2.hi + (u64) x3.hi * y3.hi;
    u64 s = c31_add<MS>(C, (L & P61) + (L >> 61)) + (H << 1);              // C 2^31 + L + H 2^62:  < (2^61 + 2^34 + 2^61 + 8) + 2^63 < 2^64
    s = (s & P61) + (s >> 61);                                             // < 2^61 + 8
    return s >= P61 ? s - P61 : s;
}
template <bool MS = false>
VP_HD F f_dot2cc(const F &a, const F &b, const F &c, const F &d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VP_MUL128)
    const Sp31 ar = split31(a.re), ai = split31(a.im), nai = split31(P61 - a.im), br = split31(b.re), bi = split31(b.im);
    const Sp31 cr = split31(c.re), ci = split31(c.im), nci = split31(P61 - c.im), dr = split31(d.re), di = split31(d.im);
    return f_make(dot4_31cc<MS>(ar, br, nai, bi, cr, dr, nci, di), dot4_31cc<MS>(ar, bi, ai, br, cr, di, ci, dr));
#else
    return f_add(f_mul128(a, b), f_mul128(c, d));
#endif
}
// a + r*(b - a): one fold step of a bookkeeping table (src/prover.cpp:483 eval + interpolate)
VP_HD F f_lerp(const F &a, const F &b, const F &r) { return f_add(a, f_mul(r, f_sub(b, a))); }

}  // namespace vp
