// Development probe (not part of the product): review item 2d of round 1 — "the three r-folds of a pair multiply every entry by the SAME
// challenge: a contraction of the entry's byte digits against a fixed matrix of r's digits, which maps onto the idle i8 MFMA pipe".
// Both forms of   out = x0 + r * d   (d a lazy difference in [0, 2p], x0 canonical, r canonical) are built here, checked against each
// other element by element, and timed with the arithmetic repeated in registers so that the VALU / MFMA cost is what is measured:
//   valu : f_mad31c (vp_field.h) — the product's form: 16 v_mad_u64_u32 + Mersenne folds per F-multiply-add
//   mfma : v_mfma_i32_32x32x16_i8.  D (32 x 32) = A (32 x 16) * B (16 x 32) + C
//          B  = the data: lane j (< 32) supplies the 8 bytes of d.re of element j, lane j + 32 the 8 bytes of d.im (one v_permlane32_swap
//               per 32-bit word builds both halves of a wave), bytes made signed by xor 0x80;
//          A  = constants of the round: A[i][k] = 7-bit digit v(i) of (2^(8 (k mod 8)) * R) mod p, R = r.re / p - r.im / r.im / r.re by
//               (output component, input component), rows laid out so that lane j receives the nine column sums of element j's real
//               part and lane j + 32 those of its imaginary part;
//          C  = 128 * rowsum(A): undoes the xor;
//          the nine 19-bit columns come back at 7-bit spacing and are carried into a 61-bit residue by the VALU (recombine()),
//          swapped back to one element per lane, the addend added, folded, canonicalised.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_fold_probe mfma_fold_probe.hip
#include <hip/hip_runtime.h>
#include "../virgo-plus_amd/csrc/vp_field.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace vp;
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 mulpow2(u64 x /* canonical */, int s /* < 61 */) {      // x * 2^s mod p: a rotation inside 61 bits
    if (s == 0) return x;
    const u64 y = ((x << s) & P61) | (x >> (61 - s));
    return y == P61 ? 0 : y;
}
// row i of D -> (component, digit): lanes < 32 hold rows (v % 4) + 8 (v / 4), lanes >= 32 the same + 4, in accumulator register v
__device__ __forceinline__ void row_of(int i, int &half, int &v) { const int q = i & 7; half = q >> 2; v = 4 * (i >> 3) + (q & 3); }
__device__ __forceinline__ unsigned a_entry(const F &r, int i, int k) {
    int half, v; row_of(i, half, v);
    if (v > 8) return 0;
    const u64 nim = r.im ? P61 - r.im : 0;
    const u64 R = half == 0 ? (k < 8 ? r.re : nim) : (k < 8 ? r.im : r.re);
    return (unsigned) ((mulpow2(R, 8 * (k & 7)) >> (7 * v)) & 127);
}
struct RoundConst { long a; v16i cin; };
__device__ __forceinline__ RoundConst make_const(const F &r) {
    const int l = threadIdx.x & 63, i = l & 31, k0 = 8 * (l >> 5);
    RoundConst c;
    u64 a = 0;
    for (int b = 0; b < 8; ++b) a |= (u64) a_entry(r, i, k0 + b) << (8 * b);
    c.a = (long) a;
    for (int v = 0; v < 16; ++v) {
        const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
        unsigned s = 0;
        for (int k = 0; k < 16; ++k) s += a_entry(r, row, k);
        c.cin[v] = (int) (128u * s);
    }
    return c;
}
// nine column sums (< 2^19 each, weights 2^(7v)) -> residue mod p, weakly reduced (< 2^63)
__device__ __forceinline__ u64 recombine(const v16i &d) {
    const unsigned p0 = (unsigned) d[0] + ((unsigned) d[1] << 7), p1 = (unsigned) d[2] + ((unsigned) d[3] << 7);
    const unsigned p2 = (unsigned) d[4] + ((unsigned) d[5] << 7), p3 = (unsigned) d[6] + ((unsigned) d[7] << 7);
    const u64 L = (u64) p1 * 16384u + p0, H = (u64) p3 * 16384u + p2;          // v_mad_u64_u32 each; H carries weight 2^28
    const unsigned c8 = (unsigned) d[8];
    return L + ((H & 0x1ffffffffull) << 28) + (H >> 33) + ((u64) (c8 & 31u) << 56) + (c8 >> 5);
}
__device__ __forceinline__ void swap32(u64 &x, u64 &y) {        // x[32..63] <-> y[0..31], both words
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned) x, (unsigned) y, false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned) (x >> 32), (unsigned) (y >> 32), false, false);
    x = ((u64) hi[0] << 32) | lo[0]; y = ((u64) hi[1] << 32) | lo[1];
}
// NO_MFMA (round 3): the same VALU work around the two matrix instructions, which are replaced by one cheap dependent operation each (results
// meaningless) — what the form would cost if the matrix pipe overlapped PERFECTLY with other waves' VALU work.
template <bool NO_MFMA = false>
__device__ __forceinline__ F fold_mfma(const RoundConst &c, const F &d, const F &x0) {
    u64 b1 = d.re, b2 = d.im;
    swap32(b1, b2);                                              // b1: elements 0-31 (re | im), b2: elements 32-63
    v16i d1, d2;
    if (NO_MFMA) {
        d1 = c.cin; d2 = c.cin;
        const u64 y1 = b1 ^ 0x8080808080808080ull, y2 = b2 ^ 0x8080808080808080ull;
        d1[0] += (int) y1; d1[3] ^= (int) (y1 >> 32); d1[8] += (int) y1; d2[0] += (int) y2; d2[5] ^= (int) (y2 >> 32); d2[8] += (int) y2;
    } else {
        d1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(c.a, (long) (b1 ^ 0x8080808080808080ull), c.cin, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(c.a, (long) (b2 ^ 0x8080808080808080ull), c.cin, 0, 0, 0);
    }
    u64 w1 = recombine(d1), w2 = recombine(d2);
    swap32(w1, w2);                                              // w1: real parts of all 64 elements, w2: imaginary parts
    u64 sr = w1 + x0.re, si = w2 + x0.im;                        // < 2^63 + 2^61
    sr = (sr & P61) + (sr >> 61); si = (si & P61) + (si >> 61);
    return f_make(sr >= P61 ? sr - P61 : sr, si >= P61 ? si - P61 : si);
}

template <int MODE> __global__ void __launch_bounds__(256) k_fold(const F *__restrict__ d_in, const F *__restrict__ x_in, F r, F *__restrict__ out, int reps) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    F d = d_in[i], acc = x_in[i];
    RoundConst c{};
    if (MODE >= 1) c = make_const(r);
    // MODE 3: waves 0, 2 of a workgroup (SIMDs are shared by the 8 workgroups of a CU) run the VALU form, waves 1, 3 the matrix form: does a SIMD
    // overlap one wave's matrix instructions with another wave's VALU work?  MODE 4: the same split with the matrix instructions taken out.
    const bool odd = (threadIdx.x >> 6) & 1;
    for (int k = 0; k < reps; ++k) {
        if (MODE == 0) acc = f_mad31c<false>(r, d, acc);
        else if (MODE == 1) acc = fold_mfma(c, d, acc);
        else if (MODE == 2) acc = fold_mfma<true>(c, d, acc);
        else if (MODE == 3) { if (odd) acc = fold_mfma(c, d, acc); else acc = f_mad31c<false>(r, d, acc); }
        else { if (odd) acc = fold_mfma<true>(c, d, acc); else acc = f_mad31c<false>(r, d, acc); }
        d.re = (d.re + 0x1234567ull * (k + 1)) & P61; d.im = (d.im ^ acc.re) & P61;       // the next "entry" (same sequence in both modes)
    }
    out[i] = acc;
}

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    const int blocks = cus * 8; const size_t n = (size_t) blocks * 256;
    std::vector<F> hd(n), hx(n);
    for (size_t i = 0; i < n; ++i) { hd[i] = f_make(rnd() % (2 * P61 + 1), rnd() % (2 * P61 + 1)); hx[i] = f_make(rnd() % P61, rnd() % P61); }
    hd[0] = f_make(2 * P61, 2 * P61); hd[1] = f_make(0, 0); hd[2] = f_make(P61, 1); hx[0] = f_make(P61 - 1, P61 - 1);      // extremes
    F *dd, *dx, *o0, *o1;
    CK(hipMalloc(&dd, n * 16)); CK(hipMalloc(&dx, n * 16)); CK(hipMalloc(&o0, n * 16)); CK(hipMalloc(&o1, n * 16));
    CK(hipMemcpy(dd, hd.data(), n * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), n * 16, hipMemcpyHostToDevice));
    int bad = 0;
    const F rs[4] = {f_make(rnd() % P61, rnd() % P61), f_make(P61 - 1, P61 - 1), f_make(1, 0), f_make(0, rnd() % P61)};
    for (int t = 0; t < 4; ++t) for (int reps : {1, 3}) {
        hipLaunchKernelGGL(k_fold<0>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[t], o0, reps);
        hipLaunchKernelGGL(k_fold<1>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[t], o1, reps);
        CK(hipDeviceSynchronize());
        std::vector<F> a(n), b(n);
        CK(hipMemcpy(a.data(), o0, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o1, n * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) if (a[i].re != b[i].re || a[i].im != b[i].im) { if (bad < 5) printf("mismatch r#%d reps %d at %zu: %llx %llx vs %llx %llx\n", t, reps, i, (unsigned long long) a[i].re, (unsigned long long) a[i].im, (unsigned long long) b[i].re, (unsigned long long) b[i].im); ++bad; }
    }
    if (bad) printf("MISMATCH (%d)\n", bad);
    else printf("mfma form == f_mad31c on %zu elements x 4 challenges x {1,3} chained steps\n", n);
    const int reps = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_fold<0>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[0], o0, reps);
            else if (mode == 1) hipLaunchKernelGGL(k_fold<1>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[0], o1, reps);
            else if (mode == 2) hipLaunchKernelGGL(k_fold<2>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[0], o1, reps);
            else if (mode == 3) hipLaunchKernelGGL(k_fold<3>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[0], o1, reps);
            else hipLaunchKernelGGL(k_fold<4>, dim3(blocks), dim3(256), 0, 0, dd, dx, rs[0], o1, reps);
            hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it) best = std::min(best, ms);
        }
        const double waveops = (double) n / 64 * reps;
        printf("%-28s %8.3f ms  %7.1f SIMD-cycles per wave-op (64 x  x0 + r*d, incl. ~12 cycles of the probe's own entry update)\n",
               mode == 0 ? "valu  f_mad31c" : mode == 1 ? "mfma  32x32x16 i8 + recombine" : mode == 2 ? "mfma form WITHOUT its mfma" :
               mode == 3 ? "half the waves each form" : "half valu, half no-mfma form", best, best * 1e-3 * ghz * 1e9 * cus * 4 / waveops);
    }
    return bad != 0;
}
