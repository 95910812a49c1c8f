// fft_gkr on the device: the circuit and the sumcheck tables of lib/virgo's fft_circuit_gkr
// (lib/virgo/src/fft_circuit_GKR.cpp:22-849).  Part of the single translation unit vpgpu.hip.
//
// The circuit:  r[lg] -> E = eq-expansion of r (2^lg values, :24-32) -> lg butterfly layers of the inverse FFT (:34-65) -> scaling by
// 1/n (:66-71) -> 64 * 2^lg products S[j] * x_i^j (:79-90) -> 64 sums (:91-100).  Every layer is a dense array of F in HBM; all of them
// are kept (the layer below a butterfly layer is the V table of that depth's two sumchecks).  The sumchecks themselves run through the
// fold / segment / closing kernels of the GKR path (run_sumcheck_seg): what is specific to fft_gkr is only how the mult / add tables
// of each sumcheck are written — closed forms over the butterfly wiring, one element per thread, no gathers beyond the partner entry.
// All kernels: one thread per element, consecutive lanes on consecutive 16-byte elements.
#pragma once
#include <hip/hip_runtime.h>
#include "vp_field.h"
#include "vp_kernels_round.h"

namespace vp {

// E[g] = prod_i (bit (lg-1-i) of g ? 1 - r[i] : r[i])   (fft_circuit_GKR.cpp:24-32: level i appends its bit at the bottom, 0 <-> r[i])
__global__ void __launch_bounds__(VP_BLOCK) k_fg_expand(const F *__restrict__ r, int lg, F *__restrict__ out) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << lg)) return;
    F acc = f_one();
    for (int i = 0; i < lg; ++i) {
        const F ri = r[i];
        acc = f_mul(acc, ((g >> (lg - 1 - i)) & 1u) ? f_sub(f_one(), ri) : ri);
    }
    out[g] = acc;
}

// out[t] = w^t for t < n, given sq[b] = w^(2^b): the powers the butterflies (x_k = w^(k 2^dep)) and the table inits use
__global__ void __launch_bounds__(VP_BLOCK) k_fg_pows(const F *__restrict__ sq, int nbits, u32 n, F *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    F acc = f_one();
    for (int b = 0; b < nbits; ++b) if ((t >> b) & 1u) acc = f_mul(acc, sq[b]);
    out[t] = acc;
}

// One butterfly layer (:44-64): pair p = (k, j), j < 2^dep:  l = pre[k << (dep+1) | j], rr = w^(k 2^dep) * pre[k << (dep+1) | 2^dep | j];
// cur[k << dep | j] = l + rr, cur[(k + half) << dep | j] = l - rr.  winv[t] = inv_rou^t, t < N/2.  `scaled` (last layer only): cur * inv_n.
__global__ void __launch_bounds__(VP_BLOCK) k_fg_butterfly(const F *__restrict__ pre, F *__restrict__ cur, const F *__restrict__ winv, int lg, int dep,
                                                          F *__restrict__ scaled, F inv_n) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x, halfN = 1u << (lg - 1);
    if (p >= halfN) return;
    const u32 J = 1u << dep, j = p & (J - 1), k = p >> dep, half = halfN >> dep;
    const F l = pre[(k << (dep + 1)) | j];
    const F rr = f_mul(winv[k << dep], pre[(k << (dep + 1)) | J | j]);
    const F a = f_add(l, rr), b = f_sub(l, rr);
    cur[(k << dep) | j] = a;
    cur[((k + half) << dep) | j] = b;
    if (scaled) { scaled[(k << dep) | j] = f_mul(a, inv_n); scaled[((k + half) << dep) | j] = f_mul(b, inv_n); }
}

// xsq[i * lg + b] = x_i^(2^b), i < 64
__global__ void k_fg_xsq(const F *__restrict__ xs, int lg, F *__restrict__ xsq) {
    const int i = threadIdx.x;
    if (i >= 64) return;
    F x = xs[i];
    for (int b = 0; b < lg; ++b) { xsq[i * lg + b] = x; x = f_mul(x, x); }
}
__device__ __forceinline__ F fg_xpow(const F *__restrict__ xsq_i, int lg, u32 j) {       // x_i^j from the squarings
    F acc = f_one();
    for (int b = 0; b < lg; ++b) if ((j >> b) & 1u) acc = f_mul(acc, xsq_i[b]);
    return acc;
}
// Powers of the 64 points by table (round 4): x_i^j = Q_i[j >> lb] * P_i[j & (2^lb - 1)], lb = min(lg, 9) — two multiplications per power instead of one per
// set bit of j.  xt: [64][2^lb] P then [64][2^(lg - lb)] Q.
__device__ __forceinline__ int fg_lb(int lg) { return lg < 9 ? lg : 9; }
__global__ void __launch_bounds__(VP_BLOCK) k_fg_xtab(const F *__restrict__ xsq, int lg, F *__restrict__ xt) {
    const int lb = fg_lb(lg), hb = lg - lb;
    const u32 nP = 64u << lb, nQ = 64u << hb, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nP) { const u32 i = t >> lb, b = t & ((1u << lb) - 1); xt[t] = fg_xpow(xsq + i * lg, lg, b); }
    else if (t < nP + nQ) { const u32 q = t - nP, i = q >> hb, a = q & ((1u << hb) - 1); xt[t] = fg_xpow(xsq + i * lg, lg, a << lb); }
}
__device__ __forceinline__ F fg_xpow_tab(const F *__restrict__ xt, int lg, u32 i, u32 j) {
    const int lb = fg_lb(lg), hb = lg - lb;
    const F p = xt[((size_t) i << lb) | (j & ((1u << lb) - 1))];
    return hb ? f_mul(xt[((size_t) 64 << lb) + ((size_t) i << hb) + (j >> lb)], p) : p;
}
// Pm[i << lg | j] = S[j] * x_i^j (:79-90), and the block's share of O[i] = sum_j Pm[i << lg | j] (:91-100): grid (ceil(N / 256), 64)
__global__ void __launch_bounds__(VP_BLOCK) k_fg_polyeval(const F *__restrict__ S, const F *__restrict__ xt, int lg, F *__restrict__ Pm, F *__restrict__ part) {
    __shared__ F lds[4];
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, N = 1u << lg;
    F v[1] = {f_zero()};
    if (j < N) { v[0] = f_mul(S[j], fg_xpow_tab(xt, lg, i, j)); Pm[((size_t) i << lg) | j] = v[0]; }
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) part[(size_t) i * gridDim.x + blockIdx.x] = v[0];
}
__global__ void __launch_bounds__(VP_BLOCK) k_fg_rowsum(const F *__restrict__ part, u32 nb, F *__restrict__ O) {      // one block per output
    __shared__ F lds[4];
    F v[1] = {f_zero()};
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) v[0] = f_add(v[0], part[(size_t) blockIdx.x * nb + b]);
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) O[blockIdx.x] = v[0];
}

// g[x] = alpha * eq(r0, x) + beta * eq(r1, x), x < 2^n, bit b of x set <-> r[b] (the beta_g half tables, :193-216).  alpha / beta are device
// pointers into the tape; NULL alpha = 1, NULL beta = 0 (engage_gkr starts with alpha = 1, beta = 0, :775-776: the r1 term is then absent).
__global__ void __launch_bounds__(VP_BLOCK) k_fg_gtab(const F *__restrict__ r0, const F *__restrict__ r1, int n, const F *__restrict__ alpha,
                                                     const F *__restrict__ beta, F *__restrict__ out) {
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1u << n)) return;
    F a = alpha ? *alpha : f_one();
    for (int b = 0; b < n; ++b) { const F rb = r0[b]; a = f_mul(a, ((x >> b) & 1u) ? rb : f_sub(f_one(), rb)); }
    if (beta) {
        F c = *beta;
        for (int b = 0; b < n; ++b) { const F rb = r1[b]; c = f_mul(c, ((x >> b) & 1u) ? rb : f_sub(f_one(), rb)); }
        a = f_add(a, c);
    }
    out[x] = a;
}

// addition layer (:231-236): mult[j] = g6[j >> lg] over the 64 * 2^lg products; the add table is identically zero (has_a = 0)
__global__ void __launch_bounds__(VP_BLOCK) k_fg_add_init(const F *__restrict__ g6, int lg, F *__restrict__ M) {
    const size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ((size_t) 64 << lg)) return;
    M[j] = g6[j >> lg];
}
// multiplication layer (:347-359) with alpha = 1, beta = 0:  mult[i] = sum_j g(j 2^lg + i) x_j^i = lo[i] * sum_j hi[j] x_j^i, g = eq(r0, .) split
// into lo = eq(r0[0..lg), .) and hi = eq(r0[lg..lg+6), .)
__global__ void __launch_bounds__(VP_BLOCK) k_fg_mul_init(const F *__restrict__ lo, const F *__restrict__ hi, const F *__restrict__ xt, int lg, F *__restrict__ M) {
    __shared__ F s_hi[64];
    if (threadIdx.x < 64) s_hi[threadIdx.x] = hi[threadIdx.x];
    __syncthreads();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << lg)) return;
    F acc = f_zero();
    for (int j = 0; j < 64; ++j) acc = f_add(acc, f_mul(s_hi[j], fg_xpow_tab(xt, lg, (u32) j, i)));
    M[i] = f_mul(lo[i], acc);
}

// inverse-FFT depth `dep`, phase 1 (:516-539): for the pair p = (k, j): u = k << (dep+1) | j, v = u | 2^dep, g1 = k << dep | j, g2 = (k + half) << dep | j;
//   mult[u] = g[g1] + g[g2],  add[u] = (g[g1] - g[g2]) * w^(k 2^dep) * pre[v],  mult[v] = add[v] = 0
__global__ void __launch_bounds__(VP_BLOCK) k_fg_ifft_p1(const F *__restrict__ g, const F *__restrict__ pre, const F *__restrict__ winv, int lg, int dep,
                                                        F *__restrict__ M, F *__restrict__ A) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x, halfN = 1u << (lg - 1);
    if (p >= halfN) return;
    const u32 J = 1u << dep, j = p & (J - 1), k = p >> dep, half = halfN >> dep;
    const u32 u = (k << (dep + 1)) | j, v = u | J;
    const F t1 = g[(k << dep) | j], t2 = g[((k + half) << dep) | j];
    M[u] = f_add(t1, t2);
    A[u] = f_mul(f_mul(f_sub(t1, t2), winv[k << dep]), pre[v]);
    M[v] = f_zero(); A[v] = f_zero();
}
// phase 2 (:574-606): mult[v] = (g[g1] - g[g2]) * eq(r_u, u) * w^(k 2^dep),  add[v] = (g[g1] + g[g2]) * eq(r_u, u) * v_u,  mult[u] = add[u] = 0.
// v_u = the claim phase 1 ended with (device pointer: the closing kernel of that sumcheck wrote it).
__global__ void __launch_bounds__(VP_BLOCK) k_fg_ifft_p2(const F *__restrict__ g, const F *__restrict__ eu, const F *__restrict__ vu, const F *__restrict__ winv,
                                                        int lg, int dep, F *__restrict__ M, F *__restrict__ A) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x, halfN = 1u << (lg - 1);
    if (p >= halfN) return;
    const u32 J = 1u << dep, j = p & (J - 1), k = p >> dep, half = halfN >> dep;
    const u32 u = (k << (dep + 1)) | j, v = u | J;
    const F t1 = g[(k << dep) | j], t2 = g[((k + half) << dep) | j], e = eu[u];
    M[v] = f_mul(f_mul(f_sub(t1, t2), e), winv[k << dep]);
    A[v] = f_mul(f_mul(f_add(t1, t2), e), *vu);
    M[u] = f_zero(); A[u] = f_zero();
}

// ---- the 2 lg sumchecks of the inverse FFT as ONE batch (round 4) ------------------------------------------------------------------------------
// Given the tape they are independent of each other: depth d's tables come from the layer values (all kept), from eq tables of tape entries, and — phase 2 —
// from v_u = the layer's extension at r_u, which is an inner product with eq(r_u, .) instead of phase 1's last fold.  The kernels above, one job per blockIdx.y:
struct FgTabJob { const F *r0, *r1, *alpha, *beta; F *out; };
__global__ void __launch_bounds__(VP_BLOCK) k_fg_gtab_multi(const FgTabJob *__restrict__ jobs, int n) {
    const FgTabJob j = jobs[blockIdx.y];
    const u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1u << n)) return;
    F a = j.alpha ? *j.alpha : f_one();
    for (int b = 0; b < n; ++b) { const F rb = j.r0[b]; a = f_mul(a, ((x >> b) & 1u) ? rb : f_sub(f_one(), rb)); }
    if (j.beta) {
        F c = *j.beta;
        for (int b = 0; b < n; ++b) { const F rb = j.r1[b]; c = f_mul(c, ((x >> b) & 1u) ? rb : f_sub(f_one(), rb)); }
        a = f_add(a, c);
    }
    j.out[x] = a;
}
struct FgDotJob { const F *x, *y; F *out; };
__global__ void __launch_bounds__(VP_BLOCK) k_fg_dot_multi(const FgDotJob *__restrict__ jobs, u32 n) {      // one workgroup per inner product (2^lg <= 2^20 terms)
    __shared__ F lds[4];
    const FgDotJob j = jobs[blockIdx.x];
    F v[1] = {f_zero()};
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) v[0] = f_add(v[0], f_mul(j.x[i], j.y[i]));
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) *j.out = v[0];
}
struct FgIfftJob { const F *g, *pre, *eu, *vu; F *M, *A; int dep, phase; };
__global__ void __launch_bounds__(VP_BLOCK) k_fg_ifft_multi(const FgIfftJob *__restrict__ jobs, const F *__restrict__ winv, int lg) {
    const FgIfftJob jb = jobs[blockIdx.y];
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x, halfN = 1u << (lg - 1);
    if (p >= halfN) return;
    const int dep = jb.dep;
    const u32 J = 1u << dep, j = p & (J - 1), k = p >> dep, half = halfN >> dep;
    const u32 u = (k << (dep + 1)) | j, v = u | J;
    const F t1 = jb.g[(k << dep) | j], t2 = jb.g[((k + half) << dep) | j];
    if (jb.phase == 1) {                                                   // k_fg_ifft_p1
        jb.M[u] = f_add(t1, t2);
        jb.A[u] = f_mul(f_mul(f_sub(t1, t2), winv[k << dep]), jb.pre[v]);
        jb.M[v] = f_zero(); jb.A[v] = f_zero();
    } else {                                                               // k_fg_ifft_p2
        const F e = jb.eu[u];
        jb.M[v] = f_mul(f_mul(f_sub(t1, t2), e), winv[k << dep]);
        jb.A[v] = f_mul(f_mul(f_add(t1, t2), e), *jb.vu);
        jb.M[u] = f_zero(); jb.A[u] = f_zero();
    }
}

}  // namespace vp
