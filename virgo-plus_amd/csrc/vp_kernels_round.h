// Reductions, eq tables, circuit evaluation and the per-round (interactive) sumcheck kernels.
// Part of the single translation unit vpgpu.hip (see vp_kernels.h for the overall layout rules).
#pragma once
#include <hip/hip_runtime.h>
#include "vp_field.h"

namespace vp {

#define VP_MAX_TAB 64          // max bookkeeping-table families per sumcheck (= max circuit depth)
#define VP_BLOCK 256
#define VP_LIGHT_MAX 16        // rows with more contributions than this go through the chunked path
#define VP_CHUNK 512           // contributions per wave in the chunked path (8 per lane: short chains, many waves)

enum { T_MUL = 0, T_ADD, T_SUB, T_ANTISUB, T_NAAB, T_ANTINAAB, T_INPUT, T_MULC, T_ADDC, T_XOR, T_NOT, T_COPY };

__device__ __forceinline__ F ldF(const F *p) { return *p; }

// ---------------------------------------------------------------------------------------------------
// wave / block reductions of field elements (exact: field addition is associative on canonical values)
// ---------------------------------------------------------------------------------------------------
// DPP data movement (no LDS round trip): row_shr:n within rows of 16 lanes, then row_bcast:15 / :31.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const u32 lo = __builtin_amdgcn_update_dpp(0u, (u32) v, CTRL, ROW_MASK, 0xf, false);
    const u32 hi = __builtin_amdgcn_update_dpp(0u, (u32) (v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((u64) hi << 32) | lo;
}
__device__ __forceinline__ u64 m_fold(u64 s) {          // any u64 -> [0, p)
    s = (s & P61) + (s >> 61);
    return s >= P61 ? s - P61 : s;
}
// Wave-wide sum of canonical field elements; the total lands in LANE 63.  Up to 8 canonical limbs fit a u64
// unreduced (8 * (2^61 - 1) < 2^64), so three butterfly steps are plain 64-bit adds followed by one fold.
__device__ __forceinline__ F wave_sum63(F x) {
    u64 a = x.re, b = x.im;
    a += dpp_u64<0x111, 0xf>(a); b += dpp_u64<0x111, 0xf>(b);      // row_shr:1
    a += dpp_u64<0x112, 0xf>(a); b += dpp_u64<0x112, 0xf>(b);      // row_shr:2
    a += dpp_u64<0x114, 0xf>(a); b += dpp_u64<0x114, 0xf>(b);      // row_shr:4
    a = m_fold(a); b = m_fold(b);
    a += dpp_u64<0x118, 0xf>(a); b += dpp_u64<0x118, 0xf>(b);      // row_shr:8   -> lane 15 of each row = row total
    a += dpp_u64<0x142, 0xa>(a); b += dpp_u64<0x142, 0xa>(b);      // row_bcast:15 into rows 1 and 3
    a += dpp_u64<0x143, 0xc>(a); b += dpp_u64<0x143, 0xc>(b);      // row_bcast:31 into rows 2 and 3
    return f_make(m_fold(a), m_fold(b));
}
// Same, result broadcast to every lane (readlane 63).
__device__ __forceinline__ F wave_sum(F x) {
    const F t = wave_sum63(x);
    F r;
    r.re = ((u64) __builtin_amdgcn_readlane((u32) (t.re >> 32), 63) << 32) | (u32) __builtin_amdgcn_readlane((u32) t.re, 63);
    r.im = ((u64) __builtin_amdgcn_readlane((u32) (t.im >> 32), 63) << 32) | (u32) __builtin_amdgcn_readlane((u32) t.im, 63);
    return r;   // every lane holds the sum
}

// Sum N field elements per thread over a 256-thread block; result valid in thread 0.
template <int N>
__device__ __forceinline__ void block_sum(F (&x)[N], F *lds /* >= N*4 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = wave_sum(x[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[w * N + i] = x[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            F s = lds[i];
            for (int k = 1; k < nw; ++k) s = f_add(s, lds[k * N + i]);
            x[i] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// a10: circuit evaluation, one layer per launch (src/prover.cpp:27-91)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VP_BLOCK)
k_evaluate_layer(int layer, u32 size, const uint8_t *__restrict__ ty, const int16_t *__restrict__ gl,
                 const u32 *__restrict__ gu, const u32 *__restrict__ gv, const F *__restrict__ gc,
                 F *const *__restrict__ vals, u32 *__restrict__ vcplx) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= size) return;
    const F *pre = vals[layer - 1];
    int t = ty[g];
    F x = pre[gu[g]];
    F y = f_zero();
    int l = gl[g];
    if (l >= 0) y = vals[l][gv[g]];
    F out;
    switch (t) {
        case T_ADD: out = f_add(x, y); break;
        case T_SUB: out = f_sub(x, y); break;
        case T_ANTISUB: out = f_sub(y, x); break;
        case T_MUL: out = f_mul(x, y); break;
        case T_NAAB: out = f_sub(y, f_mul(x, y)); break;
        case T_ANTINAAB: out = f_sub(x, f_mul(x, y)); break;
        case T_ADDC: out = f_add(x, gc[g]); break;
        case T_MULC: out = f_mul(x, gc[g]); break;
        case T_COPY: out = x; break;
        case T_NOT: out = f_sub(f_one(), x); break;
        case T_XOR: { F xy = f_mul(x, y); out = f_sub(f_add(x, y), f_dbl(xy)); break; }
        default: out = f_zero(); break;
    }
    vals[layer][g] = out;
    if (out.im) atomicOr(vcplx, 1u);                       // never taken for a circuit with real inputs and constants (vp_field.h, f_mad31c_rb)
}
// the same mark for the input layer
__global__ void __launch_bounds__(VP_BLOCK) k_mark_complex(const F *__restrict__ v, u32 n, u32 *__restrict__ vcplx) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v[i].im) atomicOr(vcplx, 1u);
}

// real parts of a layer's values as a dense 8-byte array (the inner products V_u / V_res of an all-real circuit read half the bytes)
__global__ void __launch_bounds__(VP_BLOCK) k_real_parts(const F *__restrict__ v, u32 n, unsigned long long *__restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[i].re;
}

__global__ void k_check_asserts(const u32 *__restrict__ idx, u32 n, const F *__restrict__ val, int *flag) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !f_is_zero(val[idx[i]])) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------------
// K1: eq table (src/utils.cpp:8-45).  beta[i] = init * prod_k (bit_k(i) ? r_k : 1-r_k) is kept as the
// outer product of two half tables bf (low n/2 bits, carries init) and bs (high bits); consumers that
// stream i in order multiply on the fly, consumers that gather use the expanded table.
// One block; each doubling step is one barrier.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VP_BLOCK)
k_beta_half(const F *__restrict__ r, int n, const F *__restrict__ init, F *bf, F *bs) {
    const int h1 = n >> 1, h2 = n - h1;
    if (threadIdx.x == 0) { bf[0] = *init; bs[0] = f_one(); }
    __syncthreads();
    for (int i = 0; i < h1; ++i) {
        F ri = r[i];
        for (u32 j = threadIdx.x; j < (1u << i); j += blockDim.x) {
            F t = f_mul(bf[j], ri);
            bf[j | (1u << i)] = t;
            bf[j] = f_sub(bf[j], t);
        }
        __syncthreads();
    }
    for (int i = 0; i < h2; ++i) {
        F ri = r[i + h1];
        for (u32 j = threadIdx.x; j < (1u << i); j += blockDim.x) {
            F t = f_mul(bs[j], ri);
            bs[j | (1u << i)] = t;
            bs[j] = f_sub(bs[j], t);
        }
        __syncthreads();
    }
}

// Several independent half-table builds in one launch (block b builds table b): used by the Liu init,
// which needs one eq table per later layer (src/prover.cpp:402-414).
struct BetaJob { const F *r; const F *init; F *bf; F *bs; int n; int pad; };
// Closed form of the same tables: entry j = init * prod_i (bit i of j ? r_i : 1 - r_i).  No level-by-level barriers: a
// thread owns its entries and runs <= 15 dependent multiplies (the level-synchronous build below spends ~2 us per level
// on a barrier and a lone multiply: 25 us at the head of every proof, with the rest of the chip idle).
__global__ void __launch_bounds__(VP_BLOCK) k_beta_half_direct(const BetaJob *__restrict__ jobs, u32 blocks_per_job) {
    __shared__ F sr[32], snr[32];                       // r_i and 1 - r_i
    const BetaJob jb = jobs[blockIdx.x / blocks_per_job];
    const u32 part = blockIdx.x % blocks_per_job;
    const int h1 = jb.n >> 1, h2 = jb.n - h1;
    const u32 total = (1u << h1) + (1u << h2);
    if (part * blockDim.x >= total) return;             // uniform per workgroup
    if ((int) threadIdx.x < jb.n) { const F ri = jb.r[threadIdx.x]; sr[threadIdx.x] = ri; snr[threadIdx.x] = f_sub(f_one(), ri); }
    __syncthreads();
    const u32 j = part * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const bool second = j >= (1u << h1);
    const u32 idx = second ? j - (1u << h1) : j;
    const int nb = second ? h2 : h1, base = second ? h1 : 0;
    F e = second ? f_one() : *jb.init;
    for (int i = 0; i < nb; ++i) e = f_mul(e, ((idx >> i) & 1u) ? sr[base + i] : snr[base + i]);
    (second ? jb.bs : jb.bf)[idx] = e;
}
__global__ void __launch_bounds__(VP_BLOCK) k_beta_half_multi(const BetaJob *__restrict__ jobs) {
    BetaJob jb = jobs[blockIdx.x];
    const int h1 = jb.n >> 1, h2 = jb.n - h1;
    F *bf = jb.bf, *bs = jb.bs;
    if (threadIdx.x == 0) { bf[0] = *jb.init; bs[0] = f_one(); }
    __syncthreads();
    for (int i = 0; i < h1; ++i) {
        F ri = jb.r[i];
        for (u32 j = threadIdx.x; j < (1u << i); j += blockDim.x) {
            F t = f_mul(bf[j], ri);
            bf[j | (1u << i)] = t;
            bf[j] = f_sub(bf[j], t);
        }
        __syncthreads();
    }
    for (int i = 0; i < h2; ++i) {
        F ri = jb.r[i + h1];
        for (u32 j = threadIdx.x; j < (1u << i); j += blockDim.x) {
            F t = f_mul(bs[j], ri);
            bs[j | (1u << i)] = t;
            bs[j] = f_sub(bs[j], t);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(VP_BLOCK)
k_beta_expand(const F *__restrict__ bf, const F *__restrict__ bs, int h1, u32 count, F *__restrict__ out) {
    const u32 mask = (1u << h1) - 1;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
        out[i] = f_mul(bf[i & mask], bs[i >> h1]);
}

__global__ void k_scale_entries(F *beta, const u32 *__restrict__ idx, u32 n, const F *__restrict__ s) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) beta[idx[i]] = f_mul(beta[idx[i]], *s);
}

// ---------------------------------------------------------------------------------------------------
// K2 / K3: phase-1 and phase-2 table initialisation (src/prover.cpp:214-273, 301-361).
// The reference scatters per gate into mult[u]/add[u] (phase 1) or mult[l][lv]/add[l][lv] (phase 2).
// Here the gates of a layer are pre-sorted by target at upload time (CSR), so each target row is a
// gather + local sum with one coalesced store: no atomics, identical results in any order.
// Rows with more than VP_LIGHT_MAX contributions (e.g. the single slot all unary gates of a layer feed
// in phase 2, src/prover.cpp:342-353) are cut into VP_CHUNK pieces summed by one wave each.
// ---------------------------------------------------------------------------------------------------
struct InitArgs {
    const u32 *rowptr;       // n_rows + 1
    const u32 *e_g;          // contribution -> gate index in the current layer (for beta_g[g])
    const u32 *e_x;          // phase 1: gate.v        phase 2: gate.u (for beta_u[u])
    const uint16_t *e_tl;    // (ty << 8) | (l & 0xff)   (l = 0xff for unary)
    const F *beta_g;
    const F *beta_u;         // phase 2 only
    F *const *vals;          // phase 1: circuitValue pointers
    const F *gc;             // gate constants of the current layer or nullptr
    const F *coef;           // phase 2: 12 x {cm, ca} from V_u
    F *M, *A;                // output tables
    u32 n_rows;
};

template <int PHASE>
__device__ __forceinline__ void contrib(const InitArgs &a, u32 e, F &m, F &ad) {
    const u32 g = a.e_g[e];
    const u32 x = a.e_x[e];
    const u32 tl = a.e_tl[e];
    const int ty = (tl >> 8) & 0x7f;                             // bit 15 marks assert gates (batched path)
    if (PHASE == 1) {
        const F t = a.beta_g[g];
        const int l = tl & 0xff;
        F ty_ = f_zero();
        if (l != 0xff) ty_ = f_mul(a.vals[l][x], t);          // t * V_l[v]
        switch (ty) {                                            // SURVEY.md Appendix A, phase-1 column
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
        const F t = f_mul(a.beta_g[g], a.beta_u[x]);
        F cm = a.coef[2 * ty], ca = a.coef[2 * ty + 1];          // per-type multiples of t (Appendix A, phase-2 column)
        if (ty == T_ADDC) ca = f_add(a.gc[g], ca);               // coef holds V_u      -> c + V_u
        if (ty == T_MULC) ca = f_mul(a.gc[g], ca);               // coef holds V_u      -> c * V_u
        m = f_add(m, f_mul(t, cm));
        ad = f_add(ad, f_mul(t, ca));
    }
}

template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK) k_init_light(InitArgs a) {
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.n_rows) return;
    u32 b = a.rowptr[row], e = a.rowptr[row + 1];
    if (e - b > VP_LIGHT_MAX) return;                             // written by k_init_combine
    F m = f_zero(), ad = f_zero();
    for (u32 k = b; k < e; ++k) contrib<PHASE>(a, k, m, ad);
    a.M[row] = m;
    a.A[row] = ad;
}

template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK)
k_init_chunks(InitArgs a, const u32 *__restrict__ chunk_beg, const u32 *__restrict__ chunk_end, u32 n_chunks,
              F *__restrict__ part) {
    const u32 c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;                                    // whole wave exits together
    const int lane = threadIdx.x & 63;
    F m = f_zero(), ad = f_zero();
    for (u32 k = chunk_beg[c] + lane; k < chunk_end[c]; k += 64) contrib<PHASE>(a, k, m, ad);
    m = wave_sum(m);
    ad = wave_sum(ad);
    if (lane == 0) { part[2 * c] = m; part[2 * c + 1] = ad; }
}

__device__ __forceinline__ void init_combine_body(const u32 *__restrict__ heavy_row, const u32 *__restrict__ heavy_cptr, u32 n_heavy,
                                                  const F *__restrict__ part, F *M, F *A, u32 bid) {
    const u32 h = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (h >= n_heavy) return;
    const int lane = threadIdx.x & 63;
    F m = f_zero(), ad = f_zero();
    for (u32 c = heavy_cptr[h] + lane; c < heavy_cptr[h + 1]; c += 64) {
        m = f_add(m, part[2 * c]);
        ad = f_add(ad, part[2 * c + 1]);
    }
    m = wave_sum(m);
    ad = wave_sum(ad);
    if (lane == 0) { M[heavy_row[h]] = m; A[heavy_row[h]] = ad; }
}
__global__ void __launch_bounds__(VP_BLOCK)
k_init_combine(const u32 *__restrict__ heavy_row, const u32 *__restrict__ heavy_cptr, u32 n_heavy,
               const F *__restrict__ part, F *M, F *A) { init_combine_body(heavy_row, heavy_cptr, n_heavy, part, M, A, blockIdx.x); }

// phase-2 per-type coefficients from V_u (device scalar set by finalize of phase 1)
__global__ void k_p2_coef(const F *__restrict__ Vu, F *coef) {
    if (threadIdx.x != 0) return;
    const F v = *Vu, one = f_one(), z = f_zero();
    const F nv = f_neg(v);
    for (int i = 0; i < 24; ++i) coef[i] = z;
    coef[2 * T_ADD] = one;                 coef[2 * T_ADD + 1] = v;
    coef[2 * T_SUB] = f_neg(one);          coef[2 * T_SUB + 1] = v;
    coef[2 * T_ANTISUB] = one;             coef[2 * T_ANTISUB + 1] = nv;
    coef[2 * T_MUL] = v;
    coef[2 * T_NAAB] = f_sub(one, v);
    coef[2 * T_ANTINAAB] = nv;             coef[2 * T_ANTINAAB + 1] = v;
    coef[2 * T_XOR] = f_sub(one, f_dbl(v)); coef[2 * T_XOR + 1] = v;
    coef[2 * T_COPY + 1] = v;
    coef[2 * T_NOT + 1] = f_sub(one, v);
    coef[2 * T_ADDC + 1] = v;              // + c per gate
    coef[2 * T_MULC + 1] = v;              // * c per gate
}

// phase-2 V tables: V[slot] = circuitValue[layer][dadId[layer][k]] (src/prover.cpp:301-306), flattened.
__global__ void __launch_bounds__(VP_BLOCK)
k_p2_gather_v(const u32 *__restrict__ g_slot, const uint8_t *__restrict__ g_layer, const u32 *__restrict__ g_idx,
              u32 n, F *const *__restrict__ vals, F *__restrict__ V) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = g_layer[i];
    V[g_slot[i]] = (l == 0xff) ? f_zero() : vals[l][g_idx[i]];
}

// ---------------------------------------------------------------------------------------------------
// K4: Liu init (src/prover.cpp:389-414): mult[u] = s0*eq(r_u,u) + sum_k s_k*eq(r_v[k], g) through
// dadId[k][pre][g] -> u.  dadId lists are duplicate-free, so each per-k pass is a collision-free update.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VP_BLOCK)
k_liu_first(const F *__restrict__ bf, const F *__restrict__ bs, int h1, u32 size, F *__restrict__ M) {
    const u32 mask = (1u << h1) - 1;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < size; i += gridDim.x * blockDim.x)
        M[i] = f_mul(bf[i & mask], bs[i >> h1]);
}
__global__ void __launch_bounds__(VP_BLOCK)
k_liu_scatter(const F *__restrict__ bf, const F *__restrict__ bs, int h1, const u32 *__restrict__ dad, u32 n, F *M) {
    const u32 mask = (1u << h1) - 1;
    for (u32 g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        const u32 u = dad[g];
        M[u] = f_add(M[u], f_mul(bf[g & mask], bs[g >> h1]));
    }
}

// ---------------------------------------------------------------------------------------------------
// K5: the sumcheck round (src/prover.cpp:436-492).
// ---------------------------------------------------------------------------------------------------
struct TabDesc {
    u32 off;          // element offset of the table inside the in/out buffers
    u32 len_in;       // logical (power of two) length of the INPUT table of this launch
    u32 valid_in;     // entries of the input table that can be non-zero
    u32 pair_start;   // first global pair index of this table in this launch
};
struct RoundArgs {
    const F *inV, *inM, *inA;
    F *outV, *outM, *outA;
    const F *rp;      // previous challenge (device) or nullptr -> rv
    F rv;
    int n_tab;
    int fold;         // 0: round 1 (tables are read as they are)   1: fold by the previous challenge first
    int has_a;        // 0: the add table is identically zero (Liu phase) and is neither read nor written
    u32 total_pairs;
    TabDesc t[VP_MAX_TAB];
};

// Lazy arithmetic of the throughput kernels (values are limbs of F):
//   d = x1 + p - x0            in [0, 2p]           (x0, x1 canonical)
//   a*b + c  with a, b in [0, 2p], c in [0, p]:  f_mad31 (vp_field.h), canonical result unless <true> (then < 2^61 + 4, for unreduced sums)
__device__ __forceinline__ F f_sub_lazy(const F &a, const F &b) { return f_make(a.re + P61 - b.re, a.im + P61 - b.im); }
// f_mad_lazy: a, b lazy.  f_mad_c: a canonical (one accumulator for L + 2H).  <true>: weakly reduced result for the lazy sums.
template <bool WEAK = false> __device__ __forceinline__ F f_mad_lazy(const F &a, const F &b, const F &c) { return f_mad31<WEAK, VP_MADSHIFT>(a, b, c); }
template <bool WEAK = false> __device__ __forceinline__ F f_mad_c(const F &a, const F &b, const F &c) { return f_mad31c<WEAK, VP_MADSHIFT>(a, b, c); }
struct Lz { u64 re, im; };                                           // unreduced sum of canonical values
__device__ __forceinline__ void lz_add(Lz &s, const F &x) { s.re += x.re; s.im += x.im; }
__device__ __forceinline__ void lz_fold(Lz &s) { s.re = (s.re & P61) + (s.re >> 61); s.im = (s.im & P61) + (s.im >> 61); }
__device__ __forceinline__ F lz_canon(const Lz &s) { return f_make(m_fold(s.re), m_fold(s.im)); }
// agent-scope word accesses: partial results that another workgroup (possibly on another XCD, behind another L2) picks up inside the same launch
__device__ __forceinline__ unsigned long long cf_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cf_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ F ld_or_zero(const F *p, u32 i, u32 valid) { return i < valid ? p[i] : f_zero(); }

// Main kernel: one thread per output pair, grid-stride.  For fold=1 a thread reads 4 consecutive
// entries per table (64 B), writes 2 (32 B) and accumulates the three sums X = sum dm*dv, Y = sum m1*v1 + a1, Z = sum m0*v0 + a0
// (the round polynomial is a = X, b = Y - X - Z, c = Z before add_term).  CLS only names the instantiation: CLS=1 is used for launches
// with at least VP_BIG_PAIRS pairs so that profilers report the bandwidth-relevant launches under their own name.
// Round 3: the arithmetic of the batched fold kernels (lazy differences, weakly reduced products into unreduced sums, the fold as one
// multiply-add) instead of f_lerp + canonical products and sums — the large rounds of the interactive path are VALU-bound too.
#define VP_BIG_PAIRS 32768
__device__ __forceinline__ F fold_entry(const F &e0, const F &e1, const F &r) { return f_mad_c(r, f_sub_lazy(e1, e0), e0); }     // e0 + r (e1 - e0), canonical
// pairs of workgroup `bid` of `nb`: folds, stores, and leaves the block sums (X, Y, Z) in thread 0's acc
__device__ __forceinline__ void round_main_body(const RoundArgs &a, u32 bid, u32 nb, F (&acc)[3], F *lds) {
    const F r = a.rp ? *a.rp : a.rv;
    Lz X{0, 0}, Y{0, 0}, Z{0, 0};
    for (u32 q = bid * blockDim.x + threadIdx.x; q < a.total_pairs; q += nb * blockDim.x) {
        int j = 0;
        while (j + 1 < a.n_tab && q >= a.t[j + 1].pair_start) ++j;
        const TabDesc td = a.t[j];
        const u32 p = q - td.pair_start;
        F v0, v1, m0, m1, a0 = f_zero(), a1 = f_zero();
        if (a.fold) {
            const u32 i0 = td.off + 4 * p, vi = td.off + td.valid_in;
            const u32 vo = (td.valid_in + 1) >> 1;            // valid length of the folded table
            F e0 = ld_or_zero(a.inV, i0, vi), e1 = ld_or_zero(a.inV, i0 + 1, vi);
            F e2 = ld_or_zero(a.inV, i0 + 2, vi), e3 = ld_or_zero(a.inV, i0 + 3, vi);
            v0 = fold_entry(e0, e1, r); v1 = fold_entry(e2, e3, r);
            e0 = ld_or_zero(a.inM, i0, vi); e1 = ld_or_zero(a.inM, i0 + 1, vi);
            e2 = ld_or_zero(a.inM, i0 + 2, vi); e3 = ld_or_zero(a.inM, i0 + 3, vi);
            m0 = fold_entry(e0, e1, r); m1 = fold_entry(e2, e3, r);
            const u32 o0 = td.off + 2 * p;
            const bool w1 = 2 * p + 1 < vo;
            a.outV[o0] = v0; a.outM[o0] = m0;
            if (w1) { a.outV[o0 + 1] = v1; a.outM[o0 + 1] = m1; }
            if (a.has_a) {
                e0 = ld_or_zero(a.inA, i0, vi); e1 = ld_or_zero(a.inA, i0 + 1, vi);
                e2 = ld_or_zero(a.inA, i0 + 2, vi); e3 = ld_or_zero(a.inA, i0 + 3, vi);
                a0 = fold_entry(e0, e1, r); a1 = fold_entry(e2, e3, r);
                a.outA[o0] = a0;
                if (w1) a.outA[o0 + 1] = a1;
            }
        } else {
            const u32 i0 = td.off + 2 * p, vi = td.off + td.valid_in;
            v0 = ld_or_zero(a.inV, i0, vi); v1 = ld_or_zero(a.inV, i0 + 1, vi);
            m0 = ld_or_zero(a.inM, i0, vi); m1 = ld_or_zero(a.inM, i0 + 1, vi);
            if (a.has_a) { a0 = ld_or_zero(a.inA, i0, vi); a1 = ld_or_zero(a.inA, i0 + 1, vi); }
        }
        // mult(x)*V(x) + add(x) with X(x) = X0 + x*(X1-X0): Karatsuba on the two evaluation points
        lz_add(X, f_mad_lazy<true>(f_sub_lazy(m1, m0), f_sub_lazy(v1, v0), f_zero()));
        lz_add(Y, f_mad_c<true>(m1, v1, a1));
        lz_add(Z, f_mad_c<true>(m0, v0, a0));
        lz_fold(X); lz_fold(Y); lz_fold(Z);
    }
    acc[0] = lz_canon(X); acc[1] = lz_canon(Y); acc[2] = lz_canon(Z);
    block_sum<3>(acc, lds);
}

// (X, Y, Z) of the whole round -> the round polynomial: retires the tables that have just reached length one into add_term
// (src/prover.cpp:445,462-467), adds add_term*(1-x) (:448) and emits the polynomial to the device transcript and, if given, to pinned host memory.
__device__ __forceinline__ void round_final_tail(const RoundArgs &a, const F (&acc)[3], F *add_term, F *scalarV, F *poly_dev, F *poly_host,
                                                 unsigned long long *seq_host, unsigned long long seq) {
    const F r = a.rp ? *a.rp : a.rv;
    F at = *add_term;
    if (!f_is_zero(at)) at = f_mul(at, f_sub(f_one(), r));
    for (int j = 0; j < a.n_tab; ++j) {
        const TabDesc td = a.t[j];
        const u32 len_out = a.fold ? (td.len_in >> 1) : td.len_in;
        if (len_out != 1) continue;
        F v, m, ad = f_zero();
        if (a.fold) {
            const u32 vi = td.off + td.valid_in;
            v = f_lerp(ld_or_zero(a.inV, td.off, vi), ld_or_zero(a.inV, td.off + 1, vi), r);
            m = f_lerp(ld_or_zero(a.inM, td.off, vi), ld_or_zero(a.inM, td.off + 1, vi), r);
            if (a.has_a) ad = f_lerp(ld_or_zero(a.inA, td.off, vi), ld_or_zero(a.inA, td.off + 1, vi), r);
        } else {                         // a table that starts with a single entry (always initialised)
            v = a.inV[td.off]; m = a.inM[td.off];
            if (a.has_a) ad = a.inA[td.off];
        }
        scalarV[j] = v;
        at = f_add(at, f_add(f_mul(v, m), ad));
    }
    *add_term = at;
    const F pa = acc[0], pb = f_sub(f_sub(f_sub(acc[1], acc[0]), acc[2]), at), pc = f_add(acc[2], at);
    poly_dev[0] = pa; poly_dev[1] = pb; poly_dev[2] = pc;
    if (poly_host) {
        poly_host[0] = pa; poly_host[1] = pb; poly_host[2] = pc;
        // the host polls seq_host instead of waiting for the stream: the polynomial must be visible before the ticket
        if (seq_host) { __threadfence_system(); __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

// What the closing launch of a round needs besides the sums
struct RoundOut { F *add_term, *scalarV, *poly_dev, *poly_host; unsigned long long *seq_host; unsigned long long seq; };

// Round 3: the workgroup that finishes LAST closes the round itself (no second launch: a round of the interactive path with more pairs than one
// workgroup takes cost two dependent launches, ~5 us of every such vp_round).  Every workgroup leaves its three sums with agent-scope stores
// (workgroups sit on different XCDs behind different L2s), makes sure they are out, and counts itself in; the one that reads n - 1 adds up all
// n triples (agent-scope loads), resets the counter for the next round and runs round_final_tail.
template <int CLS>
__global__ void __launch_bounds__(VP_BLOCK) k_round_main(RoundArgs a, F *__restrict__ part, unsigned int *__restrict__ arrivals, RoundOut o) {
    __shared__ F lds[12];
    __shared__ int s_last;
    F acc[3] = {f_zero(), f_zero(), f_zero()};
    round_main_body(a, blockIdx.x, gridDim.x, acc, lds);
    if (threadIdx.x == 0) {
        unsigned long long *w = reinterpret_cast<unsigned long long *>(part + (size_t) blockIdx.x * 3);
        cf_st(w, acc[0].re); cf_st(w + 1, acc[0].im); cf_st(w + 2, acc[1].re); cf_st(w + 3, acc[1].im); cf_st(w + 4, acc[2].re); cf_st(w + 5, acc[2].im);
        // Ordering of the hand-off, spelled out: the six stores above are agent-scope atomic stores (they bypass this XCD's L2 write-back policy),
        // s_waitcnt vmcnt(0) holds this wave until the memory system has acknowledged them, and only then is the arrival counted; the closing
        // workgroup reads them with agent-scope atomic loads behind __syncthreads() (a compiler barrier as well).  A release / acquire pair on the
        // counter would make the same guarantee through an L2 write-back + invalidate per WORKGROUP — measured on this chip at 4x the kernel's time
        // (HISTORY.md section 4, "finishing an inner product inside k_dot_multi": 22 -> 99 us) — for data that never sat in a non-coherent line.  A launch
        // that faults leaves the counter dirty: check_stream clears it on any stream error.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the sums are out before they are counted
        s_last = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;                                                       // uniform per workgroup
    F t[3] = {f_zero(), f_zero(), f_zero()};
    for (u32 i = threadIdx.x; i < gridDim.x; i += blockDim.x) {
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(part + (size_t) i * 3);
        t[0] = f_add(t[0], f_make(cf_ld(w), cf_ld(w + 1)));
        t[1] = f_add(t[1], f_make(cf_ld(w + 2), cf_ld(w + 3)));
        t[2] = f_add(t[2], f_make(cf_ld(w + 4), cf_ld(w + 5)));
    }
    __syncthreads();                                                           // lds is free again
    block_sum<3>(t, lds);
    if (threadIdx.x != 0) return;
    __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    round_final_tail(a, t, o.add_term, o.scalarV, o.poly_dev, o.poly_host, o.seq_host, o.seq);
}

// Closing kernel of a round without pairs (every live table has a single entry left): one block, nothing to sum.
__global__ void __launch_bounds__(VP_BLOCK) k_round_final(RoundArgs a, RoundOut o) {
    if (threadIdx.x != 0) return;
    const F z[3] = {f_zero(), f_zero(), f_zero()};
    round_final_tail(a, z, o.add_term, o.scalarV, o.poly_dev, o.poly_host, o.seq_host, o.seq);
}
// A round whose pairs fit one workgroup: fold + sums + closing in ONE launch (most rounds of a proof are this small; the
// per-round path pays launch latency, not bandwidth).
__global__ void __launch_bounds__(VP_BLOCK)
k_round_fused(RoundArgs a, F *add_term, F *scalarV, F *poly_dev, F *poly_host, unsigned long long *seq_host, unsigned long long seq) {
    __shared__ F lds[12];
    F acc[3] = {f_zero(), f_zero(), f_zero()};
    round_main_body(a, 0, 1, acc, lds);
    if (threadIdx.x != 0) return;
    round_final_tail(a, acc, add_term, scalarV, poly_dev, poly_host, seq_host, seq);
}

// Publishes a ticket in pinned host memory once everything submitted to the stream before it has completed (the batched
// entry points poll it instead of waiting on the stream: the runtime's blocking wait adds a wake-up latency with a long tail).
__global__ void k_ticket(unsigned long long *seq_host, unsigned long long seq) {
    __threadfence_system();
    __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The ends of a batched proof without the copy engine: the tape is read straight out of pinned host memory into its device buffer,
// the transcript is written straight into pinned host memory and the ticket published by the same kernel (a hipMemcpyAsync on either
// side is a separate command with ~8 us of its own latency, and the copy back needed a ticket launch behind it).
__global__ void __launch_bounds__(256) k_tape_in(const F *__restrict__ host_tape, F *__restrict__ dev_tape, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dev_tape[i] = host_tape[i];
}
__global__ void __launch_bounds__(1024) k_ship(const F *__restrict__ dev, F *__restrict__ host, u32 n, unsigned long long *seq_host, unsigned long long seq) {
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) host[i] = dev[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Finalize (src/prover.cpp:494-521): the claim of table j is its V table folded down to one value.
// `cur` holds the tables after the last round; tables that ran out earlier left their value in scalarV.
struct FinArgs {
    const F *curV;
    const F *rp; F rv;
    int n_tab; int rounds_done;
    u32 off[VP_MAX_TAB]; u32 valid[VP_MAX_TAB]; int bl[VP_MAX_TAB];
};
__global__ void k_finalize(FinArgs a, const F *__restrict__ scalarV, F *claims_dev, F *claims_host, F *Vu,
                           unsigned long long *seq_host, unsigned long long seq) {
    int j = threadIdx.x;
    if (j < a.n_tab) {
    const F r = a.rp ? *a.rp : a.rv;
    F c;
    if (a.bl[j] == a.rounds_done) {
        const u32 vi = a.off[j] + a.valid[j];
        if (a.rounds_done == 0) c = ld_or_zero(a.curV, a.off[j], vi);
        else c = f_lerp(ld_or_zero(a.curV, a.off[j], vi), ld_or_zero(a.curV, a.off[j] + 1, vi), r);
    } else {
        c = scalarV[j];
    }
    claims_dev[j] = c;
    if (claims_host) claims_host[j] = c;
    if (Vu && j == 0) *Vu = c;
    }
    // one wave: every lane's stores are issued before this point; the ticket goes out after they are visible to the host
    if (claims_host && seq_host) {
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------------
// K6: Vres (src/prover.cpp:99-129) = sum_g eq(r_0, g) * V_out[g].  The output layer is small (64*B
// entries), one block is enough.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VP_BLOCK)
k_vres(const F *__restrict__ bf, const F *__restrict__ bs, int h1, const F *__restrict__ val, u32 size, F *out_dev,
       F *out_host) {
    __shared__ F lds[4];
    const u32 mask = (1u << h1) - 1;
    F acc[1] = {f_zero()};
    for (u32 i = threadIdx.x; i < size; i += blockDim.x)
        acc[0] = f_add(acc[0], f_mul(f_mul(bf[i & mask], bs[i >> h1]), val[i]));
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) { *out_dev = acc[0]; if (out_host) *out_host = acc[0]; }
}

__global__ void k_zero_f(F *p, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = f_zero();
}

// parity-test helpers
__global__ void k_test_field(int op, const F *__restrict__ a, const F *__restrict__ b, F *__restrict__ o, u64 n) {
    u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 3) { const u64 j = i + 1 < n ? i + 1 : 0; o[i] = f_dot2cc(a[i], b[i], a[j], b[j]); return; }      // a_i b_i + a_(i+1) b_(i+1), cyclic
    o[i] = op == 0 ? f_add(a[i], b[i]) : op == 1 ? f_sub(a[i], b[i]) : f_mul(a[i], b[i]);
}

}  // namespace vp
