// HIP kernels of the GKR sumcheck hot path for gfx950 (MI355X).  Included once by vpgpu.hip.
//
// All tables are arrays of F (16 B, {real,img}); lane i of a wave touches element base+i, so every wave
// instruction is one contiguous 1 KiB global_load/store_dwordx4.  Tables carry a `valid` length: entries
// at or beyond it are zero by convention and are neither stored nor loaded (SHA-256 layer sizes are not
// powers of two, so this trims the padding the reference keeps, src/prover.cpp:472-482).
//
// Unlike the reference, which stores every bookkeeping entry as a linear polynomial {a,b} (32 B) and
// folds it lazily (src/prover.cpp:483-485), the device keeps only folded VALUES (16 B): the table of
// round k is T_k[g] = T_{k-1}[2g] + r_{k-1}*(T_{k-1}[2g+1] - T_{k-1}[2g]), and the round polynomial is
// assembled from pairs (T_k[2p], T_k[2p+1]).  Same field values, half the bytes.
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
                 F *const *__restrict__ vals) {
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

__device__ __forceinline__ F ld_or_zero(const F *p, u32 i, u32 valid) { return i < valid ? p[i] : f_zero(); }

// Main kernel: one thread per output pair, grid-stride.  For fold=1 a thread reads 4 consecutive
// entries per table (64 B), writes 2 (32 B) and accumulates the three coefficients; block partial sums
// go to part[blockIdx.x*3 + {0,1,2}].  CLS only names the instantiation: CLS=1 is used for launches with at
// least VP_BIG_PAIRS pairs so that profilers report the bandwidth-relevant launches under their own name.
#define VP_BIG_PAIRS 32768
template <int CLS>
__global__ void __launch_bounds__(VP_BLOCK) k_round_main(RoundArgs a, F *__restrict__ part) {
    __shared__ F lds[12];
    const F r = a.rp ? *a.rp : a.rv;
    F acc[3] = {f_zero(), f_zero(), f_zero()};
    for (u32 q = blockIdx.x * blockDim.x + threadIdx.x; q < a.total_pairs; q += gridDim.x * blockDim.x) {
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
            v0 = f_lerp(e0, e1, r); v1 = f_lerp(e2, e3, r);
            e0 = ld_or_zero(a.inM, i0, vi); e1 = ld_or_zero(a.inM, i0 + 1, vi);
            e2 = ld_or_zero(a.inM, i0 + 2, vi); e3 = ld_or_zero(a.inM, i0 + 3, vi);
            m0 = f_lerp(e0, e1, r); m1 = f_lerp(e2, e3, r);
            const u32 o0 = td.off + 2 * p;
            const bool w1 = 2 * p + 1 < vo;
            a.outV[o0] = v0; a.outM[o0] = m0;
            if (w1) { a.outV[o0 + 1] = v1; a.outM[o0 + 1] = m1; }
            if (a.has_a) {
                e0 = ld_or_zero(a.inA, i0, vi); e1 = ld_or_zero(a.inA, i0 + 1, vi);
                e2 = ld_or_zero(a.inA, i0 + 2, vi); e3 = ld_or_zero(a.inA, i0 + 3, vi);
                a0 = f_lerp(e0, e1, r); a1 = f_lerp(e2, e3, r);
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
        const F dm = f_sub(m1, m0), dv = f_sub(v1, v0);
        const F qa = f_mul(dm, dv), qc = f_mul(m0, v0), qe = f_mul(m1, v1);
        acc[0] = f_add(acc[0], qa);
        acc[1] = f_add(acc[1], f_add(f_sub(f_sub(qe, qa), qc), f_sub(a1, a0)));
        acc[2] = f_add(acc[2], f_add(qc, a0));
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = acc[0];
        part[blockIdx.x * 3 + 1] = acc[1];
        part[blockIdx.x * 3 + 2] = acc[2];
    }
}

// Closing kernel of a round (one block): sums the block partials, retires the tables that have just
// reached length one into add_term (src/prover.cpp:445,462-467), adds add_term*(1-x) (:448) and emits
// the round polynomial to the device transcript and, if given, to pinned host memory.
__global__ void __launch_bounds__(VP_BLOCK)
k_round_final(RoundArgs a, const F *__restrict__ part, u32 n_part, F *add_term, F *scalarV, F *poly_dev,
              F *poly_host) {
    __shared__ F lds[12];
    F acc[3] = {f_zero(), f_zero(), f_zero()};
    for (u32 i = threadIdx.x; i < n_part; i += blockDim.x) {
        acc[0] = f_add(acc[0], part[3 * i]);
        acc[1] = f_add(acc[1], part[3 * i + 1]);
        acc[2] = f_add(acc[2], part[3 * i + 2]);
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x != 0) return;
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
    const F pa = acc[0], pb = f_sub(acc[1], at), pc = f_add(acc[2], at);
    poly_dev[0] = pa; poly_dev[1] = pb; poly_dev[2] = pc;
    if (poly_host) { poly_host[0] = pa; poly_host[1] = pb; poly_host[2] = pc; }
}

// Finalize (src/prover.cpp:494-521): the claim of table j is its V table folded down to one value.
// `cur` holds the tables after the last round; tables that ran out earlier left their value in scalarV.
struct FinArgs {
    const F *curV;
    const F *rp; F rv;
    int n_tab; int rounds_done;
    u32 off[VP_MAX_TAB]; u32 valid[VP_MAX_TAB]; int bl[VP_MAX_TAB];
};
__global__ void k_finalize(FinArgs a, const F *__restrict__ scalarV, F *claims_dev, F *claims_host, F *Vu) {
    int j = threadIdx.x;
    if (j >= a.n_tab) return;
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
    o[i] = op == 0 ? f_add(a[i], b[i]) : op == 1 ? f_sub(a[i], b[i]) : f_mul(a[i], b[i]);
}

}  // namespace vp

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
};

template <int PHASE>
__device__ __forceinline__ void contrib2(const InitArgs2 &a, u32 e, F &m, F &ad) {
    const u32 g = a.e_g[e], x = a.e_x[e], tl = a.e_tl[e];
    const int ty = (tl >> 8) & 0x7f;
    F t = half_at(a.hg, g);
    if (tl & 0x8000) t = f_mul(t, *a.assert_r);
    if (PHASE == 1) {
        const int l = tl & 0xff;
        F ty_ = f_zero();
        if (l != 0xff) ty_ = f_mul(a.vals[l][x], t);
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
        const F vu = *a.Vu;
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

template <int PHASE>
__device__ __forceinline__ void init2_light_body(const InitArgs2 &a, u32 bid) {
    u32 row = bid * blockDim.x + threadIdx.x;
    if (row >= a.n_rows) return;
    if (PHASE == 2) {
        const int l = a.s_layer[row];
        if (l != 0xfe) a.V[row] = (l == 0xff) ? f_zero() : a.vals[l][a.s_idx[row]];   // 0xfe: padding slot, never read
    }
    u32 b = a.rowptr[row], e = a.rowptr[row + 1];
    if (e - b > VP_LIGHT_MAX) return;
    F m = f_zero(), ad = f_zero();
    for (u32 k = b; k < e; ++k) contrib2<PHASE>(a, k, m, ad);
    a.M[row] = m;
    a.A[row] = ad;
}
template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK) k_init2_light(InitArgs2 a) { init2_light_body<PHASE>(a, blockIdx.x); }

template <int PHASE>
__device__ __forceinline__ void init2_chunks_body(const InitArgs2 &a, const u32 *__restrict__ chunk_beg, const u32 *__restrict__ chunk_end,
                                                  u32 n_chunks, F *__restrict__ part, u32 bid) {
    const u32 c = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    F m = f_zero(), ad = f_zero();
    for (u32 k = chunk_beg[c] + lane; k < chunk_end[c]; k += 64) contrib2<PHASE>(a, k, m, ad);
    m = wave_sum(m);
    ad = wave_sum(ad);
    if (lane == 0) { part[2 * c] = m; part[2 * c + 1] = ad; }
}
template <int PHASE>
__global__ void __launch_bounds__(VP_BLOCK)
k_init2_chunks(InitArgs2 a, const u32 *__restrict__ chunk_beg, const u32 *__restrict__ chunk_end, u32 n_chunks,
               F *__restrict__ part) { init2_chunks_body<PHASE>(a, chunk_beg, chunk_end, n_chunks, part, blockIdx.x); }

// Liu init as a gather (src/prover.cpp:396-414): for every u of layer i-1 the (later layer, subset
// position) pairs that point at it were listed at upload; M[u] = s0*eq(r_u,u) + sum eq_q(g).
__device__ __forceinline__ void liu_gather_body(const u32 *__restrict__ rowptr, const uint8_t *__restrict__ e_q, const u32 *__restrict__ e_g,
                                                const Half *__restrict__ H, u32 size, F *__restrict__ M, u32 bid) {
    u32 u = bid * blockDim.x + threadIdx.x;
    if (u >= size) return;
    F m = half_at(H[0], u);
    for (u32 k = rowptr[u]; k < rowptr[u + 1]; ++k) m = f_add(m, half_at(H[e_q[k]], e_g[k]));
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
    F *outV, *outM, *outA;
    const F *r;               // r[s] = challenge of the s-th round of this launch
    F *part;                  // part[(s * part_stride) + block*3 + c]
    u32 part_stride;
    u32 total_chunks;
    int n_tab, has_a;
    u32 nblk;                 // batched launches: blocks given to this job
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
template <int N2>
__device__ __forceinline__ void sf_pairs(F (&x)[8], int s, int lane, F (&lo)[4], F (&hi)[4]) {
    const bool up = (lane >> s) & 1;
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const F A = x[2 * j], B = x[2 * j + 1];
        const F recv = shfl_xor_F(up ? A : B, 1 << s);
        lo[j] = up ? recv : A;
        hi[j] = up ? B : recv;
    }
}

template <int R, int MINW>
__global__ void __launch_bounds__(VP_BLOCK, MINW) k_sumfold(SfArgs a) {
    constexpr int G = 1 << R;
    __shared__ F lds[4 * 3 * R];
    const int lane = threadIdx.x & 63;
    const u32 wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u32 n_waves = gridDim.x * (blockDim.x >> 6);
    F acc[3 * R];
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) acc[i] = f_zero();
    F rr[R];
#pragma unroll
    for (int s = 0; s < R; ++s) rr[s] = a.r[s];
    // final lane -> element offset inside the 64 outputs of a chunk
    u32 o_fin = (u32) lane >> R;
#pragma unroll
    for (int t = 0; t < R; ++t) o_fin += ((lane >> t) & 1u) << (6 - R + t);

    for (u32 c = wave; c < a.total_chunks; c += n_waves) {
        int j = 0;
        while (j + 1 < a.n_tab && c >= a.t[j + 1].chunk_start) ++j;
        const SfTab td = a.t[j];
        const u32 cl = c - td.chunk_start;
        const u32 base = td.off + cl * 64 * G, vend = td.off + td.valid;
        F v[8], m[8], ad[8];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const u32 idx = base + 64 * q + lane;
            v[q] = ld_or_zero(a.inV, idx, vend);
            m[q] = ld_or_zero(a.inM, idx, vend);
            ad[q] = a.has_a ? ld_or_zero(a.inA, idx, vend) : f_zero();
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
            constexpr int dummy = 0; (void) dummy;
            const int n2 = G >> (s + 1);
            F vl[4], vh[4], ml[4], mh[4], al[4], ah[4];
            if (n2 == 4) { sf_pairs<4>(v, s, lane, vl, vh); sf_pairs<4>(m, s, lane, ml, mh); if (a.has_a) sf_pairs<4>(ad, s, lane, al, ah); }
            else if (n2 == 2) { sf_pairs<2>(v, s, lane, vl, vh); sf_pairs<2>(m, s, lane, ml, mh); if (a.has_a) sf_pairs<2>(ad, s, lane, al, ah); }
            else { sf_pairs<1>(v, s, lane, vl, vh); sf_pairs<1>(m, s, lane, ml, mh); if (a.has_a) sf_pairs<1>(ad, s, lane, al, ah); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= n2) break;
                const F dm = f_sub(mh[q], ml[q]), dv = f_sub(vh[q], vl[q]);
                // X += dm*dv, Y += m1*v1 + a1, Z += m0*v0 + a0: the round polynomial is (X, Y - X - Z, Z), combined once
                // per block instead of per pair
                acc[3 * s] = f_add(acc[3 * s], f_mul(dm, dv));
                F e1 = f_mul(mh[q], vh[q]), e0 = f_mul(ml[q], vl[q]);
                if (a.has_a) {
                    e1 = f_add(e1, ah[q]); e0 = f_add(e0, al[q]);
                    ad[q] = f_lerp(al[q], ah[q], rr[s]);
                }
                acc[3 * s + 1] = f_add(acc[3 * s + 1], e1);
                acc[3 * s + 2] = f_add(acc[3 * s + 2], e0);
                v[q] = f_add(vl[q], f_mul(rr[s], dv));
                m[q] = f_add(ml[q], f_mul(rr[s], dm));
            }
        }
        const u32 oi = cl * 64 + o_fin;                    // element of the folded table
        const u32 vout = (td.valid + G - 1) >> R;
        if (oi < vout) {
            a.outV[td.off + oi] = v[0];
            a.outM[td.off + oi] = m[0];
            if (a.has_a) a.outA[td.off + oi] = ad[0];
        }
    }
    // block partials
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3 * R; ++i) lds[w * 3 * R + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < R) {
        const int s = threadIdx.x;
        F X = lds[3 * s], Y = lds[3 * s + 1], Z = lds[3 * s + 2];
        for (int k = 1; k < (int) (blockDim.x >> 6); ++k) {
            X = f_add(X, lds[k * 3 * R + 3 * s]); Y = f_add(Y, lds[k * 3 * R + 3 * s + 1]); Z = f_add(Z, lds[k * 3 * R + 3 * s + 2]);
        }
        F *o = a.part + (size_t) s * a.part_stride + blockIdx.x * 3;
        o[0] = X; o[1] = f_sub(f_sub(Y, X), Z); o[2] = Z;
    }
}

// ---------------------------------------------------------------------------------------------------
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
__device__ __forceinline__ F f_sub_lazy(const F &a, const F &b) { return f_make(a.re + P61 - b.re, a.im + P61 - b.im); }
__device__ __forceinline__ F f_mad_lazy(const F &a, const F &b, const F &c) { return f_mad31(a, b, c); }
struct Lz { u64 re, im; };                                           // unreduced sum of canonical values
__device__ __forceinline__ void lz_add(Lz &s, const F &x) { s.re += x.re; s.im += x.im; }
__device__ __forceinline__ void lz_fold(Lz &s) { s.re = (s.re & P61) + (s.re >> 61); s.im = (s.im & P61) + (s.im >> 61); }
__device__ __forceinline__ F lz_canon(const Lz &s) { return f_make(m_fold(s.re), m_fold(s.im)); }

// one pair of one table family: sums into (X, Y, Z) = (sum dm*dv, sum m1*v1 + a1, sum m0*v0 + a0), folds with r
template <bool HAS_A>
__device__ __forceinline__ void sf_pair_step(const F &v0, const F &v1, const F &m0, const F &m1, const F &a0, const F &a1,
                                             const F &r, Lz &X, Lz &Y, Lz &Z, F &vo, F &mo, F &ao) {
    const F dv = f_sub_lazy(v1, v0), dm = f_sub_lazy(m1, m0);
    lz_add(X, f_mad_lazy(dm, dv, f_zero()));
    lz_add(Y, f_mad_lazy(m1, v1, HAS_A ? a1 : f_zero()));
    lz_add(Z, f_mad_lazy(m0, v0, HAS_A ? a0 : f_zero()));
    vo = f_mad_lazy(r, dv, v0);
    mo = f_mad_lazy(r, dm, m0);
    if (HAS_A) ao = f_mad_lazy(r, f_sub_lazy(a1, a0), a0);
}

struct Sf3bLds { F s1[3][256]; F s2[3][128]; F red[4][9]; Lz acc2[3][128]; Lz acc3[3][64]; };   // acc2/acc3: per-thread sums of rounds k+1, k+2
template <bool HAS_A>
__device__ __forceinline__ void sumfold3b_body(const SfArgs &a, u32 bid, u32 nb, Sf3bLds &sm) {
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
    for (u32 c = bid; c < a.total_chunks; c += nb) {
        int j = 0;
        while (j + 1 < a.n_tab && c >= a.t[j + 1].chunk_start) ++j;
        const SfTab td = a.t[j];
        const u32 cl = c - td.chunk_start;
        const u32 i0 = td.off + cl * 512 + 2 * t, vend = td.off + td.valid;
        {   // round k: one pair per thread
            const F v0 = ld_or_zero(a.inV, i0, vend), v1 = ld_or_zero(a.inV, i0 + 1, vend);
            const F m0 = ld_or_zero(a.inM, i0, vend), m1 = ld_or_zero(a.inM, i0 + 1, vend);
            F a0 = f_zero(), a1 = f_zero();
            if (HAS_A) { a0 = ld_or_zero(a.inA, i0, vend); a1 = ld_or_zero(a.inA, i0 + 1, vend); }
            F vo, mo, ao = f_zero();
            sf_pair_step<HAS_A>(v0, v1, m0, m1, a0, a1, r0, acc[0], acc[1], acc[2], vo, mo, ao);
            s1[0][t] = vo; s1[1][t] = mo;
            if (HAS_A) s1[2][t] = ao;
        }
        __syncthreads();
        if (w < 2) {   // round k+1: 128 pairs
            F vo, mo, ao = f_zero();
            Lz x = sm.acc2[0][t], y = sm.acc2[1][t], z = sm.acc2[2][t];
            sf_pair_step<HAS_A>(s1[0][2 * t], s1[0][2 * t + 1], s1[1][2 * t], s1[1][2 * t + 1],
                                HAS_A ? s1[2][2 * t] : f_zero(), HAS_A ? s1[2][2 * t + 1] : f_zero(), r1,
                                x, y, z, vo, mo, ao);
            lz_fold(x); lz_fold(y); lz_fold(z);
            sm.acc2[0][t] = x; sm.acc2[1][t] = y; sm.acc2[2][t] = z;
            s2[0][t] = vo; s2[1][t] = mo;
            if (HAS_A) s2[2][t] = ao;
        }
        __syncthreads();
        if (w == 0) {  // round k+2: 64 pairs, results are the folded table
            F vo, mo, ao = f_zero();
            Lz x = sm.acc3[0][t], y = sm.acc3[1][t], z = sm.acc3[2][t];
            sf_pair_step<HAS_A>(s2[0][2 * t], s2[0][2 * t + 1], s2[1][2 * t], s2[1][2 * t + 1],
                                HAS_A ? s2[2][2 * t] : f_zero(), HAS_A ? s2[2][2 * t + 1] : f_zero(), r2,
                                x, y, z, vo, mo, ao);
            lz_fold(x); lz_fold(y); lz_fold(z);
            sm.acc3[0][t] = x; sm.acc3[1][t] = y; sm.acc3[2][t] = z;
            const u32 oi = cl * 64 + t;
            if (oi < ((td.valid + 7) >> 3)) {
                a.outV[td.off + oi] = vo;
                a.outM[td.off + oi] = mo;
                if (HAS_A) a.outA[td.off + oi] = ao;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) lz_fold(acc[i]);
    }
    // block partials: rounds k+1 and k+2 only have contributions in waves 0-1 and 0
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
        o[0] = X; o[1] = f_sub(f_sub(Y, X), Z); o[2] = Z;
    }
}
template <bool HAS_A>
__global__ void __launch_bounds__(VP_BLOCK, 4) k_sumfold3b(SfArgs a) {
    __shared__ Sf3bLds sm;
    sumfold3b_body<HAS_A>(a, blockIdx.x, gridDim.x, sm);
}

// ---------------------------------------------------------------------------------------------------
// k_tail: one workgroup finishes a sumcheck.
// ---------------------------------------------------------------------------------------------------
#define VP_TAIL_THREADS 1024
struct TailTab {
    u32 off;          // table offset inside the ping-pong buffers
    u32 len0;         // logical length at round 1
    u32 valid0;       // valid length at round 1
    int enter;        // first round (1-based) this kernel handles for the table
    int cur;          // buffer (0/1) that holds the table at round `enter`
    int v_from_v0;    // V of round `enter` is read from V0 instead of buf[cur][0] (phase 1 / Liu, enter == 1)
};
struct TailArgs {
    const F *V0;
    F *buf[2][3];
    const F *r;                 // r[k-1] = challenge of round k
    const F *part;              // block partials written by k_sumfold: part[(k-1)*part_stride + b*3 + c]
    u32 part_stride;
    int n_tab, rounds, has_a;
    F *poly_out;                // rounds * 3
    F *claims_out;              // n_tab
    F *Vu;                      // phase 1: receives claims[0]
    uint16_t nblk[32];          // partial blocks per round
    TailTab t[VP_MAX_TAB];
};

__global__ void __launch_bounds__(VP_TAIL_THREADS) k_tail(TailArgs a) {
    __shared__ F lds[16 * 3];
    __shared__ F s_claim[VP_MAX_TAB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nth = blockDim.x;
    if (tid < a.n_tab) s_claim[tid] = f_zero();
    F at = f_zero();                                        // add_term (thread 0)
    __syncthreads();
    // tables that consist of a single entry from the start: their value is the claim (bl == 0)
    for (int k = 1; k <= (a.rounds > 0 ? a.rounds : 1); ++k) {
        const bool real_round = k <= a.rounds;
        const F rk = real_round ? a.r[k - 1] : f_zero();
        const F rprev = (k >= 2) ? a.r[k - 2] : f_zero();
        F acc[3] = {f_zero(), f_zero(), f_zero()};
        if (real_round) {
            const u32 nb = a.nblk[k - 1];
            const F *pp = a.part + (size_t) (k - 1) * a.part_stride;
            for (u32 i = tid; i < nb; i += nth) {
                acc[0] = f_add(acc[0], pp[3 * i]); acc[1] = f_add(acc[1], pp[3 * i + 1]); acc[2] = f_add(acc[2], pp[3 * i + 2]);
            }
        }
        F retire = f_zero();                                // thread 0: sum of V*M + A of tables retiring this round
        for (int j = 0; j < a.n_tab; ++j) {
            const TailTab td = a.t[j];
            if (k < td.enter) continue;
            const int sh = k - 1;
            const u32 len = sh < 32 ? (td.len0 >> sh) : 0;
            if (len == 0) continue;
            const u32 valid = (u32) (((unsigned long long) td.valid0 + (1ull << sh) - 1) >> sh);
            const int cb = td.cur ^ ((k - td.enter) & 1);       // a live table changes buffer every round
            const F *inV = (td.v_from_v0 && k == td.enter) ? a.V0 + td.off : a.buf[cb][0] + td.off;
            const F *inM = a.buf[cb][1] + td.off, *inA = a.buf[cb][2] + td.off;
            if (len == 1) {
                if (tid == 0) {
                    // always-initialised single entry (see k_round_final)
                    const F v = (td.len0 == 1) ? inV[0] : ld_or_zero(inV, 0, valid);
                    const F m = (td.len0 == 1) ? inM[0] : ld_or_zero(inM, 0, valid);
                    const F ad = a.has_a ? ((td.len0 == 1) ? inA[0] : ld_or_zero(inA, 0, valid)) : f_zero();
                    s_claim[j] = v;
                    if (real_round) retire = f_add(retire, f_add(f_mul(v, m), ad));
                }
                continue;
            }
            if (!real_round) continue;
            F *oV = a.buf[cb ^ 1][0] + td.off, *oM = a.buf[cb ^ 1][1] + td.off, *oA = a.buf[cb ^ 1][2] + td.off;
            const u32 npairs = (valid + 1) >> 1;
            for (u32 p = tid; p < npairs; p += nth) {
                const F v0 = ld_or_zero(inV, 2 * p, valid), v1 = ld_or_zero(inV, 2 * p + 1, valid);
                const F m0 = ld_or_zero(inM, 2 * p, valid), m1 = ld_or_zero(inM, 2 * p + 1, valid);
                F a0 = f_zero(), a1 = f_zero();
                if (a.has_a) { a0 = ld_or_zero(inA, 2 * p, valid); a1 = ld_or_zero(inA, 2 * p + 1, valid); }
                const F dm = f_sub(m1, m0), dv = f_sub(v1, v0);
                const F qa = f_mul(dm, dv), qc = f_mul(m0, v0), qe = f_mul(m1, v1);
                acc[0] = f_add(acc[0], qa);
                acc[1] = f_add(acc[1], f_add(f_sub(f_sub(qe, qa), qc), f_sub(a1, a0)));
                acc[2] = f_add(acc[2], f_add(qc, a0));
                const F fv = f_add(v0, f_mul(rk, dv));
                oV[p] = fv;
                oM[p] = f_add(m0, f_mul(rk, dm));
                if (a.has_a) oA[p] = f_lerp(a0, a1, rk);
                if (len == 2 && k == a.rounds) s_claim[j] = fv;      // the last fold of a full-length table is its claim
            }
        }
        if (!real_round) break;
        // block reduction of the three coefficients
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == 0) { lds[w * 3] = acc[0]; lds[w * 3 + 1] = acc[1]; lds[w * 3 + 2] = acc[2]; }
        __syncthreads();                                    // also publishes the folded tables
        if (tid == 0) {
            F s0 = lds[0], s1 = lds[1], s2 = lds[2];
            for (int q = 1; q < (nth >> 6); ++q) { s0 = f_add(s0, lds[3 * q]); s1 = f_add(s1, lds[3 * q + 1]); s2 = f_add(s2, lds[3 * q + 2]); }
            if (!f_is_zero(at)) at = f_mul(at, f_sub(f_one(), rprev));
            at = f_add(at, retire);
            a.poly_out[3 * (k - 1)] = s0;
            a.poly_out[3 * (k - 1) + 1] = f_sub(s1, at);
            a.poly_out[3 * (k - 1) + 2] = f_add(s2, at);
        }
        __syncthreads();                                    // lds reuse
    }
    // claims: tables shorter than the sumcheck left their value when they retired; a table whose last
    // fold happened in the final round stored it above; single-entry tables of a zero-round phase too.
    __syncthreads();
    if (tid < a.n_tab) {
        a.claims_out[tid] = s_claim[tid];
        if (a.Vu && tid == 0) *a.Vu = s_claim[0];
    }
}

}  // namespace vp

// ===================================================================================================
// Segment kernels (default batched path).
//
// The cost of this path is integer ALU, not bytes: one F-multiply is ~75 VALU instructions (12 of them
// v_mad_u64_u32), a lone wave issues one instruction every ~4 cycles, so the dependent chain of a round
// — not the 288 B per pair — sets the time of every table that does not fill the chip.  Hence:
//   * k_seg: a workgroup stages a SEGMENT of <= 1024 consecutive entries of V/mult/add in LDS (coalesced
//     1 KiB wave loads) and runs log2(segment) rounds on it without leaving the CU; ten rounds cost
//     48 B/entry of HBM reads and 48 B per 1024 entries of writes.  Inside a round the work is split at
//     F-multiply granularity with WAVE-UNIFORM roles (no divergence): wave role 0: dm*dv + fold V,
//     1: m0*v0 + fold mult, 2: m1*v1, 3: fold add + its two sums — the chain per round is two multiplies
//     instead of nine.  Round sums stay in registers (one accumulator per round, rounds unrolled) across
//     all segments a persistent workgroup processes and are reduced once at the end.
//   * k_emit: one workgroup finishes the sumcheck: it owns every table that is down to <= 2^e entries
//     (LDS resident), adds the block partials of the k_seg launches, retires finished tables into
//     add_term and writes all round polynomials and the claims.
// ===================================================================================================
namespace vp {

#ifndef VP_SEG_LOG
#define VP_SEG_LOG 10           // 1024-entry segments (120 KB of LDS, one workgroup per CU); 9 = 512 entries, two per CU, was measured: no gain
#endif
#define VP_SEG (1 << VP_SEG_LOG)
#define VP_SEG_THREADS 768          // 12 waves = 4 groups x 3 roles
#define VP_SEG_SLOTS 256            // a group covers 64 pair slots

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

// Round s of a 1024-entry segment has min(256, 512 >> s) active pair slots; their per-lane accumulators live in
// LDS at racc_off(s) + slot (767 slots per role in all), so the round loop stays ROLLED: the whole kernel is a
// few KB of code and stays in the instruction cache (the unrolled version was 62 KB and ran fetch-bound).
#define VP_SEG_RACC (VP_SEG_LOG >= 9 ? 256 * (VP_SEG_LOG - 9) + 512 : VP_SEG / 2)      /* slots per role: sum of racc_cnt over the rounds (+1) */
__device__ __forceinline__ u32 racc_cnt(int s) { return min(256u, (u32) (VP_SEG / 2) >> s); }       // active pair slots of round s
__device__ __forceinline__ u32 racc_off(int s) {                     // sum of racc_cnt over earlier rounds
    u32 o = 0;
    for (int q = 0; q < s; ++q) o += racc_cnt(q);
    return o;                                                        // SEG 1024: 0,256,512,640,...,766   SEG 512: 0,256,384,...,510
}

// Wave-uniform roles, one code path: role t folds table t (0: V, 1: mult, 2: add) and computes one of the three
// products of the pair: 0: (m1-m0)(v1-v0)   1: m0*v0   2: m1*v1.  Two multiplies per lane and pair.
struct SegLds { F bufA[3][VP_SEG]; F bufB[3][VP_SEG / 2]; F racc[4][VP_SEG_RACC]; };   // racc: [role][slot]; [3] = role 2's second sum (a0)
template <bool HAS_A>
__device__ __forceinline__ void seg_body(const SegArgs &a, u32 bid, u32 nb, SegLds &sm) {
    F (&bufA)[3][VP_SEG] = sm.bufA; F (&bufB)[3][VP_SEG / 2] = sm.bufB; F (&racc)[4][VP_SEG_RACC] = sm.racc;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(w % 3);
    const u32 pslot = (u32) ((w / 3) * 64 + lane);
    for (int i = tid; i < 4 * VP_SEG_RACC; i += VP_SEG_THREADS) (&racc[0][0])[i] = f_zero();
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
            bufA[0][i] = ok ? a.inV[base + i] : f_zero();
            bufA[1][i] = ok ? a.inM[base + i] : f_zero();
            if (HAS_A) bufA[2][i] = ok ? a.inA[base + i] : f_zero();
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < R; ++s) {
            const F *src = (s & 1) ? &bufB[0][0] : &bufA[0][0];
            F *dst = (s & 1) ? &bufA[0][0] : &bufB[0][0];
            const u32 sstr = (s & 1) ? VP_SEG / 2 : VP_SEG, dstr = (s & 1) ? VP_SEG : VP_SEG / 2;
            const F rs = a.r[s];
            const u32 n = S >> (s + 1);                                   // pairs of this round
            const u32 vs = (vseg + (1u << s) - 1) >> s;                   // valid entries of this round
            const u32 act = (vs + 1) >> 1;                                // pairs that can be non-zero
            const bool folds = HAS_A || role != 2;
            const u32 ai = racc_off(s) + pslot;
            F acc = f_zero(), acc2 = f_zero();
            for (u32 p = pslot; p < n; p += VP_SEG_SLOTS) {
                if (p >= act) { if (folds) dst[role * dstr + p] = f_zero(); continue; }
                F c0 = f_zero(), c1 = f_zero();
                if (folds) { c0 = src[role * sstr + 2 * p]; c1 = src[role * sstr + 2 * p + 1]; }
                const F d = f_sub(c1, c0);
                F x, y;
                if (role == 0) { x = f_sub(src[sstr + 2 * p + 1], src[sstr + 2 * p]); y = d; }
                else if (role == 1) { x = c0; y = src[2 * p]; }
                else { x = src[sstr + 2 * p + 1]; y = src[2 * p + 1]; }
                F qv = f_mul(x, y);
                if (role == 2 && HAS_A) { qv = f_add(qv, d); acc2 = f_add(acc2, c0); }
                acc = f_add(acc, qv);
                if (folds) dst[role * dstr + p] = f_add(c0, f_mul(rs, d));
            }
            if (pslot < n) {
                racc[role][ai] = f_add(racc[role][ai], acc);
                if (role == 2 && HAS_A) racc[3][ai] = f_add(racc[3][ai], acc2);
            }
            __syncthreads();
        }
        // the segment is down to one entry per table
        if (tid < 3 && q * S < td.valid) {
            const F *fin = (R & 1) ? bufB[tid] : bufA[tid];
            if (tid == 0) a.outV[td.off + q] = fin[0];
            else if (tid == 1) a.outM[td.off + q] = fin[0];
            else if (HAS_A) a.outA[td.off + q] = fin[0];
        }
        __syncthreads();
    }
    // per-round block partials: wave q sums one (round, array) list of <= 256 slots, then 3 lanes per round combine
    F *res = &bufA[0][0];                                        // [s][4]
    for (int t = w; t < a.n_rounds * 4; t += VP_SEG_THREADS / 64) {
        const int s = t >> 2, arr = t & 3;
        const u32 cnt = racc_cnt(s), o = racc_off(s);
        F x = f_zero();
        for (u32 i = lane; i < cnt; i += 64) x = f_add(x, racc[arr][o + i]);
        x = wave_sum63(x);
        if (lane == 63) res[t] = x;
    }
    __syncthreads();
    if (tid < 3 * a.n_rounds) {
        const int s = tid / 3, c = tid % 3;
        const F R0 = res[s * 4], R1 = res[s * 4 + 1], R2 = res[s * 4 + 2], R3 = res[s * 4 + 3];
        // a = sum dm*dv;  b = sum (m1*v1 + da) - a - sum m0*v0;  c = sum m0*v0 + sum a0
        const F x = c == 0 ? R0 : c == 1 ? f_sub(R2, f_add(R0, R1)) : f_add(R1, R3);
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
//   phase 1  all waves in parallel: reduce the block partials the k_seg launches wrote, one round per wave;
//   phase 2  only for rounds in which a table owned by this kernel has work: pair products + folds on the
//            LDS-resident tables (wave-uniform roles), wave sums parked in LDS, one barrier per round;
//   phase 3  totals per (round, coefficient) in parallel, the add_term recurrence (src/prover.cpp:445,
//            462-467) by one lane, polynomials and claims written out by parallel lanes.
// ---------------------------------------------------------------------------------------------------
#define VP_EMIT_THREADS 768         // 12 waves = 4 groups x 3 roles
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
    int pad;
};
struct EmitArgs {
    const F *V0;
    const F *buf[2][3];
    const F *r;                 // r[k-1] = challenge of round k
    const F *part; u32 part_stride;
    int n_tab, rounds, has_a, emit_log;
    u32 work_mask;              // bit k-1: some table of this kernel has pairs or retires in round k
    u32 enter_mask;             // bit k-1: some table is loaded from global memory in round k
    F *poly_out, *claims_out, *Vu;
    int n_pd;                   // launches that left block partials
    struct { int k0, nr; u32 nblk, off; } pd[VP_MAX_PD];   // rounds k0..k0+nr-1: part[off + s*nblk*3 + b*3 + c]
    EmitTab t[VP_MAX_TAB];
};

// dynamic LDS: tables [2][3][cap] | psum[32][3] | wred[32][12][3] | claim[64] | retv[64] | atv[32] | retk[64] (int)
#define VP_EMIT_LDS_EXTRA_F (32 * 3 + 32 * VP_EMIT_WAVES * 3 + VP_MAX_TAB + VP_MAX_TAB + 32)
__device__ __forceinline__ void emit_body(const EmitArgs &a, unsigned char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nth = blockDim.x;
    const u32 E = 1u << a.emit_log, cap = (u32) a.n_tab * E;
    F *lbuf = reinterpret_cast<F *>(smem_raw);
    F *psum = lbuf + (size_t) 6 * cap;
    F *wred = psum + 32 * 3;
    F *s_claim = wred + 32 * VP_EMIT_WAVES * 3;
    F *s_retv = s_claim + VP_MAX_TAB;
    F *s_at = s_retv + VP_MAX_TAB;
    int *s_retk = reinterpret_cast<int *>(s_at + 32);
    auto L = [&](int b, int tbl) { return lbuf + ((size_t) (b * 3 + tbl)) * cap; };
    const int role = __builtin_amdgcn_readfirstlane(w % 3);
    const u32 pslot = (u32) ((w / 3) * 64 + lane);
    const u32 pstride = (u32) (VP_EMIT_WAVES / 3) * 64;
    if (tid < VP_MAX_TAB) { s_claim[tid] = f_zero(); s_retv[tid] = f_zero(); s_retk[tid] = 0; }
    // ---- phase 1: block partials of the k_seg launches, one round per wave ----
    for (int k = w + 1; k <= a.rounds; k += VP_EMIT_WAVES) {
        F ca = f_zero(), cbv = f_zero(), cc = f_zero();
        for (int d = 0; d < a.n_pd; ++d) {
            if (k < a.pd[d].k0 || k >= a.pd[d].k0 + a.pd[d].nr) continue;
            const u32 nb = a.pd[d].nblk;
            const F *pp = a.part + a.pd[d].off + (size_t) (k - a.pd[d].k0) * nb * 3;
            for (u32 i = lane; i < nb; i += 64) { ca = f_add(ca, pp[3 * i]); cbv = f_add(cbv, pp[3 * i + 1]); cc = f_add(cc, pp[3 * i + 2]); }
        }
        ca = wave_sum63(ca); cbv = wave_sum63(cbv); cc = wave_sum63(cc);
        if (lane == 63) { psum[3 * (k - 1)] = ca; psum[3 * (k - 1) + 1] = cbv; psum[3 * (k - 1) + 2] = cc; }
    }
    __syncthreads();
    // ---- phase 2: rounds with table work ----
    const int nrounds = a.rounds > 0 ? a.rounds : 1;
    for (int k = 1; k <= nrounds; ++k) {
        if (!((a.work_mask >> (k - 1)) & 1u)) continue;                 // uniform
        const bool real_round = k <= a.rounds;
        const int cb = k & 1;
        if ((a.enter_mask >> (k - 1)) & 1u) {
            for (int j = 0; j < a.n_tab; ++j) {
                const EmitTab td = a.t[j];
                if (td.enter != k) continue;
                const F *gV = td.v_from_v0 ? a.V0 + td.off : a.buf[td.src][0] + td.off;
                const F *gM = a.buf[td.src][1] + td.off, *gA = a.buf[td.src][2] + td.off;
                const bool single = td.bl == 0;                          // always-initialised single entry
                for (u32 i = tid; i < td.len_enter; i += nth) {
                    const bool ok = single || i < td.valid_enter;
                    L(cb, 0)[j * E + i] = ok ? gV[i] : f_zero();
                    L(cb, 1)[j * E + i] = ok ? gM[i] : f_zero();
                    L(cb, 2)[j * E + i] = (ok && a.has_a) ? gA[i] : f_zero();
                }
            }
            __syncthreads();
        }
        F ca = f_zero(), cbv = f_zero(), cc = f_zero();
        if (real_round) {
            const F rk = a.r[k - 1];
            // global pair index -> (table, pair): tables are scanned with wave-uniform lengths
            for (u32 gp0 = 0;; gp0 += pstride) {
                const u32 gp = gp0 + pslot;
                u32 run = 0; int mj = -1; u32 mp = 0; u32 total = 0;
                for (int j = 0; j < a.n_tab; ++j) {
                    const EmitTab td = a.t[j];
                    if (k < td.enter) continue;
                    const int sh = k - td.enter;
                    const u32 len = sh < 32 ? (td.len_enter >> sh) : 0;
                    const u32 np = len >= 2 ? (len >> 1) : 0;
                    if (gp >= run && gp < run + np) { mj = j; mp = gp - run; }
                    run += np;
                }
                total = run;
                if (gp0 >= total) break;                                 // uniform
                if (mj >= 0) {
                    const int j = mj; const u32 p = mp;
                    const F *sV = L(cb, 0) + j * E, *sM = L(cb, 1) + j * E, *sA = L(cb, 2) + j * E;
                    F *dV = L(cb ^ 1, 0) + j * E, *dM = L(cb ^ 1, 1) + j * E, *dA = L(cb ^ 1, 2) + j * E;
                    if (role == 0) {
                        const F m0 = sM[2 * p], m1 = sM[2 * p + 1], v0 = sV[2 * p], v1 = sV[2 * p + 1];
                        const F dv = f_sub(v1, v0), qa = f_mul(f_sub(m1, m0), dv);
                        ca = f_add(ca, qa); cbv = f_sub(cbv, qa);
                        dV[p] = f_add(v0, f_mul(rk, dv));
                    } else if (role == 1) {
                        const F m0 = sM[2 * p], m1 = sM[2 * p + 1], v0 = sV[2 * p];
                        const F qc = f_mul(m0, v0);
                        cc = f_add(cc, qc); cbv = f_sub(cbv, qc);
                        dM[p] = f_add(m0, f_mul(rk, f_sub(m1, m0)));
                    } else {
                        cbv = f_add(cbv, f_mul(sM[2 * p + 1], sV[2 * p + 1]));
                        F o = f_zero();
                        if (a.has_a) {
                            const F a0 = sA[2 * p], a1 = sA[2 * p + 1];
                            const F da = f_sub(a1, a0);
                            cbv = f_add(cbv, da); cc = f_add(cc, a0);
                            o = f_add(a0, f_mul(rk, da));
                        }
                        dA[p] = o;
                    }
                }
            }
        }
        // single-entry tables: the entry is the claim; in a real round it retires into add_term.  One lane per table.
        if (w == VP_EMIT_WAVES - 1 && lane < a.n_tab) {
            const EmitTab td = a.t[lane];
            if (k >= td.enter) {
                const int sh = k - td.enter;
                const u32 len = sh < 32 ? (td.len_enter >> sh) : 0;
                if (len == 1) {
                    const F v = L(cb, 0)[lane * E], m = L(cb, 1)[lane * E], ad = L(cb, 2)[lane * E];
                    s_claim[lane] = v;
                    if (real_round) { s_retv[lane] = f_add(f_mul(v, m), ad); s_retk[lane] = k; }
                }
            }
        }
        if (real_round) {
            ca = wave_sum63(ca); cbv = wave_sum63(cbv); cc = wave_sum63(cc);
            if (lane == 63) {
                F *o = wred + ((size_t) (k - 1) * VP_EMIT_WAVES + w) * 3;
                o[0] = ca; o[1] = cbv; o[2] = cc;
            }
        }
        __syncthreads();
    }
    // ---- phase 3 ----
    if (tid < a.rounds * 3) {
        const int k = tid / 3, c = tid % 3;
        F t = psum[3 * k + c];
        if ((a.work_mask >> k) & 1u)
            for (int q = 0; q < VP_EMIT_WAVES; ++q) t = f_add(t, wred[((size_t) k * VP_EMIT_WAVES + q) * 3 + c]);
        psum[3 * k + c] = t;
    }
    if (w == VP_EMIT_WAVES - 1 && lane < a.rounds) {           // retire sum of round lane+1
        F t = f_zero();
        for (int j = 0; j < a.n_tab; ++j) if (s_retk[j] == lane + 1) t = f_add(t, s_retv[j]);
        s_at[lane] = t;
    }
    __syncthreads();
    if (tid == 0) {                                            // add_term recurrence
        F at = f_zero();
        for (int k = 1; k <= a.rounds; ++k) {
            if (k >= 2 && !f_is_zero(at)) at = f_mul(at, f_sub(f_one(), a.r[k - 2]));
            at = f_add(at, s_at[k - 1]);
            s_at[k - 1] = at;
        }
    }
    __syncthreads();
    if (tid < a.rounds * 3) {
        const int k = tid / 3, c = tid % 3;
        F t = psum[3 * k + c];
        if (c == 1) t = f_sub(t, s_at[k]); else if (c == 2) t = f_add(t, s_at[k]);
        a.poly_out[tid] = t;
    }
    if (tid < a.n_tab) {
        F c = s_claim[tid];
        if (a.rounds > 0) {
            const EmitTab td = a.t[tid];
            if (td.bl == a.rounds) {
                // as long as the sumcheck: folded to one entry by the last round — here, or already by k_seg
                if (td.enter > a.rounds) c = td.valid_enter ? (a.buf[td.src][0] + td.off)[0] : f_zero();
                else c = L((a.rounds + 1) & 1, 0)[tid * E];
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
// Batched ("plan") launches.  Every argument of every launch of a proof depends on the circuit only (the
// challenges are read from the device tape), so the job descriptors are built once per circuit, kept in
// device memory, and one launch runs the same kernel body for MANY independent sumchecks: block b looks up
// (job, block-in-job) in a map.  The hardware runs at most a handful of kernels at a time; with ~40
// independent sumchecks per proof, batching them side by side is what fills the chip.
// ---------------------------------------------------------------------------------------------------
struct BlkMap { u32 job, bid; };
struct GatherJob { const u32 *rowptr; const uint8_t *e_q; const u32 *e_g; const Half *H; F *M; u32 size; int pad; };
// phase 0: Liu gather (g), 1 / 2: phase inits (a).  A phase-1 job can carry the inner product V_u = sum_u eq(r_u,u) V[u]
// of its layer (same rows u): one more coalesced load and two multiplies in a kernel that waits on gathers anyway.
struct LightJob { InitArgs2 a; GatherJob g; Half dot_h; const F *dot_val; F *dot_part; int phase; u32 dot_size; };
struct ChunkJob { InitArgs2 a; const u32 *chunk_beg; const u32 *chunk_end; F *part; u32 n_chunks; int phase; };
struct CombineJob { const u32 *heavy_row; const u32 *heavy_cptr; const F *part; F *M; F *A; u32 n_heavy; int pad; };

__global__ void __launch_bounds__(VP_BLOCK) k_light_multi(const LightJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const LightJob &j = jobs[m.job];
    if (j.phase == 1) init2_light_body<1>(j.a, m.bid);
    else if (j.phase == 2) init2_light_body<2>(j.a, m.bid);
    else liu_gather_body(j.g.rowptr, j.g.e_q, j.g.e_g, j.g.H, j.g.size, j.g.M, m.bid);
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
struct DotJob { Half h; const F *val; F *part; F *out; u32 size, nblk; };
__global__ void __launch_bounds__(VP_BLOCK) k_dot_multi(const DotJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ F lds[4];
    const BlkMap m = map[blockIdx.x];
    const DotJob &j = jobs[m.job];
    F acc[1] = {f_zero()};
    for (u32 i = m.bid * blockDim.x + threadIdx.x; i < j.size; i += j.nblk * blockDim.x) acc[0] = f_add(acc[0], f_mul(half_at(j.h, i), j.val[i]));
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
__global__ void __launch_bounds__(VP_BLOCK) k_chunks_multi(const ChunkJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const ChunkJob &j = jobs[m.job];
    if (j.phase == 1) init2_chunks_body<1>(j.a, j.chunk_beg, j.chunk_end, j.n_chunks, j.part, m.bid);
    else init2_chunks_body<2>(j.a, j.chunk_beg, j.chunk_end, j.n_chunks, j.part, m.bid);
}
__global__ void __launch_bounds__(VP_BLOCK) k_combine_multi(const CombineJob *__restrict__ jobs, const BlkMap *__restrict__ map) {
    const BlkMap m = map[blockIdx.x];
    const CombineJob &j = jobs[m.job];
    init_combine_body(j.heavy_row, j.heavy_cptr, j.n_heavy, j.part, j.M, j.A, m.bid);
}
__global__ void __launch_bounds__(VP_BLOCK, 4) k_sumfold3b_multi(const SfArgs *__restrict__ jobs, const BlkMap *__restrict__ map) {
    __shared__ Sf3bLds sm;
    const BlkMap m = map[blockIdx.x];
    const SfArgs &a = jobs[m.job];
    if (a.has_a) sumfold3b_body<true>(a, m.bid, a.nblk, sm); else sumfold3b_body<false>(a, m.bid, a.nblk, sm);
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

// ===================================================================================================
// Virgo polynomial commitment, commit side (reference: lib/virgo/src/RS_polynomial.cpp, poly_commit.h,
// fri.cpp, merkle_tree.cpp, my_hhash.h).
// ===================================================================================================
namespace vp {

// ---- K7: NTT over F_p^2 ------------------------------------------------------------------------------
// One table of roots for the whole commitment: RT[j] = w^j, j < M/2, w = root of unity of order M = 2^lm
// (fieldElement::getRootOfUnity, fieldElement.cpp:237-249).  w^(M/2) = -1, so any power and any inverse
// power is one load and possibly one negation; smaller orders use strided indices.
__device__ __forceinline__ F root_pow(const F *__restrict__ RT, u32 half_m, u32 e /* < 2*half_m */) {
    return e < half_m ? RT[e] : f_neg(RT[e - half_m]);
}
__global__ void __launch_bounds__(VP_BLOCK)
k_root_table_step(F *RT, u32 have /* entries already filled, power of two */, F step /* w^have */) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < have) RT[have + i] = f_mul(RT[i], step);
}

// Batched in-LDS radix-2 NTT of size N = 2^ln <= 8192, one workgroup per transform (blockIdx.x = row,
// blockIdx.y = coset).  DIT: bit-reversed load, ln butterfly stages with one barrier each, natural-order store.
//   forward LDE mode (inverse = 0): input row `coef + row*N`, element j is first multiplied by w_M^(j*coset)
//       (the coset twist), and the N outputs are the evaluations at w_M^(32*a + coset): out[(row*ncoset + coset)*N + a].
//       A rate-1/32 Reed-Solomon encoding (fast_fourier_transform(coefs, N, 32N), RS_polynomial.cpp:26) is therefore
//       32 independent size-N transforms whose stores are fully coalesced; the codeword is kept COSET-MAJOR.
//   inverse mode: out[row*N + k] = N^-1 * sum_j in[row*N + j] * w_N^(-jk)   (inverse_fast_fourier_transform, :159-220).
struct NttArgs {
    const F *in; F *out;
    const F *RT; u32 half_m; int lm;      // root table of order M = 2^lm
    int ln;                               // transform size N = 2^ln
    int inverse;
    u32 in_stride;                        // elements between consecutive input rows
    F inv_n;                              // inverse mode: N^-1
};
__global__ void __launch_bounds__(1024) k_ntt_lds(NttArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    F *L = reinterpret_cast<F *>(smem_raw);
    const u32 N = 1u << a.ln, row = blockIdx.x, coset = blockIdx.y, tid = threadIdx.x, nth = blockDim.x;
    const u32 M = 2 * a.half_m;
    const u32 wstride = M >> a.ln;                          // w_N = w_M^wstride
    const F *src = a.in + (size_t) row * a.in_stride;
    for (u32 j = tid; j < N; j += nth) {
        F x = src[j];
        if (!a.inverse && coset) x = f_mul(x, root_pow(a.RT, a.half_m, (j * coset) & (M - 1)));
        L[a.ln ? (__brev(j) >> (32 - a.ln)) : 0u] = x;
    }
    __syncthreads();
    for (int s = 1; s <= a.ln; ++s) {
        const u32 half = 1u << (s - 1);
        const u32 tw = (N >> s) * wstride;                 // exponent step of this stage in units of w_M
        for (u32 idx = tid; idx < N / 2; idx += nth) {
            const u32 k = idx & (half - 1), i0 = ((idx >> (s - 1)) << s) | k, i1 = i0 + half;
            u32 e = k * tw;                                 // < M/2
            if (a.inverse) e = e ? M - e : 0;               // w^-e
            const F w = root_pow(a.RT, a.half_m, e);
            const F u = L[i0], v = f_mul(L[i1], w);
            L[i0] = f_add(u, v);
            L[i1] = f_sub(u, v);
        }
        __syncthreads();
    }
    F *dst = a.inverse ? a.out + (size_t) row * N : a.out + ((size_t) row * gridDim.y + coset) * N;
    for (u32 k = tid; k < N; k += nth) dst[k] = a.inverse ? f_mul(L[k], a.inv_n) : L[k];
}

// ---- K8: SHA3-256 on 64-byte messages (my_hhash.h:27-33; FIPS 202), leaf chains and Merkle levels -------
struct Dig { u64 w[4]; };
__device__ __forceinline__ u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
__device__ __forceinline__ void keccak_f1600(u64 (&A)[25]) {
    const u64 RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll 1
    for (int rnd = 0; rnd < 24; ++rnd) {
        u64 C0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], C1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21];
        u64 C2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], C3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23];
        u64 C4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];
        const u64 D0 = C4 ^ rotl64(C1, 1), D1 = C0 ^ rotl64(C2, 1), D2 = C1 ^ rotl64(C3, 1), D3 = C2 ^ rotl64(C4, 1), D4 = C3 ^ rotl64(C0, 1);
#pragma unroll
        for (int y = 0; y < 25; y += 5) { A[y] ^= D0; A[y + 1] ^= D1; A[y + 2] ^= D2; A[y + 3] ^= D3; A[y + 4] ^= D4; }
        // rho + pi
        u64 B[25];
        B[0] = A[0];
        B[10] = rotl64(A[1], 1);   B[20] = rotl64(A[2], 62);  B[5] = rotl64(A[3], 28);   B[15] = rotl64(A[4], 27);
        B[16] = rotl64(A[5], 36);  B[1] = rotl64(A[6], 44);   B[11] = rotl64(A[7], 6);   B[21] = rotl64(A[8], 55);
        B[6] = rotl64(A[9], 20);   B[7] = rotl64(A[10], 3);   B[17] = rotl64(A[11], 10); B[2] = rotl64(A[12], 43);
        B[12] = rotl64(A[13], 25); B[22] = rotl64(A[14], 39); B[23] = rotl64(A[15], 41); B[8] = rotl64(A[16], 45);
        B[18] = rotl64(A[17], 15); B[3] = rotl64(A[18], 21);  B[13] = rotl64(A[19], 8);  B[14] = rotl64(A[20], 18);
        B[24] = rotl64(A[21], 2);  B[9] = rotl64(A[22], 61);  B[19] = rotl64(A[23], 56); B[4] = rotl64(A[24], 14);
        // chi
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            A[y] = B[y] ^ (~B[y + 1] & B[y + 2]);
            A[y + 1] = B[y + 1] ^ (~B[y + 2] & B[y + 3]);
            A[y + 2] = B[y + 2] ^ (~B[y + 3] & B[y + 4]);
            A[y + 3] = B[y + 3] ^ (~B[y + 4] & B[y]);
            A[y + 4] = B[y + 4] ^ (~B[y] & B[y + 1]);
        }
        A[0] ^= RC[rnd];
    }
}
// h' = SHA3-256(m0..m3 || h)   — the 64-byte block of the leaf chains and of the Merkle nodes
__device__ __forceinline__ Dig hhash64(u64 m0, u64 m1, u64 m2, u64 m3, const Dig &h) {
    u64 A[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) A[i] = 0;
    A[0] = m0; A[1] = m1; A[2] = m2; A[3] = m3; A[4] = h.w[0]; A[5] = h.w[1]; A[6] = h.w[2]; A[7] = h.w[3];
    A[8] = 0x06;                                 // domain bits + first pad bit (byte 64)
    A[16] = 0x8000000000000000ull;               // last pad bit (byte 135, rate 136)
    keccak_f1600(A);
    Dig d; d.w[0] = A[0]; d.w[1] = A[1]; d.w[2] = A[2]; d.w[3] = A[3];
    return d;
}

// Leaf hashes of fri::request_init_commit (fri.cpp:95-124): leaf j chains the 64 slices' pairs
// (cw[s][j], cw[s][j + half]) and then the mask slice's pair (all zero here, src/prover.cpp:526).
// The codeword is coset-major: cw[(s*32 + b)*N + a] = value at position 32a + b; position j + half is (a + N/2, b).
// Thread t -> (b, a) with a fastest (coalesced loads); the digest goes to the natural leaf index 32a + b.
__global__ void __launch_bounds__(VP_BLOCK)
k_leaf_hash(const F *__restrict__ cw, u32 N, int n_slices, Dig *__restrict__ leaves) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 halfN = N >> 1;
    if (t >= 32 * halfN) return;
    const u32 a = t % halfN, b = t / halfN;
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < n_slices; ++s) {
        const F *row = cw + ((size_t) s * 32 + b) * N;
        const F x = row[a], y = row[a + halfN];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    h = hhash64(0, 0, 0, 0, h);                       // mask slice (zero polynomial)
    leaves[32 * a + b] = h;
}

// One Merkle level (merkle_tree.cpp:40-50): parent[i] = H(child[2i] || child[2i+1]); heap layout, root at index 1.
// Batched commit phase (vp_fri_commit): with every challenge known up front the folds of all levels run back to back, and
// ONE launch hashes the leaves of all levels — the 65 chained Keccak-f of a leaf are a fixed latency (~0.8 ms for a lone
// wave) that the per-step path pays once per level.
#define VP_FRI_MAX 32
struct FriLeafArgs { const F *cw[VP_FRI_MAX]; Dig *leaves[VP_FRI_MAX]; u32 N[VP_FRI_MAX]; u32 blk_start[VP_FRI_MAX + 1]; int n; };
__global__ void __launch_bounds__(VP_BLOCK) k_leaf_hash_multi(FriLeafArgs a) {
    int j = 0;
    while (j + 1 < a.n && blockIdx.x >= a.blk_start[j + 1]) ++j;
    const u32 t = (blockIdx.x - a.blk_start[j]) * blockDim.x + threadIdx.x;
    const u32 N = a.N[j], halfN = N >> 1;
    if (t >= 32 * halfN) return;
    const u32 p = t % halfN, b = t / halfN;
    const F *cw = a.cw[j];
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < 64; ++s) {
        const F *row = cw + ((size_t) s * 32 + b) * N;
        const F x = row[p], y = row[p + halfN];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    h = hhash64(0, 0, 0, 0, h);
    a.leaves[j][32 * p + b] = h;
}
struct MerkleArgs { Dig *tree[VP_FRI_MAX]; u32 count[VP_FRI_MAX]; u32 blk_start[VP_FRI_MAX + 1]; int n; };
__global__ void __launch_bounds__(VP_BLOCK) k_merkle_level_multi(MerkleArgs a) {
    int j = 0;
    while (j + 1 < a.n && blockIdx.x >= a.blk_start[j + 1]) ++j;
    const u32 i = (blockIdx.x - a.blk_start[j]) * blockDim.x + threadIdx.x, c = a.count[j];
    if (i >= c) return;
    Dig *tree = a.tree[j];
    const Dig l = tree[2 * (c + i)], r = tree[2 * (c + i) + 1];
    tree[c + i] = hhash64(l.w[0], l.w[1], l.w[2], l.w[3], r);
}
__global__ void __launch_bounds__(VP_BLOCK) k_merkle_top_multi(MerkleArgs a, Dig *roots) {     // one workgroup per tree
    Dig *tree = a.tree[blockIdx.x];
    for (u32 c = a.count[blockIdx.x] >> 1; c >= 1; c >>= 1) {
        for (u32 i = threadIdx.x; i < c; i += blockDim.x) {
            const Dig l = tree[2 * (c + i)], r = tree[2 * (c + i) + 1];
            tree[c + i] = hhash64(l.w[0], l.w[1], l.w[2], l.w[3], r);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) roots[blockIdx.x] = tree[1];
}
__global__ void __launch_bounds__(VP_BLOCK) k_merkle_level(Dig *tree, u32 level_start, u32 count) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Dig l = tree[2 * (level_start + i)], r = tree[2 * (level_start + i) + 1];
    tree[level_start + i] = hhash64(l.w[0], l.w[1], l.w[2], l.w[3], r);
}
// The top of the tree (<= 1024 leaves at `level_start`) in one workgroup.
__global__ void __launch_bounds__(VP_BLOCK) k_merkle_top(Dig *tree, u32 count) {
    for (u32 c = count >> 1; c >= 1; c >>= 1) {
        for (u32 i = threadIdx.x; i < c; i += blockDim.x) {
            const Dig l = tree[2 * (c + i)], r = tree[2 * (c + i) + 1];
            tree[c + i] = hhash64(l.w[0], l.w[1], l.w[2], l.w[3], r);
        }
        __syncthreads();
    }
}

__global__ void k_test_sha3(const u64 *__restrict__ in, u64 *__restrict__ out, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Dig h; h.w[0] = in[8 * i + 4]; h.w[1] = in[8 * i + 5]; h.w[2] = in[8 * i + 6]; h.w[3] = in[8 * i + 7];
    Dig d = hhash64(in[8 * i], in[8 * i + 1], in[8 * i + 2], in[8 * i + 3], h);
    out[4 * i] = d.w[0]; out[4 * i + 1] = d.w[1]; out[4 * i + 2] = d.w[2]; out[4 * i + 3] = d.w[3];
}

}  // namespace vp

// ---- commit_public (poly_commit.h:126-349) -------------------------------------------------------------
namespace vp {

// Products l*q on the two cosets the quotient needs: positions 16*j, j < 2N, are coset 0 (j even) and coset 16
// (j odd) of the coset-major codewords.  P[(2i)*N + a] = l_i*q_i at w_M^(32a), P[(2i+1)*N + a] at w_M^(32a+16).
__global__ void __launch_bounds__(VP_BLOCK)
k_pc_products(const F *__restrict__ lcw, const F *__restrict__ qcw, u32 N, F *__restrict__ P) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 128 * N) return;
    const u32 a = t % N, r = t / N, i = r >> 1, b = (r & 1) ? 16 : 0;
    const size_t src = ((size_t) i * 32 + b) * N + a;
    P[t] = f_mul(lcw[src], qcw[src]);
}
// With l*q = L + x^N H (deg L, H < N):  S = iNTT_N(products on coset 0) = L + H,  T_j * w_2N^-j = L_j - H_j for
// T = iNTT_N(products on coset 16).  h_coef = H = (S - D)/2  (poly_commit.h:283-287 takes the upper half of a 2N-point
// inverse transform; this is the same polynomial from two N-point ones), all_sum = (lq_coef[0] + h_coef[0]) * N = S_0 * N.
__global__ void __launch_bounds__(VP_BLOCK)
k_pc_quotient(const F *__restrict__ ST, u32 N, const F *__restrict__ RT, u32 half_m, F inv2, F n_as_f, F *__restrict__ H,
              F *__restrict__ all_sum) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 64 * N) return;
    const u32 j = t % N, i = t / N;
    const F S = ST[(size_t) (2 * i) * N + j], T = ST[(size_t) (2 * i + 1) * N + j];
    const u32 M = 2 * half_m;
    const u32 e = (16 * j) & (M - 1);                          // w_2N = w_M^16
    const F D = f_mul(T, root_pow(RT, half_m, e ? M - e : 0));
    H[t] = f_mul(f_sub(S, D), inv2);
    if (j == 0) { all_sum[i] = f_mul(S, n_as_f); all_sum[80 + i] = S; }     // [80..144): S_0 = lq_coef[0] + h_coef[0]
}
// prover::inner_prod (src/prover.cpp:532-540)
__global__ void __launch_bounds__(VP_BLOCK) k_pc_dot(const F *__restrict__ x, const F *__restrict__ y, u32 n, F *part) {
    __shared__ F lds[4];
    F acc[1] = {f_zero()};
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc[0] = f_add(acc[0], f_mul(x[i], y[i]));
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
__global__ void __launch_bounds__(VP_BLOCK) k_pc_sum_parts(const F *__restrict__ part, u32 n, F *out) {
    __shared__ F lds[4];
    F acc[1] = {f_zero()};
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) acc[0] = f_add(acc[0], part[i]);
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) *out = acc[0];
}

}  // namespace vp

// ---- virtual oracle + K9: FRI commit phase (poly_commit.h:294-318, fri.cpp:289-424) ------------------------
namespace vp {

// vo = (l*q - (x^N - 1)*h + const_i) * N * x^-1 at x = w_M^(32a+b); x^N = w_32^b depends on the coset only.
// Written in place over the q codeword (same coset-major index).
__global__ void __launch_bounds__(VP_BLOCK)
k_pc_virtual_oracle(const F *__restrict__ lcw, F *__restrict__ qcw, const F *__restrict__ hcw, const F *__restrict__ S0, u32 N,
                    const F *__restrict__ RT, u32 half_m, F n_as_f) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = 2 * (size_t) half_m;
    if (t >= 64 * M) return;
    const u32 a = (u32) (t % N), b = (u32) ((t / N) % 32), i = (u32) (t / M);
    const u32 k = 32 * a + b;
    const F xn_m1 = f_sub(root_pow(RT, half_m, (u32) ((size_t) b * N) & (u32) (M - 1)), f_one());   // w_M^(N*b) - 1
    const F g = f_sub(f_mul(lcw[t], qcw[t]), f_mul(xn_m1, hcw[t]));
    const F inv_x = f_mul(n_as_f, root_pow(RT, half_m, k ? (u32) M - k : 0));
    qcw[t] = f_mul(f_sub(g, S0[i]), inv_x);
}

// One FRI fold of all 64 slices: out[s][b][a] = 1/2 ((p + q) + mu^-1 r (p - q)), p = in[s][b][a], q = in[s][b][a + Nk/2],
// mu = w_k^(32a+b) with w_k = w_M^(2^k) the generator of the current domain (fri.cpp:312-331).
__global__ void __launch_bounds__(VP_BLOCK)
k_fri_fold(const F *__restrict__ in, F *__restrict__ out, u32 Nk, int k, const F *__restrict__ RT, u32 half_m, F r, F inv2) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const u32 No = Nk >> 1;                                   // per-coset length of the output (>= 1)
    if (t >= (size_t) 64 * 32 * No) return;
    const u32 a = (u32) (t % No), sb = (u32) (t / No);        // sb = slice * 32 + coset
    const u32 b = sb & 31;
    const u32 M = 2 * half_m;
    const u32 e = (u32) ((((unsigned long long) (32 * a + b)) << k) & (M - 1));
    const F inv_mu = root_pow(RT, half_m, e ? M - e : 0);
    F p, q;
    if (Nk >= 2) { p = in[(size_t) sb * Nk + a]; q = in[(size_t) sb * Nk + a + No]; }
    else { p = f_zero(); q = f_zero(); }
    out[t] = f_mul(inv2, f_add(f_add(p, q), f_mul(f_mul(inv_mu, r), f_sub(p, q))));
}
// The last fold leaves ONE value per coset (32 per slice); its 16 leaves pair coset b with coset b + 16.
__global__ void k_leaf_hash_final(const F *__restrict__ cw, int n_slices, Dig *__restrict__ leaves) {
    const u32 j = threadIdx.x;
    if (j >= 16) return;
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < n_slices; ++s) {
        const F x = cw[(size_t) s * 32 + j], y = cw[(size_t) s * 32 + j + 16];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    h = hhash64(0, 0, 0, 0, h);
    leaves[j] = h;
}

}  // namespace vp

// ---- transforms longer than the LDS (2^13 < N <= 2^17): N = N1 * N2 with N1 = 2^l1 <= 16, N2 = 2^13 --------
// j = j1*N2 + j2, k = k1 + N1*k2:  w_N^(jk) = w_N1^(j1 k1) * w_N^(j2 k1) * w_N2^(j2 k2).
//   k_ntt_split : per j2 an N1-point transform over the N1 rows (stride N2) in registers + the w_N^(j2 k1) twiddle
//                 (+ the coset twist for the encoder), written as [k1][j2] — every access coalesced;
//   k_ntt_lds   : N1 contiguous N2-point transforms per row (existing kernel, rows = original rows * N1);
//   k_ntt_unsplit: [k1][k2] -> natural k1 + N1*k2 through an LDS tile (+ the 1/N scale of the inverse).
namespace vp {

struct SplitArgs {
    const F *in; F *out;
    const F *RT; u32 half_m;      // root table of order M
    int ln, l1;                   // N = 2^ln, N1 = 2^l1
    int inverse;
    u32 in_stride;                // elements between input rows
    u32 ncoset;                   // forward: number of cosets (blockIdx.z = coset); inverse: 1
};
template <int L1>
__global__ void __launch_bounds__(VP_BLOCK) k_ntt_split(SplitArgs a) {
    constexpr u32 N1 = 1u << L1;
    const u32 N = 1u << a.ln, N2 = N >> L1, M = 2 * a.half_m;
    const u32 j2 = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y, coset = blockIdx.z;
    if (j2 >= N2) return;
    const u32 wN = M >> a.ln;                                  // w_N = w_M^wN
    const F *src = a.in + (size_t) row * a.in_stride;
    F x[N1];
#pragma unroll
    for (u32 j1 = 0; j1 < N1; ++j1) {
        const u32 j = j1 * N2 + j2;
        F v = src[j];
        if (!a.inverse && coset) v = f_mul(v, root_pow(a.RT, a.half_m, (u32) (((unsigned long long) j * coset) & (M - 1))));
        x[j1] = v;
    }
    // N1-point DFT, decimation in frequency in registers: natural in, bit-reversed out
#pragma unroll
    for (int s = L1; s >= 1; --s) {
        const u32 half = 1u << (s - 1);
#pragma unroll
        for (u32 idx = 0; idx < N1 / 2; ++idx) {
            const u32 k = idx & (half - 1), i0 = ((idx >> (s - 1)) << s) | k, i1 = i0 + half;
            u32 e = (k * (N1 >> s)) * (M >> L1);               // w_N1^(k * N1/2^s) in units of w_M
            if (a.inverse) e = e ? M - e : 0;
            const F u = x[i0], v = x[i1];
            x[i0] = f_add(u, v);
            x[i1] = f_mul(f_sub(u, v), root_pow(a.RT, a.half_m, e));
        }
    }
    F *dst = a.out + ((size_t) row * a.ncoset + coset) * N;
#pragma unroll
    for (u32 p = 0; p < N1; ++p) {
        const u32 k1 = __brev(p) >> (32 - (L1 ? L1 : 1)) >> (L1 ? 0 : 1);     // bit reversal of p in L1 bits
        u32 e = (u32) (((unsigned long long) j2 * k1 * wN) & (M - 1));         // w_N^(j2 k1)
        if (a.inverse) e = e ? M - e : 0;
        dst[(size_t) k1 * N2 + j2] = f_mul(x[p], root_pow(a.RT, a.half_m, e));
    }
}

// in: [rows][N1][N2] (k1-major), out: [rows][N] natural (k = k1 + N1*k2); tile of 64 k2 x N1 k1 through LDS
__global__ void __launch_bounds__(VP_BLOCK)
k_ntt_unsplit(const F *__restrict__ in, F *__restrict__ out, int ln, int l1, F scale, int do_scale) {
    __shared__ F tile[16][65];
    const u32 N = 1u << ln, N1 = 1u << l1, N2 = N >> l1;
    const u32 row = blockIdx.y, k2_0 = blockIdx.x * 64;
    const F *src = in + (size_t) row * N;
    F *dst = out + (size_t) row * N;
    for (u32 t = threadIdx.x; t < N1 * 64; t += blockDim.x) {
        const u32 k1 = t / 64, c = t % 64;
        tile[k1][c] = src[(size_t) k1 * N2 + k2_0 + c];
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < N1 * 64; t += blockDim.x) {
        const u32 c = t / N1, k1 = t % N1;
        F v = tile[k1][c];
        if (do_scale) v = f_mul(v, scale);
        dst[(size_t) (k2_0 + c) * N1 + k1] = v;
    }
}

}  // namespace vp

// ---- openings (fri::request_init_value_with_merkle, fri.cpp:148-205; fri::request_step_commit, :229-287) ----
namespace vp {
// One leaf of a committed oracle: the 64 slice pairs + the (zero) mask pair, and the Merkle path from the leaf up.
// Coset-major codeword with Nc values per coset: leaf i = 32a + b holds (cw[s][b][a], cw[s][b][a + Nc/2]); for Nc == 1
// (last FRI level) leaf j < 16 holds (cw[s][j], cw[s][j + 16]).
__global__ void k_pc_open(const F *__restrict__ cw, u32 Nc, const Dig *__restrict__ tree, u32 n_leaves, u32 leaf,
                          F *__restrict__ vals /* 65*2 */, Dig *__restrict__ path /* depth+1 */) {
    const u32 t = threadIdx.x;
    if (t < 64) {
        F x, y;
        if (Nc >= 2) { const u32 a = leaf >> 5, b = leaf & 31; const F *row = cw + ((size_t) t * 32 + b) * Nc; x = row[a]; y = row[a + (Nc >> 1)]; }
        else { x = cw[(size_t) t * 32 + leaf]; y = cw[(size_t) t * 32 + leaf + 16]; }
        vals[2 * t] = x; vals[2 * t + 1] = y;
    } else if (t == 64) {
        vals[128] = f_zero(); vals[129] = f_zero();
    }
    // path[k] = sibling at height k (k < depth), path[depth] = the leaf digest itself (the reference's com_hhash layout)
    u32 depth = 0;
    while ((1u << depth) < n_leaves) ++depth;
    if (t <= depth) {
        if (t == depth) path[t] = tree[n_leaves + leaf];
        else path[t] = tree[((n_leaves + leaf) >> t) ^ 1];
    }
}
}  // namespace vp
