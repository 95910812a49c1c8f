// Virgo polynomial commitment kernels: NTT, Keccak-f[1600] / SHA3-256, leaf hashes, Merkle trees, quotient, virtual oracle, FRI fold, openings.
// Part of the single translation unit vpgpu.hip (see vp_kernels.h for the overall layout rules).
#pragma once
#include "vp_kernels_round.h"
#include "vp_keccak_asm.h"
// Every product in this file is the plain split form: with the multiplier-shift form of the GKR kernels (vp_field.h, c31_add) the transforms
// lose more than the pointwise kernels gain (commit side 70.8 -> 88.9 ms at x1024, same call).  Undone at the end of the file.
#define f_mul f_mul_plain

// ===================================================================================================
// Virgo polynomial commitment, commit side (reference: lib/virgo/src/RS_polynomial.cpp, poly_commit.h,
// fri.cpp, merkle_tree.cpp, my_hhash.h).
// ===================================================================================================
namespace vp {

// ---- K7: NTT over F_p^2 ------------------------------------------------------------------------------
// One table of roots for the whole commitment: RT[j] = w^j, j < M/2, w = root of unity of order M = 2^lm
// (fieldElement::getRootOfUnity, fieldElement.cpp:237-249).  w^(M/2) = -1, so any power and any inverse
// power is one load and possibly one negation; smaller orders use strided indices.
// Branch-free on purpose: written as `e < half_m ? RT[e] : f_neg(RT[e - half_m])` the compiler makes two divergent arms, each with its own
// load and a wait for it — the three roots of a radix-4 butterfly (and the 32 output twiddles of a k_ntt_split column) were fetched one
// memory latency after the other.  One load from the masked index and a select-form negation let it issue all of them before the first use.
__device__ __forceinline__ F root_raw(const F *__restrict__ RT, u32 half_m /* power of two */, u32 e /* < 2*half_m */) { return RT[e & (half_m - 1)]; }
__device__ __forceinline__ F root_fin(const F &r, u32 half_m, u32 e) {                       // the raw entry -> w^e
    const bool neg = e >= half_m;
    const u64 nre = r.re ? P61 - r.re : 0, nim = r.im ? P61 - r.im : 0;
    return f_make(neg ? nre : r.re, neg ? nim : r.im);
}
__device__ __forceinline__ F root_pow(const F *__restrict__ RT, u32 half_m, u32 e) { return root_fin(root_raw(RT, half_m, e), half_m, e); }
// nothing moves across this point: the loads requested above it are all in flight before the first instruction that waits for one of them
__device__ __forceinline__ void loads_first() { __builtin_amdgcn_sched_barrier(0); }
__global__ void __launch_bounds__(VP_BLOCK)
k_root_table_step(F *RT, u32 have /* entries already filled, power of two */, F step /* w^have */) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < have) RT[have + i] = f_mul(RT[i], step);
}
// out[k] = RT[k * stride], k < n: the roots of a smaller order, contiguous
__global__ void __launch_bounds__(VP_BLOCK) k_root_compact(const F *__restrict__ RT, u32 stride, u32 n, F *__restrict__ out) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = RT[(size_t) k * stride];
}

// Batched in-LDS NTT of size N = 2^ln <= 8192, one workgroup per transform (blockIdx.x = row,
// blockIdx.y = coset).  DIT: bit-reversed load, radix-4 passes in registers (below), natural-order store.
//   forward LDE mode (inverse = 0): input row `coef + row*N`, element j is first multiplied by w_M^(j*coset)
//       (the coset twist), and the N outputs are the evaluations at w_M^(32*a + coset): out[(row*ncoset + coset)*N + a].
//       A rate-1/32 Reed-Solomon encoding (fast_fourier_transform(coefs, N, 32N), RS_polynomial.cpp:26) is therefore
//       32 independent size-N transforms whose stores are fully coalesced; the codeword is kept COSET-MAJOR.
//   inverse mode: out[row*N + k] = N^-1 * sum_j in[row*N + j] * w_N^(-jk)   (inverse_fast_fourier_transform, :159-220).
struct NttArgs {
    const F *in; F *out;
    const F *RT; u32 half_m; int lm;      // root table of order M = 2^lm (the coset twist indexes it with j * coset)
    // The butterflies' roots w_N^k come from a COMPACT table of order N (RTp[k] = w_N^k, k < half_p = N/2; pc_compact_roots): through the
    // order-M table they are M/N entries apart — one cache line each, a 2^12-point sub-transform of a x1024 commitment touched 256 KB of L2
    // for 32 KB of roots (and the 2^16 twiddles of k_ntt_split were spread over all 32 MB of it: the 1.28x HBM traffic of round 2's PMC).
    const F *RTp; u32 half_p;
    int ln;                               // transform size N = 2^ln
    int inverse;
    u32 in_stride;                        // elements between consecutive input rows
    F inv_n;                              // inverse mode: N^-1
    // Second pass of a long transform (N = N1 x 2^ln, see k_ntt_split) writing the result in natural order itself: block b is sub-transform
    // k1 = (b / 8) % N1 of long transform (b / 8 / N1) * 8 + b % 8 — the N1 sub-transforms of one long transform run on ONE XCD (workgroups
    // are dealt round-robin over the 8 XCDs), so that the 16-byte pieces they store at stride N1 (element k2 goes to k1 + N1 k2) meet in that
    // XCD's L2 and leave it as whole lines.  scat_l1 = log2 N1 (0: off), scat_rows = number of long transforms, scat_scale applied if nonzero.
    int scat_l1; u32 scat_rows; int scat_do_scale; F scat_scale;
};
// Multiplication by the primitive fourth root of unity iota = w_M^(M/4) = (0, +-1) of F_p[i]: a swap and a negation.
// plus: iota == (0, 1).  NEG: multiply by -iota = iota^-1 instead (inverse transforms).
__device__ __forceinline__ F mul_iota(const F &x, bool plus) {
    return plus ? f_make(x.im ? P61 - x.im : 0, x.re) : f_make(x.im, x.re ? P61 - x.re : 0);
}
// Radix-4 decimation in time: stage 1 of an odd ln rides on the bit-reversed load (its twiddles are all 1), then two stages per pass
// in registers: with W = w_4q^k the group (a, b, c, d) at distance q becomes
//     (a + B) + (C + D), (a - B) + iota (C - D), (a + B) - (C + D), (a - B) - iota (C - D),   B = W^2 b, C = W c, D = W^3 d
// — three multiplications per four points and two stages (radix 2: four), half the LDS passes and barriers; the first pass of an
// even ln has W = 1 and does not multiply at all.  ln = 13: 4.5 multiplications per point instead of 6.5, 7 LDS passes instead of 14.
// LDS slot of element i.  An element is 16 bytes, a row of the 32 LDS banks holds 8 of them: lanes that are 4, 16, ... or N/2 elements apart
// (the radix-4 passes with q = 1 and 4, the bit-reversed load) all fall into the same one or two slots of the row and serialise — 8-way in
// the first pass, 64-way on the load.  XOR-ing the three low index bits with the three-bit groups above them spreads every power-of-two
// stride over all 8 slots (consecutive indices stay a permutation of their aligned group of 8, so unit-stride accesses stay conflict-free).
// MEASURED (round 3, same-call A/B at x1024, tools/pc_ab.py): k_ntt_lds 10.6 ms without, 10.8 ms with — the bank conflicts are not what the
// kernel waits for (neither are the root-table gathers: compact tables changed nothing): ~500 VALU instructions per radix-4 butterfly (three
// multiplications at ~85, eight canonical add/sub at ~20) at the ~5 cycles per wave-instruction every kernel of this library issues at ARE
// its 41 us per 2^12-point workgroup.  The swizzle is gone; ntt_slot stays as the one place it would go.
__device__ __forceinline__ u32 ntt_slot(u32 i) { return i; }
__global__ void __launch_bounds__(1024) k_ntt_lds(NttArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    F *L = reinterpret_cast<F *>(smem_raw);
    const u32 N = 1u << a.ln, coset = blockIdx.y, tid = threadIdx.x, nth = blockDim.x;
    u32 row = blockIdx.x, long_row = 0, k1 = 0;
    if (a.scat_l1) {
        const u32 x = blockIdx.x & 7u, sidx = blockIdx.x >> 3;
        k1 = sidx & ((1u << a.scat_l1) - 1);
        long_row = (sidx >> a.scat_l1) * 8 + x;
        if (long_row >= a.scat_rows) return;                  // uniform: padding of the last group of eight
        row = (long_row << a.scat_l1) + k1;
    }
    const u32 M = 2 * a.half_m;
    const u32 Mp = 2 * a.half_p, wstride = Mp >> a.ln;      // w_N = w_Mp^wstride in the pass table (wstride = 1 for the compact one)
    const F *src = a.in + (size_t) row * a.in_stride;
    const bool twist = !a.inverse && coset;
    int s = 1;
    if (a.ln & 1) {
        // x[j] and x[j + N/2] land on the adjacent bit-reversed slots 2m, 2m+1: stage 1 (twiddle 1) on the way in
        for (u32 j = tid; j < N / 2; j += nth) {
            F x = src[j], y = src[j + N / 2];
            if (twist) {
                x = f_mul(x, root_pow(a.RT, a.half_m, (j * coset) & (M - 1)));
                y = f_mul(y, root_pow(a.RT, a.half_m, ((j + N / 2) * coset) & (M - 1)));
            }
            const u32 m = __brev(j) >> (32 - a.ln);         // even
            L[ntt_slot(m)] = f_add(x, y); L[ntt_slot(m + 1)] = f_sub(x, y);
        }
        s = 2;
    } else {
        for (u32 j = tid; j < N; j += nth) {
            F x = src[j];
            if (twist) x = f_mul(x, root_pow(a.RT, a.half_m, (j * coset) & (M - 1)));
            L[ntt_slot(a.ln ? (__brev(j) >> (32 - a.ln)) : 0u)] = x;
        }
    }
    __syncthreads();
    const bool iota_plus = a.RT[a.half_m >> 1].im == 1;    // w_M^(M/4) is (0, 1) or (0, p - 1)  (uniform; M >= 4 whenever a pass runs)
    const bool ip = a.inverse ? !iota_plus : iota_plus;
    for (; s + 1 <= a.ln; s += 2) {
        const u32 q = 1u << (s - 1);
        const u32 tw = (N >> (s + 1)) * wstride;           // W = w_4q^k = w_M^(k * tw)
        for (u32 idx = tid; idx < N / 4; idx += nth) {
            const u32 k = idx & (q - 1), i0 = ((idx >> (s - 1)) << (s + 1)) | k;
            const u32 p0 = ntt_slot(i0), p1 = ntt_slot(i0 + q), p2 = ntt_slot(i0 + 2 * q), p3 = ntt_slot(i0 + 3 * q);
            F x0 = L[p0], x1 = L[p1], x2 = L[p2], x3 = L[p3];
            if (q > 1) {                                    // uniform: the first pass of an even ln has W = 1
                const u32 e = k * tw;                       // < Mp/4
                const u32 e1 = a.inverse ? (e ? Mp - e : 0) : e, e2 = a.inverse ? (e ? Mp - 2 * e : 0) : 2 * e,
                          e3 = a.inverse ? (e ? Mp - 3 * e : 0) : 3 * e;
                const F w1 = root_raw(a.RTp, a.half_p, e1), w2 = root_raw(a.RTp, a.half_p, e2), w3 = root_raw(a.RTp, a.half_p, e3);
                loads_first();                              // three root gathers and four LDS reads in flight together
                x1 = f_mul(x1, root_fin(w2, a.half_p, e2));
                x2 = f_mul(x2, root_fin(w1, a.half_p, e1));
                x3 = f_mul(x3, root_fin(w3, a.half_p, e3));
            }
            const F s0 = f_add(x0, x1), d0 = f_sub(x0, x1), s1 = f_add(x2, x3), d1 = mul_iota(f_sub(x2, x3), ip);
            L[p0] = f_add(s0, s1); L[p1] = f_add(d0, d1); L[p2] = f_sub(s0, s1); L[p3] = f_sub(d0, d1);
        }
        __syncthreads();
    }
    if (a.scat_l1) {
        F *dst = a.out + ((size_t) long_row << (a.ln + a.scat_l1)) + k1;
        for (u32 k = tid; k < N; k += nth) dst[(size_t) k << a.scat_l1] = a.scat_do_scale ? f_mul(L[ntt_slot(k)], a.scat_scale) : L[ntt_slot(k)];
        return;
    }
    F *dst = a.inverse ? a.out + (size_t) row * N : a.out + ((size_t) row * gridDim.y + coset) * N;
    for (u32 k = tid; k < N; k += nth) dst[k] = a.inverse ? f_mul(L[ntt_slot(k)], a.inv_n) : L[ntt_slot(k)];
}

// ---- K8: SHA3-256 on 64-byte messages (my_hhash.h:27-33; FIPS 202), leaf chains and Merkle levels -------
struct Dig { u64 w[4]; };
// The state is kept as 2 x 25 32-bit halves: gfx950 has no 64-bit logic or rotate instructions, and a 64-bit rotation written
// on u64 compiles to two 64-bit shifts and two ORs (4-5 instructions); on halves it is two v_alignbit_b32.  Per round:
// 120 v_bitop3_b32 (three-input xor; chi = a ^ (~b & c) in one instruction) + 58 v_alignbit_b32 + 2 xor = 180 instructions, four
// rounds per loop iteration (was ~313 VALU instructions per round with the u64 formulation, 25 of them register moves).
__device__ __forceinline__ u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
#ifndef VP_KECCAK_UNROLL
#define VP_KECCAK_UNROLL 4
#endif
template <int N> __device__ __forceinline__ void rot(u32 lo, u32 hi, u32 &olo, u32 &ohi) {      // (ohi:olo) = rotl64(hi:lo, N)
    if (N == 0) { olo = lo; ohi = hi; }
    else if (N == 32) { olo = hi; ohi = lo; }
    else if (N < 32) { ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - N); olo = __builtin_amdgcn_alignbit(lo, hi, 32 - N); }
    else { ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - N); olo = __builtin_amdgcn_alignbit(hi, lo, 64 - N); }
}
__device__ __forceinline__ u32 chi32(u32 a, u32 b, u32 c) { return a ^ (~b & c); }            // one v_bitop3_b32
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }   // a ^ b ^ c in one instruction
__device__ __constant__ const u64 KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
// four rounds on the 2 x 25 halves, rnd0 .. rnd0 + 3
__device__ __forceinline__ void keccak_rounds4(u32 (&al)[25], u32 (&ah)[25], int rnd0) {
    const u64 *RC = KECCAK_RC;
#pragma unroll
    for (int rnd = rnd0; rnd < rnd0 + 4; ++rnd) {
        // theta
        u32 cl[5], ch[5], rl[5], rh[5];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            cl[x] = xor3(xor3(al[x], al[x + 5], al[x + 10]), al[x + 15], al[x + 20]);
            ch[x] = xor3(xor3(ah[x], ah[x + 5], ah[x + 10]), ah[x + 15], ah[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; ++x) rot<1>(cl[x], ch[x], rl[x], rh[x]);
#pragma unroll
        for (int i = 0; i < 25; ++i) {                       // A ^= D, D[x] = C[x-1] ^ rotl(C[x+1], 1), without materialising D
            al[i] = xor3(al[i], cl[(i + 4) % 5], rl[(i + 1) % 5]);
            ah[i] = xor3(ah[i], ch[(i + 4) % 5], rh[(i + 1) % 5]);
        }
        // rho + pi
        u32 bl[25], bh[25];
        rot<0>(al[0], ah[0], bl[0], bh[0]);
        rot<1>(al[1], ah[1], bl[10], bh[10]);
        rot<62>(al[2], ah[2], bl[20], bh[20]);
        rot<28>(al[3], ah[3], bl[5], bh[5]);
        rot<27>(al[4], ah[4], bl[15], bh[15]);
        rot<36>(al[5], ah[5], bl[16], bh[16]);
        rot<44>(al[6], ah[6], bl[1], bh[1]);
        rot<6>(al[7], ah[7], bl[11], bh[11]);
        rot<55>(al[8], ah[8], bl[21], bh[21]);
        rot<20>(al[9], ah[9], bl[6], bh[6]);
        rot<3>(al[10], ah[10], bl[7], bh[7]);
        rot<10>(al[11], ah[11], bl[17], bh[17]);
        rot<43>(al[12], ah[12], bl[2], bh[2]);
        rot<25>(al[13], ah[13], bl[12], bh[12]);
        rot<39>(al[14], ah[14], bl[22], bh[22]);
        rot<41>(al[15], ah[15], bl[23], bh[23]);
        rot<45>(al[16], ah[16], bl[8], bh[8]);
        rot<15>(al[17], ah[17], bl[18], bh[18]);
        rot<21>(al[18], ah[18], bl[3], bh[3]);
        rot<8>(al[19], ah[19], bl[13], bh[13]);
        rot<18>(al[20], ah[20], bl[14], bh[14]);
        rot<2>(al[21], ah[21], bl[24], bh[24]);
        rot<61>(al[22], ah[22], bl[9], bh[9]);
        rot<56>(al[23], ah[23], bl[19], bh[19]);
        rot<14>(al[24], ah[24], bl[4], bh[4]);
        // chi + iota
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
#pragma unroll
            for (int x = 0; x < 5; ++x) {
                al[y + x] = chi32(bl[y + x], bl[y + (x + 1) % 5], bl[y + (x + 2) % 5]);
                ah[y + x] = chi32(bh[y + x], bh[y + (x + 1) % 5], bh[y + (x + 2) % 5]);
            }
        }
        al[0] ^= (u32) RC[rnd]; ah[0] ^= (u32) (RC[rnd] >> 32);
    }
}
// The first and the last group of four rounds are peeled out of the loop: in the first one the compiler folds the thirteen lanes of
// the padded 64-byte message that are constants (hhash64), in the last one it drops everything that does not reach the four output
// lanes of SHA3-256 (most of round 24's rho / pi / chi).
__device__ __forceinline__ void keccak_f1600(u64 (&A)[25]) {
    u32 al[25], ah[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) { al[i] = (u32) A[i]; ah[i] = (u32) (A[i] >> 32); }
    keccak_rounds4(al, ah, 0);
#pragma unroll 1
    for (int rnd0 = 4; rnd0 < 20; rnd0 += 4) keccak_rounds4(al, ah, rnd0);
    keccak_rounds4(al, ah, 20);
#pragma unroll
    for (int i = 0; i < 25; ++i) A[i] = ((u64) ah[i] << 32) | al[i];
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
k_leaf_hash_c(const F *__restrict__ cw, u32 N, int n_slices, Dig *__restrict__ leaves) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 halfN = N >> 1;
    if (t >= 32 * halfN) return;
    const u32 a = t % halfN, b = t / halfN;
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < n_slices; ++s) {
        const F *row = cw + ((size_t) s * 32 + b) * N;
        F x = f_zero(), y = f_zero();
        if (VP_CHK(a + halfN < N && b < 32, 7, a, b, N)) { x = row[a]; y = row[a + halfN]; }
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
struct FriLeafArgs { const F *cw[VP_FRI_MAX]; Dig *leaves[VP_FRI_MAX]; u32 N[VP_FRI_MAX]; u32 blk_start[VP_FRI_MAX + 1]; u32 leaf_start[VP_FRI_MAX + 1]; int n; };     // blk_start: k_leaf_hash_multi_c, leaf_start: k_leaf_hash_multi
__global__ void __launch_bounds__(VP_BLOCK) k_leaf_hash_multi_c(FriLeafArgs a) {
    int j = 0;
    while (j + 1 < a.n && blockIdx.x >= a.blk_start[j + 1]) ++j;
    const u32 t = (blockIdx.x - a.blk_start[j]) * blockDim.x + threadIdx.x;
    const u32 N = a.N[j], halfN = N >> 1;
    const F *cw = a.cw[j];
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    if (N == 1) {                                     // last level: one value per coset, leaf j pairs cosets j and j + 16
        if (t >= 16) return;
        for (int s = 0; s < 64; ++s) {
            const F x = cw[(size_t) s * 32 + t], y = cw[(size_t) s * 32 + t + 16];
            h = hhash64(x.re, x.im, y.re, y.im, h);
        }
        h = hhash64(0, 0, 0, 0, h);
        a.leaves[j][t] = h;
        return;
    }
    if (t >= 32 * halfN) return;
    const u32 p = t % halfN, b = t / halfN;
    for (int s = 0; s < 64; ++s) {
        const F *row = cw + ((size_t) s * 32 + b) * N;
        const F x = row[p], y = row[p + halfN];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    h = hhash64(0, 0, 0, 0, h);
    a.leaves[j][32 * p + b] = h;
}
struct MerkleArgs { Dig *tree[VP_FRI_MAX]; u32 count[VP_FRI_MAX]; u32 blk_start[VP_FRI_MAX + 1]; int n; };
// Round 4: the same leaf chains by the generated fixed-register block (vp_keccak_asm.h, tools/gen_keccak_asm.py): workgroups of 1024 threads, one per
// CU, whose sixteen waves rotate (v_alignbit_b32, half rate) and do logic (v_bitop3_b32 / v_xor_b32, full rate) IN PHASE — a SIMD with waves in
// both kinds of code at once issues everything at the rotation's rate.  x1024: 14.3 -> 11.6 ms per 2^21-leaf tree, digests unchanged.  A thread past
// the end runs the chain of the last leaf (it must keep step with its workgroup) and stores nothing.  k_leaf_hash_c / k_leaf_hash_multi_c above are
// the compiler's form of the same chains: the cross-check (leaf_asm = 0) and the checked build.
#ifdef VP_LEAF_STAMPS
// Diagnostic build only (tools/leaf_in_step.py; MI355X_MICROARCH.md, DVFS item 6): every workgroup of the leaf-hash kernels leaves
// {s_memtime, s_memrealtime} from before and after its chains in a buffer of its own that nothing else reads; the effective shader clock of the
// workgroup is d(memtime) / d(memrealtime) x 100 MHz.  The product build has none of this.
__device__ unsigned long long g_leaf_stamps[4 * 65536];
__device__ unsigned int g_leaf_stamp_n;
#define VP_LEAF_STAMP_BEGIN const u64 st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define VP_LEAF_STAMP_END(tag) if (threadIdx.x == 0) { const u64 st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime(); \
        const u32 sl = atomicAdd(&g_leaf_stamp_n, 1u) & 65535u; g_leaf_stamps[4 * sl] = st_t0; g_leaf_stamps[4 * sl + 1] = st_r0; g_leaf_stamps[4 * sl + 2] = st_t1; \
        g_leaf_stamps[4 * sl + 3] = (st_r1 << 1) | (tag); }
#else
#define VP_LEAF_STAMP_BEGIN
#define VP_LEAF_STAMP_END(tag)
#endif
__global__ void __launch_bounds__(VP_LEAF_ASM_THREADS)
k_leaf_hash(const F *__restrict__ cw, u32 N, int n_slices, Dig *__restrict__ leaves) {
    VP_LEAF_STAMP_BEGIN
    const u32 halfN = N >> 1, total = 32 * halfN;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 tc = t < total ? t : total - 1;
    const u32 a = tc % halfN, b = tc / halfN;
    const F *x = cw + (size_t) b * N + a;
    vp_leaf_chain_asm(x, x + halfN, 32u * N * 16u, (unsigned) n_slices, leaves + (32 * a + b), t < total ? 1u : 0u);
    VP_LEAF_STAMP_END(0u)
}
// All levels of the FRI commit phase in one launch: thread g takes leaf g of the levels laid end to end (leaf_start), so that the launch is exactly
// ceil(leaves / workgroup) workgroups — a level of 16 leaves does not cost a workgroup (and with it a CU for the whole chain) of its own.
__global__ void __launch_bounds__(VP_LEAF_ASM_THREADS) k_leaf_hash_multi(FriLeafArgs a) {
    VP_LEAF_STAMP_BEGIN
    const u32 total = a.leaf_start[a.n];
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x, gc = g < total ? g : total - 1;
    int j = 0;
    while (j + 1 < a.n && gc >= a.leaf_start[j + 1]) ++j;
    const u32 t = gc - a.leaf_start[j], N = a.N[j];
    const F *cw = a.cw[j];
    // the last level (one value per coset, leaf t pairs cosets t and t + 16 of every slice) is the same chain with other strides
    const bool last = N == 1;
    const u32 halfN = last ? 16u : N >> 1;
    const u32 p = last ? t : t % halfN, b = last ? 0u : t / halfN;
    const F *x = cw + (size_t) b * N + p;
    vp_leaf_chain_asm(x, x + halfN, 32u * N * 16u, 64u, a.leaves[j] + (last ? t : 32 * p + b), g < total ? 1u : 0u);
    VP_LEAF_STAMP_END(1u)
}
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
// q0 / qscal != NULL: the public vector is a TENSOR (every slice a scalar multiple of slice 0: pub[i N + k] = qscal[i] pub[k] — the protocol's
// own public vector, the eq table of the opening point, src/verifier.cpp:368-369, always is), so only slice 0 was encoded (q0, one
// coset-major codeword) and q_i = qscal[i] * q0: 32 instead of 2048 transforms, 1/64 of the q traffic, one more multiplication here.
__global__ void __launch_bounds__(VP_BLOCK)
k_pc_products(const F *__restrict__ lcw, const F *__restrict__ qcw, u32 N, F *__restrict__ P, u32 n_slices, const F *__restrict__ q0,
              const F *__restrict__ qscal) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n_slices * N) return;
    const u32 a = t % N, r = t / N, i = r >> 1, b = (r & 1) ? 16 : 0;
    const size_t src = ((size_t) i * 32 + b) * N + a;
    const F q = q0 ? f_mul(qscal[i], q0[(size_t) b * N + a]) : qcw[src];
    P[t] = f_mul(lcw[src], q);
}
// Is the public vector a tensor with a non-zero corner?  pub[i N + k] * pub[0] == pub[i N] * pub[k] for every i >= 1, k (exact in a field:
// with pub[0] != 0 this IS proportionality of slice i to slice 0 with factor pub[i N] / pub[0]).  flag |= 1 on any violation.
__global__ void __launch_bounds__(VP_BLOCK) k_pc_rank1_check(const F *__restrict__ pub, u32 N, u32 n_slices, int *__restrict__ flag) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t) n_slices * N) return;
    const u32 k = (u32) (t % N), i = (u32) (t / N);
    if (i == 0) return;
    if (!f_eq(f_mul(pub[t], pub[0]), f_mul(pub[(size_t) i * N], pub[k]))) atomicOr(flag, 1);
}
// With l*q = L + x^N H (deg L, H < N):  S = iNTT_N(products on coset 0) = L + H,  T_j * w_2N^-j = L_j - H_j for
// T = iNTT_N(products on coset 16).  h_coef = H = (S - D)/2  (poly_commit.h:283-287 takes the upper half of a 2N-point
// inverse transform; this is the same polynomial from two N-point ones), all_sum = (lq_coef[0] + h_coef[0]) * N = S_0 * N.
__global__ void __launch_bounds__(VP_BLOCK)
k_pc_quotient(const F *__restrict__ ST, u32 N, const F *__restrict__ RT, u32 half_m, F inv2, F n_as_f, F *__restrict__ H,
              F *__restrict__ all_sum, u32 n_slices) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slices * N) return;
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
                    const F *__restrict__ RT, u32 half_m, F n_as_f, u32 n_slices, const F *__restrict__ q0, const F *__restrict__ qscal) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t M = 2 * (size_t) half_m;
    if (t >= n_slices * M) return;
    const u32 a = (u32) (t % N), b = (u32) ((t / N) % 32), i = (u32) (t / M);
    const u32 k = 32 * a + b;
    const F xn_m1 = f_sub(root_pow(RT, half_m, (u32) ((size_t) b * N) & (u32) (M - 1)), f_one());   // w_M^(N*b) - 1
    const F q = q0 ? f_mul(qscal[i], q0[t - (size_t) i * M]) : qcw[t];           // tensor public vector: q_i = qscal[i] * q0 (k_pc_products)
    const F g = f_sub(f_mul(lcw[t], q), f_mul(xn_m1, hcw[t]));
    const F inv_x = f_mul(n_as_f, root_pow(RT, half_m, k ? (u32) M - k : 0));
    qcw[t] = f_mul(f_sub(g, S0[i]), inv_x);
}

// One FRI fold of all 64 slices: out[s][b][a] = 1/2 ((p + q) + mu^-1 r (p - q)), p = in[s][b][a], q = in[s][b][a + Nk/2],
// mu = w_k^(32a+b) with w_k = w_M^(2^k) the generator of the current domain (fri.cpp:312-331).
// Position-sharded commitment (vpgpu_pc_shard.inc): a rank holds the positions a = a' * 2^lw + rank of every coset; Nk is then the
// LOCAL per-coset length and the twiddle uses the global position.  lw = rank = 0: the whole codeword.
// A thread takes VP_FOLD_SPT slices of one position: mu^-1 r / 2 is per position, so a fold costs ONE multiplication per output (and a halving) instead of three.
#define VP_FOLD_SPT 4
__global__ void __launch_bounds__(VP_BLOCK)
k_fri_fold(const F *__restrict__ in, F *__restrict__ out, u32 Nk, int k, const F *__restrict__ RT, u32 half_m, F r, F inv2, int lw, u32 rank) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const u32 No = Nk >> 1;                                   // per-coset length of the output (>= 1)
    if (t >= (size_t) (64 / VP_FOLD_SPT) * 32 * No) return;
    const u32 al = (u32) (t % No), gb = (u32) (t / No), b = gb & 31, ig = gb >> 5;
    const u32 a = (al << lw) + rank;
    const u32 M = 2 * half_m;
    const u32 e = (u32) ((((unsigned long long) (32 * a + b)) << k) & (M - 1));
    const F inv_mu = root_pow(RT, half_m, e ? M - e : 0);
    F p[VP_FOLD_SPT], q[VP_FOLD_SPT];
#pragma unroll
    for (int j = 0; j < VP_FOLD_SPT; ++j) {
        const size_t sb = (size_t) (ig * VP_FOLD_SPT + j) * 32 + b;
        if (Nk >= 2) { p[j] = in[sb * Nk + al]; q[j] = in[sb * Nk + al + No]; }
        else { p[j] = f_zero(); q[j] = f_zero(); }
    }
    loads_first();
    const F c = f_mul(inv2, f_mul(inv_mu, r));                // mu^-1 r / 2
#pragma unroll
    for (int j = 0; j < VP_FOLD_SPT; ++j) {
        const size_t sb = (size_t) (ig * VP_FOLD_SPT + j) * 32 + b;
        out[sb * No + al] = f_add(f_half(f_add(p[j], q[j])), f_mul(c, f_sub(p[j], q[j])));      // 1/2 ((p + q) + mu^-1 r (p - q))
    }
}
// Round 4: the FIRST fold straight from the three committed codewords — the virtual oracle (k_pc_virtual_oracle) is never written.  With
// G(a) = l q - (x^N - 1) h - S0 at position (b, a) and X = N x^-1, the oracle is G X; its partner at a + N/2 sits at -x (w_M^(16 N) = -1), so
//   p = G(a) X,  q' = -G(a') X   and   1/2 ((p + q') + mu^-1 r (p - q')) = (X / 2) ((G(a) - G(a')) + x^-1 r (G(a) + G(a'))),  mu = x at level 0
// — the same field element as k_fri_fold(k = 0) on the materialised oracle (every operation is exact), ten multiplications per output instead
// of thirteen, and 64 M x 16 B less written and read back (x1024: 8.6 GB).  Openings never read the oracle itself (fri.cpp:148-287 open l, h
// and the folded levels), so nothing else needs it.  Unsharded commitment only (lw = 0).
// A thread takes VP_VO_SPT slices of one position (b, a): x^-1, x^-1 r and (N/2) x^-1 depend on the position only (three of the eleven multiplications per
// output), and with a tensor public vector so do the two loads of its one encoded slice.
#ifndef VP_VO_SPT
#define VP_VO_SPT 4
#endif
__global__ void __launch_bounds__(VP_BLOCK)
k_fri_fold0_vo(const F *__restrict__ lcw, const F *__restrict__ qcw, const F *__restrict__ hcw, const F *__restrict__ S0, F *__restrict__ out, u32 N,
               const F *__restrict__ RTn /* w_N^k, k < N */, const F *__restrict__ cb /* [b] = w_M^-b, [32 + b] = w_32^b - 1 */, F r, F half_n /* N / 2 */,
               const F *__restrict__ q0, const F *__restrict__ qscal) {
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const u32 No = N >> 1;
    if (t >= (size_t) (64 / VP_VO_SPT) * 32 * No) return;
    const u32 al = (u32) (t % No), sb = (u32) (t / No), b = sb & 31, ig = sb >> 5;
    const F wa = RTn[(N - al) & (N - 1)];                                     // w_N^-a = w_M^-(32 a): contiguous along the lanes
    F q0a = f_zero(), q0b = f_zero();
    if (q0) { const size_t o = (size_t) b * N + al; q0a = q0[o]; q0b = q0[o + No]; }
    F la[VP_VO_SPT], lb[VP_VO_SPT], ha[VP_VO_SPT], hb[VP_VO_SPT], qa[VP_VO_SPT], qb[VP_VO_SPT];
#pragma unroll
    for (int k = 0; k < VP_VO_SPT; ++k) {
        const size_t p0 = ((size_t) (ig * VP_VO_SPT + k) * 32 + b) * N + al, p1 = p0 + No;
        la[k] = lcw[p0]; lb[k] = lcw[p1]; ha[k] = hcw[p0]; hb[k] = hcw[p1];
        if (!q0) { qa[k] = qcw[p0]; qb[k] = qcw[p1]; }
    }
    loads_first();
    const F inv_x = f_mul(wa, cb[b]);                                         // x^-1, x = w_M^(32 a + b)
    const F xn_m1 = cb[32 + b];                                               // x^N - 1 = w_32^b - 1: the same for both positions
    const F xr = f_mul(inv_x, r), hx = f_mul(half_n, inv_x);
#pragma unroll
    for (int k = 0; k < VP_VO_SPT; ++k) {
        const u32 i = ig * VP_VO_SPT + k;
        if (q0) { const F sc = qscal[i]; qa[k] = f_mul(sc, q0a); qb[k] = f_mul(sc, q0b); }
        const F s0 = S0[i];
        const F Ga = f_sub(f_sub(f_mul(la[k], qa[k]), f_mul(xn_m1, ha[k])), s0), Gb = f_sub(f_sub(f_mul(lb[k], qb[k]), f_mul(xn_m1, hb[k])), s0);
        const F D = f_sub(Ga, Gb), S = f_add(Ga, Gb);
        out[((size_t) i * 32 + b) * No + al] = f_mul(hx, f_add(D, f_mul(xr, S)));
    }
}
// Round 5: the first THREE folds in one pass (vp_fri_commit has every challenge before it starts).  A workgroup owns 64 consecutive positions al of one coset at
// the eight offsets al + j N/8: wave v (of four) does fold 0 on the pair (v, v + 4) exactly as k_fri_fold0_vo does — same loads, 1 KB contiguous per wave and
// instruction — then hands its level-1 values through LDS: waves 0, 1 do fold 1 on (v, v + 2), wave 0 fold 2 on (0, 1).  The level-1 and level-2 codewords are
// written (they are oracles: leaf hashes and openings read them) but never read back: 10.7 + 3.2 + 1.6 GB -> 12.4 GB at x1024, and two launches less.  Every
// operation is the exact field operation of k_fri_fold0_vo / k_fri_fold(k = 1, 2) on the same operands (mu_1^-1 = x^-2 and mu_2^-1 = x^-4 by squaring instead of
// a table read: the same field elements), so the three codewords are the same bytes.
// Round 6: the launch was VALU-issue bound on its own instruction stream (profiles/r06_isa_mix_and_issue_share.txt: the SIMDs spent the whole launch issuing),
// so the level-0 value of the tensor form is computed from FEWER products.  With q_i = c_i q0 and u = hx + hx xr, w = hx xr - hx:
//     f1 = hx (D + xr S),  D = Ga - Gb,  S = Ga + Gb,  Ga = la c q0a - xn ha - s0,  Gb = lb c q0b - xn hb - s0
//        = c (A la + B lb) - (C ha + E hb) - K s0,      A = u q0a,  B = w q0b,  C = xn u,  E = xn w,  K = 2 hx xr      (position constants)
// — three two-product sums with ONE reduction each (f_dot2cc) per slice instead of eight F-multiplications; the same field element, so the same bytes.
// GRP consecutive slice groups per workgroup (a loop) spread the position constants over GRP * VP_VO_SPT slices: VP_VO_GRP (8: same call at x1024, fold family
// 4.02-4.29 ms -> 3.43 / 3.52 / 3.17-3.27 / 3.01-3.06 / 3.09 ms with 1 / 2 / 4 / 8 / 16 groups, profiles/r06_ab_fri_fold0_dot2_x1024.txt) where the launch still
// has >= 2048 workgroups, one group per workgroup below that.
#ifndef VP_VO_GRP
#define VP_VO_GRP 8
#endif
static_assert(64 % (VP_VO_SPT * VP_VO_GRP) == 0, "the 64 slices are dealt to workgroups in groups of VP_VO_SPT * VP_VO_GRP");
template <bool TENSOR, int GRP> __global__ void __launch_bounds__(256)
k_fri_fold0_vo3(const F *__restrict__ lcw, const F *__restrict__ qcw, const F *__restrict__ hcw, const F *__restrict__ S0, F *__restrict__ out1, F *__restrict__ out2,
                F *__restrict__ out3, u32 N, const F *__restrict__ RTn, const F *__restrict__ cb, F r0, F r1, F r2, F half_n, F inv2,
                const F *__restrict__ q0, const F *__restrict__ qscal) {
    __shared__ F x1[VP_VO_SPT][2][64], x2[VP_VO_SPT][64];
    const u32 E = N >> 3, No = N >> 1, N2 = N >> 2;
    const u32 lane = threadIdx.x & 63, v = threadIdx.x >> 6;
    const u32 per = E >> 6;                                                   // workgroups per (slice group, coset): E >= 64
    const u32 al = (blockIdx.x % per) * 64 + lane, sb = blockIdx.x / per, b = sb & 31, ig0 = (sb >> 5) * GRP;
    const u32 a = al + v * E;                                                 // level-0 position (< N / 2) and level-1 position of its fold
    const F inv_x = f_mul(RTn[(N - a) & (N - 1)], cb[b]);                     // x^-1, x = w_M^(32 a + b)
    const F xn_m1 = cb[32 + b];
    const F xr = f_mul(inv_x, r0), hx = f_mul(half_n, inv_x);
    F cA = f_zero(), cB = f_zero(), cC = f_zero(), cE = f_zero(), cK = f_zero();
    if (TENSOR) {
        const size_t o = (size_t) b * N + a;
        const F hxr = f_mul(hx, xr), u = f_add(hx, hxr), w = f_sub(hxr, hx);
        cA = f_mul(u, q0[o]); cB = f_mul(w, q0[o + No]); cC = f_mul(xn_m1, u); cE = f_mul(xn_m1, w); cK = f_add(hxr, hxr);
    }
    // fold 1 (Nk = N / 2): level-1 positions a = al + v E and a + N / 4 = al + (v + 2) E;  mu^-1 = w_M^-(2 (32 a + b)) = x^-2.  Waves 0, 1 work; every
    // wave stays to the last barrier (a barrier in a workgroup some of whose waves have returned is undefined by HIP's rules, whatever gfx950 does)
    const F m1 = f_mul(inv_x, inv_x);
    const F c1 = f_mul(inv2, f_mul(m1, r1));
    // fold 2 (Nk = N / 4): level-2 positions al and al + E;  mu^-1 = x^-4 at al.  Wave 0.
    const F c2 = f_mul(inv2, f_mul(f_mul(m1, m1), r2));
#pragma unroll 1
    for (u32 g = 0; g < (u32) GRP; ++g) {
        const u32 ig = ig0 + g;
        F la[VP_VO_SPT], lb[VP_VO_SPT], ha[VP_VO_SPT], hb[VP_VO_SPT], qa[VP_VO_SPT], qb[VP_VO_SPT];
#pragma unroll
        for (int k = 0; k < VP_VO_SPT; ++k) {
            const size_t p0 = ((size_t) (ig * VP_VO_SPT + k) * 32 + b) * N + a, p1 = p0 + No;
            la[k] = lcw[p0]; lb[k] = lcw[p1]; ha[k] = hcw[p0]; hb[k] = hcw[p1];
            if (!TENSOR) { qa[k] = qcw[p0]; qb[k] = qcw[p1]; }
        }
        loads_first();
        F f1[VP_VO_SPT];
#pragma unroll
        for (int k = 0; k < VP_VO_SPT; ++k) {
            const u32 i = ig * VP_VO_SPT + k;
            const F s0 = S0[i];
            if (TENSOR) {
                const F X = f_dot2cc(cA, la[k], cB, lb[k]), Y = f_dot2cc(cC, ha[k], cE, hb[k]);
                f1[k] = f_sub(f_dot2cc(qscal[i], X, f_neg(s0), cK), Y);
            } else {
                const F Ga = f_sub(f_sub(f_mul(la[k], qa[k]), f_mul(xn_m1, ha[k])), s0), Gb = f_sub(f_sub(f_mul(lb[k], qb[k]), f_mul(xn_m1, hb[k])), s0);
                const F D = f_sub(Ga, Gb), S = f_add(Ga, Gb);
                f1[k] = f_mul(hx, f_add(D, f_mul(xr, S)));
            }
            out1[((size_t) i * 32 + b) * No + a] = f1[k];
            if (v >= 2) x1[k][v - 2][lane] = f1[k];
        }
        __syncthreads();
        F f2[VP_VO_SPT];
        if (v < 2) {
#pragma unroll
            for (int k = 0; k < VP_VO_SPT; ++k) {
                const u32 i = ig * VP_VO_SPT + k;
                const F gg = x1[k][v][lane];
                f2[k] = f_add(f_half(f_add(f1[k], gg)), f_mul(c1, f_sub(f1[k], gg)));
                out2[((size_t) i * 32 + b) * N2 + a] = f2[k];
                if (v == 1) x2[k][lane] = f2[k];
            }
        }
        __syncthreads();
        if (v == 0) {
#pragma unroll
            for (int k = 0; k < VP_VO_SPT; ++k) {
                const u32 i = ig * VP_VO_SPT + k;
                const F gg = x2[k][lane];
                out3[((size_t) i * 32 + b) * E + al] = f_add(f_half(f_add(f2[k], gg)), f_mul(c2, f_sub(f2[k], gg)));
            }
        }
        // the next group's x1 is written by waves 2, 3 behind the second barrier (waves 0, 1 have read theirs before it), its x2 by wave 1 behind the
        // next first barrier (which wave 0 reaches after the reads above): no barrier of its own at the end of the loop
    }
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
    const bool twist = !a.inverse && coset;
    // The coset twist w_M^(j coset), j = j1 N2 + j2, splits into a factor of j1 alone (applied here, none for j1 = 0) and the factor
    // w_M^(j2 coset) that is common to the whole column: it commutes with the N1-point transform and joins the output twiddle below.
    // (Round 3 tried the other arrangement — the whole twist on the inputs, the output twiddles w_N^(j2 k1) from a compact order-N table
    // that stays in L2 instead of a merged index striding through the 32 MB order-M table: 14 % SLOWER per byte, 2213 -> 1908 GB/s at x1024;
    // the kernel is bound by its VALU instructions, one more multiplication per column costs more than the gathers do.)
    F x[N1];
#pragma unroll
    for (u32 j1 = 0; j1 < N1; ++j1) {
        F v = src[j1 * N2 + j2];
        if (twist && j1) v = f_mul(v, root_pow(a.RT, a.half_m, (u32) (((unsigned long long) j1 * N2 * coset) & (M - 1))));
        x[j1] = v;
    }
    const bool iota_plus = a.RT[a.half_m >> 1].im == 1;        // w_M^(M/4) = (0, +-1)
    const bool ip = a.inverse ? !iota_plus : iota_plus;
    // N1-point DFT, decimation in frequency in registers: natural in, bit-reversed out.  Twiddles 1 and iota cost nothing
    // (10 of the 32 butterflies of a 16-point transform multiply).
#pragma unroll
    for (int s = L1; s >= 1; --s) {
        const u32 half = 1u << (s - 1);
#pragma unroll
        for (u32 idx = 0; idx < N1 / 2; ++idx) {
            const u32 k = idx & (half - 1), i0 = ((idx >> (s - 1)) << s) | k, i1 = i0 + half;
            const u32 e1 = k * (N1 >> s);                      // w_N1^e1, e1 < N1/2 (compile-time after unrolling)
            const F u = x[i0], v = x[i1];
            x[i0] = f_add(u, v);
            const F d = f_sub(u, v);
            if (e1 == 0) x[i1] = d;
            else if (4 * e1 == N1) x[i1] = mul_iota(d, ip);
            else {
                u32 e = e1 * (M >> L1);
                if (a.inverse) e = M - e;
                x[i1] = f_mul(d, root_pow(a.RT, a.half_m, e));
            }
        }
    }
    F *dst = a.out + ((size_t) row * a.ncoset + coset) * N;
    const u32 ec = twist ? (u32) (((unsigned long long) j2 * coset) & (M - 1)) : 0u;      // the column's share of the twist
    // output twiddles w_N^(j2 k1) (x the column's twist): per-lane gathers from the order-M table, requested four at a time ahead of their products
    constexpr u32 G = N1 < 4 ? N1 : 4;
#pragma unroll
    for (u32 p0 = 0; p0 < N1; p0 += G) {
        u32 e[G], k1[G]; F w[G];
#pragma unroll
        for (u32 i = 0; i < G; ++i) {
            const u32 p = p0 + i;
            k1[i] = __brev(p) >> (32 - (L1 ? L1 : 1)) >> (L1 ? 0 : 1);         // bit reversal of p in L1 bits
            u32 ee = (u32) (((unsigned long long) j2 * k1[i] * wN) & (M - 1));  // w_N^(j2 k1)
            if (a.inverse) ee = ee ? M - ee : 0;
            e[i] = (ee + ec) & (M - 1);
            w[i] = root_raw(a.RT, a.half_m, e[i]);
        }
        loads_first();
#pragma unroll
        for (u32 i = 0; i < G; ++i) {
            const u32 p = p0 + i;
            dst[(size_t) k1[i] * N2 + j2] = (p == 0 && !twist) ? x[p] : f_mul(x[p], root_fin(w[i], a.half_m, e[i]));
        }
    }
}

// in: [rows][N1][N2] (k1-major), out: [rows][N] natural (k = k1 + N1*k2); tile of 64 k2 x N1 k1 through LDS
__global__ void __launch_bounds__(VP_BLOCK)
k_ntt_unsplit(const F *__restrict__ in, F *__restrict__ out, int ln, int l1, F scale, int do_scale) {
    __shared__ F tile[32][65];
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

#include "vp_kernels_ntt8.h"

// ---- openings (fri::request_init_value_with_merkle, fri.cpp:148-205; fri::request_step_commit, :229-287) ----
namespace vp {
// One leaf of a committed oracle: the 64 slice pairs + the (zero) mask pair, and the Merkle path from the leaf up.
// Coset-major codeword with Nc values per coset: leaf i = 32a + b holds (cw[s][b][a], cw[s][b][a + Nc/2]); for Nc == 1
// (last FRI level) leaf j < 16 holds (cw[s][j], cw[s][j + 16]).
__global__ void k_pc_open(const F *__restrict__ cw, u32 Nc, const Dig *__restrict__ tree, u32 n_leaves, u32 leaf,
                          F *__restrict__ vals /* 65*2 */, Dig *__restrict__ path /* depth+1 */, const F *__restrict__ mask /* the mask slice at this level (coset-major), or nullptr: zeros */) {
    const u32 t = threadIdx.x;
    if (t < 64) {
        F x, y;
        if (Nc >= 2) { const u32 a = leaf >> 5, b = leaf & 31; const F *row = cw + ((size_t) t * 32 + b) * Nc; x = row[a]; y = row[a + (Nc >> 1)]; }
        else { x = cw[(size_t) t * 32 + leaf]; y = cw[(size_t) t * 32 + leaf + 16]; }
        vals[2 * t] = x; vals[2 * t + 1] = y;
    } else if (t == 64) {
        F x = f_zero(), y = f_zero();
        if (mask) {
            if (Nc >= 2) { const u32 a = leaf >> 5, b = leaf & 31; x = mask[(size_t) b * Nc + a]; y = mask[(size_t) b * Nc + a + (Nc >> 1)]; }
            else { x = mask[leaf]; y = mask[leaf + 16]; }
        }
        vals[128] = x; vals[129] = y;
    }
    // path[k] = sibling at height k (k < depth), path[depth] = the leaf digest itself (the reference's com_hhash layout)
    u32 depth = 0;
    while ((1u << depth) < n_leaves) ++depth;
    if (t <= depth) {
        if (t == depth) path[t] = tree[n_leaves + leaf];
        else path[t] = tree[((n_leaves + leaf) >> t) ^ 1];
    }
}

// ---- the mask slice WITH CONTENT (round 6; lib/virgo/src/poly_commit.h:42,55-86,138-161,187-191, fri.cpp:96-124,366-386,403-411) --------------------------------
// The protocol's own calls pass one zero (src/prover.cpp:526, src/verifier.cpp:375-377) and take the kernels above; a caller that hands vp_commit_private_masked /
// vp_commit_public_masked a non-zero mask gets a 65th slice through every piece of the commitment.  The slice lives in arrays of its own (coset-major like the
// others: m[b * Nk + a] = value at position 32 a + b), so nothing of the 64-slice path changes shape.
// Leaf chains with the mask slice's pair as the last block (the compiler's form: the generated chains end in the all-zero block).
__global__ void __launch_bounds__(VP_BLOCK)
k_leaf_hash_cm(const F *__restrict__ cw, u32 N, int n_slices, const F *__restrict__ mask, Dig *__restrict__ leaves) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (N == 1) {                                     // last FRI level: one value per coset, leaf j pairs cosets j and j + 16
        if (t >= 16) return;
        Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
        for (int s = 0; s < n_slices; ++s) { const F x = cw[(size_t) s * 32 + t], y = cw[(size_t) s * 32 + t + 16]; h = hhash64(x.re, x.im, y.re, y.im, h); }
        const F x = mask[t], y = mask[t + 16];
        leaves[t] = hhash64(x.re, x.im, y.re, y.im, h);
        return;
    }
    const u32 halfN = N >> 1;
    if (t >= 32 * halfN) return;
    const u32 a = t % halfN, b = t / halfN;
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < n_slices; ++s) {
        const F *row = cw + ((size_t) s * 32 + b) * N;
        const F x = row[a], y = row[a + halfN];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    const F x = mask[(size_t) b * N + a], y = mask[(size_t) b * N + a + halfN];
    leaves[32 * a + b] = hhash64(x.re, x.im, y.re, y.im, h);
}
__global__ void __launch_bounds__(VP_BLOCK) k_fill_f(F *__restrict__ p, u32 n, F v) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// A polynomial of ms = c N coefficients (c = 2 .. 16: a mask longer than a slice's message) on the slice's domain: on coset b, x^N = zeta_b = w_32^b is a constant, so
// P(x) = sum_t zeta_b^t P_t(x) with P_t the t-th block of N coefficients — one N-coefficient polynomial per coset, twisted by w_M^(b j) so that the plain N-point
// transform of row b yields P at w_M^(32 k + b): out[b][j] = w_M^(b j) sum_t zeta_b^t coef[t N + j]
__global__ void __launch_bounds__(VP_BLOCK)
k_mask_combine(const F *__restrict__ coef, u32 N, u32 c, const F *__restrict__ RT, u32 half_m, F *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * N) return;
    const u32 b = t / N, j = t % N, M = 2 * half_m;
    const F zeta = root_pow(RT, half_m, (u32) (((unsigned long long) N * b) & (M - 1)));
    F acc = coef[(size_t) (c - 1) * N + j];
    for (int q = (int) c - 2; q >= 0; --q) acc = f_add(f_mul(acc, zeta), coef[(size_t) q * N + j]);
    out[t] = f_mul(acc, root_pow(RT, half_m, (u32) (((unsigned long long) b * j) & (M - 1))));
}
// l q of the mask slice at the 2 ms points the quotient needs (poly_commit.h:196-204): natural position j = id * (gap / 2), coset-major index (j & 31) N + (j >> 5)
__global__ void __launch_bounds__(VP_BLOCK)
k_mask_lq(const F *__restrict__ lm, const F *__restrict__ qm, u32 N, u32 half_gap, u32 count, F *__restrict__ out) {
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    const u32 j = id * half_gap;
    const size_t at = (size_t) (j & 31) * N + (j >> 5);
    out[id] = f_mul(lm[at], qm[at]);
}
// h_coef = the upper half of lq_coef, S0 = lq_coef[0] + h_coef[0], all_sum[64] = S0 ms (poly_commit.h:213-216,247)
__global__ void __launch_bounds__(VP_BLOCK)
k_mask_hcoef(const F *__restrict__ lq_coef, u32 ms, u32 n_pad, F ms_as_f, F *__restrict__ h_coef, F *__restrict__ S0, F *__restrict__ all_sum64) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) h_coef[i] = i < ms ? lq_coef[ms + i] : f_zero();
    if (i == 0) { const F s = f_add(lq_coef[0], lq_coef[ms]); *S0 = s; *all_sum64 = f_mul(s, ms_as_f); }
}
// Virtual oracle of the mask slice (poly_commit.h:225-245): (l q - (x^ms - 1) h - S0) ms x^-1 at x = w_M^(32 a + b), coset-major like its inputs
__global__ void __launch_bounds__(VP_BLOCK)
k_mask_vo(const F *__restrict__ lm, const F *__restrict__ qm, const F *__restrict__ hm, const F *__restrict__ S0, u32 N, const F *__restrict__ RT, u32 half_m,
          F ms_as_f, u32 ms, F *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 M = 2 * half_m;
    if (t >= M) return;
    const u32 a = t % N, b = t / N, j = 32 * a + b;
    const F xn_m1 = f_sub(root_pow(RT, half_m, (u32) (((unsigned long long) j * ms) & (M - 1))), f_one());
    const F g = f_sub(f_mul(lm[t], qm[t]), f_mul(xn_m1, hm[t]));
    const F inv_x = f_mul(ms_as_f, root_pow(RT, half_m, j ? M - j : 0));
    out[t] = f_mul(f_sub(g, *S0), inv_x);
}
// One FRI fold of ONE slice (k_fri_fold's arithmetic on a single coset-major codeword): fri.cpp:366-374
__global__ void __launch_bounds__(VP_BLOCK)
k_fri_fold_one(const F *__restrict__ in, F *__restrict__ out, u32 Nk, int k, const F *__restrict__ RT, u32 half_m, F r, F inv2) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 No = Nk >> 1;
    if (t >= 32 * No) return;
    const u32 a = t % No, b = t / No, M = 2 * half_m;
    const u32 e = (u32) ((((unsigned long long) (32 * a + b)) << k) & (M - 1));
    const F inv_mu = root_pow(RT, half_m, e ? M - e : 0);
    const F p = in[(size_t) b * Nk + a], q = in[(size_t) b * Nk + a + No];
    const F c = f_mul(inv2, f_mul(inv_mu, r));
    out[(size_t) b * No + a] = f_add(f_half(f_add(p, q)), f_mul(c, f_sub(p, q)));
}
}  // namespace vp
#undef f_mul
