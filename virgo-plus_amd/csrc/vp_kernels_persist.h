// Interactive path: the persistent round kernel.  Part of the single translation unit vpgpu.hip.
#pragma once
#include "vp_kernels_round.h"

// ===================================================================================================
// The drop-in entry point vp_round (prover::sumcheckUpdate*, src/prover.cpp:422-492) answers ONE verifier message per call,
// and the verifier only draws the next challenge after it has seen the answer (src/verifier.cpp:206-215): a sumcheck of n
// rounds is n dependent host <-> device round trips.  A kernel launch per round costs ~13 us before any work is done
// (tools/mailbox_probe.hip on MI355X: launch + pinned-memory reply poll); a resident kernel that waits on a mailbox in pinned
// host memory answers in 3.0 us plus its compute (same probe).  So a phase is served by ONE launch of k_phase:
//
//   request  (host -> device, pinned host memory): three words r.re | r.im | cmd   (cmd 1: round, 2: finalize, 3: quit)
//   reply    (device -> host, pinned host memory): seven words, the six limbs of the round polynomial | status; claims go to the
//            pinned claims array.  Every word carries the low three bits of the message's sequence number in its bits 61-63 (a
//            limb is < 2^61): a message is complete when all its words show the expected tag, so neither side needs a separate
//            "ready" word behind a fence or a wait for the stores to be acknowledged.
//   Every wait is bounded: the kernel leaves after VP_PH_TIMEOUT_TICKS (100 MHz s_memrealtime) without a message (status 2,
//   the next call reports VP_EHIP); the host gives up after 15 s.
//
// One regime: SOLO rounds — every live table of the phase fits one CU's LDS (<= VP_PH_PMAX pairs): one workgroup; tables in LDS at full
// power-of-two length, folded in place; wave-uniform roles split a pair's nine multiplications over three waves (fold V / mult / add, then
// one product each), waves that can never have work again exit.  (Rounds 2-3 also had a distributed regime — G workgroups folding slices of a
// large single table, challenge relayed through device memory, partial sums through agent-scope stores and an arrival counter; measured equal
// to one launch per round, 18-20 us, and removed in round 4 together with its cross-workgroup spin waits.)
// Semantics (retiring single-entry tables into add_term, src/prover.cpp:445,462-467; claims, :494-521) are those of k_round_final /
// k_finalize, which stay the path for the rounds of multi-table phases that are too large for one CU.
// ===================================================================================================
namespace vp {

#define VP_PH_THREADS 768                       // 12 waves = 4 groups x 3 roles
#define VP_PH_PMAX 1024                         // pairs of the first solo round (LDS: 3 x 2048 entries = 96 KiB)
#define VP_PH_SLOTS 256                         // pair slots per pass (4 groups x 64 lanes)
#define VP_PH_MAXIT (VP_PH_PMAX / VP_PH_SLOTS)
#define VP_PH_TIMEOUT_TICKS 1000000000ull       // 10 s of s_memrealtime (100 MHz)

struct TailMail { unsigned long long w[3], pad[5]; };                     // r.re | r.im | cmd, each | (seq & 7) << 61
struct TailReply { unsigned long long w[7]; unsigned long long dead;     // poly limbs a.re a.im b.re b.im c.re c.im | status, tagged; dead = 1: the kernel has left on its own (time-out)
                   unsigned long long stamps[16]; };                    // stamps: -DVP_TAIL_STAMPS diagnostic build only
#define VP_TAG(seq) (((unsigned long long) (seq) & 7ull) << 61)
#define VP_UNTAG(x) ((x) & 0x1fffffffffffffffull)
#ifdef VP_TAIL_STAMPS
#define TSTAMP(i) do { if (tid == 0) __hip_atomic_store(&a.rep->stamps[i], (unsigned long long) __builtin_amdgcn_s_memtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

// per-table data of a launch: written by the host into pinned memory, read once by the kernel (kernel arguments indexed per lane
// would be copied to scratch)
struct TailAux {
    TabDesc t[VP_MAX_TAB];        // first SOLO round as do_round builds it (inputs in global memory; pair_start over FULL folded lengths)
    int bl[VP_MAX_TAB];           // log2 of the table's length at round 1 (finalize)
    u32 len_out0[VP_MAX_TAB];     // length of table j after the first solo round's fold (0: table already retired)
    u32 loff[VP_MAX_TAB];         // LDS offset (entries) of table j
};
struct PTailArgs {
    const TailAux *aux;
    const F *inV, *inM, *inA; F rv; int fold; u32 total_pairs;      // first solo round of a multi-table phase
    int k0, R, n_tab, has_a;
    u32 cap;                      // LDS entries per table family
    F *add_term, *scalarV, *claims_dev, *Vu, *poly_dev;
    TailMail *req; TailReply *rep; F *claims_host;
    unsigned long long seq0;      // the reply to the first round of the launch carries seq0, message i after it seq0 + i
    // Suspend / resume (round 3).  A resident kernel that is told to leave in the middle of a phase (another context of the process needs
    // the device for a synchronising call) or that waited longer than `timeout_ticks` for the verifier saves the phase — the LDS tables of
    // the current level, their lengths, the round counter, add_term and the polynomial answered last — into `save` ([3][cap] entries, then
    // a header) and leaves with status 5; the next vp_round / vp_finalize relaunches it with resume = 1 and the sumcheck continues where it
    // was.
    F *save; int resume;
    unsigned long long timeout_ticks;                // s_memrealtime ticks (100 MHz) without a message before the kernel leaves
};

// ---- small helpers ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// wave totals -> red[w][3]; after the caller's barrier wave 0 adds the first `nw` rows; totals valid in lane 63 of wave 0
__device__ __forceinline__ void tail_wave_partials(F (&acc)[3], F *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = wave_sum63(acc[i]);
    if (lane == 63) { red[w * 3] = acc[0]; red[w * 3 + 1] = acc[1]; red[w * 3 + 2] = acc[2]; }
}
__device__ __forceinline__ void tail_wave0_total(const F *red, int nw, F (&tot)[3]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 3; ++i) tot[i] = wave_sum63(lane < nw ? red[lane * 3 + i] : f_zero());
}
// thread 0: wait for mailbox message `expect`; returns cmd (> 0) with r, or -1 after the timeout
__device__ __forceinline__ int tail_poll(const PTailArgs &a, unsigned long long expect, F &r) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), tag = VP_TAG(expect);
    for (;;) {
        // relaxed system-scope loads read the fine-grained host memory past the caches; no acquire (it would invalidate them on every poll).
        // All three words are requested in ONE go (independent loads, one wait): every such load is a read over PCIe (~2 us); taken one
        // after the other — w0, and w1 / w2 only once w0 showed the tag, as rounds 2 did — a message cost three of them.
        const unsigned long long w0 = __hip_atomic_load(&a.req->w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long w1 = __hip_atomic_load(&a.req->w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long w2 = __hip_atomic_load(&a.req->w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((w0 & (7ull << 61)) == tag && (w1 & (7ull << 61)) == tag && (w2 & (7ull << 61)) == tag) { r = f_make(VP_UNTAG(w0), VP_UNTAG(w1)); return (int) VP_UNTAG(w2); }
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) return -1;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void tail_reply_words(const PTailArgs &a, unsigned long long seq, const F &pa, const F &pb, const F &pc, unsigned long long status) {
    const unsigned long long tag = VP_TAG(seq);
    unsigned long long *q = a.rep->w;
    __hip_atomic_store(q, pa.re | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(q + 1, pa.im | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 2, pb.re | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(q + 3, pb.im | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 4, pc.re | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(q + 5, pc.im | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 6, status | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void tail_reply(const PTailArgs &a, unsigned long long seq, const F &pa, const F &pb, const F &pc) {
    a.poly_dev[0] = pa; a.poly_dev[1] = pb; a.poly_dev[2] = pc;
    tail_reply_words(a, seq, pa, pb, pc, 0);
}
// leaving: status 0 ok (finalize done), 1 quit acknowledged, 2 timed out, 3 protocol error, 4 a workgroup did not arrive
__device__ __forceinline__ void tail_leave(const PTailArgs &a, unsigned long long seq, int cmd, unsigned long long status) {
    __threadfence_system();                                   // claims (finalize) and add_term are out before the reply
    // dead: the kernel has left on its own.  1 = gave up (not resumable), 2 = timed out with the phase saved (the host relaunches it)
    if (cmd == -1 || status == 4) __hip_atomic_store(&a.rep->dead, (cmd == -1 && status == 5) ? 2ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    tail_reply_words(a, seq, f_zero(), f_zero(), f_zero(), status);
}

// ---- the launch ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VP_PH_THREADS) k_phase(PTailArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    F *L = reinterpret_cast<F *>(smem_raw);                       // solo regime: [3][cap]
    __shared__ F red[12 * 3];
    __shared__ F s_r;
    __shared__ int s_cmd;
    __shared__ u32 s_len[VP_MAX_TAB];                              // solo: current length of table j in LDS
    __shared__ u32 s_loff[VP_MAX_TAB]; __shared__ int s_bl[VP_MAX_TAB];
    __shared__ TabDesc s_t[VP_MAX_TAB];                            // first solo round's table descriptors
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool boss = tid == 63;                                   // lane 63 of wave 0: holds add_term and the totals
    F at = f_zero();
    if (boss && blockIdx.x == 0) at = *a.add_term;
    if (blockIdx.x == 0 && tid < a.n_tab) { s_t[tid] = a.aux->t[tid]; s_loff[tid] = a.aux->loff[tid]; s_bl[tid] = a.aux->bl[tid]; s_len[tid] = a.aux->len_out0[tid]; }
    __syncthreads();
    unsigned long long expect = a.seq0;                            // sequence number of the message answered last
    int k = a.k0;                                                  // round answered last (after the first round below)

    // ======================= solo regime (workgroup 0) =====================================================================
    const u32 cap = a.cap;
    F *LV = L, *LM = L + cap, *LA = L + 2 * (size_t) cap;
    const int role = __builtin_amdgcn_readfirstlane(w % 3);
    const u32 pslot = (u32) ((w / 3) * 64 + lane);
    F *const hdr = a.save + 3 * (size_t) cap;                        // saved phase: hdr[0].re = k, hdr[1] = add_term, hdr[2..4] = last polynomial, hdr[5 + j].re = s_len[j]
    if (a.resume) {
        for (u32 i = tid; i < 3 * cap; i += blockDim.x) L[i] = a.save[i];
        if (tid < a.n_tab) s_len[tid] = (u32) hdr[5 + tid].re;
        if (boss) at = hdr[1];
        k = (int) hdr[0].re;
        expect = a.seq0 - 1;                                        // the loop below waits for message seq0
    }
    // ---- first solo round: sources in global memory, folded tables into LDS (plain one-thread-per-pair form) ----
    if (!a.resume) {
        F r;
        int n_tab, fold;
        u32 total_pairs;
        const F *inV, *inM, *inA;
        r = a.rv; n_tab = a.n_tab; fold = a.fold; total_pairs = a.total_pairs;
        inV = a.inV; inM = a.inM; inA = a.inA;
        auto ldz = [&](const F *p, u32 i, u32 valid) -> F { return i < valid ? p[i] : f_zero(); };
        F acc[3] = {f_zero(), f_zero(), f_zero()};
        for (u32 q = tid; q < total_pairs; q += blockDim.x) {
            int j = 0;
            while (j + 1 < n_tab && q >= s_t[j + 1].pair_start) ++j;
            const TabDesc td = s_t[j];
            const u32 p = q - td.pair_start;
            if (td.len_in == 0) continue;
            F v0, v1, m0, m1, a0 = f_zero(), a1 = f_zero();
            const u32 vi = td.off + td.valid_in;
            if (fold) {
                const u32 i0 = td.off + 4 * p;
                v0 = f_lerp(ldz(inV, i0, vi), ldz(inV, i0 + 1, vi), r); v1 = f_lerp(ldz(inV, i0 + 2, vi), ldz(inV, i0 + 3, vi), r);
                m0 = f_lerp(ldz(inM, i0, vi), ldz(inM, i0 + 1, vi), r); m1 = f_lerp(ldz(inM, i0 + 2, vi), ldz(inM, i0 + 3, vi), r);
                if (a.has_a) { a0 = f_lerp(ldz(inA, i0, vi), ldz(inA, i0 + 1, vi), r); a1 = f_lerp(ldz(inA, i0 + 2, vi), ldz(inA, i0 + 3, vi), r); }
            } else {
                const u32 i0 = td.off + 2 * p;
                v0 = ldz(inV, i0, vi); v1 = ldz(inV, i0 + 1, vi);
                m0 = ldz(inM, i0, vi); m1 = ldz(inM, i0 + 1, vi);
                if (a.has_a) { a0 = ldz(inA, i0, vi); a1 = ldz(inA, i0 + 1, vi); }
            }
            const u32 o = s_loff[j] + 2 * p;
            LV[o] = v0; LV[o + 1] = v1; LM[o] = m0; LM[o + 1] = m1; LA[o] = a0; LA[o + 1] = a1;
            const F dm = f_sub(m1, m0), dv = f_sub(v1, v0);
            const F qa = f_mul(dm, dv), qc = f_mul(m0, v0), qe = f_mul(m1, v1);
            acc[0] = f_add(acc[0], qa);
            acc[1] = f_add(acc[1], f_add(f_sub(f_sub(qe, qa), qc), f_sub(a1, a0)));
            acc[2] = f_add(acc[2], f_add(qc, a0));
        }
        tail_wave_partials(acc, red);
        __syncthreads();
        F tot[3];
        if (w == 0) tail_wave0_total(red, (int) (blockDim.x >> 6), tot);
        if (boss) {
            if (fold && !f_is_zero(at)) at = f_mul(at, f_sub(f_one(), r));
            for (int j = 0; j < n_tab; ++j) {                   // tables that reach length one in this round (k_round_final's loop)
                    const TabDesc td = s_t[j];
                    const u32 len_out = fold ? (td.len_in >> 1) : td.len_in;
                    if (len_out != 1) continue;
                    F v, m, ad = f_zero();
                    if (fold) {
                        const u32 vi = td.off + td.valid_in;
                        v = f_lerp(ld_or_zero(inV, td.off, vi), ld_or_zero(inV, td.off + 1, vi), r);
                        m = f_lerp(ld_or_zero(inM, td.off, vi), ld_or_zero(inM, td.off + 1, vi), r);
                        if (a.has_a) ad = f_lerp(ld_or_zero(inA, td.off, vi), ld_or_zero(inA, td.off + 1, vi), r);
                    } else { v = inV[td.off]; m = inM[td.off]; if (a.has_a) ad = inA[td.off]; }
                    a.scalarV[j] = v;
                    at = f_add(at, f_add(f_mul(v, m), ad));
                }
            tail_reply(a, expect, tot[0], f_sub(tot[1], at), f_add(tot[2], at));
        }
    }
    // ---- mailbox loop: the remaining rounds, then finalize --------------------------------------------------------------
    // Round 3: THE ANSWER FIRST.  A round's polynomial depends on the challenge only through the fold of the tables, and that dependence is
    // quadratic: with the pair (x0, x1) of the next level made of the quad (e0, e1, e2, e3) of this one, x0(r) = e0 + r (e1 - e0) and
    // x1(r) = e2 + r (e3 - e2), so   sum dm(r) dv(r) = P0 + P1 r + P2 r^2   and   sum m0(r) v0(r) + a0(r) = C0 + C1 r + C2 r^2.
    // The six sums are formed WHILE the verifier is still looking at the previous answer (three multiplications per quad each, Karatsuba on
    // the two linear factors); when the challenge arrives the reply is two Horner evaluations on one lane plus the identity
    // b = S_prev(r) - a - 2c (src/verifier.cpp:208: the check the verifier makes is the equation b is solved from), and the fold of the
    // tables — and the next round's six sums — happen behind it, in the shadow of the host's round trip.  Critical path per round: the
    // mailbox latency plus ~8 dependent multiplications, instead of fold -> barrier -> products -> reduction -> reply.
    // MEASURED (in-kernel timestamps, -DVP_TAIL_STAMPS, MI355X): six sums 9 200 shader clocks, then the kernel still WAITS 3 500 for the
    // challenge, reply chain 3 400, fold 2 100 — the device's work between two challenges (11 300 clocks) is shorter than the host's round
    // trip (reply over PCIe, the verifier's round, request back over PCIe: ~14 800 clocks = 7 us), so a resident round now costs that round
    // trip plus the reply chain: 10.2 -> 9.2 us per vp_round at x64 (with the three mailbox words polled in one go).
    F pp0 = f_zero(), pp1 = f_zero(), pp2 = f_zero();              // boss: the polynomial answered last
    if (boss) {
        if (a.resume) { pp0 = hdr[2]; pp1 = hdr[3]; pp2 = hdr[4]; }
        else { pp0 = a.poly_dev[0]; pp1 = a.poly_dev[1]; pp2 = a.poly_dev[2]; }          // written by tail_reply a few lines up (same lane)
    }
    const int role2 = __builtin_amdgcn_readfirstlane(w & 1);
    const u32 qslot = (u32) ((w >> 1) * 64 + lane);
    for (;;) {
        ++expect;
        __syncthreads();                                           // the LDS tables of this level and s_len are complete
        // (A) the coming round as a function of its challenge
        TSTAMP(4);
        F PA[3] = {f_zero(), f_zero(), f_zero()}, PC[3] = {f_zero(), f_zero(), f_zero()};
        {
            F acc[3] = {f_zero(), f_zero(), f_zero()};
            u32 nq = 0;
            for (int j = 0; j < a.n_tab; ++j) nq += s_len[j] >> 2;
            if (k < a.R)
                for (u32 q = qslot; q < nq; q += (VP_PH_THREADS / 128) * 64) {
                    u32 base = 0, i0 = 0;
                    for (int j = 0; j < a.n_tab; ++j) { const u32 np = s_len[j] >> 2; if (q >= base && q < base + np) i0 = s_loff[j] + 4 * (q - base); base += np; }
                    if (role2 == 0) {
                        const F m0 = LM[i0], m1 = LM[i0 + 1], m2 = LM[i0 + 2], m3 = LM[i0 + 3], v0 = LV[i0], v1 = LV[i0 + 1], v2 = LV[i0 + 2], v3 = LV[i0 + 3];
                        const F dm0 = f_sub(m2, m0), dv0 = f_sub(v2, v0), sm = f_sub(m3, m1), sv = f_sub(v3, v1);
                        const F p0 = f_mul(dm0, dv0), p2 = f_mul(f_sub(sm, dm0), f_sub(sv, dv0)), p1 = f_sub(f_sub(f_mul(sm, sv), p0), p2);
                        acc[0] = f_add(acc[0], p0); acc[1] = f_add(acc[1], p1); acc[2] = f_add(acc[2], p2);
                    } else {
                        const F m0 = LM[i0], m1 = LM[i0 + 1], v0 = LV[i0], v1 = LV[i0 + 1];
                        const F c0 = f_mul(m0, v0), c2 = f_mul(f_sub(m1, m0), f_sub(v1, v0)), c1 = f_sub(f_sub(f_mul(m1, v1), c0), c2);
                        acc[0] = f_add(acc[0], c0); acc[1] = f_add(acc[1], c1); acc[2] = f_add(acc[2], c2);
                        if (a.has_a) { const F a0 = LA[i0], a1 = LA[i0 + 1]; acc[0] = f_add(acc[0], a0); acc[1] = f_add(acc[1], f_sub(a1, a0)); }
                    }
                }
            tail_wave_partials(acc, red);
            __syncthreads();
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    PA[i] = wave_sum63((lane < VP_PH_THREADS / 64 && !(lane & 1)) ? red[lane * 3 + i] : f_zero());
                    PC[i] = wave_sum63((lane < VP_PH_THREADS / 64 && (lane & 1)) ? red[lane * 3 + i] : f_zero());
                }
            }
        }
        // (B) the challenge
        TSTAMP(5);
        if (tid == 0) { F rr = f_zero(); s_cmd = tail_poll(a, expect, rr); s_r = rr; }
        TSTAMP(0);
        __syncthreads();
        const int cmd = s_cmd;
        const F r = s_r;
        if (cmd == 1 && k < a.R) {
            ++k;
            // (C) the answer: one lane, from the sums formed before the challenge came
            if (boss) {
                if (!f_is_zero(at)) at = f_mul(at, f_sub(f_one(), r));
                for (int j = 0; j < a.n_tab; ++j) {                 // tables that fold to a single entry now retire into add_term
                    if (s_len[j] != 2) continue;
                    const u32 o = s_loff[j];
                    const F v = f_lerp(LV[o], LV[o + 1], r), m = f_lerp(LM[o], LM[o + 1], r);
                    const F ad = a.has_a ? f_lerp(LA[o], LA[o + 1], r) : f_zero();
                    a.scalarV[j] = v;
                    at = f_add(at, f_add(f_mul(v, m), ad));
                }
                const F qa = f_add(PA[0], f_mul(r, f_add(PA[1], f_mul(r, PA[2]))));
                const F qc = f_add(f_add(PC[0], f_mul(r, f_add(PC[1], f_mul(r, PC[2])))), at);
                const F sp = f_add(f_mul(f_add(f_mul(pp0, r), pp1), r), pp2);            // S_prev(r) = the claim this round must add up to
                const F qb = f_sub(f_sub(sp, qa), f_dbl(qc));
                tail_reply(a, expect, qa, qb, qc);
                pp0 = qa; pp1 = qb; pp2 = qc;
            }
            TSTAMP(1);
            // (D) behind the answer: fold every table by r (registers, barrier, write back)
            u32 tot_out = 0;
            for (int j = 0; j < a.n_tab; ++j) tot_out += s_len[j] >> 1;
            constexpr int FIT = (2 * VP_PH_PMAX + VP_PH_THREADS - 1) / VP_PH_THREADS;     // entries of one family per thread
            F fv[3][FIT]; u32 fo[FIT];
#pragma unroll
            for (int it = 0; it < FIT; ++it) {
                const u32 idx = (u32) it * VP_PH_THREADS + (u32) tid;
                fo[it] = 0xffffffffu;
                if (idx >= tot_out) continue;
                // entry (idx - base) of table j's next level = lerp of entries 2 (idx - base), + 1 of this level
                u32 base = 0, o = 0, src = 0;
                for (int j = 0; j < a.n_tab; ++j) {
                    const u32 no = s_len[j] >> 1;
                    if (idx >= base && idx < base + no) { o = s_loff[j] + (idx - base); src = s_loff[j] + 2 * (idx - base); }
                    base += no;
                }
                fo[it] = o;
                fv[0][it] = f_lerp(LV[src], LV[src + 1], r);
                fv[1][it] = f_lerp(LM[src], LM[src + 1], r);
                fv[2][it] = a.has_a ? f_lerp(LA[src], LA[src + 1], r) : f_zero();
            }
            TSTAMP(2);
            __syncthreads();                                       // every source entry is in registers
#pragma unroll
            for (int it = 0; it < FIT; ++it) if (fo[it] != 0xffffffffu) { LV[fo[it]] = fv[0][it]; LM[fo[it]] = fv[1][it]; LA[fo[it]] = fv[2][it]; }
            __syncthreads();                                       // nobody still reads s_len of this level
            if (tid < a.n_tab) s_len[tid] >>= 1;
            TSTAMP(3);
            continue;
        }
        // finalize (cmd 2), quit (3), timeout (-1) or a protocol error: leave
        if (cmd == 2 && tid < a.n_tab) {
            F c;
            if (s_bl[tid] == a.R) { const u32 o = s_loff[tid]; c = f_lerp(LV[o], LV[o + 1], r); }       // the table as long as the sumcheck: its last two entries
            else c = a.scalarV[tid];
            a.claims_dev[tid] = c;
            a.claims_host[tid] = c;
            if (a.Vu && tid == 0) *a.Vu = c;
        }
        // quit (another entry point needs the device) or time-out in the middle of the phase: save it, so that the next vp_round /
        // vp_finalize can continue (status 5).  The tables in LDS are those of the current level, k rounds are answered.
        const bool suspend = (cmd == 3 || cmd == -1) && a.save != nullptr;
        if (suspend) {
            for (u32 i = tid; i < 3 * cap; i += blockDim.x) a.save[i] = L[i];
            if (tid < a.n_tab) hdr[5 + tid] = f_make(s_len[tid], 0);
            if (boss) { hdr[0] = f_make((u64) k, 0); hdr[1] = at; hdr[2] = pp0; hdr[3] = pp1; hdr[4] = pp2; }
        }
        __threadfence_system();
        __syncthreads();
        if (boss) { *a.add_term = at; tail_leave(a, expect, cmd, cmd == 2 ? 0 : suspend ? 5 : cmd == 3 ? 1 : cmd == -1 ? 2 : 3); }
        return;
    }
}

}  // namespace vp
