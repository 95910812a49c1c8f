// Development probe: launch k_phase (solo regime) on a tiny synthetic table and exchange a few mailbox messages; prints HIP errors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include "../virgo-plus_amd/csrc/vp_kernels.h"
using namespace vp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
int main() {
    const u32 len0 = 256;
    F *V, *M, *A, *small;
    CK(hipMalloc(&V, len0 * 16)); CK(hipMalloc(&M, len0 * 16)); CK(hipMalloc(&A, len0 * 16)); CK(hipMalloc(&small, 4096));
    CK(hipMemset(V, 1, len0 * 16)); CK(hipMemset(M, 2, len0 * 16)); CK(hipMemset(A, 0, len0 * 16)); CK(hipMemset(small, 0, 4096));
    TailMail *req; TailReply *rep; TailAux *aux; F *claims;
    CK(hipHostMalloc(&req, sizeof(TailMail))); CK(hipHostMalloc(&rep, sizeof(TailReply))); CK(hipHostMalloc(&aux, sizeof(TailAux))); CK(hipHostMalloc(&claims, 64 * 16));
    memset(req, 0, sizeof *req); memset(rep, 0, sizeof *rep); memset(aux, 0, sizeof *aux);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * VP_PH_PMAX * (int) sizeof(F)));
    PTailArgs a{};
    a.aux = aux; a.inV = V; a.inM = M; a.inA = A; a.rv = f_zero(); a.fold = 0; a.total_pairs = len0 / 2;
    a.k0 = 1; a.R = 8; a.n_tab = 1; a.has_a = 1; a.cap = len0;
    aux->t[0].off = 0; aux->t[0].len_in = len0; aux->t[0].valid_in = len0; aux->t[0].pair_start = 0;
    aux->bl[0] = 8; aux->len_out0[0] = len0; aux->loff[0] = 0;
    a.add_term = small; a.scalarV = small + 8; a.claims_dev = small + 80; a.Vu = nullptr; a.poly_dev = small + 4;
    a.req = req; a.rep = rep; a.claims_host = claims; a.seq0 = 1;
    printf("sizeof(PTailArgs) = %zu\n", sizeof(PTailArgs));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_phase, dim3(1), dim3(VP_PH_THREADS), (size_t) 3 * a.cap * sizeof(F), st, a);
    CK(hipGetLastError());
    auto wait = [&](unsigned long long want) { auto t0 = std::chrono::steady_clock::now(); for (;;) { bool all = true; for (int q = 0; q < 7; ++q) all &= (__atomic_load_n(&rep->w[q], __ATOMIC_ACQUIRE) & (7ull << 61)) == VP_TAG(want); if (all) return true; if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { printf("timeout waiting for %llu\n", want); return false; } } };
    if (wait(1)) printf("round 1 ok: a=(%llu,%llu)\n", VP_UNTAG(rep->w[0]), VP_UNTAG(rep->w[1]));
    unsigned long long seq = 1;
    for (int k = 2; k <= 8; ++k) {
        auto t0 = std::chrono::steady_clock::now();
        ++seq; req->w[1] = 777 | VP_TAG(seq); req->w[2] = 1 | VP_TAG(seq);
        __atomic_store_n(&req->w[0], (unsigned long long) (12345 + k) | VP_TAG(seq), __ATOMIC_RELEASE);
        if (!wait(seq)) break;
        printf("round %d: %.2f us\n", k, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    ++seq; req->w[1] = 777 | VP_TAG(seq); req->w[2] = 2 | VP_TAG(seq); __atomic_store_n(&req->w[0], 5ull | VP_TAG(seq), __ATOMIC_RELEASE);
    if (wait(seq)) printf("finalize ok status %llu claim (%llu,%llu)\n", VP_UNTAG(rep->w[6]), (unsigned long long) claims[0].re, (unsigned long long) claims[0].im);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    return 0;
}
