// Development probe (not part of the product): round-trip latency of a host <-> persistent-kernel mailbox on MI355X, for the
// interactive prover path (one verifier message per sumcheck round).  A one-workgroup kernel waits for seq == i in the request
// word, does `work` dependent multiply steps, and writes i to the reply word; the host writes i, spins on the reply.  Every spin on
// either side is bounded (the kernel gives up after ~0.2 s without a message, the host after 2 s).
//   request word in: (a) pinned host memory, (b) device memory written by the host through the BAR (fine-grained allocation),
//                    (c) plain hipMalloc memory dereferenced by the host (only tried when `--raw` is given: may fault)
//   reply word in pinned host memory (the host polls its own RAM).
// Build: hipcc --offload-arch=gfx950 -O3 -o mailbox_probe mailbox_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_mailbox(volatile u64 *req, volatile u64 *rep, int n, int work, int fan) {
    __shared__ u64 sh;
    for (int i = 1; i <= n; ++i) {
        if (threadIdx.x == 0) {
            const u64 t0 = __builtin_amdgcn_s_memrealtime();              // 100 MHz
            u64 v;
            for (;;) {
                v = __hip_atomic_load((u64 *) req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v >= (u64) i) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { v = ~0ull; break; }     // 0.2 s: give up
                __builtin_amdgcn_s_sleep(1);
            }
            sh = v;
        }
        __syncthreads();
        if (sh == ~0ull) return;
        u64 x = sh + threadIdx.x;
        for (int k = 0; k < work; ++k) x = x * 6364136223846793005ull + 1442695040888963407ull;
        if (fan) { __shared__ u64 red[256]; red[threadIdx.x] = x; __syncthreads(); if (threadIdx.x == 0) { u64 s = 0; for (int q = 0; q < 256; ++q) s += red[q]; x = s; } }
        if (threadIdx.x == 0) {
            rep[1] = x;
            __hip_atomic_store((u64 *) rep, (u64) i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
}

static double run(volatile u64 *req_host_view, u64 *req_dev, volatile u64 *rep_host, u64 *rep_dev, int n, int work, int fan, const char *name) {
    *req_host_view = 0; rep_host[0] = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(256), 0, st, req_dev, rep_dev, n, work, fan);
    // let the kernel start
    auto t_start = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t_start < std::chrono::milliseconds(5)) {}
    auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (int i = 1; i <= n && ok; ++i) {
        __atomic_store_n((u64 *) req_host_view, (u64) i, __ATOMIC_RELEASE);
        auto ts = std::chrono::steady_clock::now();
        while (__atomic_load_n((u64 *) rep_host, __ATOMIC_ACQUIRE) != (u64) i) {
            if (std::chrono::steady_clock::now() - ts > std::chrono::seconds(2)) { ok = false; break; }
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    hipStreamDestroy(st);
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    printf("%-44s work %4d fan %d : %s  %.2f us per round trip\n", name, work, fan, ok ? "ok" : "TIMEOUT", us);
    return us;
}

int main(int argc, char **argv) {
    const bool raw = argc > 1 && !strcmp(argv[1], "--raw");
    const int n = 5000;
    u64 *rep_host; hipHostMalloc((void **) &rep_host, 64, hipHostMallocDefault);
    u64 *rep_dev = rep_host;
    // (a) request in pinned host memory
    u64 *req_a; hipHostMalloc((void **) &req_a, 64, hipHostMallocDefault);
    for (int work : {0, 200, 1000}) run(req_a, req_a, rep_host, rep_dev, n, work, 0, "request in pinned host memory");
    run(req_a, req_a, rep_host, rep_dev, n, 200, 1, "request in pinned host memory");
    // (b) request in fine-grained device memory, host writes through the BAR
    u64 *req_b = nullptr;
    if (hipExtMallocWithFlags((void **) &req_b, 64, hipDeviceMallocFinegrained) == hipSuccess && req_b) {
        hipPointerAttribute_t at; memset(&at, 0, sizeof at);
        hipPointerGetAttributes(&at, req_b);
        printf("fine-grained device allocation: dev ptr %p host ptr %p type %d\n", at.devicePointer, at.hostPointer, (int) at.type);
        if (raw) for (int work : {0, 200, 1000}) run(req_b, req_b, rep_host, rep_dev, n, work, 0, "request in fine-grained DEVICE memory (BAR)");
    } else printf("hipExtMallocWithFlags(finegrained) failed\n");
    // (c) launch + poll baseline: one tiny kernel per message
    {
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        *req_a = 0; rep_host[0] = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= 2000; ++i) {
            *req_a = i;
            hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(256), 0, st, req_a, rep_dev, 1, 200, 0);
            // the kernel waits for seq >= 1 (already there) and replies 1; reset the reply word each time
            while (__atomic_load_n(rep_host, __ATOMIC_ACQUIRE) != 1ull) {}
            rep_host[0] = 0;
        }
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        printf("%-44s : %.2f us per message\n", "one launch per message (pinned reply poll)", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    }
    return 0;
}
