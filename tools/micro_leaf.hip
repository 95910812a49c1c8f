// The product's two leaf-hash kernels (k_leaf_hash: generated fixed-register chains in phase; k_leaf_hash_c: the compiler's Keccak-f) on a synthetic
// codeword of the x1024 geometry (64 slices x 32 cosets x 2^17 values: 2^21 leaves), alternating.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/_build/micro_leaf tools/micro_leaf.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../virgo-plus_amd/csrc/vp_check.h"
#include "../virgo-plus_amd/csrc/vp_kernels_pc.h"
using namespace vp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(F *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        u64 s = i * 0x9E3779B97F4A7C15ull + 1; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        p[i].re = s & ((1ull << 61) - 1); s *= 0x94D049BB133111EBull; s ^= s >> 31; p[i].im = s & ((1ull << 61) - 1);
    }
}
int main(int argc, char **argv) {
    const u32 ln = argc > 1 ? atoi(argv[1]) : 17, N = 1u << ln;
    const size_t n_el = (size_t) 64 * 32 * N; const u32 n_leaves = 16 * N;
    F *cw; Dig *o1, *o2;
    CK(hipMalloc(&cw, n_el * sizeof(F))); CK(hipMalloc(&o1, (size_t) n_leaves * sizeof(Dig))); CK(hipMalloc(&o2, (size_t) n_leaves * sizeof(Dig)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, cw, n_el); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) {
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_leaf_hash_c, dim3((n_leaves + 255) / 256), dim3(256), 0, 0, cw, N, 64, o1);
            else hipLaunchKernelGGL(k_leaf_hash, dim3((n_leaves + 1023) / 1024), dim3(1024), 0, 0, cw, N, 64, o2);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best[v] = ms < best[v] ? ms : best[v];
            printf("rep %d %-28s %.3f ms  %.3e Keccak-f/s\n", rep, v ? "k_leaf_hash (asm, in phase)" : "k_leaf_hash_c (compiler)", ms, (double) n_leaves * 65 / (ms * 1e-3));
        }
    }
    std::vector<Dig> a(n_leaves), b(n_leaves);
    CK(hipMemcpy(a.data(), o1, (size_t) n_leaves * sizeof(Dig), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, (size_t) n_leaves * sizeof(Dig), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (u32 i = 0; i < n_leaves; ++i) for (int k = 0; k < 4; ++k) if (a[i].w[k] != b[i].w[k]) ++bad;
    printf("2^%d leaves: differing digest words %zu; best %.3f ms (compiler) %.3f ms (asm): %+.1f %%\n", ln + 4, bad, best[0], best[1], 100.0 * (best[1] / best[0] - 1));
    return bad != 0;
}
