// The leaf-hash chain (65 chained SHA3-256 of 64-byte blocks per leaf) with the compiler's Keccak-f and with the generated fixed-register block
// (tools/gen_keccak_asm.py): same digests, time per launch.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/micro_keccak tools/micro_keccak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../virgo-plus_amd/csrc/vp_check.h"
#include "../virgo-plus_amd/csrc/vp_kernels_pc.h"
#include "../virgo-plus_amd/csrc/vp_keccak_asm.h"
using namespace vp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_chain_c(const F *__restrict__ cw, u32 n, int n_slices, Dig *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Dig h; h.w[0] = h.w[1] = h.w[2] = h.w[3] = 0;
    for (int s = 0; s < n_slices; ++s) {
        const F x = cw[(size_t) s * 2 * n + t], y = cw[(size_t) s * 2 * n + n + t];
        h = hhash64(x.re, x.im, y.re, y.im, h);
    }
    h = hhash64(0, 0, 0, 0, h);
    out[t] = h;
}
__global__ void __launch_bounds__(256) k_chain_asm(const F *__restrict__ cw, u32 n, int n_slices, Dig *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    F x = cw[t], y = cw[n + t];
    for (int s = 0; s <= n_slices; ++s) {                       // the last block is the mask slice's pair: zeros.  ONE instance of the 33 KB block
        const unsigned m[8] = {(unsigned) x.re, (unsigned) (x.re >> 32), (unsigned) x.im, (unsigned) (x.im >> 32), (unsigned) y.re, (unsigned) (y.re >> 32), (unsigned) y.im, (unsigned) (y.im >> 32)};
        if (s + 1 < n_slices) { x = cw[(size_t) (s + 1) * 2 * n + t]; y = cw[(size_t) (s + 1) * 2 * n + n + t]; }      // next slice's pair: in flight during the block
        else { x = f_zero(); y = f_zero(); }
        vp_hhash64_asm(h, m);
    }
    Dig d;
    for (int i = 0; i < 4; ++i) d.w[i] = ((u64) h[2 * i + 1] << 32) | h[2 * i];
    out[t] = d;
}
int main() {
    const u32 n = 1u << 20; const int S = 64;
    std::vector<F> h_cw((size_t) S * 2 * n);
    u64 st = 88172645463325252ull;
    for (auto &v : h_cw) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v.re = st & ((1ull << 61) - 1); st ^= st << 13; st ^= st >> 7; st ^= st << 17; v.im = st & ((1ull << 61) - 1); }
    F *cw; Dig *o1, *o2;
    CK(hipMalloc(&cw, h_cw.size() * sizeof(F))); CK(hipMalloc(&o1, n * sizeof(Dig))); CK(hipMalloc(&o2, n * sizeof(Dig)));
    CK(hipMemcpy(cw, h_cw.data(), h_cw.size() * sizeof(F), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // occupancy of the compiler's kernel limited by dynamic LDS (a 256-thread workgroup = one wave per SIMD): 160 KB / lds -> waves per SIMD
    for (int wps = 1; wps <= 8; ++wps) {
        const size_t lds = wps == 8 ? 0 : (size_t) (160 * 1024 / wps) - 1024;
        CK(hipFuncSetAttribute((const void *) k_chain_c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        float bestw = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain_c, dim3(n / 256), dim3(256), lds, 0, cw, n, S, o1);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) bestw = ms < bestw ? ms : bestw;
        }
        printf("compiler version, at most %d waves per SIMD (LDS %zu B per workgroup): %.3f ms\n", wps, lds, bestw);
    }
    const size_t lds_c = 0;
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) {
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_chain_c, dim3(n / 256), dim3(256), lds_c, 0, cw, n, S, o1);
            else hipLaunchKernelGGL(k_chain_asm, dim3(n / 256), dim3(256), 0, 0, cw, n, S, o2);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best[v] = ms < best[v] ? ms : best[v];
            printf("rep %d %s %.3f ms  %.3e Keccak-f/s\n", rep, v ? "asm     " : "compiler", ms, (double) n * (S + 1) / (ms * 1e-3));
        }
    }
    std::vector<Dig> a(n), b(n);
    CK(hipMemcpy(a.data(), o1, n * sizeof(Dig), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, n * sizeof(Dig), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (u32 i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) if (a[i].w[k] != b[i].w[k]) ++bad;
    printf("digests differing words: %zu of %u   best compiler %.3f ms, asm %.3f ms (%.1f %%)\n", bad, 4 * n, best[0], best[1], 100.0 * (best[1] / best[0] - 1));
    return bad != 0;
}
