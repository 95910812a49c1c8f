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
#include "_build/vp_keccak_asm_variants.h"      // tools/build_micro_keccak.sh: the generator's other variants under other names
template <int V>
__global__ void __launch_bounds__(1024) k_chain_asm(const F *__restrict__ cw, u32 n, int n_slices, Dig *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 tc = t < n ? t : n - 1;
    const F *x = cw + tc;
    if (V == 0) vp_leaf_chain_asm(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
    else if (V == 1) vp_leaf_chain_asm_rot1_alignbit(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
    else if (V == 2) vp_leaf_chain_asm_nobar(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
    else if (V == 3) vp_leaf_chain_asm_msgbefore(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
    else if (V == 4) vp_leaf_chain_asm_add(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
    else vp_leaf_chain_asm_add_before(x, x + n, 2u * n * 16u, (unsigned) n_slices, out + tc, t < n ? 1u : 0u);
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
    const char *names[7] = {"compiler (256-thread workgroups)", "asm (product form: a barrier after each rotation phase)", "asm, rot(C,1) by add/shift/bitop3, one barrier per round", "asm (product form), NO barriers", "asm, barrier after rho only", "asm, barriers only BEFORE the rotation phases", "asm, barriers at both ends of the rotation phases"};
    float best[7] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    Dig *o3; CK(hipMalloc(&o3, n * sizeof(Dig)));
    for (int rep = 0; rep < 5; ++rep) {
        for (int v = 0; v < 7; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_chain_c, dim3(n / 256), dim3(256), 0, 0, cw, n, S, o1);
            else if (v == 1) hipLaunchKernelGGL(k_chain_asm<0>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o2);
            else if (v == 2) hipLaunchKernelGGL(k_chain_asm<1>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o3);
            else if (v == 3) hipLaunchKernelGGL(k_chain_asm<2>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o3);
            else if (v == 4) hipLaunchKernelGGL(k_chain_asm<3>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o3);
            else if (v == 5) hipLaunchKernelGGL(k_chain_asm<4>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o3);
            else hipLaunchKernelGGL(k_chain_asm<5>, dim3((n + 1023) / 1024), dim3(1024), 0, 0, cw, n, S, o3);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best[v] = ms < best[v] ? ms : best[v];
        }
    }
    for (int v = 0; v < 7; ++v) printf("%-58s %.3f ms  %.3e Keccak-f/s  (%+.1f %% vs compiler)\n", names[v], best[v], (double) n * (S + 1) / (best[v] * 1e-3), 100.0 * (best[v] / best[0] - 1));
    std::vector<Dig> a(n), b(n);
    CK(hipMemcpy(a.data(), o1, n * sizeof(Dig), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, n * sizeof(Dig), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (u32 i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) if (a[i].w[k] != b[i].w[k]) ++bad;
    std::vector<Dig> c3(n); CK(hipMemcpy(c3.data(), o3, n * sizeof(Dig), hipMemcpyDeviceToHost));
    size_t bad3 = 0; for (u32 i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) if (a[i].w[k] != c3[i].w[k]) ++bad3;
    printf("digests differing words: %zu (product form) %zu (last variant) of %u\n", bad, bad3, 4 * n);
    return bad != 0;
}
