// Instruction-rate probe for gfx950 (development tool, not part of the product): how many cycles a SIMD needs per
// wave-instruction of the kinds the F_p^2 multiply is made of.  Each kernel runs a long chain of 8 independent
// dependency chains per lane; 4 waves per SIMD hide the latency.  Build: hipcc --offload-arch=gfx950 -O3 -o micro_rates micro_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../virgo-plus_amd/csrc/vp_field.h"
using namespace vp;
#define ITER 4096
template <int OP> __global__ void __launch_bounds__(256) k_rate(uint64_t *out, uint64_t seed) {
    uint64_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = seed * (threadIdx.x + 1 + j * 977) + blockIdx.x;
    uint32_t m = (uint32_t) seed | 1;
    double d[8]; for (int j = 0; j < 8; ++j) d[j] = (double) x[j];
    double dm = (double) m * 1e-9;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"((uint32_t) m), "v"((uint32_t) (it + j)) : "vcc"); }
            else if (OP == 1) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[j]) : "v"(seed)); }
            else if (OP == 2) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 3) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 4) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[j]) : "v"(dm)); }
            else if (OP == 5) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 6) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 7) { asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[j])); }
            else if (OP == 8) { uint32_t lo = (uint32_t) x[j], hi = (uint32_t) (x[j] >> 32);
                                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(m) : "vcc"); x[j] = ((uint64_t) hi << 32) | lo; }
            else if (OP == 9) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 10) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo) : "v"(m), "v"((uint32_t) it)); x[j] = lo; }
            else if (OP == 11) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo) : "v"(m)); x[j] = lo; }
            else if (OP == 12) { uint32_t lo = (uint32_t) x[j]; asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(m)); x[j] = lo; }
        }
    }
    uint64_t s = 0;
    for (int j = 0; j < 8; ++j) s += x[j] + (uint64_t) d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// schoolbook variant: real = ac + C - bd, imag = ad + bc (16 mads, no 128-bit subtraction for the imaginary part)
__device__ __forceinline__ F f_mul4(const F &a, const F &b) {
    const u128 C = ((u128) P61) << 61;
    u128 re = (u128) a.re * b.re + C - (u128) a.im * b.im;
    u128 im = (u128) a.re * b.im + (u128) a.im * b.re;
    return f_make(m_red128(re), m_red128(im));
}
// weak reduction: result in [0, p] (p itself stands for 0), no compare/select
__device__ __forceinline__ u64 m_red128w(u128 x) {
    u64 lo = (u64) x & P61;
    u64 hi = (u64) (x >> 61);
    u64 s = lo + (hi & P61) + (hi >> 61);
    s = (s & P61) + (s >> 61);
    return (s & P61) + (s >> 61);
}
__device__ __forceinline__ F f_mul4w(const F &a, const F &b) {
    const u128 C = ((u128) P61) << 61;
    u128 re = (u128) a.re * b.re + C - (u128) a.im * b.im;
    u128 im = (u128) a.re * b.im + (u128) a.im * b.re;
    return f_make(m_red128w(re), m_red128w(im));
}
__device__ __forceinline__ F f_mulKw(const F &a, const F &b) {
    const u128 C = ((u128) P61) << 61;
    u128 ac = (u128) a.re * b.re;
    u128 bd = (u128) a.im * b.im;
    u128 cr = (u128) (a.re + a.im) * (b.re + b.im);
    return f_make(m_red128w(ac + C - bd), m_red128w(cr + C + C - ac - bd));
}
__device__ __forceinline__ F f_addw(const F &a, const F &b) {
    u64 s = a.re + b.re, t = a.im + b.im;
    return f_make((s & P61) + (s >> 61), (t & P61) + (t >> 61));
}
__device__ __forceinline__ F f_mad_lazy_p(const F &a, const F &b, const F &c) {
    const u128 C4 = ((u128) P61) << 63;
    const u128 re = (u128) a.re * b.re + C4 - (u128) a.im * b.im + c.re;
    const u128 im = (u128) a.re * b.im + (u128) a.im * b.re + c.im;
    return f_make(m_red128(re), m_red128(im));
}
// the product's field multiply: 4 independent chains per lane
template <int V> __global__ void __launch_bounds__(256) k_fmul(F *out, F seed) {
    F x[4];
    for (int j = 0; j < 4; ++j) x[j] = f_make((seed.re * (threadIdx.x + 1 + j)) & P61, (seed.im + threadIdx.x * 7 + j) & P61);
    F y = seed;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (V == 0) x[j] = f_mul(x[j], y);
            else if (V == 1) x[j] = f_lerp(x[j], y, x[(j + 1) & 3]);
            else if (V == 2) x[j] = f_add(x[j], y);
            else if (V == 3) x[j] = f_mul4(x[j], y);
            else if (V == 4) x[j] = f_mul4w(x[j], y);
            else if (V == 5) x[j] = f_mulKw(x[j], y);
            else if (V == 6) x[j] = f_addw(x[j], y);
            else if (V == 7) x[j] = f_sub(x[j], y);
            else if (V == 8) x[j] = f_mad_lazy_p(x[j], y, x[(j + 1) & 3]);
            else if (V == 9) x[j] = f_mad31(x[j], y, x[(j + 1) & 3]);
            else if (V == 10) x[j] = f_mad31(x[j], y, f_zero());
        }
    }
    F s = f_zero();
    for (int j = 0; j < 4; ++j) s = f_add(s, x[j]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> double time_ms(K launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    printf("device %s CUs %d clock %.3f GHz\n", pr.name, cus, ghz);
    const int blocks = cus * 8;          // 256 threads = 4 waves = one per SIMD; 8 blocks per CU -> 8 waves per SIMD
    uint64_t *o; hipMalloc(&o, (size_t) blocks * 256 * 8);
    F *of; hipMalloc(&of, (size_t) blocks * 256 * 16);
    const char *names[] = {"v_mad_u64_u32", "v_lshl_add_u64", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_add_u32", "v_mad_u32_u24", "v_lshlrev_b64", "add_co+addc (2 instr)", "v_mul_hi_u32_u24", "v_bitop3_b32", "v_alignbit_b32", "v_xor_b32"};
#define RUN(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, o, 0x9E3779B97F4A7C15ull); }); \
        double winst = (double) blocks * 4 * ITER * 8; double cyc = ms * 1e-3 * ghz * 1e9 * cus * 4 / winst; \
        printf("%-24s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[OP], ms, cyc); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    const char *fn[] = {"f_mul", "f_lerp", "f_add", "f_mul4 (schoolbook)", "f_mul4w (weak red)", "f_mulKw (karatsuba weak)", "f_addw (weak)", "f_sub", "f_mad_lazy (a*b+c, 128-bit)", "f_mad31 (a*b+c)", "f_mad31 (a*b)"};
#define RUNF(V) { double ms = time_ms([&] { hipLaunchKernelGGL(k_fmul<V>, dim3(blocks), dim3(256), 0, 0, of, f_make(123456789123ull, 987654321987ull)); }); \
        double ops = (double) blocks * 256 * ITER * 4; double cyc = ms * 1e-3 * ghz * 1e9 * cus * 4 / (ops / 64); \
        printf("%-24s %8.3f ms  %7.1f SIMD-cycles per wave-op   %.3e ops/s\n", fn[V], ms, cyc, ops / (ms * 1e-3)); }
    RUNF(0) RUNF(1) RUNF(2) RUNF(3) RUNF(4) RUNF(5) RUNF(6) RUNF(7) RUNF(8) RUNF(9) RUNF(10)
    return 0;
}
