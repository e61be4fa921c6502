// VALU throughput vs waves per SIMD and ILP: can 3 waves/SIMD reach the 2-cycle issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
template <int OP, int ILP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 77u, a4 = a0 * 5u, a5 = a0 + 9u, a6 = a0 ^ 0xf0u, a7 = a0 + 1234u;
    uint32_t b = seed * 7u + 3u, c = seed + 0x3f800000u;
    for (int i = 0; i < iters; ++i) {
#define ONE(r) \
        if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r) : "v"(b)); \
        else if constexpr (OP == 1) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else asm volatile("v_sub_f32 %0, %1, %0\n v_med3_f32 %0, %1, %2, %0\n v_xor_b32 %0, %1, %0" : "+v"(r) : "v"(c), "v"(b));
        if constexpr (ILP == 8) { REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7)) }
        else if constexpr (ILP == 2) { REP8(ONE(a0) ONE(a1) ONE(a0) ONE(a1) ONE(a0) ONE(a1) ONE(a0) ONE(a1)) }
        else { REP8(ONE(a0) ONE(a0) ONE(a0) ONE(a0) ONE(a0) ONE(a0) ONE(a0) ONE(a0)) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP, int ILP> void run(const char* name, uint32_t* d, int wps, int iters) {
    int blocks = 256 * wps; // 256-thread blocks: 4 waves = 1 per SIMD per block per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (OP == 2 ? 3.0 : 1.0) * blocks * 4.0 * iters * 64.0;
    double r = n / (ms * 1e-3) / 1024.0;
    printf("%-14s ILP%d waves/SIMD=%d : %.2f cycles/inst/SIMD @2.4GHz\n", name, ILP, wps, 2.4e9 / r);
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        run<0, 8>("xor", d, wps, 20000); run<0, 2>("xor", d, wps, 20000); run<0, 1>("xor", d, wps, 20000);
        run<1, 8>("med3", d, wps, 20000); run<1, 1>("med3", d, wps, 20000);
        run<2, 8>("sub+med3+xor", d, wps, 7000); run<2, 1>("sub+med3+xor", d, wps, 7000);
    }
    return 0;
}
