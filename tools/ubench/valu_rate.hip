// Micro-benchmark: issue rate of the VALU ops the decoder uses, in cycles per wave64 instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 77u, a4 = a0 * 5u, a5 = a0 + 9u, a6 = a0 ^ 0xf0u, a7 = a0 + 1234u;
    uint32_t b = seed * 7u + 3u, c = seed + 0x3f800000u;
    for (int i = 0; i < iters; ++i) {
#define ONE(r) \
        if constexpr (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r) : "v"(b)); \
        else if constexpr (OP == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(b)); \
        else if constexpr (OP == 2) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r) : "v"(b)); \
        else if constexpr (OP == 3) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 5) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 6) asm volatile("v_min_f32 %0, |%1|, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 7) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(r) : "v"(b)); \
        else if constexpr (OP == 8) asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 9) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 10) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 11) asm volatile("v_cmp_eq_f32 vcc, |%0|, %1" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 12) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 13) asm volatile("v_pk_min_i16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 14) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 15) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 16) asm volatile("v_pk_add_i16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 18) asm volatile("v_max_i32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 19) asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(r) : "v"(c));
        REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> double run(const char* name, uint32_t* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 64.0; // wave-instructions
    double per_simd_per_s = winst / (ms * 1e-3) / (256.0 * 4.0);
    printf("%-28s %8.3f ms  %.3f G wave-inst/s/SIMD  -> %.2f cycles/inst @2.4GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    return ms;
}

int main() {
    uint32_t* d; int blocks = 256 * 8; hipMalloc(&d, blocks * 256 * 4);
    int it = 20000;
    run<0>("v_xor_b32", d, blocks, it); run<1>("v_add_u32", d, blocks, it); run<2>("v_min_u32", d, blocks, it);
    run<3>("v_add_f32", d, blocks, it); run<14>("v_sub_f32", d, blocks, it); run<4>("v_fma_f32", d, blocks, it);
    run<5>("v_med3_f32", d, blocks, it); run<6>("v_min_f32 |x|", d, blocks, it);
    run<7>("v_cvt_f32_i32_sdwa", d, blocks, it); run<8>("v_cvt_i32_f32_sdwa preserve", d, blocks, it);
    run<9>("v_and_or_b32", d, blocks, it); run<15>("v_bfi_b32", d, blocks, it); run<10>("v_cndmask_b32", d, blocks, it); run<11>("v_cmp_eq_f32", d, blocks, it);
    run<12>("v_pk_add_f16", d, blocks, it); run<13>("v_pk_min_i16", d, blocks, it); run<16>("v_pk_add_i16", d, blocks, it); run<19>("v_pk_max_i16", d, blocks, it);
    run<17>("v_mov_b32", d, blocks, it); run<18>("v_max_i32", d, blocks, it);
    return 0;
}
