// valu_rate3.hip -- round 6: issue intervals of the 16-bit datapath candidates for the headline kernel:
// un-packed f16 with |x| / op_sel, packed f16 (two edges or two rows per VGPR), the SDWA and gfx950 fp8 <-> f16 pair
// converters, and the d16 LDS forms.  Same harness as valu_rate2.hip: 8 independent chains per wave, 8 waves per SIMD,
// cycles at a nominal 2.4 GHz (a full-rate op reads 2.25-2.5).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate3 tools/ubench/valu_rate3.hip && ./valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t sm[4096];
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 77u, a4 = a0 * 5u, a5 = a0 + 9u, a6 = a0 ^ 0xf0u, a7 = a0 + 1234u;
    uint32_t b = seed * 7u + 3u + threadIdx.x, c = seed + 0x3c003c00u + threadIdx.x;
    uint32_t sc = 0x3f800000u; // scale 1.0 for the scalef32 converters
    asm volatile("" : "+v"(sc));
    uint32_t la = (threadIdx.x & 1023u) * 4u; // LDS byte address
    if (OP >= 100) { for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i * seed; __syncthreads(); }
    for (int i = 0; i < iters; ++i) {
#define ONE(r) \
         if constexpr (OP == 0) asm volatile("v_sub_f16_e64 %0, |%1|, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 1) asm volatile("v_min_f16_e64 %0, |%1|, |%0|" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 2) asm volatile("v_max_f16_e64 %0, |%1|, |%0|" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 3) asm volatile("v_med3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 4) asm volatile("v_med3_f16 %0, |%0|, %1, %2 op_sel:[1,0,0,1]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 5) asm volatile("v_min3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 6) asm volatile("v_cmp_eq_f16_e64 vcc, |%0|, %1" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 7) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 8) asm volatile("v_pack_b32_f16 %0, %0, %1 op_sel:[1,0,0]" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 9) asm volatile("v_bitop3_b16 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 10) asm volatile("v_bitop3_b16 %0, %0, %1, %2 bitop3:0x96 op_sel:[1,0,0,1]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 11) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 12) asm volatile("v_fma_f16 %0, %0, %1, %2 op_sel:[1,0,0,1]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 13) asm volatile("v_sub_f16_e64 %0, %1, |%0| clamp" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 14) asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 15) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 16) asm volatile("v_pk_add_f16 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 17) asm volatile("v_pk_add_f16 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1] clamp" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 18) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 19) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 20) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 21) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 22) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 23) asm volatile("v_cvt_f16_i16_sdwa %0, sext(%1) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 24) asm volatile("v_cvt_i16_f16_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 25) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 26) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 27) asm volatile("v_cvt_scalef32_pk_f16_bf8 %0, %1, %2" : "=v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 28) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 29) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 30) asm volatile("v_cvt_scalef32_pk_bf8_f16 %0, %1, %2" : "+v"(r) : "v"(c), "v"(sc)); \
        else if constexpr (OP == 31) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 32) asm volatile("v_min_f16_sdwa %0, |%1|, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 33) asm volatile("v_sub_f16_sdwa %0, %1, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 34) asm volatile("v_sub_i16 %0, %1, %0 op_sel:[1,0,0]" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 35) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(r)); \
        else if constexpr (OP == 36) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 37) asm volatile("v_and_b32 %0, 0x7fff7fff, %0" : "+v"(r)); \
        else if constexpr (OP == 38) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 39) asm volatile("v_pk_add_f16 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 40) asm volatile("v_sub_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 41) asm volatile("v_min_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 42) asm volatile("v_add_f16_e64 %0, %0, %1 clamp" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 43) asm volatile("v_max3_f16 %0, %0, %1, %2 op_sel:[1,0,0,1]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 44) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 45) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(r)); \
        else if constexpr (OP == 46) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(c)); \
        else if constexpr (OP == 100) asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(r) : "v"(la)); \
        else if constexpr (OP == 101) asm volatile("ds_read_u16_d16 %0, %1 offset:64" : "+v"(r) : "v"(la)); \
        else if constexpr (OP == 102) asm volatile("ds_read_u16_d16_hi %0, %1 offset:64" : "+v"(r) : "v"(la)); \
        else if constexpr (OP == 103) asm volatile("ds_write_b32 %1, %0 offset:64" : : "v"(r), "v"(la)); \
        else if constexpr (OP == 104) asm volatile("ds_write_b16 %1, %0 offset:64" : : "v"(r), "v"(la)); \
        else if constexpr (OP == 105) asm volatile("ds_write_b16_d16_hi %1, %0 offset:64" : : "v"(r), "v"(la)); \
        else if constexpr (OP == 106) asm volatile("ds_read_u16 %0, %1 offset:64" : "=v"(r) : "v"(la));
        REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
        if constexpr (OP >= 100) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ sm[threadIdx.x];
}
template <int OP> void run(const char* name, uint32_t* d, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 64.0;
    double r = winst / (ms * 1e-3) / (256.0 * 4.0);
    printf("%-44s %8.3f ms -> %.2f cycles/inst @2.4GHz\n", name, ms, 2.4e9 / r);
}
int main() {
    uint32_t* d; int blocks = 256 * 8; (void)hipMalloc(&d, blocks * 256 * 4);
    int it = 5000;
    run<46>("v_mov_b32 (reference: full rate)", d, blocks, it);
    run<40>("v_sub_f16 e32", d, blocks, it);
    run<0>("v_sub_f16_e64 |x|", d, blocks, it);
    run<41>("v_min_f16 e32", d, blocks, it);
    run<1>("v_min_f16_e64 |x|,|y|", d, blocks, it);
    run<2>("v_max_f16_e64 |x|,|y|", d, blocks, it);
    run<3>("v_med3_f16", d, blocks, it);
    run<4>("v_med3_f16 |x| op_sel hi->hi", d, blocks, it);
    run<5>("v_min3_f16", d, blocks, it);
    run<43>("v_max3_f16 op_sel hi->hi", d, blocks, it);
    run<6>("v_cmp_eq_f16_e64 |x|", d, blocks, it);
    run<7>("v_pack_b32_f16", d, blocks, it);
    run<8>("v_pack_b32_f16 op_sel", d, blocks, it);
    run<9>("v_bitop3_b16", d, blocks, it);
    run<10>("v_bitop3_b16 op_sel hi->hi", d, blocks, it);
    run<11>("v_fma_f16", d, blocks, it);
    run<12>("v_fma_f16 op_sel hi->hi", d, blocks, it);
    run<13>("v_sub_f16_e64 |y| clamp", d, blocks, it);
    run<42>("v_add_f16_e64 clamp", d, blocks, it);
    run<34>("v_sub_i16 op_sel", d, blocks, it);
    run<44>("v_pk_add_f16", d, blocks, it);
    run<14>("v_pk_min_f16", d, blocks, it);
    run<15>("v_pk_max_f16", d, blocks, it);
    run<16>("v_pk_add_f16 neg", d, blocks, it);
    run<17>("v_pk_add_f16 neg clamp", d, blocks, it);
    run<39>("v_pk_add_f16 op_sel mix", d, blocks, it);
    run<18>("v_pk_fma_f16", d, blocks, it);
    run<19>("v_pk_mul_f16", d, blocks, it);
    run<20>("v_pk_minimum3_f16", d, blocks, it);
    run<21>("v_pk_maximum3_f16", d, blocks, it);
    run<22>("v_minimum3_f32", d, blocks, it);
    run<35>("v_pk_ashrrev_i16", d, blocks, it);
    run<37>("v_and_b32 literal 0x7fff7fff", d, blocks, it);
    run<45>("v_lshrrev_b32 16", d, blocks, it);
    run<38>("v_perm_b32 (vgpr selector)", d, blocks, it);
    run<23>("v_cvt_f16_i16_sdwa sext BYTE_2 -> WORD_1", d, blocks, it);
    run<24>("v_cvt_i16_f16_sdwa WORD_1 -> BYTE_2", d, blocks, it);
    run<25>("v_cvt_scalef32_pk_f16_fp8 lo word", d, blocks, it);
    run<26>("v_cvt_scalef32_pk_f16_fp8 hi word", d, blocks, it);
    run<27>("v_cvt_scalef32_pk_f16_bf8", d, blocks, it);
    run<28>("v_cvt_scalef32_pk_fp8_f16 -> lo", d, blocks, it);
    run<29>("v_cvt_scalef32_pk_fp8_f16 -> hi", d, blocks, it);
    run<30>("v_cvt_scalef32_pk_bf8_f16", d, blocks, it);
    run<31>("v_sat_pk_u8_i16", d, blocks, it);
    run<36>("v_cvt_pk_f16_f32", d, blocks, it);
    run<32>("v_min_f16_sdwa hi,hi->hi", d, blocks, it);
    run<33>("v_sub_f16_sdwa src1 BYTE_1", d, blocks, it);
    it = 1000;
    run<100>("ds_read_b32 (8 in flight, then wait)", d, blocks, it);
    run<106>("ds_read_u16", d, blocks, it);
    run<101>("ds_read_u16_d16", d, blocks, it);
    run<102>("ds_read_u16_d16_hi", d, blocks, it);
    run<103>("ds_write_b32", d, blocks, it);
    run<104>("ds_write_b16", d, blocks, it);
    run<105>("ds_write_b16_d16_hi", d, blocks, it);
    return 0;
}
