#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 77u, a4 = a0 * 5u, a5 = a0 + 9u, a6 = a0 ^ 0xf0u, a7 = a0 + 1234u;
    uint32_t b = seed * 7u + 3u + threadIdx.x, c = seed + 0x3f800000u + threadIdx.x; uint32_t sb = seed * 11u;
    f2 pa0 = {1.f * a0, 2.f}, pa1 = {3.f, 1.f * a1}, pa2 = pa0 + 1.f, pa3 = pa1 + 2.f, pa4 = pa0 * 3.f, pa5 = pa1 * 5.f, pa6 = pa0 - 1.f, pa7 = pa1 - 2.f, pc = {1.5f, 0.5f * seed};
    for (int i = 0; i < iters; ++i) {
#define ONE(r) \
         if constexpr (OP == 0) asm volatile("v_min_f32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 3) asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 4) asm volatile("v_sub_f32 %0, -%1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 6) asm volatile("v_or_b32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 7) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r)); \
        else if constexpr (OP == 8) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r)); \
        else if constexpr (OP == 9) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r)); \
        else if constexpr (OP == 10) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 11) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 12) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 13) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 14) asm volatile("v_cmp_eq_u32 vcc, %1, %0" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 15) asm volatile("v_rndne_f32 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 16) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 17) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 18) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 19) asm volatile("v_bfe_i32 %0, %0, 8, 8" : "+v"(r)); \
        else if constexpr (OP == 20) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r)); \
        else if constexpr (OP == 21) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 22) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 23) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 24) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 25) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 26) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 27) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 28) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 29) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 30) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 31) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p##r) : "v"(pc)); \
        else if constexpr (OP == 32) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p##r) : "v"(pc)); \
        else if constexpr (OP == 33) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p##r) : "v"(pc)); \
        else if constexpr (OP == 34) asm volatile("v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 35) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 36) asm volatile("v_sub_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 37) asm volatile("v_xor_b32 %0, 0x12345678, %0" : "+v"(r)); \
        else if constexpr (OP == 38) asm volatile("v_add_u32 %0, 0x1234, %0" : "+v"(r)); \
        else if constexpr (OP == 39) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "s"(sb)); \
        else if constexpr (OP == 40) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 41) asm volatile("v_mul_f32 %0, 0x3f400000, %0" : "+v"(r)); \
        else if constexpr (OP == 42) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 43) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 44) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 45) asm volatile("v_sub_f32_e64 %0, %1, |%0| clamp" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 46) asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 47) asm volatile("v_cvt_f32_ubyte3 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 48) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 49) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 50) asm volatile("v_max_u32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 51) asm volatile("v_min_i32 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 52) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 53) asm volatile("v_cmp_eq_f32 vcc, |%0|, %1" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 54) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 55) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 56) asm volatile("v_sub_u16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 57) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 58) asm volatile("v_med3_i16 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 59) asm volatile("v_min_i16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 60) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(p##r) : "v"(r)); \
        else if constexpr (OP == 61) asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(p##r) : "v"(r)); \
        else if constexpr (OP == 62) asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, 1.0 op_sel:[1,0,0]" : "=v"(p##r) : "v"(r)); \
        else if constexpr (OP == 63) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 64) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 65) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 66) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 67) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 68) asm volatile("v_min_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 69) asm volatile("v_max_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 70) asm volatile("v_min_u16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 71) asm volatile("v_cmp_lt_u16 vcc, %1, %0" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 72) asm volatile("v_cvt_f32_fp8_sdwa %0, %0 src0_sel:BYTE_2" : "+v"(r)); \
        else if constexpr (OP == 73) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r) : "v"(c), "v"(b)); \
        else if constexpr (OP == 74) asm volatile("v_max_i16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 75) asm volatile("v_add_f16 %0, %1, %0" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 76) asm volatile("v_cmp_eq_f16 vcc, %1, %0" : : "v"(r), "v"(c) : "vcc"); \
        else if constexpr (OP == 77) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 78) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r)); \
        else if constexpr (OP == 79) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(c), "v"(b));
        REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
    }
    f2 ps = pa0 + pa1 + pa2 + pa3 + pa4 + pa5 + pa6 + pa7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(ps.x + ps.y);
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
    printf("%-34s %8.3f ms -> %.2f cycles/inst @2.4GHz\n", name, ms, 2.4e9 / r);
}
int main() {
    uint32_t* d; int blocks = 256 * 8; (void)hipMalloc(&d, blocks * 256 * 4);
    int it = 10000;
    run<0>("v_min_f32 e32", d, blocks, it);
    run<1>("v_max_f32 e32", d, blocks, it);
    run<2>("v_mul_f32", d, blocks, it);
    run<3>("v_add_f32 |x| (e64)", d, blocks, it);
    run<4>("v_sub_f32 neg (e64)", d, blocks, it);
    run<5>("v_and_b32", d, blocks, it);
    run<6>("v_or_b32", d, blocks, it);
    run<7>("v_lshlrev_b32", d, blocks, it);
    run<8>("v_lshrrev_b32", d, blocks, it);
    run<9>("v_ashrrev_i32", d, blocks, it);
    run<10>("v_sub_u32", d, blocks, it);
    run<11>("v_subrev_u32", d, blocks, it);
    run<12>("v_cndmask_b32 (vcc const)", d, blocks, it);
    run<13>("v_cmp_lt_f32 e32", d, blocks, it);
    run<14>("v_cmp_eq_u32 e32", d, blocks, it);
    run<15>("v_rndne_f32", d, blocks, it);
    run<16>("v_cvt_f32_i32", d, blocks, it);
    run<17>("v_cvt_i32_f32", d, blocks, it);
    run<18>("v_cvt_f32_ubyte1", d, blocks, it);
    run<19>("v_bfe_i32", d, blocks, it);
    run<20>("v_bfe_u32", d, blocks, it);
    run<21>("v_perm_b32", d, blocks, it);
    run<22>("v_add3_u32", d, blocks, it);
    run<23>("v_lshl_add_u32", d, blocks, it);
    run<24>("v_lshl_or_b32", d, blocks, it);
    run<25>("v_xad_u32", d, blocks, it);
    run<26>("v_alignbit_b32", d, blocks, it);
    run<27>("v_bitop3_b32", d, blocks, it);
    run<28>("v_min3_f32", d, blocks, it);
    run<29>("v_max3_f32", d, blocks, it);
    run<30>("v_mad_u32_u24", d, blocks, it);
    run<31>("v_pk_add_f32 (2 regs)", d, blocks, it);
    run<32>("v_pk_mul_f32 (2 regs)", d, blocks, it);
    run<33>("v_pk_fma_f32 (2 regs)", d, blocks, it);
    run<34>("v_or_b32_sdwa BYTE_1", d, blocks, it);
    run<35>("v_mov_b32_sdwa B0->B2 preserve", d, blocks, it);
    run<36>("v_sub_f32_sdwa (no sel)", d, blocks, it);
    run<37>("v_xor_b32 literal", d, blocks, it);
    run<38>("v_add_u32 literal", d, blocks, it);
    run<39>("v_add_u32 sgpr", d, blocks, it);
    run<40>("v_fmac_f32", d, blocks, it);
    run<41>("v_mul_f32 x0.75 inline?", d, blocks, it);
    run<42>("v_med3_i32", d, blocks, it);
    run<43>("v_sad_u32", d, blocks, it);
    run<44>("v_cvt_pkrtz_f16_f32", d, blocks, it);
    run<45>("v_sub_f32 e64 |x| clamp", d, blocks, it);
    run<46>("v_cvt_pk_u8_f32", d, blocks, it);
    run<47>("v_cvt_f32_ubyte3", d, blocks, it);
    run<48>("v_fma_f32 (3 vgpr)", d, blocks, it);
    run<49>("v_min_u32", d, blocks, it);
    run<50>("v_max_u32", d, blocks, it);
    run<51>("v_min_i32", d, blocks, it);
    run<52>("v_med3_u32", d, blocks, it);
    run<53>("v_cmp_eq_f32 |x|", d, blocks, it);
    run<54>("v_add_f32 e64 clamp", d, blocks, it);
    run<55>("v_cvt_f32_ubyte0", d, blocks, it);
    run<56>("v_sub_u16", d, blocks, it);
    run<57>("v_pk_sub_i16", d, blocks, it);
    run<58>("v_med3_i16", d, blocks, it);
    run<59>("v_min_i16", d, blocks, it);
    run<60>("v_cvt_pk_f32_fp8 (2 regs out)", d, blocks, it);
    run<61>("v_cvt_pk_f32_fp8_sdwa WORD_1", d, blocks, it);
    run<62>("v_cvt_scalef32_pk_f32_fp8 opsel", d, blocks, it);
    run<63>("v_cvt_pk_fp8_f32 hi", d, blocks, it);
    run<64>("v_cvt_pk_fp8_f32 lo", d, blocks, it);
    run<65>("v_fma_mix_f32", d, blocks, it);
    run<66>("v_dot4_i32_i8", d, blocks, it);
    run<67>("v_dot2c_f32_bf16", d, blocks, it);
    run<68>("v_min_f16", d, blocks, it);
    run<69>("v_max_f16", d, blocks, it);
    run<70>("v_min_u16", d, blocks, it);
    run<71>("v_cmp_lt_u16", d, blocks, it);
    run<72>("v_cvt_f32_fp8_sdwa", d, blocks, it);
    run<73>("v_dot2_f32_f16", d, blocks, it);
    run<74>("v_max_i16", d, blocks, it);
    run<75>("v_add_f16", d, blocks, it);
    run<76>("v_cmp_eq_f16", d, blocks, it);
    run<77>("v_cvt_f32_f16", d, blocks, it);
    run<78>("v_cvt_f16_f32", d, blocks, it);
    run<79>("v_cndmask_b32 (independent dst)", d, blocks, it);
    return 0;
}
