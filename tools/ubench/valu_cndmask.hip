// Issue rate of v_cndmask_b32 on gfx950, measured with INDEPENDENT chains (the r02 table's v_cndmask lines
// came from valu_rate2.hip's OP 12 / 79, whose eight "independent" destinations the compiler folded into one register with
// an s_nop between writes -- 23.6 cycles was a write-after-write chain, not an issue rate).  Here every instruction reads and
// writes its own accumulator (8 chains), VCC is written once per loop trip by a real compare, and the pair the decoder
// kernels actually issue -- v_cmp_eq_f32_e64 vcc, |t|, m ; v_cndmask_b32_e32 mag, M1, M2, vcc -- is timed as a pair.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cndmask.hip -o tools/ubench/valu_cndmask && tools/ubench/valu_cndmask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 77u, a4 = a0 * 5u, a5 = a0 + 9u, a6 = a0 ^ 0xf0u, a7 = a0 + 1234u;
    uint32_t c = seed + 0x3f800000u + threadIdx.x, b = seed * 7u + threadIdx.x;
    uint64_t m = 0x5555555555555555ull ^ seed;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0 || OP == 3) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(b), "v"(c) : "vcc");
#define ONE(r) \
        if constexpr (OP == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : ); \
        else if constexpr (OP == 1) asm volatile("v_cmp_eq_f32_e64 vcc, |%0|, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r) : "v"(c), "v"(b) : "vcc"); \
        else if constexpr (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(c), "s"(m)); \
        else if constexpr (OP == 3) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(r) : "v"(c)); \
        else if constexpr (OP == 4) asm volatile("v_cmp_eq_f32_e64 vcc, |%0|, %1" : : "v"(r), "v"(c) : "vcc");
        REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char* name, uint32_t* d, int blocks, int iters, int per) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * 4 * iters * 64.0 * per;          // wave instructions
    double r = winst / (ms * 1e-3) / (256.0 * 4.0);                    // per SIMD per second
    printf("%-44s %8.3f ms -> %.2f cycles/inst @2.4GHz\n", name, ms, 2.4e9 / r);
}
int main() {
    uint32_t* d; int blocks = 256 * 8; (void)hipMalloc(&d, blocks * 256 * 4);
    int it = 10000;
    run<3>("v_mov_b32 (8 independent chains)", d, blocks, it, 1);
    run<0>("v_cndmask_b32_e32 vcc (8 independent chains)", d, blocks, it, 1);
    run<2>("v_cndmask_b32_e64 sgpr-pair mask", d, blocks, it, 1);
    run<4>("v_cmp_eq_f32_e64 vcc, |x|, y", d, blocks, it, 1);
    run<1>("v_cmp_eq_f32_e64 + v_cndmask_b32_e32 (per inst)", d, blocks, it, 2);
    return 0;
}
