// What does s_mov_b64 do with a 32-bit literal on gfx950?  The assembler accepts both spellings below and encodes the same 32
// literal bits; the hardware either zero- or sign-extends them.  hipcc (ROCm 7.2) emits this instruction for 64-bit lane-mask
// constants of BOTH kinds (lane_ge): one of the two must come out wrong.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/smov64_literal.hip -o gpurun_out/smov64 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out) {
    unsigned long long a, b, c;
    asm volatile("s_mov_b64 %0, 0xfffffffffffffc00" : "=s"(a));
    asm volatile("s_mov_b64 %0, 0xfffffc00" : "=s"(b));
    asm volatile("s_mov_b64 %0, 0xffffffff" : "=s"(c));
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; }
}
int main() {
    unsigned long long* d; unsigned long long h[3];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("s_mov_b64 0xfffffffffffffc00 -> %016llx\ns_mov_b64 0xfffffc00         -> %016llx\ns_mov_b64 0xffffffff         -> %016llx\n", h[0], h[1], h[2]);
    return 0;
}
