// nrldpc_expand.hip -- device half of the host path's int8 wire format.
//
// nrldpc_decode (host pointers) has its copy threads quantise the caller's LLRs to the decoder's fixed-point grid while
// they copy (nrldpc_host_quant.h): q = NaN ? 0 : rint(clamp(llr * scale, +-127)) as int8, +inf as -128, so that 1 byte
// per LLR crosses PCIe.  This kernel turns a chunk back into the fp16 LLRs the decoder kernels ingest: q / scale is an
// integer of at most 7 bits times a power of two, exact in fp16, and the kernels' own ingest() of it -- multiply by the
// scale, round -- is q again; -128 becomes +inf, which ingest() maps to the filler value in a core column and to +127 in
// an extension column, exactly as it does for a +inf that came in as a float (NRLDPCDecoder.m:264).  The decoder kernels
// stay as they are (a third input format in their prologue cost the headline launch 0.7 %), and every lifting size is
// served.  HBM-bound: 1 byte in, 2 out per LLR, ~60 us for 4096 headline codewords, on the chunk's own stream.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "nrldpc_kernels.h"

namespace nrldpc {

__device__ __forceinline__ uint32_t expand_pair(uint32_t w, int sh, float inv_scale) {
    const int a = (int)(int8_t)(w >> sh), b = (int)(int8_t)(w >> (sh + 8));
    const __half ha = a == -128 ? __ushort_as_half((unsigned short)0x7C00u) : __float2half((float)a * inv_scale);
    const __half hb = b == -128 ? __ushort_as_half((unsigned short)0x7C00u) : __float2half((float)b * inv_scale);
    return (uint32_t)__half_as_ushort(ha) | ((uint32_t)__half_as_ushort(hb) << 16);
}

__global__ __launch_bounds__(256) void nrldpc_expand_i8_kernel(const int8_t* __restrict__ q, __half* __restrict__ out, size_t n,
                                                               float inv_scale, int vec) {
    const size_t nv = vec ? n >> 4 : 0; // 16 LLRs per thread and trip: one 16-byte load, two 16-byte stores
    const uint4* q4 = reinterpret_cast<const uint4*>(q);
    uint4* o4 = reinterpret_cast<uint4*>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = q4[i];
        uint4 lo, hi;
        lo.x = expand_pair(v.x, 0, inv_scale); lo.y = expand_pair(v.x, 16, inv_scale);
        lo.z = expand_pair(v.y, 0, inv_scale); lo.w = expand_pair(v.y, 16, inv_scale);
        hi.x = expand_pair(v.z, 0, inv_scale); hi.y = expand_pair(v.z, 16, inv_scale);
        hi.z = expand_pair(v.w, 0, inv_scale); hi.w = expand_pair(v.w, 16, inv_scale);
        o4[2 * i] = lo; o4[2 * i + 1] = hi;
    }
    const size_t done = nv << 4;
    for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int a = (int)q[i];
        out[i] = a == -128 ? __ushort_as_half((unsigned short)0x7C00u) : __float2half((float)a * inv_scale);
    }
}

// (a chunk of odd-Z codewords may start at an address that is not a multiple of 16: one LLR per thread and trip then)
hipError_t launch_expand_i8(const int8_t* d_q, void* d_out_f16, size_t n, float inv_scale, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int vec = ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_out_f16)) & 15) == 0;
    const size_t want = ((vec ? n >> 4 : n) + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : want > 8192 ? 8192 : want);
    hipLaunchKernelGGL(nrldpc_expand_i8_kernel, dim3(grid), dim3(256), 0, stream, d_q, static_cast<__half*>(d_out_f16), n, inv_scale, vec);
    return hipGetLastError();
}

} // namespace nrldpc
