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

#include "nrldpc_hostpath.h"
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

// ---- hard decisions, one byte per bit -> one bit per bit (nrldpc_decode_packed: 8x fewer bytes over PCIe and through the
// caller's copy).  One thread per output byte; rows whose length and address allow it read their eight input bytes as one
// 64-bit word and gather the bits with one multiply (bytes are 0 / 1: byte j lands on bit 56 + j, no carries meet).
__global__ __launch_bounds__(256) void nrldpc_pack_bits_kernel(const uint8_t* __restrict__ hard, uint8_t* __restrict__ packed, int rows,
                                                               int K, int KB8, int wide) {
    const size_t total = (size_t)rows * KB8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / KB8;
        const int b = (int)(i - r * KB8);
        const uint8_t* src = hard + r * (size_t)K + 8 * (size_t)b;
        uint32_t v;
        if (wide) {
            const uint64_t x = *reinterpret_cast<const uint64_t*>(src) & 0x0101010101010101ull;
            v = (uint32_t)((x * 0x0102040810204080ull) >> 56);
        } else {
            v = 0;
            const int n = K - 8 * b < 8 ? K - 8 * b : 8;
            for (int j = 0; j < n; ++j) v |= (uint32_t)(src[j] & 1u) << j;
        }
        packed[i] = (uint8_t)v;
    }
}

hipError_t launch_pack_bits(const uint8_t* d_hard, uint8_t* d_packed, int rows, int K, hipStream_t stream) {
    if (rows <= 0 || K <= 0) return hipSuccess;
    const int KB8 = (K + 7) / 8;
    const int wide = (K % 8 == 0) && (reinterpret_cast<uintptr_t>(d_hard) & 7) == 0;
    const size_t want = ((size_t)rows * KB8 + 255) / 256;
    const int grid = (int)(want > 16384 ? 16384 : want);
    hipLaunchKernelGGL(nrldpc_pack_bits_kernel, dim3(grid), dim3(256), 0, stream, d_hard, d_packed, rows, K, KB8, wide);
    return hipGetLastError();
}

} // namespace nrldpc
