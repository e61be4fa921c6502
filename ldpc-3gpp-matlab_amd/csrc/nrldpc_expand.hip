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

// The same for a chunk whose rows were sent COMPACT: n_rows rows of `act` int8 each -> the first act LLRs of rows that are `pitch`
// fp16 apart (what follows in a row -- extension columns no active layer reads, nrldpc.h "Active layers" -- is left as it is).
__global__ __launch_bounds__(256) void nrldpc_expand_i8_rows_kernel(const int8_t* __restrict__ q, __half* __restrict__ out, unsigned n_rows,
                                                                    unsigned act, unsigned pitch, float inv_scale, int vec) {
    const unsigned total = n_rows * act;
    if (vec) { // act % 16 == 0, pitch % 8 == 0, both bases 16-byte aligned: 16 LLRs per thread and trip, never across a row end
        const unsigned nv = total >> 4;
        const uint4* q4 = reinterpret_cast<const uint4*>(q);
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
            const unsigned e = i << 4, r = e / act, c = e - r * act;
            const uint4 v = q4[i];
            uint4 lo, hi;
            lo.x = expand_pair(v.x, 0, inv_scale); lo.y = expand_pair(v.x, 16, inv_scale);
            lo.z = expand_pair(v.y, 0, inv_scale); lo.w = expand_pair(v.y, 16, inv_scale);
            hi.x = expand_pair(v.z, 0, inv_scale); hi.y = expand_pair(v.z, 16, inv_scale);
            hi.z = expand_pair(v.w, 0, inv_scale); hi.w = expand_pair(v.w, 16, inv_scale);
            uint4* o4 = reinterpret_cast<uint4*>(out + (size_t)r * pitch + c);
            o4[0] = lo; o4[1] = hi;
        }
        return;
    }
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / act, c = i - r * act;
        const int a = (int)q[i];
        out[(size_t)r * pitch + c] = a == -128 ? __ushort_as_half((unsigned short)0x7C00u) : __float2half((float)a * inv_scale);
    }
}

hipError_t launch_expand_i8_rows(const int8_t* d_q, void* d_out_f16, size_t n_rows, size_t act, size_t pitch, float inv_scale,
                                 hipStream_t stream) {
    if (n_rows == 0 || act == 0) return hipSuccess;
    if (act == pitch) return launch_expand_i8(d_q, d_out_f16, n_rows * act, inv_scale, stream);
    if (n_rows * act >= ((size_t)1 << 32)) return hipErrorInvalidValue; // chunks are tens of megabytes
    const int vec = (act % 16 == 0) && (pitch % 8 == 0) && ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_out_f16)) & 15) == 0;
    const size_t want = ((vec ? (n_rows * act) >> 4 : n_rows * act) + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : want > 8192 ? 8192 : want);
    hipLaunchKernelGGL(nrldpc_expand_i8_rows_kernel, dim3(grid), dim3(256), 0, stream, d_q, static_cast<__half*>(d_out_f16), (unsigned)n_rows,
                       (unsigned)act, (unsigned)pitch, inv_scale, vec);
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

// ---- NRLDPC_LAYERS_AUTO on device-resident LLRs: highest column block holding anything but +-0 / NaN (nrldpc_hostpath.h)
template <int DT>
__global__ __launch_bounds__(256) void nrldpc_top_block_kernel(const void* __restrict__ llr, int batch, int Z, int nblocks, int first,
                                                               int* __restrict__ best) {
    const int lane = threadIdx.x & 63;
    const long long nitems = (long long)(nblocks - first) * batch; // item = rank * batch + codeword, rank 0 = the top block
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long it = wave0; it < nitems; it += nwaves) {
        const int b = nblocks - 1 - (int)(it / batch);
        const int cw = (int)(it % batch);
        if (b <= __atomic_load_n(best, __ATOMIC_RELAXED)) return; // every later item of this wave is in the same or a lower block
        const size_t base = ((size_t)cw * nblocks + (size_t)b) * (size_t)Z;
        bool hit = false;
        for (int i = lane; i < Z; i += 64) {
            if constexpr (DT == NRLDPC_K_F16) {
                const unsigned a = static_cast<const unsigned short*>(llr)[base + i] & 0x7fffu;
                hit |= (a != 0u) & (a <= 0x7c00u);
            } else {
                const unsigned a = static_cast<const unsigned*>(llr)[base + i] & 0x7fffffffu;
                hit |= (a != 0u) & (a <= 0x7f800000u);
            }
        }
        if (__any((int)hit)) {
            if (lane == 0) atomicMax(best, b);
            return; // anything this wave would look at next is not above b
        }
    }
}

hipError_t launch_top_block(const void* d_llr, int llr_kind, int batch, int Z, int nblocks, int first, int* d_best, hipStream_t stream) {
    if (batch <= 0 || nblocks <= first) return hipSuccess;
    const long long nitems = (long long)(nblocks - first) * batch;
    const long long want = (nitems + 3) / 4;
    const int grid = (int)(want > 4096 ? 4096 : want);
    if (llr_kind == NRLDPC_K_F16)
        hipLaunchKernelGGL(nrldpc_top_block_kernel<NRLDPC_K_F16>, dim3(grid), dim3(256), 0, stream, d_llr, batch, Z, nblocks, first, d_best);
    else
        hipLaunchKernelGGL(nrldpc_top_block_kernel<NRLDPC_K_F32>, dim3(grid), dim3(256), 0, stream, d_llr, batch, Z, nblocks, first, d_best);
    return hipGetLastError();
}

} // namespace nrldpc
