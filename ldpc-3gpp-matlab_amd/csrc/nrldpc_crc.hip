// nrldpc_crc.hip -- CRC attach/check stages on the device.
//
// Replaces comm.CRCDetector at NRLDPCDecoder.m:300 (CB-CRC24B, only when C > 1) and :336 (TB CRC16/24A), and
// the payload copy loops of :303-309 / :330-332: decoded code blocks [n_tb*C][K] (bytes, straight from the
// decoder kernel) -> b_hat [n_tb][B] bytes (a_hat is its first A bytes) + one ok flag per transport block
// (ok = 0 is the reference's a_hat = []).
//
// One wave64 per code block / transport block.  Each lane runs the bit-serial CRC register over its own
// contiguous chunk; chunks are then combined pairwise in a log2(64) tree with
//     crc(A || B) = crc(A) * x^|B| mod g  xor  crc(B),
// the multiplication by x^(chunk * 2^s) being a precomputed 24x24 GF(2) matrix per tree level (host side,
// from the polynomial of get_3gpp_crc_polynomial.m:3-14).  Zero initial state => leading zero padding is free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nrldpc_kernels.h"

namespace nrldpc {

__device__ __forceinline__ uint32_t gf2_apply(const uint32_t* M, uint32_t v, int L) {
    uint32_t o = 0;
    for (int b = 0; b < L; ++b) o ^= ((v >> b) & 1u) ? M[b] : 0u;
    return o;
}

// CRC remainder of `len` bits (one per byte, stride 1) by one wave; every lane returns the result.
__device__ __forceinline__ uint32_t wave_crc(const uint8_t* bits, int len, const CrcPlan& pl) {
    const int lane = threadIdx.x & 63;
    const int chunk = pl.chunk;                       // bits per lane; 64*chunk >= len
    const int pad = 64 * chunk - len;                 // virtual leading zeros
    const uint32_t top = 1u << (pl.L - 1), mask = (1u << pl.L) - 1u;
    uint32_t reg = 0;
    int i0 = lane * chunk - pad;
    for (int i = i0; i < i0 + chunk; ++i) {
        const uint32_t bit = (i >= 0) ? (bits[i] & 1u) : 0u;
        const uint32_t fb = ((reg & top) ? 1u : 0u) ^ bit;
        reg = (reg << 1) & mask;
        if (fb) reg ^= pl.poly & mask;
    }
    // tree: after level s, lanes that are multiples of 2^(s+1) hold the CRC of 2^(s+1) chunks
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const uint32_t right = __shfl_down(reg, 1 << s, 64);
        reg = gf2_apply(pl.shiftmat[s], reg, pl.L) ^ right;
    }
    return __shfl(reg, 0, 64);
}

// One workgroup of 256 threads (4 waves) per transport block: the waves share the C code-block CRCs and the
// payload copy into b_hat (global: transport blocks can exceed LDS), then wave 0 checks the transport block.
__global__ __launch_bounds__(256) void nrldpc_crc_check_kernel(const CrcArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* cb_fail = reinterpret_cast<int*>(lds); // [C]
    const int tb = blockIdx.x;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint8_t* chat = a.c_hat + (size_t)tb * a.C * a.K;
    uint8_t* b_hat = a.b_hat + (size_t)tb * a.B;
    const int pay = a.Kp - a.Lcb; // payload bits per code block
    for (int r = wave; r < a.C; r += nw) {
        int fail = 0;
        if (a.C > 1) fail = wave_crc(chat + (size_t)r * a.K, a.Kp, a.cb) != 0; // NRLDPCDecoder.m:298-301
        if ((threadIdx.x & 63) == 0) cb_fail[r] = fail;
    }
    for (int i = threadIdx.x; i < a.C * pay; i += blockDim.x) { // :303-309 payload copy
        const int r = i / pay, k = i - r * pay;
        b_hat[i] = chat[(size_t)r * a.K + k] & 1u;
    }
    __syncthreads();
    if (wave == 0) {
        const int tb_fail = wave_crc(b_hat, a.B, a.tb) != 0; // :336
        int any_cb = 0;
        for (int r = threadIdx.x & 63; r < a.C; r += 64) any_cb |= cb_fail[r];
        any_cb = __any(any_cb);
        if ((threadIdx.x & 63) == 0) {
            a.ok[tb] = (tb_fail || any_cb) ? 0 : 1; // :337-339
            if (a.cb_pass)
                for (int r = 0; r < a.C; ++r) a.cb_pass[(size_t)tb * a.C + r] = cb_fail[r] ? 0 : 1;
        }
    }
}

// Transmit side (NRLDPCEncoder.m:70-124): transport-block CRC attachment, segmentation into C code blocks,
// CB-CRC24B attachment when C > 1, filler bits (NaN in the reference, encoded as 0, :120-122,153).
// One workgroup per transport block.
__global__ __launch_bounds__(256) void nrldpc_crc_attach_kernel(const CrcAttachArgs a) {
    __shared__ uint32_t tbcrc;
    const int tb = blockIdx.x;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint8_t* src = a.a + (size_t)tb * a.A;
    uint8_t* c = a.c + (size_t)tb * a.C * a.K;
    const int pay = a.Kp - a.Lcb, Ltb = a.B - a.A;
    if (wave == 0) {
        const uint32_t reg = wave_crc(src, a.A, a.tb); // NRLDPCEncoder.m:80-81
        if ((threadIdx.x & 63) == 0) tbcrc = reg;
    }
    __syncthreads();
    const uint32_t reg = tbcrc;
    for (int i = threadIdx.x; i < a.C * a.K; i += blockDim.x) { // :104-122
        const int r = i / a.K, k = i - r * a.K;
        uint8_t bit = 0;
        if (k < pay) {
            const int s = r * pay + k; // position in b = [a; p]
            bit = (s < a.A) ? (src[s] & 1u) : (uint8_t)((reg >> (Ltb - 1 - (s - a.A))) & 1u);
        }
        c[i] = bit; // CB CRC positions and fillers start as 0
    }
    __syncthreads();
    if (a.C > 1) {
        for (int r = wave; r < a.C; r += nw) { // :113-118
            const uint32_t cr = wave_crc(c + (size_t)r * a.K, pay, a.cb);
            const int lane = threadIdx.x & 63;
            if (lane < a.Lcb) c[(size_t)r * a.K + pay + lane] = (uint8_t)((cr >> (a.Lcb - 1 - lane)) & 1u);
        }
    }
}

hipError_t launch_crc_attach(const CrcAttachArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(nrldpc_crc_attach_kernel, dim3(a.n_tb), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_crc_check(const CrcArgs& a, hipStream_t stream) {
    const size_t lds = 4 * (size_t)a.C + 16;
    hipLaunchKernelGGL(nrldpc_crc_check_kernel, dim3(a.n_tb), dim3(256), lds, stream, a);
    return hipGetLastError();
}

} // namespace nrldpc
