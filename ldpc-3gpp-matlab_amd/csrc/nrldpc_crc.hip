// nrldpc_crc.hip -- CRC attach/check stages on the device.
//
// Replaces comm.CRCDetector at NRLDPCDecoder.m:300 (CB-CRC24B, only when C > 1) and :336 (TB CRC16/24A), and
// the payload copy loops of :303-309 / :330-332: decoded code blocks [n_tb*C][K] (bytes, straight from the
// decoder kernel) -> b_hat [n_tb][B] bytes (a_hat is its first A bytes) + one ok flag per transport block
// (ok = 0 is the reference's a_hat = []).
//
// One workgroup per transport block, one wave64 per code block (waves loop when C > 4).  A wave stages its
// code block into LDS with coalesced 16-byte loads (every HBM byte is read once and written once; the
// stages are HBM-bound copies), then:
//   * each lane runs the bit-serial CRC register over its own contiguous chunk of the LDS copy, and the 64
//     chunk remainders are combined pairwise in a log2(64) tree with
//         crc(A || B) = crc(A) * x^|B| mod g  xor  crc(B),
//     the multiplication by x^(chunk * 2^s) being a precomputed 24x24 GF(2) matrix per tree level (host side,
//     from the polynomial of get_3gpp_crc_polynomial.m:3-14).  Zero initial state => leading zeros are free;
//   * the transport-block CRC is never computed over the concatenated payload: every wave also takes the
//     TB-polynomial remainder of its own payload segment, and the C segment remainders are folded by Horner's
//     rule with the x^(segment length) matrix -- the same identity, applied across code blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nrldpc_kernels.h"
#include "nrldpc_wave.h"

namespace nrldpc {

// v * M over GF(2), M = 24 columns (zero beyond the CRC length) in kernel-argument memory.  Every lane calls
// this (wave-uniform control flow): lane b fetches column b once and the 24 columns are then broadcast with
// v_readlane, so no scalar-load latency sits inside the loop.
__device__ __forceinline__ uint32_t gf2_apply(const uint32_t* M, uint32_t v) {
    const int lane = threadIdx.x & 63;
    const uint32_t mine = M[lane < 24 ? lane : 0];
    uint32_t o = 0;
#pragma unroll
    for (int b = 0; b < 24; ++b) o ^= (0u - ((v >> b) & 1u)) & (uint32_t)__builtin_amdgcn_readlane((int)mine, b);
    return o;
}

// CRC remainder of `len` bits (one per byte, in LDS) by one wave; every lane returns the result.
// pl.chunk bits per lane with 64 * chunk >= len; the shortfall acts as leading zeros.
__device__ __forceinline__ uint32_t wave_crc(const uint8_t* bits, int len, const CrcPlan& pl) {
    const int lane = threadIdx.x & 63;
    const int chunk = pl.chunk;
    const int pad = 64 * chunk - len;
    const uint32_t top = 1u << (pl.L - 1), mask = (1u << pl.L) - 1u, poly = pl.poly & mask;
    uint32_t reg = 0;
    const int i0 = lane * chunk - pad;
#pragma unroll 4
    for (int i = i0 < 0 ? 0 : i0; i < i0 + chunk; ++i) {
        const uint32_t fb = ((reg & top) ? 1u : 0u) ^ (bits[i] & 1u);
        reg = (reg << 1) & mask;
        reg ^= fb ? poly : 0u;
    }
    // tree: after level s, lanes that are multiples of 2^(s+1) hold the CRC of 2^(s+1) chunks
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const uint32_t right = __shfl_down(reg, 1 << s, 64);
        reg = gf2_apply(pl.shiftmat[s], reg) ^ right;
    }
    return __shfl(reg, 0, 64);
}

__host__ __device__ __forceinline__ int row_capacity(int K) { return ((K + 15) & ~15) + 48; }

__global__ __launch_bounds__(256) void nrldpc_crc_check_kernel(const CrcArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    const int cap = row_capacity(a.K);
    uint8_t* base = reinterpret_cast<uint8_t*>(lds) + (size_t)wave * cap;
    int* cb_fail = reinterpret_cast<int*>(lds + (size_t)nw * cap);  // [C]
    uint32_t* part = reinterpret_cast<uint32_t*>(cb_fail + a.C);    // [C] TB-polynomial remainder per segment
    const int tb = blockIdx.x;
    const uint8_t* chat = a.c_hat + (size_t)tb * a.C * a.K;
    uint8_t* b_hat = a.b_hat + (size_t)tb * a.B;
    const int pay = a.Kp - a.Lcb; // payload bits per code block
    for (int r = wave; r < a.C; r += nw) {
        wave_lds_sync(); // previous round's readers are done with the row
        const uint8_t* row = stage_row(base, chat + (size_t)r * a.K, a.Kp);
        wave_lds_sync();
        int fail = 0;
        if (a.C > 1) fail = wave_crc(row, a.Kp, a.cb) != 0;  // NRLDPCDecoder.m:298-301
        const bool take = !fail && a.cbgti[r] != 0;          // :304: CRC holds and the block was (re)transmitted
        uint32_t p;
        if (take) {
            p = wave_crc(row, pay, a.tb);
            store_row(b_hat + (size_t)r * pay, row, pay);    // :303-309 payload copy
        } else if (a.keep_b_hat) {                           // :286-287: the segment keeps what an earlier step stored
            wave_lds_sync();
            row = stage_row(base, b_hat + (size_t)r * pay, pay);
            wave_lds_sync();
            p = wave_crc(row, pay, a.tb);
        } else {                                             // :289: b_hat = zeros(B,1)
            p = 0;
            uint8_t* z = b_hat + (size_t)r * pay;
            for (int i = lane; i < pay; i += 64) z[i] = 0;
        }
        if (lane == 0) { cb_fail[r] = take ? 0 : 1; part[r] = p; }
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t reg = 0;                                    // :336 over b_hat = segment 0 || ... || segment C-1
        for (int r = 0; r < a.C; ++r) reg = gf2_apply(a.tb.horner, reg) ^ part[r];
        int any_cb = 0;                                      // :337 any(~code_block_CRC_passed), flags sticky (:305,315)
        for (int r = lane; r < a.C; r += 64) {
            int pass = cb_fail[r] ? 0 : 1;
            if (a.cb_pass) {
                int32_t* f = a.cb_pass + (size_t)tb * a.C + r;
                if (a.sticky) pass |= (*f != 0);
                *f = pass;
            }
            any_cb |= !pass;
        }
        any_cb = __any(any_cb);
        if (lane == 0) a.ok[tb] = (reg != 0 || any_cb) ? 0 : 1; // :337-339
    }
}

// Transmit side (NRLDPCEncoder.m:70-124): transport-block CRC attachment, segmentation into C code blocks,
// CB-CRC24B attachment when C > 1, filler bits (NaN in the reference, encoded as 0, :120-122,153).
// Same structure: one wave per code block; the last code block (which carries the transport-block CRC bits)
// is finished after the segment remainders of all code blocks have been folded.
__global__ __launch_bounds__(256) void nrldpc_crc_attach_kernel(const CrcAttachArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    const int cap = row_capacity(a.K);
    uint8_t* base = reinterpret_cast<uint8_t*>(lds) + (size_t)wave * cap;
    uint8_t* row = base;
    uint32_t* part = reinterpret_cast<uint32_t*>(lds + (size_t)nw * cap); // [C]
    const int tb = blockIdx.x;
    const uint8_t* src = a.a + (size_t)tb * a.A;
    uint8_t* c = a.c + (size_t)tb * a.C * a.K;
    const int pay = a.Kp - a.Lcb, Ltb = a.B - a.A;
    auto finish = [&](int r) { // payload complete in row[0..pay): CB CRC (:113-118), fillers (:120-122), store
        if (a.C > 1) {
            const uint32_t cr = wave_crc(row, pay, a.cb);
            if (lane < a.Lcb) row[pay + lane] = (uint8_t)((cr >> (a.Lcb - 1 - lane)) & 1u);
        }
        for (int i = a.Kp + lane; i < a.K; i += 64) row[i] = 0;
        wave_lds_sync();
        store_row(c + (size_t)r * a.K, row, a.K);
    };
    for (int r = wave; r < a.C; r += nw) {
        const bool last = (r == a.C - 1);
        const int seg = last ? pay - Ltb : pay; // bits of `a` in this code block (:104-112)
        wave_lds_sync();
        row = stage_row(base, src + (size_t)r * pay, seg);
        wave_lds_sync();
        const uint32_t p = wave_crc(row, seg, a.tb);
        if (lane == 0) part[r] = p;
        if (!last) finish(r);
    }
    __syncthreads();
    if (wave == (a.C - 1) % nw) { // this wave's row still holds the last code block's share of `a`
        uint32_t reg = 0;         // NRLDPCEncoder.m:80-81 over a = segment 0 || ... || tail
        for (int r = 0; r + 1 < a.C; ++r) reg = gf2_apply(a.tb.horner, reg) ^ part[r];
        reg = gf2_apply(a.tb.horner_tail, reg) ^ part[a.C - 1];
        if (lane < Ltb) row[pay - Ltb + lane] = (uint8_t)((reg >> (Ltb - 1 - lane)) & 1u);
        wave_lds_sync();
        finish(a.C - 1);
    }
}

// ---- short transport blocks (C == 1, K' <= 512): one LANE per transport block --------------------------------------------
// A wave per code block spends 64 lanes, an LDS staging round trip and a six-level combine tree on the reference's
// demo-sized blocks (BASELINE configs[0]: 116 bits) and leaves the chip almost idle (0.02 of the HBM roofline at 65536
// transport blocks).  Here a lane walks its own block bit by bit straight from global memory (4 bytes per load where the
// rows are dword-aligned), 64 transport blocks per wave; same state machine as the wave kernels above.
constexpr int CRC_LANE_MAX_BITS = 512;

__device__ __forceinline__ uint32_t crc_step(uint32_t reg, uint32_t bit, uint32_t top, uint32_t mask, uint32_t poly) {
    const uint32_t fb = ((reg & top) ? 1u : 0u) ^ (bit & 1u);
    reg = (reg << 1) & mask;
    return reg ^ (fb ? poly : 0u);
}

__global__ __launch_bounds__(256) void nrldpc_crc_check_lane_kernel(const CrcArgs a) {
    const int tb = blockIdx.x * blockDim.x + threadIdx.x;
    if (tb >= a.n_tb) return;
    const uint8_t* row = a.c_hat + (size_t)tb * a.K;
    uint8_t* b_hat = a.b_hat + (size_t)tb * a.B;
    const int pay = a.Kp; // C == 1: no code-block CRC, the payload is B = K' bits (NRLDPCDecoder.m:298-309)
    const uint32_t top = 1u << (a.tb.L - 1), mask = (1u << a.tb.L) - 1u, poly = a.tb.poly & mask;
    const bool take = a.cbgti[0] != 0; // :304
    const bool wide = ((a.K | a.B) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.c_hat) | reinterpret_cast<uintptr_t>(a.b_hat)) & 3) == 0;
    uint32_t reg = 0;
    if (take || a.keep_b_hat) {
        const uint8_t* src = take ? row : b_hat; // :286-287: an untransmitted block keeps what an earlier step stored
        if (wide) {
            int i = 0;
            for (; i + 4 <= pay; i += 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(src + i) & 0x01010101u;
                reg = crc_step(reg, w, top, mask, poly);
                reg = crc_step(reg, w >> 8, top, mask, poly);
                reg = crc_step(reg, w >> 16, top, mask, poly);
                reg = crc_step(reg, w >> 24, top, mask, poly);
                if (take) *reinterpret_cast<uint32_t*>(b_hat + i) = w; // :303-309 payload copy
            }
            for (; i < pay; ++i) {
                const uint8_t v = src[i] & 1u;
                reg = crc_step(reg, v, top, mask, poly);
                if (take) b_hat[i] = v;
            }
        } else {
            for (int i = 0; i < pay; ++i) {
                const uint8_t v = src[i] & 1u;
                reg = crc_step(reg, v, top, mask, poly);
                if (take) b_hat[i] = v;
            }
        }
    } else { // :289: b_hat = zeros(B,1)
        for (int i = 0; i < pay; ++i) b_hat[i] = 0;
    }
    int pass = take ? 1 : 0; // :305 (C == 1: the code-block CRC cannot fail)
    if (a.cb_pass) {
        int32_t* f = a.cb_pass + tb;
        if (a.sticky) pass |= (*f != 0);
        *f = pass;
    }
    a.ok[tb] = (reg != 0 || !pass) ? 0 : 1; // :337-339
}

__global__ __launch_bounds__(256) void nrldpc_crc_attach_lane_kernel(const CrcAttachArgs a) {
    const int tb = blockIdx.x * blockDim.x + threadIdx.x;
    if (tb >= a.n_tb) return;
    const uint8_t* src = a.a + (size_t)tb * a.A;
    uint8_t* c = a.c + (size_t)tb * a.K;
    const int Ltb = a.B - a.A;
    const uint32_t top = 1u << (a.tb.L - 1), mask = (1u << a.tb.L) - 1u, poly = a.tb.poly & mask;
    uint32_t reg = 0;
    const bool wide = ((a.A | a.K | Ltb) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.a) | reinterpret_cast<uintptr_t>(a.c)) & 3) == 0;
    if (wide) { // four bits per load and store (rows are dword-aligned when A, L and K are multiples of 4)
        for (int i = 0; i < a.A; i += 4) { // NRLDPCEncoder.m:70-82
            const uint32_t w = *reinterpret_cast<const uint32_t*>(src + i) & 0x01010101u;
            reg = crc_step(reg, w, top, mask, poly);
            reg = crc_step(reg, w >> 8, top, mask, poly);
            reg = crc_step(reg, w >> 16, top, mask, poly);
            reg = crc_step(reg, w >> 24, top, mask, poly);
            *reinterpret_cast<uint32_t*>(c + i) = w;
        }
        for (int i = 0; i < Ltb; i += 4) {
            const uint32_t w = ((reg >> (Ltb - 1 - i)) & 1u) | (((reg >> (Ltb - 2 - i)) & 1u) << 8) |
                               (((reg >> (Ltb - 3 - i)) & 1u) << 16) | (((reg >> (Ltb - 4 - i)) & 1u) << 24);
            *reinterpret_cast<uint32_t*>(c + a.A + i) = w;
        }
        for (int i = a.Kp; i < a.K; i += 4) *reinterpret_cast<uint32_t*>(c + i) = 0u; // fillers (:120-122,153)
        return;
    }
    for (int i = 0; i < a.A; ++i) { // NRLDPCEncoder.m:70-82
        const uint8_t v = src[i] & 1u;
        reg = crc_step(reg, v, top, mask, poly);
        c[i] = v;
    }
    for (int i = 0; i < Ltb; ++i) c[a.A + i] = (uint8_t)((reg >> (Ltb - 1 - i)) & 1u);
    for (int i = a.Kp; i < a.K; ++i) c[i] = 0; // fillers (:120-122,153)
}

static int waves_for(int C) { return C < 4 ? C : 4; }

hipError_t launch_crc_attach(const CrcAttachArgs& a, hipStream_t stream) {
    // (the lane kernels are written for the one-code-block layout of TS 38.212: no code-block CRC, B = K')
    if (a.C == 1 && a.Lcb == 0 && a.B == a.Kp && a.Kp <= CRC_LANE_MAX_BITS && a.n_tb >= 4096) { // short blocks, enough of them to fill the chip lane-wise
        hipLaunchKernelGGL(nrldpc_crc_attach_lane_kernel, dim3((a.n_tb + 255) / 256), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    const int nw = waves_for(a.C);
    const size_t lds = (size_t)nw * row_capacity(a.K) + 4 * (size_t)a.C + 16;
    hipLaunchKernelGGL(nrldpc_crc_attach_kernel, dim3(a.n_tb), dim3(64 * nw), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_crc_check(const CrcArgs& a, hipStream_t stream) {
    if (a.C == 1 && a.Lcb == 0 && a.B == a.Kp && a.Kp <= CRC_LANE_MAX_BITS && a.n_tb >= 4096) {
        hipLaunchKernelGGL(nrldpc_crc_check_lane_kernel, dim3((a.n_tb + 255) / 256), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    const int nw = waves_for(a.C);
    const size_t lds = (size_t)nw * row_capacity(a.K) + 8 * (size_t)a.C + 16;
    hipLaunchKernelGGL(nrldpc_crc_check_kernel, dim3(a.n_tb), dim3(64 * nw), lds, stream, a);
    return hipGetLastError();
}

} // namespace nrldpc
