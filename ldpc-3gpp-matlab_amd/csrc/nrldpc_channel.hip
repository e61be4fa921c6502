// nrldpc_channel.hip -- modulation + AWGN + exact-LLR demodulation fused into one kernel.
//
// Replaces, in the Monte-Carlo loop of plot_BLER_vs_SNR.m:130-132,
//     tx = step(hMod, g);  rx = step(hChan, tx);  g_tilde = step(hDemod, rx);
// i.e. NRModulator.m:73-81 (TS 38.211 Gray maps through the reference's custom symbol tables, unit average power),
// comm.AWGNChannel at Es/N0 (plot_BLER_vs_SNR.m:50,105) and NRDemodulator.m:76-84 ('Log-likelihood ratio' = exact
// LLRs with Variance = N0 = 10^(-EsN0/10), :106).  The symbols never exist in memory: a thread owns one symbol, reads its
// Q_m bits (1 byte each, the boundary format of the rate-matching stage), draws its noise from a counter-based
// generator (Philox-4x32-10, counter = global symbol index, key = seed: reproducible for any launch geometry, and
// restated in numpy by the test oracle), and writes Q_m LLRs (f32, positive = bit 0).  HBM-bound: 1 byte in,
// 4 bytes out per bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nrldpc_kernels.h"

namespace nrldpc {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// amplitude of one I/Q rail from its NB bits, most significant (sign) first: TS 38.211 5.1.3-5.1.5,
// 16QAM (1-2b0)(2-(1-2b2)), 64QAM (1-2b0)(4-(1-2b2)(2-(1-2b4))), 256QAM one level more
template <int NB> __device__ __forceinline__ float pam_level(uint32_t code) {
    float x = 1.0f;
#pragma unroll
    for (int j = 1; j < NB; ++j) {
        const uint32_t b = (code >> (j - 1)) & 1u; // innermost (last) bit first
        x = (float)(1 << j) - (b ? -x : x);
    }
    return ((code >> (NB - 1)) & 1u) ? -x : x;
}

template <int NB> __device__ __forceinline__ void rail_llr(float y, float inv_n0, float inv_norm, float (&llr)[NB]) {
    if constexpr (NB == 1) { // two points +-p: log-sum-exp of one term each, ((y+p)^2 - (y-p)^2)/N0 = 4 p y / N0
        llr[0] = 4.0f * inv_norm * y * inv_n0;
        return;
    }
    float mx[NB][2], sm[NB][2];
#pragma unroll
    for (int k = 0; k < NB; ++k) { mx[k][0] = mx[k][1] = -3.0e38f; sm[k][0] = sm[k][1] = 0.0f; }
    float met[1 << NB];
#pragma unroll
    for (uint32_t c = 0; c < (1u << NB); ++c) {
        const float d = y - pam_level<NB>(c) * inv_norm;
        met[c] = -d * d * inv_n0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int bit = (c >> (NB - 1 - k)) & 1u;
            mx[k][bit] = fmaxf(mx[k][bit], met[c]);
        }
    }
#pragma unroll
    for (uint32_t c = 0; c < (1u << NB); ++c)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int bit = (c >> (NB - 1 - k)) & 1u;
            sm[k][bit] += __builtin_amdgcn_exp2f((met[c] - mx[k][bit]) * 1.4426950408889634f); // e^x = 2^(x log2 e): v_exp_f32
        }
#pragma unroll
    for (int k = 0; k < NB; ++k) // (sums lie in [1, 2^NB]: v_log_f32 needs no denormal care)
        llr[k] = (mx[k][0] - mx[k][1]) + 0.6931471805599453f * (__builtin_amdgcn_logf(sm[k][0]) - __builtin_amdgcn_logf(sm[k][1]));
}

// LLRs of one symbol from its bits and the two uniform words of its noise sample
template <int QM> __device__ __forceinline__ void symbol_llr(const ChanArgs& a, const uint8_t* g, uint32_t w1, uint32_t w2, float* o) {
    // Box-Muller on 24-bit uniforms in (0,1): exact in f32
    const float u1 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(w2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    // The hardware's own transcendentals: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, so sin(2 pi u2) is one
    // instruction on u2 itself -- no range reduction (the library sincosf carries a Payne-Hanek path for arguments it never gets
    // here); v_log_f32 is log2, v_sqrt_f32 is within 1 ulp.  The kernel issues VALU instructions, not bytes (438 per thread before,
    // most of them here and in Philox): it is bound by that, not by HBM.  Results move by ~1e-6 relative; the test's tolerance on an
    // LLR is 5e-4 (tests/test_chain_gpu.py).
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * a.sigma; // -2 ln u1 = -2 ln2 log2 u1; sigma = sqrt(N0/2) per rail
    const float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);
    const float ni = rad * cs, nq = rad * sn;
    if constexpr (QM == 1) { // comm.PSKModulator order 2, phase offset pi/4 (NRModulator.m:73): LLR = 4 Re(rx e^{-j pi/4}) / N0
        const float tx = (g[0] & 1u) ? -1.0f : 1.0f;
        const float y = tx + (ni + nq) * 0.70710678118654752f; // noise projected onto the signalling axis
        o[0] = 4.0f * y * a.inv_n0;
    } else {
        constexpr int NB = QM / 2;
        uint32_t wi = 0, wq = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { wi = (wi << 1) | (g[2 * k] & 1u); wq = (wq << 1) | (g[2 * k + 1] & 1u); }
        const float yi = pam_level<NB>(wi) * a.inv_norm + ni, yq = pam_level<NB>(wq) * a.inv_norm + nq;
        float li[NB], lq[NB];
        rail_llr<NB>(yi, a.inv_n0, a.inv_norm, li);
        rail_llr<NB>(yq, a.inv_n0, a.inv_norm, lq);
#pragma unroll
        for (int k = 0; k < NB; ++k) { o[2 * k] = li[k]; o[2 * k + 1] = lq[k]; }
    }
}

// A thread owns the pair of symbols (2c, 2c+1) of the global symbol count: one Philox call -- counter c, four words -- feeds
// both (words 0,1 the even symbol, words 2,3 the odd one); a generator call per symbol threw half of its output away and was
// what the kernel spent most of its instructions on.  Pairs are aligned to the global count, so the noise of a symbol does not
// depend on how the symbols are split over launches.
template <int QM> __global__ __launch_bounds__(256) void nrldpc_awgn_llr_kernel(const ChanArgs a) {
    const uint64_t c = (a.first_symbol >> 1) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // pair index = Philox counter
    const int64_t s0 = (int64_t)(2 * c - a.first_symbol);                                        // local index of the even symbol
    if (s0 >= a.n_sym) return;
    uint32_t r[4];
    philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 0u, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
    const bool v0 = s0 >= 0, v1 = s0 + 1 < a.n_sym;
    float o[2 * QM];
    uint8_t bits[2 * QM];
    const uint8_t* g = a.g + s0 * QM;
    float* dst = a.llr + s0 * QM;
    const bool both = v0 && v1;
    if (QM == 2 && both && (reinterpret_cast<uintptr_t>(g) & 3) == 0) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(g);
#pragma unroll
        for (int k = 0; k < 4; ++k) bits[k < 2 * QM ? k : 0] = (uint8_t)(w >> (8 * k));
    } else {
#pragma unroll
        for (int k = 0; k < 2 * QM; ++k) bits[k] = ((k < QM ? v0 : v1) ? g[k] : (uint8_t)0);
    }
    symbol_llr<QM>(a, bits, r[0], r[1], o);
    symbol_llr<QM>(a, bits + QM, r[2], r[3], o + QM);
    if (QM == 2 && both && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[QM < 2 ? 0 : 2], o[QM < 2 ? 0 : 3]);
    } else {
#pragma unroll
        for (int k = 0; k < QM; ++k) {
            if (v0) dst[k] = o[k];
            if (v1) dst[QM + k] = o[QM + k];
        }
    }
}

// ---- payload bits of the Monte-Carlo loop (plot_BLER_vs_SNR.m:118: a = round(rand(A,1)) per block) ---------------------------------
// Bit i of transport block b = bit (i mod 64) of splitmix64(seed + (b*W + i div 64 + 1) * golden), W = ceil(A/64): a function of the
// GLOBAL block index alone (harness.payload_bits_np is the definition), so any split of a batch over devices draws the same payloads.
// One kernel instead of the fifteen small tensor operations the harness made of it (round 6: a fifth of a demo-sized step).
// A thread writes four consecutive bits of a block as four bytes (64 % 4 == 0: they never straddle a hash word).
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void nrldpc_payload_bits_kernel(uint64_t seed, uint64_t first_block, int32_t n_tb, int32_t A, uint8_t* out) {
    const int q4 = (A + 3) >> 2; // four-bit chunks per block
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_tb * q4) return;
    const int b = (int)(t / q4), c = (int)(t - (int64_t)b * q4);
    const int i0 = 4 * c, W = (A + 63) >> 6;
    const uint64_t h = splitmix64(seed + (((first_block + (uint64_t)b) * (uint64_t)W + (uint64_t)(i0 >> 6)) + 1ull) * 0x9E3779B97F4A7C15ull);
    const uint32_t nib = (uint32_t)(h >> (i0 & 63)) & 0xFu;
    uint8_t* o = out + (size_t)b * A + i0;
    if (i0 + 4 <= A && (reinterpret_cast<uintptr_t>(o) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(o) = (nib * 0x00204081u) & 0x01010101u; // bit k of the nibble -> byte k
    } else {
        for (int k = 0; k < 4 && i0 + k < A; ++k) o[k] = (uint8_t)((nib >> k) & 1u);
    }
}
hipError_t launch_payload_bits(uint64_t seed, uint64_t first_block, int32_t n_tb, int32_t A, uint8_t* out, hipStream_t stream) {
    const int64_t n = (int64_t)n_tb * ((A + 3) >> 2);
    hipLaunchKernelGGL(nrldpc_payload_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seed, first_block, n_tb, A, out);
    return hipGetLastError();
}

hipError_t launch_awgn_llr(const ChanArgs& a, hipStream_t stream) {
    // pairs of the global symbol count that the launch touches
    const uint64_t npairs = ((a.first_symbol + (uint64_t)a.n_sym - 1) >> 1) - (a.first_symbol >> 1) + 1;
    const dim3 grid((unsigned)((npairs + 255) / 256)), block(256);
    switch (a.Qm) {
        case 1: hipLaunchKernelGGL(nrldpc_awgn_llr_kernel<1>, grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL(nrldpc_awgn_llr_kernel<2>, grid, block, 0, stream, a); break;
        case 4: hipLaunchKernelGGL(nrldpc_awgn_llr_kernel<4>, grid, block, 0, stream, a); break;
        case 6: hipLaunchKernelGGL(nrldpc_awgn_llr_kernel<6>, grid, block, 0, stream, a); break;
        case 8: hipLaunchKernelGGL(nrldpc_awgn_llr_kernel<8>, grid, block, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace nrldpc
