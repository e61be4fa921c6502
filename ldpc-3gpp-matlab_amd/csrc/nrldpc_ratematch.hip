// nrldpc_ratematch.hip -- rate recovery on the device.
//
// Fuses the three stages in front of the decoder core -- code-block de-concatenation
// (NRLDPCDecoder.m:143-169), bit de-interleaving (:172-197), circular-buffer bit de-selection with
// soft combining of repetitions and the HARQ buffer (:200-242) -- with the core's input conventions
// (2Z zero prefix :262, filler NaN -> +inf :264) into one gather kernel: raw demodulator LLRs g_tilde in
// HBM -> codeword LLR blocks [n_tb*C][ncols*Z] ready for nrldpc_decode_dev, no host loop in between.
//
// The reference walks the circular buffer bit by bit (O(E) interpreted iterations per block); here each
// thread owns pairs of adjacent output positions p and gathers:  the q-th non-filler position from k_0 receives
// e[q], e[q+P], e[q+2P], ... (P = non-filler positions in the buffer), added in ascending order -- the
// reference's accumulation order -- and e[k] = f[(k mod E/Qm)*Qm + k div (E/Qm)].  HBM-bound gather:
// every g_tilde word is read exactly once (reads of one block interleave Qm streams), every output word
// written once.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <utility>

#include "nrldpc_kernels.h"

namespace nrldpc {

template <class F, int... I> __device__ __forceinline__ void rm_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void rm_static_for(F&& f) { rm_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <typename T> __device__ __forceinline__ T to_out(float v);
template <> __device__ __forceinline__ float to_out<float>(float v) { return v; }
template <> __device__ __forceinline__ __half to_out<__half>(float v) { return __float2half(v); }

constexpr int RR_PAIRS = 4;                    // position pairs per thread
constexpr int RR_TILE = 64 * 2 * RR_PAIRS;     // positions per wave: 4 sweeps of 128 consecutive positions

// Thread mapping: in sweep s a wave covers 128 consecutive positions, lane l the pair (2l, 2l+1) -- a 4-byte
// (fp16) or 8-byte store per lane, 256 / 512 contiguous bytes per wave instruction, and g_tilde reads that
// advance by 2*Qm floats from lane to lane (the two reads of a pair fill the gaps).
template <typename OutT>
__global__ __launch_bounds__(256) void nrldpc_rate_recover_kernel(const RmArgs a) {
    const int blk = blockIdx.y;              // tb * C + r
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int ncwz = 2 * a.Z + a.N;
    const int lane = threadIdx.x & 63;
    const int tile0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RR_TILE;
    if (tile0 >= ncwz) return;
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z; // fillers (:224)
    // fillers that lie inside the circular buffer, and non-filler count before a position
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int F = f_hi > lo_f ? f_hi - lo_f : 0;
    const int P = a.N_cb - F;
    auto nf = [&](int x) { int c = x - lo_f; c = c < 0 ? 0 : (c > F ? F : c); return x - c; };
    const int nfk0 = nf(a.k0);
    const int E = a.E[r];
    const int rows = E > 0 ? E / a.Qm : 1;
    const int Pq = P / rows, Pr = P - Pq * rows;  // a repetition is P positions of e further on
    const float* f = a.g + (size_t)tb * a.G + a.off[r];
    float* hb = a.harq ? a.harq + (size_t)blk * a.N_cb : nullptr;
    OutT* out = static_cast<OutT*>(a.out) + (size_t)blk * ncwz;
    const bool vec = (ncwz % 2) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
#pragma unroll
    for (int s = 0; s < RR_PAIRS; ++s) {
        const int pos0 = tile0 + s * 128 + 2 * lane;
        if (pos0 >= ncwz) break;
        OutT o[2];
        // q: index among the buffer's non-filler positions counted from k_0; e index q = i*rows + j
        int q = -1, j = 0, i = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = pos0 + t - 2 * a.Z;
            float val = 0.0f;
            const bool filler = (p >= lo_f && p < hi_f);
            if (p >= 0 && !filler && p < a.N_cb) {
                if (q < 0) {
                    q = nf(p) - nfk0;
                    if (q < 0) q += P;
                    for (int m = 1; m < a.Qm; ++m) i += (q >= m * rows); // q / rows when q < E = Qm * rows
                    j = q - i * rows;
                }
                if (q < E) {
                    val = f[j * a.Qm + i];
                    int jj = j, ii = i;
                    for (int k = q + P; k < E; k += P) { // soft combining of repetitions, ascending k (:229-231)
                        jj += Pr; ii += Pq;
                        if (jj >= rows) { jj -= rows; ++ii; }
                        val += f[jj * a.Qm + ii];
                    }
                }
                if (hb) { // :236-239
                    val += hb[p];
                    hb[p] = val;
                }
                ++q; ++j;
                if (j == rows) { j = 0; ++i; }
                if (q == P) { q = 0; j = 0; i = 0; }
            }
            // (fillers inside the buffer: the reference keeps NaN there; the position is forced to +inf every time)
            o[t] = to_out<OutT>(filler ? __builtin_inff() : val);
        }
        if (vec) {
            struct alignas(2 * sizeof(OutT)) Pair { OutT x, y; };
            *reinterpret_cast<Pair*>(out + pos0) = Pair{o[0], o[1]};
        } else {
            out[pos0] = o[0];
            if (pos0 + 1 < ncwz) out[pos0 + 1] = o[1];
        }
    }
}

// The same gather for the usual case -- no repetition (E_r <= non-filler positions of the buffer for every code block) --
// with Q_m a template argument and no data-dependent branch: the general kernel above is 1259 instructions per wave and
// 512 positions, most of them divergent control flow around the repetition walk, and runs at the rate it can issue them
// (0.46 of the HBM roofline); this one computes both positions of a pair with selects and clamps the load index of a
// position that receives nothing.  Same values: a position takes exactly one e(k) or none.
template <typename OutT, int QM>
__global__ __launch_bounds__(256) void nrldpc_rate_recover_fast_kernel(const RmArgs a) {
    const int blk = blockIdx.y;
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int ncwz = 2 * a.Z + a.N;
    const int lane = threadIdx.x & 63;
    const int tile0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RR_TILE;
    if (tile0 >= ncwz) return;
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z;
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int F = f_hi > lo_f ? f_hi - lo_f : 0;
    const int P = a.N_cb - F;
    int nfk0 = a.k0 - lo_f;
    nfk0 = a.k0 - (nfk0 < 0 ? 0 : (nfk0 > F ? F : nfk0));
    const int E = a.E[r];
    const int rows = E > 0 ? E / QM : 1;
    const float* f = a.g + (size_t)tb * a.G + a.off[r];
    float* hb = a.harq ? a.harq + (size_t)blk * a.N_cb : nullptr;
    OutT* out = static_cast<OutT*>(a.out) + (size_t)blk * ncwz;
    const bool vec = (ncwz % 2) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
#pragma unroll
    for (int s = 0; s < RR_PAIRS; ++s) {
        const int pos0 = tile0 + s * 128 + 2 * lane;
        if (pos0 >= ncwz) break;
        float v[2];
        bool fill[2], live[2];
        int idx[2], pp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = pos0 + t - 2 * a.Z;
            fill[t] = p >= lo_f && p < hi_f;
            const bool inbuf = p >= 0 && p < a.N_cb && !fill[t];
            int c = p - lo_f;
            c = c < 0 ? 0 : (c > F ? F : c);
            int q = p - c - nfk0;
            q += q < 0 ? P : 0;
            int i = 0;
#pragma unroll
            for (int m = 1; m < QM; ++m) i += (q >= m * rows);
            const int j = q - i * rows;
            live[t] = inbuf && q < E;
            idx[t] = live[t] ? j * QM + i : 0;
            pp[t] = inbuf ? p : -1;
        }
        // predicated: a position that receives nothing must not touch g_tilde at all -- for a trailing code block with
        // E_r == 0 (not retransmitted under CBGTI) f already points one past the transport block's LLRs
        v[0] = live[0] ? f[idx[0]] : 0.0f;
        v[1] = live[1] ? f[idx[1]] : 0.0f;
        OutT o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float val = v[t];
            if (hb && pp[t] >= 0) { // :236-239
                val += hb[pp[t]];
                hb[pp[t]] = val;
            }
            o[t] = to_out<OutT>(fill[t] ? __builtin_inff() : val);
        }
        if (vec) {
            struct alignas(2 * sizeof(OutT)) Pair { OutT x, y; };
            *reinterpret_cast<Pair*>(out + pos0) = Pair{o[0], o[1]};
        } else {
            out[pos0] = o[0];
            if (pos0 + 1 < ncwz) out[pos0 + 1] = o[1];
        }
    }
}

// ---- round 6: the no-repetition case INPUT-driven (serves calls with a HARQ buffer at Q_m <= 2: launch_rate_recover) ----------------
// The gather above reads g_tilde at a stride: the two positions of an output pair are neighbours in e, and e(i*rows + j) = f(j*Qm + i)
// puts neighbours Qm floats apart, so a wave's load instruction uses 1/Qm ... 1/(2 Qm) of every line it touches and the kernel sat at
// 0.53 of the HBM roofline with 79 % of its wave-cycles waiting (profiles/r05_stage_kernels_pmc.txt).  Here a lane owns J consecutive
// interleaver columns j of ALL Qm rows: J*Qm consecutive floats of g_tilde, read as whole 16-byte words (a wave instruction = one
// contiguous kilobyte, every byte used), and scatters them: row i's J values are J neighbouring positions of the circular buffer
// (unless the run crosses its end or the filler gap), i.e. Qm short contiguous stores per lane, each wave-store a contiguous run.
// Positions that receive nothing -- the 2Z punctured columns, fillers (+inf), what lies beyond E or beyond N_cb -- are written by a
// second set of waves of the same launch that only store (with HARQ: echo the buffer).  Same values as the gather: without
// repetition a position takes exactly one e(k) or none, so there is no order of additions to preserve.
template <int QM> struct RrJ { static constexpr int J = QM == 1 ? 4 : (QM == 2 || QM == 6) ? 2 : 1; };
typedef float __attribute__((ext_vector_type(4), aligned(4))) float4_a4; // 16-byte load that only needs dword alignment
constexpr int RR_FILL_TILE = 512;                                          // positions per wave of the fill part: 4 sweeps of 64 pairs

template <typename OutT, int QM>
__global__ __launch_bounds__(256) void nrldpc_rate_recover_scatter_kernel(const RmArgs a, const int in_blocks) {
    constexpr int J = RrJ<QM>::J, NV = J * QM; // floats per lane
    const int blk = blockIdx.y;
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int ncwz = 2 * a.Z + a.N;
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z;
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int F = f_hi > lo_f ? f_hi - lo_f : 0;
    const int P = a.N_cb - F;
    int nfk0 = a.k0 - lo_f;
    nfk0 = a.k0 - (nfk0 < 0 ? 0 : (nfk0 > F ? F : nfk0));
    const int E = a.E[r];
    const int rows = E / QM;
    float* hb = a.harq ? a.harq + (size_t)blk * a.N_cb : nullptr;
    OutT* out = static_cast<OutT*>(a.out) + (size_t)blk * ncwz;
    if ((int)blockIdx.x < in_blocks) {
        // ---- input part: lane -> columns j0 .. j0+J-1
        const int j0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * J;
        if (j0 >= rows) return;
        const float* f = a.g + (size_t)tb * a.G + a.off[r] + (size_t)j0 * QM;
        const int nj = rows - j0 < J ? rows - j0 : J;
        float v[NV];
        if (nj == J) {
            if constexpr (NV % 4 == 0) {
#pragma unroll
                for (int k = 0; k < NV / 4; ++k) {
                    const float4_a4 w = reinterpret_cast<const float4_a4*>(f)[k];
                    v[4 * k] = w.x; v[4 * k + 1] = w.y; v[4 * k + 2] = w.z; v[4 * k + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k] = f[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = k < nj * QM ? f[k] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < QM; ++i) {
            // non-filler index of e(i*rows + j0) counted from 0, then its position in d (fillers skipped), then in the core's input
            int n = i * rows + j0 + nfk0;
            n -= n >= P ? P : 0;
            const int p = n < lo_f ? n : n + F;
            // the J values are neighbours in d unless the run crosses the end of the buffer or the filler gap
            const bool run = nj == J && n + J <= P && (n >= lo_f || n + J <= lo_f);
            if (run) {
                float val[J];
#pragma unroll
                for (int t = 0; t < J; ++t) val[t] = v[t * QM + i];
                if (hb) { // NRLDPCDecoder.m:236-239
#pragma unroll
                    for (int t = 0; t < J; ++t) { val[t] += hb[p + t]; hb[p + t] = val[t]; }
                }
                OutT* o = out + 2 * a.Z + p;
                if constexpr (J == 1) {
                    o[0] = to_out<OutT>(val[0]);
                } else if constexpr (J == 2) {
                    struct alignas(2 * sizeof(OutT)) Pair { OutT x, y; };
                    if ((reinterpret_cast<uintptr_t>(o) & (2 * sizeof(OutT) - 1)) == 0) *reinterpret_cast<Pair*>(o) = Pair{to_out<OutT>(val[0]), to_out<OutT>(val[1])};
                    else { o[0] = to_out<OutT>(val[0]); o[1] = to_out<OutT>(val[1]); }
                } else {
                    struct alignas(4 * sizeof(OutT)) Quad { OutT x, y, z, w; };
                    if ((reinterpret_cast<uintptr_t>(o) & (4 * sizeof(OutT) - 1)) == 0)
                        *reinterpret_cast<Quad*>(o) = Quad{to_out<OutT>(val[0]), to_out<OutT>(val[1]), to_out<OutT>(val[2]), to_out<OutT>(val[3])};
                    else {
#pragma unroll
                        for (int t = 0; t < J; ++t) o[t] = to_out<OutT>(val[t]);
                    }
                }
            } else {
                for (int t = 0; t < nj; ++t) {
                    int nn = n + t;
                    nn -= nn >= P ? P : 0;
                    const int pp = nn < lo_f ? nn : nn + F;
                    float val = v[t * QM + i];
                    if (hb) { val += hb[pp]; hb[pp] = val; }
                    out[2 * a.Z + pp] = to_out<OutT>(val);
                }
            }
        }
        return;
    }
    // ---- fill part: positions no e(k) lands on.  Pairs of neighbouring positions per lane as in the gather kernels, whole tiles of
    // covered positions skipped by a wave-uniform test (at R = 1/3 without fillers only the 2Z punctured columns are left).
    const int wave = ((int)blockIdx.x - in_blocks) * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile0 = wave * RR_FILL_TILE;
    if (tile0 >= ncwz) return;
    // d positions [c0, c1) (cyclically, in non-filler index space [n0, n0 + E)) are covered
    auto covered = [&](int p) -> bool { // p: position in d, 0 <= p < N_cb, not a filler
        int c = p - lo_f;
        c = c < 0 ? 0 : (c > F ? F : c);
        int q = p - c - nfk0;
        q += q < 0 ? P : 0;
        return q < E;
    };
    {
        // wave-uniform skip: the tile lies inside one stretch of d between two boundaries and that stretch is covered
        const int x0 = tile0, x1 = (tile0 + RR_FILL_TILE < ncwz ? tile0 + RR_FILL_TILE : ncwz) - 1; // inclusive
        const int p0 = x0 - 2 * a.Z, p1 = x1 - 2 * a.Z;
        bool plain = p0 >= 0 && p1 < a.N_cb && !(p1 >= lo_f && p0 < hi_f); // no punctured column, nothing beyond the buffer, no filler
        if (plain) {
            // covered positions form a cyclic interval of non-filler indices: both ends covered and the same distance apart in q as in p
            int c0 = p0 - lo_f; c0 = c0 < 0 ? 0 : (c0 > F ? F : c0);
            int c1 = p1 - lo_f; c1 = c1 < 0 ? 0 : (c1 > F ? F : c1);
            int q0 = p0 - c0 - nfk0; q0 += q0 < 0 ? P : 0;
            int q1 = p1 - c1 - nfk0; q1 += q1 < 0 ? P : 0;
            if (q0 < E && q1 < E && q1 - q0 == p1 - p0) return;
        }
    }
    const bool vec = (ncwz % 2) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
#pragma unroll
    for (int s = 0; s < RR_FILL_TILE / 128; ++s) {
        const int pos0 = tile0 + s * 128 + 2 * lane;
        if (pos0 >= ncwz) break;
        bool skip[2];
        OutT o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = pos0 + t - 2 * a.Z;
            const bool filler = p >= lo_f && p < hi_f;
            const bool inbuf = p >= 0 && p < a.N_cb && !filler;
            skip[t] = (inbuf && covered(p)) || pos0 + t >= ncwz;
            float val = 0.0f;
            if (hb && inbuf && !skip[t]) val = hb[p]; // :236-239 with nothing received: the buffer as it is
            o[t] = to_out<OutT>(filler ? __builtin_inff() : val);
        }
        if (vec && !skip[0] && !skip[1]) {
            struct alignas(2 * sizeof(OutT)) Pair { OutT x, y; };
            *reinterpret_cast<Pair*>(out + pos0) = Pair{o[0], o[1]};
        } else {
            if (!skip[0]) out[pos0] = o[0];
            if (!skip[1]) out[pos0 + 1] = o[1];
        }
    }
}

// Transmit side: bit selection + interleaving + concatenation (NRLDPCEncoder.m:168-256) as one gather.
// Output bit x of code block r is f(x) = e(i*E/Qm + j) with i = x mod Qm, j = x div Qm (:219-223), and
// e(k) is the (k mod P)-th non-filler position of the circular buffer counted from k_0 (:186-195).
//
// A thread owns four consecutive j and all Q_m interleaver rows i: Q_m runs of four consecutive e indices, each of which is
// four consecutive bytes of the code block unless it crosses the end of the circular buffer, the filler gap or (with
// repetition) the end of P -- then that run is gathered byte by byte.  So the common case is Q_m (unaligned) dword loads
// and 4*Q_m contiguous output bytes per thread instead of one byte load per output byte (the first version of this
// kernel: 0.23 of the HBM roofline, bound by load instructions, not bytes).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

template <int QM>
__global__ __launch_bounds__(256) void nrldpc_rate_match_kernel(const TxRmArgs a) {
    const int blk = blockIdx.y;
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int E = a.E[r];
    const int rows = E / QM;
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j0 >= rows) return;
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z;
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int F = f_hi > lo_f ? f_hi - lo_f : 0;
    const int P = a.N_cb - F;
    auto nf = [&](int p) { int c = p - lo_f; c = c < 0 ? 0 : (c > F ? F : c); return p - c; };
    const int nfk0 = nf(a.k0);
    const uint8_t* cw = a.cw + (size_t)blk * (2 * a.Z + a.N) + 2 * a.Z;
    const int nj = rows - j0 < 4 ? rows - j0 : 4; // j of this thread that exist
    uint32_t run[QM]; // byte t of run[i] = e(i*rows + j0 + t)
    // No repetition (E <= P: every code block of a launch whose rate is above the mother code's -- the usual case): k < P always,
    // and the index arithmetic is adds and compares.  With repetition k wraps around P: a division and a modulo per run and per
    // gathered byte, ~35 VALU instructions each -- as one code path for both (the first version) the kernel issued 233 VALU
    // instructions per thread for its 8 output bytes and ran at the rate it could issue them, not at the rate of its bytes.
    const bool rep = E > P; // uniform over the workgroup
#pragma unroll
    for (int i = 0; i < QM; ++i) {
        const int k = i * rows + j0;
        int q, kend; // q: position among the buffer's non-filler bits; kend: first k of the next repetition
        if (rep) { const int d = k / P; q = k - d * P + nfk0; kend = P * (d + 1); }
        else { q = k + nfk0; kend = P; }
        if (q >= P) q -= P;
        // four consecutive e indices are four consecutive bytes when the run stays below P in k (no repetition wrap) and
        // in q (no buffer wrap) and on one side of the filler gap
        if (nj == 4 && k + 3 < kend && q + 3 < P && (q >= lo_f || q + 3 < lo_f)) {
            run[i] = *reinterpret_cast<const u32_unaligned*>(cw + (q < lo_f ? q : q + F)) & 0x01010101u;
        } else {
            uint32_t w = 0;
            for (int t = 0; t < nj; ++t) {
                const int kk = k + t;
                int qq = (rep ? kk % P : kk) + nfk0;
                if (qq >= P) qq -= P;
                w |= (uint32_t)(cw[qq < lo_f ? qq : qq + F] & 1u) << (8 * t);
            }
            run[i] = w;
        }
    }
    // interleave: output byte t*QM + i = byte t of run[i].  v_perm_b32 picks each byte of its result from the eight bytes of two
    // registers (selector byte 0..3: the second source's byte, 4..7: the first's), so an output dword is one instruction where its
    // four bytes come from two runs (Q_m <= 2) and three otherwise -- two pairs, then their low halves -- instead of twelve shifts,
    // masks and ors.
    uint32_t out[QM];
    rm_static_for<QM>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        constexpr int t0 = (4 * o) / QM, i0 = (4 * o) % QM, t1 = (4 * o + 1) / QM, i1 = (4 * o + 1) % QM;
        constexpr int t2 = (4 * o + 2) / QM, i2 = (4 * o + 2) % QM, t3 = (4 * o + 3) / QM, i3 = (4 * o + 3) % QM;
        if constexpr (QM == 1) {
            out[o] = run[0];
        } else if constexpr (QM == 2) { // bytes (run0.t0, run1.t1, run0.t2, run1.t3)
            out[o] = __builtin_amdgcn_perm(run[1], run[0], (uint32_t)(t0 | ((4 + t1) << 8) | (t2 << 16) | ((4 + t3) << 24)));
        } else {
            const uint32_t p01 = __builtin_amdgcn_perm(run[i1], run[i0], (uint32_t)(t0 | ((4 + t1) << 8) | 0x0c0c0000u));
            const uint32_t p23 = __builtin_amdgcn_perm(run[i3], run[i2], (uint32_t)(t2 | ((4 + t3) << 8) | 0x0c0c0000u));
            out[o] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        }
    });
    uint8_t* g = a.g + (size_t)tb * a.G + a.off[r] + (size_t)j0 * QM;
    if (nj == 4 && (reinterpret_cast<uintptr_t>(g) & 3) == 0) {
#pragma unroll
        for (int o = 0; o < QM; ++o) reinterpret_cast<uint32_t*>(g)[o] = out[o];
    } else {
        for (int ob = 0; ob < nj * QM; ++ob) g[ob] = (uint8_t)(out[ob >> 2] >> (8 * (ob & 3)));
    }
}

hipError_t launch_rate_match(const TxRmArgs& a, hipStream_t stream) {
    int emax = 0;
    for (int r = 0; r < a.C; ++r) emax = a.E[r] > emax ? a.E[r] : emax;
    if (emax == 0) return hipSuccess;
    const int rows = emax / a.Qm;
    const int per_block = 256 * 4;
    dim3 grid((rows + per_block - 1) / per_block, a.n_tb * a.C);
    switch (a.Qm) {
        case 1: hipLaunchKernelGGL(nrldpc_rate_match_kernel<1>, grid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(nrldpc_rate_match_kernel<2>, grid, dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL(nrldpc_rate_match_kernel<4>, grid, dim3(256), 0, stream, a); break;
        case 6: hipLaunchKernelGGL(nrldpc_rate_match_kernel<6>, grid, dim3(256), 0, stream, a); break;
        case 8: hipLaunchKernelGGL(nrldpc_rate_match_kernel<8>, grid, dim3(256), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename OutT> static void launch_rr_fast(const RmArgs& a, dim3 grid, hipStream_t stream) {
    switch (a.Qm) {
        case 1: hipLaunchKernelGGL((nrldpc_rate_recover_fast_kernel<OutT, 1>), grid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((nrldpc_rate_recover_fast_kernel<OutT, 2>), grid, dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((nrldpc_rate_recover_fast_kernel<OutT, 4>), grid, dim3(256), 0, stream, a); break;
        case 6: hipLaunchKernelGGL((nrldpc_rate_recover_fast_kernel<OutT, 6>), grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((nrldpc_rate_recover_fast_kernel<OutT, 8>), grid, dim3(256), 0, stream, a); break;
    }
}

template <typename OutT> static void launch_rr_scatter(const RmArgs& a, int emax, hipStream_t stream) {
    const int ncwz = 2 * a.Z + a.N;
    const int fill_blocks = (ncwz + 4 * RR_FILL_TILE - 1) / (4 * RR_FILL_TILE);
    auto go = [&](auto qc) {
        constexpr int QM = decltype(qc)::value;
        const int rows = emax / QM;
        const int in_blocks = (rows + 256 * RrJ<QM>::J - 1) / (256 * RrJ<QM>::J);
        hipLaunchKernelGGL((nrldpc_rate_recover_scatter_kernel<OutT, QM>), dim3(in_blocks + fill_blocks, a.n_tb * a.C), dim3(256), 0, stream, a, in_blocks);
    };
    switch (a.Qm) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 6: go(std::integral_constant<int, 6>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
    }
}

hipError_t launch_rate_recover(const RmArgs& a, hipStream_t stream) {
    const int ncwz = 2 * a.Z + a.N;
    const int per_block = 4 * RR_TILE;
    dim3 grid((ncwz + per_block - 1) / per_block, a.n_tb * a.C);
    // non-filler positions of the circular buffer: a code block with more bits than that repeats (soft combining walk)
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z;
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int P = a.N_cb - (f_hi > lo_f ? f_hi - lo_f : 0);
    static const bool force_general = getenv("NRLDPC_RR_GENERAL") != nullptr; // A/B: force the general kernel
    // the input-driven form where it measured faster than the gather -- with the HARQ buffer and Q_m <= 2: -25 % on the headline code;
    // it loses 6-17 % without the buffer, 21 % at 64QAM with it, and 2.4 x on kilobyte-sized blocks (profiles/r06_rate_recover_scatter_ab.txt).
    // NRLDPC_RR_SCATTER=0 / 1 forces the gather / the scatter wherever it applies (A/B)
    static const int env_scatter = getenv("NRLDPC_RR_SCATTER") ? atoi(getenv("NRLDPC_RR_SCATTER")) : -1;
    bool repeats = force_general;
    int emax = 0;
    for (int r = 0; r < a.C; ++r) { repeats = repeats || a.E[r] > P; emax = a.E[r] > emax ? a.E[r] : emax; }
    const bool qm_ok = a.Qm == 1 || a.Qm == 2 || a.Qm == 4 || a.Qm == 6 || a.Qm == 8;
    const bool scatter = env_scatter >= 0 ? env_scatter != 0 : (a.harq != nullptr && a.Qm <= 2 && a.N >= 4096);
    if (!repeats && qm_ok && scatter) {
        if (a.out_f16) launch_rr_scatter<__half>(a, emax, stream);
        else launch_rr_scatter<float>(a, emax, stream);
    } else if (!repeats && qm_ok) {
        if (a.out_f16) launch_rr_fast<__half>(a, grid, stream);
        else launch_rr_fast<float>(a, grid, stream);
    } else if (a.out_f16) {
        hipLaunchKernelGGL(nrldpc_rate_recover_kernel<__half>, grid, dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(nrldpc_rate_recover_kernel<float>, grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace nrldpc
