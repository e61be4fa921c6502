// nrldpc_ratematch.hip -- rate recovery on the device.
//
// Fuses the three stages in front of the decoder core -- code-block de-concatenation
// (NRLDPCDecoder.m:143-169), bit de-interleaving (:172-197), circular-buffer bit de-selection with
// soft combining of repetitions and the HARQ buffer (:200-242) -- with the core's input conventions
// (2Z zero prefix :262, filler NaN -> +inf :264) into one gather kernel: raw demodulator LLRs g_tilde in
// HBM -> codeword LLR blocks [n_tb*C][ncols*Z] ready for nrldpc_decode_dev, no host loop in between.
//
// The reference walks the circular buffer bit by bit (O(E) interpreted iterations per block); here each
// thread owns one output position p and gathers:  the q-th non-filler position from k_0 receives
// e[q], e[q+P], e[q+2P], ... (P = non-filler positions in the buffer), added in ascending order -- the
// reference's accumulation order -- and e[k] = f[(k mod E/Qm)*Qm + k div (E/Qm)].  HBM-bound gather:
// every g_tilde word is read exactly once (reads of one block interleave Qm streams), every output word
// written once.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "nrldpc_kernels.h"

namespace nrldpc {

__global__ __launch_bounds__(256) void nrldpc_rate_recover_kernel(const RmArgs a) {
    const int blk = blockIdx.y;              // tb * C + r
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x; // position in the core's input, 0 .. 2Z+N-1
    const int ncwz = 2 * a.Z + a.N;
    if (pos >= ncwz) return;
    float val = 0.0f;
    bool filler = false;
    if (pos >= 2 * a.Z) {
        const int p = pos - 2 * a.Z;
        const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z; // fillers (:224)
        filler = (p >= lo_f && p < hi_f);
        if (!filler && p < a.N_cb) {
            // fillers that lie inside the circular buffer, and non-filler count before a position
            const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
            const int F = f_hi > lo_f ? f_hi - lo_f : 0;
            const int P = a.N_cb - F;
            auto nf = [&](int x) { int c = x - lo_f; c = c < 0 ? 0 : (c > F ? F : c); return x - c; };
            int q = nf(p) - nf(a.k0);
            if (q < 0) q += P;
            const int E = a.E[r];
            if (E > 0) {
                const int rows = E / a.Qm;
                const float* f = a.g + (size_t)tb * a.G + a.off[r];
                for (int k = q; k < E; k += P) val += f[(k % rows) * a.Qm + k / rows];
            }
            if (a.harq) {
                float* hb = a.harq + ((size_t)blk) * a.N_cb + p;
                val += *hb;
                *hb = val;
            }
        } else if (filler && p < a.N_cb && a.harq) {
            // the reference stores NaN here; the position is forced to +inf every time, nothing to keep
        }
    }
    const float o = filler ? __builtin_inff() : val;
    const size_t oi = (size_t)blk * ncwz + pos;
    if (a.out_f16) static_cast<__half*>(a.out)[oi] = __float2half(o);
    else static_cast<float*>(a.out)[oi] = o;
}

// Transmit side: bit selection + interleaving + concatenation (NRLDPCEncoder.m:168-256) as one gather.
// Output bit x of code block r is f(x) = e(i*E/Qm + j) with i = x mod Qm, j = x div Qm (:219-223), and
// e(k) is the (k mod P)-th non-filler position of the circular buffer counted from k_0 (:186-195).
__global__ __launch_bounds__(256) void nrldpc_rate_match_kernel(const TxRmArgs a) {
    const int blk = blockIdx.y;
    const int tb = blk / a.C, r = blk - tb * a.C;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = a.E[r];
    if (x >= E) return;
    const int rows = E / a.Qm;
    const int k = (x % a.Qm) * rows + x / a.Qm;
    const int lo_f = a.Kp - 2 * a.Z > 0 ? a.Kp - 2 * a.Z : 0, hi_f = a.K - 2 * a.Z;
    const int f_hi = hi_f < a.N_cb ? hi_f : a.N_cb;
    const int F = f_hi > lo_f ? f_hi - lo_f : 0;
    const int P = a.N_cb - F;
    auto nf = [&](int p) { int c = p - lo_f; c = c < 0 ? 0 : (c > F ? F : c); return p - c; };
    int q = (k % P) + nf(a.k0);
    if (q >= P) q -= P;
    const int pos = (q < lo_f) ? q : q + F; // q-th non-filler position
    const int ncwz = 2 * a.Z + a.N;
    a.g[(size_t)tb * a.G + a.off[r] + x] = a.cw[(size_t)blk * ncwz + 2 * a.Z + pos] & 1u;
}

hipError_t launch_rate_match(const TxRmArgs& a, hipStream_t stream) {
    int emax = 0;
    for (int r = 0; r < a.C; ++r) emax = a.E[r] > emax ? a.E[r] : emax;
    if (emax == 0) return hipSuccess;
    dim3 grid((emax + 255) / 256, a.n_tb * a.C);
    hipLaunchKernelGGL(nrldpc_rate_match_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rate_recover(const RmArgs& a, hipStream_t stream) {
    const int ncwz = 2 * a.Z + a.N;
    dim3 grid((ncwz + 255) / 256, a.n_tb * a.C);
    hipLaunchKernelGGL(nrldpc_rate_recover_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace nrldpc
