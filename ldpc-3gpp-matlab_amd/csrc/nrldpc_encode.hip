// nrldpc_encode.hip -- systematic NR LDPC encoder for gfx950.
//
// Replaces step(obj.hLDPCEncoder, c) (NRLDPCEncoder.m:158; comm.LDPCEncoder built at :49 from the H of
// NRLDPC.m:438-440).  The systematic codeword [c; w] with H*[c; w] = 0 is unique, so this kernel is
// bit-identical to the toolbox encoder by construction; tests check H*cw = 0 and equality with the
// oracle's encoder.
//
// One workgroup per codeword, thread z owns row z of every base-graph row.  Systematic and
// core-parity bits sit in LDS as bytes [kb+4][Z]; the 4 core rows are solved through the
// dual-diagonal structure (sum of the four rows isolates p0; the other three blocks follow by
// substitution in an order the host derives from the table), then each extension row's parity is a
// plain XOR of rotated reads.  This is not the hot path: Z-wide byte rotations from LDS, ~300 LDS
// reads per thread.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nrldpc_kernels.h"

namespace nrldpc {

__device__ __forceinline__ int rotz(int z, int p, int Z) {
    int v = z + p;
    return v >= Z ? v - Z : v;
}

__global__ __launch_bounds__(384) void nrldpc_encode_kernel(const EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Z = a.Z, kb = a.kb;
    uint8_t* x = reinterpret_cast<uint8_t*>(lds);          // [kb+4][Z]
    uint8_t* lam = x + (size_t)(kb + 4) * Z;               // [4][Z]
    const int cw = blockIdx.x;
    const uint8_t* info = a.info + (size_t)cw * kb * Z;
    uint8_t* out = a.cw + (size_t)cw * a.ncols * Z;

    for (int i = threadIdx.x; i < kb * Z; i += blockDim.x) {
        const uint8_t b = info[i] & 1;
        x[i] = b;
        out[i] = b;
    }
    __syncthreads();
    for (int z = threadIdx.x; z < Z; z += blockDim.x) {
        uint8_t tot = 0;
        for (int i = 0; i < 4; ++i) {
            uint8_t s = 0;
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e)
                if (a.col[e] < kb) s ^= x[a.col[e] * Z + rotz(z, a.shift[e], Z)];
            lam[i * Z + z] = s;
            tot ^= s;
        }
        x[kb * Z + rotz(z, a.p0_shift, Z)] = tot;
    }
    __syncthreads();
    for (int st = 0; st < 3; ++st) {
        const int i = a.step_row[st], u = a.step_col[st];
        for (int z = threadIdx.x; z < Z; z += blockDim.x) {
            uint8_t s = lam[i * Z + z];
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) {
                const int c = a.col[e] - kb;
                if (c >= 0 && c < 4 && c != u) {
                    bool known = (c == 0);
                    for (int q = 0; q < st; ++q) known |= (a.step_col[q] == c);
                    if (known) s ^= x[a.col[e] * Z + rotz(z, a.shift[e], Z)];
                }
            }
            x[(kb + u) * Z + rotz(z, a.step_shift[st], Z)] = s;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 4 * Z; i += blockDim.x) out[kb * Z + i] = x[kb * Z + i];
    for (int z = threadIdx.x; z < Z; z += blockDim.x)
        for (int i = 4; i < a.nrows; ++i) {
            uint8_t s = 0;
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e)
                if (a.col[e] < kb + 4) s ^= x[a.col[e] * Z + rotz(z, a.shift[e], Z)];
            out[(size_t)(kb + i) * Z + z] = s;
        }
}

hipError_t launch_encode(const EncArgs& a, hipStream_t stream) {
    int threads = ((a.Z + 63) / 64) * 64;
    if (threads > 384) threads = 384;
    const size_t lds = (size_t)(a.kb + 8) * a.Z;
    hipLaunchKernelGGL(nrldpc_encode_kernel, dim3(a.batch), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

} // namespace nrldpc
