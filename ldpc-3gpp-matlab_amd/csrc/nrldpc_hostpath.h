// nrldpc_hostpath.h -- device helpers of the host-pointer entry points (nrldpc_expand.hip): the int8 wire format on the way in,
// bit-packed hard decisions on the way out.
#ifndef NRLDPC_HOSTPATH_H
#define NRLDPC_HOSTPATH_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nrldpc {
// hard decisions, one byte per bit ([rows][K], values 0 / 1) -> [rows][ceil(K/8)] bytes, bit k of a row in byte k / 8 at bit
// k % 8 (least significant first; the unused high bits of a row's last byte are zero).  HBM-bound: K bytes in, K/8 out per row.
hipError_t launch_pack_bits(const uint8_t* d_hard, uint8_t* d_packed, int rows, int K, hipStream_t stream);
} // namespace nrldpc
#endif
