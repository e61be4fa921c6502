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
// int8 wire format of rows sent compact: [n_rows][act] int8 -> the first act fp16 LLRs of rows `pitch` apart (nrldpc_expand.hip)
hipError_t launch_expand_i8_rows(const int8_t* d_q, void* d_out_f16, size_t n_rows, size_t act, size_t pitch, float inv_scale,
                                 hipStream_t stream);
// NRLDPC_LAYERS_AUTO for device-resident LLRs (nrldpc.h "Active layers"): *d_best = max(*d_best, highest block b in
// [first, nblocks) in which any of the `batch` codewords holds a value other than +-0 and NaN); a block = Z consecutive values of
// `llr_kind` (NRLDPC_K_F32 / _F16), a codeword = nblocks blocks.  The caller sets *d_best = first - 1 on the same stream before.
// One wave per (block, codeword) pair, top block first; a wave whose block is not above *d_best leaves at once, so the pass reads
// the all-zero column blocks once and almost nothing else.  HBM-bound.
hipError_t launch_top_block(const void* d_llr, int llr_kind, int batch, int Z, int nblocks, int first, int* d_best, hipStream_t stream);
struct DecArgs;
// the shared launch of nrldpc_decode_multi_dev (nrldpc_decode.hip) with the workgroup size given: the configurations of one launch
// share its workgroup size and its dynamic LDS, so the caller groups them by what their schedules ask for (nrldpc_sched.h) and a
// 256-thread configuration is not held to the residency of a 512-thread one.  threads: a multiple of 64, at least every
// configuration's Schedule::threads, at most 768
hipError_t launch_decode_multi_wg(int bg, int llr_kind, const DecArgs* d_tab, const int32_t* d_start, int nb, int grid, int threads,
                                  size_t lds_bytes, hipStream_t stream);
} // namespace nrldpc
#endif
