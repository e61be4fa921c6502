// nrldpc_host_quant.h -- host-side LLR quantisation for the host-pointer decode path (plain C++, no HIP).
#ifndef NRLDPC_HOST_QUANT_H
#define NRLDPC_HOST_QUANT_H
#include <stddef.h>
#include <stdint.h>

#define NRLDPC_HQ_F32 0
#define NRLDPC_HQ_F16 1
#define NRLDPC_HQ_F64 2

// dst[i] = the decoder kernels' ingest() of src[i] as int8: NaN ? 0 : rint(clamp(float(src[i]) * scale, +-127)), +inf as
// -128 (the kernel knows whether the position is a core column, where +inf means a filler bit, NRLDPCDecoder.m:264).
// The arithmetic is the device's, operation for operation (one f32 multiply, compare-selects, round to nearest even),
// so a batch quantised here decodes bit for bit like the same batch ingested on the device.  Returns true when a -inf
// was met: int8 has no code left for it, and the caller sends that chunk in its own format instead.
bool nrldpc_quantise_i8(int8_t* dst, const void* src, size_t n, int src_kind, float scale);
// the same through one named code path -- 0: plain C++, 1: AVX2 + F16C, 2: AVX-512; -1: the best the CPU has (= the call above) --
// falling back to the next one down when the CPU lacks it (tests compare the paths with each other)
bool nrldpc_quantise_i8_path(int8_t* dst, const void* src, size_t n, int src_kind, float scale, int path);

// Highest block b in [first, nblocks) -- a block = Z consecutive values, a codeword = nblocks blocks -- in which any of the n_cw
// codewords [cw0, cw0 + cw_step, ...) < n_total at src holds a value other than +-0 and NaN; first - 1 when there is none.
// NRLDPC_LAYERS_AUTO's scan (nrldpc.h "Active layers"): from the top block down, per codeword, never below *best + 1 -- `best`
// is shared by the threads that scan one call (relaxed atomic maximum; start it at first - 1).
void nrldpc_top_block(const void* src, int src_kind, size_t n_total, size_t cw0, size_t cw_step, int Z, int nblocks, int first,
                      int* best);

#endif
