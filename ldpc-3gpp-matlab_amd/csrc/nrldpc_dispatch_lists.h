// nrldpc_dispatch_lists.h -- the interleaved block geometry's entries and what each serves (read by the dispatch, nrldpc_decode.hip,
// and mirrored by build.py: Z64I; tests/test_capi_symbols.py checks that the two agree).
#ifndef NRLDPC_DISPATCH_LISTS_H
#define NRLDPC_DISPATCH_LISTS_H
#include "nrldpc_kernels.h"

namespace nrldpc {
// The INTERLEAVED block geometry (z64_ilv, nrldpc_decode_z64.h / nrldpc_decode_z64p.h): (BG, Zr, NCW, mode) =
// NCW codewords of the lifting size Zr in one workgroup of the block geometry of the virtual size Zr * NCW (256, 384, 240, 224,
// 208; 252 = 4 x 63, 220 = 4 x 55, 360, 440, 480: shapes no lifting size has).  mode = what the entry serves, and is compiled for,
// each bit set where it beat the kernel that served the size before (tools/ab_ilv.py, profiles/r04_ilv_ab.txt): 1 fixed
// iteration counts (any layer count), 2 the parity stop with every row active, 4 the parity stop with pruned rows -- a
// workgroup lives until the LAST of its NCW codewords stops, so with the stop most sizes keep their previous kernels.
// Measured and not listed: BG1 88 x 5, 144 x 3, 160 x 2 / x 3; BG2 3, 6, 12, 24, 48, 192 on the 384 shape, 72 x 5, 144 x 3; NCW = 1
// (the lifting size itself in this kernel) for 144 ... 384 except the two entries below.
// Round 5: the parity-stop builds REFILL their codeword slots from the batch (nrldpc_decode_z64p.h), as the packed geometry's do; the
// stop bits below are from the A/B of that round -- every entry with the stop against the kernel (with refill, where it has it) that
// serves the size otherwise, at each size's waterfall, all rows and a pruned count (tools/ab_ilv.py, profiles/r05_ilv_stop_ab.txt):
// a bit is set where the entry took at least 3 % less time.
#define NRLDPC_Z64I_LIST(X) \
    X(1, 2, 128, 5) X(1, 3, 128, 7) X(1, 4, 64, 5) X(1, 5, 48, 7) X(1, 6, 64, 7) X(1, 7, 32, 5) X(1, 8, 32, 5) X(1, 9, 28, 5) X(1, 10, 24, 5) \
    X(1, 11, 20, 5) X(1, 12, 32, 1) X(1, 13, 16, 1) X(1, 14, 16, 1) X(1, 15, 16, 1) X(1, 16, 16, 5) X(1, 18, 14, 5) X(1, 20, 12, 1) X(1, 22, 10, 1) \
    X(1, 24, 16, 1) X(1, 26, 8, 1) X(1, 28, 8, 1) X(1, 30, 8, 1) X(1, 32, 8, 5) X(1, 36, 7, 7) X(1, 40, 6, 5) X(1, 44, 5, 7) X(1, 48, 8, 1) \
    X(1, 52, 4, 5) X(1, 56, 4, 5) X(1, 60, 4, 7) X(1, 64, 4, 7) X(1, 72, 5, 1) X(1, 80, 3, 7) X(1, 96, 4, 7) X(1, 104, 2, 7) X(1, 112, 2, 3) \
    X(1, 120, 2, 3) X(1, 128, 2, 7) X(1, 160, 1, 6) X(1, 192, 2, 1) X(2, 2, 128, 3) X(2, 4, 64, 7) X(2, 5, 48, 7) X(2, 7, 32, 6) X(2, 8, 32, 7) \
    X(2, 9, 28, 7) X(2, 10, 24, 7) X(2, 11, 20, 7) X(2, 13, 16, 5) X(2, 14, 16, 5) X(2, 15, 16, 5) X(2, 16, 16, 5) X(2, 18, 14, 5) X(2, 20, 12, 5) \
    X(2, 22, 10, 5) X(2, 26, 8, 5) X(2, 28, 8, 5) X(2, 30, 8, 5) X(2, 32, 8, 5) X(2, 36, 7, 7) X(2, 40, 6, 5) X(2, 44, 5, 7) X(2, 52, 4, 1) \
    X(2, 56, 4, 1) X(2, 60, 4, 1) X(2, 64, 4, 1) X(2, 80, 3, 7) X(2, 88, 5, 1) X(2, 96, 4, 1) X(2, 104, 2, 1) X(2, 112, 2, 1) X(2, 120, 2, 1) \
    X(2, 128, 2, 1) X(2, 160, 3, 1) X(2, 176, 1, 1)
#define NRLDPC_Z64I_DECL(bg, z, ncw, et) hipError_t launch_decode_z64i_##bg##_##z(const DecArgs& a, hipStream_t stream);
NRLDPC_Z64I_LIST(NRLDPC_Z64I_DECL)
#undef NRLDPC_Z64I_DECL
} // namespace nrldpc
#endif
