// nrldpc_decode_z64_inst.hip -- one (BG, Z) instantiation of the compile-time-Z decoder (nrldpc_decode_z64.h).
// build.py compiles this file once per supported pair, in parallel:
//     hipcc -c -DNRLDPC_Z64_BG=1 -DNRLDPC_Z64_Z=384 nrldpc_decode_z64_inst.hip -o z64_1_384.o
#ifndef NRLDPC_Z64_BG
#define NRLDPC_Z64_BG 1
#endif
#ifndef NRLDPC_Z64_Z
#define NRLDPC_Z64_Z 384
#endif
#include "nrldpc_decode_z64.h"

#define NRLDPC_CAT_(a, b, c) a##b##_##c
#define NRLDPC_CAT(a, b, c) NRLDPC_CAT_(a, b, c)

namespace nrldpc {
#ifdef NRLDPC_Z64_NL
// -DNRLDPC_Z64_NL=<count>: the pipelined kernels of one pruned layer count (NRLDPC_Z64_NL_LIST)
#define NRLDPC_CAT4_(a, b, c, d) a##b##_##c##_nl##d
#define NRLDPC_CAT4(a, b, c, d) NRLDPC_CAT4_(a, b, c, d)
hipError_t NRLDPC_CAT4(launch_decode_z64_, NRLDPC_Z64_BG, NRLDPC_Z64_Z, NRLDPC_Z64_NL)(const DecArgs& a, hipStream_t stream) {
    return launch_z64_pruned<NRLDPC_Z64_BG, NRLDPC_Z64_Z, z64_ncwg_nl<NRLDPC_Z64_BG, NRLDPC_Z64_Z, NRLDPC_Z64_NL>(), NRLDPC_Z64_NL>(a, stream);
}
#else
hipError_t NRLDPC_CAT(launch_decode_z64_, NRLDPC_Z64_BG, NRLDPC_Z64_Z)(const DecArgs& a, hipStream_t stream) {
    return launch_z64<NRLDPC_Z64_BG, NRLDPC_Z64_Z, z64_ncwg<NRLDPC_Z64_BG, NRLDPC_Z64_Z>()>(a, stream);
}
#endif
} // namespace nrldpc
