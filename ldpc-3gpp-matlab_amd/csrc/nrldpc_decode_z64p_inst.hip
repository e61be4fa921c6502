// nrldpc_decode_z64p_inst.hip -- one (BG, Z) instantiation of the packed-geometry decoder (nrldpc_decode_z64p.h).
// build.py compiles this file once per pair of NRLDPC_Z64P_LIST:
//     hipcc -c -DNRLDPC_Z64_BG=2 -DNRLDPC_Z64_Z=16 nrldpc_decode_z64p_inst.hip -o z64p_2_16.o
#ifndef NRLDPC_Z64_BG
#define NRLDPC_Z64_BG 1
#endif
#ifndef NRLDPC_Z64_Z
#define NRLDPC_Z64_Z 32
#endif
#ifndef NRLDPC_Z64_ILV
#define NRLDPC_Z64_PACK 1 // the geometry of this unit (z64_packed, nrldpc_decode_z64.h)
#endif
// -DNRLDPC_Z64_ILV=<NCW> -DNRLDPC_Z64_Z=<Zr * NCW> -DNRLDPC_Z64I_ZR=<Zr> -DNRLDPC_Z64S_DUAL=0: the interleaved BLOCK geometry instead
// (z64_ilv): NCW codewords of the lifting size Zr in one workgroup of the virtual size's block geometry (NRLDPC_Z64I_LIST)
#include "nrldpc_decode_z64p.h"

#define NRLDPC_CAT_(a, b, c) a##b##_##c
#define NRLDPC_CAT(a, b, c) NRLDPC_CAT_(a, b, c)

namespace nrldpc {
#ifdef NRLDPC_Z64_ILV
#ifndef NRLDPC_Z64I_MODE
#define NRLDPC_Z64I_MODE 7
#endif
// only what the list entry's mode serves is instantiated (1 fixed iteration counts, 2 parity stop with every row active, 4 parity
// stop with pruned rows: NRLDPC_Z64I_LIST); launch_decode asks for nothing else
hipError_t NRLDPC_CAT(launch_decode_z64i_, NRLDPC_Z64_BG, NRLDPC_Z64I_ZR)(const DecArgs& a, hipStream_t stream) {
    constexpr int BG = NRLDPC_Z64_BG, ZC = NRLDPC_Z64_Z, MODE = NRLDPC_Z64I_MODE;
    const bool pruned = a.n_layers != BGT<BG>::ROWS; // any other layer count: the run-time-prefix builds (NL_RT)
    if (!a.early_term) {
        if constexpr ((MODE & 1) != 0) return pruned ? launch_z64p_t<BG, ZC, false, NL_RT>(a, stream) : launch_z64p_t<BG, ZC, false>(a, stream);
    } else if (!pruned) {
        if constexpr ((MODE & 2) != 0) return launch_z64p_t<BG, ZC, true>(a, stream);
    } else {
        if constexpr ((MODE & 4) != 0) return launch_z64p_t<BG, ZC, true, NL_RT>(a, stream);
    }
    return hipErrorInvalidValue;
}
#elif defined(NRLDPC_Z64P_ROW)
// -DNRLDPC_Z64P_ROW=1 (with -DNRLDPC_Z64P_RW=<waves>): the pipelined one-thread-per-row builds of this (BG, Z) only (NRLDPC_Z64PR_LIST)
hipError_t NRLDPC_CAT(launch_decode_z64pr_, NRLDPC_Z64_BG, NRLDPC_Z64_Z)(const DecArgs& a, hipStream_t stream) {
    return launch_z64pr<NRLDPC_Z64_BG, NRLDPC_Z64_Z>(a, stream);
}
#elif defined(NRLDPC_Z64_NL)
// -DNRLDPC_Z64_NL=<count>: the builds of one pruned layer count (NRLDPC_Z64P_NL_LIST)
#define NRLDPC_CAT4_(a, b, c, d) a##b##_##c##_nl##d
#define NRLDPC_CAT4(a, b, c, d) NRLDPC_CAT4_(a, b, c, d)
hipError_t NRLDPC_CAT4(launch_decode_z64p_, NRLDPC_Z64_BG, NRLDPC_Z64_Z, NRLDPC_Z64_NL)(const DecArgs& a, hipStream_t stream) {
    return launch_z64p_pruned<NRLDPC_Z64_BG, NRLDPC_Z64_Z, NRLDPC_Z64_NL>(a, stream);
}
#else
hipError_t NRLDPC_CAT(launch_decode_z64p_, NRLDPC_Z64_BG, NRLDPC_Z64_Z)(const DecArgs& a, hipStream_t stream) {
    return launch_z64p<NRLDPC_Z64_BG, NRLDPC_Z64_Z>(a, stream);
}
// any layer count, soft output: the general kernel of this geometry -- BG2 only (z64pg_serves; no BG1 kernel is instantiated)
hipError_t NRLDPC_CAT(launch_decode_z64pg_, NRLDPC_Z64_BG, NRLDPC_Z64_Z)(const DecArgs& a, hipStream_t stream) {
    if constexpr (z64pg_serves<NRLDPC_Z64_BG>()) return launch_z64pg<NRLDPC_Z64_BG, NRLDPC_Z64_Z>(a, stream);
    else return hipErrorInvalidValue; // not reached: launch_decode asks first
}
#endif
} // namespace nrldpc
