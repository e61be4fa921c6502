// nrldpc_host_quant.cpp -- see nrldpc_host_quant.h.  One loop, built twice: for AVX2 + F16C (eight LLRs per
// instruction; picked at run time when the CPU has them) and for the baseline x86-64 / any other host.
//
// Why: the host-pointer entry point (the MEX gateway's call, NRLDPCDecoder.m:257-266 batched) is PCIe-bound -- a
// MATLAB-double batch of 4096 headline codewords is 855 MB on the host and was 428 MB on the wire as f32.  The copy
// threads have to touch every LLR anyway on the way into the pinned staging slot; quantising there puts 107 MB on
// the wire and takes the same work off the kernel's prologue.
#include "nrldpc_host_quant.h"

#include <math.h>
#include <string.h>

namespace {

template <int KIND> struct Src;
template <> struct Src<NRLDPC_HQ_F32> { typedef float T; static inline float get(const T* p, size_t i) { return p[i]; } };
template <> struct Src<NRLDPC_HQ_F64> { typedef double T; static inline float get(const T* p, size_t i) { return (float)p[i]; } };
template <> struct Src<NRLDPC_HQ_F16> {
    typedef _Float16 T;
    static inline float get(const T* p, size_t i) { return (float)p[i]; }
};

#define NRLDPC_QUANT_BODY(KIND)                                                        \
    const typename Src<KIND>::T* s = static_cast<const typename Src<KIND>::T*>(src);   \
    int neg = 0;                                                                       \
    for (size_t i = 0; i < n; ++i) {                                                   \
        const float x = Src<KIND>::get(s, i);                                          \
        float y = x * scale;                                                           \
        y = (y != y) ? 0.0f : y;                                                       \
        y = y < -127.0f ? -127.0f : y;                                                 \
        y = y > 127.0f ? 127.0f : y;                                                   \
        int q = (int)__builtin_rintf(y);                                               \
        q = (x == __builtin_inff()) ? -128 : q;                                        \
        neg |= (x == -__builtin_inff());                                               \
        dst[i] = (int8_t)q;                                                            \
    }                                                                                  \
    return neg != 0;

template <int KIND> bool quant_base(int8_t* dst, const void* src, size_t n, float scale) { NRLDPC_QUANT_BODY(KIND) }

#if defined(__x86_64__)
template <int KIND> __attribute__((target("avx2,fma,f16c"))) bool quant_avx2(int8_t* dst, const void* src, size_t n, float scale) {
    NRLDPC_QUANT_BODY(KIND)
}
#endif

template <int KIND> bool quant(int8_t* dst, const void* src, size_t n, float scale) {
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c") && __builtin_cpu_supports("fma");
    if (wide) return quant_avx2<KIND>(dst, src, n, scale);
#endif
    return quant_base<KIND>(dst, src, n, scale);
}

} // namespace

bool nrldpc_quantise_i8(int8_t* dst, const void* src, size_t n, int src_kind, float scale) {
    switch (src_kind) {
        case NRLDPC_HQ_F16: return quant<NRLDPC_HQ_F16>(dst, src, n, scale);
        case NRLDPC_HQ_F64: return quant<NRLDPC_HQ_F64>(dst, src, n, scale);
        default: return quant<NRLDPC_HQ_F32>(dst, src, n, scale);
    }
}
