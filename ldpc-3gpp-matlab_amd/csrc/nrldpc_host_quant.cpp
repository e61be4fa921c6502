// nrldpc_host_quant.cpp -- see nrldpc_host_quant.h.  Three builds of one conversion, picked at run time: AVX-512 (sixteen LLRs
// per instruction), AVX2 + F16C (eight), and plain C++ for any other host.  All three perform the device's arithmetic operation
// for operation (one f32 multiply, NaN -> 0, clamp, round to nearest even), so they agree with each other and with the kernels'
// ingest() bit for bit (tests/test_capi_symbols.py::test_host_quantiser_paths_agree).
//
// Why: the host-pointer entry point (the MEX gateway's call, NRLDPCDecoder.m:257-266 batched) is bound by the host: a
// MATLAB-double batch of 4096 headline codewords is 855 MB in the caller's array.  The copy threads have to touch every LLR
// anyway on the way into the pinned staging slot; quantising there puts 107 MB on the wire and takes the same work off the
// kernel's prologue.  Round 4: written with intrinsics -- the compiler's own vectorisation of the scalar loop ran at 1.9 (fp16) /
// 1.3 (f32) / 0.66 (f64) G LLRs per second and thread, which made the conversion, not PCIe or the kernel, the longest phase of
// a call (profiles/r04_host_trace.txt).
#include "nrldpc_host_quant.h"

#include <math.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

template <int KIND> struct Src;
template <> struct Src<NRLDPC_HQ_F32> { typedef float T; static inline float get(const T* p, size_t i) { return p[i]; } };
template <> struct Src<NRLDPC_HQ_F64> { typedef double T; static inline float get(const T* p, size_t i) { return (float)p[i]; } };
template <> struct Src<NRLDPC_HQ_F16> {
    typedef _Float16 T;
    static inline float get(const T* p, size_t i) { return (float)p[i]; }
};

// the definition: every other path must produce exactly this
template <int KIND> bool quant_base(int8_t* dst, const void* src, size_t i0, size_t n, float scale) {
    const typename Src<KIND>::T* s = static_cast<const typename Src<KIND>::T*>(src);
    int neg = 0;
    for (size_t i = i0; i < n; ++i) {
        const float x = Src<KIND>::get(s, i);
        float y = x * scale;
        y = (y != y) ? 0.0f : y;
        y = y < -127.0f ? -127.0f : y;
        y = y > 127.0f ? 127.0f : y;
        int q = (int)__builtin_rintf(y);
        q = (x == __builtin_inff()) ? -128 : q;
        neg |= (x == -__builtin_inff());
        dst[i] = (int8_t)q;
    }
    return neg != 0;
}

#if defined(__x86_64__)
// ---- AVX-512: 16 LLRs per step.  cvtps_epi32 rounds by MXCSR (nearest even unless the caller changed it: the scalar path's
// rintf obeys the same register); min / max after the NaN -> 0 select, so their NaN rules never matter.
#define NRLDPC_T512 __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,f16c")))
template <int KIND> NRLDPC_T512 static inline __m512 load16(const void* src, size_t i);
template <> NRLDPC_T512 inline __m512 load16<NRLDPC_HQ_F32>(const void* src, size_t i) { return _mm512_loadu_ps(static_cast<const float*>(src) + i); }
template <> NRLDPC_T512 inline __m512 load16<NRLDPC_HQ_F16>(const void* src, size_t i) {
    return _mm512_cvtph_ps(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(static_cast<const uint16_t*>(src) + i)));
}
template <> NRLDPC_T512 inline __m512 load16<NRLDPC_HQ_F64>(const void* src, size_t i) {
    const double* d = static_cast<const double*>(src) + i;
    const __m256 lo = _mm512_cvtpd_ps(_mm512_loadu_pd(d)), hi = _mm512_cvtpd_ps(_mm512_loadu_pd(d + 8));
    return _mm512_insertf32x8(_mm512_castps256_ps512(lo), hi, 1);
}
// (A software prefetch 1 KB / 4 KB ahead of the reads was measured in round 6, the paths alternated call by call on one array: within
// +-2 % of this loop for doubles, fp16 and singles -- profiles/r06_host_copy_thread_polling.txt.  Not kept.)
template <int KIND> NRLDPC_T512 bool quant_avx512(int8_t* dst, const void* src, size_t n, float scale) {
    const __m512 vs = _mm512_set1_ps(scale), lo = _mm512_set1_ps(-127.0f), hi = _mm512_set1_ps(127.0f), pinf = _mm512_set1_ps(__builtin_inff()),
                 ninf = _mm512_set1_ps(-__builtin_inff());
    const __m512i m128 = _mm512_set1_epi32(-128);
    __mmask16 neg = 0;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512 x = load16<KIND>(src, i);
        __m512 y = _mm512_mul_ps(x, vs);
        y = _mm512_maskz_mov_ps(_mm512_cmp_ps_mask(y, y, _CMP_ORD_Q), y); // NaN -> 0
        y = _mm512_min_ps(_mm512_max_ps(y, lo), hi);
        __m512i q = _mm512_cvtps_epi32(y);
        q = _mm512_mask_mov_epi32(q, _mm512_cmp_ps_mask(x, pinf, _CMP_EQ_OQ), m128);
        neg |= _mm512_cmp_ps_mask(x, ninf, _CMP_EQ_OQ);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm512_cvtepi32_epi8(q));
    }
    return quant_base<KIND>(dst, src, i, n, scale) | (neg != 0);
}

// ---- AVX2 + F16C: 32 LLRs per step (four vectors of eight, packed to bytes with saturating packs: the values are in range)
#define NRLDPC_T256 __attribute__((target("avx2,fma,f16c")))
template <int KIND> NRLDPC_T256 static inline __m256 load8(const void* src, size_t i);
template <> NRLDPC_T256 inline __m256 load8<NRLDPC_HQ_F32>(const void* src, size_t i) { return _mm256_loadu_ps(static_cast<const float*>(src) + i); }
template <> NRLDPC_T256 inline __m256 load8<NRLDPC_HQ_F16>(const void* src, size_t i) {
    return _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i*>(static_cast<const uint16_t*>(src) + i)));
}
template <> NRLDPC_T256 inline __m256 load8<NRLDPC_HQ_F64>(const void* src, size_t i) {
    const double* d = static_cast<const double*>(src) + i;
    return _mm256_set_m128(_mm256_cvtpd_ps(_mm256_loadu_pd(d + 4)), _mm256_cvtpd_ps(_mm256_loadu_pd(d)));
}
template <int KIND> NRLDPC_T256 static inline __m256i quant8(const void* src, size_t i, __m256 vs, __m256& negacc) {
    const __m256 x = load8<KIND>(src, i);
    __m256 y = _mm256_mul_ps(x, vs);
    y = _mm256_and_ps(y, _mm256_cmp_ps(y, y, _CMP_ORD_Q)); // NaN -> 0
    y = _mm256_min_ps(_mm256_max_ps(y, _mm256_set1_ps(-127.0f)), _mm256_set1_ps(127.0f));
    __m256i q = _mm256_cvtps_epi32(y);
    const __m256 isp = _mm256_cmp_ps(x, _mm256_set1_ps(__builtin_inff()), _CMP_EQ_OQ);
    q = _mm256_blendv_epi8(q, _mm256_set1_epi32(-128), _mm256_castps_si256(isp));
    negacc = _mm256_or_ps(negacc, _mm256_cmp_ps(x, _mm256_set1_ps(-__builtin_inff()), _CMP_EQ_OQ));
    return q;
}
template <int KIND> NRLDPC_T256 bool quant_avx2(int8_t* dst, const void* src, size_t n, float scale) {
    const __m256 vs = _mm256_set1_ps(scale);
    __m256 neg = _mm256_setzero_ps();
    const __m256i fix = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7); // undo the 128-bit-lane interleave of the two packs
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i a = quant8<KIND>(src, i, vs, neg), b = quant8<KIND>(src, i + 8, vs, neg), c = quant8<KIND>(src, i + 16, vs, neg),
                      d = quant8<KIND>(src, i + 24, vs, neg);
        const __m256i bytes = _mm256_packs_epi16(_mm256_packs_epi32(a, b), _mm256_packs_epi32(c, d));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_permutevar8x32_epi32(bytes, fix));
    }
    return quant_base<KIND>(dst, src, i, n, scale) | (_mm256_movemask_ps(neg) != 0);
}
#endif

template <int KIND> bool quant(int8_t* dst, const void* src, size_t n, float scale, int path) {
#if defined(__x86_64__)
    static const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
                               __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("f16c");
    static const bool has256 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c") && __builtin_cpu_supports("fma");
    if ((path < 0 || path == 2) && has512) return quant_avx512<KIND>(dst, src, n, scale);
    if ((path < 0 || path == 1) && has256) return quant_avx2<KIND>(dst, src, n, scale);
#endif
    (void)path;
    return quant_base<KIND>(dst, src, 0, n, scale);
}

} // namespace

// ---- NRLDPC_LAYERS_AUTO: "is there anything but +-0 and NaN in this block" on the raw bits (integer compares vectorise for every
// element type: |x| != 0 and |x| <= inf)
namespace {
template <class U> struct Bits;
template <> struct Bits<uint16_t> { static constexpr uint16_t ABS = 0x7fffu, INF = 0x7c00u; };
template <> struct Bits<uint32_t> { static constexpr uint32_t ABS = 0x7fffffffu, INF = 0x7f800000u; };
template <> struct Bits<uint64_t> { static constexpr uint64_t ABS = 0x7fffffffffffffffull, INF = 0x7ff0000000000000ull; };
template <class U> bool any_set(const U* p, size_t n) {
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        unsigned acc = 0;
        for (int k = 0; k < 64; ++k) {
            const U a = (U)(p[i + k] & Bits<U>::ABS);
            acc |= (unsigned)((a != 0) & (a <= Bits<U>::INF));
        }
        if (acc) return true;
    }
    for (; i < n; ++i) {
        const U a = (U)(p[i] & Bits<U>::ABS);
        if (a != 0 && a <= Bits<U>::INF) return true;
    }
    return false;
}
// The scan's bulk is blocks that ARE all zero (what rate matching left untransmitted): those are settled by OR-ing the words
// together with the sign bits masked off -- no compares, one pass at memory speed (the first build compared element by element
// with baseline-x86-64 code: 516 MB of zeros took 9 ms on the MI355X box's host, longer than quantising the whole batch).  Only a
// block whose OR is not zero is looked at value by value (it may hold nothing but NaNs).
template <class U> static inline bool or_is_zero_body(const U* p, size_t n) {
    constexpr uint64_t M = sizeof(U) == 8 ? 0x7fffffffffffffffull : sizeof(U) == 4 ? 0x7fffffff7fffffffull : 0x7fff7fff7fff7fffull;
    const char* b = reinterpret_cast<const char*>(p);
    const size_t bytes = n * sizeof(U);
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        uint64_t w[8];
        memcpy(w, b + i, 64);
        a0 |= w[0]; a1 |= w[1]; a2 |= w[2]; a3 |= w[3]; a4 |= w[4]; a5 |= w[5]; a6 |= w[6]; a7 |= w[7];
    }
    uint64_t acc = (a0 | a1 | a2 | a3 | a4 | a5 | a6 | a7) & M;
    for (; i + sizeof(U) <= bytes; i += sizeof(U)) {
        U v;
        memcpy(&v, b + i, sizeof(U));
        acc |= (uint64_t)(v & Bits<U>::ABS);
    }
    return acc == 0;
}
#if defined(__x86_64__)
template <class U> __attribute__((target("avx2"))) static bool or_is_zero_avx2(const U* p, size_t n) { return or_is_zero_body(p, n); }
#endif
template <class U> static bool or_is_zero(const U* p, size_t n) {
#if defined(__x86_64__)
    static const bool has = __builtin_cpu_supports("avx2");
    if (has) return or_is_zero_avx2(p, n);
#endif
    return or_is_zero_body(p, n);
}

template <class U> void top_block(const U* src, size_t n_total, size_t cw0, size_t cw_step, int Z, int nblocks, int first, int* best) {
    for (size_t cw = cw0; cw < n_total; cw += cw_step) {
        const U* base = src + cw * (size_t)nblocks * (size_t)Z;
        // the usual case first -- every codeword of a call was rate-matched alike, so nothing lies above what an earlier codeword
        // found: one ASCENDING pass over the whole tail (block by block from the top the hardware prefetcher restarts every Z values)
        const int lo = std::max(first, __atomic_load_n(best, __ATOMIC_RELAXED) + 1);
        if (lo >= nblocks) return;
        if (or_is_zero(base + (size_t)lo * Z, (size_t)(nblocks - lo) * (size_t)Z)) continue;
        for (int b = nblocks - 1; b >= first; --b) {
            if (b <= __atomic_load_n(best, __ATOMIC_RELAXED)) break;
            if (!or_is_zero(base + (size_t)b * Z, (size_t)Z) && any_set(base + (size_t)b * Z, (size_t)Z)) {
                int cur = __atomic_load_n(best, __ATOMIC_RELAXED);
                while (cur < b && !__atomic_compare_exchange_n(best, &cur, b, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                break;
            }
        }
        if (__atomic_load_n(best, __ATOMIC_RELAXED) >= nblocks - 1) return;
    }
}
} // namespace

void nrldpc_top_block(const void* src, int src_kind, size_t n_total, size_t cw0, size_t cw_step, int Z, int nblocks, int first,
                      int* best) {
    if (cw_step == 0) cw_step = 1;
    switch (src_kind) {
        case NRLDPC_HQ_F16: top_block(static_cast<const uint16_t*>(src), n_total, cw0, cw_step, Z, nblocks, first, best); break;
        case NRLDPC_HQ_F64: top_block(static_cast<const uint64_t*>(src), n_total, cw0, cw_step, Z, nblocks, first, best); break;
        default: top_block(static_cast<const uint32_t*>(src), n_total, cw0, cw_step, Z, nblocks, first, best); break;
    }
}

bool nrldpc_quantise_i8_path(int8_t* dst, const void* src, size_t n, int src_kind, float scale, int path) {
    switch (src_kind) {
        case NRLDPC_HQ_F16: return quant<NRLDPC_HQ_F16>(dst, src, n, scale, path);
        case NRLDPC_HQ_F64: return quant<NRLDPC_HQ_F64>(dst, src, n, scale, path);
        default: return quant<NRLDPC_HQ_F32>(dst, src, n, scale, path);
    }
}

bool nrldpc_quantise_i8(int8_t* dst, const void* src, size_t n, int src_kind, float scale) {
    // NRLDPC_HOST_QUANT_PATH=0/1/2 names the code path (tests compare them; A/B of the host path); read per call, a call is a chunk
    const char* e = getenv("NRLDPC_HOST_QUANT_PATH");
    return nrldpc_quantise_i8_path(dst, src, n, src_kind, scale, e ? atoi(e) : -1);
}
