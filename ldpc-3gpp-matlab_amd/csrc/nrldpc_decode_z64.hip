// nrldpc_decode_z64.hip -- compile-time-Z specialisation of the layered NMS-Q decoder (Z = 384).
//
// Same algorithm and results as nrldpc_decode.hip (the generic kernel is the reference for this one
// in tests); what changes is where the circulant rotation is paid.  gfx950 issues add/sub/mul/fma and
// and/or/xor at one wave64 instruction per 2 cycles but min/max/med3/cmp/cvt/SDWA -- and any VALU op
// with an SGPR operand -- at one per 4 (tools/ubench/valu_rate*.hip), so the generic kernel's
// three-op ring address (SGPR add, SGPR sub, v_min_u32 = 12 cycles per edge) was its largest single
// cost.  With Z a compile-time multiple of 64 the rotation is free:
//
//   * every base-graph shift is a constant P = 64a + b.  Wave w of a codeword owns rows 64w..64w+63,
//     so it needs ring positions 64((w+a) mod Z/64) + b + lane: a contiguous run of 64 words that
//     crosses at most one 64-word block boundary;
//   * LDS is column-major (unit stride in ring position => no bank conflicts); a column is
//     [64-word guard][Z-word ring][64-word mirror of ring block 0], so a run never wraps;
//   * each thread keeps Z/64 = 6 loop-invariant base addresses R[k] = codeword base + guard +
//     256((w+k) mod 6) + 4 lane; an edge's address is R[a] plus the immediate (col*stride + 4b): zero
//     VALU ops;
//   * coherence of the mirror costs one extra full-wave ds_write in the wave whose run spills into the
//     mirror (twin at -Z words; its low lanes fall into the column's guard, which nobody reads) and in
//     the wave whose run starts in block 0 (twin at +Z words; its high lanes fall into the next
//     column's guard).  Both are wave-uniform branches on compile-time constants; no lane masks.
#include <cstdlib>

#include "nrldpc_device.h"

namespace nrldpc {

constexpr int z64_set_index(int Z) {
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9; ++k)
            if (nr_lifting_sets[s][k] == Z) return s;
    return -1;
}

template <int BG, int ZC, int NCWG_ = 768 / ZC> struct Z64 : BGD<BG> {
    static_assert(ZC % 64 == 0, "specialisation needs whole waves per codeword");
    static constexpr int NWV = ZC / 64;                 // waves per codeword
    static constexpr int GUARD = 256;                   // bytes: 64 never-read words in front of every ring
    static constexpr int CS = GUARD + (ZC + 64) * 4;    // column stride in bytes (guard + ring + mirror)
    static constexpr int CWS = BGD<BG>::NC * CS;        // codeword stride in bytes
    static constexpr int NCWG = NCWG_;                  // codewords per workgroup
    static constexpr int ILS = z64_set_index(ZC);
    static constexpr int shift(int e) { return (BG == 1 ? nr_bg1_shift[ILS][e] : nr_bg2_shift[ILS][e < NR_BG2_NNZ ? e : 0]) % ZC; }
    // + one trailing guard (the last column's block-0 twin write overshoots into it) + termination flags
    static constexpr size_t lds_bytes() { return (size_t)NCWG * CWS + GUARD + 16 * ((NCWG + 1 + 3) / 4); }
};

template <int BG, int ZC, int L>
__device__ __forceinline__ void layer_z64(DecState<BG>& st, char* lds, const uint32_t (&R)[ZC / 64], uint32_t RA,
                                          uint32_t RB, int w, const DecArgs& a, uint32_t& esign_lo,
                                          uint32_t& esign_hi, float* app_ext) {
    using G = Z64<BG, ZC>;
    constexpr int e0 = G::row_ptr(L);
    constexpr int deg = G::row_ptr(L + 1) - e0;
    constexpr bool HAS_EXT = (L >= 4);
    constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    constexpr int ce0 = G::core_base(L);

    float t[ncore];
    float m1 = __builtin_inff(), m2 = __builtin_inff();
    uint32_t S = 0;
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        constexpr int ce = ce0 + j;
        constexpr int P = G::shift(e0 + j);
        constexpr int off = c * G::CS + 4 * (P % 64);
        const float app = *reinterpret_cast<const float*>(lds + R[P / 64] + off);
        const float r = byte_to_f32<ce & 3>(st.rm[ce >> 2]);
        const float tj = app - r;
        t[j] = tj;
        const float aj = fabsf(tj);
        m2 = __builtin_amdgcn_fmed3f(aj, m1, m2);
        m1 = fminf(m1, aj);
        S ^= fbits(tj);
    });
    float lam = 0.0f;
    if constexpr (HAS_EXT) {
        lam = byte_to_f32<(L - 4) & 3>(st.xq[(L - 4) >> 2]);
        const float al = fabsf(lam);
        m2 = __builtin_amdgcn_fmed3f(al, m1, m2);
        m1 = fminf(m1, al);
        S ^= fbits(lam);
    }
    const uint32_t Sm = S & 0x80000000u;
    const float M1 = __uint_as_float(fbits(fminf(rintf(a.alpha * m1), 127.0f)) | Sm);
    const float M2 = __uint_as_float(fbits(fminf(rintf(a.alpha * m2), 127.0f)) | Sm);
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        constexpr int ce = ce0 + j;
        constexpr int P = G::shift(e0 + j);
        constexpr int ka = P / 64, kb = P % 64;
        constexpr int off = c * G::CS + 4 * kb;
        const float tj = t[j];
        const float mag = (fabsf(tj) == m1) ? M2 : M1;
        const float r = __uint_as_float(fbits(mag) ^ (fbits(tj) & 0x80000000u));
        f32_to_byte<ce & 3>(st.rm[ce >> 2], r);
        const float v = tj + r;
        t[j] = v; // kept for the mirror pass below
        *reinterpret_cast<float*>(lds + R[ka] + off) = v;
    });
    // Mirror coherence (ring block 0 lives at ring words [0,64) and again at [ZC, ZC+64)), grouped by the
    // wave that owes the twin write so that each wave takes at most one taken branch per layer:
    //   wave (NWV-1-ka): its run started in the last block and ran into the mirror -> twin at RA + off
    //   wave (NWV-ka)  : its run started in block 0                                -> twin at RB + off
    static_for<G::NWV>([&](auto wc) {
        constexpr int wv = decltype(wc)::value;
        if (w == wv) {
            static_for<ncore>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int c = G::col(e0 + j);
                constexpr int P = G::shift(e0 + j);
                constexpr int ka = P / 64, kb = P % 64;
                constexpr int off = c * G::CS + 4 * kb;
                if constexpr (kb != 0 && wv == (2 * G::NWV - 1 - ka) % G::NWV)
                    *reinterpret_cast<float*>(lds + RA + off) = t[j];
                if constexpr (wv == (G::NWV - ka) % G::NWV)
                    *reinterpret_cast<float*>(lds + RB + off) = t[j];
            });
        }
    });
    if constexpr (HAS_EXT) {
        if (a.need_ext) {
            const float mag = (fabsf(lam) == m1) ? M2 : M1;
            const float r = __uint_as_float(fbits(mag) ^ (fbits(lam) & 0x80000000u));
            const float ae = lam + r;
            if constexpr (L - 4 < 32) esign_lo |= (fbits(ae) >> 31) << (L - 4);
            else esign_hi |= (fbits(ae) >> 31) << (L - 36);
            if (app_ext) {
                float* p = app_ext;
                asm volatile("" : "+v"(p));
                p[(size_t)(G::NC + L - 4) * ZC] = ae * a.inv_scale;
            }
        }
    }
}

template <int BG, int ZC, int L>
__device__ __forceinline__ uint32_t row_parity_z64(char* lds, const uint32_t (&R)[ZC / 64], uint32_t esign_lo,
                                                   uint32_t esign_hi) {
    using G = Z64<BG, ZC>;
    constexpr int e0 = G::row_ptr(L);
    constexpr int deg = G::row_ptr(L + 1) - e0;
    constexpr bool HAS_EXT = (L >= 4);
    constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    uint32_t p = 0;
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        constexpr int P = G::shift(e0 + j);
        p ^= fbits(*reinterpret_cast<const float*>(lds + R[P / 64] + c * G::CS + 4 * (P % 64)));
    });
    p >>= 31;
    if constexpr (HAS_EXT) p ^= (L - 4 < 32 ? esign_lo >> ((L - 4) & 31) : esign_hi >> ((L - 36) & 31)) & 1u;
    return p;
}

template <int BG, int ZC, int NCWG>
__global__ __launch_bounds__(NCWG * ZC, 4) void nrldpc_decode_z64_kernel(const DecArgs a) {
    using G = Z64<BG, ZC, NCWG>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cwl = wave / G::NWV, w = wave % G::NWV, lane = tid & 63;
    const int z = w * 64 + lane;
    const int cw = blockIdx.x * G::NCWG + cwl;
    const bool active = cw < a.batch; // wave-uniform: whole waves belong to one codeword
    const uint32_t cwbase = (uint32_t)cwl * (uint32_t)G::CWS;
    int* flags = reinterpret_cast<int*>(lds + (size_t)G::NCWG * G::CWS + G::GUARD);
    constexpr size_t ncwz = (size_t)G::COLS * ZC;

    uint32_t R[G::NWV];
#pragma unroll
    for (int k = 0; k < G::NWV; ++k)
        R[k] = cwbase + G::GUARD + 256u * (uint32_t)((w + k) % G::NWV) + 4u * (uint32_t)lane;
    // twin addresses: a run in the last block mirrors to ring word (kb+lane-64) => column base + 4(kb+lane);
    // a run in block 0 mirrors to ring word ZC+kb+lane
    const uint32_t RA = cwbase + 4u * (uint32_t)lane;
    const uint32_t RB = cwbase + G::GUARD + 4u * ZC + 4u * (uint32_t)lane;

    DecState<BG> st;
#pragma unroll
    for (int i = 0; i < G::NW; ++i) st.rm[i] = 0;
#pragma unroll
    for (int i = 0; i < G::NXW; ++i) st.xq[i] = 0;
    uint32_t esign_lo = 0, esign_hi = 0;
    float* app_row = nullptr;

    if (active) {
        const size_t base = (size_t)cw * ncwz;
        if (a.app) app_row = a.app + base + z;
        const bool f16 = a.llr_kind == NRLDPC_K_F16;
        char* home = lds + cwbase + G::GUARD + 4 * z; // ring position z of column 0
        {
            float x[G::NC];
            static_for<G::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const size_t i = base + (size_t)c * ZC + z;
                x[c] = f16 ? load_llr<NRLDPC_K_F16>(a.llr, i) : load_llr<NRLDPC_K_F32>(a.llr, i);
            });
            static_for<G::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const float q = ingest(x[c], a.scale, true);
                *reinterpret_cast<float*>(home + c * G::CS) = q;
                if (w == 0) *reinterpret_cast<float*>(home + c * G::CS + ZC * 4) = q; // mirror of block 0
            });
        }
        {
            float x[G::NEXT];
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const size_t gi = base + (size_t)(G::NC + i) * ZC + z;
                x[i] = f16 ? load_llr<NRLDPC_K_F16>(a.llr, gi) : load_llr<NRLDPC_K_F32>(a.llr, gi);
            });
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                f32_to_byte<i & 3>(st.xq[i >> 2], ingest(x[i], a.scale, false));
            });
        }
        if (app_row) {
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                app_row[(size_t)(G::NC + i) * ZC] = byte_to_f32<i & 3>(st.xq[i >> 2]) * a.inv_scale;
            });
        }
    }
    __syncthreads();

    bool done = !active;
    int my_iters = a.max_iter;
    for (int it = 1; it <= a.max_iter; ++it) {
        if (!done) { esign_lo = 0; esign_hi = 0; }
        static_for<G::ROWS>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            if (L < launder(a.n_layers)) {
                if (!done) layer_z64<BG, ZC, L>(st, lds, R, RA, RB, w, a, esign_lo, esign_hi, app_row);
            }
            if constexpr (LayerGroups<BG>::group_end(L)) { // see LayerGroups: one barrier per column-disjoint group
                constexpr int gs = LayerGroups<BG>::group_start(L); // forced compile-time evaluation
                if (gs < launder(a.n_layers)) __syncthreads();
            }
        });
        if (a.early_term) {
            if (tid <= G::NCWG) flags[tid] = 0;
            __syncthreads();
            if (!done) {
                uint32_t bad = 0;
                static_for<G::ROWS>([&](auto lc) {
                    constexpr int L = decltype(lc)::value;
                    if (L < launder(a.n_layers)) bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                });
                if (bad) { flags[cwl] = 1; flags[G::NCWG] = 1; }
            }
            __syncthreads();
            if (!done && flags[cwl] == 0) { done = true; my_iters = it; }
            if (flags[G::NCWG] == 0) break;
        }
    }

    if (active) {
        if (a.iters && z == 0) a.iters[cw] = my_iters;
        uint8_t* hard = a.hard + (size_t)cw * ((size_t)G::KB * ZC);
        const char* home = lds + cwbase + G::GUARD + 4 * z;
        static_for<G::NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const float val = *reinterpret_cast<const float*>(home + c * G::CS);
            if (c < G::KB) hard[(size_t)c * ZC + z] = val < 0.0f ? 1 : 0;
            if (app_row) app_row[(size_t)c * ZC] = val * a.inv_scale;
        });
    }
}

template <int BG, int ZC, int NCWG> static hipError_t launch_z64(const DecArgs& a, hipStream_t s) {
    using G = Z64<BG, ZC, NCWG>;
    auto k = nrldpc_decode_z64_kernel<BG, ZC, NCWG>;
    constexpr size_t lds = G::lds_bytes();
    static_assert(lds <= 160 * 1024, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = (a.batch + G::NCWG - 1) / G::NCWG;
    hipLaunchKernelGGL(k, dim3(grid), dim3(G::NCWG * ZC), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_decode_z384(int bg, const DecArgs& a, hipStream_t stream) {
    // Codewords per workgroup: 2 (12 waves, three per SIMD, one workgroup per CU) measured faster than 1
    // (two independent 6-wave workgroups per CU).  NRLDPC_Z384_NCWG=1 selects the latter for experiments.
    static const int ncwg = [] { const char* e = getenv("NRLDPC_Z384_NCWG"); return (e && e[0] == '1') ? 1 : 2; }();
    if (ncwg == 1) return bg == 1 ? launch_z64<1, 384, 1>(a, stream) : launch_z64<2, 384, 1>(a, stream);
    return bg == 1 ? launch_z64<1, 384, 2>(a, stream) : launch_z64<2, 384, 2>(a, stream);
}

} // namespace nrldpc
