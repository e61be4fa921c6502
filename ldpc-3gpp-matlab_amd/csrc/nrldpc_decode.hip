// nrldpc_decode.hip -- layered normalised-min-sum NR LDPC decoder for gfx950 (MI355X / CDNA4).
//
// Replaces the inner loop of the reference's decode path: step(obj.hLDPCDecoder, cw_tilde) at
// NRLDPCDecoder.m:265 (object constructed at :120).  Algorithm "NMS-Q" as restated in
// oracle/nrldpc_oracle.c (orc_decode_nmsq); the two must agree bit for bit.
//
// Mapping (CDNA4-first, not a streaming design):
//   * one workgroup owns `ncw` whole codewords for ALL iterations; thread (cwl, z) owns check row z of
//     every base-graph layer of codeword cwl (Z = 384 -> 6 wave64 per codeword, ncw = 1);
//   * a-posteriori LLRs of the kb+4 core columns live in LDS for the whole decode, ring-position
//     major with an odd dword stride (bank-conflict free for ds_read_b32/ds_write_b32 at unit
//     position stride); check row z reads and writes ring position (z + P) mod Z of each of its
//     columns: three VALU ops per edge (add, add, v_min_u32 -- the unsigned-underflow trick), the
//     column offset rides in the ds instruction's immediate, and every LDS word is touched by exactly
//     one thread per layer, so one barrier per layer suffices;
//   * check-to-variable messages live in VGPRs as int8 (4 per register, SDWA byte convert/insert);
//   * channel LLRs of the degree-1 extension-parity columns never enter LDS: such a column is only
//     ever touched by its own row at ring position z, so it is thread-private (int8 in VGPRs);
//   * all values are integers carried in fp32 (exact below 2^24), so abs/neg modifiers, v_med3_f32
//     and v_min_f32 do the min-sum in 8 VALU ops per edge for pass 1 and 7 for pass 2.
// HBM sees each codeword once on the way in (ncols*Z LLRs) and K hard bits on the way out.
#include <cstdlib>

#include "nrldpc_device.h"
#include "nrldpc_dispatch_lists.h"
#include "nrldpc_hostpath.h"

namespace nrldpc {

// One base-graph layer for this thread's check row.  With ncw codewords per workgroup the LDS ring of a
// column has ncw*Z slots, slot u = z*ncw + cwl; rotating z by P is rotating u by P*ncw, so the same
// unsigned-min wrap works on zb = u * (NCP*4) with rot[e] = P_e * sbw and ring size Z*sbw bytes.
template <int BG, int L>
__device__ __forceinline__ void layer(DecState<BG>& st, char* lds, uint32_t zb, const DecArgs& a, ctab_t rot,
                                      uint32_t& esign_lo, uint32_t& esign_hi, float* app_ext) {
    using G = BGD<BG>;
    constexpr int e0 = G::row_ptr(L);
    constexpr int deg = G::row_ptr(L + 1) - e0;
    constexpr bool HAS_EXT = (L >= 4);
    constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    constexpr int ce0 = G::core_base(L);
    const uint32_t zsb = (uint32_t)a.Z * (uint32_t)a.sbw;

    float t[ncore];
    uint32_t ad[ncore];
    float m1 = __builtin_inff(), m2 = __builtin_inff();
    uint32_t S = 0;
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        constexpr int ce = ce0 + j;
        const uint32_t w1 = zb + (uint32_t)rot[e0 + j];
        const uint32_t w2 = w1 - zsb; // wraps to a huge value unless w1 >= Z*sbw
        const uint32_t wa = min(w1, w2);
        ad[j] = wa;
        const float app = *reinterpret_cast<const float*>(lds + wa + 4 * c);
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
    // magnitudes carrying the row's sign parity; the edge's own sign is xor-ed in per edge
    const uint32_t Sm = S & 0x80000000u;
    const float M1 = __uint_as_float(fbits(scale_mag(a, m1)) | Sm);
    const float M2 = __uint_as_float(fbits(scale_mag(a, m2)) | Sm);
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        constexpr int ce = ce0 + j;
        const float tj = t[j];
        const float mag = (fabsf(tj) == m1) ? M2 : M1;
        const float r = __uint_as_float(fbits(mag) ^ (fbits(tj) & 0x80000000u));
        f32_to_byte<ce & 3>(st.rm[ce >> 2], r);
        *reinterpret_cast<float*>(lds + ad[j] + 4 * c) = tj + r;
    });
    if constexpr (HAS_EXT) {
        if (a.need_ext) { // early termination and/or soft output: a-posteriori value of the extension bit
            const float mag = (fabsf(lam) == m1) ? M2 : M1;
            const float r = __uint_as_float(fbits(mag) ^ (fbits(lam) & 0x80000000u));
            const float ae = lam + r;
            // esign is cleared at the start of every iteration, so a plain OR rebuilds bit L-4
            if constexpr (L - 4 < 32) esign_lo |= (fbits(ae) >> 31) << (L - 4);
            else esign_hi |= (fbits(ae) >> 31) << (L - 36);
            if (app_ext) { // soft output is a test/debug path: keep its address arithmetic out of the hot loop
                float* p = app_ext;
                asm volatile("" : "+v"(p));
                p[(size_t)(G::NC + L - 4) * launder(a.Z)] = ae * a.inv_scale;
            }
        }
    }
}

// parity of check row (L, z) on the iteration-end snapshot
template <int BG, int L>
__device__ __forceinline__ uint32_t row_parity(char* lds, uint32_t zb, const DecArgs& a, ctab_t rot, uint32_t esign_lo,
                                               uint32_t esign_hi) {
    using G = BGD<BG>;
    constexpr int e0 = G::row_ptr(L);
    constexpr int deg = G::row_ptr(L + 1) - e0;
    constexpr bool HAS_EXT = (L >= 4);
    constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    const uint32_t zsb = (uint32_t)a.Z * (uint32_t)a.sbw;
    uint32_t p = 0;
    static_for<ncore>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int c = G::col(e0 + j);
        const uint32_t w1 = zb + (uint32_t)rot[e0 + j];
        const uint32_t w2 = w1 - zsb;
        const uint32_t ra = min(w1, w2);
        p ^= fbits(*reinterpret_cast<const float*>(lds + ra + 4 * c));
    });
    p >>= 31;
    if constexpr (HAS_EXT) p ^= (L - 4 < 32 ? esign_lo >> ((L - 4) & 31) : esign_hi >> ((L - 36) & 31)) & 1u;
    return p;
}

// The whole decode of workgroup `blk` of one (BG, Z) configuration.
// CRC: the CRC-aided stop compiled in (nrldpc_cfg.early_term = 2) -- builds of their own: as a run-time option it cost the other
// calls of this kernel registers (21 -> 75 spilled on BG1) and the mixed-batch launch 10-17 %
template <int BG, int DT, bool CRC = false>
__device__ __forceinline__ void decode_body(const DecArgs& a, const int32_t* __restrict__ rot_tab, int blk) {
    using G = BGD<BG>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int Z = a.Z;
    const int cwl = tid / Z, z = tid - cwl * Z;
    const int cw = blk * a.ncw + cwl;
    const bool active = (cwl < a.ncw) && (cw < a.batch) && (z < Z);
    const uint32_t zb = (uint32_t)z * (uint32_t)a.sbw + (uint32_t)cwl * (uint32_t)(G::NCP * 4);
    int* flags = reinterpret_cast<int*>(lds + (size_t)Z * a.sbw);
    const size_t ncwz = (size_t)G::COLS * Z;

    DecState<BG> st;
#pragma unroll
    for (int i = 0; i < G::NW; ++i) st.rm[i] = 0;
#pragma unroll
    for (int i = 0; i < G::NXW; ++i) st.xq[i] = 0;
    uint32_t esign_lo = 0, esign_hi = 0;
    float* app_row = nullptr; // &app[cw][z]

    if (active) {
        const size_t base = (size_t)cw * ncwz;
        if (a.app) app_row = a.app + base + z;
        // Issue every global load of a group before the first use so the HBM round trips overlap.
        {   // core columns -> LDS, ring position z
            float x[G::NC];
            static_for<G::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                x[c] = load_llr<DT>(a.llr, base + (size_t)c * Z + z);
            });
            static_for<G::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                *reinterpret_cast<float*>(lds + zb + 4 * c) = ingest(x[c], a.scale, true);
            });
        }
        {   // extension columns -> int8 registers (thread-private: shift 0, degree 1)
            // the extension-parity LLR of a pruned row is never used (only soft output echoes it): blocks of 8
            // rows behind wave-uniform branches keep those columns out of the HBM traffic
            float x[G::NEXT];
            const int next_used = a.app ? G::NEXT : launder(a.n_layers) - 4;
            static_for<(G::NEXT + 7) / 8>([&](auto bc) {
                constexpr int i0 = decltype(bc)::value * 8;
                constexpr int i1 = i0 + 8 < G::NEXT ? i0 + 8 : G::NEXT;
                if (i0 < next_used) {
                    static_for<i1 - i0>([&](auto ic) {
                        constexpr int i = i0 + decltype(ic)::value;
                        x[i] = load_llr<DT>(a.llr, base + (size_t)(G::NC + i) * Z + z);
                    });
                } else {
                    static_for<i1 - i0>([&](auto ic) { x[i0 + decltype(ic)::value] = 0.0f; });
                }
            });
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                f32_to_byte<i & 3>(st.xq[i >> 2], ingest(x[i], a.scale, false));
            });
        }
        if (app_row) { // soft output of columns whose layer is inactive = the ingested channel value
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                app_row[(size_t)(G::NC + i) * Z] = byte_to_f32<i & 3>(st.xq[i >> 2]) * a.inv_scale;
            });
        }
    }
    __syncthreads();

    bool done = !active;
    int my_iters = a.max_iter;
    for (int it = 1; it <= a.max_iter; ++it) {
        const ctab_t rot = launder(as_ctab(rot_tab));
        if (!done) { esign_lo = 0; esign_hi = 0; }
        static_for<G::ROWS>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            if (L < launder(a.n_layers)) {
                if (!done) layer<BG, L>(st, lds, zb, a, rot, esign_lo, esign_hi, app_row);
            }
            if constexpr (LayerGroups<BG>::group_end(L)) { // see LayerGroups: one barrier per column-disjoint group
                constexpr int gs = LayerGroups<BG>::group_start(L); // forced compile-time evaluation
                if (gs < launder(a.n_layers)) __syncthreads();
            }
        });
        if (a.early_term) {
            if (tid <= a.ncw) flags[tid] = 0; // flags[ncw] = "some codeword of this workgroup still fails"
            // CRC-aided stop (early_term = 2): per codeword CRC_SLOTS words behind the flags, at the next 16-byte boundary of LDS --
            // the flags themselves sit at Z*sbw, only a multiple of 4 when Z*ncw is odd (BG2 Z = 7 ...), and crc_holds() reads the
            // slots as int4 (nrldpc_create leaves 16 bytes of slack for this)
            const uint32_t slots_off = (uint32_t)(((size_t)Z * a.sbw + 4 * (size_t)(a.ncw + 1) + 15) & ~(size_t)15);
            int* crc_base = reinterpret_cast<int*>(lds + slots_off);
            int* crc_slots = crc_base + cwl * CRC_SLOTS;
            if constexpr (CRC)
                for (int i = tid; i < a.ncw * CRC_SLOTS; i += (int)blockDim.x) crc_base[i] = 0;
            __syncthreads();
            if (CRC && !done) { // every thread folds the information bits at its own ring position z of each column
                CrcFold f;
                static_for<G::KB>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    f.bit(*reinterpret_cast<const float*>(lds + zb + 4 * c), a.crc_tab, c * Z + z, a.crc_bits);
                });
                f.publish(crc_slots);
            }
            if (!done) {
                // One violated check settles a codeword's answer.  A lane that has found one publishes it at
                // once; at every vote point (each core row, then every 4 rows) a lane whose codeword is already
                // flagged -- by itself or by any other wave -- has nothing left to learn, and the wave stops
                // reading when that holds for all its lanes.  Far from convergence: ~19 LDS reads instead of all.
                uint32_t bad = 0;
                bool stop = false; // wave-uniform
                static_for<G::ROWS>([&](auto lc) {
                    constexpr int L = decltype(lc)::value;
                    if (!stop && L < launder(a.n_layers)) {
                        bad |= row_parity<BG, L>(lds, zb, a, rot, esign_lo, esign_hi);
                        if constexpr (L < 4 || (L % 4) == 3) {
                            if (bad) { flags[cwl] = 1; flags[a.ncw] = 1; }
                            stop = __all((int)(bad | (uint32_t)__atomic_load_n(&flags[cwl], __ATOMIC_RELAXED))) != 0;
                        }
                    }
                });
                if (bad) { flags[cwl] = 1; flags[a.ncw] = 1; }
            }
            __syncthreads();
            if constexpr (CRC) {
                // a codeword whose CRC holds is done although a parity check fails; the workgroup leaves when none is left
                if (!done && flags[cwl] != 0 && crc_holds(crc_slots)) { done = true; my_iters = it; }
                if (tid == 0) flags[a.ncw] = 0;
                __syncthreads();
                if (!done && flags[cwl] != 0) flags[a.ncw] = 1;
                __syncthreads();
            }
            if (!done && flags[cwl] == 0) { done = true; my_iters = it; }
            if (flags[a.ncw] == 0) break;
        }
    }

    if (active) {
        if (a.iters && z == 0) a.iters[cw] = my_iters;
        uint8_t* hard = a.hard + (size_t)cw * ((size_t)G::KB * Z);
        static_for<G::NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const float val = *reinterpret_cast<const float*>(lds + zb + 4 * c);
            if (c < G::KB) hard[(size_t)c * Z + z] = val < 0.0f ? 1 : 0;
            if (app_row) app_row[(size_t)c * Z] = val * a.inv_scale;
        });
    }
}

template <int BG, int DT, bool CRC = false>
__global__ __launch_bounds__(768, BG == 2 ? NRLDPC_GEN_WPE_BG2 : NRLDPC_GEN_WPE_BG1) void nrldpc_decode_kernel(const DecArgs a, const int32_t* __restrict__ rot_tab) {
    decode_body<BG, DT, CRC>(a, rot_tab, blockIdx.x);
}

// Mixed-(Z) batches in ONE launch: workgroup -> (configuration, local workgroup) through a prefix table, the
// configuration's argument block is fetched with scalar loads, then the same body runs.  A small bucket alone
// is a one-workgroup kernel that leaves 255 CUs idle for >100 us; a hundred of them queue behind each other
// (BASELINE configuration 4: 3.6 ms as 102 launches on 8 streams).
template <int BG, int DT>
__global__ __launch_bounds__(768, BG == 2 ? NRLDPC_GEN_WPE_BG2 : NRLDPC_GEN_WPE_BG1) void nrldpc_decode_multi_kernel(const DecArgs* __restrict__ tab,
                                                                                  const int32_t* __restrict__ wg_start, int nb) {
    const ctab_t st = as_ctab(wg_start);
    int lo = 0, hi = nb; // wg_start[lo] <= blockIdx.x < wg_start[hi], wg_start[nb] = grid size
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= st[mid]) lo = mid; else hi = mid;
    }
    constexpr int NW = (int)(sizeof(DecArgs) / 4);
    static_assert(sizeof(DecArgs) % 4 == 0, "argument block is copied as dwords");
    const ctab_t src = as_ctab(reinterpret_cast<const int32_t*>(tab + lo));
    int32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = src[i];
    DecArgs a;
    __builtin_memcpy(&a, w, sizeof a);
    decode_body<BG, DT>(a, a.rot, (int)blockIdx.x - st[lo]);
}

template <int BG, int DT> static hipError_t launch_multi_t(const DecArgs* d_tab, const int32_t* d_start, int nb, int grid, int threads,
                                                           size_t lds, hipStream_t s) {
    auto k = nrldpc_decode_multi_kernel<BG, DT>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    if (threads <= 0 || threads > 768 || threads % 64) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, d_tab, d_start, nb);
    return hipGetLastError();
}

hipError_t launch_decode_multi_wg(int bg, int llr_kind, const DecArgs* d_tab, const int32_t* d_start, int nb, int grid, int threads,
                                  size_t lds_bytes, hipStream_t stream) {
    const bool f16 = llr_kind == NRLDPC_K_F16;
    if (bg == 1)
        return f16 ? launch_multi_t<1, NRLDPC_K_F16>(d_tab, d_start, nb, grid, threads, lds_bytes, stream)
                   : launch_multi_t<1, NRLDPC_K_F32>(d_tab, d_start, nb, grid, threads, lds_bytes, stream);
    return f16 ? launch_multi_t<2, NRLDPC_K_F16>(d_tab, d_start, nb, grid, threads, lds_bytes, stream)
               : launch_multi_t<2, NRLDPC_K_F32>(d_tab, d_start, nb, grid, threads, lds_bytes, stream);
}

hipError_t launch_decode_multi(int bg, int llr_kind, const DecArgs* d_tab, const int32_t* d_start, int nb, int grid,
                               size_t lds_bytes, hipStream_t stream) {
    return launch_decode_multi_wg(bg, llr_kind, d_tab, d_start, nb, grid, bg == 1 ? NRLDPC_GEN_THREADS_BG1 : NRLDPC_GEN_THREADS_BG2,
                                  lds_bytes, stream);
}

template <int BG, int DT, bool CRC = false> static hipError_t launch_t(const DecArgs& a, int grid, int threads, size_t lds, hipStream_t s) {
    auto k = nrldpc_decode_kernel<BG, DT, CRC>;
    static bool attr_set[64] = {}; // per device: raising the dynamic-LDS limit is a slow host call, do it once
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a, a.rot);
    return hipGetLastError();
}

static bool force_generic_env() {
    static const bool f = getenv("NRLDPC_FORCE_GENERIC") != nullptr; // A/B of the two kernels (tools/bench_all_z.py)
    return f;
}

bool has_z64_kernel(int bg, int Z) {
    if (force_generic_env()) return false;
#define NRLDPC_Z64_CASE(b, z) if (bg == b && Z == z) return true;
    NRLDPC_Z64_LIST(NRLDPC_Z64_CASE)
#undef NRLDPC_Z64_CASE
    return false;
}

bool has_z64p_kernel(int bg, int Z, bool early_term) {
    static const bool no_packed = getenv("NRLDPC_NO_PACKED") != nullptr; // A/B against the block-geometry / run-time-Z kernels
    if (force_generic_env() || no_packed) return false;
#define NRLDPC_Z64P_CASE(b, z) if (early_term && bg == b && Z == z) return false;
    NRLDPC_Z64P_NOT_ET(NRLDPC_Z64P_CASE)
#undef NRLDPC_Z64P_CASE
#define NRLDPC_Z64P_CASE(b, z) if (bg == b && Z == z) return true;
    NRLDPC_Z64P_LIST(NRLDPC_Z64P_CASE)
#undef NRLDPC_Z64P_CASE
    return false;
}

hipError_t launch_decode(int bg, const DecArgs& a, int threads, size_t lds_bytes, hipStream_t stream) {
    const bool force_generic = force_generic_env();
    static const bool no_packed = getenv("NRLDPC_NO_PACKED") != nullptr;
    // (the CRC-aided stop, early_term = 2, lives in the block-geometry and run-time-Z kernels: packed sizes take those then)
    const bool crc = a.crc_bits != 0;
    // the interleaved block geometry: hard output, any layer count -- for what each entry's mode lists (1 fixed iteration counts,
    // 2 parity stop with every row active, 4 parity stop with pruned rows: measured per size, profiles/r04_ilv_ab.txt).
    // NRLDPC_NO_ILV=1: A/B against the kernels that served these sizes before
    static const bool no_ilv = getenv("NRLDPC_NO_ILV") != nullptr;
    if (!force_generic && !no_ilv && !a.app && !crc) {
        const int want = !a.early_term ? 1 : a.n_layers == (bg == 1 ? BGT<1>::ROWS : BGT<2>::ROWS) ? 2 : 4;
#ifdef NRLDPC_Z64I_FORCE_MODE // the experiment build (build.py: NRLDPC_BUILD_ALLMODES): every entry serves these modes, whatever the list says
#define NRLDPC_Z64I_CASE(b, z, ncw, mode) if (bg == b && a.Z == z && ((NRLDPC_Z64I_FORCE_MODE) & want)) return launch_decode_z64i_##b##_##z(a, stream);
#else
#define NRLDPC_Z64I_CASE(b, z, ncw, mode) if (bg == b && a.Z == z && ((mode) & want)) return launch_decode_z64i_##b##_##z(a, stream);
#endif
        NRLDPC_Z64I_LIST(NRLDPC_Z64I_CASE)
#undef NRLDPC_Z64I_CASE
    }
    if (!force_generic && !no_packed && !a.app && !crc) { // pruned layer counts with packed builds of their own
#define NRLDPC_Z64P_NL_CASE(b, z, nl) if (bg == b && a.Z == z && a.n_layers == nl) return launch_decode_z64p_##b##_##z##_nl##nl(a, stream);
        NRLDPC_Z64P_NL_LIST(NRLDPC_Z64P_NL_CASE)
#undef NRLDPC_Z64P_NL_CASE
    }
    static const bool no_packed_row = getenv("NRLDPC_NO_PACKED_ROW") != nullptr; // A/B against the block-geometry / run-time-Z kernels
    if (!force_generic && !no_packed && !no_packed_row && !a.app && !crc) { // packed geometry, pipelined one-thread-per-row builds
#define NRLDPC_Z64PR_CASE(b, z, rw) if (bg == b && a.Z == z) return launch_decode_z64pr_##b##_##z(a, stream);
        NRLDPC_Z64PR_LIST(NRLDPC_Z64PR_CASE)
#undef NRLDPC_Z64PR_CASE
    }
    // the packed builds: hard output; every row active, or any other layer count as a run-time prefix (NL_RT; NRLDPC_NO_RT=1
    // sends those to the kernels that served them before -- A/B)
    static const bool no_rt = getenv("NRLDPC_NO_RT") != nullptr;
    // (BG2 with pruned rows AND the parity stop stays with the general kernel of this geometry below: 8-18 % faster there,
    // profiles/r04_bench_nl_packed.txt)
    const bool all_rows = a.n_layers == (bg == 1 ? BGT<1>::ROWS : BGT<2>::ROWS);
    if (!a.app && !crc && (all_rows || (!no_rt && !(bg == 2 && a.early_term))) && has_z64p_kernel(bg, a.Z, a.early_term != 0)) {
#define NRLDPC_Z64P_CASE(b, z) if (bg == b && a.Z == z) return launch_decode_z64p_##b##_##z(a, stream);
        NRLDPC_Z64P_LIST(NRLDPC_Z64P_CASE)
#undef NRLDPC_Z64P_CASE
    }
    static const bool no_packed_general = getenv("NRLDPC_NO_PACKED_GENERAL") != nullptr; // A/B against the run-time-Z kernel
    if (bg == 2 && !crc && !force_generic && !no_packed && !no_packed_general && !has_z64_kernel(bg, a.Z)) {
        // the packed geometry's general kernel: any layer count, soft output -- BG2 (nrldpc_decode_z64p.h: z64pg_serves); sizes
        // with a block-geometry build keep that one's
#define NRLDPC_Z64P_CASE(b, z) if (bg == b && a.Z == z) return launch_decode_z64pg_##b##_##z(a, stream);
        NRLDPC_Z64P_LIST(NRLDPC_Z64P_CASE)
#undef NRLDPC_Z64P_CASE
    }
#define NRLDPC_Z64_CASE(b, z) if (!force_generic && bg == b && a.Z == z) return launch_decode_z64_##b##_##z(a, stream);
    NRLDPC_Z64_LIST(NRLDPC_Z64_CASE)
#undef NRLDPC_Z64_CASE
    const int grid = (a.batch + a.ncw - 1) / a.ncw;
    const bool f16 = a.llr_kind == NRLDPC_K_F16;
    if (crc) {
        if (bg == 1)
            return f16 ? launch_t<1, NRLDPC_K_F16, true>(a, grid, threads, lds_bytes, stream)
                       : launch_t<1, NRLDPC_K_F32, true>(a, grid, threads, lds_bytes, stream);
        return f16 ? launch_t<2, NRLDPC_K_F16, true>(a, grid, threads, lds_bytes, stream)
                   : launch_t<2, NRLDPC_K_F32, true>(a, grid, threads, lds_bytes, stream);
    }
    if (bg == 1)
        return f16 ? launch_t<1, NRLDPC_K_F16>(a, grid, threads, lds_bytes, stream)
                   : launch_t<1, NRLDPC_K_F32>(a, grid, threads, lds_bytes, stream);
    return f16 ? launch_t<2, NRLDPC_K_F16>(a, grid, threads, lds_bytes, stream)
               : launch_t<2, NRLDPC_K_F32>(a, grid, threads, lds_bytes, stream);
}

} // namespace nrldpc
