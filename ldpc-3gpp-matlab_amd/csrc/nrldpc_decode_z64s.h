// nrldpc_decode_z64s.h -- "split" form of the compile-time-Z decoder (nrldpc_decode_z64.h): TWO threads per check row.
//
// The one-thread-per-row kernel keeps all of a row's check-to-variable messages in its thread's registers (BG1: 69 + 11
// VGPRs of state), which leaves room for 3 waves per SIMD at Z = 384, all of one workgroup and in step with each other:
// its VALU pipes are busy ~70 % of the time.  Here the column-disjoint barrier groups of the
// layered schedule ALTERNATE between two threads of the same row ("halves" H = 0 / 1 of the workgroup):
//
//   interval gi (between two workgroup barriers):   half gi % 2       finishes group gi   (late reads, pass 2, LDS writes)
//                                                   half (gi+1) % 2   prepares group gi+1 (early reads + min search)
//
// -- exactly the two parts the software pipeline of pipeline_z64 runs back to back in ONE wave, now in different waves.
// Each thread holds the messages of every other group only (half the state registers), so twice the waves fit:
// one codeword = 2 * Z/64 waves = one workgroup, two workgroups per CU, 6 waves per SIMD.  The instruction count per
// codeword is unchanged (every edge is still processed once, by one thread); what changes is that the preparing half's
// work is always there to fill the finishing half's LDS round trips, and that the two codewords of a CU no longer share
// a barrier.  Results are identical (same arithmetic in the same order per row; min / med3 / xor are order-independent).
//
// Requires an even number of barrier groups (so that the alternation is the same in every iteration): BG1 all rows 32,
// BG2 all rows 28.
#ifndef NRLDPC_DECODE_Z64S_H
#define NRLDPC_DECODE_Z64S_H
#include "nrldpc_decode_z64.h"

namespace nrldpc {
inline namespace NRLDPC_UNIT { // one name space per translation unit: see NRLDPC_UNIT in nrldpc_decode_z64.h

#ifndef NRLDPC_Z64S_WPE
#define NRLDPC_Z64S_WPE 6 // waves per SIMD the register allocation is sized for (two 12-wave workgroups per CU at Z = 384)
#endif
#ifndef NRLDPC_Z64S_RULE_SGPR
#define NRLDPC_Z64S_RULE_SGPR 1 // alpha, 2^23 - beta as scalar operands (two 4-cycle ops per row instead of 2 VGPRs)
#endif
#ifndef NRLDPC_Z64S_PRIO
#define NRLDPC_Z64S_PRIO 2
#endif
// The hand-over between the halves.  -DNRLDPC_EXP_NOBARRIER=1 builds a TIMING-ONLY kernel (results wrong): the workgroup
// barriers of the iteration loop become a wait for the wave's own LDS operations, which bounds what any finer-grained
// synchronisation (per-column flags between the waves that share a ring segment) could return
// (profiles/r04_headline_nobarrier.txt).
#if defined(NRLDPC_EXP_NOBARRIER) && NRLDPC_EXP_NOBARRIER == 1
#define NRLDPC_Z64S_HANDOVER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define NRLDPC_Z64S_HANDOVER() __syncthreads()
#endif

template <int BG, int ZC, int NL> struct Z64S : Z64<BG, ZC, 1, NL> {
    using B = Z64<BG, ZC, 1, NL>;
    static constexpr int NG = LGof<BG, ZC, NL, 0>::ngroups();
    static constexpr int THREADS = 2 * B::TPC;
    static constexpr bool usable() { return THREADS <= 1024 && NG >= 2; }
    // rings + trailing guard + flags [+ the extension-column channel LLRs, one int8 per extension row and row-thread]
    // (16 bytes of flags, then the CRC-aided stop's CRC_SLOTS words, nrldpc_device.h: CrcFold)
    static constexpr size_t XOFF = (size_t)B::CWS + B::GUARD + 16 + 4 * (size_t)CRC_SLOTS;
    static constexpr size_t XBYTES = (size_t)(B::NLT - 4) * ZC;
    // Workgroups a CU holds: 24 wave slots at the 80-VGPR budget, 160 KB of LDS.  The extension LLRs move from 6 registers
    // per thread into LDS exactly where that does not cost a workgroup (Z = 384, 288, 256, 240 ... : the wave slots bind;
    // Z <= 192 with 2- to 6-wave workgroups: LDS binds, and one workgroup fewer per CU costs 4-8 %, measured).
    static constexpr int wgs_per_cu(size_t lds) {
        const int by_waves = 24 / (2 * B::NWV) > 0 ? 24 / (2 * B::NWV) : 1, by_lds = (int)((160 * 1024) / lds);
        return by_waves < by_lds ? by_waves : by_lds;
    }
    // dual rows (Own::dual): the exchange buffer of the two halves' partial searches, 8 bytes per thread
    static constexpr bool DUAL = Own<BG, NL, 0, z64s_variant<BG, ZC, NL>()>::dual(0);
    static constexpr size_t XCHBYTES = DUAL ? (size_t)2 * ZC * 8 : 0;
    static constexpr bool XL = wgs_per_cu(XOFF + XCHBYTES + XBYTES) == wgs_per_cu(XOFF + XCHBYTES);
    static constexpr size_t XCHOFF = (XOFF + (XL ? XBYTES : 0) + 7) & ~(size_t)7;
    static constexpr size_t lds_bytes() { return XCHOFF + XCHBYTES; }
    // waves per SIMD the register allocation is sized for: NRLDPC_Z64S_WPE, or fewer where the LDS image holds a CU below that
    // anyway (BG1 Z = 60, 64: eight 2-wave workgroups = 4 per SIMD, so 128 VGPRs instead of 80 and no scratch)
    static constexpr int wpe() {
        const int by_lds = (int)((160 * 1024) / lds_bytes()) * (THREADS / 64) / 4;
        return by_lds >= NRLDPC_Z64S_WPE ? NRLDPC_Z64S_WPE : by_lds >= 1 ? by_lds : 1;
    }
};

template <int BG, int ZC, int NL, int H, bool ET, bool XF, int GI, class St>
__device__ __forceinline__ void s_early(GroupZ64<BG, ZC, 0, NL, H>& next0, St& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE],
                                        uint32_t RA, uint32_t RB, int w, const DecArgs& a, float cap, uint32_t& esign_lo,
                                        uint32_t& esign_hi);

// interval GI, this half owns group GI: `cur` arrives with its early part done
template <int BG, int ZC, int NL, int H, bool ET, bool XF, int GI, class St>
__device__ __forceinline__ void s_crit(GroupZ64<BG, ZC, GI, NL, H>& cur, GroupZ64<BG, ZC, 0, NL, H>& next0, St& st, char* lds,
                                       const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA, uint32_t RB, int w, const DecArgs& a,
                                       float cap, uint32_t& esign_lo, uint32_t& esign_hi) {
    using LG = LGof<BG, ZC, NL, 0>;
    constexpr int NG = LG::ngroups();
    constexpr bool RT = NL == NL_RT; // run-time layer count: the iteration ends after the last active layer's group (the kernel prepares group 0)
    // ends interval GI-1: group GI-1's writes are visible.  With early termination the parity pass between two iterations
    // ends with a barrier of its own (and the first iteration follows the prologue's), so interval 0 needs none.
    if constexpr (!(ET && GI == 0)) NRLDPC_Z64S_HANDOVER();
    __builtin_amdgcn_s_setprio(NRLDPC_Z64S_PRIO); // the next barrier waits for this half
    cur.template loads<true>(lds, R);
    cur.template track<true, XF>(st, cap);
    cur.finish(st, lds, R, a);
    cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
    __builtin_amdgcn_s_setprio(0);
    if constexpr (ET) {
        cur.ext(a, esign_lo, esign_hi);
        asm volatile("" : "+v"(esign_lo), "+v"(esign_hi)); // see pipeline_z64
    }
    if constexpr (GI + 1 < NG) {
        if (!RT || LG::group_first(GI + 1) < launder(a.n_layers))
            s_early<BG, ZC, NL, H, ET, XF, GI + 1>(next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
    } else if constexpr (RT) {
    } else if constexpr (H == 0 || Own<BG, NL, H, z64s_variant<BG, ZC, NL>()>::dual(0)) { // (a dual row 0: both halves prepare their own edges of it)
        // odd group count: this half owns the last group AND group 0, whose early part it runs right here (the one
        // interval per iteration that is software-pipelined within a wave, as in pipeline_z64)
        next0.template loads<false>(lds, R);
        next0.template track<false, XF>(st, cap);
    }
}

// interval GI, the other half owns group GI: this half prepares group GI+1 (cyclically: group 0 of the next iteration)
template <int BG, int ZC, int NL, int H, bool ET, bool XF, int GI, class St>
__device__ __forceinline__ void s_early(GroupZ64<BG, ZC, 0, NL, H>& next0, St& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE],
                                        uint32_t RA, uint32_t RB, int w, const DecArgs& a, float cap, uint32_t& esign_lo,
                                        uint32_t& esign_hi) {
    using LG = LGof<BG, ZC, NL, 0>;
    constexpr int NG = LG::ngroups();
    constexpr bool RT = NL == NL_RT; // see s_crit
    if constexpr (!(ET && GI == 0)) NRLDPC_Z64S_HANDOVER(); // see s_crit
    if constexpr (GI + 1 < NG) {
        if (!RT || LG::group_first(GI + 1) < launder(a.n_layers)) {
            GroupZ64<BG, ZC, GI + 1, NL, H> nxt;
            nxt.template loads<false>(lds, R); // columns group GI does not write
            nxt.template track<false, XF>(st, cap);
            s_crit<BG, ZC, NL, H, ET, XF, GI + 1>(nxt, next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
        }
    } else if constexpr (RT) {
    } else if constexpr (H == 0 || Own<BG, NL, H, z64s_variant<BG, ZC, NL>()>::dual(0)) { // even group count: the last group is the other half's, group 0 is this one's
        next0.template loads<false>(lds, R);
        next0.template track<false, XF>(st, cap);
    }
}

// interval(s) of a dual row GI (a one-layer group worked on by both halves, Own::dual): `cur` arrives with this half's early
// edges done.  Two barriers: the usual one in front of the group, and one between the halves' partial searches and pass 2.
template <int BG, int ZC, int NL, int H, bool ET, bool XF, int GI, class St>
__device__ __forceinline__ void s_dense(GroupZ64<BG, ZC, GI, NL, H>& cur, GroupZ64<BG, ZC, 0, NL, H>& next0, St& st, char* lds,
                                        const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA, uint32_t RB, int w, const DecArgs& a,
                                        float cap, uint32_t& esign_lo, uint32_t& esign_hi, uint32_t xmine, uint32_t xother) {
    using O = Own<BG, NL, H, z64s_variant<BG, ZC, NL>()>;
    using LG = LGof<BG, ZC, NL, 0>;
    constexpr int NG = LG::ngroups();
    constexpr bool RT = NL == NL_RT; // see s_crit (the dual rows 0..3 themselves are always active)
    static_assert(LG::group_last(LG::group_first(GI)) == LG::group_first(GI), "a dual row is a barrier group of its own");
    if constexpr (!(ET && GI == 0)) NRLDPC_Z64S_HANDOVER(); // see s_crit
    __builtin_amdgcn_s_setprio(NRLDPC_Z64S_PRIO);
    cur.template loads<true>(lds, R);
    cur.template track<true, XF>(st, cap);
    cur.l0.publish(lds, xmine);
    NRLDPC_Z64S_HANDOVER();
    cur.l0.merge(lds, xother);
    cur.finish(st, lds, R, a);
    cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
    __builtin_amdgcn_s_setprio(0);
    if constexpr (GI + 1 < NG) {
        if (RT && LG::group_first(GI + 1) >= launder(a.n_layers)) {
        } else if constexpr (O::dual(LG::group_first(GI + 1))) {
            GroupZ64<BG, ZC, GI + 1, NL, H> nxt;
            nxt.template loads<false>(lds, R);
            nxt.template track<false, XF>(st, cap);
            s_dense<BG, ZC, NL, H, ET, XF, GI + 1>(nxt, next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi, xmine, xother);
        } else if constexpr ((GI + 1) % 2 == H) { // the next group is this half's alone: its early part, here
            GroupZ64<BG, ZC, GI + 1, NL, H> nxt;
            nxt.template loads<false>(lds, R);
            nxt.template track<false, XF>(st, cap);
            s_crit<BG, ZC, NL, H, ET, XF, GI + 1>(nxt, next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
        } else {
            s_early<BG, ZC, NL, H, ET, XF, GI + 1>(next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
        }
    } else if constexpr (!RT) { // the dual row was the last group: row 0 (dual as well) follows
        next0.template loads<false>(lds, R);
        next0.template track<false, XF>(st, cap);
    }
}

// CRC: the parity-stop build with the CRC-aided stop compiled in (nrldpc_cfg.early_term = 2, nrldpc_device.h: CrcFold) -- a twin of
// its own: as a run-time option inside the ETP build it cost every parity-stop call 3-5 % (registers, the slot reset)
template <int BG, int ZC, bool ETP, int NL = BGT<BG>::ROWS, bool CRC = false>
__global__ __launch_bounds__(2 * z64_nwv(ZC) * 64, (Z64S<BG, ZC, NL>::wpe())) void nrldpc_decode_z64s_kernel(const DecArgs a) {
    static_assert(!CRC || ETP, "the CRC-aided stop is a mode of the parity-stop build");
    using G = Z64S<BG, ZC, NL>;
    using LGN = LGof<BG, ZC, NL, 0>;
    static_assert(G::usable(), "split kernel: at most 1024 threads");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave / G::NWV, w = wave % G::NWV, lane = tid & 63;
    if constexpr (G::BLK < 64) {
        if (lane >= G::BLK) return; // these lanes own no row; barriers count waves, not lanes
    }
    const int z = w * G::BLK + lane;
    const int u = half * ZC + z; // dense index of this thread among the 2 Z threads that own rows
    const int cw = blockIdx.x;
    int* flags = reinterpret_cast<int*>(lds + (size_t)G::CWS + G::GUARD);
    constexpr size_t ncwz = (size_t)G::COLS * ZC;
    constexpr bool XF = false; // extension LLRs: int8 in LDS or in registers (DecStateS, Z64S::XL), never floats

    uint32_t R[G::NWV];
#pragma unroll
    for (int k = 0; k < G::NWV; ++k) R[k] = G::GUARD + (uint32_t)(4 * G::BLK) * (uint32_t)((w + k) % G::NWV) + 4u * (uint32_t)lane;
    const uint32_t RA = (uint32_t)(G::GUARD - 4 * G::BLK) + 4u * (uint32_t)lane;
    const uint32_t RB = G::GUARD + 4u * ZC + 4u * (uint32_t)lane;
    const size_t base = (size_t)cw * ncwz;

    // ---- core columns -> LDS: ZC/4 threads cover a column with 4 consecutive ring positions each, 8 columns per pass
    {
        constexpr int QW = ZC / 4;
        const int qs = u / QW, qq = u - qs * QW;
        constexpr int NP = (G::NC + 7) / 8;
        const bool wide = (reinterpret_cast<uintptr_t>(a.llr) & 15) == 0;
        auto ingest_as = [&](auto kind_c) {
            constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
            if (wide) {
                uint4 x[NP];
                static_for<NP>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 8 * k + qs;
                    x[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (8 * k + 7 < G::NC || c < G::NC) {
                        const size_t i = base + (size_t)c * ZC + 4 * qq;
                        if constexpr (F16) {
                            const uint2 r = *reinterpret_cast<const uint2*>(static_cast<const __half*>(a.llr) + i);
                            x[k].x = r.x; x[k].y = r.y;
                        } else {
                            x[k] = *reinterpret_cast<const uint4*>(static_cast<const float*>(a.llr) + i);
                        }
                    }
                });
                static_for<NP>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 8 * k + qs;
                    if (8 * k + 7 < G::NC || c < G::NC) {
                        float4 v;
                        if constexpr (F16) {
                            const __half2 lo = *reinterpret_cast<const __half2*>(&x[k].x), hi = *reinterpret_cast<const __half2*>(&x[k].y);
                            v = make_float4(__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi));
                        } else {
                            v = make_float4(__uint_as_float(x[k].x), __uint_as_float(x[k].y), __uint_as_float(x[k].z), __uint_as_float(x[k].w));
                        }
                        const float4 q = make_float4(ingest(v.x, a.scale, true), ingest(v.y, a.scale, true),
                                                     ingest(v.z, a.scale, true), ingest(v.w, a.scale, true));
                        char* col = lds + G::GUARD + c * G::CS;
                        *reinterpret_cast<float4*>(col + 16 * qq) = q;
                        if (qq < G::BLK / 4) *reinterpret_cast<float4*>(col + 4 * ZC + 16 * qq) = q; // mirror of block 0
                    }
                });
            } else { // unaligned LLR pointer: one ring position per thread, the halves take alternate columns
                constexpr int NPU = (G::NC + 1) / 2;
                uint32_t x[NPU];
                static_for<NPU>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 2 * k + half;
                    x[k] = 0u;
                    if (2 * k + 1 < G::NC || c < G::NC) {
                        if constexpr (F16) x[k] = static_cast<const uint16_t*>(a.llr)[base + (size_t)c * ZC + z];
                        else x[k] = static_cast<const uint32_t*>(a.llr)[base + (size_t)c * ZC + z];
                    }
                });
                static_for<NPU>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 2 * k + half;
                    if (2 * k + 1 < G::NC || c < G::NC) {
                        float v;
                        if constexpr (F16) v = __half2float(__ushort_as_half((unsigned short)x[k]));
                        else v = __uint_as_float(x[k]);
                        const float q = ingest(v, a.scale, true);
                        char* home = lds + G::GUARD + 4 * z + c * G::CS;
                        *reinterpret_cast<float*>(home) = q;
                        if (w == 0) *reinterpret_cast<float*>(home + ZC * 4) = q;
                    }
                });
            }
        };
        if (a.llr_kind == NRLDPC_K_F16) ingest_as(std::integral_constant<int, NRLDPC_K_F16>{});
        else ingest_as(std::integral_constant<int, NRLDPC_K_F32>{});
    }

    int my_iters = a.max_iter;
    // One half's whole decode.  Instantiated twice; the branch on `half` is wave-uniform.
    auto run = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        using O = Own<BG, NL, H, z64s_variant<BG, ZC, NL>()>;
        DecStateS<BG, NL, H, ZC, G::XL, z64s_variant<BG, ZC, NL>()> st;
#pragma unroll
        for (int i = 0; i < O::NW; ++i) st.rm[i] = 0;
        if constexpr (G::XL) { // half 0's extension rows first, then half 1's: [row][thread] bytes
            st.xp = (lds_i8_t)(lds + G::XOFF + (size_t)(H == 0 ? 0 : Own<BG, NL, 0, z64s_variant<BG, ZC, NL>()>::NEXT) * ZC + z);
        } else {
#pragma unroll
            for (int i = 0; i < O::NXW; ++i) st.xq[i] = 0;
        }
        // extension LLRs of this half's rows: thread-private, one load per row; raw bits first, conversions after, in a
        // body per LLR format (no format test per load: see the prologue of the one-thread-per-row kernel)
        auto load_ext = [&](auto kind_c) {
            constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
            uint32_t xe[O::NEXT > 0 ? O::NEXT : 1];
            // (run-time layer count: the extension LLR of a pruned row is never used -- blocks of 8 rows behind wave-uniform
            // branches keep those columns out of the HBM traffic: at R = 8/9 that is 41 of BG1's 68 columns)
            static_for<(G::NLT - 4 + 7) / 8>([&](auto bc) {
                constexpr int L0 = 4 + 8 * decltype(bc)::value;
                constexpr int L1 = L0 + 8 < G::NLT ? L0 + 8 : G::NLT;
                if (!G::RT || L0 < launder(a.n_layers)) {
                    static_for<L1 - L0>([&](auto ic) {
                        constexpr int L = L0 + decltype(ic)::value;
                        if constexpr (O::mine(L)) {
                            constexpr int xi = O::ext_index(L);
                            const size_t i = base + (size_t)(G::NC + L - 4) * ZC + z;
                            if constexpr (F16) xe[xi] = static_cast<const uint16_t*>(a.llr)[i];
                            else xe[xi] = static_cast<const uint32_t*>(a.llr)[i];
                        }
                    });
                } else {
                    static_for<L1 - L0>([&](auto ic) {
                        constexpr int L = L0 + decltype(ic)::value;
                        if constexpr (O::mine(L)) xe[O::ext_index(L)] = 0u;
                    });
                }
            });
            static_for<O::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float v;
                if constexpr (F16) v = __half2float(__ushort_as_half((unsigned short)xe[i]));
                else v = __uint_as_float(xe[i]);
                st.template set_ext<i>(ingest(v, a.scale, false)); // thread-private: no barrier needed before its reads
            });
        };
        if (a.llr_kind == NRLDPC_K_F16) load_ext(std::integral_constant<int, NRLDPC_K_F16>{});
        else load_ext(std::integral_constant<int, NRLDPC_K_F32>{});
        __syncthreads(); // the a-posteriori rings are complete
        const float cap = (127.49f + a.beta) / a.alpha; // see LayerZ64::track3
        DecArgs av = a;                                  // alpha, 2^23 - beta as VGPR values: see the one-thread-per-row kernel
        av.beta = 8388608.0f - a.beta;
#if !NRLDPC_Z64S_RULE_SGPR
        asm volatile("" : "+v"(av.alpha), "+v"(av.beta));
#endif
        uint32_t esign_lo = 0, esign_hi = 0;
        GroupZ64<BG, ZC, 0, NL, H> g0;
        constexpr bool D0 = O::dual(0); // row 0 is worked on by both halves
        // this thread's slot of the dual rows' exchange buffer, and its partner's (the other half's thread of the same row)
        const uint32_t xmine = (uint32_t)G::XCHOFF + 8u * (uint32_t)(H * ZC + z), xother = (uint32_t)G::XCHOFF + 8u * (uint32_t)((1 - H) * ZC + z);
        if constexpr (H == 0 || D0) {
            g0.template loads<false>(lds, R);
            g0.template track<false, XF>(st, cap);
        }
        for (int it = 1; it <= a.max_iter; ++it) {
            if constexpr (ETP) { esign_lo = 0; esign_hi = 0; }
            if constexpr (D0) {
                GroupZ64<BG, ZC, 0, NL, H> nx;
                s_dense<BG, ZC, NL, H, ETP, XF, 0>(g0, nx, st, lds, R, RA, RB, w, av, cap, esign_lo, esign_hi, xmine, xother);
                if constexpr (G::RT) { // wherever the iteration ended: group 0's early part (none of its edges is early here)
                    nx.template loads<false>(lds, R);
                    nx.template track<false, XF>(st, cap);
                }
                g0 = nx;
            } else if constexpr (H == 0) {
                GroupZ64<BG, ZC, 0, NL, H> nx;
                s_crit<BG, ZC, NL, H, ETP, XF, 0>(g0, nx, st, lds, R, RA, RB, w, av, cap, esign_lo, esign_hi);
                if constexpr (G::RT) {
                    nx.template loads<false>(lds, R);
                    nx.template track<false, XF>(st, cap);
                }
                g0 = nx;
            } else {
                s_early<BG, ZC, NL, H, ETP, XF, 0>(g0, st, lds, R, RA, RB, w, av, cap, esign_lo, esign_hi);
            }
            if constexpr (ETP) {
                // parity check of this half's rows (see parity_pass of the one-thread-per-row kernel)
                if (u == 0) flags[0] = 0;
                int* crc_slots = flags + 4; // CRC-aided stop (early_term = 2)
                if constexpr (CRC) { if (u < CRC_SLOTS) crc_slots[u] = 0; }
                __syncthreads();
                if constexpr (CRC) { // this half's columns (alternate ones), the bit at this thread's own ring position z (primary copy)
                    CrcFold f;
                    int tz = threadIdx.x; // derived again from the thread id: z is not kept live across the iteration for this
                    asm volatile("" : "+v"(tz));
                    const int zz = ((tz >> 6) % G::NWV) * G::BLK + (tz & 63);
                    const char* home = lds + G::GUARD + 4 * zz;
                    static_for<(G::KB + 1) / 2>([&](auto kc) {
                        constexpr int c = 2 * decltype(kc)::value + H;
                        if constexpr (c < G::KB) f.bit(*reinterpret_cast<const float*>(home + c * G::CS), a.crc_tab, c * ZC + zz, a.crc_bits);
                    });
                    f.publish(crc_slots);
                }
                uint32_t bad = 0;
                bool stop = false; // wave-uniform
                if constexpr (G::RT) {
                    // run-time layer count: highest row first, pruned rows skipped eight at a time (Own::parity_order_desc)
                    constexpr auto PD = O::parity_order_desc();
                    static_for<(PD.n + 7) / 8>([&](auto bc) {
                        constexpr int b0 = 8 * decltype(bc)::value, b1 = b0 + 8 < PD.n ? b0 + 8 : PD.n;
                        const int nlb = launder(a.n_layers);
                        if (!stop && PD.v[b1 - 1] < nlb) { // the block's lowest row is active: the block has work
                            static_for<b1 - b0>([&](auto ic) {
                                constexpr int i = b0 + decltype(ic)::value;
                                constexpr int L = PD.v[i];
                                if (!stop && L < nlb) {
                                    bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                                    if constexpr ((i % 4) == 3 || i + 1 == b1 || O::ncore(L) > 10 || (i + 1 < PD.n && O::ncore(PD.v[i + 1 < PD.n ? i + 1 : i]) > 10))
                                        stop = __any((int)bad) != 0;
                                }
                            });
                        }
                    });
                }
                constexpr auto PO = O::parity_order(); // cheapest rows first
                if constexpr (!G::RT) static_for<PO.n>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int L = PO.v[i];
                    if (!stop && (!G::RT || L < launder(a.n_layers))) {
                        bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                        // (a vote right after every dense row as well: a vote is a point the compiler cannot move loads across,
                        // and 19 + 16 a-posteriori words in flight spill at 80 VGPRs)
                        if constexpr (i < 3 || (i % 4) == 3 || i + 1 == PO.n || O::ncore(L) > 10 || (i + 1 < PO.n && O::ncore(PO.v[i + 1 < PO.n ? i + 1 : i]) > 10))
                            stop = __any((int)bad) != 0;
                    }
                });
                if (bad) flags[0] = 1;
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(flags[0]) == 0) { my_iters = it; break; }
                if constexpr (CRC) { // the CRC of the information bits holds although a parity check does not: done as well
                    // (the slots are reset in the next parity pass, a whole iteration of barriers away from this read)
                    if (__builtin_amdgcn_readfirstlane((int)crc_holds(crc_slots)) != 0) { my_iters = it; break; }
                }
            }
        }
        if constexpr (!ETP) __syncthreads(); // the last group's writes
    };
    if (half == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});

    // Write-back.  Every index is derived AGAIN from the thread id, through an opaque copy: values computed before the iteration
    // loop and needed only here would otherwise stay live across it, and at 80 VGPRs the compiler parks them in scratch
    // (8 dwords per thread written to and read from HBM: measured as 1.5x the compulsory traffic).
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int wave2 = tid2 >> 6;
    const int half2 = wave2 / G::NWV, z2 = (wave2 % G::NWV) * G::BLK + (tid2 & 63), u2 = half2 * ZC + z2;
    if (a.iters && u2 == 0) a.iters[cw] = my_iters;
    uint8_t* hard = a.hard + (size_t)cw * ((size_t)G::KB * ZC);
    if ((reinterpret_cast<uintptr_t>(a.hard) & 3) == 0) {
        constexpr int QW = ZC / 4;
        const int qs = u2 / QW, qq = u2 - qs * QW;
        static_for<(G::KB + 7) / 8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int c = 8 * k + qs;
            if (8 * k + 7 < G::KB || c < G::KB) {
                const float4 v = *reinterpret_cast<const float4*>(lds + G::GUARD + c * G::CS + 16 * qq);
                const uint32_t bits = (v.x < 0.0f ? 1u : 0u) | (v.y < 0.0f ? 0x100u : 0u) | (v.z < 0.0f ? 0x10000u : 0u) |
                                      (v.w < 0.0f ? 0x1000000u : 0u);
                *reinterpret_cast<uint32_t*>(hard + (size_t)c * ZC + 4 * qq) = bits;
            }
        });
    } else {
        static_for<(G::KB + 1) / 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int c = 2 * k + half2;
            if (2 * k + 1 < G::KB || c < G::KB)
                hard[(size_t)c * ZC + z2] = *reinterpret_cast<const float*>(lds + G::GUARD + 4 * z2 + c * G::CS) < 0.0f ? 1 : 0;
        });
    }
}

template <int BG, int ZC, bool ETP, int NL = BGT<BG>::ROWS, bool CRC = false> static hipError_t launch_z64s(const DecArgs& a, hipStream_t s) {
    using G = Z64S<BG, ZC, NL>;
    auto k = nrldpc_decode_z64s_kernel<BG, ZC, ETP, NL, CRC>;
    constexpr size_t lds = G::lds_bytes();
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(k, dim3(a.batch), dim3(G::THREADS), lds, s, a);
    return hipGetLastError();
}

} // inline namespace NRLDPC_UNIT
} // namespace nrldpc
#endif
