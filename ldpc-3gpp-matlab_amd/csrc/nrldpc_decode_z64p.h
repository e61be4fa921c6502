// nrldpc_decode_z64p.h -- the split decoder (nrldpc_decode_z64s.h: two threads per check row, barrier groups alternating
// between them) in the PACKED geometry (z64_packed, nrldpc_decode_z64.h): lifting sizes too small to fill a wave with one
// codeword.  A workgroup is 2 x RW waves and decodes NCW = floor(64 RW / Z) codewords at once, row lane g = z*NCW + c.  The
// per-layer code (LayerZ64 / GroupZ64 / s_crit / s_early) is the block geometry's, unchanged: only the base addresses differ.
// Replaces the run-time-Z kernel (nrldpc_decode.hip) for these sizes when every row is active and no soft output is asked for.
//
// Early termination: the codewords of a workgroup converge at different iterations.  Nothing is frozen: a codeword that passes
// its parity check has its hard decisions (and iteration count) written out AT THAT ITERATION by its own lanes, and then simply
// keeps iterating until the last codeword of the workgroup is done -- what happens to its values afterwards reaches nobody.  So
// the iteration loop has no divergent control flow (a per-lane `done` around it would wrap 80 state registers in exec-mask phis).
#ifndef NRLDPC_DECODE_Z64P_H
#define NRLDPC_DECODE_Z64P_H
#include "nrldpc_decode_z64.h"

namespace nrldpc {
inline namespace NRLDPC_UNIT { // one name space per translation unit: see NRLDPC_UNIT in nrldpc_decode_z64.h

template <int BG, int ZC, int NL = BGT<BG>::ROWS> struct Z64P : Z64<BG, ZC, 1, NL> {
    using B = Z64<BG, ZC, 1, NL>;
    static constexpr bool ILVM = z64_ilvm();          // interleaved BLOCK geometry (z64_ilv): ZC is the virtual size
    static_assert(ILVM ? !B::PACKED : (B::PACKED && B::NWV == 1), "packed geometry, or the block geometry with interleaved codewords");
    static constexpr int RW = z64p_rw(BG, ZC);       // row waves per half
    static constexpr int NCW = z64p_ncw(BG, ZC);     // codewords per workgroup
    static constexpr int NROW = ILVM ? ZC : ZC * NCW; // row lanes in use (of 64 RW; interleaved: BLK of every wave's 64)
    static constexpr int THREADS = 2 * RW * 64;
    static_assert(ILVM || (NCW >= 1 && NROW <= 64 * RW && 64 * RW - NROW < 64 && NCW + 1 <= NROW), "packed workgroup shape");
    static_assert(2 * RW <= 16 && NCW + 1 <= NROW, "workgroup shape");
    static_assert(ILVM || (B::NROWP == NROW && B::po(B::NC - 1, ZC - 1) + 4 * NROW < 65536 && B::po(B::HICOL < B::NC ? B::HICOL : 0, 0) - 4 * NROW >= 0), "LDS immediate offsets");
    static constexpr size_t FLAGS = (size_t)B::GUARD + B::CWS; // [guard][NC columns of ring | mirror][flags]
    // Interleaved block geometry: the extension-column channel LLRs live in LDS behind the flags, one int8 per extension row and
    // row lane ([row][lane] bytes, as in the split kernels of the block geometry: DecStateS, Z64S::XL), where that does not
    // cost a workgroup per CU -- without them the parity-stop build spills 130 registers at 80 VGPRs
    // flags: [NCW] "slot c's codeword has a violated check", "some slot goes on", "some slot took a new codeword"; then the slots of
    // the parity-stop builds: [NCW] codeword index, [NCW] iterations it has had (nrldpc_decode_z64p_kernel)
    static constexpr int SLOT0 = (NCW + 3 + 3) / 4 * 4; // (+ one sticky word: "the batch counter has run out")
    static constexpr size_t XOFF = FLAGS + 4 * (size_t)((SLOT0 + 2 * NCW + 3) / 4 * 4);
    static constexpr size_t XBYTES = (size_t)(B::NLT - 4) * ZC;
    static constexpr int wgs_per_cu(size_t lds) {
        const int by_waves = 24 / (2 * RW) > 0 ? 24 / (2 * RW) : 1, by_lds = (int)((160 * 1024) / lds);
        return by_waves < by_lds ? by_waves : by_lds;
    }
    static constexpr bool XL = ILVM && wgs_per_cu(XOFF + XBYTES) == wgs_per_cu(XOFF);
    static constexpr size_t lds_bytes() { return XOFF + (XL ? XBYTES : 0); }
};

// waves per SIMD the register allocation is sized for: what the LDS image lets a CU hold anyway (BG1: 4, i.e. 128 VGPRs), at most 6
template <int BG, int ZC, int NL> constexpr int z64p_wpe() {
#ifdef NRLDPC_Z64P_WPE
    return NRLDPC_Z64P_WPE;
#endif
    constexpr int by_lds = (int)((160 * 1024) / Z64P<BG, ZC, NL>::lds_bytes()) * 2 * Z64P<BG, ZC, NL>::RW / 4;
    return by_lds >= 6 ? 6 : by_lds >= 1 ? by_lds : 1;
}

// NL: the active rows 0..NL-1, a compile-time fact (all rows, or one of the pruned counts of NRLDPC_Z64P_NL_LIST)
// ILVT: the unit's interleave factor, a template argument ONLY so that it is part of the kernel's name -- the units of 128 x 2,
// 64 x 4, 32 x 8 ... all instantiate <BG, 256, ...>, a kernel handle is a weak symbol, and the linker would keep the first one's
template <int BG, int ZC, bool ETP, int NL = BGT<BG>::ROWS, int ILVT = (z64_ilvm() ? 1000 + z64_ilv() : 0)>
__global__ __launch_bounds__(2 * z64p_rw(BG, ZC) * 64, (z64p_wpe<BG, ZC, NL>())) void nrldpc_decode_z64p_kernel(const DecArgs a) {
    using G = Z64P<BG, ZC, NL>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave / G::RW, lane = tid & 63;
    const int rw = wave % G::RW;         // row wave
    const int g = G::ILVM ? rw * G::BLK + lane : rw * 64 + lane; // row lane (interleaved: a wave owns BLK consecutive ring positions)
    if constexpr (G::ILVM) {
        if constexpr (G::BLK < 64) {
            if (lane >= G::BLK) return;
        }
    } else if constexpr (G::NROW < 64 * G::RW) {
        if (g >= G::NROW) return; // these lanes own no row; barriers count waves, not lanes (no wave is empty: fewer than Z lanes retire, and the shape rule keeps that below 64)
    }
    const int z = g / G::NCW, c = g - z * G::NCW;
    const int cw = blockIdx.x * G::NCW + c;
    const bool present = cw < a.batch; // the last workgroup of a launch may hold fewer codewords: the others decode zeros
    int* flags = reinterpret_cast<int*>(lds + G::FLAGS);
    constexpr int ZR = G::ZR;          // the code's own lifting size: what the LLR and hard-decision arrays are laid out by
    constexpr size_t ncwz = (size_t)G::COLS * ZR;
    constexpr bool XF = false;
    constexpr int V = z64s_variant<BG, ZC, NL>();
    static_assert((V & SPLIT_DUAL) == 0, "no dual rows in this kernel (interleaved units are built with -DNRLDPC_Z64S_DUAL=0)");

    // base registers of the edge addresses (Z64::pb / po).  Packed geometry: the thread's word of the guard in front of column 0
    // [and the same HIOFF bytes further for the columns an immediate offset from R[0] does not reach]; opaque, so that the compiler
    // keeps them as they are.  Interleaved block geometry: the block geometry's own -- one base per ring block, RA / RB for the twins.
    uint32_t R[G::NBASE];
    uint32_t RA = 0, RB = 0;
    if constexpr (G::ILVM) {
#pragma unroll
        for (int k = 0; k < G::NBASE; ++k) R[k] = (uint32_t)G::GUARD + (uint32_t)(4 * G::BLK) * (uint32_t)((rw + k) % G::RW) + 4u * (uint32_t)lane;
        RA = (uint32_t)(G::GUARD - 4 * G::BLK) + 4u * (uint32_t)lane;
        RB = (uint32_t)G::GUARD + 4u * ZC + 4u * (uint32_t)lane;
    } else {
        R[0] = 4u * (uint32_t)g;
        if constexpr (G::NBASE > 1) {
            R[1] = R[0] + (uint32_t)G::HIOFF;
            asm volatile("" : "+v"(R[1]));
        }
    }
    // Where this thread sits, derived AGAIN from the thread id through an opaque copy: the parity pass, the write-back and the
    // refill need the codeword slot and the row position once per iteration at most, and values computed before the iteration loop
    // and used only there would stay live across it -- at 80 VGPRs the compiler parks them in scratch (as in the block geometry's
    // kernel).
    struct Where { int half, g, z, c; uint32_t home0; };
    auto where = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        Where p;
        const int wv = t >> 6, ln = t & 63;
        p.half = wv / G::RW;
        p.g = G::ILVM ? (wv % G::RW) * G::BLK + ln : (wv % G::RW) * 64 + ln;
        p.z = p.g / G::NCW;
        p.c = p.g - p.z * G::NCW;
        p.home0 = (uint32_t)G::GUARD + 4u * (uint32_t)p.g;
        return p;
    };

    // ---- core columns of codeword `cwi` -> LDS (ring and mirror), by the lanes with `wr` set (the prologue: every lane; a refill:
    // the lanes of the slots that take a new codeword); have = the codeword exists (else zeros: the lane decodes nothing).  The
    // halves take alternate columns; raw bits first, conversions after
    auto load_core = [&](const Where& p, bool wr, bool have, int cwi) {
        const size_t base = (size_t)(have ? cwi : 0) * ncwz;
        auto ingest_as = [&](auto kind_c) {
            constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
            constexpr int NPU = (G::NC + 1) / 2;
            uint32_t x[NPU];
            static_for<NPU>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int col = 2 * k + p.half;
                x[k] = 0u;
                if (wr && have && (2 * k + 1 < G::NC || col < G::NC)) {
                    if constexpr (F16) x[k] = static_cast<const uint16_t*>(a.llr)[base + (size_t)col * ZR + p.z];
                    else x[k] = static_cast<const uint32_t*>(a.llr)[base + (size_t)col * ZR + p.z];
                }
            });
            static_for<NPU>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int col = 2 * k + p.half;
                if (wr && (2 * k + 1 < G::NC || col < G::NC)) {
                    float v;
                    if constexpr (F16) v = __half2float(__ushort_as_half((unsigned short)x[k]));
                    else v = __uint_as_float(x[k]);
                    const float q = have ? ingest(v, a.scale, true) : 0.0f;
                    char* home = lds + p.home0 + col * G::CS;
                    *reinterpret_cast<float*>(home) = q;
                    if constexpr (G::ILVM) {
                        if (p.g < G::BLK) *reinterpret_cast<float*>(home + 4 * ZC) = q; // mirror of ring block 0
                    } else {
                        *reinterpret_cast<float*>(home + 4 * G::NROW) = q;
                    }
                }
            });
        };
        if (a.llr_kind == NRLDPC_K_F16) ingest_as(std::integral_constant<int, NRLDPC_K_F16>{});
        else ingest_as(std::integral_constant<int, NRLDPC_K_F32>{});
    };
    {
        Where p0;
        p0.half = half; p0.g = g; p0.z = z; p0.c = c; p0.home0 = (uint32_t)G::GUARD + 4u * (uint32_t)g;
        load_core(p0, true, present, cw);
    }
    // Parity-stop builds: a workgroup's NCW codeword SLOTS, each with the index of the codeword it holds and the iterations that one
    // has had (LDS words behind the flags).  A slot whose codeword is finished -- its checks hold, or it has had max_iter iterations --
    // writes its result and takes the next codeword of the batch from a counter in device memory (DecArgs::work); the other slots
    // keep their messages and go on.  Until round 4 a workgroup lived until the LAST of its NCW codewords stopped.
    int* slot_cw = flags + G::SLOT0;
    int* slot_it = slot_cw + G::NCW;
    if constexpr (ETP) {
        if (half == 0 && z == 0) { slot_cw[c] = present ? cw : a.batch; slot_it[c] = 0; }
        if (tid == 0) flags[G::NCW + 2] = a.work ? 0 : 1; // sticky: no counter, or the counter has run past the batch -- nothing left to take
    }

    // hard decisions of this thread's columns of codeword `cwi` (the halves take alternate columns) + the iteration count
    auto write_out = [&](const Where& p, int cwi, int it) {
        uint8_t* hard = a.hard + (size_t)cwi * ((size_t)G::KB * ZR) + p.z;
        static_for<(G::KB + 1) / 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int col = 2 * k + p.half;
            if (2 * k + 1 < G::KB || col < G::KB)
                hard[(size_t)col * ZR] = *reinterpret_cast<const float*>(lds + p.home0 + col * G::CS) < 0.0f ? 1 : 0;
        });
        if (a.iters && p.z == 0 && p.half == 0) a.iters[cwi] = it;
    };

    auto run = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        using O = Own<BG, NL, H, V>;
        DecStateS<BG, NL, H, ZC, G::XL, V> st;
#pragma unroll
        for (int i = 0; i < O::NW; ++i) st.rm[i] = 0;
        if constexpr (G::XL) { // half 0's extension rows first, then half 1's: [row][lane] bytes
            st.xp = (lds_i8_t)(lds + G::XOFF + (size_t)(H == 0 ? 0 : Own<BG, NL, 0, V>::NEXT) * ZC + g);
        } else {
#pragma unroll
            for (int i = 0; i < O::NXW; ++i) st.xq[i] = 0;
        }
        // extension LLRs of this half's rows of codeword `cwi`, by the lanes with `wr` set (see load_core)
        auto load_ext = [&](int zz, bool wr, bool have, int cwi) {
            const size_t base = (size_t)(have ? cwi : 0) * ncwz;
            auto as = [&](auto kind_c) {
                constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
                uint32_t xe[O::NEXT > 0 ? O::NEXT : 1];
                // (run-time layer count: pruned rows' extension LLRs are never used and never loaded, in blocks of 8 rows)
                static_for<(G::NLT - 4 + 7) / 8>([&](auto bc) {
                    constexpr int L0 = 4 + 8 * decltype(bc)::value;
                    constexpr int L1 = L0 + 8 < G::NLT ? L0 + 8 : G::NLT;
                    const bool used = wr && have && (!G::RT || L0 < launder(a.n_layers));
                    static_for<L1 - L0>([&](auto ic) {
                        constexpr int L = L0 + decltype(ic)::value;
                        if constexpr (O::mine(L)) {
                            constexpr int xi = O::ext_index(L);
                            const size_t i = base + (size_t)(G::NC + L - 4) * ZR + zz;
                            xe[xi] = 0u;
                            if (used) {
                                if constexpr (F16) xe[xi] = static_cast<const uint16_t*>(a.llr)[i];
                                else xe[xi] = static_cast<const uint32_t*>(a.llr)[i];
                            }
                        }
                    });
                });
                static_for<O::NEXT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    float v;
                    if constexpr (F16) v = __half2float(__ushort_as_half((unsigned short)xe[i]));
                    else v = __uint_as_float(xe[i]);
                    if (wr) st.template set_ext<i>(have ? ingest(v, a.scale, false) : 0.0f);
                });
            };
            if (a.llr_kind == NRLDPC_K_F16) as(std::integral_constant<int, NRLDPC_K_F16>{});
            else as(std::integral_constant<int, NRLDPC_K_F32>{});
        };
        load_ext(z, true, present, cw);
        __syncthreads(); // the a-posteriori rings are complete
        const float cap = (127.49f + a.beta) / a.alpha; // see LayerZ64::track3
        DecArgs av = a;
        av.beta = 8388608.0f - a.beta;
#if !NRLDPC_Z64S_RULE_SGPR
        asm volatile("" : "+v"(av.alpha), "+v"(av.beta));
#endif
        uint32_t esign_lo = 0, esign_hi = 0;
        GroupZ64<BG, ZC, 0, NL, H> g0;
        if constexpr (H == 0) {
            g0.template loads<false>(lds, R);
            g0.template track<false, XF>(st, cap);
        }
        // one iteration over the active layers (the halves alternate barrier groups: nrldpc_decode_z64s.h)
        auto iteration = [&]() {
            if constexpr (ETP) { esign_lo = 0; esign_hi = 0; }
            if constexpr (H == 0) {
                GroupZ64<BG, ZC, 0, NL, H> nx;
                s_crit<BG, ZC, NL, H, ETP, XF, 0>(g0, nx, st, lds, R, RA, RB, rw, av, cap, esign_lo, esign_hi);
                if constexpr (G::RT) { // wherever the iteration ended: group 0's early part (see the block geometry's split kernel)
                    nx.template loads<false>(lds, R);
                    nx.template track<false, XF>(st, cap);
                }
                g0 = nx;
            } else {
                s_early<BG, ZC, NL, H, ETP, XF, 0>(g0, st, lds, R, RA, RB, rw, av, cap, esign_lo, esign_hi);
            }
        };
        if constexpr (!ETP) {
            for (int it = 1; it <= a.max_iter; ++it) iteration();
            __syncthreads(); // the last group's writes
            if (present) write_out(where(), cw, a.max_iter);
        } else {
            // Refills are taken every (refill_mask + 1)-th iteration only: a refill stalls the whole workgroup for an atomic's round
            // trip and an HBM load latency, and with NCW >= 16 some slot finishes in almost every iteration -- a finished slot then
            // waits half an iteration on average, the workgroup stalls half as often (DecArgs::refill_mask: the launcher's policy)
            for (int wit = 0;; ++wit) {
                iteration();
                // parity check of this half's rows, per slot: flags[c] = "the codeword in slot c has a violated check",
                // flags[NCW] = "some slot goes on", flags[NCW + 1] = "some slot took a new codeword"
                // (raw thread ids: the lanes >= BLK of a wave have retired, so the NCW + 2 flags must fit below BLK)
                static_assert(!G::ILVM || G::BLK == 64 || G::NCW + 2 <= G::BLK, "flags are cleared by raw thread id: lanes >= BLK have retired");
                if ((int)threadIdx.x <= G::NCW + 1) flags[threadIdx.x] = 0;
                __syncthreads();
                const Where p = where();
                const int c = p.c;                      // (this shadows the prologue's copy on purpose: see `where`)
                const int cwi = slot_cw[c];             // the slot's codeword and the iterations it has had, this one included
                const int iti = slot_it[c] + 1;
                const bool empty = cwi >= a.batch;      // a free slot (waiting for its refill, or the batch has run out): its lanes decode nothing anybody reads
                const bool exhausted = flags[G::NCW + 2] != 0;
                const bool refill_now = (wit & a.refill_mask) == a.refill_mask;
                uint32_t bad = 0;
                bool stop = false; // wave-uniform: every lane's codeword is settled (violated, or out of the vote)
                auto vote = [&]() {
                    if (bad && !empty) flags[c] = 1;
                    stop = __all((int)(bad | (uint32_t)empty | (uint32_t)__atomic_load_n(&flags[c], __ATOMIC_RELAXED))) != 0;
                };
                if constexpr (G::RT) { // run-time layer count: highest row first, pruned rows skipped eight at a time (Own::parity_order_desc)
                    constexpr auto PD = O::parity_order_desc();
                    static_for<(PD.n + 7) / 8>([&](auto bc) {
                        constexpr int b0 = 8 * decltype(bc)::value, b1 = b0 + 8 < PD.n ? b0 + 8 : PD.n;
                        const int nlb = launder(a.n_layers);
                        if (!stop && PD.v[b1 - 1] < nlb) {
                            static_for<b1 - b0>([&](auto ic) {
                                constexpr int i = b0 + decltype(ic)::value;
                                constexpr int L = PD.v[i];
                                if (!stop && L < nlb) {
                                    bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                                    if constexpr ((i % 4) == 3 || i + 1 == b1 || O::ncore(L) > 10 || (i + 1 < PD.n && O::ncore(PD.v[i + 1 < PD.n ? i + 1 : i]) > 10)) vote();
                                }
                            });
                        }
                    });
                }
                constexpr auto PO = O::parity_order(); // cheapest rows first
                if constexpr (!G::RT) static_for<PO.n>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int L = PO.v[i];
                    if (!stop) {
                        bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                        if constexpr (i < 3 || (i % 4) == 3 || i + 1 == PO.n || O::ncore(L) > 10 || (i + 1 < PO.n && O::ncore(PO.v[i + 1 < PO.n ? i + 1 : i]) > 10)) vote();
                    }
                });
                if (bad && !empty) flags[c] = 1;
                __syncthreads();
                // a finished codeword's result leaves now, by its own lanes ...
                const bool fin = !empty && (flags[c] == 0 || iti >= a.max_iter);
                if (fin) write_out(p, cwi, iti);
                // ... and the first lane of half 0 of a FREE slot (just finished, or waiting since an earlier iteration) takes the next
                // codeword of the batch on a refill iteration; otherwise the slot stays free and keeps the workgroup alive until then
                const bool free_slot = fin || empty;
                if (p.half == 0 && p.z == 0) {
                    if (!free_slot) {
                        slot_it[c] = iti;
                        flags[G::NCW] = 1;
                    } else {
                        int nxt = a.batch;
                        if (!exhausted) {
                            if (refill_now) {
                                nxt = atomicAdd(a.work, 1);
                                if (nxt >= a.batch) { nxt = a.batch; flags[G::NCW + 2] = 1; }
                                else { flags[G::NCW] = 1; flags[G::NCW + 1] = 1; }
                            } else {
                                flags[G::NCW] = 1; // its refill comes
                            }
                        }
                        slot_cw[c] = nxt;
                        slot_it[c] = 0;
                    }
                }
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(flags[G::NCW]) == 0) break; // every slot is empty
                if (__builtin_amdgcn_readfirstlane(flags[G::NCW + 1]) != 0) {  // some slot starts a new codeword: its lanes load it
                    const Where q = where();
                    const int nw = slot_cw[q.c];
                    const bool refill = free_slot && nw < a.batch;
                    if (__any((int)refill)) {
                        load_core(q, refill, true, nw);
                        load_ext(q.z, refill, true, nw);
                        if (refill) {
#pragma unroll
                            for (int i = 0; i < O::NW; ++i) st.rm[i] = 0;
                        }
                    }
                    __syncthreads(); // the new codewords' rings are complete
                    if constexpr (H == 0) { // group 0's early part again: it was read before the refill
                        g0.template loads<false>(lds, R);
                        g0.template track<false, XF>(st, cap);
                    }
                }
            }
        }
    };
    if (half == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// Workgroups of one launch.  Parity-stop builds refill their slots from the batch, so a launch of them is PERSISTENT: as many
// workgroups as the device holds at once (occupancy x compute units, asked of the runtime once per kernel and device) when the
// batch has more codewords than those hold, each starting with NCW codewords and pulling the rest through DecArgs::work, which
// this function sets to the first index nobody starts with.  Fixed-iteration builds: one workgroup per NCW codewords, as before.
template <int BG, int ZC, bool ETP, int NL = BGT<BG>::ROWS> static hipError_t launch_z64p_t(const DecArgs& a, hipStream_t s) {
    using G = Z64P<BG, ZC, NL>;
    auto k = nrldpc_decode_z64p_kernel<BG, ZC, ETP, NL, (z64_ilvm() ? 1000 + z64_ilv() : 0)>;
    constexpr size_t lds = G::lds_bytes();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {};
    static int resident[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), G::THREADS, lds) != hipSuccess) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        resident[dev & 63] = per_cu * cus;
        attr_set[dev & 63] = true;
    }
    int grid = (a.batch + G::NCW - 1) / G::NCW;
    DecArgs b = a;
    static const bool no_refill = getenv("NRLDPC_NO_REFILL") != nullptr; // A/B: every workgroup decodes its own NCW codewords and leaves
    // refills every iteration for the workgroups of few codewords, every second one from 16 slots on (measured: profiles/r05_refill_period.txt);
    // NRLDPC_REFILL_MASK=0/1/3 forces every / every second / every fourth iteration (A/B)
    static const int env_mask = getenv("NRLDPC_REFILL_MASK") ? atoi(getenv("NRLDPC_REFILL_MASK")) : -1;
    b.refill_mask = env_mask >= 0 ? env_mask : (G::NCW >= 16 ? 1 : 0);
    // NRLDPC_REFILL_GRID=<n>: at most n workgroups per launch -- a TEST hook (a small batch then goes through the refill path; the
    // tests change it from call to call), looked at only in a process that was started with NRLDPC_TEST_HOOKS set: production launches
    // read no environment here, and a stray NRLDPC_REFILL_GRID cannot cap them
    static const bool test_hooks = getenv("NRLDPC_TEST_HOOKS") != nullptr;
    int cap = resident[dev & 63];
    if (test_hooks) {
        if (const char* e = getenv("NRLDPC_REFILL_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }
    }
    if (ETP && a.work && !no_refill && cap > 0 && grid > cap) {
        grid = cap;
        hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.work), grid * G::NCW, 1, s);
        if (e != hipSuccess) return e;
    } else {
        b.work = nullptr;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(G::THREADS), lds, s, b);
    return hipGetLastError();
}


// ---- the general kernel of the packed geometry: one thread per check row, run-time layer count, soft output ------------------
// What the pipelined builds above do not serve (pruned rows other than NRLDPC_Z64P_NL_LIST, per-iteration soft values): the
// unpipelined layer loop of the block geometry's general kernel (group_z64 with both twin writes, one barrier per group) on
// the packed LDS image.  A workgroup is RW waves holding the same NCW codewords.  Early termination as above: nothing is
// frozen; a codeword's outputs leave at the iteration it converges, and its soft-output stores stop there.
// Measured against the run-time-Z kernel it would replace (tools/exp_check.py with a layer count, one session): BG2 +1...+34 %
// (Z = 8 / 20 / 32: +34 / +12 / +20 %; Z = 56, 80: -5...+1 % fixed, +2...+19 % with the parity stop); BG1 -3...-17 %: one thread
// per row keeps all 80 message registers (142-161 VGPRs) and the doubled rings allow only 11 waves per CU, three per SIMD,
// where the run-time-Z kernel runs four.  So BG2 only.
template <int BG> constexpr bool z64pg_serves() { return BG == 2; }

template <int BG, int ZC> constexpr int z64pg_wpe() {
    constexpr int by_lds = (int)((160 * 1024) / Z64P<BG, ZC>::lds_bytes()) * Z64P<BG, ZC>::RW / 4;
    constexpr int cap = BG == 2 ? 6 : 4;
    return by_lds >= cap ? cap : by_lds >= 1 ? by_lds : 1;
}

// MODE 0: the general kernel described above.  MODE 1 / 2 ("row form" of the packed geometry, nrldpc_decode_z64pr_*): the same
// workgroup -- one thread per check row, RW waves, NCW whole codewords -- with the SOFTWARE-PIPELINED layer loop of the block
// geometry's one-thread-per-row kernel (pipeline_z64: fixed iteration count / parity-check stop), hard output only; NL = every
// row, or NL_RT (a run-time prefix).  For BG2's large lifting sizes that do not split into full waves (11 x 2^k: 88, 176, 352
// as 352 row lanes; 96 as 4 x 96 = 384): BG2's one-thread-per-row form fits 80 registers, so four 6-wave workgroups fill a
// CU exactly as Z = 384's do, with 92-100 % of the lanes at work instead of 69-75 %.
template <int BG, int ZC, int MODE = 0, int NL = BGT<BG>::ROWS>
__global__ __launch_bounds__(z64p_rw(BG, ZC) * 64, (z64pg_wpe<BG, ZC>())) void nrldpc_decode_z64pg_kernel(const DecArgs a) {
    using G = Z64P<BG, ZC, NL>;
    using LG = LayerGroups<BG>;
    static_assert(MODE != 0 || NL == BGT<BG>::ROWS, "the general kernel reads its layer count at run time from the all-rows tables");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int rw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid; // row lane (one thread per row: no halves)
    if constexpr (G::NROW < 64 * G::RW) {
        if (g >= G::NROW) return;
    }
    const int z = g / G::NCW, c = g - z * G::NCW;
    const int cw = blockIdx.x * G::NCW + c;
    const bool present = cw < a.batch;
    int* flags = reinterpret_cast<int*>(lds + G::FLAGS);
    constexpr size_t ncwz = (size_t)G::COLS * ZC;
    // base registers of the edge addresses (Z64::pb / po): the thread's word of the guard in front of column 0 [and the same HIOFF
    // bytes further for the columns an immediate offset from R[0] does not reach]; opaque, so that the compiler keeps them as they are
    uint32_t R[G::NBASE];
    R[0] = 4u * (uint32_t)g;
    if constexpr (G::NBASE > 1) {
        R[1] = R[0] + (uint32_t)G::HIOFF;
        asm volatile("" : "+v"(R[1]));
    }
    constexpr uint32_t RA = 0, RB = 0;                           // (the block geometry's twin bases: unused here)
    const uint32_t home0 = (uint32_t)G::GUARD + 4u * (uint32_t)g; // this thread's word of column 0's ring
    const size_t base = (size_t)(present ? cw : 0) * ncwz;
    float* app_row = (present && a.app) ? a.app + base + z : nullptr;

    DecState<BG> st;
#pragma unroll
    for (int i = 0; i < G::NW; ++i) st.rm[i] = 0;
#pragma unroll
    for (int i = 0; i < G::NXW; ++i) st.xq[i] = 0;
    {   // all loads of a thread as raw bits first, conversions after (see the block geometry's prologue)
        const int next_used = a.app ? G::NEXT : launder(a.n_layers) - 4; // a pruned row's extension LLR is never used
        auto ingest_as = [&](auto kind_c) {
            constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
            auto raw = [&](size_t i) -> uint32_t {
                if (!present) return 0u;
                if constexpr (F16) return static_cast<const uint16_t*>(a.llr)[i];
                else return static_cast<const uint32_t*>(a.llr)[i];
            };
            auto val = [&](uint32_t r) -> float {
                if constexpr (F16) return __half2float(__ushort_as_half((unsigned short)r));
                else return __uint_as_float(r);
            };
            uint32_t x[G::NC], xe[G::NEXT];
            static_for<G::NC>([&](auto cc) {
                constexpr int col = decltype(cc)::value;
                x[col] = raw(base + (size_t)col * ZC + z);
            });
            static_for<(G::NEXT + 7) / 8>([&](auto bc) {
                constexpr int i0 = decltype(bc)::value * 8;
                constexpr int i1 = i0 + 8 < G::NEXT ? i0 + 8 : G::NEXT;
                if (i0 < next_used) {
                    static_for<i1 - i0>([&](auto ic) {
                        constexpr int i = i0 + decltype(ic)::value;
                        xe[i] = raw(base + (size_t)(G::NC + i) * ZC + z);
                    });
                } else {
                    static_for<i1 - i0>([&](auto ic) { xe[i0 + decltype(ic)::value] = 0u; });
                }
            });
            static_for<G::NC>([&](auto cc) {
                constexpr int col = decltype(cc)::value;
                const float q = present ? ingest(val(x[col]), a.scale, true) : 0.0f;
                char* home = lds + home0 + col * G::CS;
                *reinterpret_cast<float*>(home) = q;
                *reinterpret_cast<float*>(home + 4 * G::NROW) = q;
            });
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                f32_to_byte<i & 3>(st.xq[i >> 2], present ? ingest(val(xe[i]), a.scale, false) : 0.0f);
            });
        };
        if (a.llr_kind == NRLDPC_K_F16) ingest_as(std::integral_constant<int, NRLDPC_K_F16>{});
        else ingest_as(std::integral_constant<int, NRLDPC_K_F32>{});
        if (app_row) { // soft output of columns whose layer is inactive = the ingested channel value
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                app_row[(size_t)(G::NC + i) * ZC] = byte_to_f32<i & 3>(st.xq[i >> 2]) * a.inv_scale;
            });
        }
    }
    __syncthreads();

    auto write_out = [&](int it) {
        uint8_t* hard = a.hard + (size_t)cw * ((size_t)G::KB * ZC) + z;
        static_for<G::NC>([&](auto cc) {
            constexpr int col = decltype(cc)::value;
            const float v = *reinterpret_cast<const float*>(lds + home0 + col * G::CS);
            if constexpr (col < G::KB) hard[(size_t)col * ZC] = v < 0.0f ? 1 : 0;
            if (app_row) app_row[(size_t)col * ZC] = v * a.inv_scale;
        });
        if (a.iters && z == 0) a.iters[cw] = it;
    };

    uint32_t esign_lo = 0, esign_hi = 0;
    bool done = !present; // per lane = per codeword
    // parity check of every codeword of the workgroup; a codeword that passes leaves now (write_out); true: nobody is left
    auto parity_and_retire = [&](int it) -> bool {
        if (tid <= G::NCW) flags[tid] = 0;
        __syncthreads();
        uint32_t bad = 0;
        bool stop = false; // wave-uniform
        static_for<G::ROWS>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            if (!stop && L < launder(a.n_layers)) {
                bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                if constexpr (L < 4 || (L % 4) == 3) {
                    if (bad && !done) flags[c] = 1;
                    stop = __all((int)(bad | (uint32_t)done | (uint32_t)__atomic_load_n(&flags[c], __ATOMIC_RELAXED))) != 0;
                }
            }
        });
        if (bad && !done) { flags[c] = 1; flags[G::NCW] = 1; }
        __syncthreads();
        if (!done && flags[c] == 0) {
            done = true;
            write_out(it);
        }
        return __builtin_amdgcn_readfirstlane(flags[G::NCW]) == 0;
    };
    if constexpr (MODE != 0) {
        constexpr bool ETP = MODE == 2;
        const float cap = (127.49f + a.beta) / a.alpha; // see LayerZ64::track3
        DecArgs av = a;                                  // alpha, 2^23 - beta as VGPR values (the one-thread-per-row kernel's choice)
        av.beta = 8388608.0f - a.beta;
        asm volatile("" : "+v"(av.alpha), "+v"(av.beta));
        GroupZ64<BG, ZC, 0, NL> g0;
        g0.template loads<false>(lds, R);
        g0.template track<false>(st, cap);
        for (int it = 1; it <= a.max_iter; ++it) {
            if constexpr (ETP) { esign_lo = 0; esign_hi = 0; }
            GroupZ64<BG, ZC, 0, NL> nx;
            pipeline_z64<BG, ZC, 0, ETP, NL>(g0, nx, st, lds, R, RA, RB, rw, av, cap, esign_lo, esign_hi);
            if constexpr (G::RT) { // wherever the iteration ended: group 0's early part (none of its edges is early then)
                nx.template loads<false>(lds, R);
                nx.template track<false>(st, cap);
            }
            g0 = nx;
            if constexpr (ETP) {
                if (parity_and_retire(it)) break;
            }
        }
        if constexpr (!ETP) __syncthreads(); // the last group's writes
        if (!done) write_out(a.max_iter);
    } else {
    for (int it = 1; it <= a.max_iter; ++it) {
        esign_lo = 0; esign_hi = 0;
        float* app_ext = done ? nullptr : app_row; // a converged codeword keeps iterating, but its soft output is final
        static_for<G::ROWS>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            if constexpr (LG::group_start(L) == L) { // L leads a barrier group
                constexpr int GE = LG::group_last(L);
                const int nl = launder(a.n_layers);
                if (L < nl) {
                    if (GE < nl) {
                        group_z64<BG, ZC, L, GE, false, false>(st, lds, R, RA, RB, rw, a, esign_lo, esign_hi, app_ext);
                    } else { // the layer count cuts this group: its active layers one by one
                        static_for<GE - L>([&](auto ic) {
                            constexpr int LL = L + decltype(ic)::value;
                            if (LL < nl) group_z64<BG, ZC, LL, LL, false, false>(st, lds, R, RA, RB, rw, a, esign_lo, esign_hi, app_ext);
                        });
                    }
                    __syncthreads();
                }
            }
        });
        if (a.early_term && parity_and_retire(it)) break;
    }
    if (!done) write_out(a.max_iter);
    }
}

template <int BG, int ZC, int MODE = 0, int NL = BGT<BG>::ROWS> static hipError_t launch_z64pg(const DecArgs& a, hipStream_t s) {
    using G = Z64P<BG, ZC, NL>;
    auto k = nrldpc_decode_z64pg_kernel<BG, ZC, MODE, NL>;
    constexpr size_t lds = G::lds_bytes();
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(k, dim3((a.batch + G::NCW - 1) / G::NCW), dim3(G::RW * 64), lds, s, a);
    return hipGetLastError();
}

// the pipelined one-thread-per-row builds of the packed geometry (MODE 1 / 2 above): hard output, any layer count
template <int BG, int ZC> static hipError_t launch_z64pr(const DecArgs& a, hipStream_t s) {
    if (a.n_layers != BGT<BG>::ROWS) return a.early_term ? launch_z64pg<BG, ZC, 2, NL_RT>(a, s) : launch_z64pg<BG, ZC, 1, NL_RT>(a, s);
    return a.early_term ? launch_z64pg<BG, ZC, 2>(a, s) : launch_z64pg<BG, ZC, 1>(a, s);
}

// hard output only, every row active or the layer count of the build (the caller checks: anything else is the run-time-Z kernel's)
template <int BG, int ZC> constexpr bool z64p_not_et() {
#define NRLDPC_Z64P_CASE(b, z) if (BG == b && ZC == z) return true;
    NRLDPC_Z64P_NOT_ET(NRLDPC_Z64P_CASE)
#undef NRLDPC_Z64P_CASE
    return false;
}
// a pruned layer count with builds of its own (NRLDPC_Z64P_NL_LIST)
template <int BG, int ZC, int NL> static hipError_t launch_z64p_pruned(const DecArgs& a, hipStream_t s) {
    return a.early_term ? launch_z64p_t<BG, ZC, true, NL>(a, s) : launch_z64p_t<BG, ZC, false, NL>(a, s);
}
template <int BG, int ZC> static hipError_t launch_z64p(const DecArgs& a, hipStream_t s) {
    if (a.n_layers != BGT<BG>::ROWS) { // a pruned layer count without a build of its own: the run-time-prefix kernels (NL_RT)
        if (!a.early_term) return launch_z64p_t<BG, ZC, false, NL_RT>(a, s);
        if constexpr (z64p_not_et<BG, ZC>()) return hipErrorInvalidValue; // not reached
        else return launch_z64p_t<BG, ZC, true, NL_RT>(a, s);
    }
    if (!a.early_term) return launch_z64p_t<BG, ZC, false>(a, s);
    if constexpr (z64p_not_et<BG, ZC>()) return hipErrorInvalidValue; // not reached: launch_decode asks has_z64p_kernel first
    else return launch_z64p_t<BG, ZC, true>(a, s);
}

} // inline namespace NRLDPC_UNIT
} // namespace nrldpc
#endif
