// nrldpc_device.h -- device-side helpers shared by the decoder kernels (gfx950).
#ifndef NRLDPC_DEVICE_H
#define NRLDPC_DEVICE_H
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "nr_bg_tables.h"
#include "nrldpc_kernels.h"

namespace nrldpc {

template <int BG> struct BGT;
template <> struct BGT<1> {
    static constexpr int ROWS = NR_BG1_ROWS, COLS = NR_BG1_COLS, KB = 22, NNZ = NR_BG1_NNZ;
    static constexpr int row_ptr(int r) { return nr_bg1_row_ptr[r]; }
    static constexpr int col(int e) { return nr_bg1_col[e]; }
};
template <> struct BGT<2> {
    static constexpr int ROWS = NR_BG2_ROWS, COLS = NR_BG2_COLS, KB = 10, NNZ = NR_BG2_NNZ;
    static constexpr int row_ptr(int r) { return nr_bg2_row_ptr[r]; }
    static constexpr int col(int e) { return nr_bg2_col[e]; }
};
template <int BG> struct BGD : BGT<BG> {
    static constexpr int NC = BGT<BG>::KB + 4;                    // core columns (LDS resident)
    static constexpr int NCP = NC | 1;                            // odd dword stride
    static constexpr int NEXT = BGT<BG>::ROWS - 4;                // extension rows / columns
    static constexpr int NCORE = BGT<BG>::NNZ - NEXT;             // core edges (messages stored)
    static constexpr int NW = (NCORE + 3) / 4;                    // message registers
    static constexpr int NXW = (NEXT + 3) / 4;                    // extension-LLR registers
    // number of core edges before row L (every row >= 4 carries exactly one extension edge, last)
    static constexpr int core_base(int L) { return BGT<BG>::row_ptr(L) - (L > 4 ? L - 4 : 0); }
};

// Barrier groups.  Layer L only needs a workgroup barrier in front of it if it shares a core column with
// a layer processed since the last barrier: column-disjoint layers touch disjoint LDS words, so running
// them back to back is exactly the sequential schedule.  From row ~20 on, consecutive rows of both base
// graphs alternate between column 0 and column 1 and are otherwise sparse, which pairs them up: BG1 needs
// 32 barriers per iteration instead of 46, BG2 28 instead of 42.  Groups are formed greedily in table
// order (the processing order is NOT changed).  NL = number of active layers (rows 0..NL-1): a pruned layer count
// known at compile time gets its own group table, so the cyclic "next group" of the last group is group 0 again.
// SG ("single"): every layer a group of its own -- always legal (a barrier in front of every layer IS the sequential schedule);
// the split kernels of the 12-wave workgroups use it (z64s_single, nrldpc_decode_z64s.h): their two halves alternate per group,
// and one-layer groups keep the two halves' shares of every interval closer than the merged pairs do.
//
// NL == NL_RT: the layer count is a RUN-TIME prefix (DecArgs::n_layers) of the all-rows tables.  Pruning is always a row
// prefix, so the groups, the ownership of layers by the halves of a split kernel and every message slot are those of the
// all-rows build; what the kernels of such a build decide at run time is where an iteration ends (scalar branches), and which
// copy of a ring's block 0 the column's next reader will read when the compile-time next reader may be pruned (LayerZ64::twins).
constexpr int NL_RT = 0;
template <int BG> constexpr int nl_rows(int NL) { return NL == NL_RT ? BGT<BG>::ROWS : NL; }

template <int BG, int NL_ = BGT<BG>::ROWS, bool SG = false> struct LayerGroups {
    using G = BGD<BG>;
    static constexpr int NL = nl_rows<BG>(NL_);
    static_assert(NL >= 4 && NL <= BGT<BG>::ROWS, "active layer count");
    static constexpr unsigned long long colmask(int L) {
        unsigned long long m = 0;
        for (int e = G::row_ptr(L); e < G::row_ptr(L + 1); ++e)
            if (G::col(e) < G::NC) m |= 1ull << G::col(e);
        return m;
    }
    struct Tab {
        int gstart[G::ROWS];              // first layer of the group containing layer L
        int glast[G::ROWS];               // last layer of that group
        int gindex[G::ROWS];              // group number of layer L
        int gfirst[G::ROWS];              // first layer of group number gi
        unsigned long long gmask[G::ROWS]; // core columns written by group number gi
        int n;                            // number of groups (all layers active)
    };
    static constexpr Tab make() {
        Tab t{};
        int start = 0, gi = -1;
        unsigned long long acc = 0;
        for (int l = 0; l < NL; ++l) {
            const unsigned long long m = colmask(l);
            if (SG || l == 0 || (acc & m)) { start = l; acc = m; ++gi; t.gfirst[gi] = l; t.gmask[gi] = 0; } else acc |= m;
            t.gstart[l] = start;
            t.gindex[l] = gi;
            t.gmask[gi] |= m;
        }
        t.n = gi + 1;
        for (int l = NL - 1; l >= 0; --l)
            t.glast[l] = (l + 1 < NL && t.gstart[l + 1] == t.gstart[l]) ? t.glast[l + 1] : l;
        return t;
    }
    static constexpr Tab T = make();
    static constexpr int group_start(int L) { return T.gstart[L]; }
    static constexpr bool group_end(int L) { return T.glast[L] == L; }
    static constexpr int group_last(int L) { return T.glast[L]; }
    static constexpr int group_index(int L) { return T.gindex[L]; }
    static constexpr int ngroups() { return T.n; }
    static constexpr int group_first(int gi) { return T.gfirst[gi]; }
    static constexpr unsigned long long group_mask(int gi) { return T.gmask[gi]; }
};

// Ownership of layers by threads.  H < 0: a thread owns its check row in every layer (one thread per row of a codeword).
// H = 0 / 1 ("split" kernels, nrldpc_decode_z64s.h): a row has TWO threads; the barrier groups alternate between them, so
// each holds the messages (and extension LLRs) of every other group only -- about half the registers, twice the waves.
// V: variant bits of a split kernel (z64s_variant<BG, ZC, NL>()): 1 = dual dense rows, 2 = one-layer barrier groups.
constexpr int SPLIT_DUAL = 1, SPLIT_SINGLE = 2;
template <int BG, int NL_, int H, int V = 0> struct Own {
    using G = BGD<BG>;
    static constexpr int NL = nl_rows<BG>(NL_); // rows the tables cover (NL_RT: all of them)
    using LG = LayerGroups<BG, NL_, (H >= 0 && (V & SPLIT_SINGLE) != 0)>;
    static constexpr bool mine(int L) { return H < 0 || LG::group_index(L) % 2 == H; }
    static constexpr int ncore(int L) { return G::row_ptr(L + 1) - G::row_ptr(L) - (L >= 4 ? 1 : 0); }
    // "Dual" rows: the dense core rows 0..3 (BG1: 19 edges each, a quarter of all edges, each a barrier group of its own) are
    // worked on by BOTH threads of a check row at once -- the edges alternate between the halves, each half searches its own
    // edges for the two smallest magnitudes and the sign parity, the two partial results are exchanged through LDS (8 bytes
    // per thread, one extra barrier), and each half then updates its own edges.  Without this one half carries a dense row
    // alone while the other finishes the next row's few early edges and idles.
    // V & SPLIT_DUAL: the kernel variant works its dense rows dually (z64s_dual<BG, ZC, NL>(): where the exchange buffer costs no workgroup)
    static constexpr bool dual(int L) { return (V & SPLIT_DUAL) != 0 && H >= 0 && L < 4 && L < NL && ncore(L) >= 12; }
    static constexpr bool owned(int L, int j) { return dual(L) ? (j % 2 == H) : mine(L); }
    static constexpr int ncore_own(int L) {
        int n = 0;
        for (int j = 0; j < ncore(L); ++j) n += owned(L, j) ? 1 : 0;
        return n;
    }
    // index of the first message byte of layer L in this thread's compact message store
    static constexpr int core_base(int L) {
        if (H < 0) return G::core_base(L);
        int n = 0;
        for (int l = 0; l < L; ++l) n += ncore_own(l);
        return n;
    }
    // index of layer L's extension LLR (L >= 4) in this thread's store
    static constexpr int ext_index(int L) {
        if (H < 0) return L - 4;
        int n = 0;
        for (int l = 4; l < L; ++l) n += mine(l) ? 1 : 0;
        return n;
    }
    // this thread's rows among the NL active ones, cheapest (fewest core edges) first: the order of the parity pass, which
    // stops reading at the first violated check of its wave -- far from convergence almost any row fails, and a
    // two-edge row costs 2 LDS reads where row 0 costs 19
    struct Order { int v[BGT<BG>::ROWS]; int n; };
    static constexpr Order parity_order() {
        Order o{};
        o.n = 0;
        for (int l = 0; l < NL; ++l)
            if (mine(l)) o.v[o.n++] = l;
        for (int i = 1; i < o.n; ++i) {
            const int x = o.v[i];
            int j = i - 1;
            while (j >= 0 && ncore(o.v[j]) > ncore(x)) { o.v[j + 1] = o.v[j]; --j; }
            o.v[j + 1] = x;
        }
        return o;
    }
    // ... and highest row first: the parity-pass order of the builds with a run-time layer count.  The rows a layer count prunes
    // are then the FIRST ones of the list, in whole blocks that one scalar branch skips (a taken branch per pruned row was a
    // third of those builds' distance to the compile-time ones under the parity stop), and what is left runs cheap rows (the
    // high, sparse ones) before the dense core rows all the same.
    static constexpr Order parity_order_desc() {
        Order o{};
        o.n = 0;
        for (int l = NL - 1; l >= 0; --l)
            if (mine(l)) o.v[o.n++] = l;
        return o;
    }
    static constexpr int NCORE = H < 0 ? G::NCORE : core_base(NL);
    static constexpr int NEXT = H < 0 ? G::NEXT : ext_index(NL);
    static constexpr int NW = (NCORE + 3) / 4, NXW = (NEXT + 3) / 4;
};

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ uint32_t fbits(float x) { return __float_as_uint(x); }

// Schedule tables are read through the constant address space so that every access is a scalar
// (s_load) instruction: the index is compile-time, the base wave-uniform.
typedef const int32_t __attribute__((address_space(4))) * ctab_t;
__device__ __forceinline__ ctab_t as_ctab(const int32_t* p) { return reinterpret_cast<ctab_t>(reinterpret_cast<uintptr_t>(p)); }
// Opaque identity on a wave-uniform value: stops LLVM hoisting iteration-invariant scalar loads,
// write addresses and layer predicates out of the iteration loop (which costs >230 VGPRs + SGPR spills).
__device__ __forceinline__ ctab_t launder(ctab_t p) {
    uintptr_t v = reinterpret_cast<uintptr_t>(p);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<ctab_t>(v);
}
__device__ __forceinline__ int launder(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// signed byte B of w -> float (one SDWA VALU op)
template <int B> __device__ __forceinline__ float byte_to_f32(uint32_t w) {
    float f;
    if constexpr (B == 0)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f) : "v"(w));
    else if constexpr (B == 1)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f) : "v"(w));
    else if constexpr (B == 2)
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f) : "v"(w));
    else
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f) : "v"(w));
    return f;
}
// (int)r -> byte B of w, other bytes preserved (one SDWA VALU op); r is an integer-valued float in [-127,127]
template <int B> __device__ __forceinline__ void f32_to_byte(uint32_t& w, float r) {
    if constexpr (B == 0)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(r));
    else if constexpr (B == 1)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(r));
    else if constexpr (B == 2)
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(r));
    else
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w) : "v"(r));
}

template <int DT> __device__ __forceinline__ float load_llr(const void* p, size_t i) {
    if constexpr (DT == NRLDPC_K_F16)
        return __half2float(static_cast<const __half*>(p)[i]);
    else
        return static_cast<const float*>(p)[i];
}

// channel LLR -> fixed-point grid (integer-valued float).  Mirrors ingest() of the oracle.
__device__ __forceinline__ float ingest(float x, float scale, bool core) {
    float y = x * scale;
    y = (y != y) ? 0.0f : y;
    y = fminf(fmaxf(y, -127.0f), 127.0f);
    y = rintf(y) + 0.0f; // +0.0f canonicalises -0
    if (core && fabsf(x) == __builtin_inff()) y = copysignf(1048576.0f, x);
    return y;
}

// Row minimum m -> message magnitude clamp(rint(alpha*m - beta), 0, 127), alpha*m - beta rounded ONCE (nearest, ties to
// even), as scale_mag() of the oracle: the rounding is done by the fused multiply-add itself.  2^23 - beta is exact
// because nrldpc_create keeps beta on a grid of half fixed-point units; alpha*m + (2^23 - beta) then rounds at an ulp
// of 1, i.e. to 2^23 + rint(alpha*m - beta); anything below 2^23 is a negative magnitude and is clamped (m <= 2^20, so
// the sum stays below 2^24).  v_fma + v_med3 + v_sub.
__device__ __forceinline__ float scale_mag(const DecArgs& a, float m) {
    const float y = __builtin_fmaf(a.alpha, m, 8388608.0f - a.beta);
    return __builtin_amdgcn_fmed3f(y, 8388608.0f, 8388608.0f + 127.0f) - 8388608.0f;
}

template <int BG> struct DecState {
    uint32_t rm[BGD<BG>::NW];  // check-to-variable messages, int8 x4
    uint32_t xq[BGD<BG>::NXW]; // extension-column channel LLRs, int8 x4
    // channel LLR of this thread's XI-th extension bit (XF: from the float copy)
    template <int XI, bool XF> __device__ __forceinline__ float ext() const {
        if constexpr (XF) return xf[XI];
        else return byte_to_f32<XI & 3>(xq[XI >> 2]);
    }
    // the same LLRs as floats, for the builds whose register budget has room for them (one v_cvt_f32_i32_sdwa per
    // extension row and iteration saved); never touched -- so never allocated -- by the others
    float xf[BGD<BG>::NEXT];
};

// per-thread state of a split kernel's half H (see Own).  The extension-column channel LLRs live in LDS, one int8 per
// row and thread ([row][thread]: a wave reads 64 consecutive bytes, conflict-free) -- they are read once per row and
// iteration (ds_read_i8 + v_cvt_f32_i32, the cost of the SDWA unpack they replace), and the 6 registers they would take
// are what the 80-VGPR budget of 6 waves per SIMD is short of: with them in registers the compiler spilled message words
// to scratch, and the scratch traffic showed up as 1.5x the compulsory HBM bytes.
typedef int8_t __attribute__((address_space(3))) * lds_i8_t;
template <int BG, int NL, int H, int XS, bool XL, int V> struct DecStateS;
template <int BG, int NL, int H, int XS, int V> struct DecStateS<BG, NL, H, XS, true, V> { // extension LLRs in LDS
    uint32_t rm[Own<BG, NL, H, V>::NW];
    lds_i8_t xp; // this thread's byte of its half's extension row 0; row XI is XI * XS bytes further (an immediate offset)
    template <int XI, bool XF> __device__ __forceinline__ float ext() const {
        static_assert(!XF, "the split kernels keep no float copy of the extension LLRs");
        return (float)(int)xp[XI * XS];
    }
    template <int XI> __device__ __forceinline__ void set_ext(float q) { xp[XI * XS] = (int8_t)(int)q; }
};
template <int BG, int NL, int H, int XS, int V> struct DecStateS<BG, NL, H, XS, false, V> { // extension LLRs in registers, int8 x4
    uint32_t rm[Own<BG, NL, H, V>::NW];
    uint32_t xq[Own<BG, NL, H, V>::NXW > 0 ? Own<BG, NL, H, V>::NXW : 1];
    template <int XI, bool XF> __device__ __forceinline__ float ext() const {
        static_assert(!XF, "the split kernels keep no float copy of the extension LLRs");
        return byte_to_f32<XI & 3>(xq[XI >> 2]);
    }
    template <int XI> __device__ __forceinline__ void set_ext(float q) { f32_to_byte<XI & 3>(xq[XI >> 2], q); }
};

// ---- CRC-aided stop (nrldpc_cfg.early_term = 2; SURVEY 8f row N2, NRLDPCDecoder.m:298-301,336) ----------------------------------
// A codeword also stops iterating when the CRC over its first crc_bits hard decisions holds although some parity check does
// not (the information bits are usually right an iteration before the last parity bit is).  The CRC is linear: the remainder
// of the block is the XOR of x^(n-1-i) mod g over its set bits i, a table the host builds once (DecArgs::crc_tab).  Every thread
// folds the bits it can read from its own ring positions, the lanes of a wave XOR their partial remainders into 16 LDS words
// (ds_xor_b32: lanes that own no row simply take no part), and after the barrier the parity pass needs anyway every thread
// reads the 16 words back.  An all-zero block has remainder 0 whatever it means -- punctured systematic bits whose a-posteriori
// value is still 0 decide "0" -- so a block passes only if at least one of its bits is set (slot 16).
struct CrcFold {
    uint32_t acc = 0, any = 0;
    __device__ __forceinline__ void bit(float app, const uint32_t* __restrict__ tab, int pos, int nbits) {
        if (pos < nbits) {
            const uint32_t neg = 0u - (fbits(app) >> 31); // all ones when the hard decision is 1
            acc ^= tab[pos] & neg;
            any |= neg;
        }
    }
    __device__ __forceinline__ void publish(int* slots) const {
        if (acc) atomicXor(&slots[threadIdx.x & 15], (int)acc);
        if (any) slots[16] = 1;
    }
};
__device__ __forceinline__ bool crc_holds(const int* slots) {
    const int4 a = *reinterpret_cast<const int4*>(slots), b = *reinterpret_cast<const int4*>(slots + 4),
               c = *reinterpret_cast<const int4*>(slots + 8), d = *reinterpret_cast<const int4*>(slots + 12);
    const int r = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    return r == 0 && slots[16] != 0;
}

// The same for the software-pipelined builds, with nbeta23 = 2^23 - beta held in a VGPR: v_fma + v_max + v_sub
// (2.5 + 4 + 2.3 issue cycles; a separate v_rndne would cost 4 more).  The caller's search started from
// (127.49 + beta)/alpha, so no upper clamp is needed.
__device__ __forceinline__ float scale_mag_magic(float alpha, float nbeta23, float m) {
    return fmaxf(__builtin_fmaf(alpha, m, nbeta23), 8388608.0f) - 8388608.0f;
}

} // namespace nrldpc
#endif
