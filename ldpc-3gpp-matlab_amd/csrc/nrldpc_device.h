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
template <int BG, int NL = BGT<BG>::ROWS> struct LayerGroups {
    using G = BGD<BG>;
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
            if (l == 0 || (acc & m)) { start = l; acc = m; ++gi; t.gfirst[gi] = l; t.gmask[gi] = 0; } else acc |= m;
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
template <int BG, int NL, int H> struct Own {
    using G = BGD<BG>;
    using LG = LayerGroups<BG, NL>;
    static constexpr bool mine(int L) { return H < 0 || LG::group_index(L) % 2 == H; }
    static constexpr int ncore(int L) { return G::row_ptr(L + 1) - G::row_ptr(L) - (L >= 4 ? 1 : 0); }
    // index of the first message byte of layer L in this thread's compact message store
    static constexpr int core_base(int L) {
        if (H < 0) return G::core_base(L);
        int n = 0;
        for (int l = 0; l < L; ++l) n += mine(l) ? ncore(l) : 0;
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

// ---- rate recovery inside the decoder's prologue (llr_kind == NRLDPC_K_RR) ------------------------------------------------
// One decoder-input value straight from the demodulator's LLRs g_tilde: code_block_concatenation + bit_interleaving +
// bit_selection with soft combining of repetitions + the HARQ buffer + the core's input conventions
// (NRLDPCDecoder.m:143-242, 262-264), the arithmetic of nrldpc_rate_recover_kernel (nrldpc_ratematch.hip) per position, in
// the same fp32 summation order (repetitions in ascending k, then the buffer).  The arguments live in device memory and are
// read with scalar loads (wave-uniform addresses).
struct RrBlock {
    const float* f; // g_tilde segment of this code block
    float* hb;      // its HARQ buffer row or null
    int lo_f, hi_f, F, P, nfk0, E, rows, Qm, N_cb, Z2;
};
typedef const RmArgs __attribute__((address_space(4))) * rr_ctab_t;
__device__ __forceinline__ RrBlock rr_block(const RmArgs* rrp, int blk) {
    rr_ctab_t a = reinterpret_cast<rr_ctab_t>(reinterpret_cast<uintptr_t>(rrp));
    RrBlock b;
    const int C = a->C, tb = blk / C, r = blk - tb * C;
    b.Z2 = 2 * a->Z;
    b.N_cb = a->N_cb; b.Qm = a->Qm;
    b.lo_f = a->Kp - b.Z2 > 0 ? a->Kp - b.Z2 : 0;
    b.hi_f = a->K - b.Z2;
    const int f_hi = b.hi_f < b.N_cb ? b.hi_f : b.N_cb;
    b.F = f_hi > b.lo_f ? f_hi - b.lo_f : 0;
    b.P = b.N_cb - b.F;
    int c0 = a->k0 - b.lo_f;
    c0 = c0 < 0 ? 0 : (c0 > b.F ? b.F : c0);
    b.nfk0 = a->k0 - c0;
    b.E = a->E[r];
    b.rows = b.E > 0 ? b.E / b.Qm : 1;
    b.f = a->g + (size_t)tb * a->G + a->off[r];
    b.hb = a->harq ? a->harq + (size_t)blk * b.N_cb : nullptr;
    return b;
}
// n: index into the decoder's input (0 .. ncols*Z-1); the first 2Z positions are the punctured columns (:262)
__device__ __forceinline__ float rr_value(const RrBlock& b, int n) {
    const int p = n - b.Z2;
    if (p < 0) return 0.0f;
    if (p >= b.lo_f && p < b.hi_f) return __builtin_inff(); // filler (:224, :264)
    if (p >= b.N_cb) return 0.0f;                            // beyond the (limited) circular buffer
    int c = p - b.lo_f;
    c = c < 0 ? 0 : (c > b.F ? b.F : c);
    int q = p - c - b.nfk0; // index among the buffer's non-filler positions counted from k_0
    q += q < 0 ? b.P : 0;
    float val = 0.0f;
    for (int k = q; k < b.E; k += b.P) { // soft combining of repetitions, ascending k (:229-231)
        int i = 0;
        for (int m = 1; m < b.Qm; ++m) i += (k >= m * b.rows); // k / rows
        val += b.f[(k - i * b.rows) * b.Qm + i];               // e(i*rows + j) = f(i + j*Qm)  (:191-195)
    }
    if (b.hb) { // :236-239
        val += b.hb[p];
        b.hb[p] = val;
    }
    return val;
}

// The same for N positions of one thread at once, arranged so that the prologue does not become a chain of dependent memory
// round trips: without repetition (E_r <= non-filler positions of the buffer: a position takes one e(k) or none) all
// index arithmetic comes first, then all loads are issued back to back, then the HARQ buffer's, then the values are
// handed over.  pos(i) = decoder-input index of item i (negative: skip the item), put(i, value).  With repetition the
// per-position walk of rr_value is used (rare: code rates below the mother code's).
template <int N, class PosF, class PutF>
__device__ __forceinline__ void rr_gather_n(const RrBlock& b, PosF&& pos, PutF&& put);
template <int N, class PosF, class PutF>
__device__ __forceinline__ void rr_gather(const RrBlock& b, PosF&& pos, PutF&& put) {
    if constexpr (N > 0) rr_gather_n<N>(b, pos, put);
}
template <int N, class PosF, class PutF>
__device__ __forceinline__ void rr_gather_n(const RrBlock& b, PosF&& pos, PutF&& put) {
    if (b.E > b.P) { // wave-uniform
        static_for<N>([&](auto ic) {
            const int n = pos(ic);
            if (n >= 0) put(ic, rr_value(b, n));
        });
        return;
    }
    int idx[N], pp[N]; // idx: index into f or -1 (nothing received) / -2 (filler); pp: index into the HARQ buffer or -1
    static_for<N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int n = pos(ic);
        const int p = n - b.Z2;
        const bool fill = p >= b.lo_f && p < b.hi_f;
        const bool inbuf = n >= 0 && p >= 0 && p < b.N_cb && !fill;
        int c = p - b.lo_f;
        c = c < 0 ? 0 : (c > b.F ? b.F : c);
        int q = p - c - b.nfk0;
        q += q < 0 ? b.P : 0;
        int r = 0;
        for (int m = 1; m < 8; ++m) r += (m < b.Qm && q >= m * b.rows) ? 1 : 0; // q / rows (Q_m <= 8)
        idx[i] = (n >= 0 && fill) ? -2 : (inbuf && q < b.E) ? (q - r * b.rows) * b.Qm + r : -1;
        pp[i] = inbuf ? p : -1;
        if (n < 0) { idx[i] = -3; pp[i] = -1; }
    });
    float v[N];
    static_for<N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        v[i] = idx[i] >= 0 ? b.f[idx[i]] : 0.0f;
    });
    if (b.hb) { // :236-239
        float h[N];
        static_for<N>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            h[i] = pp[i] >= 0 ? b.hb[pp[i]] : 0.0f;
        });
        static_for<N>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (pp[i] >= 0) {
                v[i] += h[i];
                b.hb[pp[i]] = v[i];
            }
        });
    }
    static_for<N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (idx[i] != -3) put(ic, idx[i] == -2 ? __builtin_inff() : v[i]);
    });
}

template <int BG> struct DecState {
    uint32_t rm[BGD<BG>::NW];  // check-to-variable messages, int8 x4
    uint32_t xq[BGD<BG>::NXW]; // extension-column channel LLRs, int8 x4
    // the same LLRs as floats, for the builds whose register budget has room for them (one v_cvt_f32_i32_sdwa per
    // extension row and iteration saved); never touched -- so never allocated -- by the others
    float xf[BGD<BG>::NEXT];
};

// per-thread state of a split kernel's half H (see Own)
template <int BG, int NL, int H> struct DecStateS {
    uint32_t rm[Own<BG, NL, H>::NW];
    uint32_t xq[Own<BG, NL, H>::NXW];
    float xf[Own<BG, NL, H>::NEXT > 0 ? Own<BG, NL, H>::NEXT : 1];
};

// The same for the software-pipelined builds, with nbeta23 = 2^23 - beta held in a VGPR: v_fma + v_max + v_sub
// (2.5 + 4 + 2.3 issue cycles; a separate v_rndne would cost 4 more).  The caller's search started from
// (127.49 + beta)/alpha, so no upper clamp is needed.
__device__ __forceinline__ float scale_mag_magic(float alpha, float nbeta23, float m) {
    return fmaxf(__builtin_fmaf(alpha, m, nbeta23), 8388608.0f) - 8388608.0f;
}

} // namespace nrldpc
#endif
