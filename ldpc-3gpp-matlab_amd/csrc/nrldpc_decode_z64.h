// nrldpc_decode_z64.h -- compile-time-Z specialisation of the layered NMS-Q decoder.  Described for Z a multiple
// of 64; for other Z every "64" below is the block size B = z64_blk(Z) < 64 and lanes B..63 of each wave retire.
// Instantiated once per (BG, Z) by nrldpc_decode_z64_inst.hip (compiled with -DNRLDPC_Z64_BG / -DNRLDPC_Z64_Z).
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

#ifndef NRLDPC_DECODE_Z64_H
#define NRLDPC_DECODE_Z64_H
#include "nrldpc_device.h"

// Everything in this header and in nrldpc_decode_z64s.h / _z64p.h depends on the -D flags of the translation unit that includes it
// (NRLDPC_Z64_PACK, NRLDPC_Z64_ILV, NRLDPC_Z64S_DUAL, ...: Z64<1, 256> is the block geometry of Z = 256 in one unit and four interleaved
// codewords of Z = 64 in another), so the same template names would mean different things in different units of one library -- a
// violation of the one-definition rule that only inlining hid.  Each unit therefore puts these headers' contents into an
// inline name space of its own, named by build.py after the unit's object file: distinct symbols, same spelling at the point of use.
#ifndef NRLDPC_UNIT
#define NRLDPC_UNIT u_default
#endif

// scheduling experiments of the software pipeline (see pipeline_z64 / LayerZ64::track3); 0 = the shipped schedule
#ifndef NRLDPC_Z64_POSTBAR
#define NRLDPC_Z64_POSTBAR 0
#endif
#ifndef NRLDPC_Z64_DEFER
#define NRLDPC_Z64_DEFER 0
#endif
#ifndef NRLDPC_Z64_DEFER_EXT
#define NRLDPC_Z64_DEFER_EXT 0
#endif

// lane >= T of the executing wave, T a compile-time constant in 1..63: a constant exec mask applied by scalar instructions
// (s_and_saveexec_b64), no VALU work.  The two halves of the mask are made opaque 32-bit scalars on purpose: as a 64-bit
// constant the compiler of this ROCm release materialises it with s_mov_b64 and a 32-bit literal both for values that need
// zero extension (0x00000000ffffffff) and for values that need sign extension (0xfffffffffffffc00); the hardware does one of
// the two, and the packed kernels decoded wrongly until the halves were separated (tools/dbg_packed.py found it: bit errors
// after ONE iteration that later iterations mostly repaired -- tests/test_decode_gpu.py::test_packed_kernels_iteration_by_iteration).
#ifdef NRLDPC_EXP_LANE_CMP
#define NRLDPC_LANE_GE(T) ((int)(threadIdx.x & 63) >= (T))
#else
#define NRLDPC_LANE_GE(T) nrldpc::lane_ge<(T)>()
#endif

namespace nrldpc {
inline namespace NRLDPC_UNIT { // one name space per translation unit: see NRLDPC_UNIT in nrldpc_decode_z64.h

template <int T> __device__ __forceinline__ bool lane_ge() {
    static_assert(T >= 1 && T <= 63, "a proper subset of the wave");
    uint32_t lo = T < 32 ? (~0u << (T & 31)) : 0u, hi = T <= 32 ? ~0u : (~0u << ((T - 32) & 31));
    asm("" : "+s"(lo));
    asm("" : "+s"(hi));
    return __builtin_amdgcn_inverse_ballot_w64(((uint64_t)hi << 32) | lo);
}

constexpr int z64_set_index(int Z) {
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9; ++k)
            if (nr_lifting_sets[s][k] == Z) return s;
    return -1;
}

// "Packed" geometry (nrldpc_decode_z64p.h): lifting sizes that do not fill waves with one codeword's rows in blocks.  A
// workgroup's row lanes g = 0 .. Z*NCW-1 (RW waves of 64 per half) carry NCW whole codewords, codeword index fastest:
// g = z*NCW + c.  A column of the workgroup's LDS image is [ring: Z*NCW words][mirror: the same again], word g + P*NCW holding
// ring position (z + P) of codeword c -- consecutive lanes touch consecutive words (no bank conflicts for any Z), the rotation
// is an instruction immediate exactly as in the block geometry (its "block" is the whole ring: one base address per thread),
// and reads never wrap because the mirror is a full copy.  The twin writes that keep the two copies coherent are owed by
// every wave here, each by PART of its lanes (the rows whose run wrapped, or the others): the lane sets are compile-time
// constants per (shift, wave), applied as exec masks by scalar instructions (__builtin_amdgcn_inverse_ballot_w64: s_mov_b64 +
// s_and_saveexec_b64, no VALU work), so the columns need no guard or pad words between them.
// The geometry is a property of the translation unit: nrldpc_decode_z64p_inst.hip defines NRLDPC_Z64_PACK, the block-geometry
// units do not, so one (BG, Z) can have builds in both (nrldpc_decode.hip: the packed one serves hard-output calls with every
// row active where it is the faster, NRLDPC_Z64P_LIST; pruned rows / soft output stay with the block or run-time-Z kernels).
constexpr bool z64_packed(int) {
#ifdef NRLDPC_Z64_PACK
    return NRLDPC_Z64_PACK != 0;
#else
    return false;
#endif
}
// "Interleaved" block geometry (round 4; nrldpc_decode_z64p.h runs it): NCW whole codewords of a SMALL lifting size Zr share one
// workgroup of the BLOCK geometry of the virtual size ZC = Zr * NCW -- row lane g = z * NCW + c as in the packed geometry, but the
// ring of a column is the block geometry's [guard][ZC words][64-word mirror of block 0]: a base-graph shift P of the real code is
// the shift P * NCW of the virtual one, a compile-time constant like any other, so base registers per ring block, immediates,
// the twin analysis (one extra store per edge in one or two waves instead of a masked one in every wave), LDS size (no doubled
// rings) are those of Z = 256 / 384 / 240 ...: the shapes that run fastest.  -DNRLDPC_Z64_ILV=<NCW> with -DNRLDPC_Z64_Z=<ZC>.
constexpr int z64_ilv() {
#ifdef NRLDPC_Z64_ILV
    return NRLDPC_Z64_ILV;
#else
    return 1;
#endif
}
// ... NCW = 1 included (-DNRLDPC_Z64_ILV=1): one codeword of the lifting size itself in that kernel.  Measured for Z = 144 ... 384
// against the split / row kernels of nrldpc_decode_z64s.h / this file: within +-3 % or slower everywhere except BG1 Z = 160
// with the parity stop and BG2 Z = 176 (profiles/r04_ilv_ab.txt) -- the interleaved units gain from their shape, not from the kernel
constexpr bool z64_ilvm() {
#ifdef NRLDPC_Z64_ILV
    return true;
#else
    return false;
#endif
}

// Row waves per half of a packed workgroup: 1, 2 or 4 -- the first of these that fills 80 % of its lanes, else the best filled.
// Measured (one session each, fixed 25 / parity stop): for Z <= 32 two-wave halves gain 2-10 % at fixed iterations where they
// fill more lanes but lose 10-25 % with the parity stop on BG1 (a workgroup lives until its last codeword converges, and
// holds twice as many), so there two waves are taken only below 80 % (Z = 22, 24); three-wave halves (6-wave workgroups: 2 + 2
// + 1 + 1 waves on the four SIMDs) lose 14-48 % everywhere; four-wave halves are fine (Z = 72, 80); five and more (BG2 Z = 52
// ... 352 with 6-8 codewords or 10-16 waves per workgroup) lose to the block geometry.  BG1's last column would also lie
// beyond the 64 KB an LDS instruction's immediate offset reaches with more than 4.
constexpr int z64_blk(int Z);
constexpr int z64p_rw(int BG, int Z) {
#ifdef NRLDPC_Z64P_RW
    return NRLDPC_Z64P_RW;
#endif
    (void)BG;
    if (z64_ilvm()) return Z / z64_blk(Z); // interleaved block geometry: the virtual size's waves per codeword
    // Large lifting sizes of BG1 that do not split into full waves in the block geometry (11 or 3 times a power of two: blocks of
    // 44-48 rows leave 25-31 % of the lanes idle): 6-wave halves -- the 12-wave workgroup shape of Z = 384's split kernel -- carry
    // 352 rows (4 x 88, 2 x 176, 1 x 352) or 384 (4 x 96).  Measured against the kernels these sizes ran before, one session
    // (profiles/r04_packed_large.json, edge updates per ns): Z = 88 2507 -> 2830, 96 2730 -> 3247, 176 2905 -> 3035, 352 2412 ->
    // 3001.  Not adopted after the same measurement: 5-wave halves (Z = 144, 160, 288, 320: 10-wave workgroups, 1522-1994 against
    // 2708-3155) and every BG2 size (88 ... 352: 2278-2780 against 2847-3266 -- BG2's one-thread-per-row form already runs 6
    // waves per SIMD with 80 registers, and the packed image doubles its rings).
    if (BG == 1 && (Z == 88 || Z == 96 || Z == 176 || Z == 352)) return 6;
    int best = 1, fill = -1;
    for (int rw = 1; rw <= 4; rw *= 2) {
        const int f = (64 * rw / Z) * Z * 1000 / (64 * rw);
        if (f >= 800) return rw;
        if (f > fill) { fill = f; best = rw; }
    }
    return best;
}
constexpr int z64p_ncw(int BG, int Z) { return z64_ilvm() ? z64_ilv() : 64 * z64p_rw(BG, Z) / Z; } // codewords per packed workgroup

// Rows per wave ("block"): 64 when 64 | Z, else the largest divisor of Z below 64 that is a multiple of 4.
// A wave then owns B consecutive rows and its lanes B..63 retire at kernel entry (Z = 240 -> 4 waves of 60
// rows, 94 % of the lanes busy; the run-time-Z kernel fills every lane but pays 12 VALU cycles per edge for
// the ring address and runs at 0.55-0.7 of this kernel's rate).
constexpr int z64_blk(int Z) {
    if (z64_packed(Z)) return Z; // the whole ring: every shift is an offset from one base address
    if (Z % 64 == 0) return 64;
    if (z64_ilvm()) { // interleaved units read and write single words only: any divisor will do (252 = 4 x 63, 220 = 4 x 55)
        for (int b = 63; b >= 4; --b)
            if (Z % b == 0) return b;
    }
    for (int b = 60; b >= 4; b -= 4)
        if (Z % b == 0) return b;
    return 0;
}
constexpr int z64_nwv(int Z) { return Z / z64_blk(Z); } // waves per codeword

// Codewords per workgroup and waves per SIMD the register allocation is sized for (second
// __launch_bounds__ argument in HIP), by waves per codeword: measured optima on MI355X (tools/exp_z64.sh
// builds one variant, tools/bench_one.py times it).  What decides: waves resident per CU (BG1 needs ~128 VGPRs
// -> 4 waves per SIMD = 16 per CU when the codeword's wave count divides into it, else 3; BG2 fits 80 VGPRs
// -> 6 per SIMD), the 160 KB of LDS, an even spread of a workgroup's waves over the 4 SIMDs, barrier width.
template <int BG, int ZC> constexpr int z64_ncwg() {
#ifdef NRLDPC_Z64_NCWG
    return NRLDPC_Z64_NCWG;
#endif
    constexpr int n = z64_nwv(ZC);
    if (BG == 1) return n == 8 ? 2 : n == 6 ? 2 : n == 5 ? 3 : n == 4 ? 2 : n == 3 ? 1 : 2;
    return n == 8 ? 1 : n == 6 ? 2 : n == 5 ? 1 : n == 4 ? 3 : n == 3 ? 4 : n == 2 ? 2 : 4;
}

// Pruned layer counts (NRLDPC_Z64_NL_LIST) carry less message state.  Measured on MI355X (tools/exp_z64.sh with a
// layer count): BG2 keeps its shape (one-codeword workgroups lose 10-20 %); BG1 at 5 rows (R = 8/9) gains 13-15 % as
// one-codeword workgroups, three per CU (LDS-bound), at a register budget of 6 waves per SIMD.
template <int BG, int ZC, int NL> constexpr int z64_ncwg_nl() {
#ifdef NRLDPC_Z64_NCWG
    return NRLDPC_Z64_NCWG;
#endif
    return (BG == 1 && z64_nwv(ZC) == 6 && NL != NL_RT && NL <= 6) ? 1 : z64_ncwg<BG, ZC>();
}

// Extension LLRs as floats in VGPRs (DecState::xf) for the fixed-iteration builds with register room: BG1 shapes sized
// for 3 waves per SIMD (168 VGPRs allowed, ~130 used) with every row active.
template <int BG, int ZC, int NCWG, int NL> constexpr bool z64_ext_float();

// Early-termination builds: the codewords of a workgroup leave the decoding loop at different iterations and the
// workgroup lives until the last one; with one codeword per workgroup nothing waits.  Measured at waterfall points
// (QPSK/AWGN, mean 7.4-7.9 iterations): BG1 four-wave codewords (Z = 256) +9 %; six-wave codewords lose 12-50 % as
// one-codeword workgroups (uneven spread over the 4 SIMDs), so they keep the fixed-iteration shape.
template <int BG, int ZC> constexpr int z64_ncwg_et() {
#ifdef NRLDPC_Z64_NCWG
    return NRLDPC_Z64_NCWG;
#endif
    return (BG == 1 && z64_nwv(ZC) == 4) ? 1 : z64_ncwg<BG, ZC>();
}

template <int BG, int ZC, int NCWG, int NL = BGT<BG>::ROWS> constexpr int z64_wpe() {
#ifdef NRLDPC_Z64_WPE
    return NRLDPC_Z64_WPE;
#endif
    if (BG == 1 && z64_nwv(ZC) == 6 && NL != NL_RT && NL <= 6) return 6;
    return BG == 2 ? 6 : (z64_nwv(ZC) == 8 || z64_nwv(ZC) == 5 || z64_nwv(ZC) == 4 || z64_nwv(ZC) == 3) ? 4 : 3;
}

// NL: active layers (rows 0..NL-1) when that count is a compile-time fact (all rows, or one of the pruned counts the
// pipelined kernels are instantiated for); only the mirror-coherence analysis depends on it.
template <int BG, int ZC, int NCWG_ = z64_ncwg<BG, ZC>(), int NL_ = BGT<BG>::ROWS> struct Z64 : BGD<BG> {
    static constexpr int NL = NL_;
    static constexpr bool RT = NL_ == NL_RT;           // the layer count is a run-time prefix of the all-rows tables (nrldpc_device.h)
    static constexpr int NLT = nl_rows<BG>(NL_);       // rows the tables cover
    static constexpr int NNZA = BGD<BG>::row_ptr(NLT); // edges of the active rows
    static constexpr bool PACKED = z64_packed(ZC);
    static constexpr int BLK = z64_blk(ZC);             // rows (ring words) per wave
    static_assert(BLK >= (PACKED ? 2 : 4) && ZC % BLK == 0, "no usable block size for this lifting size");
    static constexpr int NWV = ZC / BLK;                // waves per codeword
    static constexpr int TPC = NWV * 64;                // threads per codeword (lanes BLK..63 of a wave retire)
    static constexpr int PW = PACKED ? z64p_ncw(BG, ZC) : 1; // packed: words between consecutive ring positions (= codewords per workgroup)
    static constexpr int GUARD = PACKED ? 4 * ZC * PW : 256; // bytes: never-read words in front of every ring (packed: of column 0 only, so that no address is negative)
    static constexpr int CS = PACKED ? 8 * ZC * PW : GUARD + (ZC + 64) * 4; // column stride in bytes (guard + ring + mirror | ring + mirror)
    static constexpr int CWS = BGD<BG>::NC * CS;        // codeword stride in bytes (packed: of the workgroup's whole image)
    // a shift P as (index of the thread's base address, byte offset from it)
    static constexpr int ridx(int P) { return P / BLK; }
    static constexpr int roff(int P) { return 4 * (P % BLK) * PW; }
    // Where edge (column c, shift P) of this thread's row lives: base register R[pb(c, P)] plus the immediate po(c, P).
    // Block geometry: R[k] = the thread's word of ring block (w + k) mod NWV of column 0.  Packed geometry: ONE base, R[0] = 4 g
    // -- the thread's word of the guard in front of column 0 -- so that the twin of a word (NROWP words below it / above it)
    // is the same register with another immediate; an LDS instruction's immediate offset is 16 bits, so a workgroup image
    // beyond 64 KB (BG1 with 5- or 6-wave halves) addresses its columns from HICOL on from a second base, R[1] = R[0] + HIOFF.
    static constexpr int NROWP = PACKED ? ZC * PW : 0; // packed: row lanes of the workgroup = words of one copy of a ring
    static constexpr int hicol() {
        if (!PACKED) return BGD<BG>::NC;
        for (int c = 0; c < BGD<BG>::NC; ++c)
            if (GUARD + c * CS + 4 * (ZC - 1) * PW + 4 * NROWP > 65535) return c; // (the immediate of its twin above the last ring position)
        return BGD<BG>::NC;
    }
    static constexpr int HICOL = hicol();
    static constexpr int HIOFF = HICOL * CS;
    static constexpr int NBASE = PACKED ? (HICOL < BGD<BG>::NC ? 2 : 1) : NWV;
    static constexpr int pb(int c, int P) { return PACKED ? (c >= HICOL ? 1 : 0) : ridx(P); }
    static constexpr int po(int c, int P) { return c * CS + roff(P) + (PACKED ? GUARD - (c >= HICOL ? HIOFF : 0) : 0); }
    static constexpr int NCWG = NCWG_;                  // codewords per workgroup
    static constexpr int ILV = z64_ilv();               // interleaved block geometry: codewords of the real size ZR per workgroup
    static constexpr int ZR = ZC / ILV;                 // the lifting size of the code (= ZC unless interleaved)
    static_assert(ZC % ILV == 0 && z64_set_index(ZR) >= 0, "the virtual size is a multiple of a real lifting size");
    static constexpr int ILS = z64_set_index(ZR);
    static constexpr int shift(int e) { return ((BG == 1 ? nr_bg1_shift[ILS][e] : nr_bg2_shift[ILS][e < NR_BG2_NNZ ? e : 0]) % ZR) * ILV; }
    // Mirror coherence analysis (all layers active).  After edge e rewrites its column, ring words i >= kb_e
    // are fresh in the primary copy of block 0 and words i < kb_e are fresh in the mirror.  The next edge on
    // the same column (cyclic layer order) reads words i >= kb' from the primary copy and words i < kb' from
    // the mirror, so exactly one twin write is owed: mirror -> primary ("A") if kb' < kb_e, primary ->
    // mirror ("B") if kb' > kb_e, none if equal.
    static constexpr int next_on_column(int e) {
        const int c = BGD<BG>::col(e);
        for (int i = e + 1; i < NNZA; ++i)
            if (BGD<BG>::col(i) == c) return i;
        for (int i = 0; i < e; ++i)
            if (BGD<BG>::col(i) == c) return i;
        return e;
    }
    // The last writer of a column in an iteration makes both copies fresh: the parity check, the soft
    // output and the hard decision read the column through every edge / through the primary copy.
    static constexpr bool last_on_column(int e) { return next_on_column(e) <= e; }
    static constexpr bool twin_a(int e, bool full) {
        return shift(e) % BLK != 0 && (!full || last_on_column(e) || shift(next_on_column(e)) % BLK < shift(e) % BLK);
    }
    static constexpr bool twin_b(int e, bool full) {
        return !full || last_on_column(e) || shift(next_on_column(e)) % BLK > shift(e) % BLK;
    }
    // Run-time layer count (RT).  Edge e's column is read next by the next edge n on it in a LATER row -- if that row is active --
    // and otherwise (e is the column's last ACTIVE writer: n_layers <= row(n)) by the column's first edge in the next iteration,
    // by the parity pass through every active edge, and by the hard decision through the primary copy: both twins then.  So a
    // twin is owed never, always (n needs it, or e is the column's last edge in the table), or only when e is the last active
    // writer -- a scalar comparison of n_layers with twin_last_row(e) around that one store.  With every row active this writes
    // exactly what the all-rows build writes.
    static constexpr int TW_NEVER = 0, TW_ALWAYS = 1, TW_IF_LAST = 2;
    static constexpr int row_of(int e) {
        int r = 0;
        while (BGD<BG>::row_ptr(r + 1) <= e) ++r;
        return r;
    }
    static constexpr int later_on_column(int e) {
        const int c = BGD<BG>::col(e);
        for (int i = e + 1; i < BGD<BG>::NNZ; ++i)
            if (BGD<BG>::col(i) == c) return i;
        return -1;
    }
    static constexpr int twin_last_row(int e) { return later_on_column(e) < 0 ? BGD<BG>::ROWS : row_of(later_on_column(e)); }
    static constexpr int twin_mode_a(int e, bool full) {
        if (!RT) return twin_a(e, full) ? TW_ALWAYS : TW_NEVER;
        if (shift(e) % BLK == 0) return TW_NEVER;
        const int n = later_on_column(e);
        return (n < 0 || shift(n) % BLK < shift(e) % BLK) ? TW_ALWAYS : TW_IF_LAST;
    }
    static constexpr int twin_mode_b(int e, bool full) {
        if (!RT) return twin_b(e, full) ? TW_ALWAYS : TW_NEVER;
        const int n = later_on_column(e);
        return (n < 0 || shift(n) % BLK > shift(e) % BLK) ? TW_ALWAYS : TW_IF_LAST;
    }
    // + one trailing guard (the last column's block-0 twin write overshoots into it) + termination flags
    // ... + the CRC-aided stop's partial remainders (CrcFold, nrldpc_device.h), CRC_SLOTS words per codeword
    static constexpr size_t FLAG_BYTES = 16 * ((NCWG + 1 + 3) / 4);
    static constexpr size_t lds_bytes() { return (size_t)NCWG * CWS + GUARD + FLAG_BYTES + 4 * (size_t)CRC_SLOTS * NCWG; }
};

// binary decision tree on the (wave-uniform) wave index: at most ceil(log2(HI-LO)) scalar branches
template <int LO, int HI, class F> __device__ __forceinline__ void dispatch_w(int w, F&& f) {
    if constexpr (HI - LO == 1) {
        f(std::integral_constant<int, LO>{});
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (w < MID) dispatch_w<LO, MID>(w, f);
        else dispatch_w<MID, HI>(w, f);
    }
}

// Split kernels (nrldpc_decode_z64s.h): are the dense core rows 0..3 worked on by both halves at once (Own::dual)?  BG1 only
// (19 edges per row; BG2's rows 0..3 have 8-10), and only where the halves' exchange buffer (16 Z bytes of LDS) does not cost
// a workgroup per CU, and only for the 12-wave workgroups of which a CU holds two (Z = 384, 288): with three or more (smaller)
// workgroups per CU the other workgroups already fill a dense row's idle half, and the extra barrier costs more than it saves.
// Measured, one session each, against the same build without it -- Z = 384: all rows -1.8 % (fixed 25) / -2 % (parity stop);
// 5 / 13 / 24 rows -2 / -5 / -3 % (fixed), -5 / -2 / -2 % (parity stop); Z = 288: -2.4 % / -4.9 %; Z = 208 / 224 / 240 / 256 (8-wave
// workgroups): 0 / +2.3 / +1.3 / +0.3 % (fixed), +1...3 % (parity stop): off there.  -DNRLDPC_Z64S_DUAL=0/1 forces it (A/B).
template <int BG, int ZC, int NL> constexpr bool z64s_dual() {
#ifdef NRLDPC_Z64S_DUAL
    return NRLDPC_Z64S_DUAL != 0 && BG == 1;
#endif
    if (BG != 1) return false;
    using B = Z64<BG, ZC, 1, NL>;
    constexpr size_t base = (size_t)B::CWS + B::GUARD + 16;
    constexpr int by_waves = 24 / (2 * B::NWV) > 0 ? 24 / (2 * B::NWV) : 1;
    constexpr int w0 = (int)((160 * 1024) / base) < by_waves ? (int)((160 * 1024) / base) : by_waves;
    constexpr int w1 = (int)((160 * 1024) / (base + 16 * ZC)) < by_waves ? (int)((160 * 1024) / (base + 16 * ZC)) : by_waves;
    return w1 == w0 && by_waves <= 2;
}
// ... and are its barrier groups single layers (LayerGroups<.., SG>: 46 / 42 barriers per iteration instead of 32 / 28)?  The row
// form wants as few barriers as possible; the split form's barriers are hand-overs between the halves, and a merged pair of
// layers makes one half's interval twice as long as the other's next one.  Measured against the merged groups, one session:
// BG1 Z = 384 -3.2 % (fixed 25) / -3.9 % (parity stop), 5 / 13 / 24 rows 0 / 0 / -1 % and -1 %; Z = 288 -3.0 / -1.5 %; the 8-wave
// workgroups Z = 256 / 240 / 208: 0...-1 % fixed but +1...4 % with the parity stop -- off; Z <= 128: within +-1.4 %, off.
// BG2 (rows 0..3 are not dual there): Z = 256 / 240 / 224 / 208: 0 / -1.1 / -3.2 / -3.4 % fixed, -4.5 / -4.7 / -9.0 / -8.8 % parity
// stop; Z = 64 / 60 / 52: 0 / -2.8 / +1 % and 0 / -2.3 / -2 %: on.  -DNRLDPC_Z64S_SINGLE=0/1 forces it (A/B).
template <int BG, int ZC, int NL> constexpr bool z64s_single() {
#ifdef NRLDPC_Z64S_SINGLE
    return NRLDPC_Z64S_SINGLE != 0;
#endif
    if (NL == NL_RT) return true; // a run-time layer count may end an iteration after any layer: no merged groups to cut
    return BG == 2 || 24 / (2 * Z64<BG, ZC, 1, NL>::NWV) <= 2;
}
template <int BG, int ZC, int NL> constexpr int z64s_variant() {
    return (z64s_dual<BG, ZC, NL>() ? SPLIT_DUAL : 0) | (z64s_single<BG, ZC, NL>() ? SPLIT_SINGLE : 0);
}
// the barrier-group table of a kernel form (H < 0: one thread per check row)
// (a run-time layer count may end an iteration after any layer, so its builds -- both forms -- have one-layer groups)
template <int BG, int ZC, int NL, int H> using LGof = LayerGroups<BG, NL, (NL == NL_RT || (H >= 0 && z64s_single<BG, ZC, NL>()))>;

// One base-graph layer for this thread's check row, split into phases so that the layers of a
// column-disjoint barrier group can issue all their LDS reads first and share one mirror dispatch.
template <int BG, int ZC, int L, bool FULL, int NL = BGT<BG>::ROWS, int H = -1> struct LayerZ64 {
    using G = Z64<BG, ZC, z64_ncwg<BG, ZC>(), NL>;
    static constexpr int e0 = G::row_ptr(L);
    static constexpr int deg = G::row_ptr(L + 1) - e0;
    static constexpr bool HAS_EXT = (L >= 4);
    static constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    using OW = Own<BG, NL, H, (H >= 0 ? z64s_variant<BG, ZC, NL>() : 0)>;
    static constexpr int ce0 = OW::core_base(L); // first message byte of this layer in the thread's store
    static constexpr int XI = HAS_EXT ? OW::ext_index(L) : 0; // its extension LLR there
    static constexpr bool DUALROW = OW::dual(L);  // the row's edges alternate between the two halves
    static constexpr bool owned(int j) { return !DUALROW || (j % 2 == H); }
    static constexpr int cidx(int j) {                                    // message byte of edge j in the thread's store
        int n = ce0;
        for (int i = 0; i < j; ++i) n += owned(i) ? 1 : 0;
        return n;
    }
    float t[ncore];
    float lam, m1, M1, M2;

    __device__ __forceinline__ void load(const char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE]) {
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int P = G::shift(e0 + j);
            t[j] = *reinterpret_cast<const float*>(lds + R[G::pb(G::col(e0 + j), P)] + G::po(G::col(e0 + j), P));
        });
    }

    // ---- software-pipelined form (all layers active): an edge is "late" if its column is written by the
    // barrier group in front of this layer's group (cyclically), otherwise "early": early edges may be read
    // and folded into the min search BEFORE the barrier that separates the two groups.
    static constexpr unsigned long long prev_written() {
        using LG = LGof<BG, ZC, NL, H>;
        // run-time layer count: which group ends an iteration is not a compile-time fact, so all of group 0's edges are late.
        // (Measured on BG1 Z = 384, one session, against timing-only builds that took the late set of ONE layer count as a
        // compile-time fact: 0...1.5 % -- and a build that decided per edge at run time, behind scalar branches on a column mask,
        // was 4-10 % SLOWER than this: the branches split the layer into basic blocks the scheduler cannot interleave.)
        if (G::RT && LG::group_index(L) == 0) return ~0ull;
        return LG::group_mask((LG::group_index(L) + LG::ngroups() - 1) % LG::ngroups());
    }
    static constexpr bool is_late(int j) { return (prev_written() >> G::col(e0 + j)) & 1ull; }
    float pm1, pm2;  // partial two-smallest search
    uint32_t pS;     // partial sign parity

    template <bool LATE> __device__ __forceinline__ void load_part(const char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE]) {
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (LayerZ64::is_late(j) == LATE && LayerZ64::owned(j)) {
                constexpr int P = G::shift(e0 + j);
                t[j] = *reinterpret_cast<const float*>(lds + R[G::pb(G::col(e0 + j), P)] + G::po(G::col(e0 + j), P));
            }
        });
    }
    // number of edges of the requested part among edges 0..j-1 (compile time): pairs of sign words are folded
    // into the parity with one three-input xor (v_bitop3_b32 0x96) instead of two v_xor
    template <bool LATE> static constexpr int part_count_before(int j) {
        int n = 0;
        for (int i = 0; i < j; ++i) n += (is_late(i) == LATE);
        return n;
    }
    // Early edges whose min search is DEFERRED until after the barrier (their LDS reads still precede it): the first
    // NRLDPC_Z64_DEFER early edges of a layer (and, with NRLDPC_Z64_DEFER_EXT, its thread-private extension bit).  Every
    // wave of a SIMD leaves a barrier at the same moment and then waits one LDS round trip for its late reads; the
    // deferred edges are work that is ready to issue in that shadow.  min / med3 / xor are order-independent on exact
    // integers, so results do not depend on the split.
    static constexpr bool is_deferred(int j) { return !is_late(j) && part_count_before<false>(j) < NRLDPC_Z64_DEFER; }
    // PART 0: early, tracked before the barrier (starts the search); 1: early, deferred; 2: late
    static constexpr int part_of(int j) { return !owned(j) ? 3 : is_late(j) ? 2 : is_deferred(j) ? 1 : 0; }
    template <int PART> static constexpr int pcount_before(int j) {
        int n = 0;
        for (int i = 0; i < j; ++i) n += (part_of(i) == PART);
        return n;
    }
    template <int PART, bool XF = false, class St> __device__ __forceinline__ void track3(const St& st, float cap) {
        // cap = (127.49 + beta)/alpha: the search starts from it, so alpha*m - beta never rounds above 127 and needs no upper clamp
        if constexpr (PART == 0) { pm1 = cap; pm2 = cap; pS = 0; }
        uint32_t pend = 0;
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (LayerZ64::part_of(j) == PART) {
                constexpr int ce = LayerZ64::cidx(j);
                const float tj = t[j] - byte_to_f32<ce & 3>(st.rm[ce >> 2]);
                t[j] = tj;
                const float aj = fabsf(tj);
                pm2 = __builtin_amdgcn_fmed3f(aj, pm1, pm2);
                pm1 = fminf(pm1, aj);
                if constexpr (LayerZ64::template pcount_before<PART>(j) % 2 == 0) pend = fbits(tj);
                else pS = __builtin_amdgcn_bitop3_b32(pS, pend, fbits(tj), 0x96);
            }
        });
        constexpr int npart = pcount_before<PART>(ncore);
        if constexpr (PART == (NRLDPC_Z64_DEFER_EXT ? 1 : 0) && HAS_EXT) { // the extension bit is thread-private: never "late"
            lam = st.template ext<XI, XF>();
            const float al = fabsf(lam);
            pm2 = __builtin_amdgcn_fmed3f(al, pm1, pm2);
            pm1 = fminf(pm1, al);
            if constexpr (npart % 2 == 1) pS = __builtin_amdgcn_bitop3_b32(pS, pend, fbits(lam), 0x96);
            else pS ^= fbits(lam);
        } else {
            if constexpr (npart % 2 == 1) pS ^= pend;
        }
    }
    template <bool LATE, bool XF = false, class St> __device__ __forceinline__ void track_part(const St& st, float cap) {
        if constexpr (LATE) {
            if constexpr (NRLDPC_Z64_DEFER > 0 || NRLDPC_Z64_DEFER_EXT) track3<1, XF>(st, cap);
            track3<2, XF>(st, cap);
        } else {
            track3<0, XF>(st, cap);
        }
    }
    // Dual rows: this half's partial search result -> LDS (m1, m2 with the sign parity in m2's sign bit: both are >= 0) ...
    __device__ __forceinline__ void publish(char* lds, uint32_t xmine) const {
        *reinterpret_cast<float2*>(lds + xmine) = make_float2(pm1, __uint_as_float(fbits(pm2) | (pS & 0x80000000u)));
    }
    // ... and the other half's merged in: two smallest of {pm1 <= pm2, o1 <= o2}, parities xor-ed
    __device__ __forceinline__ void merge(const char* lds, uint32_t xother) {
        const float2 o = *reinterpret_cast<const float2*>(lds + xother);
        const float o2 = fabsf(o.y);
        pS ^= fbits(o.y);
        pm2 = fminf(fminf(fmaxf(pm1, o.x), pm2), o2);
        pm1 = fminf(pm1, o.x);
    }
    // pass 2 for all (owned) edges after every part has been tracked
    template <class St> __device__ __forceinline__ void finish(St& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], const DecArgs& a) {
        m1 = pm1;
        // magnitudes carrying the row's sign parity: M | (S & signbit) in one v_bitop3_b32 (0xF8 = a | (b & c))
        // a.beta holds 2^23 - beta here (set up by the pipelined kernels, see scale_mag_magic)
        M1 = __uint_as_float(__builtin_amdgcn_bitop3_b32(fbits(scale_mag_magic(a.alpha, a.beta, pm1)), pS, 0x80000000u, 0xF8));
        M2 = __uint_as_float(__builtin_amdgcn_bitop3_b32(fbits(scale_mag_magic(a.alpha, a.beta, pm2)), pS, 0x80000000u, 0xF8));
        bool ismin[ncore]; // all compares first: keeps v_cmp -> v_cndmask hazard slots filled with useful work
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (LayerZ64::owned(j)) ismin[j] = fabsf(t[j]) == m1;
        });
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (LayerZ64::owned(j)) {
                constexpr int ce = LayerZ64::cidx(j);
                constexpr int P = G::shift(e0 + j);
                const float tj = t[j];
                const float mag = ismin[j] ? M2 : M1;
                const float r = __uint_as_float(fbits(mag) ^ (fbits(tj) & 0x80000000u));
                f32_to_byte<ce & 3>(st.rm[ce >> 2], r);
                const float v = tj + r;
                t[j] = v;
                *reinterpret_cast<float*>(lds + R[G::pb(G::col(e0 + j), P)] + G::po(G::col(e0 + j), P)) = v;
            }
        });
    }

    __device__ __forceinline__ void update(DecState<BG>& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], const DecArgs& a) {
        float mm1 = __builtin_inff(), mm2 = __builtin_inff();
        uint32_t S = 0, pend = 0;
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int ce = ce0 + j;
            const float tj = t[j] - byte_to_f32<ce & 3>(st.rm[ce >> 2]);
            t[j] = tj;
            const float aj = fabsf(tj);
            mm2 = __builtin_amdgcn_fmed3f(aj, mm1, mm2);
            mm1 = fminf(mm1, aj);
            if constexpr (j % 2 == 0) pend = fbits(tj);
            else S = __builtin_amdgcn_bitop3_b32(S, pend, fbits(tj), 0x96); // three-input xor
        });
        lam = 0.0f;
        if constexpr (HAS_EXT) {
            lam = byte_to_f32<XI & 3>(st.xq[XI >> 2]);
            const float al = fabsf(lam);
            mm2 = __builtin_amdgcn_fmed3f(al, mm1, mm2);
            mm1 = fminf(mm1, al);
            if constexpr (ncore % 2 == 1) S = __builtin_amdgcn_bitop3_b32(S, pend, fbits(lam), 0x96);
            else S ^= fbits(lam);
        } else {
            if constexpr (ncore % 2 == 1) S ^= pend;
        }
        m1 = mm1;
        // magnitudes carrying the row's sign parity (M | (S & signbit), one v_bitop3_b32); the edge's own sign is
        // xor-ed in per edge
        M1 = __uint_as_float(__builtin_amdgcn_bitop3_b32(fbits(scale_mag(a, mm1)), S, 0x80000000u, 0xF8));
        M2 = __uint_as_float(__builtin_amdgcn_bitop3_b32(fbits(scale_mag(a, mm2)), S, 0x80000000u, 0xF8));
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int ce = ce0 + j;
            constexpr int P = G::shift(e0 + j);
            const float tj = t[j];
            const float mag = (fabsf(tj) == m1) ? M2 : M1;
            const float r = __uint_as_float(fbits(mag) ^ (fbits(tj) & 0x80000000u));
            f32_to_byte<ce & 3>(st.rm[ce >> 2], r);
            const float v = tj + r;
            t[j] = v; // kept for the mirror pass
            *reinterpret_cast<float*>(lds + R[G::pb(G::col(e0 + j), P)] + G::po(G::col(e0 + j), P)) = v;
        });
    }

    // Mirror coherence (ring block 0 lives at ring words [0,64) and again at [ZC, ZC+64)) owed by wave WV:
    //   wave (NWV-1-ka): its run started in the last block and ran into the mirror -> twin at RA + off
    //   wave (NWV-ka)  : its run started in block 0                                -> twin at RB + off
    // With every layer active (FULL) the next reader of the column is known at compile time and only the
    // twin it will actually read is written (Z64::twin_a/twin_b); with pruned layers both are.
    // nl: the run-time layer count (read only by the builds with G::RT, Z64::twin_mode_a / _b)
    template <int WV> __device__ __forceinline__ void twins(char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA, uint32_t RB, int nl) const {
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int P = G::shift(e0 + j);
            constexpr int ka = P / G::BLK;
            constexpr int off = G::col(e0 + j) * G::CS + G::roff(P);                     // block geometry: from RA / RB
            constexpr int pbi = G::pb(G::col(e0 + j), P), poi = G::po(G::col(e0 + j), P); // packed geometry: the word NROWP below / above
            constexpr int MA = LayerZ64::owned(j) ? G::twin_mode_a(e0 + j, FULL) : G::TW_NEVER;
            constexpr int MB = LayerZ64::owned(j) ? G::twin_mode_b(e0 + j, FULL) : G::TW_NEVER;
            constexpr int TL = G::twin_last_row(e0 + j);
            if constexpr (G::PACKED) {
                // WV = row wave; lane T of it is the first whose row wrapped into the mirror (ring position z + P >= Z)
                constexpr int T = (ZC - P) * G::PW - 64 * WV;
                if constexpr (MA != G::TW_NEVER && T < 64) { // wrapped rows: mirror -> ring
                    if (MA == G::TW_ALWAYS || nl <= TL) {
                        if constexpr (T <= 0) {
                            *reinterpret_cast<float*>(lds + R[pbi] + (poi - 4 * G::NROWP)) = t[j];
                        } else {
                            if (NRLDPC_LANE_GE(T)) *reinterpret_cast<float*>(lds + R[pbi] + (poi - 4 * G::NROWP)) = t[j];
                        }
                    }
                }
                if constexpr (MB != G::TW_NEVER && T > 0) { // the other rows: ring -> mirror
                    if (MB == G::TW_ALWAYS || nl <= TL) {
                        if constexpr (T >= 64) {
                            *reinterpret_cast<float*>(lds + R[pbi] + (poi + 4 * G::NROWP)) = t[j];
                        } else {
                            if (!NRLDPC_LANE_GE(T)) *reinterpret_cast<float*>(lds + R[pbi] + (poi + 4 * G::NROWP)) = t[j];
                        }
                    }
                }
            } else {
                constexpr bool DA = MA != G::TW_NEVER && WV == (2 * G::NWV - 1 - ka) % G::NWV;
                constexpr bool DB = MB != G::TW_NEVER && WV == (G::NWV - ka) % G::NWV;
                if constexpr (DA || DB) {
                    if constexpr (DA) {
                        if (MA == G::TW_ALWAYS || nl <= TL) *reinterpret_cast<float*>(lds + RA + off) = t[j];
                    }
                    if constexpr (DB) {
                        if (MB == G::TW_ALWAYS || nl <= TL) *reinterpret_cast<float*>(lds + RB + off) = t[j];
                    }
                    // A unique (empty) asm per store: without it SimplifyCFG sinks the six per-wave store
                    // sequences into one store that indexes t[] dynamically, which pushes t[] to scratch.
                    asm volatile("; twin L%c0 e%c1 w%c2" ::"i"(L), "i"(j), "i"(WV));
                }
            }
        });
    }

    // early termination / soft output need the a-posteriori value of the row's extension-parity bit
    __device__ __forceinline__ void ext(const DecArgs& a, uint32_t& esign_lo, uint32_t& esign_hi, float* app_ext) const {
        if constexpr (HAS_EXT) {
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
};

// Layers GS..GE (a column-disjoint barrier group, see LayerGroups) processed as one block of code.
template <int BG, int ZC, int GS, int GE, bool FULL, bool PLAIN>
__device__ __forceinline__ void group_z64(DecState<BG>& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA,
                                          uint32_t RB, int w, const DecArgs& a, uint32_t& esign_lo,
                                          uint32_t& esign_hi, float* app_ext) {
    constexpr int N = GE - GS + 1;
    static_assert(N >= 1 && N <= 3, "group size");
    constexpr int NWV = z64_packed(ZC) ? z64p_rw(BG, ZC) : z64_nwv(ZC); // packed geometry: w is the row-wave index
    LayerZ64<BG, ZC, GS, FULL> l0;
    LayerZ64<BG, ZC, (N > 1 ? GS + 1 : GS), FULL> l1;
    LayerZ64<BG, ZC, (N > 2 ? GS + 2 : GS), FULL> l2;
    l0.load(lds, R);
    if constexpr (N > 1) l1.load(lds, R);
    if constexpr (N > 2) l2.load(lds, R);
    l0.update(st, lds, R, a);
    if constexpr (N > 1) l1.update(st, lds, R, a);
    if constexpr (N > 2) l2.update(st, lds, R, a);
    dispatch_w<0, NWV>(w, [&](auto wc) {
        constexpr int WV = decltype(wc)::value;
        l0.template twins<WV>(lds, R, RA, RB, a.n_layers);
        if constexpr (N > 1) l1.template twins<WV>(lds, R, RA, RB, a.n_layers);
        if constexpr (N > 2) l2.template twins<WV>(lds, R, RA, RB, a.n_layers);
    });
    if constexpr (!PLAIN) if (a.need_ext) {
        l0.ext(a, esign_lo, esign_hi, app_ext);
        if constexpr (N > 1) l1.ext(a, esign_lo, esign_hi, app_ext);
        if constexpr (N > 2) l2.ext(a, esign_lo, esign_hi, app_ext);
    }
}

// Wave priority inside the software pipeline.  Between two barriers a wave first finishes its group (late reads, min search
// over them, pass 2, LDS writes: what every other wave of the workgroup waits for at the next barrier) and then runs the
// early part of the next group (work that merely has to be done some time before that group's own barrier).  With equal
// priorities the SIMD's arbiter lets a wave that is already in the second part take issue slots from a sibling that is
// still in the first, and the barrier opens later.  s_setprio 2 for the first part, 0 for the second: the headline launch
// went from 3.96 to 3.67 ms (8.7 -> 9.4 Gbit/s; levels 1, 2, 3 and a window that starts at pass 2 measure the same).
// -DNRLDPC_Z64_PRIO=0 builds the kernel without the effect (A/B).
#ifndef NRLDPC_Z64_PRIO
#define NRLDPC_Z64_PRIO 2
#endif

// ---- software pipeline over barrier groups (FULL && PLAIN kernels) -------------------------------------
// Group gi's layers; `early` = loads + min search over the edges that do not depend on the previous group.
template <int BG, int ZC, int GI, int NL = BGT<BG>::ROWS, int H = -1> struct GroupZ64 {
    using LG = LGof<BG, ZC, NL, H>;
    static constexpr int GS = LG::group_first(GI);
    static constexpr int N = LG::group_last(GS) - GS + 1;
    static_assert(N >= 1 && N <= 3, "group size");
    struct NoLayer {}; // absent second / third layer: no storage, so copying a group copies only live state
    LayerZ64<BG, ZC, GS, true, NL, H> l0;
    std::conditional_t<(N > 1), LayerZ64<BG, ZC, (N > 1 ? GS + 1 : GS), true, NL, H>, NoLayer> l1;
    std::conditional_t<(N > 2), LayerZ64<BG, ZC, (N > 2 ? GS + 2 : GS), true, NL, H>, NoLayer> l2;

    template <bool LATE> __device__ __forceinline__ void loads(const char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE]) {
        l0.template load_part<LATE>(lds, R);
        if constexpr (N > 1) l1.template load_part<LATE>(lds, R);
        if constexpr (N > 2) l2.template load_part<LATE>(lds, R);
    }
    template <bool LATE, bool XF = false, class St> __device__ __forceinline__ void track(const St& st, float cap) {
        l0.template track_part<LATE, XF>(st, cap);
        if constexpr (N > 1) l1.template track_part<LATE, XF>(st, cap);
        if constexpr (N > 2) l2.template track_part<LATE, XF>(st, cap);
    }
    template <class St> __device__ __forceinline__ void finish(St& st, char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], const DecArgs& a) {
        l0.finish(st, lds, R, a);
        if constexpr (N > 1) l1.finish(st, lds, R, a);
        if constexpr (N > 2) l2.finish(st, lds, R, a);
    }
    __device__ __forceinline__ void ext(const DecArgs& a, uint32_t& esign_lo, uint32_t& esign_hi) const {
        l0.ext(a, esign_lo, esign_hi, nullptr);
        if constexpr (N > 1) l1.ext(a, esign_lo, esign_hi, nullptr);
        if constexpr (N > 2) l2.ext(a, esign_lo, esign_hi, nullptr);
    }
    // w: the wave's index within its codeword (block geometry) / its row-wave index (packed geometry)
    __device__ __forceinline__ void twins(char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA, uint32_t RB, int w, int nl) const {
        dispatch_w<0, (z64_packed(ZC) ? z64p_rw(BG, ZC) : z64_nwv(ZC))>(w, [&](auto wc) {
            constexpr int WV = decltype(wc)::value;
            l0.template twins<WV>(lds, R, RA, RB, nl);
            if constexpr (N > 1) l1.template twins<WV>(lds, R, RA, RB, nl);
            if constexpr (N > 2) l2.template twins<WV>(lds, R, RA, RB, nl);
        });
    }
};

// One iteration: group GI arrives with its early part done; barrier; late part; pass 2; the NEXT group's early
// part is started before this group's pass 2 so that its LDS latency and min search overlap the barrier wait.
// Returns (through `next0`) group 0 with its early part done for the following iteration.
// ET: also record the sign of every extension-parity bit's a-posteriori value (the parity pass of the
// early-termination kernel needs it).
template <int BG, int ZC, int GI, bool ET = false, int NL = BGT<BG>::ROWS, bool XF = false>
__device__ __forceinline__ void pipeline_z64(GroupZ64<BG, ZC, GI, NL>& cur, GroupZ64<BG, ZC, 0, NL>& next0, DecState<BG>& st,
                                             char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t RA, uint32_t RB,
                                             int w, const DecArgs& a, float cap, uint32_t& esign_lo,
                                             uint32_t& esign_hi) {
    using LG = LGof<BG, ZC, NL, -1>;
    constexpr int NG = LG::ngroups();
    constexpr bool RT = NL == NL_RT; // run-time layer count: the iteration ends after the last active layer (the kernel prepares group 0)
    static_assert(!RT || !NRLDPC_Z64_POSTBAR, "the run-time layer count is built for the shipped schedule only");
    // ends group GI-1.  With early termination the parity pass between two iterations ends with a barrier of its own (and the
    // first iteration follows the prologue's), so group 0 needs none -- the waves that only keep the barrier count skip it too.
    if constexpr (!(ET && GI == 0)) __syncthreads();
#if NRLDPC_Z64_POSTBAR
    // Variant: the group's own early part (its LDS reads were issued before the barrier) is tracked AFTER the barrier, in
    // the shadow of the late reads' LDS round trip, instead of before it; only the next group's early READS precede the
    // next barrier.  Every wave of a SIMD leaves a barrier at the same time, so with the early part already done nothing
    // is left to issue while the late reads are in flight.
    cur.template loads<true>(lds, R);
    cur.template track<false, XF>(st, cap);
    cur.template track<true, XF>(st, cap);
    if constexpr (GI + 1 < NG) {
        GroupZ64<BG, ZC, GI + 1, NL> nxt;
        nxt.template loads<false>(lds, R);
        cur.finish(st, lds, R, a);
        cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
        if constexpr (ET) {
            cur.ext(a, esign_lo, esign_hi);
            asm volatile("" : "+v"(esign_lo), "+v"(esign_hi));
        }
        pipeline_z64<BG, ZC, GI + 1, ET, NL, XF>(nxt, next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
    } else {
        next0.template loads<false>(lds, R);
        cur.finish(st, lds, R, a);
        cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
        if constexpr (ET) {
            cur.ext(a, esign_lo, esign_hi);
            asm volatile("" : "+v"(esign_lo), "+v"(esign_hi));
        }
    }
#else
    __builtin_amdgcn_s_setprio(NRLDPC_Z64_PRIO); // urgent until this group's writes are out (see NRLDPC_Z64_PRIO)
    cur.template loads<true>(lds, R);
    if constexpr (GI + 1 < NG) {
        GroupZ64<BG, ZC, GI + 1, NL> nxt;
        nxt.template loads<false>(lds, R); // columns untouched by group GI: safe before its writes
        cur.template track<true, XF>(st, cap);
        cur.finish(st, lds, R, a);
        cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
        __builtin_amdgcn_s_setprio(0); // the next group's early part only fills gaps
        if constexpr (ET) {
            cur.ext(a, esign_lo, esign_hi);
            // pin the bits here: their only reader is the parity pass, and LLVM otherwise sinks all 42 rows'
            // sign computations (with lam, m1, M1, M2 of every row kept alive in scratch) down to it
            asm volatile("" : "+v"(esign_lo), "+v"(esign_hi));
        }
        nxt.template track<false, XF>(st, cap);
        // (run-time layer count: when layer GI was the last active one, the early part above was speculative -- reads of valid
        // LDS words into registers nobody uses -- and the iteration ends here)
        if (!RT || LG::group_first(GI + 1) < launder(a.n_layers))
            pipeline_z64<BG, ZC, GI + 1, ET, NL, XF>(nxt, next0, st, lds, R, RA, RB, w, a, cap, esign_lo, esign_hi);
    } else {
        if constexpr (!RT) next0.template loads<false>(lds, R);
        cur.template track<true, XF>(st, cap);
        cur.finish(st, lds, R, a);
        cur.twins(lds, R, RA, RB, w, launder(a.n_layers));
        __builtin_amdgcn_s_setprio(0); // the next group's early part only fills gaps
        if constexpr (ET) {
            cur.ext(a, esign_lo, esign_hi);
            asm volatile("" : "+v"(esign_lo), "+v"(esign_hi));
        }
        if constexpr (!RT) next0.template track<false, XF>(st, cap);
    }
#endif
}

// N values live in registers at one point of the program, behind a compiler-level memory fence: every load that produced them has
// been issued before it, none of the loads after it has (the LDS reads of a parity check go out in batches of N, neither one round
// trip per edge nor all nineteen of a dense row at once)
template <int N> __device__ __forceinline__ void pin_batch(uint32_t (&v)[8]) {
    static_assert(N >= 1 && N <= 8, "batch size");
    if constexpr (N == 1) asm volatile("" : "+v"(v[0]) : : "memory");
    else if constexpr (N == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1]) : : "memory");
    else if constexpr (N == 3) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) : : "memory");
    else if constexpr (N == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
    else if constexpr (N == 5) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]) : : "memory");
    else if constexpr (N == 6) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : : "memory");
    else if constexpr (N == 7) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) : : "memory");
    else asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}

// Parity of check row L on the signs of the a-posteriori values.  The row's LDS reads go out in batches of up to NRLDPC_PARITY_BATCH
// (pin_batch): left alone, the compiler of this ROCm release either waited for every read before issuing the next (one LDS round
// trip per edge) or -- dense rows of BG1 under the 80-VGPR budget of the split kernels -- issued all nineteen and parked them in
// SCRATCH one by one (16 spilled registers in <1, 384, ETP>: 75 MB of the 109 MB that launch wrote to HBM were these, round 6;
// profiles/r06_parity_pass_read_batching.txt: -1 ... -5 % of the stop's time at the waterfall).  -DNRLDPC_PARITY_BATCH=0: as before (A/B).
#ifndef NRLDPC_PARITY_BATCH
#define NRLDPC_PARITY_BATCH 6
#endif
template <int BG, int ZC, int L>
__device__ __forceinline__ uint32_t row_parity_z64(char* lds, const uint32_t (&R)[Z64<BG, ZC>::NBASE], uint32_t esign_lo,
                                                   uint32_t esign_hi) {
    using G = Z64<BG, ZC>;
    constexpr int e0 = G::row_ptr(L);
    constexpr int deg = G::row_ptr(L + 1) - e0;
    constexpr bool HAS_EXT = (L >= 4);
    constexpr int ncore = deg - (HAS_EXT ? 1 : 0);
    constexpr int CH = NRLDPC_PARITY_BATCH > 0 ? NRLDPC_PARITY_BATCH : 8;
    static_assert(NRLDPC_PARITY_BATCH >= 0 && CH <= 8, "NRLDPC_PARITY_BATCH");
    uint32_t p = 0;
    if constexpr (NRLDPC_PARITY_BATCH == 0) { // A/B: the reads left to the compiler (rounds 1-5)
        static_for<ncore>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int c = G::col(e0 + j);
            constexpr int P = G::shift(e0 + j);
            p ^= fbits(*reinterpret_cast<const float*>(lds + R[G::pb(c, P)] + G::po(c, P)));
        });
    } else static_for<(ncore + CH - 1) / CH>([&](auto bc) {
        constexpr int b0 = decltype(bc)::value * CH;
        constexpr int n = ncore - b0 < CH ? ncore - b0 : CH;
        uint32_t v[8];
        static_for<n>([&](auto ic) {
            constexpr int j = b0 + decltype(ic)::value;
            constexpr int c = G::col(e0 + j);
            constexpr int P = G::shift(e0 + j);
            v[decltype(ic)::value] = fbits(*reinterpret_cast<const float*>(lds + R[G::pb(c, P)] + G::po(c, P)));
        });
        pin_batch<n>(v);
        static_for<n>([&](auto ic) { p ^= v[decltype(ic)::value]; });
        // ... and the batch is folded before the next one is read (the fence again, now behind the xors: without it the scheduler
        // issued the next batch's reads first and parked this batch's values in scratch until "later")
        if constexpr (b0 + n < ncore) asm volatile("" : "+v"(p) : : "memory");
    });
    p >>= 31;
    if constexpr (HAS_EXT) p ^= (L - 4 < 32 ? esign_lo >> ((L - 4) & 31) : esign_hi >> ((L - 36) & 31)) & 1u;
    return p;
}

// FULL : every layer of the base graph is active (n_layers == rows): no per-layer predicates, single mirror twin.
// PLAIN: additionally no early termination and no soft output (the fixed-iteration throughput path): no
//        per-thread `done` predicate, no extension-bit bookkeeping, no parity pass.
// ETP  : FULL with early termination (no soft output): the pipelined iteration of PLAIN plus the sign of every
//        extension-parity bit, then the parity pass; a finished codeword's waves only keep the barriers.
template <int BG, int ZC, int NCWG, int NL> constexpr bool z64_ext_float() {
#ifdef NRLDPC_Z64_XF
    return NRLDPC_Z64_XF != 0;
#endif
    return BG == 1 && NL == BGT<BG>::ROWS && z64_wpe<BG, ZC, NCWG, NL>() == 3;
}

// NL   : the compile-time layer count of a FULL build: all rows, or one of the pruned counts of NRLDPC_Z64_NL_LIST
//        (the rate-matching points BASELINE.json names), each with its own barrier-group table and prefetch plan.
// CRC  : ETP with the CRC-aided stop compiled in (nrldpc_cfg.early_term = 2; a twin of its own, see the split kernel)
template <int BG, int ZC, int NCWG, bool FULL, bool PLAIN, bool ETP = false, int NL = BGT<BG>::ROWS, bool CRC = false>
__global__ __launch_bounds__(NCWG * z64_nwv(ZC) * 64, (z64_wpe<BG, ZC, NCWG, NL>())) void nrldpc_decode_z64_kernel(const DecArgs a) {
    static_assert(!CRC || ETP, "the CRC-aided stop is a mode of the parity-stop build");
    static_assert(!PLAIN || FULL, "PLAIN implies FULL");
    static_assert(!ETP || (FULL && !PLAIN), "ETP implies FULL and excludes PLAIN");
    static_assert(FULL || NL == BGT<BG>::ROWS, "a run-time layer count uses the all-rows tables");
    using G = Z64<BG, ZC, NCWG, NL>;
    using LGN = LGof<BG, ZC, NL, -1>;
    constexpr bool RT = NL == NL_RT; // FULL with the layer count as a run-time prefix of the all-rows tables
    static_assert(!RT || FULL, "a run-time layer count is a mode of the pipelined builds");
    // barriers a wave without a codeword keeps per iteration: one per group with an active layer
    auto idle_barriers = [&](int skip) {
        if constexpr (RT) {
            static_for<LGN::ngroups()>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (g >= skip && LGN::group_first(g) < launder(a.n_layers)) __syncthreads();
            });
        } else {
            for (int g = skip; g < LGN::ngroups(); ++g) __syncthreads();
        }
    };
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cwl = wave / G::NWV, w = wave % G::NWV, lane = tid & 63;
    if constexpr (G::BLK < 64) {
        if (lane >= G::BLK) return; // these lanes own no row; barriers count waves, not lanes
    }
    const int z = w * G::BLK + lane;
    const int cw = blockIdx.x * G::NCWG + cwl;
    const bool active = cw < a.batch; // wave-uniform: whole waves belong to one codeword
    const uint32_t cwbase = (uint32_t)cwl * (uint32_t)G::CWS;
    int* flags = reinterpret_cast<int*>(lds + (size_t)G::NCWG * G::CWS + G::GUARD);
    constexpr size_t ncwz = (size_t)G::COLS * ZC;

    uint32_t R[G::NWV];
#pragma unroll
    for (int k = 0; k < G::NWV; ++k)
        R[k] = cwbase + G::GUARD + (uint32_t)(4 * G::BLK) * (uint32_t)((w + k) % G::NWV) + 4u * (uint32_t)lane;
    // twin addresses: a run in the last block mirrors to ring word (kb+lane-64) => column base + 4(kb+lane);
    // a run in block 0 mirrors to ring word ZC+kb+lane
    const uint32_t RA = cwbase + (uint32_t)(G::GUARD - 4 * G::BLK) + 4u * (uint32_t)lane;
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
        char* home = lds + cwbase + G::GUARD + 4 * z; // ring position z of column 0
        // Core columns -> LDS.  A quarter of the codeword's threads covers one column with 4 consecutive ring
        // positions each (8- or 16-byte loads, ds_write_b128), four columns per pass: 7 load instructions per
        // thread instead of 26 two-byte ones.  Unaligned LLR pointers take the one-position-per-thread path.
        // Extension columns -> registers, one load per column and thread.
        //
        // All loads of a thread are issued as RAW bits before the first one is converted, in a body instantiated per LLR
        // format: with the format tested per load (the first version of this prologue) the fp16 branch converted right
        // behind its load, so every one of the 49 loads of a thread was its own HBM round trip -- 19 us per workgroup,
        // one iteration's worth, against ~7 us now.
        constexpr int QW = ZC / 4;
        const int qs = z / QW, qq = z - qs * QW; // column within a pass, quad within the column
        const bool wide = (reinterpret_cast<uintptr_t>(a.llr) & 15) == 0;
        // The extension-parity LLR of a pruned row is never used (only soft output echoes it): at R = 8/9
        // that is 41 of 68 columns of HBM input saved.  Blocks of 8 rows, wave-uniform branches.
        const int next_used = a.app ? G::NEXT : (FULL && !RT) ? NL - 4 : launder(a.n_layers) - 4;
        auto ingest_as = [&](auto kind_c) {
            constexpr bool F16 = decltype(kind_c)::value == NRLDPC_K_F16;
            auto raw = [&](size_t i) -> uint32_t { // one LLR, raw bits
                if constexpr (F16) return static_cast<const uint16_t*>(a.llr)[i];
                else return static_cast<const uint32_t*>(a.llr)[i];
            };
            auto val = [&](uint32_t r) -> float {
                if constexpr (F16) return __half2float(__ushort_as_half((unsigned short)r));
                else return __uint_as_float(r);
            };
            uint32_t xe[G::NEXT];
            auto load_ext = [&]() {
                static_for<(G::NEXT + 7) / 8>([&](auto bc) {
                    constexpr int i0 = decltype(bc)::value * 8;
                    constexpr int i1 = i0 + 8 < G::NEXT ? i0 + 8 : G::NEXT;
                    if ((FULL && NL == G::ROWS) || i0 < next_used) {
                        static_for<i1 - i0>([&](auto ic) {
                            constexpr int i = i0 + decltype(ic)::value;
                            xe[i] = raw(base + (size_t)(G::NC + i) * ZC + z);
                        });
                    } else {
                        static_for<i1 - i0>([&](auto ic) { xe[i0 + decltype(ic)::value] = 0u; });
                    }
                });
            };
            if (wide) {
                constexpr int NP = (G::NC + 3) / 4;
                uint4 x[NP];
                static_for<NP>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 4 * k + qs;
                    x[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (4 * k + 3 < G::NC || c < G::NC) {
                        const size_t i = base + (size_t)c * ZC + 4 * qq;
                        if constexpr (F16) {
                            const uint2 r = *reinterpret_cast<const uint2*>(static_cast<const __half*>(a.llr) + i);
                            x[k].x = r.x; x[k].y = r.y;
                        } else {
                            x[k] = *reinterpret_cast<const uint4*>(static_cast<const float*>(a.llr) + i);
                        }
                    }
                });
                load_ext();
                static_for<NP>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int c = 4 * k + qs;
                    if (4 * k + 3 < G::NC || c < G::NC) {
                        float4 v;
                        if constexpr (F16) {
                            const __half2 lo = *reinterpret_cast<const __half2*>(&x[k].x), hi = *reinterpret_cast<const __half2*>(&x[k].y);
                            v = make_float4(__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi));
                        } else {
                            v = make_float4(__uint_as_float(x[k].x), __uint_as_float(x[k].y), __uint_as_float(x[k].z), __uint_as_float(x[k].w));
                        }
                        const float4 q = make_float4(ingest(v.x, a.scale, true), ingest(v.y, a.scale, true),
                                                     ingest(v.z, a.scale, true), ingest(v.w, a.scale, true));
                        char* col = lds + cwbase + G::GUARD + c * G::CS;
                        *reinterpret_cast<float4*>(col + 16 * qq) = q;
                        if (qq < 16) *reinterpret_cast<float4*>(col + 4 * ZC + 16 * qq) = q; // mirror of block 0
                    }
                });
            } else {
                uint32_t x[G::NC];
                static_for<G::NC>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    x[c] = raw(base + (size_t)c * ZC + z);
                });
                load_ext();
                static_for<G::NC>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const float q = ingest(val(x[c]), a.scale, true);
                    *reinterpret_cast<float*>(home + c * G::CS) = q;
                    if (w == 0) *reinterpret_cast<float*>(home + c * G::CS + ZC * 4) = q; // mirror of block 0
                });
            }
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                f32_to_byte<i & 3>(st.xq[i >> 2], ingest(val(xe[i]), a.scale, false));
            });
        };
        if (a.llr_kind == NRLDPC_K_F16) ingest_as(std::integral_constant<int, NRLDPC_K_F16>{});
        else ingest_as(std::integral_constant<int, NRLDPC_K_F32>{});
        if (app_row) {
            static_for<G::NEXT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                app_row[(size_t)(G::NC + i) * ZC] = byte_to_f32<i & 3>(st.xq[i >> 2]) * a.inv_scale;
            });
        }
    }
    __syncthreads();

    if constexpr (PLAIN) {
        // fixed-iteration path: barrier groups software-pipelined (see pipeline_z64)
        if (active) {
            const float cap = (127.49f + a.beta) / a.alpha; // see track_part
            // alpha and beta as VGPR values: the two fused multiply-adds per row then issue at the full rate (any VALU
            // op with an SGPR operand takes 4 cycles) and need no v_mov for their second scalar operand
            DecArgs av = a;
            av.beta = 8388608.0f - a.beta; // see scale_mag_magic
            asm volatile("" : "+v"(av.alpha), "+v"(av.beta));
            constexpr bool XF = z64_ext_float<BG, ZC, NCWG, NL>();
            if constexpr (XF) {
                static_for<G::NEXT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    st.xf[i] = byte_to_f32<i & 3>(st.xq[i >> 2]);
                });
            }
            GroupZ64<BG, ZC, 0, NL> g0;
            g0.template loads<false>(lds, R);
#if !NRLDPC_Z64_POSTBAR
            g0.template track<false, XF>(st, cap);
#endif
            for (int it = 1; it <= a.max_iter; ++it) {
                GroupZ64<BG, ZC, 0, NL> nx;
                pipeline_z64<BG, ZC, 0, false, NL, XF>(g0, nx, st, lds, R, RA, RB, w, av, cap, esign_lo, esign_hi);
                if constexpr (RT) { // wherever the iteration ended: group 0's early part (none of its edges is early here)
                    nx.template loads<false>(lds, R);
                    nx.template track<false, XF>(st, cap);
                }
                g0 = nx;
            }
        } else {
            for (int it = 1; it <= a.max_iter; ++it) idle_barriers(0);
        }
        __syncthreads();
    }
    bool done = !active;
    int my_iters = a.max_iter;
    // violated-check vote of one codeword's waves -> flags; returns after the closing barrier
    auto parity_pass = [&](int it) {
        static_assert(G::BLK == 64 || G::NCWG + 1 <= G::BLK, "flags are cleared by raw thread id: lanes >= BLK have retired");
        if (tid <= G::NCWG) flags[tid] = 0;
        int* crc_slots = flags + G::FLAG_BYTES / 4 + cwl * CRC_SLOTS; // CRC-aided stop (early_term = 2)
        if constexpr (CRC) {
            // by the DENSE index of the lanes that are still here: lanes >= BLK of every wave returned at entry, so a raw thread
            // id skips words (BG2 Z = 144: BLK 48, four codewords -> 80 words, tids 48..63 absent)
            for (int i = wave * G::BLK + lane; i < G::NCWG * CRC_SLOTS; i += G::NCWG * G::NWV * G::BLK) flags[G::FLAG_BYTES / 4 + i] = 0;
        }
        __syncthreads();
        if (CRC && !done) { // the information bits at this thread's own ring position z of every column (primary copy:
            CrcFold f;             // a column's last writer of an iteration leaves both copies fresh)
            const char* home = lds + cwbase + G::GUARD + 4 * z;
            static_for<G::KB>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                f.bit(*reinterpret_cast<const float*>(home + c * G::CS), a.crc_tab, c * ZC + z, a.crc_bits);
            });
            f.publish(crc_slots);
        }
        if (!done) {
            // A violated check anywhere settles the answer, so a wave stops reading as soon as one of its
            // 64 rows has failed (voted after each core row, then every 4 rows; this also bounds the loads in flight):
            // far from convergence the pass costs ~19 LDS reads per thread instead of all 274.
            uint32_t bad = 0;
            bool stop = false; // wave-uniform
            if constexpr (FULL) { // the active rows are a compile-time fact: cheapest rows first (Own::parity_order)
                if constexpr (RT) { // run-time layer count: highest row first, pruned rows skipped eight at a time (Own::parity_order_desc)
                    constexpr auto PD = Own<BG, NL, -1>::parity_order_desc();
                    static_for<(PD.n + 7) / 8>([&](auto bc) {
                        constexpr int b0 = 8 * decltype(bc)::value, b1 = b0 + 8 < PD.n ? b0 + 8 : PD.n;
                        const int nlb = launder(a.n_layers);
                        if (!stop && PD.v[b1 - 1] < nlb) {
                            static_for<b1 - b0>([&](auto ic) {
                                constexpr int i = b0 + decltype(ic)::value;
                                constexpr int L = PD.v[i];
                                if (!stop && L < nlb) {
                                    bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                                    if constexpr ((i % 4) == 3 || i + 1 == b1 || L < 4) stop = __any((int)bad) != 0;
                                }
                            });
                        }
                    });
                }
                constexpr auto PO = Own<BG, NL, -1>::parity_order();
                if constexpr (!RT) static_for<PO.n>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int L = PO.v[i];
                    if (!stop) {
                        bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                        if constexpr (i < 3 || (i % 4) == 3 || i + 1 == PO.n) stop = __any((int)bad) != 0;
                    }
                });
            } else {
                static_for<NL>([&](auto lc) {
                    constexpr int L = decltype(lc)::value;
                    if (!stop && L < launder(a.n_layers)) {
                        bad |= row_parity_z64<BG, ZC, L>(lds, R, esign_lo, esign_hi);
                        if constexpr (L < 4 || (L % 4) == 3) stop = __any((int)bad) != 0;
                    }
                });
            }
            if (bad) { flags[cwl] = 1; flags[G::NCWG] = 1; }
        }
        __syncthreads();
        // readfirstlane: the flags are wave-uniform by construction, and `done` has to be *provably* so --
        // as a divergent predicate it wraps the whole iteration in exec-mask control flow with phi copies
        // of all 80 state registers (measured: 168 VGPRs + 234 spills instead of 129 and none)
        int mine = __builtin_amdgcn_readfirstlane(flags[cwl]);
        int any = __builtin_amdgcn_readfirstlane(flags[G::NCWG]);
        if (CRC && any != 0) {
            // a codeword whose CRC holds is done although one of its parity checks fails; then "is anybody left" is asked again
            if (!done && mine != 0 && __builtin_amdgcn_readfirstlane((int)crc_holds(crc_slots))) mine = 0;
            __syncthreads();
            if (tid == 0) flags[G::NCWG] = 0;
            __syncthreads();
            if (!done && mine != 0) flags[G::NCWG] = 1;
            __syncthreads();
            any = __builtin_amdgcn_readfirstlane(flags[G::NCWG]);
        }
        if (!done && mine == 0) { done = true; my_iters = it; }
        return any == 0; // every codeword of the workgroup has converged
    };
    if constexpr (ETP) {
        // Two loops, so that everything live in the decoding loop is unconditional (as in PLAIN): a wave
        // whose codeword has converged (or does not exist) drops into the second loop and only keeps the
        // barrier count of its workgroup until every codeword of it is done.
        const float cap = (127.49f + a.beta) / a.alpha;
        DecArgs av = a; // see the fixed-iteration path
        av.beta = 8388608.0f - a.beta;
        asm volatile("" : "+v"(av.alpha), "+v"(av.beta));
        int it = 1;
        bool all_done = false;
        if (active) {
            GroupZ64<BG, ZC, 0, NL> g0;
            g0.template loads<false>(lds, R);
#if !NRLDPC_Z64_POSTBAR
            g0.template track<false>(st, cap);
#endif
            for (; it <= a.max_iter; ++it) {
                esign_lo = 0; esign_hi = 0;
                GroupZ64<BG, ZC, 0, NL> nx;
                pipeline_z64<BG, ZC, 0, true, NL>(g0, nx, st, lds, R, RA, RB, w, av, cap, esign_lo, esign_hi);
                if constexpr (RT) {
                    nx.template loads<false>(lds, R);
                    nx.template track<false>(st, cap);
                }
                g0 = nx;
                all_done = parity_pass(it);
                if (all_done || done) { ++it; break; }
            }
        }
        if (!all_done) {
            done = true;
            for (; it <= a.max_iter; ++it) {
                idle_barriers(1); // (see pipeline_z64: group 0 has no barrier here)
                if (parity_pass(it)) break;
            }
        }
    }
    if constexpr (!PLAIN && !ETP) for (int it = 1; it <= a.max_iter; ++it) {
        if (!done) { esign_lo = 0; esign_hi = 0; }
        static_for<G::ROWS>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            using LG = LayerGroups<BG>;
            if constexpr (LG::group_start(L) == L) { // L leads a barrier group
                constexpr int GE = LG::group_last(L);
                static_assert(PLAIN || ETP || !FULL, "the unpipelined loop is built for run-time layer counts only");
                {
                    const int nl = launder(a.n_layers);
                    if (L < nl) {
                        if (!done) {
                            if (GE < nl) {
                                group_z64<BG, ZC, L, GE, FULL, PLAIN>(st, lds, R, RA, RB, w, a, esign_lo, esign_hi, app_row);
                            } else { // the layer count cuts this group: its active layers one by one
                                static_for<GE - L>([&](auto ic) {
                                    constexpr int LL = L + decltype(ic)::value;
                                    if (LL < nl) group_z64<BG, ZC, LL, LL, FULL, PLAIN>(st, lds, R, RA, RB, w, a, esign_lo, esign_hi, app_row);
                                });
                            }
                        }
                        __syncthreads();
                    }
                }
            }
        });
        if (a.early_term && parity_pass(it)) break;
    }

    if (active) {
        if (a.iters && z == 0) a.iters[cw] = my_iters;
        uint8_t* hard = a.hard + (size_t)cw * ((size_t)G::KB * ZC);
        if (!app_row && (reinterpret_cast<uintptr_t>(a.hard) & 3) == 0) {
            // hard decisions, 4 ring positions per thread: one ds_read_b128 and one dword store per column quarter
            constexpr int QW = ZC / 4;
            const int qs = z / QW, qq = z - qs * QW;
            static_for<(G::KB + 3) / 4>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int c = 4 * k + qs;
                if (4 * k + 3 < G::KB || c < G::KB) {
                    const float4 v = *reinterpret_cast<const float4*>(lds + cwbase + G::GUARD + c * G::CS + 16 * qq);
                    const uint32_t bits = (v.x < 0.0f ? 1u : 0u) | (v.y < 0.0f ? 0x100u : 0u) | (v.z < 0.0f ? 0x10000u : 0u) |
                                          (v.w < 0.0f ? 0x1000000u : 0u);
                    *reinterpret_cast<uint32_t*>(hard + (size_t)c * ZC + 4 * qq) = bits;
                }
            });
        } else {
            const char* home = lds + cwbase + G::GUARD + 4 * z;
            static_for<G::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const float val = *reinterpret_cast<const float*>(home + c * G::CS);
                if (c < G::KB) hard[(size_t)c * ZC + z] = val < 0.0f ? 1 : 0;
                if (app_row) app_row[(size_t)c * ZC] = val * a.inv_scale;
            });
        }
    }
}

template <int BG, int ZC, int NCWG, bool FULL, bool PLAIN, bool ETP = false, int NL = BGT<BG>::ROWS, bool CRC = false>
static hipError_t launch_z64f(const DecArgs& a, hipStream_t s) {
    using G = Z64<BG, ZC, NCWG, NL>;
    auto k = nrldpc_decode_z64_kernel<BG, ZC, NCWG, FULL, PLAIN, ETP, NL, CRC>;
    constexpr size_t lds = G::lds_bytes();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {}; // per device: raising the dynamic-LDS limit is a slow host call, do it once
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const int grid = (a.batch + G::NCWG - 1) / G::NCWG;
    hipLaunchKernelGGL(k, dim3(grid), dim3(G::NCWG * G::TPC), lds, s, a);
    return hipGetLastError();
}

} // inline namespace NRLDPC_UNIT
} // namespace nrldpc
#include "nrldpc_decode_z64s.h" // the two-threads-per-row form of the same decoder
namespace nrldpc {
inline namespace NRLDPC_UNIT { // one name space per translation unit: see NRLDPC_UNIT in nrldpc_decode_z64.h

// Which form serves a (BG, Z, layer count): the measured choice, z64_split_default -- both forms timed on the MI355X for
// every pair at 25 fixed iterations and with the parity-check stop, in one session (tools/bench_forms.py,
// profiles/r03_forms.json), and for every BASELINE configuration (r03_forms_configs_*.json).  The split form wins where two of
// its workgroups (or more) fill a CU's wave slots evenly: BG1 -4...-19 % of the row form's time (Z = 384: -15 %, 288: -19 %),
// BG2 Z <= 128 and 208...256 -3...-18 % (-2...-26 % with the parity stop); it loses where a workgroup of 2 Z/B waves leaves
// slots empty (Z = 320: 10-wave workgroups, Z = 352: 16) or spreads unevenly over the 4 SIMDs (6-wave workgroups, Z = 144,
// 192), and on BG2 Z = 384, whose one-thread-per-row form already runs 6 waves per SIMD with four codewords per CU.
// A library built with -DNRLDPC_Z64_AB carries both forms for every pair and takes NRLDPC_SPLIT=0 / 1 from the
// environment (the A/B build of those measurements); the shipped build compiles only the form it uses.
template <int BG, int ZC, int NL> constexpr bool z64_split_default() {
#ifdef NRLDPC_Z64_SPLIT
    return NRLDPC_Z64_SPLIT != 0;
#endif
    // pruned layer counts: BG1 Z = 384 {5, 13, 24} -- with the parity-check stop +10 % / +21 % / +14 %, fixed-25 -3 % /
    // +6 % / +4 % (profiles/r03_forms_nl.txt); the BG2 counts of BASELINE configs[2] lose 10-19 % and stay with the row form
    // BG2 Z = 208 with 21 rows (the reference's default operating point): split 1.42 / 0.94 ms, row form 1.49 / 1.04, general kernel 1.71 / 1.27
    if (NL != BGT<BG>::ROWS) return (BG == 1 && ZC == 384) || (BG == 2 && ZC == 208);
    if (BG == 1)
        return ZC == 60 || ZC == 64 || ZC == 104 || ZC == 112 || ZC == 120 || ZC == 128 || ZC == 176 || ZC == 208 || ZC == 224 ||
               ZC == 240 || ZC == 256 || ZC == 288 || ZC == 384;
    return ZC == 52 || ZC == 60 || ZC == 64 || ZC == 88 || ZC == 96 || ZC == 104 || ZC == 112 || ZC == 120 || ZC == 128 ||
           ZC == 208 || ZC == 224 || ZC == 240 || ZC == 256;
}
#ifdef NRLDPC_Z64_AB
constexpr bool z64_ab = true;
#else
constexpr bool z64_ab = false;
#endif
inline int split_env() {
    static const int v = getenv("NRLDPC_SPLIT") ? atoi(getenv("NRLDPC_SPLIT")) : -1;
    return v;
}
template <int BG, int ZC, int NL> constexpr bool z64_has_split() {
    if constexpr (!Z64S<BG, ZC, NL>::usable()) return false;
    else return z64_ab || z64_split_default<BG, ZC, NL>();
}
template <int BG, int ZC, int NL> constexpr bool z64_has_row() { return z64_ab || !z64_has_split<BG, ZC, NL>(); }
template <int BG, int ZC, int NL> static bool use_split() {
    if constexpr (!z64_has_split<BG, ZC, NL>()) return false;
    else if constexpr (!z64_has_row<BG, ZC, NL>()) return true;
    else {
        const int e = split_env();
        return e < 0 ? z64_split_default<BG, ZC, NL>() : e != 0;
    }
}

// the pipelined pair (fixed iteration count / early termination) for a compile-time pruned layer count: its own
// translation unit (nrldpc_decode_z64_inst.hip with -DNRLDPC_Z64_NL=<count>)
template <int BG, int ZC, int NCWG, int NL> static hipError_t launch_z64_pruned(const DecArgs& a, hipStream_t s) {
    if constexpr (z64_has_split<BG, ZC, NL>()) {
        if (use_split<BG, ZC, NL>()) return a.early_term ? launch_z64s<BG, ZC, true, NL>(a, s) : launch_z64s<BG, ZC, false, NL>(a, s);
    }
    if constexpr (z64_has_row<BG, ZC, NL>()) {
        if (a.early_term) return launch_z64f<BG, ZC, NCWG, true, false, true, NL>(a, s);
        return launch_z64f<BG, ZC, NCWG, true, true, false, NL>(a, s);
    }
    return hipErrorUnknown; // not reached: one of the two forms exists
}

template <int BG, int ZC, int NCWG> static hipError_t launch_z64(const DecArgs& a, hipStream_t s) {
    constexpr int ROWS = BGT<BG>::ROWS;
    if (a.crc_bits && !a.app) { // CRC-aided stop (early_term = 2): the parity-stop twins with the CRC compiled in, all rows or a run-time prefix
        if constexpr (z64_has_split<BG, ZC, ROWS>()) {
            if (use_split<BG, ZC, ROWS>())
                return a.n_layers == ROWS ? launch_z64s<BG, ZC, true, ROWS, true>(a, s) : launch_z64s<BG, ZC, true, NL_RT, true>(a, s);
        }
        if constexpr (z64_has_row<BG, ZC, ROWS>()) {
            return a.n_layers == ROWS ? launch_z64f<BG, ZC, z64_ncwg_et<BG, ZC>(), true, false, true, ROWS, true>(a, s)
                                      : launch_z64f<BG, ZC, z64_ncwg_et<BG, ZC>(), true, false, true, NL_RT, true>(a, s);
        }
    }
    if constexpr (z64_has_split<BG, ZC, ROWS>()) {
        if (a.n_layers == ROWS && !a.app && use_split<BG, ZC, ROWS>())
            return a.early_term ? launch_z64s<BG, ZC, true>(a, s) : launch_z64s<BG, ZC, false>(a, s);
    }
    static const bool no_pruned = getenv("NRLDPC_NO_PRUNED_PIPELINE") != nullptr; // A/B against the general kernel
    if (!a.app && !no_pruned) {
        // layer counts of the rate-matching points BASELINE.json names have pipelined builds of their own
#define NRLDPC_Z64_NL_CASE(bg, z, nl) \
        if constexpr (BG == bg && ZC == z) if (a.n_layers == nl) return launch_decode_z64_##bg##_##z##_nl##nl(a, s);
        NRLDPC_Z64_NL_LIST(NRLDPC_Z64_NL_CASE)
#undef NRLDPC_Z64_NL_CASE
    }
    // every other pruned layer count: the same pipelined / split kernels with the layer count as a run-time prefix of the
    // all-rows tables (NL_RT); NRLDPC_NO_RT=1 sends them to the general kernel instead (A/B)
    static const bool no_rt = getenv("NRLDPC_NO_RT") != nullptr;
    if (a.n_layers != ROWS && !a.app && !no_rt) {
        if constexpr (z64_has_split<BG, ZC, ROWS>()) {
            if (use_split<BG, ZC, ROWS>()) return a.early_term ? launch_z64s<BG, ZC, true, NL_RT>(a, s) : launch_z64s<BG, ZC, false, NL_RT>(a, s);
        }
        if constexpr (z64_has_row<BG, ZC, ROWS>()) {
            if (a.early_term) return launch_z64f<BG, ZC, z64_ncwg_et<BG, ZC>(), true, false, true, NL_RT>(a, s);
            return launch_z64f<BG, ZC, NCWG, true, true, false, NL_RT>(a, s);
        }
    }
    // soft output (a test / debug feature) and what is left of the pruned layer counts: the unpipelined general kernel
    if (a.n_layers != ROWS || a.app) return launch_z64f<BG, ZC, NCWG, false, false>(a, s);
    if constexpr (z64_has_row<BG, ZC, ROWS>()) {
        if (a.early_term) return launch_z64f<BG, ZC, z64_ncwg_et<BG, ZC>(), true, false, true>(a, s);
        return launch_z64f<BG, ZC, NCWG, true, true>(a, s);
    }
    return hipErrorUnknown; // not reached
}

} // inline namespace NRLDPC_UNIT
} // namespace nrldpc
#endif
