// nrldpc_encode.hip -- systematic NR LDPC encoder for gfx950.
//
// Replaces step(obj.hLDPCEncoder, c) (NRLDPCEncoder.m:158; comm.LDPCEncoder built at :49 from the H of
// NRLDPC.m:438-440).  The systematic codeword [c; w] with H*[c; w] = 0 is unique, so this kernel is
// bit-identical to the toolbox encoder by construction; tests check H*cw = 0 and equality with the
// oracle's encoder.
//
// NWC waves per codeword -- one (four codewords per workgroup, no workgroup barrier), or FOUR for BG1 from Z = 128 on (round 6: a
// workgroup per codeword, every phase below a quarter as long; measured 5-8 % faster there and slower elsewhere: launch_encode) --,
// GF(2) arithmetic on bit-packed columns (32 check rows per XOR):
//   1. the K systematic bytes are copied to the output;
//   2. every column the parity equations read is packed with __ballot into a *doubled* bit ring
//      (bits 0 .. 2Z+31 of the periodic extension), so that the 32 bits  x[(32m + P + t) mod Z], t = 0..31,
//      of a circulant edge are one unaligned 32-bit window: two LDS words + v_alignbit, for any Z;
//   3. work items are (base row, word m) pairs spread over the lanes: the 4 core rows give lambda_i, the
//      dual-diagonal core is solved in the byte domain (4 columns, substitution order derived on the host
//      from the table), re-packed, and the 42/38 extension rows are plain XORs of windows;
//   4. parity words are expanded back to one byte per bit on the way out (dword stores when 4 | Z).
// NWC = 1: no workgroup barrier after the table load, waves are independent; NWC = 4: the phases are separated by workgroup barriers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "nrldpc_kernels.h"
#include "nrldpc_wave.h"

namespace nrldpc {

__device__ __forceinline__ int rotz(int z, int p, int Z) {
    int v = z + p;
    return v >= Z ? v - Z : v;
}

struct EncLayout { // LDS words; shared table first, then one region per wave
    int W, DW, tab, D, lam, PP, xc, wave_words;
    __host__ __device__ EncLayout(int Z, int kb, int nrows, int nnz) {
        W = (Z + 31) >> 5;                    // packed words per column
        DW = 2 * ((2 * Z + 32 + 63) >> 6);    // words of a doubled ring (whole ballots)
        tab = ((nnz + nrows + 1 + 3) & ~3);   // table: nnz edge words + nrows+1 row pointers
        D = 0;
        lam = D + (kb + 4) * DW;
        PP = lam + 4 * W;
        xc = PP + (nrows - 4) * W;
        wave_words = ((xc + Z + 2 + 3) & ~3); // xc: 4*Z bytes + slack
    }
};

template <int NWC> __global__ __launch_bounds__(256) void nrldpc_encode_kernel(const EncArgs a) {
    static_assert(NWC == 1 || NWC == 4, "a codeword is served by one wave or by the whole workgroup");
    constexpr int GT = 64 * NWC; // threads that serve one codeword
    // what separates two phases of a codeword: its own wave's LDS order, or the workgroup's barrier
    auto group_sync = [] {
        if constexpr (NWC == 1) wave_lds_sync();
        else __syncthreads();
    };
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Z = a.Z, kb = a.kb, nrows = a.nrows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const EncLayout L(Z, kb, nrows, a.nnz);
    const int W = L.W, DW = L.DW;
    uint32_t* tw = reinterpret_cast<uint32_t*>(lds);      // col | shift << 8 per edge
    uint32_t* rp = tw + a.nnz;                            // row pointers
    for (int e = threadIdx.x; e < a.nnz; e += blockDim.x) tw[e] = (uint32_t)a.col[e] | ((uint32_t)a.shift[e] << 8);
    for (int i = threadIdx.x; i <= nrows; i += blockDim.x) rp[i] = a.row_ptr[i];
    __syncthreads();
    const int slot = wave / NWC, wv = wave % NWC; // codeword of the workgroup, wave within its group
    const int gl = wv * 64 + lane;                // thread within the group
    const int cw = blockIdx.x * (nw / NWC) + slot;
    if (cw >= a.batch) return;                    // (NWC = 4: the whole workgroup, together)
    uint32_t* wbase = tw + L.tab + (size_t)slot * L.wave_words;
    uint32_t* D = wbase + L.D;
    uint32_t* lamP = wbase + L.lam;
    uint32_t* PP = wbase + L.PP;
    uint8_t* xc = reinterpret_cast<uint8_t*>(wbase + L.xc);
    const uint8_t* info = a.info + (size_t)cw * kb * Z;
    uint8_t* out = a.cw + (size_t)cw * a.ncols * Z;

    // 1. systematic part straight to the output, widest copies both alignments allow
    {
        const int n = kb * Z;
        const uintptr_t al = reinterpret_cast<uintptr_t>(info) | reinterpret_cast<uintptr_t>(out);
        int done = 0;
        if ((al & 15) == 0) {
            for (int i = gl; i < (n >> 4); i += GT) {
                uint4 v = reinterpret_cast<const uint4*>(info)[i];
                v.x &= 0x01010101u; v.y &= 0x01010101u; v.z &= 0x01010101u; v.w &= 0x01010101u;
                reinterpret_cast<uint4*>(out)[i] = v;
            }
            done = n & ~15;
        } else if ((al & 3) == 0) {
            for (int i = gl; i < (n >> 2); i += GT)
                reinterpret_cast<uint32_t*>(out)[i] = reinterpret_cast<const uint32_t*>(info)[i] & 0x01010101u;
            done = n & ~3;
        }
        for (int i = done + gl; i < n; i += GT) out[i] = info[i] & 1u;
    }

    // 2. doubled bit rings of `ncol` byte columns at src -> D[c0 ..]; the byte reads of a batch of ballots are
    // issued together (global loads for the systematic part, LDS reads for the core parity)
    const int idx0 = lane % Z, step = 64 % Z;
    auto pack_b = [&](auto src, int c0, int ncol, auto batch) {
        constexpr int PB = decltype(batch)::value;
        for (int c = wv; c < ncol; c += NWC) { // a ballot is a wave's: whole columns per wave
            int idx = idx0;
            for (int w0 = 0; w0 < DW / 2; w0 += PB) {
                uint8_t b[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    b[k] = src[c * Z + idx];
                    idx += step;
                    if (idx >= Z) idx -= Z;
                }
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const unsigned long long m = __ballot(b[k] & 1u);
                    if (w0 + k < DW / 2 && lane < 2) D[(c0 + c) * DW + 2 * (w0 + k) + lane] = (uint32_t)(m >> (32 * lane));
                }
            }
        }
    };
    auto pack = [&](auto src, int c0, int ncol) { // batch size: 2 ballots cover Z <= 48, 7 + 6 cover Z = 384
        if (DW <= 4) pack_b(src, c0, ncol, std::integral_constant<int, 2>{});
        else pack_b(src, c0, ncol, std::integral_constant<int, 7>{});
    };
    // XOR over the edges of base row i with column < col_limit of the 32-bit windows starting at ring bit 32m;
    // edge words and windows are read in batches so that their LDS latencies overlap
    constexpr int EB = 5;
    auto row_word = [&](int i, int m, int col_limit) {
        uint32_t acc = 0;
        const int e1 = rp[i + 1];
        for (int e0 = rp[i]; e0 < e1; e0 += EB) {
            uint32_t t[EB], lo[EB], hi[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) t[k] = tw[e0 + k < e1 ? e0 + k : e1 - 1];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const int s = 32 * m + (int)(t[k] >> 8);
                const uint32_t* d = D + (t[k] & 0xff) * DW + (s >> 5);
                const bool on = (e0 + k < e1) && (int)(t[k] & 0xff) < col_limit;
                lo[k] = on ? d[0] : 0u;
                hi[k] = on ? d[1] : 0u;
                t[k] = s & 31;
            }
#pragma unroll
            for (int k = 0; k < EB; ++k) acc ^= __builtin_amdgcn_alignbit(hi[k], lo[k], t[k]);
        }
        return acc;
    };
    auto bit_of = [&](const uint32_t* P, int z) { return (P[z >> 5] >> (z & 31)) & 1u; };

    // Z a multiple of 32 (every large lifting size): a ring word is 32 consecutive bytes of a column, so a lane builds
    // whole words from two 16-byte loads -- ((w & 0x01010101) * 0x01020408) >> 24 squeezes 4 byte-bits into a nibble -- instead
    // of one byte load and one ballot per bit: 264 word items per codeword at Z = 384 against 286 byte loads + ballots per lane.
    const bool words32 = (Z & 31) == 0;
    auto squeeze = [](uint32_t w) { return ((w & 0x01010101u) * 0x01020408u) >> 24; };
    auto put_word = [&](int c, int m, uint32_t v) { // ring word m of column c into every slot of the doubled ring it fills
        for (int w = m; w < DW; w += W) D[c * DW + w] = v;
    };
    if (words32 && (reinterpret_cast<uintptr_t>(info) & 15) == 0) {
        for (int it = gl; it < kb * W; it += GT) {
            const int c = it / W, m = it - c * W;
            const uint4* s4 = reinterpret_cast<const uint4*>(info + (size_t)c * Z + 32 * m);
            const uint4 lo = s4[0], hi = s4[1];
            put_word(c, m, squeeze(lo.x) | squeeze(lo.y) << 4 | squeeze(lo.z) << 8 | squeeze(lo.w) << 12 |
                               squeeze(hi.x) << 16 | squeeze(hi.y) << 20 | squeeze(hi.z) << 24 | squeeze(hi.w) << 28);
        }
    } else {
        pack(info, 0, kb);
    }
    group_sync();
    // 3a. lambda_i = systematic part of core row i
    for (int it = gl; it < 4 * W; it += GT) {
        const int i = it / W, m = it - i * W;
        lamP[it] = row_word(i, m, kb);
    }
    group_sync();
    // 3b. dual-diagonal core in the byte domain: the sum of the four rows isolates p0, the other three
    // blocks follow by substitution
    for (int z = gl; z < Z; z += GT) {
        const uint32_t tot = bit_of(lamP, z) ^ bit_of(lamP + W, z) ^ bit_of(lamP + 2 * W, z) ^ bit_of(lamP + 3 * W, z);
        xc[rotz(z, a.p0_shift, Z)] = (uint8_t)tot;
    }
    group_sync();
    for (int st = 0; st < 3; ++st) {
        const int i = a.step_row[st], u = a.step_col[st], nk = a.step_nk[st];
        for (int z = gl; z < Z; z += GT) {
            uint32_t s = bit_of(lamP + i * W, z);
            for (int k = 0; k < nk; ++k) s ^= xc[a.step_kcol[st][k] * Z + rotz(z, a.step_kshift[st][k], Z)];
            xc[u * Z + rotz(z, a.step_shift[st], Z)] = (uint8_t)s;
        }
        group_sync();
    }
    if constexpr (NWC == 1) store_row(out + (size_t)kb * Z, xc, 4 * Z);
    else store_row(out + (size_t)(kb + wv) * Z, xc + wv * Z, Z); // one core-parity column per wave
    if (words32) { // xc starts on a dword boundary
        const uint32_t* x32 = reinterpret_cast<const uint32_t*>(xc);
        for (int it = gl; it < 4 * W; it += GT) {
            const int c = it / W, m = it - c * W;
            const uint32_t* q = x32 + (c * Z + 32 * m) / 4;
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v |= squeeze(q[k]) << (4 * k);
            put_word(kb + c, m, v);
        }
    } else {
        pack(xc, kb, 4);
    }
    group_sync();
    // 3c. extension rows
    const int next = nrows - 4;
    for (int it = gl; it < next * W; it += GT) {
        const int i = it / W, m = it - i * W;
        PP[it] = row_word(4 + i, m, kb + 4);
    }
    group_sync();
    // 4. expand the extension parity to bytes
    uint8_t* ext = out + (size_t)(kb + 4) * Z;
    if ((Z & 3) == 0 && (reinterpret_cast<uintptr_t>(ext) & 3) == 0) {
        const int o0 = 4 * gl;
        int i = o0 / Z, z = o0 - i * Z;
        const int qi = (4 * GT) / Z, qz = 4 * GT - qi * Z;
        for (int o = o0; o < next * Z; o += 4 * GT) {
            const uint32_t nib = (PP[i * W + (z >> 5)] >> (z & 31)) & 0xFu;
            *reinterpret_cast<uint32_t*>(ext + o) = (nib * 0x00204081u) & 0x01010101u;
            i += qi; z += qz;
            if (z >= Z) { z -= Z; ++i; }
        }
    } else {
        int i = gl / Z, z = gl - i * Z;
        const int qi = GT / Z, qz = GT - qi * Z;
        for (int o = gl; o < next * Z; o += GT) {
            ext[o] = (uint8_t)bit_of(PP + i * W, z);
            i += qi; z += qz;
            if (z >= Z) { z -= Z; ++i; }
        }
    }
}

hipError_t launch_encode(const EncArgs& a, hipStream_t stream) {
    const EncLayout L(a.Z, a.kb, a.nrows, a.nnz);
    const int nw = 4; // waves per workgroup
    // A workgroup per codeword for BG1 from Z = 128 on, four codewords per workgroup (a wave each) otherwise -- as measured, 4096 transport
    // blocks, one session (profiles/r06_encode_waves_per_codeword.txt): BG1 Z = 384 0.0478 -> 0.0440 ms (0.37 -> 0.40 of 8 TB/s), BG1 Z = 320
    // C = 4 0.0402 -> 0.0380, but BG2 Z = 384 0.0322 -> 0.0335 and the kilobyte codewords of Z = 20 2.1 x slower.  So one wave walking its
    // codeword alone was NOT what holds the encoder at 0.37: a workgroup per codeword buys 5-8 % where the codeword is largest.
    // NRLDPC_ENC_NWC=1/4 forces either (A/B)
    static const int env_nwc = getenv("NRLDPC_ENC_NWC") ? atoi(getenv("NRLDPC_ENC_NWC")) : 0;
    const int nwc = env_nwc == 1 || env_nwc == 4 ? env_nwc : (a.kb == 22 && a.Z >= 128 ? 4 : 1);
    const int per_wg = nw / nwc;
    const size_t lds = 4 * ((size_t)L.tab + (size_t)per_wg * L.wave_words);
    const dim3 grid((a.batch + per_wg - 1) / per_wg);
    if (nwc == 4) hipLaunchKernelGGL(nrldpc_encode_kernel<4>, grid, dim3(64 * nw), lds, stream, a);
    else hipLaunchKernelGGL(nrldpc_encode_kernel<1>, grid, dim3(64 * nw), lds, stream, a);
    return hipGetLastError();
}

} // namespace nrldpc
