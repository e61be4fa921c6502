// nrldpc_sched.h -- host-side construction of the decoder's per-(BG, Z, n_layers) layer schedule.
//
// The reference materialises a sparse H with get_pcm.m:1-11 from get_3gpp_base_graph.m's table
// (NRLDPC.m:433-440).  The MI355X decoder never builds H; it needs, per base-graph edge, only
// rotation amounts.  Check (l, z) touches variable (c, (z + P_lc) mod Z)  (get_pcm.m:8).
//
// LDS convention: the a-posteriori block of core column c sits at ring positions 0..Z-1 in natural
// order; check row z reads AND writes ring position (z + P_e) mod Z for edge e.  Within one layer
// every LDS word is therefore touched by exactly one thread (no intra-layer hazard, one barrier per
// layer).  A "write-rotated" variant that reads at position z and writes at (z + D_e) mod Z saves
// the live address registers but lets one wave's pass-2 write race another wave's pass-1 read of the
// same layer, so it is not used.
#ifndef NRLDPC_SCHED_H
#define NRLDPC_SCHED_H

#include <cstdint>
#include <vector>

#include "nr_bg_tables.h"

#include "nrldpc_kernels.h"

namespace nrldpc {

struct BaseGraph {
    int bg, nrows, ncols, kb, nnz;
    const uint16_t* row_ptr;
    const uint8_t* col;
    const uint16_t (*shift)[NR_BG1_NNZ];  // only valid for bg == 1
    const uint16_t (*shift2)[NR_BG2_NNZ]; // only valid for bg == 2
    int raw_shift(int ils, int e) const { return bg == 1 ? shift[ils][e] : shift2[ils][e]; }
};

inline bool base_graph(int bg, BaseGraph* g) {
    if (bg == 1) {
        *g = BaseGraph{1, NR_BG1_ROWS, NR_BG1_COLS, 22, NR_BG1_NNZ, nr_bg1_row_ptr, nr_bg1_col, nr_bg1_shift, nullptr};
        return true;
    }
    if (bg == 2) {
        *g = BaseGraph{2, NR_BG2_ROWS, NR_BG2_COLS, 10, NR_BG2_NNZ, nr_bg2_row_ptr, nr_bg2_col, nullptr, nr_bg2_shift};
        return true;
    }
    return false;
}

// get_3gpp_set_index.m:5-11
inline int set_index(int Z) {
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9 && nr_lifting_sets[s][k]; ++k)
            if (nr_lifting_sets[s][k] == Z) return s;
    return -1;
}

// get_3gpp_lifting_size.m:5-16
inline int lifting_size(int K_b, int K_prime) {
    int best = -1;
    for (int s = 0; s < 8; ++s)
        for (int k = 0; k < 9 && nr_lifting_sets[s][k]; ++k) {
            int z = nr_lifting_sets[s][k];
            if (K_b * z >= K_prime && (best < 0 || z < best)) best = z;
        }
    return best;
}

// Check-node rule used when the caller does not give one (nrldpc_cfg.alpha == 0): message magnitude =
// max(alpha*min - beta, 0), beta in LLR units.  The reference defines no such parameters (its decoder is
// sum-product, NRLDPCDecoder.m:120); these are the values whose block-error rate sits closest to flooding
// sum-product at equal iteration caps, measured per base graph and fraction of rows decoded
// (tools/alg_search).
inline void default_rule(int bg, int n_layers, float* alpha, float* beta) {
    const int rows = bg == 1 ? NR_BG1_ROWS : NR_BG2_ROWS;
    const double frac = (double)n_layers / rows;
    // alpha = 7/8 everywhere; the offset grows with the share of degree-1 extension rows in the decoded graph.
    // BG1 Z=384 R=1/3, 25 iterations, BLER 1e-2: plain alpha = 0.625 (round 1) sits 0.20 dB from flooding sum-product
    // at the same cap, this rule 0.03 dB; at high rates (few extension rows) it beats sum-product at equal caps.
    *alpha = 0.875f;
    if (bg == 1) *beta = frac > 0.2 ? 0.375f : 0.25f;
    else *beta = frac > 0.25 ? 0.3125f : 0.25f;
}

struct Schedule {
    BaseGraph g;
    int Z = 0, ils = 0, n_layers = 0;
    int nc = 0;      // core columns kb+4 (kept in LDS)
    int ncp = 0;     // padded (odd) dword stride between ring positions
    int ncw = 1;     // codewords per workgroup
    int threads = 0; // workgroup size
    int sbw = 0;     // bytes between consecutive ring positions = ncw*ncp*4
    size_t lds_bytes = 0;
    // per base-graph edge (indexed by table edge id; extension-parity edges hold 0):
    std::vector<int32_t> shift;  // P_e = table shift mod Z
    std::vector<int32_t> rot;    // P_e * sbw  (ring rotation in bytes)
};

// Returns false for an unsupported (bg, Z, n_layers).
inline bool build_schedule(int bg, int Z, int n_layers, Schedule* s) {
    if (!base_graph(bg, &s->g)) return false;
    const BaseGraph& g = s->g;
    s->ils = set_index(Z);
    if (s->ils < 0) return false;
    if (n_layers == 0) n_layers = g.nrows;
    if (n_layers < 4 || n_layers > g.nrows) return false;
    s->Z = Z;
    s->n_layers = n_layers;
    s->nc = g.kb + 4;
    s->ncp = s->nc | 1;
    // Codewords per workgroup.  The run-time-Z kernel's registers are sized for 4 waves per SIMD (BG1, 128
    // VGPRs) / 6 (BG2, 80 VGPRs), i.e. 16 / 24 wave slots per CU; pick the ncw that keeps most of them doing
    // useful work: lanes used per wave x wave slots the resulting workgroups can fill (also bounded by LDS).
    // Consecutive ring positions of one codeword are ncw*ncp dwords apart in LDS and only an odd stride spreads
    // a wave's lanes over all banks: an even ncw above 4 means 8-way or worse conflicts (Z = 24 ran 3x slower),
    // so those are skipped.  Measured against the former "as many as fit 768 threads": +0..43 % (BG1).
    const int slots = 4 * ((bg == 1) ? NRLDPC_GEN_WPE_BG1 : NRLDPC_GEN_WPE_BG2), tmax = (bg == 1) ? NRLDPC_GEN_THREADS_BG1 : NRLDPC_GEN_THREADS_BG2;
    int ncw = 1;
    double best = -1.0;
    for (int n = 1; n * Z <= tmax || n == 1; ++n) {
        if (n > 4 && (n & 1) == 0) continue;
        const int waves = (n * Z + 63) / 64;
        const size_t lds = (size_t)Z * n * s->ncp * 4 + 4 * (size_t)(n + 1) + 16;
        int wgs = slots / waves;
        const int by_lds = (int)((160 * 1024) / lds);
        if (wgs > by_lds) wgs = by_lds;
        if (wgs < 1) wgs = 1;
        double score = (double)(n * Z) / (waves * 64.0) * (double)(waves * wgs > slots ? slots : waves * wgs) / slots;
        if (waves % 4) score *= (waves > 4 ? 0.85 : 0.97); // uneven spread over the 4 SIMDs (6-wave groups: -15 %)
        if (waves < 4) score *= 0.97; // per-workgroup overheads weigh more on tiny workgroups
        if (score >= best - 1e-9) { best = score; ncw = n; } // ties: the larger workgroup
        if (n * Z > tmax) break;
    }
    s->ncw = ncw;
    s->threads = ((ncw * Z + 63) / 64) * 64;
    s->sbw = ncw * s->ncp * 4;
    s->lds_bytes = (((size_t)Z * s->sbw + 4 * (size_t)(ncw + 1)) + 15) / 16 * 16; // + early-termination flags
    s->shift.assign(g.nnz, 0);
    s->rot.assign(g.nnz, 0);
    for (int e = 0; e < g.nnz; ++e) {
        s->shift[e] = g.raw_shift(s->ils, e) % Z;
        s->rot[e] = (g.col[e] < s->nc) ? s->shift[e] * s->sbw : 0;
    }
    return true;
}

} // namespace nrldpc
#endif
